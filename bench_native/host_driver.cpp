// host_driver.cpp — the caller's side of bench.py's pipelined legs, in native code.
//
// The reference's host is Rust: one blocking thread per request around TextSearcher / ParagraphSearcher::search
// (nidx/src/searcher/shard_search.rs:139-153,176-248).  bench.py used to play that role with Python threads; with six of them the
// interpreter lock and the glue between two ctypes calls cost ~30 us per batch — more than the kernels of a batch — so the figure
// measured the interpreter.  This file is the same loop (submit / wait, `depth` tickets per thread, `threads` threads) written against
// include/nidx_gpu.h only: it is a CLIENT of the C ABI like the Rust shim of INTEGRATION.md, it is not part of the product library
// and nothing in nucliadb_amd/ uses it.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <deque>
#include <thread>
#include <vector>

#include "../include/nidx_gpu.h"

namespace {
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" {

// Every thread: submit batches (round robin over `batches`, all of shape clause_offsets / n_queries) and wait for the oldest once
// `depth` tickets are out, until `min_steps` batches in total AND `min_seconds` have passed.  Four untimed batches per thread first
// (the slots' buffers and streams exist before the clock starts).
// -> rc of the first failing call (0 = ok); elapsed_out seconds, batches_out, postings_out (sum of the per-query posting counts),
//    per_batch_postings_out: NULL or [cap] postings of each timed batch (n written = min(batches, cap)).
// submit_fn / wait_fn: nidx_gpu_bm25_search_submit / _wait of the library the caller has loaded (the driver links against nothing).
typedef int32_t (*bm25_submit_fn)(nidx_gpu_bm25_index_t *, const nidx_gpu_bm25_clause_t *, const uint64_t *, uint32_t, const nidx_gpu_bm25_search_options_t *, uint64_t *);
typedef int32_t (*bm25_wait_fn)(nidx_gpu_bm25_index_t *, uint64_t, uint64_t *, float *, uint32_t *, uint64_t *, uint64_t *);
int32_t nidx_bench_bm25_pipeline(bm25_submit_fn submit_fn, bm25_wait_fn wait_fn, nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *const *batches, uint32_t n_batches,
                                 const uint64_t *clause_offsets, uint32_t n_queries, const nidx_gpu_bm25_search_options_t *options,
                                 uint32_t threads, uint32_t depth, uint64_t min_steps, double min_seconds, double *elapsed_out,
                                 uint64_t *batches_out, double *postings_out, double *per_batch_postings_out, uint64_t cap) {
    const uint32_t k = options->k > 0 ? options->k : 1;
    std::atomic<int32_t> rc{0};
    std::atomic<uint32_t> ready{0};
    std::atomic<int> go{0};
    std::atomic<uint64_t> steps{0};
    std::atomic<uint64_t> n_logged{0};
    double t0 = 0.0;
    std::vector<double> t_end(threads, 0.0), posts(threads, 0.0);
    std::vector<uint64_t> done(threads, 0);
    auto worker = [&](uint32_t w) {
        std::vector<uint64_t> docaddr((size_t)n_queries * k), total(n_queries), post(n_queries);
        std::vector<float> score((size_t)n_queries * k);
        std::vector<uint32_t> count(n_queries);
        std::deque<uint64_t> pending;
        auto submit = [&](uint64_t i) -> bool {
            uint64_t t = 0;
            const int32_t r = submit_fn(index, batches[i % n_batches], clause_offsets, n_queries, options, &t);
            if (r != 0) { int32_t z = 0; rc.compare_exchange_strong(z, r); return false; }
            pending.push_back(t);
            return true;
        };
        auto wait_oldest = [&](double *p) -> bool {
            const uint64_t t = pending.front();
            pending.pop_front();
            const int32_t r = wait_fn(index, t, docaddr.data(), score.data(), count.data(), total.data(), post.data());
            if (r != 0) { int32_t z = 0; rc.compare_exchange_strong(z, r); return false; }
            if (p) {
                double s = 0.0;
                for (uint32_t q = 0; q < n_queries; q++) s += (double)post[q];
                *p = s;
            }
            return true;
        };
        bool ok = true;
        for (uint32_t i = 0; i < 4 && ok; i++) {
            ok = submit(w + (uint64_t)i * threads);
            if (ok && pending.size() >= depth) ok = wait_oldest(nullptr);
        }
        while (ok && !pending.empty()) ok = wait_oldest(nullptr);
        ready.fetch_add(1);
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        uint64_t n = 0;
        double sum = 0.0;
        auto log = [&](double p) {
            sum += p;
            const uint64_t at = n_logged.fetch_add(1);
            if (per_batch_postings_out && at < cap) per_batch_postings_out[at] = p;
        };
        while (ok && rc.load() == 0) {
            if (steps.load(std::memory_order_relaxed) >= min_steps && now_s() - t0 >= min_seconds) break;
            ok = submit(w + n * threads);
            steps.fetch_add(1, std::memory_order_relaxed);
            n++;
            if (ok && pending.size() >= depth) {
                double p = 0.0;
                ok = wait_oldest(&p);
                if (ok) log(p);
            }
        }
        while (ok && !pending.empty()) {
            double p = 0.0;
            ok = wait_oldest(&p);
            if (ok) log(p);
        }
        if (!ok)   // give the tickets back so the index can be closed
            while (!pending.empty()) { (void)wait_fn(index, pending.front(), docaddr.data(), score.data(), count.data(), total.data(), post.data()); pending.pop_front(); }
        t_end[w] = now_s();
        posts[w] = sum;
        done[w] = n;
    };
    std::vector<std::thread> ths;
    for (uint32_t w = 0; w < threads; w++) ths.emplace_back(worker, w);
    while (ready.load() < threads) std::this_thread::yield();
    t0 = now_s();
    go.store(1, std::memory_order_release);
    for (auto &t : ths) t.join();
    double end = t0, p = 0.0;
    uint64_t n = 0;
    for (uint32_t w = 0; w < threads; w++) {
        if (t_end[w] > end) end = t_end[w];
        p += posts[w];
        n += done[w];
    }
    if (elapsed_out) *elapsed_out = end - t0;
    if (batches_out) *batches_out = n;
    if (postings_out) *postings_out = p;
    return rc.load();
}

// The same loop around nidx_gpu_vector_search_submit / _wait with HOST query rows (what every caller of the reference hands over:
// VectorSearchRequest.vector, nidx_vector/src/request_types.rs:19-35): `threads` threads, `depth` tickets each, batches[i] = n_queries x
// dimension f32 rows in ordinary host memory.  Runs exactly `steps` batches in total after `warm` untimed ones per thread.
typedef int32_t (*vec_submit_fn)(nidx_gpu_vector_index_t *, const float *, uint32_t, uint32_t, const nidx_gpu_vector_search_params_t *, const uint64_t *const *, uint64_t *);
typedef int32_t (*vec_wait_fn)(nidx_gpu_vector_index_t *, uint64_t, uint32_t *, uint32_t *, uint32_t *, float *, uint32_t *, uint32_t *);
int32_t nidx_bench_vector_pipeline(vec_submit_fn submit_fn, vec_wait_fn wait_fn, nidx_gpu_vector_index_t *index, const float *const *batches,
                                   uint32_t n_batches, uint32_t n_queries, uint32_t dimension, const nidx_gpu_vector_search_params_t *params,
                                   uint32_t threads, uint32_t depth, uint32_t warm, uint64_t steps, double *elapsed_out, uint32_t *last_vec_out,
                                   float *last_score_out, uint32_t *last_count_out) {
    const uint32_t k = params->k > 0 ? params->k : 1;
    std::atomic<int32_t> rc{0};
    std::atomic<uint32_t> ready{0};
    std::atomic<int> go{0};
    std::atomic<uint64_t> next{0};
    std::vector<double> t_end(threads, 0.0);
    auto worker = [&](uint32_t w) {
        std::vector<uint32_t> vec((size_t)n_queries * k), count(n_queries);
        std::vector<float> score((size_t)n_queries * k);
        std::deque<uint64_t> pending;
        auto submit = [&](uint64_t i) -> bool {
            uint64_t t = 0;
            const int32_t r = submit_fn(index, batches[i % n_batches], n_queries, dimension, params, nullptr, &t);
            if (r != 0) { int32_t z = 0; rc.compare_exchange_strong(z, r); return false; }
            pending.push_back(t);
            return true;
        };
        auto wait_oldest = [&]() -> bool {
            const uint64_t t = pending.front();
            pending.pop_front();
            const int32_t r = wait_fn(index, t, nullptr, nullptr, vec.data(), score.data(), count.data(), nullptr);
            if (r != 0) { int32_t z = 0; rc.compare_exchange_strong(z, r); return false; }
            return true;
        };
        bool ok = true;
        for (uint32_t i = 0; i < warm && ok; i++) {
            ok = submit(w + (uint64_t)i * threads);
            if (ok && pending.size() >= depth) ok = wait_oldest();
        }
        while (ok && !pending.empty()) ok = wait_oldest();
        ready.fetch_add(1);
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        while (ok && rc.load() == 0) {
            const uint64_t i = next.fetch_add(1);
            if (i >= steps) break;
            ok = submit(i);
            if (ok && pending.size() >= depth) ok = wait_oldest();
        }
        while (ok && !pending.empty()) ok = wait_oldest();
        if (!ok)
            while (!pending.empty()) { (void)wait_fn(index, pending.front(), nullptr, nullptr, vec.data(), score.data(), count.data(), nullptr); pending.pop_front(); }
        t_end[w] = now_s();
        if (w == 0 && ok && last_vec_out) {   // one more batch (batch 0), handed back for the caller's comparison with the device-resident entry
            if (submit(0) && wait_oldest()) {
                memcpy(last_vec_out, vec.data(), vec.size() * 4);
                memcpy(last_score_out, score.data(), score.size() * 4);
                memcpy(last_count_out, count.data(), count.size() * 4);
            }
        }
    };
    std::vector<std::thread> ths;
    for (uint32_t w = 0; w < threads; w++) ths.emplace_back(worker, w);
    while (ready.load() < threads) std::this_thread::yield();
    const double t0 = now_s();
    go.store(1, std::memory_order_release);
    for (auto &t : ths) t.join();
    double end = t0;
    for (uint32_t w = 0; w < threads; w++) end = t_end[w] > end ? t_end[w] : end;
    if (elapsed_out) *elapsed_out = end - t0;
    return rc.load();
}

}  // extern "C"
