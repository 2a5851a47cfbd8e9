// coalescer.cpp — request coalescing for single-query callers (nidx_gpu_vector_search_one).
//
// The reference serves every Search request on its own blocking thread with ONE query vector
// (nidx/src/searcher/shard_search.rs:139-153, nodereader.proto:402); a GPU wants batches.  Concurrent
// callers of nidx_gpu_vector_search_one are therefore merged: the first caller to arrive becomes the
// leader, waits a short window (or until the batch is full) for more requests with the same
// parameters, runs ONE batched search, and hands every caller its rows.  Results are identical to
// calling nidx_gpu_vector_search with a batch of one (the kernels are per-query deterministic).
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>

#include "host_common.h"
#include "vector_index.h"

namespace nidx {

struct OneRequest {
    const float *query;
    nidx_gpu_vector_search_params_t params;
    uint32_t *out_segment, *out_paragraph, *out_vector;
    float *out_score;
    uint32_t *out_count;
    int32_t rc = NIDX_OK;
    char error[512] = {0};
    bool done = false;
};

struct Coalescer {
    std::mutex mu;
    // Two condition variables, so that an arriving request wakes at most the leader.  With a single one every arrival woke every
    // parked caller (each re-takes the mutex to find nothing to do): 64 callers arriving behind a running batch are ~4 000 mutex
    // hand-overs, and the leader coming back from the GPU queued behind them for 25-75 ms (measured: the p99 of 64 callers).
    std::condition_variable cv_leader;  // the gathering leader waits here for arrivals
    std::condition_variable cv_done;    // everyone else waits here for a batch to be handed out
    std::deque<OneRequest *> pending;
    bool leader_active = false;
    uint64_t n_batches = 0, n_queries = 0;
    uint32_t window_us = 100, max_batch = 1024;
};

static bool same_params(const nidx_gpu_vector_search_params_t &a, const nidx_gpu_vector_search_params_t &b) {
    return a.k == b.k && a.with_duplicates == b.with_duplicates && a.method == b.method &&
           std::memcmp(&a.min_score, &b.min_score, 4) == 0;
}

int32_t VectorIndex::search_one(const float *query, const nidx_gpu_vector_search_params_t &p, uint32_t *out_segment,
                                uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count) {
    Coalescer &c = *coalescer;  // created with the handle (nidx_gpu_vector_open)
    {
        // staging for a full batch, once (not under the coalescer's lock: it takes the index's)
        uint32_t mb;
        {
            std::lock_guard<std::mutex> g(c.mu);
            mb = c.max_batch;
        }
        if (mb > reserved_nq || p.k > reserved_k) {
            const int32_t rc = reserve_search(mb, p.k);
            if (rc != NIDX_OK) return rc;
        }
    }
    OneRequest req{query, p, out_segment, out_paragraph, out_vector, out_score, out_count};
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t_in = std::chrono::steady_clock::now();
    auto t_lead = t_in, t_gathered = t_in, t_searched = t_in, t_posted = t_in;
    uint32_t led = 0;
    std::unique_lock<std::mutex> lk(c.mu);
    // everything that can fail for lack of memory happens before the request is visible to other callers: `req` lives on this
    // stack frame, and a leader that unwinds would strand its followers
    std::vector<OneRequest *> batch;
    batch.reserve(c.max_batch);
    c.pending.push_back(&req);
    if (c.leader_active) c.cv_leader.notify_one();
    while (!req.done) {
        if (c.leader_active) {
            c.cv_done.wait(lk);
            continue;
        }
        // become the leader: gather a batch of requests that share this request's parameters
        c.leader_active = true;
        t_lead = std::chrono::steady_clock::now();
        auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(c.window_us);
        while (c.pending.size() < c.max_batch && c.cv_leader.wait_until(lk, deadline) != std::cv_status::timeout) {
        }
        const nidx_gpu_vector_search_params_t lead = c.pending.front()->params;
        const size_t room = std::min<size_t>(c.max_batch, batch.capacity());
        batch.clear();
        for (auto it = c.pending.begin(); it != c.pending.end() && batch.size() < room;) {
            if (same_params((*it)->params, lead)) {
                batch.push_back(*it);
                it = c.pending.erase(it);
            } else {
                ++it;
            }
        }
        lk.unlock();
        t_gathered = std::chrono::steady_clock::now();
        const uint32_t B = (uint32_t)batch.size(), k = lead.k, d = cfg.dimension;
        led = B;
        const size_t kk = std::max<uint32_t>(k, 1);
        std::vector<uint32_t> seg, par, vec, cnt;
        std::vector<float> q, sc;
        int32_t rc;
        // the followers of this batch are parked on the condition variable: whatever happens here, they must be released
        try {
            q.resize((size_t)B * d);
            for (uint32_t i = 0; i < B; i++) std::memcpy(&q[(size_t)i * d], batch[i]->query, (size_t)d * 4);
            seg.resize(B * kk), par.resize(B * kk), vec.resize(B * kk), cnt.resize(B), sc.resize(B * kk);
            rc = search_host(q.data(), B, lead, nullptr, nullptr, seg.data(), par.data(), vec.data(), sc.data(), cnt.data(), nullptr,
                             nullptr);
        } catch (...) {
            rc = abi_exception();
        }
        char err[512] = {0};
        if (rc != NIDX_OK) nidx_gpu_last_error(err, sizeof(err));
        t_searched = std::chrono::steady_clock::now();
        lk.lock();
        for (uint32_t i = 0; i < B; i++) {
            OneRequest *r = batch[i];
            r->rc = rc;
            if (rc != NIDX_OK) std::memcpy(r->error, err, sizeof(err));
            else {
                *r->out_count = cnt[i];
                for (uint32_t j = 0; j < cnt[i]; j++) {
                    if (r->out_segment) r->out_segment[j] = seg[i * kk + j];
                    if (r->out_paragraph) r->out_paragraph[j] = par[i * kk + j];
                    if (r->out_vector) r->out_vector[j] = vec[i * kk + j];
                    if (r->out_score) r->out_score[j] = sc[i * kk + j];
                }
            }
            r->done = true;
        }
        c.n_batches++;
        c.n_queries += B;
        c.leader_active = false;
        c.cv_done.notify_all();
        t_posted = std::chrono::steady_clock::now();
    }
    if (trace_slow_us() > 0) {
        const auto t_out = std::chrono::steady_clock::now();
        if (us(t_in, t_out) > trace_slow_us())
            fprintf(stderr, "[nidx_gpu slow search_one] total %.0f us: led a batch of %u (0 = follower only); until leader %.0f, gather %.0f, search %.0f, hand-out %.0f, after %.0f\n",
                    us(t_in, t_out), led, us(t_in, t_lead), us(t_lead, t_gathered), us(t_gathered, t_searched), us(t_searched, t_posted), us(t_posted, t_out));
    }
    if (req.rc != NIDX_OK) set_error("%s", req.error);
    return req.rc;
}

void VectorIndex::coalescer_stats(uint64_t &batches, uint64_t &queries) {
    batches = queries = 0;
    if (!coalescer) return;
    std::lock_guard<std::mutex> lk(coalescer->mu);
    batches = coalescer->n_batches;
    queries = coalescer->n_queries;
}

std::shared_ptr<Coalescer> make_coalescer() { return std::make_shared<Coalescer>(); }

void VectorIndex::coalescer_config(int32_t window_us, int32_t max_batch) {
    std::lock_guard<std::mutex> lk(coalescer->mu);
    if (window_us >= 0) coalescer->window_us = (uint32_t)window_us;
    if (max_batch > 0) coalescer->max_batch = (uint32_t)max_batch;
}

}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_vector_search_one(nidx_gpu_vector_index_t *index, const float *query, uint32_t query_dimension,
                                   const nidx_gpu_vector_search_params_t *params, uint32_t *out_segment,
                                   uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !query || !params || !out_count) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (query_dimension != idx->cfg.dimension)
        return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "Inconsistent dimensions. Index=%u Vector=%u", idx->cfg.dimension,
                    query_dimension);
    *out_count = 0;
    if (params->k == 0) return NIDX_OK;
    return idx->search_one(query, *params, out_segment, out_paragraph, out_vector, out_score, out_count);
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_coalescer_stats(nidx_gpu_vector_index_t *index, uint64_t *batches_out, uint64_t *queries_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !batches_out || !queries_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    idx->coalescer_stats(*batches_out, *queries_out);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_spill_stats(nidx_gpu_vector_index_t *index, uint64_t *queries_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !queries_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *queries_out = idx->spill_queries;
    return NIDX_OK;
} NIDX_ABI_CATCH

}  // extern "C"
