// coalescer.cpp — request coalescing for single-query callers (nidx_gpu_vector_search_one).
//
// The reference serves every Search request on its own blocking thread with ONE query vector
// (nidx/src/searcher/shard_search.rs:139-153, nodereader.proto:402); a GPU wants batches.  Concurrent
// callers of nidx_gpu_vector_search_one are therefore merged: one caller at a time is the GATHERER — it waits until its
// batch is due, takes the pending requests that share its parameters, hands the gatherer role to the next pending caller
// and only then runs its batch through the serving pipeline (serving.cpp: its own stream and staging) — so the next batch
// gathers, and up to `max_in_flight` batches run, while this one is on the device.  A batch is due when its window has passed
// (or it is full) AND fewer than max_in_flight batches are running: under load a batch closes when an earlier launch finishes
// and its size follows the arrival rate.  Results are identical to calling nidx_gpu_vector_search with a batch of one
// (the kernels are per-query deterministic).
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
    // every parked caller sleeps on its own condition variable: a finished batch wakes exactly its members, an arrival wakes at
    // most the gatherer (one shared variable made 64 arrivals behind a running batch ~4 000 mutex hand-overs: 25-75 ms stalls)
    std::condition_variable cv;
};

struct Coalescer {
    std::mutex mu;
    std::condition_variable cv_gather;  // the gatherer waits here for arrivals, for its window and for a free slot
    std::deque<OneRequest *> pending;
    bool gatherer_active = false;
    uint32_t in_flight = 0;
    uint64_t n_batches = 0, n_queries = 0;
    uint32_t window_us = 50, max_batch = 1024, max_in_flight = 4;
};

static bool same_params(const nidx_gpu_vector_search_params_t &a, const nidx_gpu_vector_search_params_t &b) {
    return a.k == b.k && a.with_duplicates == b.with_duplicates && a.method == b.method &&
           std::memcmp(&a.min_score, &b.min_score, 4) == 0;
}

int32_t VectorIndex::search_one(const float *query, const nidx_gpu_vector_search_params_t &p, uint32_t *out_segment,
                                uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count) {
    Coalescer &c = *coalescer;  // created with the handle (nidx_gpu_vector_open)
    if (p.k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "result_per_page > %d is not supported (got %u)", NIDX_K_MAX, p.k);
    if (p.method < 0 || p.method > 6) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown search method %d", p.method);
    OneRequest req{query, p, out_segment, out_paragraph, out_vector, out_score, out_count};
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t_in = std::chrono::steady_clock::now();
    auto t_lead = t_in, t_gathered = t_in, t_searched = t_in, t_posted = t_in;
    uint32_t led = 0;
    std::unique_lock<std::mutex> lk(c.mu);
    // everything that can fail for lack of memory happens before the request is visible to other callers (`req` lives on this
    // stack frame: a caller that unwinds while others hold a pointer to it would strand them) or inside the try block below
    std::vector<OneRequest *> batch;
    batch.reserve(c.max_batch);
    c.pending.push_back(&req);
    if (c.gatherer_active) c.cv_gather.notify_one();
    while (!req.done) {
        if (c.gatherer_active) {   // parked until this request is done or this caller is asked to gather
            req.cv.wait(lk);
            continue;
        }
        // ---- gatherer: wait until the batch is due, take it, pass the role on -------------------------------------------
        c.gatherer_active = true;
        t_lead = std::chrono::steady_clock::now();
        const auto deadline = t_lead + std::chrono::microseconds(c.window_us);
        for (;;) {
            const bool full = c.pending.size() >= c.max_batch;
            const bool due = full || std::chrono::steady_clock::now() >= deadline;
            if (due && c.in_flight < c.max_in_flight) break;
            if (due) c.cv_gather.wait(lk);            // only a finishing batch can make it runnable
            else c.cv_gather.wait_until(lk, deadline);
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
        c.in_flight++;
        c.gatherer_active = false;
        // whoever is first in line (this caller runs its batch now) gathers the next one meanwhile; a woken caller that finds
        // the role taken by a new arrival simply parks again
        for (OneRequest *r : c.pending)
            if (r != &req) {
                r->cv.notify_one();
                break;
            }
        lk.unlock();
        t_gathered = std::chrono::steady_clock::now();
        const uint32_t B = (uint32_t)batch.size(), k = lead.k, d = cfg.dimension;
        led = B;
        const size_t kk = std::max<uint32_t>(k, 1);
        std::vector<uint32_t> seg, par, vec, cnt;
        std::vector<float> q, sc;
        int32_t rc;
        // the members of this batch are parked: whatever happens here, they must be released
        try {
            q.resize((size_t)B * d);
            for (uint32_t i = 0; i < B; i++) std::memcpy(&q[(size_t)i * d], batch[i]->query, (size_t)d * 4);
            seg.resize(B * kk), par.resize(B * kk), vec.resize(B * kk), cnt.resize(B), sc.resize(B * kk);
            uint64_t ticket = 0;
            rc = pipeline_submit(q.data(), B, lead, nullptr, /*blocking=*/true, &ticket);
            if (rc == NIDX_OK) rc = pipeline_wait(ticket, seg.data(), par.data(), vec.data(), sc.data(), cnt.data(), nullptr);
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
            if (r != &req) r->cv.notify_one();
        }
        c.n_batches++;
        c.n_queries += B;
        c.in_flight--;
        c.cv_gather.notify_one();   // a gatherer waiting for a free slot
        // (this caller's own request may have had other parameters than the batch it led: it is still pending then, and the
        // loop either parks it behind the current gatherer or makes it the gatherer again)
        t_posted = std::chrono::steady_clock::now();
    }
    if (trace_slow_us() > 0) {
        const auto t_out = std::chrono::steady_clock::now();
        if (us(t_in, t_out) > trace_slow_us())
            fprintf(stderr, "[nidx_gpu slow search_one] total %.0f us: led a batch of %u (0 = member only); until gatherer %.0f, gather %.0f, search %.0f, hand-out %.0f, after %.0f\n",
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

void VectorIndex::coalescer_config(int32_t window_us, int32_t max_batch, int32_t in_flight) {
    std::lock_guard<std::mutex> lk(coalescer->mu);
    if (window_us >= 0) coalescer->window_us = (uint32_t)window_us;
    if (max_batch > 0) coalescer->max_batch = (uint32_t)max_batch;
    if (in_flight > 0) coalescer->max_in_flight = (uint32_t)std::min(in_flight, 16);
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
