// serving.cpp — pipelined serving of query batches: nidx_gpu_vector_search_submit / _wait.
//
// Replaces the loop the reference runs around VectorSearcher::search (nidx_vector/src/lib.rs:120-148, called once per request
// from nidx/src/searcher/shard_search.rs:139-153).  A GPU launch of the HNSW kernel lasts as long as the longest walk of its
// batch, so ONE batch at a time leaves most of the device idle behind the tail of every launch.  The library
// therefore owns `depth` slots — a HIP stream, pinned staging and a device result block each — and a caller keeps several
// batches in flight:  ticket = submit(batch i + 2);  wait(ticket of batch i);  ...
//
//   submit  stages the queries (host rows through pinned memory, or a device pointer as it is), launches every segment's search
//           on the slot's stream into the slot's result block, queues ONE device-to-host transfer of the block
//           ([flag words | per segment: vectors, scores, counts]) and returns;
//   wait    blocks until the transfer has landed, re-runs — only when a segment's flag word says a bounded on-chip structure
//           overflowed — that segment through the exact fallback (segment_search_exact), merges the segments with Fssc and
//           fills the caller's arrays.
//
// Results are those of nidx_gpu_vector_search for the same batch (tests/test_serving_gpu.py).
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "host_common.h"
#include "vector_index.h"

namespace nidx {

namespace {
// Batches in flight live on separate HIP streams; with the runtime's default of 4 hardware queues two streams can share a queue
// and serialise (measured: 3 batches in flight = 2.2 M queries/s with 4 queues, 3.5 M with 8).  The runtime reads the variable
// when it initialises, i.e. at the first HIP call of the process: this library's load is early enough for a host that links it.
__attribute__((constructor)) void more_hardware_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
}  // namespace

struct SearchSlot {
    bool busy = false, waiting = false;
    uint64_t ticket = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    DevBuf d_queries, d_block, d_filter;
    PinBuf pin_in, pin_out;
    // the batch in flight
    uint32_t nq = 0;
    nidx_gpu_vector_search_params_t params{};
    const float *dq = nullptr;               // the device rows the launches read
    std::vector<int> method;                 // per segment; 0 = not searched (nothing can match)
    std::vector<const uint64_t *> d_seg_filter;
    bool launched = false;
    ~SearchSlot() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (done) (void)hipEventDestroy(done);
    }
};

struct Pipeline {
    std::mutex mu;
    std::condition_variable cv_free;
    std::vector<std::unique_ptr<SearchSlot>> slots;
    uint32_t depth = 4;
    uint64_t next_ticket = 1;
};

std::shared_ptr<Pipeline> make_pipeline() { return std::make_shared<Pipeline>(); }

void VectorIndex::pipeline_config(int32_t depth) {
    std::lock_guard<std::mutex> lk(pipe->mu);
    if (depth >= 1) pipe->depth = (uint32_t)std::min(depth, 16);
}

namespace {
// result block of a slot: [flag word per segment, padded to 16 words][per segment: nq*k vectors | nq*k scores | nq counts]
inline size_t flag_words(size_t S) { return (S + 15) / 16 * 16; }
inline size_t seg_words(uint32_t nq, uint32_t k) { return (size_t)nq * k * 2 + nq; }

struct SlotRelease {   // hands the slot back on every exit path
    Pipeline &P;
    SearchSlot *slot;
    ~SlotRelease() {
        if (!slot) return;
        std::lock_guard<std::mutex> lk(P.mu);
        slot->busy = slot->waiting = false;
        slot->ticket = 0;
        P.cv_free.notify_one();
    }
};

bool is_device_pointer(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary host pointer is "invalid value" to the runtime: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}
}  // namespace

int32_t VectorIndex::pipeline_submit(const float *queries, uint32_t nq, const nidx_gpu_vector_search_params_t &p,
                                     const uint64_t *const *segment_filters, bool blocking, uint64_t *ticket_out) {
    Pipeline &P = *pipe;
    const uint32_t k = p.k, d = cfg.dimension, dp = (d + 3u) & ~3u;
    const size_t S = segs.size();
    if (k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "result_per_page > %d is not supported (got %u)", NIDX_K_MAX, k);
    if (p.method < 0 || p.method > 6) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown search method %d", p.method);
    NIDX_HIP(hipSetDevice(device));

    SearchSlot *slot = nullptr;
    {
        std::unique_lock<std::mutex> lk(P.mu);
        for (;;) {
            // `depth` bounds the tickets outstanding, also after it was lowered below the number of slots that exist
            uint32_t n_busy = 0;
            for (auto &s : P.slots) n_busy += s->busy ? 1u : 0u;
            if (n_busy < P.depth)
                for (auto &s : P.slots)
                    if (!s->busy) { slot = s.get(); break; }
            if (slot) break;
            if (n_busy < P.depth) {
                std::unique_ptr<SearchSlot> ns(new SearchSlot());
                NIDX_HIP(hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking));
                NIDX_HIP(hipEventCreateWithFlags(&ns->done, hipEventDisableTiming));
                P.slots.push_back(std::move(ns));
                slot = P.slots.back().get();
                break;
            }
            if (!blocking)
                return fail(NIDX_ERR_BUSY, "all %u pipeline slots hold a ticket that has not been waited for", P.depth);
            P.cv_free.wait(lk);
        }
        slot->busy = true;
        slot->waiting = false;
        slot->ticket = P.next_ticket++;
    }
    SlotRelease release{P, slot};
    SearchSlot &sl = *slot;
    sl.nq = nq;
    sl.params = p;
    sl.method.assign(S, 0);
    sl.d_seg_filter.assign(S, nullptr);
    sl.launched = false;
    sl.dq = nullptr;
    if (nq > 0 && k > 0 && S > 0) {
        // ---- queries: a device pointer is searched where it lies; host rows go through the slot's pinned staging --------------
        if (is_device_pointer(queries)) {
            if ((d & 3u) || cfg.normalize_vectors)
                return fail(NIDX_ERR_UNSUPPORTED, "device-resident queries need a dimension that is a multiple of 4 and an index that does not normalise its queries");
            sl.dq = queries;
        } else {
            NIDX_HIP(sl.pin_in.reserve((size_t)nq * dp * 4));
            float *qpad = sl.pin_in.as<float>();
            for (uint32_t q = 0; q < nq; q++) {
                float *row = qpad + (size_t)q * dp;
                if (cfg.normalize_vectors) normalize_row(queries + (size_t)q * d, row, d);
                else memcpy(row, queries + (size_t)q * d, (size_t)d * 4);
                for (uint32_t i = d; i < dp; i++) row[i] = 0.f;
            }
            NIDX_HIP(sl.d_queries.reserve((size_t)nq * dp * 4));
            NIDX_HIP(hipMemcpyAsync(sl.d_queries.p, qpad, (size_t)nq * dp * 4, hipMemcpyHostToDevice, sl.stream));
            sl.dq = sl.d_queries.as<float>();
        }
        // ---- result block; its flag words are zero whenever the slot is idle (a flagged batch clears them in wait) -------------
        const size_t fw = flag_words(S), sw = seg_words(nq, k), words = fw + S * sw;
        if (words * 4 > sl.d_block.bytes) {
            NIDX_HIP(sl.d_block.reserve(words * 4));
            NIDX_HIP(hipMemsetAsync(sl.d_block.p, 0, fw * 4, sl.stream));
        }
        NIDX_HIP(sl.pin_out.reserve(words * 4));
        // ---- per-segment filters (bitsets over paragraph addresses) ------------------------------------------------------------
        if (segment_filters) {
            size_t fwords = 0;
            for (size_t s = 0; s < S; s++)
                if (segment_filters[s]) fwords += (segs[s].n_paragraphs + 63) / 64;
            NIDX_HIP(sl.d_filter.reserve(fwords * 8));
            size_t at = 0;
            for (size_t s = 0; s < S; s++) {
                if (!segment_filters[s]) continue;
                const size_t w = (segs[s].n_paragraphs + 63) / 64;
                uint64_t *dst = sl.d_filter.as<uint64_t>() + at;
                NIDX_HIP(hipMemcpyAsync(dst, segment_filters[s], w * 8, hipMemcpyHostToDevice, sl.stream));
                sl.d_seg_filter[s] = dst;
                at += w;
            }
        }
        uint32_t *blk = sl.d_block.as<uint32_t>();
        {
            // launches read index state (tunables, the scratch the scans stage through): under the index lock, which is held for
            // the launch calls only — never across a synchronisation
            std::lock_guard<std::mutex> lock(mu);
            for (size_t s = 0; s < S; s++) {
                VectorSegment &seg = segs[s];
                const uint64_t *filt = segment_filters ? segment_filters[s] : nullptr;
                // matching = |filter ∩ alive| (segment.rs:516-531)
                const uint64_t matching = filt ? popcount_filter((uint32_t)s, filt) : seg.alive_count;
                if (matching == 0 || seg.n == 0) continue;
                int method = p.method;
                if (method == NIDX_METHOD_AUTO) {
                    // OpenSegment::_search (segment.rs:506-513,535-555), like search_host
                    const bool rabitq = rabitq_enabled(seg) && k <= 256;
                    const bool hnsw = seg.has_graph && use_hnsw(seg.n_paragraphs, matching, k, rabitq);
                    method = rabitq ? (hnsw ? NIDX_METHOD_RABITQ_HNSW : NIDX_METHOD_RABITQ_BRUTE_FORCE)
                                    : (hnsw ? NIDX_METHOD_HNSW : NIDX_METHOD_BRUTE_FORCE);
                }
                if ((method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW) && !seg.has_graph)
                    return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %zu has no HNSW graph", s);
                uint32_t *d_vec = blk + fw + s * sw;
                float *d_score = reinterpret_cast<float *>(d_vec + (size_t)nq * k);
                uint32_t *d_count = d_vec + (size_t)nq * k * 2;
                scan_matching_hint = matching;
                const int32_t rc = segment_search_device((uint32_t)s, sl.dq, nq, k, p.min_score, p.with_duplicates != 0, method, sl.d_seg_filter[s],
                                                         d_vec, d_score, d_count, nullptr, default_vis_log2, sl.stream, blk + s);
                scan_matching_hint = ~0ull;
                if (rc != NIDX_OK) return rc;
                sl.method[s] = method;
            }
        }
        NIDX_HIP(hipMemcpyAsync(sl.pin_out.p, sl.d_block.p, words * 4, hipMemcpyDeviceToHost, sl.stream));
        NIDX_HIP(hipEventRecord(sl.done, sl.stream));
        sl.launched = true;
    }
    *ticket_out = sl.ticket;
    release.slot = nullptr;   // the ticket owns the slot until it is waited for
    return NIDX_OK;
}

int32_t VectorIndex::pipeline_wait(uint64_t ticket, uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                                   uint32_t *out_count, uint32_t *n_retried_out) {
    Pipeline &P = *pipe;
    SearchSlot *slot = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto &s : P.slots)
            if (s->busy && s->ticket == ticket && ticket != 0) { slot = s.get(); break; }
        if (!slot || slot->waiting) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown ticket %llu (a ticket is waited for once)", (unsigned long long)ticket);
        slot->waiting = true;
    }
    SlotRelease release{P, slot};
    SearchSlot &sl = *slot;
    const uint32_t nq = sl.nq, k = sl.params.k;
    const size_t S = segs.size();
    if (n_retried_out) *n_retried_out = 0;
    for (uint32_t q = 0; q < nq; q++) out_count[q] = 0;
    if (!sl.launched) return NIDX_OK;
    NIDX_HIP(hipSetDevice(device));
    NIDX_HIP(hipEventSynchronize(sl.done));
    const size_t fw = flag_words(S), sw = seg_words(nq, k);
    uint32_t *host = sl.pin_out.as<uint32_t>();
    bool flagged = false;
    for (size_t s = 0; s < S; s++) {
        if (!host[s] || !sl.method[s]) continue;
        flagged = true;
        if (sl.method[s] != NIDX_METHOD_HNSW && sl.method[s] != NIDX_METHOD_RABITQ_HNSW) continue;
        // a bounded on-chip structure overflowed for some query of this segment: the complete OpenSegment::search (larger visited
        // table / HBM-resident walk for the flagged queries), synchronously, into the index's own block, then over the slot's rows
        std::lock_guard<std::mutex> lock(mu);
        const size_t bw = out_block_words(nq, k);
        NIDX_HIP(scratch_out_block.reserve(bw * 4));
        NIDX_HIP(pin_out.reserve(bw * 4));
        uint32_t retried = 0;
        const int32_t rc = segment_search_exact((uint32_t)s, sl.dq, nq, k, sl.params.min_score, sl.params.with_duplicates != 0, sl.method[s],
                                                sl.d_seg_filter[s], scratch_out_block.as<uint32_t>(), pin_out.as<uint32_t>(), sl.stream, &retried);
        if (rc != NIDX_OK) return rc;
        memcpy(host + fw + s * sw, pin_out.p, sw * 4);
        if (n_retried_out) *n_retried_out += retried;
    }
    if (flagged) {
        NIDX_HIP(hipMemsetAsync(sl.d_block.p, 0, fw * 4, sl.stream));
        NIDX_HIP(hipStreamSynchronize(sl.stream));
    }
    if (k == 0) return NIDX_OK;
    // Fssc across the segments (searcher.rs:149-199, 270-287)
    std::vector<const uint32_t *> pv(S, nullptr), pc(S, nullptr);
    std::vector<const float *> ps(S, nullptr);
    for (size_t s = 0; s < S; s++) {
        if (!sl.method[s]) continue;
        const uint32_t *b = host + fw + s * sw;
        pv[s] = b;
        ps[s] = reinterpret_cast<const float *>(b + (size_t)nq * k);
        pc[s] = b + (size_t)nq * k * 2;
    }
    return fssc_merge(nq, sl.params, pv.data(), ps.data(), pc.data(), out_segment, out_paragraph, out_vector, out_score, out_count);
}

}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_vector_search_submit(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries, uint32_t query_dimension,
                                      const nidx_gpu_vector_search_params_t *params, const uint64_t *const *segment_filters,
                                      uint64_t *ticket_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || !ticket_out || (n_queries && !queries)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *ticket_out = 0;
    if (query_dimension != idx->cfg.dimension)
        return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "Inconsistent dimensions. Index=%u Vector=%u", idx->cfg.dimension, query_dimension);
    return idx->pipeline_submit(queries, n_queries, *params, segment_filters, false, ticket_out);
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_search_wait(nidx_gpu_vector_index_t *index, uint64_t ticket, uint32_t *out_segment, uint32_t *out_paragraph,
                                    uint32_t *out_vector, float *out_score, uint32_t *out_count, uint32_t *n_retried_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !out_count) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    return idx->pipeline_wait(ticket, out_segment, out_paragraph, out_vector, out_score, out_count, n_retried_out);
} NIDX_ABI_CATCH

}  // extern "C"
