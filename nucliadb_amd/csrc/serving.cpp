// serving.cpp — pipelined serving of query batches: nidx_gpu_vector_search_submit / _wait.
//
// Replaces the loop the reference runs around VectorSearcher::search (nidx_vector/src/lib.rs:120-148, called once per request
// from nidx/src/searcher/shard_search.rs:139-153).  A GPU launch of the HNSW kernel lasts as long as the longest walk of its
// batch, so ONE batch at a time leaves most of the device idle behind the tail of every launch.  The library
// therefore owns `depth` slots — a HIP stream, pinned staging and a device result block each — and a caller keeps several
// batches in flight:  ticket = submit(batch i + 2);  wait(ticket of batch i);  ...
//
//   submit  stages the queries (host rows through pinned memory, or a device pointer as it is), searches the segments on the
//           slot's stream into the slot's result block — every segment that takes the plain HNSW arm in ONE launch of
//           n_queries x n_segments walks (hnsw_search_segments_kernel: the reference's index at 10 M vectors IS 50 segments,
//           nidx/src/settings.rs:258-278), the other arms one launch each — merges them per query on the device (Fssc,
//           fssc_device.hip), queues ONE device-to-host transfer of [flag words | merged hits] and returns;
//   wait    blocks until the transfer has landed and fills the caller's arrays; only when a segment's flag word says a bounded
//           on-chip structure overflowed it fetches the per-segment rows, re-runs that segment through the exact fallback
//           (segment_search_exact) and merges on the host (the same Fssc, csrc/vector_index.cpp: fssc_merge).
//
// Results are those of nidx_gpu_vector_search for the same batch (tests/test_serving_gpu.py).
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "host_common.h"
#include "vector_index.h"

namespace nidx {

namespace {
// Batches in flight live on separate HIP streams; with the runtime's default of 4 hardware queues two streams can share a queue
// and serialise (measured: 3 batches in flight = 2.2 M queries/s with 4 queues, 3.5 M with 8).  The runtime reads the variable
// when it initialises, i.e. at the first HIP call of the process: this library's load is early enough for a host that links it.
__attribute__((constructor)) void more_hardware_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
}  // namespace

// ---- staging of host query rows ---------------------------------------------------------------------------------------------
// The seam hands over HOST memory (VectorSearchRequest.vector: Vec<f32>, nidx_vector/src/request_types.rs:19-35), and a batch of
// 1 024 x 768 rows is 3 MiB: copied row by row into the slot's pinned staging by the submitting thread alone that is ~0.3 ms — as long
// as the device needs for the whole batch, so the copy, not the GPU, set the rate of the host-buffer entries (2.0 M queries/s against
// 3.5 M from device-resident rows, round 4).  A few helper threads (process-wide, started on first use) share the rows of a batch with
// the submitting thread in chunks; small batches are copied inline.
namespace {
struct StageJob {
    const float *src;
    float *dst;
    uint32_t nq, d, dp, chunk_rows, n_chunks;
    bool normalize;
    std::atomic<uint32_t> next{0}, done{0};
    uint32_t refs = 0;   // helper threads inside run() (under the pool's mutex)
    void run() {         // claims chunks until none is left
        for (;;) {
            const uint32_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= n_chunks) return;
            const uint32_t q0 = c * chunk_rows, q1 = std::min(nq, q0 + chunk_rows);
            if (!normalize && d == dp) memcpy(dst + (size_t)q0 * dp, src + (size_t)q0 * d, (size_t)(q1 - q0) * d * 4);
            else
                for (uint32_t q = q0; q < q1; q++) {
                    float *row = dst + (size_t)q * dp;
                    if (normalize) normalize_row(src + (size_t)q * d, row, d);
                    else memcpy(row, src + (size_t)q * d, (size_t)d * 4);
                    for (uint32_t i = d; i < dp; i++) row[i] = 0.f;
                }
            done.fetch_add(1, std::memory_order_release);
        }
    }
};

struct StagePool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<StageJob *> jobs;
    std::vector<std::thread> threads;
    bool stop = false;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
            if (stop) return;
            StageJob *j = jobs.front();
            if (j->next.load(std::memory_order_relaxed) >= j->n_chunks) {   // every chunk is claimed: nothing left to help with
                jobs.pop_front();
                continue;
            }
            j->refs++;
            lk.unlock();
            j->run();
            lk.lock();
            if (--j->refs == 0) cv_done.notify_all();
        }
    }
    void ensure(uint32_t n) {
        std::lock_guard<std::mutex> lk(mu);
        while (threads.size() < n) threads.emplace_back([this] { worker(); });
    }
    // copies the rows of `j` with the helpers' help; returns when every row is in place and no helper still looks at the job
    void run(StageJob &j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(&j);
        }
        cv_work.notify_all();
        j.run();
        std::unique_lock<std::mutex> lk(mu);
        for (auto it = jobs.begin(); it != jobs.end(); ++it)
            if (*it == &j) { jobs.erase(it); break; }
        cv_done.wait(lk, [&] { return j.refs == 0; });
        lk.unlock();
        while (j.done.load(std::memory_order_acquire) < j.n_chunks) std::this_thread::yield();   // (all claimed, refs == 0: already true)
    }
};
// The pool lives on the heap and is never destroyed: no join at static destruction (a forked child would join threads it does not
// have), and fork() is survived — the child inherits the object but none of its threads, and its mutex may be held by a thread that
// does not exist there: the child's atfork handler drops the pointer (the object is leaked) and the next staging call starts a pool
// of its own.  Python's multiprocessing forks by default.
std::mutex g_pool_mu;
StagePool *g_pool = nullptr;
void pool_atfork_prepare() { g_pool_mu.lock(); }
void pool_atfork_parent() { g_pool_mu.unlock(); }
void pool_atfork_child() {
    g_pool = nullptr;
    g_pool_mu.unlock();
}
StagePool &stage_pool() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool) {
        static bool registered = false;
        if (!registered) {
            (void)pthread_atfork(pool_atfork_prepare, pool_atfork_parent, pool_atfork_child);
            registered = true;
        }
        g_pool = new StagePool();
    }
    return *g_pool;
}
std::atomic<int> g_stage_threads{-1};   // helpers besides the submitting thread; -1 = not yet read from the environment
}  // namespace

// dst[q][0..dp) = src[q][0..d) (normalised when the index normalises its queries), zero padded
void stage_query_rows(const float *src, float *dst, uint32_t nq, uint32_t d, uint32_t dp, bool normalize) {
    int helpers = g_stage_threads.load(std::memory_order_relaxed);
    if (helpers < 0) {
        const char *e = getenv("NIDX_GPU_STAGE_THREADS");
        helpers = e ? std::max(0, std::min(atoi(e), 16)) : 3;
        g_stage_threads.store(helpers, std::memory_order_relaxed);
    }
    StageJob j;
    j.src = src, j.dst = dst, j.nq = nq, j.d = d, j.dp = dp, j.normalize = normalize;
    j.chunk_rows = std::max<uint32_t>(1, (64u << 10) / (dp * 4u));   // ~64 KiB per chunk
    j.n_chunks = (nq + j.chunk_rows - 1) / j.chunk_rows;
    if (helpers == 0 || (size_t)nq * dp * 4 < ((size_t)256 << 10)) {
        j.run();
        return;
    }
    StagePool &P = stage_pool();
    P.ensure((uint32_t)helpers);
    P.run(j);
}

void set_stage_threads(int32_t n) { g_stage_threads.store(std::max(0, std::min(n, 16)), std::memory_order_relaxed); }

struct SearchSlot {
    bool busy = false, waiting = false;
    uint64_t ticket = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    DevBuf d_queries, d_block, d_filter, d_tables, d_offered;
    DevBuf d_rq, d_rq_vis, d_rq_ties, d_rq_entry, d_rq_table;   // RaBitQ segments of a one-launch batch: encoded queries, visited bitsets, entry points, argument table
    DevBuf d_bf_partial, d_bf_table;                 // brute-force segments of a one-launch batch: per-block lists, argument tables
    PinBuf pin_in, pin_out, pin_tables, pin_rq_table, pin_bf_table;
    bool dirty = false;                      // work may be queued on `stream` / the flag words may be set: clean before reuse
    bool merged = false;                     // the block carries the device Fssc's hits
    size_t flag_bytes = 0;                   // flag words at the head of d_block
    size_t mw = 0;                           // words of the merged region behind them (0: the batch is merged on the host)
    // the batch in flight
    uint32_t nq = 0;
    nidx_gpu_vector_search_params_t params{};
    const float *dq = nullptr;               // the device rows the launches read
    std::vector<int> method;                 // per segment; 0 = not searched (nothing can match)
    std::vector<const uint64_t *> d_seg_filter;
    std::atomic<bool> launched{false};   // (read by other submitters: are there batches on the device?)
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
    // At most `walks` batches have their search launches on the device at once, however many tickets are outstanding: the launches of
    // batch i wait (on the device, hipStreamWaitEvent) for the completion of batch i - walks.  Four or more 1 024-walk launches
    // oversubscribe the wave slots (3.5 M queries/s with three in flight, < 2 M with four: DESIGN 4.1) — but a batch that arrives as
    // HOST rows first spends ~0.1 ms in its upload, and with only `walks` tickets outstanding the device then runs one launch short
    // for that long.  So: more tickets than walks; the uploads of the extra ones run ahead, their launches take their turn.
    static constexpr uint32_t GATES = 32;
    uint32_t walks = 3;
    uint64_t launched = 0;            // batches whose launches were queued
    hipEvent_t gate[GATES] = {};      // gate[i % GATES]: completion of batch i
    ~Pipeline() {
        for (hipEvent_t e : gate)
            if (e) (void)hipEventDestroy(e);
    }
};

std::shared_ptr<Pipeline> make_pipeline() { return std::make_shared<Pipeline>(); }

void VectorIndex::pipeline_config(int32_t depth, int32_t walks) {
    std::lock_guard<std::mutex> lk(pipe->mu);
    if (depth >= 1) pipe->depth = (uint32_t)std::min(depth, 16);
    if (walks >= 1) pipe->walks = (uint32_t)std::min(walks, 16);
}

namespace {
// result block of a slot: [flag word per segment, padded to 16 words][merged hits: nq*k segments | paragraphs | vectors | scores,
// nq counts][per segment: nq*k vectors | nq*k scores | nq counts]
inline size_t flag_words(size_t S) { return (S + 15) / 16 * 16; }
inline size_t merged_words(uint32_t nq, uint32_t k) { return (size_t)nq * k * 4 + nq; }
inline size_t seg_words(uint32_t nq, uint32_t k) { return (size_t)nq * k * 2 + nq; }

struct SlotRelease {   // hands the slot back on every exit path
    Pipeline &P;
    SearchSlot *slot;
    ~SlotRelease() {
        if (!slot) return;
        if (slot->dirty) {
            // an error left the slot half way: nothing may still be queued that reads its staging, and the invariant "flag words are
            // zero while the slot is idle" is restored before anybody else takes it
            (void)hipStreamSynchronize(slot->stream);
            if (slot->d_block.p) (void)hipMemsetAsync(slot->d_block.p, 0, std::min(slot->d_block.bytes, slot->flag_bytes), slot->stream);
            (void)hipStreamSynchronize(slot->stream);
            slot->dirty = false;
        }
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
    bool crowded = false;
    {
        std::unique_lock<std::mutex> lk(P.mu);
        for (;;) {
            // `depth` bounds the tickets outstanding, also after it was lowered below the number of slots that exist
            // A blocking search (nidx_gpu_vector_search of a multi-segment index) may take one of 4 slots beyond the budget: its caller
            // may itself hold `depth` tickets it has not waited for yet and would otherwise wait here for a slot only it can free;
            // the extra slots are held by blocking calls alone, which give them back by themselves.
            uint32_t n_busy = 0;
            for (auto &s : P.slots) n_busy += s->busy ? 1u : 0u;
            const uint32_t budget = blocking ? P.depth + 4u : P.depth;
            if (n_busy < budget)
                for (auto &s : P.slots)
                    if (!s->busy) { slot = s.get(); break; }
            if (slot) break;
            if (n_busy < budget) {
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
        // other batches still on the device?  (their launches + this one's oversubscribe the workgroup slots: VectorIndex::crowded_launch)
        // (asked only for the batches whose shape depends on it: small batches take the latency shape whatever else runs)
        if (nq > 256)
            for (auto &s : P.slots)
                if (!crowded && s.get() != slot && s->busy && s->launched && hipEventQuery(s->done) == hipErrorNotReady) crowded = true;
    }
    struct CrowdedScope {
        explicit CrowdedScope(bool on) { VectorIndex::set_crowded_launch(on); }
        ~CrowdedScope() { VectorIndex::set_crowded_launch(false); }
    } crowded_scope(crowded);
    SlotRelease release{P, slot};
    SearchSlot &sl = *slot;
    sl.nq = nq;
    sl.params = p;
    sl.method.assign(S, 0);
    sl.d_seg_filter.assign(S, nullptr);
    sl.launched = false;
    sl.merged = false;
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
            stage_query_rows(queries, qpad, nq, d, dp, cfg.normalize_vectors);
            NIDX_HIP(sl.d_queries.reserve((size_t)nq * dp * 4));
            NIDX_HIP(hipMemcpyAsync(sl.d_queries.p, qpad, (size_t)nq * dp * 4, hipMemcpyHostToDevice, sl.stream));
            sl.dq = sl.d_queries.as<float>();
        }
        // ---- result block; its flag words are zero whenever the slot is idle (a flagged batch clears them in wait) -------------
        // One plain segment (no cross-segment paragraph keys): Fssc is the identity on a falling score list, which the host checks while it
        // copies the row (fssc_merge's fast path) — a merge kernel would only add a launch that, with other batches in flight, waits for
        // workgroup slots behind their walks (8 us alone, 60 us in the hybrid trace) and a larger transfer.
        // Fssc's `seen` set (with_duplicates = false, the default) compares a candidate with everything offered before it: on the device
        // that is one wave scanning the offered list per candidate, O((segments x k)^2 / 64) dependent steps — fine at the 500 candidates
        // of 50 segments x k = 10, seconds at nucliadb's page sizes (k up to 500).  Past 4 096 candidates per query the per-segment rows go
        // to the host merge (a hash set there).
        const bool big_dedup = p.with_duplicates == 0 && (uint64_t)S * k > 4096;
        sl.merged = !(S == 1 && segs[0].key_ids.empty()) && !big_dedup && !getenv("NIDX_GPU_FSSC_HOST");   // the variable: merge on the host (comparison)
        const size_t fw = flag_words(S), mw = sl.merged ? merged_words(nq, k) : 0, sw = seg_words(nq, k), words = fw + mw + S * sw;
        sl.mw = mw;
        sl.dirty = true;   // from here on work is queued on the slot's stream: an error path must drain it (SlotRelease)
        sl.flag_bytes = fw * 4;
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
        // this batch's turn on the device (see Pipeline::walks): its launches queue behind the completion of the batch `walks` before it
        uint64_t my_seq = 0;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            my_seq = P.launched++;
            if (my_seq >= P.walks && P.gate[(my_seq - P.walks) % Pipeline::GATES])
                NIDX_HIP(hipStreamWaitEvent(sl.stream, P.gate[(my_seq - P.walks) % Pipeline::GATES], 0));
        }
        uint32_t *blk = sl.d_block.as<uint32_t>();
        // the argument tables of the two table-driven kernels travel as one pinned block: [HnswSearchArgs x S | FsscSegDev x S]
        const size_t hnsw_tab_bytes = (S * sizeof(HnswSearchArgs) + 63) & ~(size_t)63, tab_bytes = hnsw_tab_bytes + S * sizeof(FsscSegDev);
        NIDX_HIP(sl.pin_tables.reserve(tab_bytes));
        NIDX_HIP(sl.d_tables.reserve(tab_bytes));
        HnswSearchArgs *h_hnsw = sl.pin_tables.as<HnswSearchArgs>();
        FsscSegDev *h_fssc = reinterpret_cast<FsscSegDev *>(sl.pin_tables.as<unsigned char>() + hnsw_tab_bytes);
        uint32_t n_hnsw = 0, n_searched = 0;
        const bool one_launch = S > 1 && !getenv("NIDX_GPU_SEGMENT_LAUNCHES");   // the variable: a launch per segment (comparison)
        // RaBitQ is the reference's default arm of a Dot index with D % 64 == 0 (config.rs:170-173, segment.rs:506-513): the walks of
        // every such segment join ONE table-driven launch too (rabitq_hnsw_segments_kernel), their closest_up_nodes the plain
        // segments' grid in entry mode.  The visited bitsets of the launch (n_queries x vectors of those segments bits) stay under 4 GiB.
        std::vector<uint32_t> rq_segs, bf_segs;
        // (a single RaBitQ segment takes the same path: its scratch is the slot's own, so batches in flight overlap — the per-segment
        // route stages through index-owned scratch, one batch at a time)
        bool rq_one_launch = !getenv("NIDX_GPU_SEGMENT_LAUNCHES") && k <= NIDX_K_MAX;
        if (rq_one_launch) {
            uint64_t vis_words = 0;
            for (size_t s = 0; s < S; s++)
                if (segs[s].has_quant) vis_words += (segs[s].n + 31u) / 32u;
            if (vis_words * 4ull * nq > (4ull << 30)) rq_one_launch = false;
        }
        {
            // launches read index state (tunables, the scratch the scans stage through): under the index lock, which is held for
            // the launch calls only — never across a synchronisation
            std::lock_guard<std::mutex> lock(mu);
            const uint32_t walks = one_launch ? nq * (uint32_t)std::min<size_t>(S, 64) : nq;   // the launch shape follows the grid
            for (size_t s = 0; s < S; s++) {
                VectorSegment &seg = segs[s];
                uint32_t *d_vec = blk + fw + mw + s * sw;
                float *d_score = reinterpret_cast<float *>(d_vec + (size_t)nq * k);
                uint32_t *d_count = d_vec + (size_t)nq * k * 2;
                h_fssc[s] = FsscSegDev{seg.vectors.as<float>(), seg.identity_para ? nullptr : seg.para_of_vec.as<uint32_t>(),
                                       seg.key_ids.empty() ? nullptr : seg.key_ids_dev.as<unsigned long long>(), d_vec, nullptr, d_score, seg.dp, 0u};
                const uint64_t *filt = segment_filters ? segment_filters[s] : nullptr;
                // matching = |filter ∩ alive| (segment.rs:516-531)
                const uint64_t matching = filt ? popcount_filter((uint32_t)s, filt) : seg.alive_count;
                if (matching == 0 || seg.n == 0) continue;
                int method = p.method;
                if (method == NIDX_METHOD_AUTO) {
                    // OpenSegment::_search (segment.rs:506-513,535-555), like search_host
                    const bool rabitq = rabitq_enabled(seg);
                    const bool hnsw = seg.has_graph && use_hnsw(seg.n_paragraphs, matching, k, rabitq);
                    method = rabitq ? (hnsw ? NIDX_METHOD_RABITQ_HNSW : NIDX_METHOD_RABITQ_BRUTE_FORCE)
                                    : (hnsw ? NIDX_METHOD_HNSW : NIDX_METHOD_BRUTE_FORCE);
                }
                if ((method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW) && !seg.has_graph)
                    return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %zu has no HNSW graph", s);
                h_fssc[s].count = d_count;
                n_searched++;
                sl.method[s] = method;
                if (method == NIDX_METHOD_HNSW && one_launch) {
                    // its walks join the one grid below
                    h_hnsw[n_hnsw++] = hnsw_args((uint32_t)s, sl.dq, nq, walks, k, p.min_score, p.with_duplicates != 0, sl.d_seg_filter[s], d_vec, d_score,
                                                 d_count, nullptr, vis_for(walks), blk + s);
                    continue;
                }
                if (method == NIDX_METHOD_RABITQ_HNSW && rq_one_launch && seg.has_quant) {
                    rq_segs.push_back((uint32_t)s);   // launched together behind this loop
                    continue;
                }
                if (method == NIDX_METHOD_BRUTE_FORCE && one_launch && k <= NIDX_K_MAX && scan_takes_tile_kernel((uint32_t)s, nq, k, matching)) {
                    bf_segs.push_back((uint32_t)s);   // their scans share one launch, their merges another
                    continue;
                }
                scan_matching_hint = matching;
                const int32_t rc = segment_search_device((uint32_t)s, sl.dq, nq, k, p.min_score, p.with_duplicates != 0, method, sl.d_seg_filter[s],
                                                         d_vec, d_score, d_count, nullptr, vis_for(nq), sl.stream, blk + s);
                scan_matching_hint = ~0ull;
                if (rc != NIDX_OK) return rc;
            }
            if (!bf_segs.empty()) {
                // Brute-force segments (a selective filter sends every segment of an index there, segment.rs:506-555): ONE launch scans
                // them all (blockIdx.z = segment), one more merges every segment's per-block lists.  The lists of the launch stay under
                // 2 GiB; past that (pages of hundreds of hits over dozens of large segments) the segments keep a launch pair each.
                uint32_t nblk = 1;
                for (uint32_t s : bf_segs) nblk = std::max(nblk, scan_num_blocks(segs[s].n));
                const size_t per_seg = (size_t)nq * nblk * k * 8, n_bf = bf_segs.size();
                if (per_seg * n_bf > ((size_t)2 << 30)) {
                    for (uint32_t s : bf_segs) {
                        uint32_t *d_vec = blk + fw + mw + (size_t)s * sw;
                        const uint64_t *filt = segment_filters ? segment_filters[s] : nullptr;
                        scan_matching_hint = filt ? popcount_filter(s, filt) : segs[s].alive_count;
                        const int32_t rc = segment_search_device(s, sl.dq, nq, k, p.min_score, p.with_duplicates != 0, NIDX_METHOD_BRUTE_FORCE, sl.d_seg_filter[s],
                                                                 d_vec, reinterpret_cast<float *>(d_vec + (size_t)nq * k), d_vec + (size_t)nq * k * 2, nullptr,
                                                                 vis_for(nq), sl.stream, blk + s);
                        scan_matching_hint = ~0ull;
                        if (rc != NIDX_OK) return rc;
                    }
                } else {
                    const size_t scan_tab = (n_bf * sizeof(ScanArgs) + 63) & ~(size_t)63, tab = scan_tab + n_bf * sizeof(ScanMergeTab);
                    NIDX_HIP(sl.d_bf_partial.reserve(per_seg * n_bf));
                    NIDX_HIP(sl.pin_bf_table.reserve(tab));
                    NIDX_HIP(sl.d_bf_table.reserve(tab));
                    ScanArgs *h_scan = sl.pin_bf_table.as<ScanArgs>();
                    ScanMergeTab *h_merge = reinterpret_cast<ScanMergeTab *>(sl.pin_bf_table.as<unsigned char>() + scan_tab);
                    for (size_t i = 0; i < n_bf; i++) {
                        const uint32_t s = bf_segs[i];
                        uint32_t *d_vec = blk + fw + mw + (size_t)s * sw;
                        ScanArgs a = scan_args(s, sl.dq, nq, k, p.min_score, sl.d_seg_filter[s]);
                        a.partial = reinterpret_cast<uint64_t *>(sl.d_bf_partial.as<unsigned char>() + per_seg * i);
                        a.qt = scan_query_tile(nq, a.dp, k);
                        h_scan[i] = a;
                        h_merge[i] = ScanMergeTab{a.partial, d_vec, reinterpret_cast<float *>(d_vec + (size_t)nq * k), d_vec + (size_t)nq * k * 2};
                    }
                    NIDX_HIP(hipMemcpyAsync(sl.d_bf_table.p, sl.pin_bf_table.p, tab, hipMemcpyHostToDevice, sl.stream));
                    NIDX_HIP(launch_scan_segments(sl.d_bf_table.as<ScanArgs>(), (uint32_t)n_bf, h_scan[0], nblk, sl.stream));
                    NIDX_HIP(launch_merge_topk_segments(reinterpret_cast<const ScanMergeTab *>(sl.d_bf_table.as<unsigned char>() + scan_tab), (uint32_t)n_bf, nq, nblk,
                                                        k, sl.stream));
                }
            }
            if (!rq_segs.empty()) {
                const uint32_t dim = segs[rq_segs[0]].dim, nw = dim / 64u, n_rq = (uint32_t)rq_segs.size();
                // the queries' 4-bit codes and constants once per batch (they depend on the query alone)
                const size_t qd_bytes = ((size_t)nq * sizeof(RabitqQueryDev) + 63) & ~(size_t)63;
                NIDX_HIP(sl.d_rq.reserve(qd_bytes + (size_t)nq * 4 * nw * 8));
                RabitqQueryDev *d_qd = sl.d_rq.as<RabitqQueryDev>();
                uint64_t *d_planes = reinterpret_cast<uint64_t *>(sl.d_rq.as<unsigned char>() + qd_bytes);
                NIDX_HIP(launch_rabitq_query(sl.dq, nq, dp, dim, d_qd, d_planes, sl.stream));
                uint64_t vis_words = 0;
                for (uint32_t s : rq_segs) vis_words += (segs[s].n + 31u) / 32u;
                NIDX_HIP(sl.d_rq_vis.reserve((size_t)vis_words * 4 * nq));
                NIDX_HIP(hipMemsetAsync(sl.d_rq_vis.p, 0, (size_t)vis_words * 4 * nq, sl.stream));
                const uint32_t tie_stride = rabitq_tie_stride(std::min<uint32_t>(k * 100u, 2000u));   // (rabitq_hnsw_args: ef)
                NIDX_HIP(sl.d_rq_ties.reserve((size_t)n_rq * nq * tie_stride * 8));
                const size_t entry_words = (size_t)nq * k * 2 + nq;   // per segment: vectors | scores | counts
                NIDX_HIP(sl.d_rq_entry.reserve((size_t)n_rq * entry_words * 4));
                NIDX_HIP(sl.pin_rq_table.reserve((size_t)n_rq * sizeof(RabitqSearchArgs)));
                NIDX_HIP(sl.d_rq_table.reserve((size_t)n_rq * sizeof(RabitqSearchArgs)));
                RabitqSearchArgs *h_rq = sl.pin_rq_table.as<RabitqSearchArgs>();
                uint64_t vis_at = 0;
                for (uint32_t i = 0; i < n_rq; i++) {
                    const uint32_t s = rq_segs[i];
                    uint32_t *e_vec = sl.d_rq_entry.as<uint32_t>() + (size_t)i * entry_words;
                    float *e_score = reinterpret_cast<float *>(e_vec + (size_t)nq * k);
                    uint32_t *e_count = e_vec + (size_t)nq * k * 2;
                    RabitqSearchArgs r = rabitq_hnsw_args(s, sl.dq, nq, k, p.min_score, blk + s);
                    r.qd = d_qd;
                    r.planes = d_planes;
                    r.visited = sl.d_rq_vis.as<uint32_t>() + vis_at * nq;
                    vis_at += r.vis_words;
                    r.tie_stride = tie_stride;
                    r.tie_spill = rabitq_tie_spill_enabled() ? sl.d_rq_ties.as<uint64_t>() + (size_t)i * nq * tie_stride : nullptr;
                    r.out_vec = e_vec, r.out_score = e_score, r.out_count = e_count;
                    h_rq[i] = r;
                    // closest_up_nodes from the re-ranked entry points, on the raw query (search.rs:369-375): an entry-mode record of the grid
                    uint32_t *d_vec = blk + fw + mw + (size_t)s * sw;
                    HnswSearchArgs ha = hnsw_args(s, sl.dq, nq, walks, k, p.min_score, p.with_duplicates != 0, sl.d_seg_filter[s], d_vec,
                                                  reinterpret_cast<float *>(d_vec + (size_t)nq * k), d_vec + (size_t)nq * k * 2, nullptr, vis_for(walks), blk + s);
                    ha.entry_vec = e_vec, ha.entry_score = e_score, ha.entry_count = e_count;
                    h_hnsw[n_hnsw++] = ha;
                }
                NIDX_HIP(hipMemcpyAsync(sl.d_rq_table.p, sl.pin_rq_table.p, (size_t)n_rq * sizeof(RabitqSearchArgs), hipMemcpyHostToDevice, sl.stream));
                NIDX_HIP(launch_rabitq_hnsw_segments(sl.d_rq_table.as<RabitqSearchArgs>(), n_rq, h_rq[0], sl.stream));
            }
            NIDX_HIP(hipMemcpyAsync(sl.d_tables.p, sl.pin_tables.p, tab_bytes, hipMemcpyHostToDevice, sl.stream));
            if (n_hnsw) {
                HnswSearchArgs a = h_hnsw[0];
                a.seg_table = sl.d_tables.as<HnswSearchArgs>();
                a.n_table = n_hnsw;
                NIDX_HIP(launch_hnsw_search(a, waves_per_query, sl.stream));
            }
        }
        // ---- Fssc on the device: the hits a caller gets are k per query, whatever the number of segments -------------------------
        if (sl.merged) {
            FsscArgs f;
            f.segs = reinterpret_cast<const FsscSegDev *>(sl.d_tables.as<unsigned char>() + hnsw_tab_bytes);
            f.n_segs = (uint32_t)S, f.nq = nq, f.k = k, f.dim = d;
            f.with_duplicates = p.with_duplicates != 0;
            f.offered = nullptr;
            f.offered_stride = std::max<uint32_t>(n_searched, 1) * k;
            if (!f.with_duplicates) {
                NIDX_HIP(sl.d_offered.reserve((size_t)nq * f.offered_stride * 12));
                f.offered = sl.d_offered.as<uint32_t>();
            }
            f.out_seg = blk + fw;
            f.out_para = f.out_seg + (size_t)nq * k;
            f.out_vec = f.out_para + (size_t)nq * k;
            f.out_score = reinterpret_cast<float *>(f.out_vec + (size_t)nq * k);
            f.out_count = f.out_vec + (size_t)nq * k * 2;
            NIDX_HIP(launch_fssc_merge(f, sl.stream));
            NIDX_HIP(hipMemcpyAsync(sl.pin_out.p, sl.d_block.p, (fw + mw) * 4, hipMemcpyDeviceToHost, sl.stream));
        } else {
            NIDX_HIP(hipMemcpyAsync(sl.pin_out.p, sl.d_block.p, words * 4, hipMemcpyDeviceToHost, sl.stream));
        }
        NIDX_HIP(hipEventRecord(sl.done, sl.stream));
        {
            std::lock_guard<std::mutex> lk(P.mu);
            hipEvent_t &g = P.gate[my_seq % Pipeline::GATES];
            if (!g) NIDX_HIP(hipEventCreateWithFlags(&g, hipEventDisableTiming));
            NIDX_HIP(hipEventRecord(g, sl.stream));
        }
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
    NIDX_HIP(hipEventSynchronize(sl.done));   // (a failure leaves the slot dirty: SlotRelease drains its stream)
    const size_t fw = flag_words(S), mw = sl.mw, sw = seg_words(nq, k);
    uint32_t *host = sl.pin_out.as<uint32_t>();
    bool flagged = false;
    for (size_t s = 0; s < S; s++)
        if (host[s] && sl.method[s]) flagged = true;
    if (!flagged) {
        sl.dirty = false;   // everything queued has completed and no flag word is set
        if (k == 0) return NIDX_OK;
        if (sl.merged) {
            // the device merged the segments: k hits per query in [segments | paragraphs | vectors | scores | counts]
            const uint32_t *m = host + fw, *cnt = m + (size_t)nq * k * 4;
            for (uint32_t q = 0; q < nq; q++) {
                const uint32_t c = std::min(cnt[q], k);
                out_count[q] = c;
                const size_t at = (size_t)q * k;
                if (out_segment) memcpy(out_segment + at, m + at, (size_t)c * 4);
                if (out_paragraph) memcpy(out_paragraph + at, m + (size_t)nq * k + at, (size_t)c * 4);
                if (out_vector) memcpy(out_vector + at, m + (size_t)nq * k * 2 + at, (size_t)c * 4);
                if (out_score) memcpy(out_score + at, m + (size_t)nq * k * 3 + at, (size_t)c * 4);
            }
            return NIDX_OK;
        }
    } else {
        if (sl.merged) {
            // the merged hits rest on a walk that overflowed: fetch the per-segment rows, repair, merge again on the host
            NIDX_HIP(hipMemcpyAsync(host + fw + mw, sl.d_block.as<uint32_t>() + fw + mw, S * sw * 4, hipMemcpyDeviceToHost, sl.stream));
            NIDX_HIP(hipStreamSynchronize(sl.stream));
        }
        for (size_t s = 0; s < S; s++) {
            if (!host[s] || !sl.method[s]) continue;
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
            memcpy(host + fw + mw + s * sw, pin_out.p, sw * 4);
            if (n_retried_out) *n_retried_out += retried;
        }
        NIDX_HIP(hipMemsetAsync(sl.d_block.p, 0, fw * 4, sl.stream));
        NIDX_HIP(hipStreamSynchronize(sl.stream));
        sl.dirty = false;
    }
    if (k == 0) return NIDX_OK;
    // Fssc across the segments (searcher.rs:149-199, 270-287)
    std::vector<const uint32_t *> pv(S, nullptr), pc(S, nullptr);
    std::vector<const float *> ps(S, nullptr);
    for (size_t s = 0; s < S; s++) {
        if (!sl.method[s]) continue;
        const uint32_t *b = host + fw + mw + s * sw;
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
