// coalescer.cpp — request coalescing for single-query callers (nidx_gpu_vector_search_one).
//
// The reference serves every Search request on its own blocking thread with ONE query vector
// (nidx/src/searcher/shard_search.rs:139-153, nodereader.proto:402); a GPU wants batches.  Concurrent
// callers of nidx_gpu_vector_search_one are therefore merged into batches that run through the serving pipeline
// (serving.cpp: a stream and staging per batch in flight):
//
//   * an arrival JOINS the open batch of its parameters (or opens one and becomes its gatherer): a slot index under the
//     coalescer's mutex — nothing else happens under it — then it copies its own query row into the batch (the gatherer copies
//     the rows of members that have not got that far when it closes the batch);
//   * the gatherer waits until the batch is due — one window after its first request if it then holds its share of the requests
//     in the system, at most six windows (or it is full), AND fewer than max_in_flight batches are running — closes it (later arrivals open the next batch, whose first arrival gathers it while this one is on the device),
//     runs it (submit + wait) and publishes "done" with ONE futex wake for all members;
//   * every member copies its own hits out of the batch.
// Under load a batch closes when an earlier launch finishes, so its size follows the arrival rate.  The mutex is taken twice per
// request for a few instructions; completion takes no lock at all (round 2's version distributed the results and woke every
// member under the mutex: at 256 callers the convoy on that mutex, not the device, set the latency).
// Results are identical to calling nidx_gpu_vector_search with a batch of one (the kernels are per-query deterministic).
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "host_common.h"
#include "vector_index.h"

namespace nidx {

namespace {
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t expected) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t> *w) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
}  // namespace

struct CoBatch {
    nidx_gpu_vector_search_params_t params{};
    uint32_t cap = 0, d = 0;
    uint32_t n = 0;                       // members (under Coalescer::mu while the batch is open, fixed afterwards)
    bool open = false;
    // per member: where its query lies and who copies it into `q` — the member itself as soon as it has joined, or the gatherer
    // when it closes the batch first (with more callers than cores a member can lose its time slice between taking its slot and
    // copying its row: the gatherer must never wait for a descheduled thread).  0 = not copied, 1 = being copied, 2 = in `q`.
    std::unique_ptr<const float *[]> src;
    std::unique_ptr<std::atomic<uint8_t>[]> row_state;
    std::atomic<uint32_t> done{0};        // futex word: 0 = gathering / on the device, 1 = results (or rc) published
    std::unique_ptr<float[]> q;           // [cap][d]
    std::vector<uint32_t> seg, par, vec, cnt;
    std::vector<float> sc;
    int32_t rc = NIDX_OK;
    char error[512] = {0};
};

struct Coalescer {
    std::mutex mu;
    std::condition_variable cv_gather;    // gatherers: a batch filled up, or a batch in flight finished
    std::vector<std::shared_ptr<CoBatch>> open;   // at most one per parameter set
    std::vector<std::shared_ptr<CoBatch>> pool;   // batch objects are recycled (their query block is max_batch rows)
    uint32_t in_flight = 0;
    std::atomic<uint32_t> active{0};      // requests inside search_one (joined, on the device or collecting their hits)
    uint64_t n_batches = 0, n_queries = 0;
    uint32_t window_us = 50, max_batch = 1024, max_in_flight = 4;
    // Admission: at most max_callers requests are inside the coalescer (joined, on the device or collecting their hits); the others
    // wait at the door in ARRIVAL ORDER — a ticket each; ticket t enters once t - max_callers requests have left — or are turned
    // away (NIDX_ERR_BUSY) when the host prefers to shed load.  Without the bound 1 024 blocked callers (16 threads per core) all
    // contend for batches, mutex and time slices at once: 100 k queries/s at p99 87 ms, where 256 callers get 318 k at p99 2.4 ms
    // (round 3).  The hand-over is targeted: a request that leaves wakes exactly the one waiter whose turn it makes (its own futex
    // word), never the crowd; and a caller that comes straight back with its next request queues behind the waiters (a free-for-all
    // door let the callers inside re-enter for ever while the woken ones found the door shut again: 10 k queries/s, p50 92 ms).
    static constexpr uint32_t DOOR_SLOTS = 4096;
    std::atomic<uint32_t> max_callers{256};   // 0 = unbounded
    std::atomic<uint32_t> reject_when_full{0};
    std::atomic<uint64_t> next_ticket{0}, left{0};   // requests that took a ticket / that have left
    std::unique_ptr<std::atomic<uint32_t>[]> door{new std::atomic<uint32_t>[DOOR_SLOTS]()};   // futex word of ticket t: door[t % DOOR_SLOTS]
    std::atomic<uint64_t> n_waited{0}, n_rejected{0};
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
    const uint32_t d = cfg.dimension;
    const auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t_in = std::chrono::steady_clock::now();
    auto t_joined = t_in, t_gathered = t_in, t_searched = t_in;
    // ---- admission (see Coalescer::max_callers) ----
    {
        const uint32_t cap0 = c.max_callers.load(std::memory_order_relaxed);
        if (cap0 && c.reject_when_full.load(std::memory_order_relaxed)) {
            // `left` first: requests that leave between the two loads can only make the crowd look larger by the tickets taken
            // meanwhile, never negative (the other order let `left` overtake the ticket count read before it: a huge unsigned
            // difference and a spurious NIDX_ERR_BUSY on an idle coalescer)
            const uint64_t gone0 = c.left.load(std::memory_order_seq_cst);
            const int64_t inside = (int64_t)(c.next_ticket.load(std::memory_order_seq_cst) - gone0);
            if (inside >= (int64_t)cap0) {
                c.n_rejected.fetch_add(1, std::memory_order_relaxed);
                return fail(NIDX_ERR_BUSY, "%u single-query requests are already inside the coalescer (coalesce_max_callers)", cap0);
            }
        }
        // the ticket / left hand-shake is a store-buffering pattern (a waiter publishes its ticket, then reads `left`; a leaver
        // publishes `left`, then reads the tickets): sequentially consistent on both sides, so one of the two always sees the other
        const uint64_t t = c.next_ticket.fetch_add(1, std::memory_order_seq_cst);
        std::atomic<uint32_t> &word = c.door[t % Coalescer::DOOR_SLOTS];
        bool waited = false;
        for (;;) {
            // the futex word BEFORE the cap and `left`: whoever changes either afterwards bumps the word (a leaver whose leaving
            // makes it this ticket's turn; coalescer_admission when the cap changes), so the wait below returns at once instead of
            // sleeping on a decision taken from stale values
            const uint32_t v = word.load(std::memory_order_seq_cst);
            const uint32_t cap = c.max_callers.load(std::memory_order_seq_cst);
            if (cap == 0) break;
            // a crowd of more than two doors' worth outside: half the door (every request needs a core to get in and out, and with
            // that many blocked threads the cores are the bottleneck: measured at 1 024 callers on 64 cores, door 128 serves 191 k
            // queries/s at p99 27 ms, door 256 153 k at 68 ms — while 256 callers want the whole 256: 340 k against 223 k)
            const uint64_t gone = c.left.load(std::memory_order_seq_cst);
            const uint32_t eff = c.next_ticket.load(std::memory_order_relaxed) - gone > 2ull * cap ? std::max(1u, cap / 2) : cap;
            if (t < gone + eff) break;
            waited = true;
            futex_wait(&word, v);   // (the request whose leaving makes it this ticket's turn bumps the word first)
        }
        if (waited) c.n_waited.fetch_add(1, std::memory_order_relaxed);
        c.active.fetch_add(1, std::memory_order_relaxed);
    }
    struct ActiveCount {
        Coalescer &c;
        ~ActiveCount() {
            c.active.fetch_sub(1, std::memory_order_relaxed);
            const uint64_t gone = c.left.fetch_add(1, std::memory_order_seq_cst) + 1;
            const uint32_t cap = c.max_callers.load(std::memory_order_seq_cst);
            if (cap) {
                // the tickets whose turn this makes — under the whole door and under the halved one (a waiter judges by the crowd it
                // sees when it looks: both are told, so none sleeps through its turn) — are woken if they wait
                const uint64_t tickets = c.next_ticket.load(std::memory_order_seq_cst);
                const uint64_t turns[2] = {gone + cap - 1, gone + std::max(1u, cap / 2) - 1};
                for (int i = 0; i < (turns[0] == turns[1] ? 1 : 2); i++) {
                    if (turns[i] >= tickets) continue;
                    std::atomic<uint32_t> &word = c.door[turns[i] % Coalescer::DOOR_SLOTS];
                    word.fetch_add(1, std::memory_order_release);
                    futex_wake_all(&word);   // (tickets DOOR_SLOTS apart share a word: all of them look again)
                }
            }
        }
    } active_count{c};
    std::shared_ptr<CoBatch> b;
    uint32_t slot = 0;
    bool gatherer = false;
    {
        // ---- join: everything that can fail for lack of memory happens before the batch is visible to other callers ----------------
        std::unique_lock<std::mutex> lk(c.mu);
        for (auto &ob : c.open)
            if (ob->n < ob->cap && same_params(ob->params, p)) { b = ob; break; }
        if (!b) {
            for (auto it = c.pool.begin(); it != c.pool.end(); ++it)
                if (it->use_count() == 1 && (*it)->cap == c.max_batch && (*it)->d == d) { b = *it; break; }
            if (!b) {
                b = std::make_shared<CoBatch>();
                b->cap = c.max_batch;
                b->d = d;
                b->q.reset(new float[(size_t)b->cap * d]);
                b->src.reset(new const float *[b->cap]);
                b->row_state.reset(new std::atomic<uint8_t>[b->cap]);
                if (c.pool.size() < 32) c.pool.push_back(b);
            }
            b->params = p;
            b->n = 0;
            for (uint32_t i = 0; i < b->cap; i++) b->row_state[i].store(0, std::memory_order_relaxed);
            b->done.store(0, std::memory_order_relaxed);
            b->rc = NIDX_OK;
            b->open = true;
            c.open.push_back(b);
            gatherer = true;
        }
        slot = b->n++;
        b->src[slot] = query;
        if (!gatherer && b->n == b->cap) c.cv_gather.notify_all();   // full: its gatherer need not sit out the window
    }
    {
        uint8_t unclaimed = 0;
        if (b->row_state[slot].compare_exchange_strong(unclaimed, 1, std::memory_order_acquire)) {
            std::memcpy(b->q.get() + (size_t)slot * d, query, (size_t)d * 4);
            b->row_state[slot].store(2, std::memory_order_release);
        }
    }
    t_joined = std::chrono::steady_clock::now();
    uint32_t led = 0;
    if (gatherer) {
        uint32_t B;
        {
            std::unique_lock<std::mutex> lk(c.mu);
            // The batch is due one window after its first request IF it then holds its share of the requests that are inside
            // search_one right now (active / (max_in_flight + 1): with F batches on the device and one gathering, that share keeps
            // every caller either in a launch or in the next one); a batch short of its share waits on, window by window, at most
            // six.  A lone caller waits one window; 64 callers pipeline through batches of ~13; 1 024 through batches of ~200.
            const auto w = std::chrono::microseconds(c.window_us);
            auto deadline = t_in + w;
            const auto latest = t_in + 6 * w;
            for (;;) {
                const auto now = std::chrono::steady_clock::now();
                const uint32_t share = std::max<uint32_t>(1u, c.active.load(std::memory_order_relaxed) / (c.max_in_flight + 1u));
                const bool due = b->n >= b->cap || (now >= deadline && (b->n >= share || now >= latest));
                if (due && c.in_flight < c.max_in_flight) break;
                if (due) c.cv_gather.wait(lk);            // only a finishing batch can make it runnable
                else {
                    if (now >= deadline) deadline = std::min(latest, now + w);
                    c.cv_gather.wait_until(lk, deadline);
                }
            }
            // close: later arrivals open the next batch and gather it while this one is on the device
            for (auto it = c.open.begin(); it != c.open.end(); ++it)
                if (it->get() == b.get()) { c.open.erase(it); break; }
            b->open = false;
            B = b->n;
            c.in_flight++;
        }
        for (uint32_t i = 0; i < B; i++) {   // rows their members have not copied yet
            uint8_t unclaimed = 0;
            if (b->row_state[i].compare_exchange_strong(unclaimed, 1, std::memory_order_acquire)) {
                std::memcpy(b->q.get() + (size_t)i * d, b->src[i], (size_t)d * 4);
                b->row_state[i].store(2, std::memory_order_release);
            }
        }
        for (uint32_t i = 0; i < B; i++)     // (a member inside its 3 KiB copy right now)
            while (b->row_state[i].load(std::memory_order_acquire) != 2) sched_yield();
        t_gathered = std::chrono::steady_clock::now();
        led = B;
        const uint32_t k = b->params.k;
        const size_t kk = std::max<uint32_t>(k, 1);
        int32_t rc;
        // the members of this batch are parked: whatever happens here, they must be released
        try {
            b->seg.resize(B * kk), b->par.resize(B * kk), b->vec.resize(B * kk), b->cnt.resize(B), b->sc.resize(B * kk);
            uint64_t ticket = 0;
            rc = pipeline_submit(b->q.get(), B, b->params, nullptr, /*blocking=*/true, &ticket);
            if (rc == NIDX_OK) rc = pipeline_wait(ticket, b->seg.data(), b->par.data(), b->vec.data(), b->sc.data(), b->cnt.data(), nullptr);
        } catch (...) {
            rc = abi_exception();
        }
        b->rc = rc;
        if (rc != NIDX_OK) nidx_gpu_last_error(b->error, sizeof(b->error));
        t_searched = std::chrono::steady_clock::now();
        b->done.store(1, std::memory_order_release);
        futex_wake_all(&b->done);
        {
            std::lock_guard<std::mutex> lk(c.mu);
            c.n_batches++;
            c.n_queries += B;
            c.in_flight--;
        }
        c.cv_gather.notify_all();   // gatherers waiting for a free slot
    } else {
        while (b->done.load(std::memory_order_acquire) == 0) futex_wait(&b->done, 0);
    }
    // ---- every member takes its own hits ------------------------------------------------------------------------------------------
    const int32_t rc = b->rc;
    if (rc == NIDX_OK) {
        const size_t kk = std::max<uint32_t>(b->params.k, 1);
        const uint32_t cnt = b->cnt[slot];
        *out_count = cnt;
        for (uint32_t j = 0; j < cnt; j++) {
            if (out_segment) out_segment[j] = b->seg[slot * kk + j];
            if (out_paragraph) out_paragraph[j] = b->par[slot * kk + j];
            if (out_vector) out_vector[j] = b->vec[slot * kk + j];
            if (out_score) out_score[j] = b->sc[slot * kk + j];
        }
    } else {
        set_error("%s", b->error);
    }
    if (trace_slow_us() > 0) {
        const auto t_out = std::chrono::steady_clock::now();
        if (us(t_in, t_out) > trace_slow_us())
            fprintf(stderr, "[nidx_gpu slow search_one] total %.0f us: led a batch of %u (0 = member only); join %.0f, gather %.0f, search %.0f, after %.0f\n",
                    us(t_in, t_out), led, us(t_in, t_joined), us(t_joined, t_gathered), us(t_gathered, t_searched), us(t_searched, t_out));
    }
    return rc;
}

void VectorIndex::coalescer_stats(uint64_t &batches, uint64_t &queries) {
    batches = queries = 0;
    if (!coalescer) return;
    std::lock_guard<std::mutex> lk(coalescer->mu);
    batches = coalescer->n_batches;
    queries = coalescer->n_queries;
}

std::shared_ptr<Coalescer> make_coalescer() {
    auto c = std::make_shared<Coalescer>();
    if (const char *e = getenv("NIDX_GPU_COALESCE_MAX_CALLERS")) c->max_callers.store((uint32_t)std::max(0, atoi(e)));   // the tunable's default
    return c;
}

void VectorIndex::coalescer_admission(int32_t max_callers, int32_t reject_when_full) {
    if (max_callers >= 0) {
        coalescer->max_callers.store((uint32_t)max_callers, std::memory_order_seq_cst);
        for (uint32_t i = 0; i < Coalescer::DOOR_SLOTS; i++) {   // a changed bound lets everybody at the door look again (the cap first,
                                                                 // then the words: a waiter reads its word first, then the cap)
            coalescer->door[i].fetch_add(1, std::memory_order_seq_cst);
            futex_wake_all(&coalescer->door[i]);
        }
    }
    if (reject_when_full >= 0) coalescer->reject_when_full.store(reject_when_full ? 1u : 0u, std::memory_order_relaxed);
}

void VectorIndex::coalescer_config(int32_t window_us, int32_t max_batch, int32_t in_flight) {
    std::lock_guard<std::mutex> lk(coalescer->mu);
    if (window_us >= 0) coalescer->window_us = (uint32_t)window_us;
    if (max_batch > 0) coalescer->max_batch = (uint32_t)std::min(max_batch, 4096);   // batches already open keep their size
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
