// shard_merge.cpp — multi-shard result merge (host), include/nidx_gpu.h "Shard merge".
//
// Restates nidx/src/searcher/shard_merge.rs: merge_vector_responses (:332-348) and the BM25
// comparators sort_documents_fn / sort_paragraphs_fn (:211-234, :289-312), both driven through
// itertools::kmerge_by — a binary heap of list heads ordered by the "comes first" predicate,
// rebuilt with sift_down after every pop (third-party algorithm, restated).  The per-shard lists are
// what each GPU produced; with one shard per GPU they arrive through an RCCL all-gather.
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "device_common.h"
#include "host_common.h"

namespace nidx {
namespace {

// A few persistent host threads for per-query host work that is too small for a kernel launch and too large to leave serial
// (rank fusion of a 1024-query batch).  run(n, grain, f) calls f(begin, end) over [0, n) in chunks claimed from an atomic
// counter, on the workers and on the caller; it returns when every chunk is done.  One job at a time (callers queue on a mutex).
class HostPool {
  public:
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = hw > 2 ? std::min(hw / 2, 8u) : 0u;
        if (const char *e = getenv("NIDX_GPU_HOST_THREADS")) n = (unsigned)std::max(0, std::min(64, atoi(e) - 1));
        for (unsigned i = 0; i < n; i++) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void run(uint32_t n, uint32_t grain, const std::function<void(uint32_t, uint32_t)> &f) {
        if (n == 0) return;
        if (workers_.empty() || n <= grain || getpid() != pid_) {   // a forked child inherits the object but not the threads
            f(0, n);
            return;
        }
        std::lock_guard<std::mutex> job_lock(job_mu_);
        {
            std::lock_guard<std::mutex> g(mu_);
            f_ = &f;
            n_ = n;
            grain_ = grain;
            next_.store(0);
            pending_ = (n + grain - 1) / grain;
            generation_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [this] { return pending_ == 0 && active_ == 0; });   // no worker is still inside work() when the next job is set up
        f_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            const uint32_t b = next_.fetch_add(grain_);
            if (b >= n_) return;
            (*f_)(b, std::min(n_, b + grain_));
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (pending_ == 0) continue;   // the job finished before this worker woke up
                active_++;
            }
            work();
            std::lock_guard<std::mutex> g(mu_);
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
    const pid_t pid_ = getpid();
    std::vector<std::thread> workers_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(uint32_t, uint32_t)> *f_ = nullptr;
    std::atomic<uint32_t> next_{0};
    uint32_t n_ = 0, grain_ = 1, pending_ = 0, active_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};
HostPool &fuse_pool() {
    // never destroyed: worker threads must not be joined from a static destructor (exit order; a forked child has no workers)
    static HostPool *pool = new HostPool;
    return *pool;
}

struct Head {
    uint32_t list, pos;
};

template <typename First>
void sift_down(std::vector<Head> &heap, size_t len, size_t index, const First &first) {
    size_t pos = index, child = 2 * pos + 1;
    while (child + 1 < len) {
        if (first(heap[child + 1], heap[child])) child++;
        if (!first(heap[child], heap[pos])) return;
        std::swap(heap[pos], heap[child]);
        pos = child;
        child = 2 * pos + 1;
    }
    if (child + 1 == len && first(heap[child], heap[pos])) std::swap(heap[pos], heap[child]);
}

template <typename First>
uint32_t kmerge(const uint32_t *lens, uint32_t n_lists, uint32_t limit, const First &first, std::vector<Head> &order) {
    std::vector<Head> heap;
    for (uint32_t l = 0; l < n_lists; l++)
        if (lens[l] > 0) heap.push_back(Head{l, 0});
    size_t len = heap.size();
    for (size_t i = len / 2; i-- > 0;) sift_down(heap, len, i, first);
    order.clear();
    while (len > 0 && order.size() < limit) {
        order.push_back(heap[0]);
        if (heap[0].pos + 1 < lens[heap[0].list]) heap[0].pos++;
        else {
            heap[0] = heap[len - 1];
            len--;
        }
        sift_down(heap, len, 0, first);
    }
    return (uint32_t)order.size();
}

int cmp_bytes(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb) {
    uint32_t m = la < lb ? la : lb;
    int c = m ? memcmp(a, b, m) : 0;
    if (c) return c;
    return (la > lb) - (la < lb);
}

}  // namespace
}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_merge_vector(const float *const *scores, const uint64_t *const *ids, const uint32_t *lens,
                              uint32_t n_lists, uint32_t limit, float *out_score, uint64_t *out_id, uint32_t *out_list,
                              uint32_t *n_out) try {
    if (!n_out || (n_lists && (!scores || !lens))) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<Head> order;
    // kmerge_by(|a, b| a.score >= b.score)
    auto first = [&](const Head &a, const Head &b) { return scores[a.list][a.pos] >= scores[b.list][b.pos]; };
    uint32_t n = kmerge(lens, n_lists, limit, first, order);
    for (uint32_t i = 0; i < n; i++) {
        if (out_score) out_score[i] = scores[order[i].list][order[i].pos];
        if (out_id && ids) out_id[i] = ids[order[i].list][order[i].pos];
        if (out_list) out_list[i] = order[i].list;
    }
    *n_out = n;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_merge_bm25(const float *const *scores, const uint64_t *const *docaddrs, const uint32_t *lens,
                            const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists,
                            uint32_t limit, float *out_score, uint64_t *out_docaddr, uint32_t *out_list,
                            uint32_t *n_out) try {
    if (!n_out || (n_lists && (!scores || !docaddrs || !lens || !shard_ids || !shard_id_lens)))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<Head> order;
    // a_bm25.total_cmp(b_bm25).then(a.shard_id.cmp(b.shard_id)).then(a_docaddr.cmp(b_docaddr).reverse()).is_gt()
    auto first = [&](const Head &a, const Head &b) {
        int32_t ka = total_key(scores[a.list][a.pos]), kb = total_key(scores[b.list][b.pos]);
        if (ka != kb) return ka > kb;
        int c = cmp_bytes(shard_ids[a.list], shard_id_lens[a.list], shard_ids[b.list], shard_id_lens[b.list]);
        if (c) return c > 0;
        return docaddrs[a.list][a.pos] < docaddrs[b.list][b.pos];
    };
    uint32_t n = kmerge(lens, n_lists, limit, first, order);
    for (uint32_t i = 0; i < n; i++) {
        if (out_score) out_score[i] = scores[order[i].list][order[i].pos];
        if (out_docaddr) out_docaddr[i] = docaddrs[order[i].list][order[i].pos];
        if (out_list) out_list[i] = order[i].list;
    }
    *n_out = n;
    return NIDX_OK;
} NIDX_ABI_CATCH

// ---- the same merges for a whole batch in host memory (layout of the device forms: [n_lists][n_queries][k]) ---------------------
int32_t nidx_gpu_merge_vector_batch(const float *scores, const uint64_t *ids, const uint32_t *counts, uint32_t n_lists, uint32_t n_queries,
                                    uint32_t k, uint32_t limit, float *out_score, uint64_t *out_id, uint32_t *out_count) try {
    if (!out_count || (n_lists && n_queries && (!scores || !ids || !counts))) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<Head> order;
    std::vector<uint32_t> lens(n_lists);
    for (uint32_t q = 0; q < n_queries; q++) {
        for (uint32_t l = 0; l < n_lists; l++) lens[l] = std::min(counts[(size_t)l * n_queries + q], k);
        auto at = [&](const Head &h) { return ((size_t)h.list * n_queries + q) * k + h.pos; };
        auto first = [&](const Head &a, const Head &b) { return scores[at(a)] >= scores[at(b)]; };
        const uint32_t n = kmerge(lens.data(), n_lists, limit, first, order);
        for (uint32_t i = 0; i < n; i++) {
            if (out_score) out_score[(size_t)q * limit + i] = scores[at(order[i])];
            if (out_id) out_id[(size_t)q * limit + i] = ids[at(order[i])];
        }
        out_count[q] = n;
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_merge_bm25_batch(const float *scores, const uint64_t *docaddrs, const int64_t *order_values, const uint32_t *counts,
                                  const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists, uint32_t n_queries,
                                  uint32_t k, uint32_t limit, int32_t order_by, float *out_score, uint64_t *out_docaddr,
                                  int64_t *out_order_value, uint32_t *out_list, uint32_t *out_count) try {
    if (!out_count || (n_lists && (!shard_ids || !shard_id_lens)) || (n_lists && n_queries && (!scores || !docaddrs || !counts)))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (order_by < 0 || order_by > NIDX_MERGE_ORDER_VALUE_ASC) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown merge order %d", order_by);
    if (order_by != NIDX_MERGE_ORDER_SCORE && !order_values) return fail(NIDX_ERR_INVALID_ARGUMENT, "ordering by value needs order_values");
    std::vector<Head> order;
    std::vector<uint32_t> lens(n_lists);
    for (uint32_t q = 0; q < n_queries; q++) {
        for (uint32_t l = 0; l < n_lists; l++) lens[l] = std::min(counts[(size_t)l * n_queries + q], k);
        auto at = [&](const Head &h) { return ((size_t)h.list * n_queries + q) * k + h.pos; };
        auto first = [&](const Head &a, const Head &b) {
            if (order_by == NIDX_MERGE_ORDER_VALUE_DESC) return order_values[at(a)] > order_values[at(b)];
            if (order_by == NIDX_MERGE_ORDER_VALUE_ASC) return order_values[at(a)] < order_values[at(b)];
            const int32_t ka = total_key(scores[at(a)]), kb = total_key(scores[at(b)]);
            if (ka != kb) return ka > kb;
            const int c = cmp_bytes(shard_ids[a.list], shard_id_lens[a.list], shard_ids[b.list], shard_id_lens[b.list]);
            if (c) return c > 0;
            return docaddrs[at(a)] < docaddrs[at(b)];
        };
        const uint32_t n = kmerge(lens.data(), n_lists, limit, first, order);
        for (uint32_t i = 0; i < n; i++) {
            if (out_score) out_score[(size_t)q * limit + i] = scores[at(order[i])];
            if (out_docaddr) out_docaddr[(size_t)q * limit + i] = docaddrs[at(order[i])];
            if (out_order_value && order_values) out_order_value[(size_t)q * limit + i] = order_values[at(order[i])];
            if (out_list) out_list[(size_t)q * limit + i] = order[i].list;
        }
        out_count[q] = n;
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

// ---- merge_facets (shard_merge.rs:380-414) ----------------------------------------------------------------------------------------
int32_t nidx_gpu_merge_facets(const nidx_gpu_facet_count_t *const *shard_facets, const uint32_t *shard_lens, uint32_t n_shards,
                              nidx_gpu_facet_count_t *out, uint32_t capacity, uint32_t *n_out) try {
    if (!n_out || (n_shards && (!shard_facets || !shard_lens)) || (capacity && !out)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<nidx_gpu_facet_count_t> all;
    for (uint32_t s = 0; s < n_shards; s++)
        for (uint32_t i = 0; i < shard_lens[s]; i++) all.push_back(shard_facets[s][i]);
    auto cmp = [](const nidx_gpu_facet_count_t &a, const nidx_gpu_facet_count_t &b) {
        const int g = cmp_bytes(a.group, a.group_len, b.group, b.group_len);
        return g ? g : cmp_bytes(a.tag, a.tag_len, b.tag, b.tag_len);
    };
    // counts.entry((group, tag)).and_modify(|total| *total += ..).or_insert(..): equal keys are summed (i32, wrapping like release Rust)
    std::stable_sort(all.begin(), all.end(), [&](const nidx_gpu_facet_count_t &a, const nidx_gpu_facet_count_t &b) { return cmp(a, b) < 0; });
    uint32_t n = 0;
    for (size_t i = 0; i < all.size();) {
        nidx_gpu_facet_count_t acc = all[i];
        size_t j = i + 1;
        for (; j < all.size() && cmp(all[j], acc) == 0; j++) acc.total = (int32_t)((uint32_t)acc.total + (uint32_t)all[j].total);
        if (n < capacity) out[n] = acc;
        n++;
        i = j;
    }
    *n_out = n;
    return NIDX_OK;
} NIDX_ABI_CATCH

}  // extern "C"

// ---- rank fusion (nucliadb rank_fusion.py:60-254), batched on the host -----------------------------------------------------------
// comb_sum = false: ReciprocalRankFusion._fuse (:139-181); true: WeightedCombSum._fuse (:216-252) — the hits in the order given, the
// term is score * weight (f64, like the Python expression over the f32 scores), a hit's first occurrence is the one kept
static int32_t rank_fusion(const nidx_gpu_ranked_list_t *lists, uint32_t n_lists, uint32_t n_queries, double k, uint32_t window, bool comb_sum,
                           uint64_t *out_ids, double *out_scores, uint32_t *out_counts) {
    if ((n_lists && !lists) || !out_ids || !out_scores || !out_counts) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t l = 0; l < n_lists; l++) {
        if (!lists[l].counts || (lists[l].stride && !lists[l].ids)) return fail(NIDX_ERR_INVALID_ARGUMENT, "list %u: NULL arrays", l);
        if (comb_sum && lists[l].stride && !lists[l].scores) return fail(NIDX_ERR_INVALID_ARGUMENT, "list %u: wCombSUM needs the scores", l);
    }
    struct Item { uint64_t id; double score; };
    // _fuse ranks every source by its OWN scores, descending, with a stable sort (rank_fusion.py:139-147): a list that carries
    // scores and is not already in that order (BM25 hits ordered by a fast field, say) is ranked through a per-query
    // permutation below; a list without scores is taken as ranked
    // Queries are independent: ranges of them go to worker threads (a batch of 1024 hybrid queries spends more time here than in
    // either search kernel otherwise).  Inside a range: first-seen position of every id of one query by open addressing (sized for
    // the query's hits), then a stable order by fused score — an insertion sort for the usual few dozen hits.
    // an exception must not leave a worker thread (that would be std::terminate, not an error code): remember it and report after the job
    std::atomic<int> worker_failure{0};   // 0 none, 1 bad_alloc, 2 anything else
    auto fuse_range = [&](uint32_t q_begin, uint32_t q_end) {
      try {
        std::vector<Item> acc;
        std::vector<uint32_t> order, slot_at, perm;
        std::vector<uint64_t> slot_id;
        for (uint32_t q = q_begin; q < q_end; q++) {
            uint32_t non_empty = 0, only = 0;
            size_t hits = 0;
            for (uint32_t l = 0; l < n_lists; l++) {
                const uint32_t c = std::min(lists[l].counts[q], lists[l].stride);
                if (c) { non_empty++; only = l; }
                hits += c;
            }
            acc.clear();
            if (non_empty == 1) {
                // fuse(): the single source's hits, unchanged (then the same stable sort by their own scores)
                const nidx_gpu_ranked_list_t &L = lists[only];
                const uint32_t c = std::min(L.counts[q], L.stride);
                for (uint32_t r = 0; r < c; r++)
                    acc.push_back({L.ids[(size_t)q * L.stride + r], L.scores ? (double)L.scores[(size_t)q * L.stride + r] : 0.0});
            } else {
                size_t cap = 16;
                while (cap < 2 * hits) cap <<= 1;
                slot_at.assign(cap, 0xffffffffu);
                slot_id.resize(cap);
                for (uint32_t l = 0; l < n_lists; l++) {
                    const nidx_gpu_ranked_list_t &L = lists[l];
                    const uint32_t c = std::min(L.counts[q], L.stride);
                    // sorted(values, key=score, reverse=True): only when the list is not in that order already
                    bool ranked = true;
                    if (L.scores && !comb_sum)
                        for (uint32_t r = 1; r < c && ranked; r++)
                            ranked = !(L.scores[(size_t)q * L.stride + r] > L.scores[(size_t)q * L.stride + r - 1]);
                    if (!ranked) {
                        perm.resize(c);
                        for (uint32_t i = 0; i < c; i++) perm[i] = i;
                        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
                            return L.scores[(size_t)q * L.stride + a] > L.scores[(size_t)q * L.stride + b];
                        });
                    }
                    for (uint32_t r = 0; r < c; r++) {
                        const uint64_t id = L.ids[(size_t)q * L.stride + (ranked ? r : perm[r])];
                        const double term = comb_sum ? (double)L.scores[(size_t)q * L.stride + r] * L.weight : (1.0 / (k + (double)r)) * L.weight;
                        size_t h = (size_t)((id * 0x9E3779B97F4A7C15ull) >> 32) & (cap - 1);
                        while (slot_at[h] != 0xffffffffu && slot_id[h] != id) h = (h + 1) & (cap - 1);
                        if (slot_at[h] == 0xffffffffu) {
                            slot_at[h] = (uint32_t)acc.size();
                            slot_id[h] = id;
                            acc.push_back({id, term});
                        } else {
                            acc[slot_at[h]].score += term;
                        }
                    }
                }
            }
            const uint32_t m = (uint32_t)acc.size();
            order.resize(m);
            if (m <= 96) {   // stable insertion sort, descending by score
                for (uint32_t i = 0; i < m; i++) {
                    const double sc = acc[i].score;
                    uint32_t j = i;
                    while (j > 0 && acc[order[j - 1]].score < sc) {
                        order[j] = order[j - 1];
                        j--;
                    }
                    order[j] = i;
                }
            } else {
                for (uint32_t i = 0; i < m; i++) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return acc[a].score > acc[b].score; });
            }
            const uint32_t n = std::min<uint32_t>(m, window);
            for (uint32_t i = 0; i < n; i++) {
                out_ids[(size_t)q * window + i] = acc[order[i]].id;
                out_scores[(size_t)q * window + i] = acc[order[i]].score;
            }
            out_counts[q] = n;
        }
      } catch (const std::bad_alloc &) {
        worker_failure.store(1);
      } catch (...) {
        worker_failure.store(2);
      }
    };
    fuse_pool().run(n_queries, 64, fuse_range);
    if (worker_failure.load() == 1) return fail(NIDX_ERR_OUT_OF_MEMORY, "rank fusion: a host allocation failed");
    if (worker_failure.load() == 2) return fail(NIDX_ERR_INTERNAL, "rank fusion: unexpected exception in a worker");
    return NIDX_OK;
}

extern "C" {

int32_t nidx_gpu_rank_fusion_rrf(const nidx_gpu_ranked_list_t *lists, uint32_t n_lists, uint32_t n_queries, double k, uint32_t window,
                                 uint64_t *out_ids, double *out_scores, uint32_t *out_counts) try {
    return rank_fusion(lists, n_lists, n_queries, k, window, false, out_ids, out_scores, out_counts);
} NIDX_ABI_CATCH

int32_t nidx_gpu_rank_fusion_wcombsum(const nidx_gpu_ranked_list_t *lists, uint32_t n_lists, uint32_t n_queries, uint32_t window,
                                      uint64_t *out_ids, double *out_scores, uint32_t *out_counts) try {
    return rank_fusion(lists, n_lists, n_queries, 0.0, window, true, out_ids, out_scores, out_counts);
} NIDX_ABI_CATCH

}  // extern "C"
