// shard_merge_device.hip — the k-way merges of nidx/src/searcher/shard_merge.rs for a whole query batch on the device: after the
// RCCL all-gather every GPU holds the P per-shard top-k lists of each query; one thread per query runs the same merge the host
// entry points run (itertools::kmerge_by: binary heap of list heads ordered by the "comes first" predicate, sift_down after
// every pop), so ties resolve exactly as in nidx_gpu_merge_vector / nidx_gpu_merge_bm25.
//
//   vector   merge_vector_responses (:332-348)             first(a, b) = a.score >= b.score
//   bm25     sort_documents_fn / sort_paragraphs_fn, SortExpr::Score (:211-234, :289-312)
//                                                          first(a, b) = bm25 greater (total_cmp), else shard_id greater (bytes:
//                                                          the caller passes every list's rank in that order), else docaddr smaller
//   date     the same two functions, SortExpr::Date        first(a, b) = value strictly greater (descending) / smaller (ascending)
//
// The lists of shard l start at base + l * stride (bytes): separate [P][B][k] arrays and the packed per-rank blocks the
// exchange gathers (shard_comm.cpp) are the same kernel.
#include <string.h>

#include "device_common.h"
#include "host_common.h"
#include "shard_merge_device.h"

namespace nidx {

struct MergeHead {
    uint32_t list, pos;
};

enum { MERGE_VECTOR = 0, MERGE_BM25 = 1, MERGE_VALUE_DESC = 2, MERGE_VALUE_ASC = 3 };

template <int MODE>
__global__ void merge_lists_kernel(MergeListsArgs a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t B = a.n_queries, k = a.k, limit = a.limit;
    if (q >= B) return;
    MergeHead heap[MERGE_MAX_LISTS];
    uint32_t len = 0;
    auto scores_of = [&](uint32_t l) { return reinterpret_cast<const float *>(a.scores + (size_t)l * a.scores_stride) + (size_t)q * k; };
    auto ids_of = [&](uint32_t l) { return reinterpret_cast<const uint64_t *>(a.ids + (size_t)l * a.ids_stride) + (size_t)q * k; };
    auto first = [&](const MergeHead &x, const MergeHead &y) {
        if (MODE == MERGE_VECTOR) return scores_of(x.list)[x.pos] >= scores_of(y.list)[y.pos];
        if (MODE == MERGE_BM25) {
            const int32_t kx = total_key(scores_of(x.list)[x.pos]), ky = total_key(scores_of(y.list)[y.pos]);
            if (kx != ky) return kx > ky;
            const uint32_t sx = a.shard_order[x.list], sy = a.shard_order[y.list];
            if (sx != sy) return sx > sy;
            return ids_of(x.list)[x.pos] < ids_of(y.list)[y.pos];
        }
        const int64_t vx = reinterpret_cast<const int64_t *>(a.values + (size_t)x.list * a.values_stride)[(size_t)q * k + x.pos];
        const int64_t vy = reinterpret_cast<const int64_t *>(a.values + (size_t)y.list * a.values_stride)[(size_t)q * k + y.pos];
        return MODE == MERGE_VALUE_DESC ? vx > vy : vx < vy;
    };
    auto sift_down = [&](uint32_t index) {
        uint32_t pos = index, child = 2 * pos + 1;
        while (child + 1 < len) {
            if (first(heap[child + 1], heap[child])) child++;
            if (!first(heap[child], heap[pos])) return;
            MergeHead t = heap[pos];
            heap[pos] = heap[child];
            heap[child] = t;
            pos = child;
            child = 2 * pos + 1;
        }
        if (child + 1 == len && first(heap[child], heap[pos])) {
            MergeHead t = heap[pos];
            heap[pos] = heap[child];
            heap[child] = t;
        }
    };
    // a shard's count can never exceed the k slots its row has: an oversized value (a bad gather) must not read past the row
    auto count_of = [&](uint32_t l) {
        const uint32_t c = reinterpret_cast<const uint32_t *>(a.counts + (size_t)l * a.counts_stride)[q];
        return c < k ? c : k;
    };
    for (uint32_t l = 0; l < a.n_lists; l++)
        if (count_of(l) > 0) heap[len++] = MergeHead{l, 0};
    for (uint32_t i = len / 2; i-- > 0;) sift_down(i);
    uint32_t n = 0;
    while (len > 0 && n < limit) {
        const MergeHead h = heap[0];
        if (a.out_score) a.out_score[(size_t)q * limit + n] = scores_of(h.list)[h.pos];
        if (a.out_id) a.out_id[(size_t)q * limit + n] = ids_of(h.list)[h.pos];
        if (a.out_list) a.out_list[(size_t)q * limit + n] = h.list;
        if (a.out_value)
            a.out_value[(size_t)q * limit + n] = reinterpret_cast<const int64_t *>(a.values + (size_t)h.list * a.values_stride)[(size_t)q * k + h.pos];
        n++;
        if (h.pos + 1 < count_of(h.list)) heap[0].pos++;
        else {
            heap[0] = heap[len - 1];
            len--;
        }
        sift_down(0);
    }
    a.out_count[q] = n;
}

int32_t launch_merge_lists(const MergeListsArgs &a, int mode, hipStream_t st) {
    if (a.n_lists > MERGE_MAX_LISTS) return fail(NIDX_ERR_UNSUPPORTED, "more than %d shard lists", MERGE_MAX_LISTS);
    if (a.n_queries == 0) return NIDX_OK;
    const dim3 grid((a.n_queries + 63) / 64), block(64);
    switch (mode) {
    case MERGE_VECTOR: hipLaunchKernelGGL(merge_lists_kernel<MERGE_VECTOR>, grid, block, 0, st, a); break;
    case MERGE_BM25: hipLaunchKernelGGL(merge_lists_kernel<MERGE_BM25>, grid, block, 0, st, a); break;
    case MERGE_VALUE_DESC: hipLaunchKernelGGL(merge_lists_kernel<MERGE_VALUE_DESC>, grid, block, 0, st, a); break;
    case MERGE_VALUE_ASC: hipLaunchKernelGGL(merge_lists_kernel<MERGE_VALUE_ASC>, grid, block, 0, st, a); break;
    default: return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown merge order %d", mode);
    }
    NIDX_HIP(hipGetLastError());
    return NIDX_OK;
}

// rank of every list's shard id in bytewise order (equal ids share a rank): what `a.shard_id.cmp(&b.shard_id)` decides
int32_t shard_order_from_ids(const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n, uint32_t *order) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t la = shard_id_lens[j], lb = shard_id_lens[i], m = la < lb ? la : lb;
            int c = m ? memcmp(shard_ids[j], shard_ids[i], m) : 0;
            if (c == 0) c = (la > lb) - (la < lb);
            if (c < 0) r++;   // ids smaller than i's
        }
        order[i] = r;
    }
    return NIDX_OK;
}

}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_merge_vector_device(const float *d_scores, const uint64_t *d_ids, const uint32_t *d_counts, uint32_t n_lists,
                                     uint32_t n_queries, uint32_t k, uint32_t limit, float *d_out_score, uint64_t *d_out_id,
                                     uint32_t *d_out_count, void *stream) try {
    if (!d_scores || !d_ids || !d_counts || !d_out_score || !d_out_id || !d_out_count)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    MergeListsArgs a{};
    a.scores = reinterpret_cast<const uint8_t *>(d_scores);
    a.scores_stride = (size_t)n_queries * k * 4;
    a.ids = reinterpret_cast<const uint8_t *>(d_ids);
    a.ids_stride = (size_t)n_queries * k * 8;
    a.counts = reinterpret_cast<const uint8_t *>(d_counts);
    a.counts_stride = (size_t)n_queries * 4;
    a.n_lists = n_lists, a.n_queries = n_queries, a.k = k, a.limit = limit;
    a.out_score = d_out_score, a.out_id = d_out_id, a.out_count = d_out_count;
    return launch_merge_lists(a, MERGE_VECTOR, (hipStream_t)stream);
} NIDX_ABI_CATCH

int32_t nidx_gpu_merge_bm25_device(const float *d_scores, const uint64_t *d_docaddrs, const int64_t *d_order_values, const uint32_t *d_counts,
                                   const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists, uint32_t n_queries,
                                   uint32_t k, uint32_t limit, int32_t order, float *d_out_score, uint64_t *d_out_docaddr,
                                   int64_t *d_out_order_value, uint32_t *d_out_list, uint32_t *d_out_count, void *stream) try {
    if (!d_scores || !d_docaddrs || !d_counts || !d_out_count || (n_lists && (!shard_ids || !shard_id_lens)))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_lists > MERGE_MAX_LISTS) return fail(NIDX_ERR_UNSUPPORTED, "more than %d shard lists", MERGE_MAX_LISTS);
    if (order != NIDX_MERGE_ORDER_SCORE && !d_order_values) return fail(NIDX_ERR_INVALID_ARGUMENT, "ordering by value needs d_order_values");
    MergeListsArgs a{};
    a.scores = reinterpret_cast<const uint8_t *>(d_scores);
    a.scores_stride = (size_t)n_queries * k * 4;
    a.ids = reinterpret_cast<const uint8_t *>(d_docaddrs);
    a.ids_stride = (size_t)n_queries * k * 8;
    a.values = reinterpret_cast<const uint8_t *>(d_order_values);
    a.values_stride = (size_t)n_queries * k * 8;
    a.counts = reinterpret_cast<const uint8_t *>(d_counts);
    a.counts_stride = (size_t)n_queries * 4;
    a.n_lists = n_lists, a.n_queries = n_queries, a.k = k, a.limit = limit;
    shard_order_from_ids(shard_ids, shard_id_lens, n_lists, a.shard_order);
    a.out_score = d_out_score, a.out_id = d_out_docaddr, a.out_list = d_out_list, a.out_count = d_out_count;
    a.out_value = d_order_values ? d_out_order_value : nullptr;
    const int mode = order == NIDX_MERGE_ORDER_SCORE ? MERGE_BM25 : (order == NIDX_MERGE_ORDER_VALUE_DESC ? MERGE_VALUE_DESC : MERGE_VALUE_ASC);
    if (order < 0 || order > NIDX_MERGE_ORDER_VALUE_ASC) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown merge order %d", order);
    return launch_merge_lists(a, mode, (hipStream_t)stream);
} NIDX_ABI_CATCH

}  // extern "C"
