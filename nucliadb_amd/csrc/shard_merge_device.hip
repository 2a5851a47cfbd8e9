// shard_merge_device.hip — merge_vector_responses (nidx/src/searcher/shard_merge.rs:332-348) for a
// whole query batch on the device: after the RCCL all-gather every GPU holds the P per-shard top-k
// lists of each query; one thread per query runs the same k-way merge the host entry point runs
// (itertools::kmerge_by(|a, b| a.score >= b.score).take(limit): binary heap of list heads,
// sift_down after every pop), so ties resolve exactly as in nidx_gpu_merge_vector.
#include "device_common.h"
#include "host_common.h"

namespace nidx {

#define MERGE_MAX_LISTS 64

struct MergeHead {
    uint32_t list, pos;
};

__global__ void merge_vector_kernel(const float *__restrict__ scores,    // [P][B][k]
                                    const uint64_t *__restrict__ ids,     // [P][B][k]
                                    const uint32_t *__restrict__ counts,  // [P][B]
                                    uint32_t P, uint32_t B, uint32_t k, uint32_t limit,
                                    float *__restrict__ out_score, uint64_t *__restrict__ out_id,
                                    uint32_t *__restrict__ out_count) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    MergeHead heap[MERGE_MAX_LISTS];
    uint32_t len = 0;
    auto score_of = [&](const MergeHead &h) { return scores[((size_t)h.list * B + q) * k + h.pos]; };
    auto first = [&](const MergeHead &a, const MergeHead &b) { return score_of(a) >= score_of(b); };
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
    auto count_of = [&](uint32_t l) { const uint32_t c = counts[(size_t)l * B + q]; return c < k ? c : k; };
    for (uint32_t l = 0; l < P; l++)
        if (count_of(l) > 0) heap[len++] = MergeHead{l, 0};
    for (uint32_t i = len / 2; i-- > 0;) sift_down(i);
    uint32_t n = 0;
    while (len > 0 && n < limit) {
        MergeHead h = heap[0];
        out_score[(size_t)q * limit + n] = score_of(h);
        out_id[(size_t)q * limit + n] = ids[((size_t)h.list * B + q) * k + h.pos];
        n++;
        if (h.pos + 1 < count_of(h.list)) heap[0].pos++;
        else {
            heap[0] = heap[len - 1];
            len--;
        }
        sift_down(0);
    }
    out_count[q] = n;
}

}  // namespace nidx

using namespace nidx;

extern "C" int32_t nidx_gpu_merge_vector_device(const float *d_scores, const uint64_t *d_ids, const uint32_t *d_counts,
                                                uint32_t n_lists, uint32_t n_queries, uint32_t k, uint32_t limit,
                                                float *d_out_score, uint64_t *d_out_id, uint32_t *d_out_count,
                                                void *stream) {
    if (!d_scores || !d_ids || !d_counts || !d_out_score || !d_out_id || !d_out_count)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_lists > MERGE_MAX_LISTS) return fail(NIDX_ERR_UNSUPPORTED, "more than %d shard lists", MERGE_MAX_LISTS);
    if (n_queries == 0) return NIDX_OK;
    hipLaunchKernelGGL(merge_vector_kernel, dim3((n_queries + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_scores,
                       d_ids, d_counts, n_lists, n_queries, k, limit, d_out_score, d_out_id, d_out_count);
    NIDX_HIP(hipGetLastError());
    return NIDX_OK;
}
