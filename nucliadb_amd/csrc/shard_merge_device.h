// shard_merge_device.h — arguments of the batched k-way merge kernel (shard_merge_device.hip), shared with the RCCL exchange
// (shard_comm.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nidx {

#define MERGE_MAX_LISTS 64

struct MergeListsArgs {
    // list l of array X lives at X + l * X_stride (bytes): rows [n_queries][k] (counts: [n_queries])
    const uint8_t *scores;   // f32
    size_t scores_stride;
    const uint8_t *ids;      // u64
    size_t ids_stride;
    const uint8_t *values;   // i64 sort values (date orders), may be nullptr
    size_t values_stride;
    const uint8_t *counts;   // u32
    size_t counts_stride;
    uint32_t n_lists, n_queries, k, limit;
    uint32_t shard_order[MERGE_MAX_LISTS];   // bm25: rank of list l's shard id in bytewise order
    float *out_score;        // [n_queries][limit], each may be nullptr
    uint64_t *out_id;
    int64_t *out_value;
    uint32_t *out_list;
    uint32_t *out_count;     // [n_queries]
};

// mode: 0 vector, 1 bm25 by score, 2 by value descending, 3 by value ascending
int32_t launch_merge_lists(const MergeListsArgs &a, int mode, hipStream_t st);
int32_t shard_order_from_ids(const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n, uint32_t *order);

}  // namespace nidx
