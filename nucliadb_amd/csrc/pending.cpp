// pending.cpp — entry points declared in include/nidx_gpu.h whose kernels are not written yet.
// They fail loudly (NIDX_ERR_UNSUPPORTED); nothing falls back to a CPU path.
#include "host_common.h"

using namespace nidx;

extern "C" {

int32_t nidx_gpu_bm25_open(const nidx_gpu_bm25_segment_t *, uint32_t, nidx_gpu_bm25_index_t **) {
    return fail(NIDX_ERR_UNSUPPORTED, "nidx_gpu_bm25_open: BM25 kernels not implemented yet");
}
void nidx_gpu_bm25_close(nidx_gpu_bm25_index_t *) {}
int32_t nidx_gpu_bm25_space_usage(const nidx_gpu_bm25_index_t *, uint64_t *) {
    return fail(NIDX_ERR_UNSUPPORTED, "BM25 kernels not implemented yet");
}
int32_t nidx_gpu_bm25_search(nidx_gpu_bm25_index_t *, const nidx_gpu_bm25_clause_t *, const uint64_t *, uint32_t, uint32_t,
                             const nidx_gpu_bm25_search_after_t *, uint64_t *, float *, uint32_t *, uint64_t *, uint64_t *) {
    return fail(NIDX_ERR_UNSUPPORTED, "BM25 kernels not implemented yet");
}
float nidx_gpu_bm25_idf(uint64_t, uint64_t) { return 0.f; }
uint32_t nidx_gpu_fieldnorm_from_id(uint8_t) { return 0; }
uint8_t nidx_gpu_fieldnorm_to_id(uint32_t) { return 0; }
}
