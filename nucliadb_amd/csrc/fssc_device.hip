// fssc_device.hip — Fssc on the device: the merge of the segments' hits of one query (gfx950).
//
// Replaces the loop `for segment in segments { for result in segment.search(..) { ffsv.add(..) } }` + Fssc::add /
// into_iter of nidx_vector/src/searcher.rs:149-199,270-287 for a whole batch: one wave per query replays the reference's
// SEQUENTIAL insertion — segments in index order, a segment's hits best first —
//   seen      vector bytes already offered (with_duplicates == false): a candidate whose bytes were seen is dropped; equal bytes
//             imply equal score bits for one query, so rows are compared only on a bit-identical score;
//   eviction  the buffer holds k hits: a candidate must beat the LOWEST score strictly (first of equal lowest), which leaves by
//             swap-remove — the buffer order is part of the semantics (it decides which of two equal lowest scores goes);
//   identity  one hit per paragraph key (key_ids, or (segment, paragraph) when the segment carries none) — checked AFTER the
//             eviction, like the reference;
// then the stable sort by score (ties: segment, vector — the order csrc/vector_index.cpp: fssc_merge and the oracle fix).
// The buffer lives in LDS (k <= NIDX_K_MAX = 512: 12 KiB per wave); the `seen` list in HBM scratch.  Integer / compare work on
// <= n_segments * k candidates per query: microseconds per batch; what it buys is ONE small device-to-host transfer per batch
// (k hits per query) instead of one block per segment, and no host loop over n_queries * n_segments * k candidates.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

__global__ __launch_bounds__(64) void fssc_merge_kernel(FsscArgs a) {
    __shared__ float b_score[NIDX_K_MAX];
    __shared__ uint32_t b_seg[NIDX_K_MAX], b_vec[NIDX_K_MAX], b_para[NIDX_K_MAX];
    __shared__ unsigned long long b_key[NIDX_K_MAX];
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x, k = a.k;
    uint32_t len = 0, n_off = 0;
    uint32_t *off = a.offered ? a.offered + (size_t)q * a.offered_stride * 3 : nullptr;
    for (uint32_t s = 0; s < a.n_segs; s++) {
        const FsscSegDev sg = a.segs[s];
        if (!sg.count) continue;   // not searched: nothing of it could match
        const uint32_t cnt = sg.count[q];
        for (uint32_t i = 0; i < cnt; i++) {
            const float sc = sg.score[(size_t)q * k + i];
            const uint32_t v = sg.vec[(size_t)q * k + i];
            const uint32_t para = sg.para_of_vec ? sg.para_of_vec[v] : v;
            const unsigned long long key = sg.key_ids ? sg.key_ids[para] : (((unsigned long long)s << 32) | para);
            if (!a.with_duplicates) {
                bool dup = false;
                for (uint32_t base = 0; base < n_off && !dup; base += 64) {
                    const uint32_t j = base + (uint32_t)lane;
                    unsigned long long m = __ballot(j < n_off && off[3 * j] == __float_as_uint(sc));
                    while (m && !dup) {
                        const uint32_t jj = base + (uint32_t)(__ffsll((long long)m) - 1);
                        m &= m - 1;
                        const uint32_t os = off[3 * jj + 1], ov = off[3 * jj + 2];
                        const FsscSegDev og = a.segs[os];
                        const float *ra = og.vectors + (size_t)ov * og.dp, *rb = sg.vectors + (size_t)v * sg.dp;
                        bool eq = true;
                        for (uint32_t e = (uint32_t)lane; e < a.dim; e += 64) eq = eq && __float_as_uint(ra[e]) == __float_as_uint(rb[e]);
                        dup = __all(eq);
                    }
                }
                if (dup) continue;
                if (lane == 0) {
                    off[3 * n_off] = __float_as_uint(sc);
                    off[3 * n_off + 1] = s;
                    off[3 * n_off + 2] = v;
                }
                n_off++;
                __threadfence_block();   // the other lanes read the list on the next candidate
            }
            if (len == k) {
                // the lowest score strictly below the candidate's, the first of equal ones
                float best = 0.f;
                int bi = -1;
                for (uint32_t j = (uint32_t)lane; j < len; j += 64) {
                    const float bs = b_score[j];
                    if (sc > bs && (bi < 0 || bs < best)) best = bs, bi = (int)j;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (oi >= 0 && (bi < 0 || ob < best || (ob == best && oi < bi))) best = ob, bi = oi;
                }
                if (bi < 0) continue;
                __syncthreads();
                if (lane == 0) {   // swap-remove
                    b_score[bi] = b_score[len - 1], b_seg[bi] = b_seg[len - 1], b_vec[bi] = b_vec[len - 1];
                    b_para[bi] = b_para[len - 1], b_key[bi] = b_key[len - 1];
                }
                len--;
                __syncthreads();
            }
            bool present = false;
            for (uint32_t j = (uint32_t)lane; j < len; j += 64) present = present || b_key[j] == key;
            if (__any(present)) continue;
            __syncthreads();
            if (lane == 0) b_score[len] = sc, b_seg[len] = s, b_vec[len] = v, b_para[len] = para, b_key[len] = key;
            len++;
            __syncthreads();
        }
    }
    // sort desc by score; equal scores by (segment, vector)
    for (uint32_t e = (uint32_t)lane; e < len; e += 64) {
        const float se = b_score[e];
        const uint32_t ge = b_seg[e], ve = b_vec[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < len; j++) {
            const float sj = b_score[j];
            const bool before = sj > se || (!(sj < se) && (b_seg[j] < ge || (b_seg[j] == ge && b_vec[j] < ve)));
            rank += (j != e && before) ? 1u : 0u;
        }
        a.out_seg[(size_t)q * k + rank] = ge;
        a.out_para[(size_t)q * k + rank] = b_para[e];
        a.out_vec[(size_t)q * k + rank] = ve;
        a.out_score[(size_t)q * k + rank] = se;
    }
    if (lane == 0) a.out_count[q] = len;
}

hipError_t launch_fssc_merge(const FsscArgs &a, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (a.k == 0 || a.k > NIDX_K_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fssc_merge_kernel, dim3(a.nq), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace nidx
