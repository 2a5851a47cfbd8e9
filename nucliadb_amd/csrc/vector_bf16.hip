// vector_bf16.hip — batched brute-force FALLBACK on the bf16 matrix cores with exact f32 re-scoring
// (gfx950).  BASELINE.json configs[4]; SURVEY.md §7 step 5.
//
// Stage 1 (bf16_scan_kernel): S~ = Q~·X~ᵀ with bf16 copies of queries and corpus on
// v_mfma_f32_32x32x16_bf16 (f32 accumulate), fused per-query top-K' by the APPROXIMATE score
// (K' = 32 candidates per query and stripe).  The bf16 corpus copy is half the bytes of the f32 one
// and the matrix cores run 16x the f32 rate, so the corpus streams once per batch at ~HBM speed.
// Stage 2 (merge_topk_kernel, K' per query) + Stage 3 (rescore_select_kernel): the K' survivors of a
// query are re-scored from the f32 rows in the WAVE64 order of the scan/HNSW kernels and the final
// top-k is selected by those exact scores.  Returned scores are therefore bit-identical to
// brute_force_search's for the returned ids; the id set equals the exact top-k unless a true top-k
// row fell outside the approximate top-K' (bf16 rounding: relative 2^-9 per operand) — recall is
// measured, not assumed (tests/test_vector_gpu.py).  This method is explicit
// (NIDX_METHOD_BRUTE_FORCE_BF16), never chosen by the cost model.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BF_BM 128
#define BF_BN 128
#define BF_BK 32                 /* bf16 elements per chunk = 64 B per row */
#define BF_PITCH_B 80            /* bytes per LDS row: 64 + 16 pad => b128 reads hit 16 distinct slots (5i mod 16) */
#define BF_KP NIDX_BF16_CAND     /* candidates kept per query */

struct Bf16Shared {
    unsigned char q[2][BF_BM][BF_PITCH_B];
    unsigned char x[2][BF_BN][BF_PITCH_B];
    uint64_t lists[BF_BM][BF_KP];
    uint64_t thr_key[BF_BM];
    float thr_score[BF_BM];
    float q_rinv[BF_BM];
    float row_rinv[BF_BN];
    uint32_t row_ok[BF_BN];
};

__device__ inline unsigned short f32_to_bf16_rne(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// rows [n][dp] f32 -> [n][dp16] bf16 (dp16 = dp rounded up to 64, zero padded)
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *__restrict__ in, uint32_t n, uint32_t dp, uint32_t dp16,
                                                      unsigned short *__restrict__ out) {
    const size_t total = (size_t)n * dp16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r = (uint32_t)(i / dp16), c = (uint32_t)(i % dp16);
        out[i] = c < dp ? f32_to_bf16_rne(in[(size_t)r * dp + c]) : (unsigned short)0;
    }
}

__device__ inline void bf_stage_load(const unsigned short *base, uint32_t n_rows, uint32_t row0, uint32_t dp16, uint32_t k0,
                                     int tid, uint4 (&regs)[2]) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        uint32_t row = row0 + (uint32_t)(tid >> 2) + 64u * it;
        uint32_t k = k0 + 8u * (uint32_t)(tid & 3);
        if (row < n_rows && k < dp16) regs[it] = *reinterpret_cast<const uint4 *>(base + (size_t)row * dp16 + k);
        else regs[it] = make_uint4(0, 0, 0, 0);
    }
}
__device__ inline void bf_stage_store(unsigned char (&tile)[BF_BM][BF_PITCH_B], int tid, const uint4 (&regs)[2]) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        int row = (tid >> 2) + 64 * it;
        *reinterpret_cast<uint4 *>(&tile[row][16 * (tid & 3)]) = regs[it];
    }
}

__global__ __launch_bounds__(256, 2) void bf16_scan_kernel(Bf16ScanArgs a) {
    __shared__ Bf16Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const uint32_t q0 = blockIdx.y * BF_BM;
    const bool cosine = a.similarity == 1;
    const uint32_t n_tiles = (a.n + BF_BN - 1) / BF_BN;
    const uint32_t nk = a.dp16 / BF_BK;

    if (tid < BF_BM) {
        uint32_t qi = q0 + tid < a.n_queries ? q0 + tid : a.n_queries - 1;
        sh.q_rinv[tid] = cosine ? 1.0f / sqrtf(a.q_norm2[qi]) : 1.0f;
        sh.thr_key[tid] = NIDX_EMPTY_KEY;
        sh.thr_score[tid] = -INFINITY;
    }
    for (int i = tid; i < BF_BM * BF_KP; i += 256) (&sh.lists[0][0])[i] = NIDX_EMPTY_KEY;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t r0 = tile * BF_BN;
        __syncthreads();
        if (tid < BF_BN) {
            uint32_t r = r0 + tid;
            bool ok = r < a.n;
            float rinv = 1.0f;
            if (ok) {
                uint32_t p = a.para_of_vec ? a.para_of_vec[r] : r;
                if (a.alive && !bit_test(a.alive, p)) ok = false;
                if (ok && a.filter && !bit_test(a.filter, p)) ok = false;
                if (cosine) rinv = 1.0f / sqrtf(a.norm2[r]);
            }
            sh.row_ok[tid] = ok ? 1u : 0u;
            sh.row_rinv[tid] = rinv;
        }
        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

        uint4 gq[2], gx[2], nq_[2], nx_[2];
        bf_stage_load(a.queries16, a.n_queries, q0, a.dp16, 0, tid, gq);
        bf_stage_load(a.vectors16, a.n, r0, a.dp16, 0, tid, gx);
        bf_stage_store(sh.q[0], tid, gq);
        bf_stage_store(sh.x[0], tid, gx);
        if (nk > 1) {
            bf_stage_load(a.queries16, a.n_queries, q0, a.dp16, BF_BK, tid, gq);
            bf_stage_load(a.vectors16, a.n, r0, a.dp16, BF_BK, tid, gx);
        }
        __syncthreads();
        for (uint32_t kc = 0; kc < nk; kc++) {
            const int st = (int)(kc & 1);
            if (kc + 2 < nk) {
                bf_stage_load(a.queries16, a.n_queries, q0, a.dp16, (kc + 2) * BF_BK, tid, nq_);
                bf_stage_load(a.vectors16, a.n, r0, a.dp16, (kc + 2) * BF_BK, tid, nx_);
            }
            // BF_BK/16 k-steps of 16: lane (li, half) supplies k = 16*s + 8*half .. +7 of its row
            bf16x8 av[BF_BK / 16];
#pragma unroll
            for (int s = 0; s < BF_BK / 16; s++)
                av[s] = *reinterpret_cast<const bf16x8 *>(&sh.q[st][32 * wave + li][32 * s + 16 * half]);
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int s = 0; s < BF_BK / 16; s++) {
                    bf16x8 bv = *reinterpret_cast<const bf16x8 *>(&sh.x[st][32 * t + li][32 * s + 16 * half]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s], bv, acc[t], 0, 0, 0);
                }
            }
            if (kc + 1 < nk) {
                bf_stage_store(sh.q[st ^ 1], tid, gq);
                bf_stage_store(sh.x[st ^ 1], tid, gx);
            }
#pragma unroll
            for (int it = 0; it < 2; it++) {
                gq[it] = nq_[it];
                gx[it] = nx_[it];
            }
            __syncthreads();
        }

        // ---- epilogue: approximate scores -> per-query candidate lists ----
        // per-lane copies of the 16 queries' 1/|q| and current thresholds (a stale threshold only lets a
        // few extra candidates through to the exact key test below)
        float qinv[16], thr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int qi = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            qinv[r] = sh.q_rinv[qi];
            thr[r] = sh.thr_score[qi];
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = 32 * t + li;
            const uint32_t row = r0 + (uint32_t)j;
            const bool row_ok = sh.row_ok[j] != 0;
            const float rinv = sh.row_rinv[j];
            uint32_t mask = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float approx = acc[t][r] * rinv * qinv[r];
                mask |= (row_ok && approx > thr[r]) ? (1u << r) : 0u;
            }
            if (!__ballot(mask != 0)) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                unsigned long long m = __ballot((mask >> r) & 1u);
                if (!m) continue;
                const int qi = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                const uint64_t key = rank_key(acc[t][r] * rinv * qinv[r], row);
                while (m) {
                    int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int sq = __shfl(qi, src, 64);
                    const uint64_t nk_ = shfl_u64(key, src);
                    if (!(nk_ > sh.thr_key[sq])) continue;
                    WaveSortedList l;
                    l.key = lane < BF_KP ? sh.lists[sq][lane] : NIDX_EMPTY_KEY;
                    l.insert(nk_, lane);
                    if (lane < BF_KP) sh.lists[sq][lane] = l.key;
                    uint64_t kth = l.at(BF_KP - 1);
                    if (lane == 0 && kth != NIDX_EMPTY_KEY) {
                        sh.thr_key[sq] = kth;
                        sh.thr_score[sq] = rank_key_score(kth);
                    }
                }
                // refresh this lane's view of the thresholds it just may have raised
                thr[r] = sh.thr_score[qi];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < BF_BM * BF_KP; i += 256) {
        int q = i / BF_KP, e = i % BF_KP;
        if (q0 + q < a.n_queries) a.partial[((size_t)(q0 + q) * gridDim.x + blockIdx.x) * BF_KP + e] = sh.lists[q][e];
    }
}

// Stage 3: one wave per query re-scores its candidates exactly (WAVE64 order) and keeps the best k.
template <int NJ>
__global__ __launch_bounds__(256) void rescore_select_kernel(RescoreArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= a.n_queries) return;
    const bool cosine = a.similarity == 1;
    float4 qv[NJ];
    float qacc = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        qv[j] = load_row_chunk(a.queries + (size_t)q * a.dp, a.dp, j, lane);
        qacc = fma4(qv[j], qv[j], qacc);
    }
    const float qq = wave_butterfly_sum(qacc);
    const uint32_t n_cand = a.cand_count[q];
    WaveSortedList top;
    top.init();
    for (uint32_t c = 0; c < n_cand; c++) {
        const uint32_t row = a.cand_vec[(size_t)q * a.n_cand_max + c];
        const float *r = a.vectors + (size_t)row * a.dp;
        float ab = 0.f, xx = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            float4 x = load_row_chunk(r, a.dp, j, lane);
            ab = fma4(x, qv[j], ab);
            xx = fma4(x, x, xx);
        }
        ab = wave_butterfly_sum(ab);
        float score = ab;
        if (cosine) {
            xx = wave_butterfly_sum(xx);
            score = cosine_from_sums(ab, xx, qq);
        }
        if (score >= a.min_score) top.insert(rank_key(score, row), lane);
    }
    const int k = (int)a.k;
    const bool v = top.key != NIDX_EMPTY_KEY && lane < k;
    unsigned long long vm = __ballot(v);
    if (lane < k) {
        a.out_vec[(size_t)q * k + lane] = v ? rank_key_addr(top.key) : 0xffffffffu;
        a.out_score[(size_t)q * k + lane] = v ? rank_key_score(top.key) : 0.f;
    }
    if (lane == 0) a.out_count[q] = (uint32_t)__popcll(vm);
}

hipError_t launch_to_bf16(const float *in, uint32_t n, uint32_t dp, uint32_t dp16, unsigned short *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t total = (size_t)n * dp16;
    uint32_t blocks = (uint32_t)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, s, in, n, dp, dp16, out);
    return hipGetLastError();
}

uint32_t bf16_scan_stripes(uint32_t n, uint32_t n_queries) {
    uint32_t tiles = (n + BF_BN - 1) / BF_BN, qb = (n_queries + BF_BM - 1) / BF_BM;
    uint32_t s = 512 / (qb ? qb : 1);  // two workgroups per CU
    if (s < 1) s = 1;
    if (s > tiles) s = tiles;
    return s ? s : 1;
}

hipError_t launch_bf16_scan(const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(bf16_scan_kernel, dim3(stripes, (a.n_queries + BF_BM - 1) / BF_BM), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_rescore_select(const RescoreArgs &a, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    const dim3 grid((a.n_queries + 3) / 4), block(256);
    int nj = (int)((a.dp + 255u) / 256u);
#define NIDX_RS_CASE(N) hipLaunchKernelGGL((rescore_select_kernel<N>), grid, block, 0, s, a); return hipGetLastError()
    if (nj <= 1) { NIDX_RS_CASE(1); }
    if (nj <= 2) { NIDX_RS_CASE(2); }
    if (nj <= 3) { NIDX_RS_CASE(3); }
    if (nj <= 4) { NIDX_RS_CASE(4); }
    if (nj <= 6) { NIDX_RS_CASE(6); }
    if (nj <= 8) { NIDX_RS_CASE(8); }
    if (nj <= 12) { NIDX_RS_CASE(12); }
#undef NIDX_RS_CASE
    return hipErrorInvalidValue;
}

}  // namespace nidx
