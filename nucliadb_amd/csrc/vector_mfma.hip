// vector_mfma.hip — batched exact k-NN as a true dense GEMM on the f32 matrix cores (gfx950).
//
// The batched form of OpenSegment::brute_force_search (nidx_vector/src/segment.rs:569-623): for a
// batch of B queries the score matrix S[B][N] = Q·Xᵀ is a GEMM, so the corpus is read ONCE per batch
// instead of once per 8-query tile (vector_scan.hip).  v_mfma_f32_32x32x2_f32 is exact f32: each
// output element is a k-ordered fmaf chain, i.e. the oracle's ORC_ORDER_SERIAL_FMA, so scores and
// top-k are bit-identical to the oracle in that order (the scan/HNSW kernels use WAVE64 order; the two
// orders differ in the last bits, which is why the method is chosen explicitly, never silently).
//
// Tiling: workgroup = 4 waves, block tile 128 queries x 128 rows, K chunk 16, LDS double-buffered
// and split into even/odd-k planes so that a lane fetches the operands of 4 consecutive MFMAs with
// one conflict-free ds_read_b128 (12-float line pitch).  Wave w owns queries 32w..32w+31 and all
// 128 rows: 4 independent 32x32 accumulators (64 VGPRs).  The epilogue of every tile compares each
// score with its query's current k-th score (cheap f32 pre-test, exact f64 cosine only on the rare
// survivors) and inserts survivors into per-query sorted lists in LDS; a block walks a stripe of row
// tiles and leaves [query][stripe][k] partial lists for merge_topk_kernel.
//
// Bound: f32 MFMA (2*B*N*D flop at 157.3 TFLOP/s); HBM traffic N*D*4 per 128-query block row.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MF_BM 128
#define MF_BN 128
#define MF_BK 16
#define MF_PITCH 12 /* floats per (row, parity) line: 8 + 4 pad => b128 reads hit 16 distinct slots */

// KMAX = 16: 70 KiB of LDS, two workgroups per CU (the k <= 16 pages the reference serves by default); KMAX = 64: the lists take
// 64 KiB, one workgroup per CU — the same tile with pages up to a full wave of ranks (and k x vectors-per-paragraph of a multi-vector
// segment, reduced to one hit per paragraph by para_best_kernel afterwards).
template <int KMAX>
struct MfmaShared {
    float q[2][2][MF_BM][MF_PITCH];  // [stage][k parity][query][k/2]
    float x[2][2][MF_BN][MF_PITCH];  // [stage][k parity][row][k/2]
    uint64_t lists[MF_BM][KMAX];     // per-query sorted top-k (rank keys), owned by the query's wave
    uint64_t thr_key[MF_BM];         // k-th key (EMPTY while the list is short)
    float thr_score[MF_BM];          // its score (-inf while short)
    float q_qq[MF_BM], q_rinv[MF_BM];
    float row_xx[MF_BN], row_rinv[MF_BN];
    uint32_t row_ok[MF_BN];
};

// serial-order norms: out[r] = fmaf chain over k ascending (the xx/yy terms of ORC_ORDER_SERIAL_FMA)
__global__ __launch_bounds__(256) void serial_norms_kernel(const float *__restrict__ rows, uint32_t n, uint32_t dp,
                                                           float *__restrict__ out) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float4 *p = reinterpret_cast<const float4 *>(rows + (size_t)r * dp);
    float acc = 0.f;
    for (uint32_t i = 0; i < dp / 4; i++) {
        float4 v = p[i];
        acc = fmaf(v.x, v.x, acc);
        acc = fmaf(v.y, v.y, acc);
        acc = fmaf(v.z, v.z, acc);
        acc = fmaf(v.w, v.w, acc);
    }
    out[r] = acc;
}

__device__ inline void stage_load(const float *base, uint32_t n_rows, uint32_t row0, uint32_t dp, uint32_t k0, int tid,
                                  float4 (&regs)[2]) {
    // 128 rows x 16 floats: thread t loads float4 #(t%4) of rows t/4 and t/4 + 64
#pragma unroll
    for (int it = 0; it < 2; it++) {
        uint32_t row = row0 + (uint32_t)(tid >> 2) + 64u * it;
        uint32_t k = k0 + 4u * (uint32_t)(tid & 3);
        if (row < n_rows && k < dp) regs[it] = *reinterpret_cast<const float4 *>(base + (size_t)row * dp + k);
        else regs[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ inline void stage_store(float (&plane)[2][MF_BM][MF_PITCH], int tid, const float4 (&regs)[2]) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
        int row = (tid >> 2) + 64 * it;
        int c = 2 * (tid & 3);
        *reinterpret_cast<float2 *>(&plane[0][row][c]) = make_float2(regs[it].x, regs[it].z);  // even k
        *reinterpret_cast<float2 *>(&plane[1][row][c]) = make_float2(regs[it].y, regs[it].w);  // odd k
    }
}

template <int KMAX>
__global__ __launch_bounds__(256, (KMAX <= 16 ? 2 : 1)) void mfma_scan_kernel(MfmaScanArgs a) {
    __shared__ MfmaShared<KMAX> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const uint32_t q0 = blockIdx.y * MF_BM;
    const bool cosine = a.similarity == 1;
    const int k = (int)a.k;
    const uint32_t n_tiles = (a.n + MF_BN - 1) / MF_BN;
    const uint32_t nk = (a.dp + MF_BK - 1) / MF_BK;

    if (tid < MF_BM) {
        uint32_t qi = q0 + tid < a.n_queries ? q0 + tid : a.n_queries - 1;
        float qq = a.q_norm2[qi];
        sh.q_qq[tid] = qq;
        sh.q_rinv[tid] = 1.0f / sqrtf(qq);
        sh.thr_key[tid] = NIDX_EMPTY_KEY;
        sh.thr_score[tid] = -INFINITY;
    }
    for (int i = tid; i < MF_BM * KMAX; i += 256) (&sh.lists[0][0])[i] = NIDX_EMPTY_KEY;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t r0 = tile * MF_BN;
        __syncthreads();  // previous tile's epilogue is done with row_* and the stage buffers
        if (tid < MF_BN) {
            uint32_t r = r0 + tid;
            bool ok = r < a.n;
            float xx = 0.f;
            if (ok) {
                uint32_t p = a.para_of_vec ? a.para_of_vec[r] : r;
                if (a.alive && !bit_test(a.alive, p)) ok = false;
                if (ok && a.filter && !bit_test(a.filter, p)) ok = false;
                xx = cosine ? a.norm2[r] : 0.f;
            }
            sh.row_ok[tid] = ok ? 1u : 0u;
            sh.row_xx[tid] = xx;
            sh.row_rinv[tid] = 1.0f / sqrtf(xx);
        }
        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

        // software pipeline: the global loads of chunk kc+2 are issued before the MFMAs of chunk kc and
        // only consumed (written to LDS) at the end of chunk kc+1 — two MFMA phases (~4k cycles) of cover
        float4 gq[2], gx[2], nq_[2], nx_[2];
        stage_load(a.queries, a.n_queries, q0, a.dp, 0, tid, gq);
        stage_load(a.vectors, a.n, r0, a.dp, 0, tid, gx);
        stage_store(sh.q[0], tid, gq);
        stage_store(sh.x[0], tid, gx);
        if (nk > 1) {
            stage_load(a.queries, a.n_queries, q0, a.dp, MF_BK, tid, gq);
            stage_load(a.vectors, a.n, r0, a.dp, MF_BK, tid, gx);
        }
        __syncthreads();
        for (uint32_t kc = 0; kc < nk; kc++) {
            const int st = (int)(kc & 1);
            if (kc + 2 < nk) {
                stage_load(a.queries, a.n_queries, q0, a.dp, (kc + 2) * MF_BK, tid, nq_);
                stage_load(a.vectors, a.n, r0, a.dp, (kc + 2) * MF_BK, tid, nx_);
            }
            // operands of the 8 k-steps of this chunk: lane (li, half) takes parity `half`
            const float4 a0 = *reinterpret_cast<const float4 *>(&sh.q[st][half][32 * wave + li][0]);
            const float4 a1 = *reinterpret_cast<const float4 *>(&sh.q[st][half][32 * wave + li][4]);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 b0 = *reinterpret_cast<const float4 *>(&sh.x[st][half][32 * t + li][0]);
                const float4 b1 = *reinterpret_cast<const float4 *>(&sh.x[st][half][32 * t + li][4]);
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int s = 0; s < 8; s++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[t], 0, 0, 0);
            }
            if (kc + 1 < nk) {
                stage_store(sh.q[st ^ 1], tid, gq);
                stage_store(sh.x[st ^ 1], tid, gx);
            }
#pragma unroll
            for (int it = 0; it < 2; it++) {
                gq[it] = nq_[it];
                gx[it] = nx_[it];
            }
            __syncthreads();
        }

        // ---- epilogue: fold the 32 x 128 scores of this wave into its queries' top-k lists ----
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = 32 * t + li;
            const uint32_t row = r0 + (uint32_t)j;
            const bool row_ok = sh.row_ok[j] != 0;
            const float rinv = sh.row_rinv[j];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int qi = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float ab = acc[t][r];
                // cheap pre-test in f32 (cosine: ab/(|x||q|) within 1e-5 of the exact value)
                const float approx = cosine ? ab * rinv * sh.q_rinv[qi] + 1e-5f : ab;
                const bool pre = row_ok && !(approx < sh.thr_score[qi]);
                unsigned long long m = __ballot(pre);
                if (!m) continue;
                float score = ab;
                if (pre && cosine) score = cosine_from_sums(ab, sh.row_xx[j], sh.q_qq[qi]);
                uint64_t key = rank_key(score, row);
                bool pass = pre && (score >= a.min_score) && key > sh.thr_key[qi];
                m = __ballot(pass);
                while (m) {
                    int src = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int sq = (int)lane_bcast_u32((uint32_t)qi, src);
                    const uint64_t nk_ = lane_bcast_u64(key, src);
                    if (!(nk_ > sh.thr_key[sq])) continue;  // an earlier insert of this round raised the bar
                    WaveSortedList l;
                    l.key = lane < KMAX ? sh.lists[sq][lane] : NIDX_EMPTY_KEY;
                    l.insert(nk_, lane);
                    if (lane < k) sh.lists[sq][lane] = l.key;
                    uint64_t kth = l.at(k - 1);
                    if (lane == 0 && kth != NIDX_EMPTY_KEY) {
                        sh.thr_key[sq] = kth;
                        sh.thr_score[sq] = rank_key_score(kth);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- partial lists: [query][stripe][k] ----
    for (int i = tid; i < MF_BM * k; i += 256) {
        int q = i / k, e = i % k;
        if (q0 + q < a.n_queries) a.partial[((size_t)(q0 + q) * gridDim.x + blockIdx.x) * k + e] = sh.lists[q][e];
    }
}

hipError_t launch_serial_norms(const float *rows, uint32_t n, uint32_t dp, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(serial_norms_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rows, n, dp, out);
    return hipGetLastError();
}

uint32_t mfma_scan_stripes(uint32_t n, uint32_t n_queries) {
    uint32_t tiles = (n + MF_BN - 1) / MF_BN, qb = (n_queries + MF_BM - 1) / MF_BM;
    uint32_t s = 512 / (qb ? qb : 1);
    if (s < 1) s = 1;
    if (s > tiles) s = tiles;
    return s ? s : 1;
}

hipError_t launch_mfma_scan(const MfmaScanArgs &a, uint32_t stripes, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    const dim3 grid(stripes, (a.n_queries + MF_BM - 1) / MF_BM);
    if (a.k <= 16) hipLaunchKernelGGL(mfma_scan_kernel<16>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mfma_scan_kernel<NIDX_MFMA_KMAX>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace nidx
