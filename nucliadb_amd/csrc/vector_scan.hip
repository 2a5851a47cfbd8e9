// vector_scan.hip — exact k-NN scan kernels (gfx950), WAVE64 summation order.
//
// Replaces OpenSegment::brute_force_search (nidx_vector/src/segment.rs:569-623) and the
// dense_f32::{dot,cosine}_similarity calls under it (vector_types/dense_f32.rs:29-39):
//   for every alive/filter-passing paragraph: similarity(query, vector); keep >= min_score;
//   order by (score desc, address asc); take k.
//
// Layout: vectors[N][Dp] f32 in HBM, Dp = dimension rounded up to 4 floats, zero padded, so one
// wave reads one row as Dp/256 fully coalesced 1 KiB transactions (16 B per lane).  A tile of QT
// queries lives in registers (QT*NJ float4 per lane); the row is read once per tile.  The QT dot
// products of a row are reduced with a transposed butterfly (one shuffle per PAIR of queries at
// the first log2(QT) levels) that is bit-identical to a per-query xor butterfly.  Every wave keeps
// a sorted top-k per query one-entry-per-lane; a block merges its 4 waves through LDS; a second
// kernel merges the per-block lists.
//
// Bound: HBM.  Algorithmic bytes per (row, query tile) = 4*D.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

// ---------------------------------------------------------------------------------------------
// row norms: norm2[r] = sum x^2 in WAVE64 order (the `xx` term of the oracle's orc_sums)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ vectors, uint32_t n, uint32_t dp,
                                                        float *__restrict__ norm2) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nj = (int)((dp + 255u) / 256u);
    for (uint32_t r = wave; r < n; r += nwaves) {
        const float *row = vectors + (size_t)r * dp;
        float acc = 0.f;
        for (int j = 0; j < nj; j++) {
            float4 x = load_row_chunk(row, dp, j, lane);
            acc = fma4(x, x, acc);
        }
        acc = wave_butterfly_sum(acc);
        if (lane == 0) norm2[r] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// pairwise similarity (a1 numerics check): out[i] = sim(x[i], y[i]), one wave per pair
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_similarity_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                              uint32_t n, uint32_t dp, int similarity,
                                                              float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nj = (int)((dp + 255u) / 256u);
    for (uint32_t r = wave; r < n; r += nwaves) {
        float ab = 0.f, xx = 0.f, yy = 0.f;
        for (int j = 0; j < nj; j++) {
            float4 a = load_row_chunk(x + (size_t)r * dp, dp, j, lane);
            float4 b = load_row_chunk(y + (size_t)r * dp, dp, j, lane);
            // (ab, xx, yy) interleaved per component exactly like orc_sums' wave64 order
            ab = fma4(a, b, ab);
            xx = fma4(a, a, xx);
            yy = fma4(b, b, yy);
        }
        ab = wave_butterfly_sum(ab);
        xx = wave_butterfly_sum(xx);
        yy = wave_butterfly_sum(yy);
        if (lane == 0) out[r] = similarity == 1 ? cosine_from_sums(ab, xx, yy) : ab;
    }
}

// ---------------------------------------------------------------------------------------------
// scan + per-block top-k
// ---------------------------------------------------------------------------------------------
template <int NJ, int QT, int KL>
__device__ inline void scan_topk_body(const ScanArgs &a) {
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;  // wave in block
    const uint32_t tile = blockIdx.y;
    const uint32_t q0 = tile * QT;
    const uint32_t wave = blockIdx.x * 4 + wib;
    const uint32_t nwaves = gridDim.x * 4;
    const int myq = QReduce<QT>::query_of_lane(lane);
    const bool cosine = a.similarity == 1;

    // query tile -> registers; queries past n_queries replicate the last one (their lists are dropped)
    float4 qv[QT][NJ];
    double sqrt_qq = 0.0;  // sqrt(|q|^2) of this lane's query
    float qq_mine = 0.f;
    {
        float qq[QT];
#pragma unroll
        for (int q = 0; q < QT; q++) {
            uint32_t qi = q0 + q < a.n_queries ? q0 + q : a.n_queries - 1;
            const float *qrow = a.queries + (size_t)qi * a.dp;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                qv[q][j] = load_row_chunk(qrow, a.dp, j, lane);
                acc = fma4(qv[q][j], qv[q][j], acc);
            }
            qq[q] = acc;
        }
        if (cosine) {
            qq_mine = QReduce<QT>::run(qq, lane);
            sqrt_qq = sqrt((double)qq_mine);
        }
    }

    WaveTopK<KL> top[QT];  // k <= 64*KL
#pragma unroll
    for (int q = 0; q < QT; q++) top[q].init();
    uint64_t thr = NIDX_EMPTY_KEY;  // k-th key of this lane's query (EMPTY while the list is short)
    const int k = (int)a.k;

    auto passes = [&](uint32_t r) -> bool {
        uint32_t p = a.para_of_vec ? a.para_of_vec[r] : r;
        if (a.alive && !bit_test(a.alive, p)) return false;
        if (a.filter && !bit_test(a.filter, p)) return false;
        return true;
    };

    uint32_t r = wave;
    while (r < a.n && !passes(r)) r += nwaves;
    float4 cur[NJ];
    if (r < a.n) {
#pragma unroll
        for (int j = 0; j < NJ; j++) cur[j] = load_row_chunk(a.vectors + (size_t)r * a.dp, a.dp, j, lane);
    }
    while (r < a.n) {
        uint32_t rn = r + nwaves;
        while (rn < a.n && !passes(rn)) rn += nwaves;
        float4 nxt[NJ];
        if (rn < a.n) {
#pragma unroll
            for (int j = 0; j < NJ; j++) nxt[j] = load_row_chunk(a.vectors + (size_t)rn * a.dp, a.dp, j, lane);
        }
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; q++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s = fma4(cur[j], qv[q][j], s);
            acc[q] = s;
        }
        float ab = QReduce<QT>::run(acc, lane);
        float score;
        if (cosine) {
            float xx = a.norm2[r];
            // cosine_from_sums with sqrt(|q|^2) hoisted (same f64 operations, same order)
            double dab = (double)ab, dxx = (double)xx;
            double dist;
            if (dxx == 0.0 && (double)qq_mine == 0.0) dist = 0.0;
            else if (dab == 0.0) dist = 1.0;
            else {
                double d = 1.0 - dab / (sqrt(dxx) * sqrt_qq);
                dist = d > 0.0 ? d : 0.0;
            }
            score = 1.0f - (float)dist;
        } else {
            score = ab;
        }
        uint64_t ck = rank_key(score, r);
        bool ok = (score >= a.min_score) && (ck > thr) && ((lane & QReduce<QT>::group_mask()) == 0);
        unsigned long long m = __ballot(ok);
        while (m) {
            int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            uint64_t nk = lane_bcast_u64(ck, src);
            int q = QReduce<QT>::query_of_lane(src);
#pragma unroll
            for (int qq = 0; qq < QT; qq++) {
                if (qq == q) {
                    uint64_t kth = top[qq].insert_kth(nk, k, lane);
                    if (myq == qq) thr = kth;
                }
            }
        }
        r = rn;
#pragma unroll
        for (int j = 0; j < NJ; j++) cur[j] = nxt[j];
    }

    // block merge through LDS: waves 1..3 publish, wave 0 folds them in
    __shared__ uint64_t lds[3][QT][64 * KL];
    if (wib > 0) {
#pragma unroll
        for (int q = 0; q < QT; q++)
#pragma unroll
            for (int i = 0; i < KL; i++) lds[wib - 1][q][64 * i + lane] = top[q].mine(i);
    }
    __syncthreads();
    if (wib == 0) {
#pragma unroll
        for (int q = 0; q < QT; q++) {
            uint64_t kth = top[q].at(k - 1);
            for (int w = 0; w < 3; w++) {
                for (int i = 0; i < k; i++) {
                    uint64_t nk = lds[w][q][i];
                    if (nk == NIDX_EMPTY_KEY) break;
                    if (nk > kth) kth = top[q].insert_kth(nk, k, lane);
                    else break;  // lists are sorted: the rest rank even lower
                }
            }
            if (q0 + q < a.n_queries) {
#pragma unroll
                for (int i = 0; i < KL; i++)
                    if (64 * i + lane < k) a.partial[((size_t)(q0 + q) * gridDim.x + blockIdx.x) * k + 64 * i + lane] = top[q].mine(i);
            }
        }
    }
}

template <int NJ, int QT, int KL>
__global__ __launch_bounds__(256) void scan_topk_kernel(ScanArgs a) {
    scan_topk_body<NJ, QT, KL>(a);
}
// The exact scans of SEVERAL segments in one launch (a multi-segment index under a selective filter: OpenSegment::_search routes
// every segment to brute force, segment.rs:506-555 — a launch pair per segment was 100 launches per batch on the reference's
// 50-segment layout): blockIdx.z names the segment, whose arguments come from a table in HBM (uniform address: scalar loads).
// Every record carries the same launch shape (dp, k, query tile, n_queries, grid); a block past a small segment's rows leaves
// empty lists.
template <int NJ, int QT, int KL>
__global__ __launch_bounds__(256) void scan_topk_segments_kernel(const ScanArgs *table) {
    scan_topk_body<NJ, QT, KL>(table[blockIdx.z]);
}

// merge per-block lists: one block per query
template <int KL>
__device__ inline void merge_topk_body(const uint64_t *__restrict__ partial, uint32_t lists_per_query,
                                       uint32_t k, uint32_t *__restrict__ out_vec,
                                       float *__restrict__ out_score,
                                       uint32_t *__restrict__ out_count) {
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint64_t *src = partial + (size_t)q * lists_per_query * k;
    const uint32_t total = lists_per_query * k;
    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    for (uint32_t base = wib * 64; base < total; base += 256) {
        uint32_t i = base + lane;
        uint64_t ck = i < total ? src[i] : NIDX_EMPTY_KEY;
        unsigned long long m = __ballot(ck > kth);
        while (m) {
            int s = __ffsll((long long)m) - 1;
            m &= m - 1;
            uint64_t nk = shfl_u64(ck, s);
            if (nk > kth) kth = top.insert_kth(nk, (int)k, lane);
        }
    }
    __shared__ uint64_t lds[3][64 * KL];
    if (wib > 0) {
#pragma unroll
        for (int i = 0; i < KL; i++) lds[wib - 1][64 * i + lane] = top.mine(i);
    }
    __syncthreads();
    if (wib == 0) {
        for (int w = 0; w < 3; w++)
            for (uint32_t i = 0; i < k; i++) {
                uint64_t nk = lds[w][i];
                if (nk == NIDX_EMPTY_KEY) break;
                if (nk > kth) kth = top.insert_kth(nk, (int)k, lane);
                else break;
            }
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < KL; i++) {
            const uint32_t e = 64 * i + lane;
            const uint64_t key = top.mine(i);
            const bool v = key != NIDX_EMPTY_KEY && e < k;
            cnt += (uint32_t)__popcll(__ballot(v));
            if (e < k) {
                out_vec[(size_t)q * k + e] = v ? rank_key_addr(key) : 0xffffffffu;
                out_score[(size_t)q * k + e] = v ? rank_key_score(key) : 0.f;
            }
        }
        if (lane == 0) out_count[q] = cnt;
    }
}
template <int KL>
__global__ __launch_bounds__(256) void merge_topk_kernel(const uint64_t *__restrict__ partial, uint32_t lists_per_query, uint32_t k,
                                                         uint32_t *__restrict__ out_vec, float *__restrict__ out_score,
                                                         uint32_t *__restrict__ out_count) {
    merge_topk_body<KL>(partial, lists_per_query, k, out_vec, out_score, out_count);
}
template <int KL>
__global__ __launch_bounds__(256) void merge_topk_segments_kernel(const ScanMergeTab *table, uint32_t lists_per_query, uint32_t k) {
    const ScanMergeTab t = table[blockIdx.y];
    merge_topk_body<KL>(t.partial, lists_per_query, k, t.out_vec, t.out_score, t.out_count);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <int NJ, bool WIDE>
static hipError_t launch_scan_nj(const ScanArgs &a, uint32_t nblk, hipStream_t s) {
    if (a.k > 256) {  // k up to 512 (rank-fusion / reranker windows reach 500): 8 chained lists, one query per pass
        hipLaunchKernelGGL((scan_topk_kernel<NJ, 1, 8>), dim3(nblk, a.n_queries), dim3(256), 0, s, a);
    } else if (a.k > 64) {  // large result pages: 4 chained lists per query, at most 4 queries per pass
        if (a.qt == 1) hipLaunchKernelGGL((scan_topk_kernel<NJ, 1, 4>), dim3(nblk, a.n_queries), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((scan_topk_kernel<NJ, 4, 4>), dim3(nblk, (a.n_queries + 3) / 4), dim3(256), 0, s, a);
    } else if (a.qt == 1) {
        hipLaunchKernelGGL((scan_topk_kernel<NJ, 1, 1>), dim3(nblk, a.n_queries), dim3(256), 0, s, a);
    } else if (a.qt == 4 || !WIDE) {
        hipLaunchKernelGGL((scan_topk_kernel<NJ, 4, 1>), dim3(nblk, (a.n_queries + 3) / 4), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((scan_topk_kernel<NJ, (WIDE ? 8 : 4), 1>), dim3(nblk, (a.n_queries + 7) / 8), dim3(256), 0, s,
                           a);
    }
    return hipGetLastError();
}

template <int NJ, bool WIDE>
static hipError_t launch_scan_segments_nj(const ScanArgs *table, uint32_t n_seg, const ScanArgs &a, uint32_t nblk, hipStream_t s) {
    const uint32_t nq = a.n_queries;
    if (a.k > 256) {
        hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, 1, 8>), dim3(nblk, nq, n_seg), dim3(256), 0, s, table);
    } else if (a.k > 64) {
        if (a.qt == 1) hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, 1, 4>), dim3(nblk, nq, n_seg), dim3(256), 0, s, table);
        else hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, 4, 4>), dim3(nblk, (nq + 3) / 4, n_seg), dim3(256), 0, s, table);
    } else if (a.qt == 1) {
        hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, 1, 1>), dim3(nblk, nq, n_seg), dim3(256), 0, s, table);
    } else if (a.qt == 4 || !WIDE) {
        hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, 4, 1>), dim3(nblk, (nq + 3) / 4, n_seg), dim3(256), 0, s, table);
    } else {
        hipLaunchKernelGGL((scan_topk_segments_kernel<NJ, (WIDE ? 8 : 4), 1>), dim3(nblk, (nq + 7) / 8, n_seg), dim3(256), 0, s, table);
    }
    return hipGetLastError();
}

// queries per pass over the rows: 8 while the tile fits the register file (D <= 1024, k <= 64), else 4
uint32_t scan_query_tile(uint32_t n_queries, uint32_t dp, uint32_t k) {
    if (n_queries == 1 || k > 256) return 1;
    if (n_queries <= 4 || dp > 1024 || k > 64) return 4;
    return 8;
}

uint32_t scan_num_blocks(uint32_t n) {
    uint32_t b = (n + 15) / 16;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
}

hipError_t launch_scan(ScanArgs a, uint32_t nblk, hipStream_t s) {
    if (a.k == 0 || a.k > NIDX_K_MAX) return hipErrorInvalidValue;
    a.qt = scan_query_tile(a.n_queries, a.dp, a.k);
    int nj = (int)((a.dp + 255u) / 256u);
    if (nj <= 1) return launch_scan_nj<1, true>(a, nblk, s);
    if (nj <= 2) return launch_scan_nj<2, true>(a, nblk, s);
    if (nj <= 3) return launch_scan_nj<3, true>(a, nblk, s);
    if (nj <= 4) return launch_scan_nj<4, true>(a, nblk, s);
    if (nj <= 6) return launch_scan_nj<6, false>(a, nblk, s);
    if (nj <= 8) return launch_scan_nj<8, false>(a, nblk, s);
    if (nj <= 12) return launch_scan_nj<12, false>(a, nblk, s);
    if (nj <= 16) return launch_scan_nj<16, false>(a, nblk, s);
    return hipErrorInvalidValue;
}

// `table` (device): n_seg records that agree with `shape` (host: one of them, qt filled like launch_scan does) in dp, k, n_queries, qt;
// every record's `partial` has room for [n_queries][nblk][k] keys.  The grid's y dimension carries the query tiles (<= 65 535).
hipError_t launch_scan_segments(const ScanArgs *table, uint32_t n_seg, ScanArgs shape, uint32_t nblk, hipStream_t s) {
    if (shape.k == 0 || shape.k > NIDX_K_MAX || n_seg == 0 || n_seg > 65535u) return hipErrorInvalidValue;
    shape.qt = scan_query_tile(shape.n_queries, shape.dp, shape.k);
    int nj = (int)((shape.dp + 255u) / 256u);
    if (nj <= 1) return launch_scan_segments_nj<1, true>(table, n_seg, shape, nblk, s);
    if (nj <= 2) return launch_scan_segments_nj<2, true>(table, n_seg, shape, nblk, s);
    if (nj <= 3) return launch_scan_segments_nj<3, true>(table, n_seg, shape, nblk, s);
    if (nj <= 4) return launch_scan_segments_nj<4, true>(table, n_seg, shape, nblk, s);
    if (nj <= 6) return launch_scan_segments_nj<6, false>(table, n_seg, shape, nblk, s);
    if (nj <= 8) return launch_scan_segments_nj<8, false>(table, n_seg, shape, nblk, s);
    if (nj <= 12) return launch_scan_segments_nj<12, false>(table, n_seg, shape, nblk, s);
    if (nj <= 16) return launch_scan_segments_nj<16, false>(table, n_seg, shape, nblk, s);
    return hipErrorInvalidValue;
}

hipError_t launch_merge_topk_segments(const ScanMergeTab *table, uint32_t n_seg, uint32_t n_queries, uint32_t lists_per_query, uint32_t k, hipStream_t s) {
    if (n_seg == 0 || n_queries == 0) return hipSuccess;
    if (k > 256) hipLaunchKernelGGL(merge_topk_segments_kernel<8>, dim3(n_queries, n_seg), dim3(256), 0, s, table, lists_per_query, k);
    else if (k > 64) hipLaunchKernelGGL(merge_topk_segments_kernel<4>, dim3(n_queries, n_seg), dim3(256), 0, s, table, lists_per_query, k);
    else hipLaunchKernelGGL(merge_topk_segments_kernel<1>, dim3(n_queries, n_seg), dim3(256), 0, s, table, lists_per_query, k);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const uint64_t *partial, uint32_t n_queries, uint32_t lists_per_query, uint32_t k,
                             uint32_t *out_vec, float *out_score, uint32_t *out_count, hipStream_t s) {
    if (k > 256)
        hipLaunchKernelGGL(merge_topk_kernel<8>, dim3(n_queries), dim3(256), 0, s, partial, lists_per_query, k, out_vec,
                           out_score, out_count);
    else if (k > 64)
        hipLaunchKernelGGL(merge_topk_kernel<4>, dim3(n_queries), dim3(256), 0, s, partial, lists_per_query, k, out_vec,
                           out_score, out_count);
    else
        hipLaunchKernelGGL(merge_topk_kernel<1>, dim3(n_queries), dim3(256), 0, s, partial, lists_per_query, k, out_vec,
                           out_score, out_count);
    return hipGetLastError();
}

// ---- multi-vector paragraphs (VectorCardinality::Multi): "only return the best vector match per paragraph"
//      (segment.rs:582-593).  Input: each query's k_in best VECTORS (score desc, addr asc); output: the k best
//      paragraphs, each represented by its best vector — on equal scores the LAST one of the paragraph, like
//      Iterator::max_by — ordered by (score desc, representative addr asc).  One wave per query. ----
__global__ __launch_bounds__(64) void para_best_kernel(const uint32_t *in_vec, const float *in_score, const uint32_t *in_count,
                                                       uint32_t k_in, const uint32_t *para_of_vec, uint32_t k, uint32_t *out_vec,
                                                       float *out_score, uint32_t *out_count) {
    __shared__ uint32_t a_para[NIDX_K_MAX], a_vec[NIDX_K_MAX];
    __shared__ float a_score[NIDX_K_MAX];
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t cnt = in_count[q];
    int na = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t v = in_vec[(size_t)q * k_in + i];
        const float s = in_score[(size_t)q * k_in + i];
        const uint32_t p = para_of_vec[v];
        int found = -1;
        for (int base = 0; base < na; base += 64) {
            const int j = base + lane;
            const unsigned long long m = __ballot(j < na && a_para[j] == p);
            if (m) {
                found = base + __ffsll((long long)m) - 1;
                break;
            }
        }
        if (found >= 0) {
            // a later vector of an accepted paragraph: it replaces the representative only on an exactly equal score
            if (lane == 0 && __builtin_bit_cast(uint32_t, a_score[found]) == __builtin_bit_cast(uint32_t, s)) a_vec[found] = v;
        } else if (na < (int)k) {
            if (lane == 0) {
                a_para[na] = p;
                a_vec[na] = v;
                a_score[na] = s;
            }
            na++;
        }
    }
    // order by (score desc, representative addr asc): rank of every entry among the na accepted
    for (int base = 0; base < (int)k; base += 64) {
        const int e = base + lane;
        if (e < na) {
            const uint64_t key = rank_key(a_score[e], a_vec[e]);
            int rank = 0;
            for (int j = 0; j < na; j++) rank += rank_key(a_score[j], a_vec[j]) > key ? 1 : 0;
            out_vec[(size_t)q * k + rank] = a_vec[e];
            out_score[(size_t)q * k + rank] = a_score[e];
        } else if (e < (int)k) {
            out_vec[(size_t)q * k + e] = 0xffffffffu;
            out_score[(size_t)q * k + e] = 0.f;
        }
    }
    if (lane == 0) out_count[q] = (uint32_t)na;
}

// ---- maxsim_similarity (nidx_vector/src/multivector.rs:33-46): score of one paragraph for one multi-vector query =
//      sum over the query's vectors of the best similarity among the paragraph's vectors (0.0 when none is positive).
//      One wave per (query, paragraph) candidate; every similarity in the WAVE64 order. ----
__global__ __launch_bounds__(64) void maxsim_kernel(const float *vectors, const float *norm2, uint32_t dp, int similarity,
                                                    const float *queries, const uint32_t *cand_qfirst, const uint32_t *cand_qnum,
                                                    const uint32_t *cand_first, const uint32_t *cand_num, float *out) {
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t q0 = cand_qfirst[c], qn = cand_qnum[c], v0 = cand_first[c], vn = cand_num[c];
    const int nj = (int)((dp + 255u) / 256u);
    float summaxsim = 0.0f;
    for (uint32_t qi = 0; qi < qn; qi++) {
        const float *qrow = queries + (size_t)(q0 + qi) * dp;
        float qq = 0.f;
        if (similarity == 1) {
            float acc = 0.f;
            for (int j = 0; j < nj; j++) {
                const float4 qv = load_row_chunk(qrow, dp, j, lane);
                acc = fma4(qv, qv, acc);
            }
            qq = wave_butterfly_sum(acc);
        }
        float maxsim = 0.0f;
        for (uint32_t vi = 0; vi < vn; vi++) {
            const float *row = vectors + (size_t)(v0 + vi) * dp;
            float acc = 0.f;
            for (int j = 0; j < nj; j++) acc = fma4(load_row_chunk(row, dp, j, lane), load_row_chunk(qrow, dp, j, lane), acc);
            const float ab = wave_butterfly_sum(acc);
            const float sim = similarity == 1 ? cosine_from_sums(ab, norm2[v0 + vi], qq) : ab;
            if (sim > maxsim) maxsim = sim;
        }
        summaxsim = summaxsim + maxsim;
    }
    if (lane == 0) out[c] = summaxsim;
}

hipError_t launch_maxsim(const float *vectors, const float *norm2, uint32_t dp, int similarity, const float *queries,
                         const uint32_t *cand_qfirst, const uint32_t *cand_qnum, const uint32_t *cand_first, const uint32_t *cand_num,
                         uint32_t n_cand, float *out, hipStream_t s) {
    if (n_cand == 0) return hipSuccess;
    hipLaunchKernelGGL(maxsim_kernel, dim3(n_cand), dim3(64), 0, s, vectors, norm2, dp, similarity, queries, cand_qfirst, cand_qnum, cand_first,
                       cand_num, out);
    return hipGetLastError();
}

hipError_t launch_para_best(const uint32_t *in_vec, const float *in_score, const uint32_t *in_count, uint32_t n_queries, uint32_t k_in,
                            const uint32_t *para_of_vec, uint32_t k, uint32_t *out_vec, float *out_score, uint32_t *out_count,
                            hipStream_t s) {
    if (n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(para_best_kernel, dim3(n_queries), dim3(64), 0, s, in_vec, in_score, in_count, k_in, para_of_vec, k, out_vec, out_score,
                       out_count);
    return hipGetLastError();
}

hipError_t launch_row_norms(const float *vectors, uint32_t n, uint32_t dp, float *norm2, hipStream_t s) {
    uint32_t blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, s, vectors, n, dp, norm2);
    return hipGetLastError();
}

hipError_t launch_pair_similarity(const float *x, const float *y, uint32_t n, uint32_t dp, int similarity, float *out,
                                  hipStream_t s) {
    uint32_t blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pair_similarity_kernel, dim3(blocks), dim3(256), 0, s, x, y, n, dp, similarity, out);
    return hipGetLastError();
}

}  // namespace nidx
