// rabitq.hip — RaBitQ on gfx950: 1-bit stored codes, 4-bit query codes, popcount estimates and the
// error-bounded re-rank with the raw f32 rows.
//
// Replaces, for quantizable indexes (Dot similarity, dimension % 64 == 0, config.rs:170-173):
//   EncodedVector::encode            nidx_vector/src/vector_types/rabitq.rs:75-106   rabitq_encode_kernel
//   QueryVector::from_vector         rabitq.rs:124-157                               rabitq_query_kernel
//   QueryVector::similarity          rabitq.rs:163-218                               rabitq_estimate()
//   rerank_top                       rabitq.rs:221-244                               Reranker
//   brute_force_search, RaBitQ arm   nidx_vector/src/segment.rs:569-623              rabitq_bf_kernel
//   HnswSearcher::search, RaBitQ arm nidx_vector/src/hnsw/search.rs:306-366          rabitq_hnsw_kernel
//     (closest_up_nodes then runs in hnsw_search_kernel's entry mode on the re-ranked entry points)
//
// All of it is integer popcounts and plain f32 arithmetic in the reference's operation order (the
// file is compiled with -ffp-contract=off; f32 division and sqrt are correctly rounded), so estimates,
// error bounds and therefore every admission / re-rank decision are bit-identical to the CPU path.
// The only f32 sums are dot_quant_original (encode) and the re-rank's raw dot products: both in the
// canonical WAVE64 order of device_common.h.
//
// Work shape: one WAVE per query.  The traversal and the re-rank are sequential decision chains
// (ef = min(100 k, 2000) results, candidates visited best-first; re-rank thresholds move with
// every accepted row), so the 64 lanes are spent on the 60 neighbours of one expansion (one
// 8+D/8-byte code per lane — no cross-lane reduction at all) and on the 64-wide raw dot products.
// Bound: HBM latency/bytes (D/8+8 bytes per estimate, 4 D per re-ranked row, 256 B per expansion).
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

#include "hnsw_device.h"

namespace nidx {

#define RABITQ_EPSILON 1.9f /* rabitq.rs:30 */

// ------------------------------------------------------------------------------------------------
// bit plumbing: lane l of chunk j owns elements 256 j + 4 l .. + 3; ballots b[c] hold component c of
// every lane.  Word s (0..3) of the chunk = elements 64 s .. 64 s + 63 = lanes 16 s .. 16 s + 15.
// ------------------------------------------------------------------------------------------------
__device__ inline uint64_t spread4(uint64_t x) {  // bit i of a 16-bit value -> bit 4 i
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}
__device__ inline uint64_t chunk_word(const unsigned long long (&b)[4], int s) {
    uint64_t w = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) w |= spread4((b[c] >> (16 * s)) & 0xffffull) << c;
    return w;
}

// ---- EncodedVector::encode: one wave per row ------------------------------------------------------
__global__ __launch_bounds__(256) void rabitq_encode_kernel(const float *vectors, uint32_t n, uint32_t dp, uint32_t dim,
                                                            uint8_t *out, uint32_t rec_len) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *x = vectors + (size_t)row * dp;
    uint8_t *rec = out + (size_t)row * rec_len;
    const float root_dim = sqrtf((float)dim);
    const float pos = 1.0f / root_dim, neg = -1.0f / root_dim;
    const int nj = (int)((dim + 255u) / 256u);
    float acc = 0.f;
    uint32_t sum_bits = 0;
    for (int j = 0; j < nj; j++) {
        const uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
        const bool in = e < dim;
        float4 v = in ? *reinterpret_cast<const float4 *>(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
            acc = fmaf(v.x, v.x > 0.0f ? pos : neg, acc);
            acc = fmaf(v.y, v.y > 0.0f ? pos : neg, acc);
            acc = fmaf(v.z, v.z > 0.0f ? pos : neg, acc);
            acc = fmaf(v.w, v.w > 0.0f ? pos : neg, acc);
        }
        unsigned long long b[4];
        b[0] = __ballot(in && v.x > 0.0f);
        b[1] = __ballot(in && v.y > 0.0f);
        b[2] = __ballot(in && v.z > 0.0f);
        b[3] = __ballot(in && v.w > 0.0f);
        sum_bits += (uint32_t)(__popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]));
        if (lane < 4) {
            uint32_t w = (uint32_t)j * 4u + (uint32_t)lane;
            if (w < dim / 64u) *reinterpret_cast<uint64_t *>(rec + 8 + (size_t)w * 8) = chunk_word(b, lane);
        }
    }
    float dot = wave_butterfly_sum(acc);
    if (lane == 0) {
        *reinterpret_cast<float *>(rec) = dot;
        *reinterpret_cast<uint32_t *>(rec + 4) = sum_bits;
    }
}

// ---- QueryVector::from_vector: one wave per query ---------------------------------------------------
__device__ inline unsigned long long f32_as_u64(float x) {  // Rust `as u64`: saturating, NaN -> 0
    if (!(x > 0.0f)) return 0ull;
    if (x >= 18446744073709551616.0f) return ~0ull;
    return (unsigned long long)x;
}

__global__ __launch_bounds__(256) void rabitq_query_kernel(const float *queries, uint32_t nq, uint32_t dp, uint32_t dim,
                                                           RabitqQueryDev *qd, uint64_t *planes) {
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const float *x = queries + (size_t)qi * dp;
    const uint32_t nw = dim / 64u;
    uint64_t *pl = planes + (size_t)qi * 4u * nw;
    const int nj = (int)((dim + 255u) / 256u);
    // fold (min, max) — order independent
    float lo = x[0], hi = x[0];
    for (int j = 0; j < nj; j++) {
        const uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
        if (e < dim) {
            float4 v = *reinterpret_cast<const float4 *>(x + e);
            float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (c[i] < lo) lo = c[i];
                if (c[i] > hi) hi = c[i];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        if (ol < lo) lo = ol;
        if (oh > hi) hi = oh;
    }
    hi += 0.00001f;
    const float delta = (hi - lo) / 16.0f;
    unsigned long long sumq = 0;
    for (int j = 0; j < nj; j++) {
        const uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
        const bool in = e < dim;
        float4 v = in ? *reinterpret_cast<const float4 *>(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long wq[4] = {0, 0, 0, 0};
        if (in) {
            wq[0] = f32_as_u64((v.x - lo) / delta);
            wq[1] = f32_as_u64((v.y - lo) / delta);
            wq[2] = f32_as_u64((v.z - lo) / delta);
            wq[3] = f32_as_u64((v.w - lo) / delta);
            sumq += wq[0] + wq[1] + wq[2] + wq[3];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            unsigned long long b[4];
#pragma unroll
            for (int c = 0; c < 4; c++) b[c] = __ballot(in && ((wq[c] >> p) & 1ull));
            if (lane < 4) {
                uint32_t w = (uint32_t)j * 4u + (uint32_t)lane;
                if (w < nw) pl[(size_t)p * nw + w] = chunk_word(b, lane);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t l = __shfl_xor((uint32_t)sumq, off, 64), h = __shfl_xor((uint32_t)(sumq >> 32), off, 64);
        sumq += ((unsigned long long)h << 32) | l;
    }
    if (lane == 0) {
        const float root_dim = sqrtf((float)dim);
        const float sum_quantized = (float)(uint32_t)sumq;  // `sum_quantized as u32`, then `as f32` (rabitq.rs:152,207)
        RabitqQueryDev o;
        o.c_dot = 2.0f * delta / root_dim;               // 2.0 * delta / root_dim   (* dot)
        o.two_low = 2.0f * lo;                           // 2.0 * low                (* sum_bits / root_dim)
        o.c_sumq = delta * sum_quantized / root_dim;     // delta * sum_quantized / root_dim
        o.c_low = lo * root_dim;                         // low * root_dim
        o.root_dim = root_dim;
        o.low = lo;
        o.delta = delta;
        o.sum_quantized = (uint32_t)sumq;
        qd[qi] = o;
    }
}

// ---- QueryVector::similarity for one stored code (per lane) -------------------------------------------
// A code is fetched with all of its 8-byte loads in flight at once (NW of them when the word count is a
// template constant, groups of four otherwise) and scored against this query's four bit planes in LDS
// ([4][nw]; every lane reads the same words: broadcast).
template <int NW>
struct RqCode {
    uint64_t s[NW > 0 ? NW : 1];
    float dqo;
    uint32_t sum_bits;
};
template <int NW>
__device__ inline void rq_load_code(const uint8_t *rec, RqCode<NW> &c) {
    const uint2 hdr = *reinterpret_cast<const uint2 *>(rec);
    if (NW > 0) {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(rec + 8);
#pragma unroll
        for (int w = 0; w < (NW > 0 ? NW : 1); w++) c.s[w] = src[w];
    }
    c.dqo = __builtin_bit_cast(float, hdr.x);
    c.sum_bits = hdr.y;
}
__device__ inline void rq_finish(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, float dqo, uint32_t sum_bits,
                                 const RabitqQueryDev &c, float &est, float &err) {
    const float dot = (float)(d0 + d1 * 2u + d2 * 4u + d3 * 8u);
    const float dqq = c.c_dot * dot + c.two_low * (float)sum_bits / c.root_dim - c.c_sumq - c.c_low;
    est = dqq / dqo;
    const float dd = dqo * dqo;
    err = sqrtf((1.0f - dd) / dd) * RABITQ_EPSILON / c.root_dim;
}
template <int NW>
__device__ inline void rq_score_code(const RqCode<NW> &code, const uint8_t *rec, const uint64_t *qp, uint32_t nw,
                                     const RabitqQueryDev &c, float &est, float &err) {
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    if (NW > 0) {
#pragma unroll
        for (int w = 0; w < (NW > 0 ? NW : 1); w++) {
            const uint64_t sw = code.s[w];
            d0 += (uint32_t)__popcll(qp[w] & sw);
            d1 += (uint32_t)__popcll(qp[NW + w] & sw);
            d2 += (uint32_t)__popcll(qp[2 * NW + w] & sw);
            d3 += (uint32_t)__popcll(qp[3 * NW + w] & sw);
        }
    } else {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(rec + 8);
        for (uint32_t w0 = 0; w0 < nw; w0 += 4) {
            uint64_t sw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) sw[u] = w0 + u < nw ? src[w0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (w0 + u < nw) {
                    d0 += (uint32_t)__popcll(qp[w0 + u] & sw[u]);
                    d1 += (uint32_t)__popcll(qp[nw + w0 + u] & sw[u]);
                    d2 += (uint32_t)__popcll(qp[2 * nw + w0 + u] & sw[u]);
                    d3 += (uint32_t)__popcll(qp[3 * nw + w0 + u] & sw[u]);
                }
            }
        }
    }
    rq_finish(d0, d1, d2, d3, code.dqo, code.sum_bits, c, est, err);
}
template <int NW>
__device__ inline void rabitq_estimate(const uint8_t *rec, const uint64_t *qp, uint32_t nw, const RabitqQueryDev &c,
                                       float &est, float &err) {
    RqCode<NW> code;
    rq_load_code<NW>(rec, code);
    rq_score_code<NW>(code, rec, qp, nw, c, est, err);
}
__device__ inline float rabitq_error(const uint8_t *rec, const RabitqQueryDev &c) {
    const float dqo = *reinterpret_cast<const float *>(rec);
    const float dd = dqo * dqo;
    return sqrtf((1.0f - dd) / dd) * RABITQ_EPSILON / c.root_dim;
}

// Wave-uniform reads of one lane's value.  Unlike __shfl they return SGPRs, so the sequential replay
// loops below (whose every decision is wave-uniform) compile to scalar branches instead of exec-masked
// vector code.
__device__ inline uint32_t lane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ inline float lane_f32(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ inline uint64_t lane_u64(uint64_t v, int l) {
    return ((uint64_t)lane_u32((uint32_t)(v >> 32), l) << 32) | lane_u32((uint32_t)v, l);
}
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)uni((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)v);
}
__device__ inline float unif(float v) { return __builtin_bit_cast(float, uni(__builtin_bit_cast(int, v))); }

// ---- sorted key array in LDS (best first), one wave -----------------------------------------------------
// Keys: score bits << 32 | (~addr & 0x7fffffff) << 1 | unexpanded flag — ordered like rank_key()
// (higher score first, then lower address; the flag never decides: addresses are unique in a list).
__device__ inline uint64_t rq_key(float score, uint32_t addr, uint32_t flag) {
    uint32_t k = (uint32_t)total_key(score) ^ 0x80000000u;
    return ((uint64_t)k << 32) | ((uint64_t)((~addr) & 0x7fffffffu) << 1) | flag;
}
__device__ inline uint32_t rq_addr(uint64_t key) { return (~(uint32_t)(key >> 1)) & 0x7fffffffu; }

// Inserts nk — which must rank before the current worst key when the list is full — keeping at most `cap`
// keys.  Returns the evicted key (0 if none); `worst` is the list's last key and is kept up to date (both
// come out of registers: no dependent LDS read).  `keys` must be readable up to the next multiple of 64
// past cap + 1.  The list is walked from its end in blocks of eight 64-key chunks whose LDS reads are all
// in flight together.  (A variant that located the landing chunk through per-chunk pivots held in
// registers and moved the chunks behind it wholesale was measured 12 % SLOWER: the walk is bound by
// taken scalar branches and LDS round trips of a lone wave, not by compares.)
__device__ inline uint64_t sorted_insert(uint64_t *keys, int &len_io, int cap, uint64_t nk, int lane, int &pos_out,
                                        uint64_t &worst) {
    const int len = uni(len_io);
    int after = 0;  // entries ranking after nk (they move down by one)
    bool done = len == 0;
    uint64_t old_prev = 0;  // keys[len - 2] before the insert
    const int top = (len - 1) >> 6;
    for (int cb = top; cb >= 0 && !done; cb -= 8) {
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = cb - u < 0 ? 0 : cb - u;
            v[u] = keys[c * 64 + lane];
        }
        if (cb == top && len >= 2) {
            const uint64_t a = lane_u64(v[0], (len - 2) & 63), b = lane_u64(v[1], (len - 2) & 63);
            old_prev = ((len - 2) >> 6) == top ? a : b;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = cb - u;
            if (!done && c >= 0) {
                const int i = c * 64 + lane;
                const bool mv = i < len && v[u] < nk;
                if (mv) keys[i + 1] = v[u];
                const int cnt = __popcll(__ballot(mv));
                after += cnt;
                const int n_in = len - c * 64 < 64 ? len - c * 64 : 64;
                if (cnt < n_in) done = true;  // an entry of this chunk ranks before nk: so does everything above
            }
        }
    }
    const int pos = len - after;
    if (lane == 0) keys[pos] = nk;
    pos_out = pos;
    uint64_t evicted = 0;
    if (len == cap) {
        evicted = worst;                       // the old last key moved to keys[cap]
        worst = after >= 2 ? old_prev : nk;    // new last = the old keys[cap - 2], or nk when it landed at the end
    } else {
        if (after == 0) worst = nk;
        len_io = len + 1;
    }
    return evicted;
}

// ---- rerank_top ---------------------------------------------------------------------------------------
// `best` (LDS, k + 1 keys) holds the k best real scores so far.  feed() takes up to 64 candidates in
// the reference's iteration order (lane order) and replays `if best.len() < top_k || best_k < upper_bound`
// / `if real_score >= min_score && (best.len() < top_k || best_k < real_score)` one by one; raw dot products
// are computed four rows at a time (speculatively — a row evaluated but skipped by the replay costs bytes only).
struct Reranker {
    uint64_t *best;
    uint64_t worst;
    int len, k;
    float best_k, min_score;
    uint32_t n_eval;
    const float *vectors;
    uint32_t dp;
    const float *q;  // LDS copy of the raw query, [dp]

    __device__ inline void init(uint64_t *best_, int k_, float min_score_, const float *vectors_, uint32_t dp_, const float *q_) {
        best = best_;
        worst = 0;
        len = 0;
        k = k_;
        best_k = 0.0f;
        min_score = min_score_;
        n_eval = 0;
        vectors = vectors_;
        dp = dp_;
        q = q_;
    }
    __device__ inline void feed(bool cand, uint32_t addr, float ub, int lane) {
        len = uni(len);
        best_k = unif(best_k);
        unsigned long long todo = __ballot(cand && (len < k || best_k < ub));
        const int nj = (int)((dp + 255u) / 256u);
        while (todo) {
            todo = uni64(todo);
            len = uni(len);
            worst = uni64(worst);
            best_k = unif(best_k);
            int idx[4];
            int cnt = 0;
            unsigned long long t = todo;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                idx[i] = t ? __ffsll((long long)t) - 1 : -1;
                if (t) {
                    t &= t - 1;
                    cnt++;
                }
            }
            uint32_t ra[4];
#pragma unroll
            for (int i = 0; i < 4; i++) ra[i] = lane_u32(addr, idx[i] < 0 ? 0 : idx[i]);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < nj; j++) {
                const uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
                if (e < dp) {
                    const float4 qv = *reinterpret_cast<const float4 *>(q + e);
                    float4 x[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        x[i] = i < cnt ? *reinterpret_cast<const float4 *>(vectors + (size_t)ra[i] * dp + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] = fma4(x[i], qv, acc[i]);
                }
            }
            const float red = QReduce<4>::run(acc, lane);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i >= cnt) break;
                const float real = lane_f32(red, (i >> 1) * 32 + (i & 1) * 16);
                const float ubi = lane_f32(ub, idx[i]);
                if (len < k || best_k < ubi) {
                    n_eval++;
                    if (real >= min_score && (len < k || best_k < real)) {
                        int pos;
                        sorted_insert(best, len, k, rq_key(real, ra[i], 0), lane, pos, worst);
                        best_k = rank_key_score(worst);
                    }
                }
            }
            todo = t;
            // thresholds only rise: drop the lanes that can no longer pass
            if (len >= k) todo &= __ballot(best_k < ub);
        }
    }
    __device__ inline void write(uint32_t *out_vec, float *out_score, uint32_t *out_count, int lane) const {
        for (int i = lane; i < k; i += 64) {
            out_vec[i] = i < len ? rq_addr(best[i]) : 0xffffffffu;
            out_score[i] = i < len ? rank_key_score(best[i]) : 0.f;
        }
        if (lane == 0) *out_count = (uint32_t)len;
    }
};

// dynamic LDS carve-up shared by the two search kernels
struct RqShared {
    uint64_t *planes;  // [4][nw]
    float *q;          // [dp]
    uint64_t *best;    // [k + 1]
    uint64_t *res;     // [ef + 1]            (hnsw only)
    uint64_t *ties;    // [RABITQ_TIE_CAP]    (hnsw only)
    uint32_t *vis;     // [1 << RABITQ_UPPER_VIS_LOG2] (hnsw only)
};
#define RABITQ_TIE_CAP 64
#define RABITQ_UPPER_VIS_LOG2 11

// a sorted list of `cap` keys: cap + 1 slots, readable in whole 64-key chunks
__host__ __device__ inline size_t rq_list_bytes(uint32_t cap) { return (size_t)(((cap + 1 + 63) / 64) * 64) * 8; }
// physical 64-key chunks of a two-level list of `cap` keys: every chunk but the last holds >= 32 keys
__host__ __device__ inline uint32_t rq_chunks(uint32_t cap) { return cap / 32u + 2u < 64u ? cap / 32u + 2u : 64u; }

// (q_in_lds = false: the raw query stays in HBM / L2 — only the re-rank reads it, beside the rows it fetches anyway — and 4 dp bytes of LDS
// are another resident walk per CU at ef = 1 000)
__device__ inline RqShared rq_carve(unsigned char *smem, uint32_t nw, uint32_t dp, uint32_t k, uint32_t ef, bool q_in_lds = true) {
    RqShared s;
    size_t off = 0;
    s.planes = reinterpret_cast<uint64_t *>(smem + off);
    off += (size_t)4 * nw * 8;
    s.best = reinterpret_cast<uint64_t *>(smem + off);
    off += rq_list_bytes(k);
    s.res = reinterpret_cast<uint64_t *>(smem + off);
    off += (size_t)rq_chunks(ef) * 512;
    s.ties = reinterpret_cast<uint64_t *>(smem + off);
    off += (size_t)RABITQ_TIE_CAP * 8;
    off = (off + 15) & ~(size_t)15;
    s.q = reinterpret_cast<float *>(smem + off);
    if (q_in_lds) off += (size_t)dp * 4;
    s.vis = reinterpret_cast<uint32_t *>(smem + off);
    return s;
}
static size_t rq_smem_bytes(uint32_t nw, uint32_t dp, uint32_t k, uint32_t ef, bool hnsw, bool q_in_lds = true) {
    size_t off = (size_t)4 * nw * 8 + rq_list_bytes(k) + (size_t)rq_chunks(ef) * 512 + (size_t)RABITQ_TIE_CAP * 8;
    off = (off + 15) & ~(size_t)15;
    if (q_in_lds) off += (size_t)dp * 4;
    if (hnsw) off += (size_t)4 << RABITQ_UPPER_VIS_LOG2;
    return off;
}

// LDS of rabitq_hnsw3_kernel (see its body)
static size_t rq_smem3_bytes(uint32_t nw, uint32_t k, uint32_t ef, uint32_t seen_log2) {
    size_t off = (size_t)4 * nw * 8 + rq_list_bytes(k) + (size_t)rq_chunks(ef) * 512 + (size_t)RABITQ_TIE_CAP * 8;
    off = (off + 15) & ~(size_t)15;
    if ((rq_chunks(ef) - 2u) * 512u < (4u << RABITQ_UPPER_VIS_LOG2)) off += (size_t)4 << RABITQ_UPPER_VIS_LOG2;
    if (seen_log2) off += (size_t)4 << seen_log2;
    return off;
}

__device__ inline void rq_load_query(const RqShared &sh, const RabitqSearchArgs &a, uint32_t qi, uint32_t nw, int lane, bool q_in_lds = true) {
    const uint64_t *gp = a.planes + (size_t)qi * 4u * nw;
    for (uint32_t i = lane; i < 4u * nw; i += 64) sh.planes[i] = gp[i];
    if (!q_in_lds) return;
    const float *gq = a.queries + (size_t)qi * a.seg.dp;
    for (uint32_t i = lane; i < a.seg.dp; i += 64) sh.q[i] = gq[i];
}

// ---- brute force, RaBitQ arm (segment.rs:569-623): one wave per query -------------------------------
// Rows are visited in address order (= the bitset iteration order); every passing row's estimate gives
// an upper bound, `upper_bound >= min_score` admits it to the candidate list, and rerank_top consumes
// that list in the same order.
template <int NW>
__global__ __launch_bounds__(64) void rabitq_bf_kernel(RabitqSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const uint32_t nw = a.seg.dim / 64u;
    RqShared sh = rq_carve(smem, nw, a.seg.dp, a.k, 0);
    rq_load_query(sh, a, qi, nw, lane);
    const RabitqQueryDev qc = a.qd[qi];
    Reranker rr;
    rr.init(sh.best, (int)a.k, a.min_score, a.seg.vectors, a.seg.dp, sh.q);
    uint32_t n_est = 0;
    // one lane per PARAGRAPH: its best vector by estimate (Iterator::max_by keeps the last of equal maxima,
    // segment.rs:586-593); single-vector stores have paragraph p = vector p
    const uint32_t n_para = a.para_first ? a.n_paragraphs : a.seg.n;
    for (uint32_t base = 0; base < n_para; base += 64) {
        const uint32_t p = base + (uint32_t)lane;
        bool ok = p < n_para;
        if (ok) {
            if (a.seg.alive && !bit_test(a.seg.alive, p)) ok = false;
            if (ok && a.filter && !bit_test(a.filter, p)) ok = false;
        }
        const uint32_t first = ok ? (a.para_first ? a.para_first[p] : p) : 0u;
        const uint32_t num = ok ? (a.para_first ? a.para_num[p] : 1u) : 0u;
        if (num == 0) ok = false;
        if (!__any(ok)) continue;
        float est = 0.f, err = 0.f;
        uint32_t r = first;
        if (ok) {
            rabitq_estimate<NW>(a.quant + (size_t)first * a.rec_len, sh.planes, nw, qc, est, err);
            for (uint32_t v = first + 1; v < first + num; v++) {
                float e2, r2;
                rabitq_estimate<NW>(a.quant + (size_t)v * a.rec_len, sh.planes, nw, qc, e2, r2);
                if (total_key(e2) >= total_key(est)) {
                    est = e2;
                    err = r2;
                    r = v;
                }
            }
        }
        n_est += (uint32_t)__popcll(__ballot(ok)) ;
        const float ub = est + err;
        rr.feed(ok && ub >= a.min_score, r, ub, lane);
    }
    rr.write(a.out_vec + (size_t)qi * a.k, a.out_score + (size_t)qi * a.k, a.out_count + qi, lane);
    if (a.stats && lane == 0) {
        uint32_t *o = a.stats + (size_t)qi * NIDX_STAT_STRIDE;
        o[NIDX_STAT_EVALS] = n_est;
        o[NIDX_STAT_EXPANSIONS] = 0;
        o[NIDX_STAT_VISITED] = rr.n_eval;  // raw rows re-ranked
        o[NIDX_STAT_FLAGS] = 0;
    }
}

// ---- HNSW, RaBitQ arm (hnsw/search.rs:242-366): one wave per query ------------------------------------
// layer_search on estimates.  `res` is the result set, sorted, each key carrying an "unexpanded" flag:
// the candidate heap of the reference is exactly the unexpanded part of the result set, plus entries
// evicted from it whose score still EQUALS the worst result's (`cs < ws` does not stop on those) — kept
// in `ties`.  An entry evicted with a lower score than ws can only ever terminate the search when it is
// popped, and by then nothing better is left, so it is dropped on the spot.
// The result set is a two-level sorted list: 64-key chunks in LDS (each sorted, best first) and a directory held
// one entry per lane in registers — lane d: the first key of the d-th chunk, its physical slot and fill.  An
// admission touches ONE chunk: a ballot over the directory finds it, the chunk is read, split at the key's
// position and written back; a full chunk is first halved into a free slot.  Evictions only ever shorten the last
// chunk, so every other chunk holds >= 32 keys and the directory never needs more than cap / 32 + 2 <= 64 lanes.
struct RqLayer {
    uint64_t *res, *ties;
    // Evicted ties beyond the 64 of `ties`: a per-query region in HBM (RabitqSearchArgs::tie_spill; nullptr = none).  It never needs
    // more than ef entries: a tie is an entry evicted while its score still EQUALS the worst result's, the result set holds at
    // most ef entries of one score when it is full, nothing of that score is admitted from then on (`similarity.score > ws.score`),
    // and every entry of a lower score is stale for good (ws only rises) — so stale entries are dropped when the list is full and
    // what is left, all of ONE score, fits ef slots.  The region only ever holds entries of one score (it is dropped as a whole when
    // that score falls below ws).
    uint64_t *spill;
    int n_spill, spill_cap;
    uint64_t dir_first;      // lane d: first (best) key of chunk d
    uint32_t dir_meta;       // lane d: physical slot | fill << 8
    uint64_t free_mask;      // physical slots not in use
    uint64_t worst;          // the last key of the last chunk
    int n_dir, len, n_ties;
    int dcur;                // every chunk before it holds expanded keys only

    __device__ inline void init(int n_phys) {
        dir_first = 0;
        dir_meta = 0;
        free_mask = n_phys >= 64 ? ~0ull : ((1ull << n_phys) - 1ull);
        worst = 0;
        n_dir = 0;
        len = 0;
        n_ties = 0;
        n_spill = 0;
        dcur = 0;
    }
    __device__ inline uint64_t *chunk(uint32_t phys) const { return res + (size_t)phys * 64; }
};

// chunk d is full: its upper half moves to a free physical slot which becomes chunk d + 1
__device__ inline void rq_split(RqLayer &L, int d, int lane) {
    const uint32_t meta = lane_u32(L.dir_meta, d);
    const uint32_t p = meta & 0xffu;
    const int q = __ffsll((long long)L.free_mask) - 1;
    L.free_mask &= ~(1ull << q);
    const uint64_t row = L.chunk(p)[lane];
    if (lane >= 32) L.chunk((uint32_t)q)[lane - 32] = row;
    const uint64_t up_first = shfl_up_u64(L.dir_first, 1);
    const uint32_t up_meta = (uint32_t)__shfl_up((int)L.dir_meta, 1, 64);
    if (lane > d + 1) {
        L.dir_first = up_first;
        L.dir_meta = up_meta;
    }
    const uint64_t mid = lane_u64(row, 32);
    if (lane == d + 1) {
        L.dir_first = mid;
        L.dir_meta = (uint32_t)q | (32u << 8);
    }
    if (lane == d) L.dir_meta = p | (32u << 8);
    L.n_dir++;
    if (L.dcur > d) L.dcur++;
}

// The wave re-reads spill slots its lanes wrote a few instructions earlier: workgroup-scope accesses keep that coherent through the CU's L1
__device__ inline uint64_t rq_ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void rq_st64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The 64-entry list of evicted ties is full.  Entries whose score fell below the worst result's (`ws`) can never be expanded any more
// (hnsw/search.rs:271-277: popping one ends the search, and nothing above it is left by then): they go.  If all 64 still tie with
// ws they move to the query's region in HBM, which holds entries of that one score only.
__device__ inline void rq_ties_make_room(RqLayer &L, float ws, int lane) {
    const uint64_t t = L.ties[lane];
    const bool keep = !(rank_key_score(t) < ws);
    const unsigned long long m = __ballot(keep);
    const int n_keep = __popcll(m);
    if (n_keep < RABITQ_TIE_CAP) {
        const int at = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (keep) L.ties[at] = t;
        L.n_ties = n_keep;
        return;
    }
    if (!L.spill) return;
    if (L.n_spill > 0 && rank_key_score(rq_ld64(L.spill)) < ws) L.n_spill = 0;   // the region's score fell behind: all of it is stale
    if (L.n_spill + RABITQ_TIE_CAP > L.spill_cap) return;
    rq_st64(L.spill + L.n_spill + lane, t);
    L.n_spill += RABITQ_TIE_CAP;
    L.n_ties = 0;
}

__device__ inline void rq_admit(RqLayer &L, int kk, float est, uint32_t addr, int lane, uint32_t &flags) {
    L.worst = uni64(L.worst);
    L.free_mask = uni64(L.free_mask);
    L.n_dir = uni(L.n_dir);
    L.len = uni(L.len);
    L.n_ties = uni(L.n_ties);
    L.n_spill = uni(L.n_spill);
    L.dcur = uni(L.dcur);
    const uint64_t nk = rq_key(est, addr, 1u);
    if (L.n_dir == 0) {  // first key of the layer
        const int q = __ffsll((long long)L.free_mask) - 1;
        L.free_mask &= ~(1ull << q);
        if (lane == 0) {
            L.chunk((uint32_t)q)[0] = nk;
            L.dir_first = nk;
            L.dir_meta = (uint32_t)q | (1u << 8);
        }
        L.n_dir = 1;
        L.len = 1;
        L.worst = nk;
        L.dcur = 0;
        return;
    }
    int d = __popcll(__ballot(lane < L.n_dir && L.dir_first > nk));
    d = d > 0 ? d - 1 : 0;
    if ((lane_u32(L.dir_meta, d) >> 8) == 64u) {
        rq_split(L, d, lane);
        if (lane_u64(L.dir_first, d + 1) > nk) d++;
    }
    const uint32_t meta = lane_u32(L.dir_meta, d);
    const uint32_t phys = meta & 0xffu;
    const int cnt = (int)(meta >> 8);
    const int dl = L.n_dir - 1;
    const uint32_t meta_l = lane_u32(L.dir_meta, dl);
    // the landing chunk and the last chunk are read together (one LDS round trip)
    const uint64_t row = L.chunk(phys)[lane];
    const uint64_t tail = L.chunk(meta_l & 0xffu)[lane];
    const bool valid = lane < cnt;
    const int pos = __popcll(__ballot(valid && row > nk));
    if (valid && lane >= pos) L.chunk(phys)[lane + 1] = row;
    if (lane == 0) L.chunk(phys)[pos] = nk;
    if (lane == d) {
        L.dir_meta = phys | ((uint32_t)(cnt + 1) << 8);
        if (pos == 0) L.dir_first = nk;
    }
    if (d < L.dcur) L.dcur = d;
    const bool at_end = d == dl && pos == cnt;  // nk is the new last key of the list
    if (L.len < kk) {
        L.len++;
        if (at_end) L.worst = nk;
        return;
    }
    // full: the last key of the last chunk leaves (nk ranks before it, so it is never nk)
    const uint64_t ev = L.worst;
    const int cnt_l = (int)(meta_l >> 8) + (d == dl ? 1 : 0);  // fill of the last chunk after the insert
    if (cnt_l >= 2) {
        uint64_t nw;
        if (d == dl) nw = pos == cnt - 1 ? nk : lane_u64(row, cnt - 2);  // arrangement after the insert, minus its last key
        else nw = lane_u64(tail, cnt_l - 2);
        L.worst = nw;
        if (lane == dl) L.dir_meta = (meta_l & 0xffu) | ((uint32_t)(cnt_l - 1) << 8);
    } else {
        // the last chunk held only the evicted key: free it, the previous chunk's last key is the new worst
        L.free_mask |= 1ull << (meta_l & 0xffu);
        L.n_dir--;
        const uint32_t meta_p = lane_u32(L.dir_meta, dl - 1);
        const int cnt_p = (int)(meta_p >> 8) + (d == dl - 1 ? 0 : 0);
        L.worst = L.chunk(meta_p & 0xffu)[cnt_p - 1];
        if (L.dcur > L.n_dir) L.dcur = L.n_dir;
    }
    if (ev & 1ull) {
        // the evicted entry is still a candidate only while its score is not below the worst result's
        const float ws = rank_key_score(L.worst);
        if (!(rank_key_score(ev) < ws)) {
            if (L.n_ties == RABITQ_TIE_CAP) rq_ties_make_room(L, ws, lane);
            if (L.n_ties < RABITQ_TIE_CAP) {
                if (lane == 0) L.ties[L.n_ties] = ev;
                L.n_ties++;
            } else {
                flags |= NIDX_FLAG_POOL_INEXACT;   // (no spill region, or one smaller than ef: not reachable through the library)
            }
        }
    }
}

// lane i's word of the edge record of `node` (no cross-lane use: the load stays in flight)
__device__ inline uint32_t load_edge_raw(const GraphDev &g, uint32_t node, int layer, int lane) {
    if (layer == 0) return g.l0[(size_t)node * NIDX_L0_STRIDE + lane];
    const uint32_t base = g.upper_base[node];
    if (base != 0xffffffffu && lane < NIDX_UP_STRIDE) return g.upper[((size_t)base + (layer - 1)) * NIDX_UP_STRIDE + lane];
    return 0u;
}

// pops the best candidate; returns false when the search is over.  `next` = the runner-up when it sits in
// the same chunk (0xffffffff otherwise): its edge record is prefetched by the caller.
__device__ inline bool rq_pop(RqLayer &L, int lane, uint32_t &node, uint32_t &next) {
    next = 0xffffffffu;
    L.dcur = uni(L.dcur);
    L.n_dir = uni(L.n_dir);
    while (L.dcur < L.n_dir) {
        const uint32_t meta = lane_u32(L.dir_meta, L.dcur);
        const uint64_t mine = lane < (int)(meta >> 8) ? L.chunk(meta & 0xffu)[lane] : 0ull;
        unsigned long long m = __ballot((mine & 1ull) != 0);
        if (m) {
            const int j = __ffsll((long long)m) - 1;
            const uint64_t key = lane_u64(mine, j);
            if (lane == j) L.chunk(meta & 0xffu)[lane] = mine & ~1ull;
            m &= m - 1;
            if (m) next = rq_addr(lane_u64(mine, __ffsll((long long)m) - 1));
            node = rq_addr(key);
            return true;  // a member of the result set never scores below its worst entry
        }
        L.dcur++;
    }
    L.n_spill = uni(L.n_spill);
    if (L.n_ties == 0 && L.n_spill == 0) return false;
    // best of the evicted ties
    uint64_t b = lane < L.n_ties ? L.ties[lane] : 0ull;
    const uint64_t best = wave_max_u64(b);
    if (L.n_spill > 0) {
        // ... and of those that were moved to HBM (keys are unique: the address is part of them)
        uint64_t sb = 0ull;
        int si = 0;
        for (int i = lane; i < L.n_spill; i += 64) {
            const uint64_t v = rq_ld64(L.spill + i);
            if (v > sb) sb = v, si = i;
        }
        const uint64_t sbest = wave_max_u64(sb);
        if (sbest > best) {
            const int src = __ffsll((long long)__ballot(sb == sbest)) - 1;
            const int idx = __shfl(si, src, 64);
            const uint64_t last = rq_ld64(L.spill + (L.n_spill - 1));
            if (lane == 0) rq_st64(L.spill + idx, last);
            L.n_spill--;
            if (rank_key_score(sbest) < rank_key_score(L.worst)) return false;  // `cs < ws => break` (search.rs:271-277)
            node = rq_addr(sbest);
            return true;
        }
    }
    if (L.n_ties == 0) return false;   // (the region only held stale entries below the list's)
    const unsigned long long m = __ballot(b == best);
    const int idx = __ffsll((long long)m) - 1;
    const uint64_t last = L.ties[L.n_ties - 1];
    if (lane == 0) L.ties[idx] = last;
    L.n_ties--;
    const float ws = rank_key_score(L.worst);
    if (rank_key_score(best) < ws) return false;  // `cs < ws => break` (search.rs:271-277)
    node = rq_addr(best);
    return true;
}

template <int NW>
__device__ inline void rabitq_hnsw1_body(const RabitqSearchArgs &a, const uint32_t qi, unsigned char *smem) {
    const int lane = threadIdx.x;
    const uint32_t nw = a.seg.dim / 64u;
    RqShared sh = rq_carve(smem, nw, a.seg.dp, a.k, a.ef);
    rq_load_query(sh, a, qi, nw, lane);
    const RabitqQueryDev qc = a.qd[qi];
    uint32_t *gvis = a.visited + (size_t)qi * a.vis_words;  // layer-0 visited bitset (zeroed by the host)
    uint32_t n_est = 0, n_exp = 0, flags = 0;
    uint64_t cyc_pop = 0, cyc_vis = 0, cyc_est = 0, cyc_ins = 0;
    // (s_memtime is a scalar MEMORY instruction, ~100+ cycles each with its s_waitcnt: the cycle split is taken only when the caller asked
    // for counters — the timed launches of bench.py do not)
    const bool timing = a.stats != nullptr;
    auto now = [&]() -> uint64_t { return timing ? (uint64_t)clock64() : 0ull; };
    const uint64_t t_start = now();

    uint32_t ep = a.g.ep_node;
    RqLayer L;
    L.res = sh.res;
    L.ties = sh.ties;
    L.spill = a.tie_spill ? a.tie_spill + (size_t)qi * a.tie_stride : nullptr;
    L.spill_cap = (int)a.tie_stride;
    for (int layer = (int)a.g.ep_layer; layer >= 0; layer--) {
        const int kk = layer == 0 ? (int)a.ef : 1;
        L.init((int)rq_chunks((uint32_t)kk));
        uint32_t vis_count = 0;
        if (layer > 0) {
            for (uint32_t i = lane; i < (1u << RABITQ_UPPER_VIS_LOG2); i += 64) sh.vis[i] = NIDX_VIS_EMPTY;
            if (lane == 0) vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, ep);
            vis_count = 1;
        } else if (lane == 0) {
            atomicOr(&gvis[ep >> 5], 1u << (ep & 31));
        }
        {  // the entry point is admitted unconditionally (search.rs:256-261)
            float est, err;
            rabitq_estimate<NW>(a.quant + (size_t)ep * a.rec_len, sh.planes, nw, qc, est, err);
            n_est++;
            rq_admit(L, kk, est, ep, lane, flags);
        }
        uint32_t node, next, pf_node = 0xffffffffu, pf_word = 0;
        for (;;) {
            const uint64_t t0 = now();
            if (!rq_pop(L, lane, node, next)) break;
            // edge record: one coalesced 256 B (128 B above layer 0) load, usually already here — the runner-up's
            // record is requested one expansion ahead
            uint32_t w = node == pf_node ? pf_word : load_edge_raw(a.g, node, layer, lane);
            pf_node = next;
            if (next != 0xffffffffu) pf_word = load_edge_raw(a.g, next, layer, lane);
            const uint32_t deg = lane_u32(w, 0);
            const bool is_edge = lane >= 1 && lane <= (int)deg;
            const uint64_t t1 = now();
            cyc_pop += t1 - t0;
            // the code of every neighbour is requested together with the visited test (one round trip instead of
            // two); codes of already-visited neighbours are dropped
            RqCode<NW> code;
            const uint8_t *rec = a.quant + (size_t)(is_edge ? w : 0u) * a.rec_len;
            if (is_edge) rq_load_code<NW>(rec, code);
            bool fresh = false;
            if (is_edge) {
                if (layer > 0) fresh = vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, w);
                else fresh = (atomicOr(&gvis[w >> 5], 1u << (w & 31)) & (1u << (w & 31))) == 0;
            }
            n_exp++;
            const unsigned long long fm = __ballot(fresh);
            const uint64_t t2 = now();
            cyc_vis += t2 - t1;
            if (layer > 0) {
                vis_count += (uint32_t)__popcll(fm);
                if (vis_count > (3u << RABITQ_UPPER_VIS_LOG2) / 4u) {
                    flags |= NIDX_FLAG_VISITED_OVERFLOW;
                    break;
                }
            }
            float est = 0.f, err = 0.f;
            if (fresh) rq_score_code<NW>(code, rec, sh.planes, nw, qc, est, err);
            n_est += (uint32_t)__popcll(fm);
            const uint64_t t3 = now();
            cyc_est += t3 - t2;
            // `if similarity.score > ws.score || len < k` replayed in edge order (search.rs:287-295)
            unsigned long long todo = fm;
            while (todo) {
                todo = uni64(todo);
                L.len = uni(L.len);
                L.worst = uni64(L.worst);
                const float ws = rank_key_score(L.worst);
                if (L.len >= kk) {
                    todo &= __ballot(fresh && est > ws);
                    if (!todo) break;
                }
                const int j = __ffsll((long long)todo) - 1;
                todo &= ~(1ull << j);
                const float sj = lane_f32(est, j);
                if (sj > ws || L.len < kk) rq_admit(L, kk, sj, lane_u32(w, j), lane, flags);
            }
            cyc_ins += now() - t3;
        }
        ep = rq_addr(lane_u64(L.dir_first, 0));  // layer result (k = 1) = next entry point; layer 0 keeps the whole list
    }

    // ---- rerank_top over the ef neighbours, best estimate first (search.rs:354-363) ----
    const uint64_t t_rr = now();
    Reranker rr;
    rr.init(sh.best, (int)a.k, a.min_score, a.seg.vectors, a.seg.dp, sh.q);
    for (int dch = 0; dch < uni(L.n_dir); dch++) {  // the chunks in rank order = the neighbours best first
        const uint32_t meta = lane_u32(L.dir_meta, dch);
        const bool ok = lane < (int)(meta >> 8);
        uint32_t addr = 0;
        float ub = 0.f;
        if (ok) {
            const uint64_t key = L.chunk(meta & 0xffu)[lane];
            addr = rq_addr(key);
            ub = rank_key_score(key) + rabitq_error(a.quant + (size_t)addr * a.rec_len, qc);
        }
        rr.feed(ok, addr, ub, lane);
    }
    rr.write(a.out_vec + (size_t)qi * a.k, a.out_score + (size_t)qi * a.k, a.out_count + qi, lane);
    if (flags && a.flag_word && lane == 0) atomicOr(a.flag_word, flags);
    if (a.stats && lane == 0) {
        uint32_t *o = a.stats + (size_t)qi * NIDX_STAT_STRIDE;
        o[NIDX_STAT_EVALS] = n_est;
        o[NIDX_STAT_EXPANSIONS] = n_exp;
        o[NIDX_STAT_VISITED] = rr.n_eval;
        o[NIDX_STAT_FLAGS] = flags;
        // cycle split of the walk: pop + edge record / visited test / estimates / admission; [7] = total incl. re-rank
        o[NIDX_STAT_CYC_CTL] = (uint32_t)(cyc_pop + cyc_vis);
        o[NIDX_STAT_EDGE_HITS] = (uint32_t)cyc_est;   // this kernel: cycles in the estimate phase
        o[NIDX_STAT_CYC_INS] = (uint32_t)cyc_ins;
        o[NIDX_STAT_CYC_TOTAL] = (uint32_t)(now() - t_start);
        (void)t_rr;
    }
}

template <int NW>
__global__ __launch_bounds__(64) void rabitq_hnsw_kernel(RabitqSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    rabitq_hnsw1_body<NW>(a, blockIdx.x, smem);
}
// every RaBitQ segment of an index in ONE launch (round 5): block b walks query b % n_queries of segment b / n_queries, whose
// arguments come from a table in HBM (uniform address, read only: scalar loads) — like hnsw_search_segments_kernel.  RaBitQ is the
// reference's default arm of a Dot index with D % 64 == 0 (config.rs:170-173): a launch per segment was 50 launches per batch on the
// reference's 10 M-vector layout.
template <int NW>
__global__ __launch_bounds__(64) void rabitq_hnsw_segments_kernel(const RabitqSearchArgs *table, uint32_t n_queries) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    rabitq_hnsw1_body<NW>(table[blockIdx.x / n_queries], blockIdx.x % n_queries, smem);
}

#define RQ_NONE 0xffffffffu

#ifdef NIDX_RABITQ_EXPERIMENTS   /* make EXPERIMENTS=1: the two-wave walk, measured slower (DESIGN-LOG: RaBitQ, round 5), kept for the comparison */
// ---- HNSW, RaBitQ arm, TWO waves per query (round 5; NOT the default: see the measurements below) ------------------------------
// The walk above is a chain of ~1 100 dependent expansions per query, and a lone wave pays every link in full: the edge record and
// the neighbours' codes are two memory round trips (43 % of the walk's cycles), the admissions ~150 dependent instructions each
// (36 %), with nothing to overlap either (one wave per SIMD at batch 1 024: 0.026 of the HBM roofline, rounds 1-4).  Here a
// workgroup of two waves walks one query:
//   wave 0, the controller  keeps the result set (RqLayer: the directory in its registers), pops, replays the admission rule
//                           `score > ws || len < k` in edge order (search.rs:287-295) — the code of the one-wave kernel;
//   wave 1, the fetcher     expands a node: edge record, the visited test-and-set of every neighbour (layer 0: the per-query
//                           bitset in HBM, atomicOr; above: the LDS hash), the codes of the fresh ones, their estimates; it
//                           leaves (address, estimate) of the fresh neighbours in edge order in LDS.
// While the controller admits the neighbours of expansion i, the fetcher already expands the node the NEXT pop will return —
// predicted exactly: the best of {best unexpanded entry of the result set, best admissible new neighbour}; the best new neighbour
// that beats the current worst result is always admitted, and nothing admitted can rank above it.  The speculation is complete
// (it sets the visited bits), so a confirmed one costs nothing more; the pop that follows the admissions verifies it, and a
// mismatch — possible with exactly tied scores or when the ties side list decides — rolls it back: the fetcher clears exactly
// the bits it set (nobody else writes this query's bitset) and expands the popped node instead.  Upper layers (k = 1, a handful
// of expansions, the LDS hash has no removal) run without speculation.  The fetcher also keeps the edge record of the runner-up
// candidate (read-only, so no rollback): when that node is expanded next, its fetch starts at the codes.
// Results are the one-wave kernel's bit for bit — same admission replay on the same values in the same order; only the moment a
// visited bit is set moves, never whether it is set when an expansion that was really popped tests it
// (tests/test_rabitq_gpu.py, tests/test_serving_gpu.py::test_rabitq_segments_share_one_launch run both kernels).
//
// MEASURED (1 M x 768 clustered, batch 1 024, k = 10; gpurun_out/r5ab): the prediction is right for 99.2 % of the
// expansions and 84 % of them find their edge record held — and the launch is SLOWER than the one-wave kernel's: 7.6 ms against 6.6 ms.
// Per expansion (NIDX_GPU_RABITQ_DEBUG): the fetcher's expansion takes 4.3 k cycles and the controller's admissions + pop 3.6 k, but the
// two stand at their second meeting point for another 2.1 k / 2.7 k cycles each — a meeting costs ~2 k cycles whatever implements it
// (s_barrier, or two sequence words polled in LDS: 8.0 ms), with the waves on one SIMD or on two (a four-wave workgroup whose other
// two waves leave at once), so the overlap the design was made for (4.3 k beside 3.6 k instead of behind it) is spent on the meetings,
// and the prediction + hand-over add 1.7 k more.  The one-wave kernel therefore stays the default (NIDX_GPU_RABITQ_WAVES=2 selects this
// one); what would help is fewer meetings per walk — a fetcher that runs several expansions ahead through a queue — which needs the
// speculation to be exact more than one step ahead (it is not: the best new neighbour of expansion i + 1 is unknown at i).
struct RqFetchBuf {      // one expansion, written by the fetcher
    uint32_t node, n, flags, pad;
    uint32_t addr[64];   // the fresh neighbours in edge order
    float est[64];
};
#define RQ_EDGE_CACHE 3
struct RqCtl {
    uint32_t pred, pred2, pred3, state, fetch_node, abort, ep, pad;
    uint32_t seq[4];                        // [0] / [1]: meeting counts of the controller / the fetcher
    uint32_t cache_node[4];                 // the layer-0 edge records the fetcher holds (RQ_NONE = free)
    uint32_t cache_w[RQ_EDGE_CACHE][64];
};
enum { RQ_STATE_HIT = 0, RQ_STATE_MISS = 1, RQ_STATE_DONE = 2 };

static size_t rq_smem2_bytes(uint32_t nw, uint32_t dp, uint32_t k, uint32_t ef) {
    return rq_smem_bytes(nw, dp, k, ef, true) + 2 * sizeof(RqFetchBuf) + sizeof(RqCtl);
}
#endif  // NIDX_RABITQ_EXPERIMENTS

// the best and (when they sit in the same chunk) second- and third-best unexpanded keys of the result set; 0 = none.  Changes nothing
// but the hint dcur (chunks before it hold expanded keys only — still true afterwards).
__device__ inline void rq_peek3(RqLayer &L, int lane, uint64_t &k1, uint64_t &k2, uint64_t &k3) {
    k1 = 0;
    k2 = 0;
    k3 = 0;
    L.dcur = uni(L.dcur);
    L.n_dir = uni(L.n_dir);
    while (L.dcur < L.n_dir) {
        const uint32_t meta = lane_u32(L.dir_meta, L.dcur);
        const uint64_t mine = lane < (int)(meta >> 8) ? L.chunk(meta & 0xffu)[lane] : 0ull;
        unsigned long long m = __ballot((mine & 1ull) != 0);
        if (m) {
            k1 = lane_u64(mine, __ffsll((long long)m) - 1);
            m &= m - 1;
            if (m) k2 = lane_u64(mine, __ffsll((long long)m) - 1);
            m &= m - 1;
            if (m) k3 = lane_u64(mine, __ffsll((long long)m) - 1);
            return;
        }
        L.dcur++;
    }
}

// the fetcher's expansion of `node` on `layer` -> out.  n2 / n3 (layer 0, or RQ_NONE): the candidates most likely to be expanded after
// it — their edge records are requested along with this expansion's loads and kept in a three-record cache (read-only data: nothing to
// roll back), so that an expansion whose node was foreseen starts at its neighbours' codes: one memory round trip instead of two.
#ifdef NIDX_RABITQ_EXPERIMENTS
template <int NW>
__device__ inline void rq_fetch(const RabitqSearchArgs &a, const RqShared &sh, RqCtl *ctl, RqFetchBuf *out, uint32_t node, uint32_t n2, uint32_t n3,
                                int layer, uint32_t *gvis, const RabitqQueryDev &qc, uint32_t nw, uint32_t &vis_count, uint32_t &cache_hits, int lane) {
    uint32_t w;
    uint32_t pf_node[2] = {RQ_NONE, RQ_NONE}, pf_w[2] = {0u, 0u};
    int pf_slot[2] = {-1, -1};
    if (layer == 0) {
        uint32_t cn[RQ_EDGE_CACHE];
#pragma unroll
        for (int i = 0; i < RQ_EDGE_CACHE; i++) cn[i] = (uint32_t)uni((int)ctl->cache_node[i]);
        int hit = -1;
#pragma unroll
        for (int i = 0; i < RQ_EDGE_CACHE; i++)
            if (cn[i] == node) hit = i;
        // what to request ahead: n2 / n3 unless already held (or the node itself); they take the slots that hold neither of them
        // (the record of `node` is read into registers first, so its slot is free too)
        const uint32_t want[2] = {n2, n3 == n2 ? RQ_NONE : n3};
        bool keep[RQ_EDGE_CACHE];
#pragma unroll
        for (int i = 0; i < RQ_EDGE_CACHE; i++) keep[i] = cn[i] != RQ_NONE && cn[i] != node && (cn[i] == want[0] || cn[i] == want[1]);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (want[j] == RQ_NONE || want[j] == node) continue;
            bool held = false;
#pragma unroll
            for (int i = 0; i < RQ_EDGE_CACHE; i++) held |= cn[i] == want[j];
            if (held) continue;
            int slot = -1;
#pragma unroll
            for (int i = RQ_EDGE_CACHE - 1; i >= 0; i--)
                if (!keep[i]) slot = i;
            if (slot < 0) continue;
            keep[slot] = true;
            pf_node[j] = want[j];
            pf_slot[j] = slot;
            pf_w[j] = load_edge_raw(a.g, want[j], 0, lane);
        }
        // (two separate loads and a select of their VALUES: a select of the two addresses made the compiler emit one FLAT load behind an
        // s_waitcnt vmcnt(0) — the records requested ahead just above had to land before this expansion's own loads could even start)
        const uint32_t w_held = ctl->cache_w[hit >= 0 ? hit : 0][lane];
        uint32_t w_mem = 0;
        if (hit < 0) w_mem = load_edge_raw(a.g, node, 0, lane);
        else cache_hits++;
        asm volatile("" : "+v"(w_mem));   // keeps the two loads apart
        w = hit >= 0 ? w_held : w_mem;
    } else {
        w = load_edge_raw(a.g, node, layer, lane);
    }
    const uint32_t deg = lane_u32(w, 0);
    const bool is_edge = lane >= 1 && lane <= (int)deg;
    // the code of every neighbour is requested together with the visited test (one round trip); codes of visited ones are dropped
    RqCode<NW> code;
    const uint8_t *rec = a.quant + (size_t)(is_edge ? w : 0u) * a.rec_len;
    if (is_edge) rq_load_code<NW>(rec, code);
    bool fresh = false;
    if (is_edge) {
        if (layer > 0) fresh = vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, w);
        else fresh = (atomicOr(&gvis[w >> 5], 1u << (w & 31)) & (1u << (w & 31))) == 0;
    }
    const unsigned long long fm = __ballot(fresh);
    uint32_t oflags = 0;
    if (layer > 0) {
        vis_count += (uint32_t)__popcll(fm);
        if (vis_count > (3u << RABITQ_UPPER_VIS_LOG2) / 4u) oflags = NIDX_FLAG_VISITED_OVERFLOW;
    }
    float est = 0.f, err = 0.f;
    if (fresh) rq_score_code<NW>(code, rec, sh.planes, nw, qc, est, err);
    (void)err;
    const uint32_t pos = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
    if (fresh) {
        out->addr[pos] = w;
        out->est[pos] = est;
    }
    if (lane == 0) {
        out->node = node;
        out->n = (uint32_t)__popcll(fm);
        out->flags = oflags;
    }
    if (layer == 0) {
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (pf_slot[j] >= 0) {
                ctl->cache_w[pf_slot[j]][lane] = pf_w[j];
                if (lane == 0) ctl->cache_node[pf_slot[j]] = pf_node[j];
            }
    }
}

// undo a speculative layer-0 fetch that was not confirmed: clear exactly the visited bits it set; returns after they are cleared
__device__ inline void rq_rollback(const RqFetchBuf *b, uint32_t *gvis, int lane) {
    const uint32_t n = (uint32_t)uni((int)b->n);
    uint32_t old = 0;
    if ((uint32_t)lane < n) {
        const uint32_t x = b->addr[lane];
        old = atomicAnd(&gvis[x >> 5], ~(1u << (x & 31)));
    }
    asm volatile("" ::"v"(old));   // the returned words are waited for: the next fetch must find the bits cleared
}

template <int NW>
__device__ inline void rabitq_hnsw2_body(const RabitqSearchArgs &a, uint32_t qi, unsigned char *smem) {
    const int lane = threadIdx.x & 63;
    const bool w0 = (threadIdx.x >> 6) == 0;
    const uint32_t nw = a.seg.dim / 64u;
    RqShared sh = rq_carve(smem, nw, a.seg.dp, a.k, a.ef);
    RqFetchBuf *buf = reinterpret_cast<RqFetchBuf *>(reinterpret_cast<unsigned char *>(sh.vis) + ((size_t)4 << RABITQ_UPPER_VIS_LOG2));
    RqCtl *ctl = reinterpret_cast<RqCtl *>(buf + 2);
    {
        const uint64_t *gp = a.planes + (size_t)qi * 4u * nw;
        for (uint32_t i = threadIdx.x; i < 4u * nw; i += 128) sh.planes[i] = gp[i];
        const float *gq = a.queries + (size_t)qi * a.seg.dp;
        for (uint32_t i = threadIdx.x; i < a.seg.dp; i += 128) sh.q[i] = gq[i];
    }
    const RabitqQueryDev qc = a.qd[qi];
    uint32_t *gvis = a.visited + (size_t)qi * a.vis_words;  // layer-0 visited bitset (zeroed by the host)
    uint32_t n_est = 0, n_exp = 0, n_hit = 0, flags = 0;   // controller
    uint32_t vis_count = 0, cache_hits = 0;                // fetcher (upper layers' visited count; layer-0 expansions whose edge record was held)
    uint64_t cyc_ctl = 0, cyc_wait = 0, cyc_ins = 0, cyc_fetch = 0;
    uint64_t dbg_b1 = 0, dbg_b2 = 0, dbg_nf = 0;   // this wave's cycles inside barrier 1 / barrier 2 / the need_fetch barrier (NIDX_GPU_RABITQ_DEBUG)
    // (s_memtime is a scalar MEMORY instruction, ~100+ cycles each with its s_waitcnt: the cycle split is taken only when the caller asked
    // for counters or the debug totals — the timed launches of bench.py do not)
    const bool timing = a.stats != nullptr || a.dbg != nullptr;
    auto now = [&]() -> uint64_t { return timing ? (uint64_t)clock64() : 0ull; };
    const uint64_t t_start = now();
    if (threadIdx.x == 0) ctl->seq[0] = ctl->seq[1] = 0;
    __syncthreads();
    // NIDX_GPU_RABITQ_SPEC=0 (measurement): no speculation — the two waves take turns (admit + pop, then fetch)
    const bool speculate = (a.no_speculation & 1u) == 0;
    // The two waves meet twice per expansion: s_barrier.  NIDX_GPU_RABITQ_SPEC=2/3 meets through two sequence words in LDS instead — each
    // wave publishes its count and polls the other's (measured slower: 8.0 ms against 7.6 ms).
    const bool spin = (a.no_speculation & 2u) != 0;
    uint32_t my_seq = 0;
    typedef volatile __attribute__((address_space(3))) uint32_t lds_u32;   // (a generic pointer made these FLAT accesses with system-scope bits)
    lds_u32 *seqw = (lds_u32 *)(&ctl->seq[0]);
    auto meet = [&]() {
        if (!spin) {
            __syncthreads();
            return;
        }
        my_seq++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        seqw[w0 ? 0 : 1] = my_seq;
        while ((uint32_t)uni((int)seqw[w0 ? 1 : 0]) < my_seq) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    uint32_t ep = a.g.ep_node;
    RqLayer L;
    L.res = sh.res;
    L.ties = sh.ties;
    L.spill = a.tie_spill ? a.tie_spill + (size_t)qi * a.tie_stride : nullptr;
    L.spill_cap = (int)a.tie_stride;
    for (int layer = (int)a.g.ep_layer; layer >= 0; layer--) {
        const int kk = layer == 0 ? (int)a.ef : 1;
        if (w0) {
            L.init((int)rq_chunks((uint32_t)kk));
            // the entry point is admitted unconditionally (search.rs:256-261) and is the first pop
            float est, err;
            rabitq_estimate<NW>(a.quant + (size_t)ep * a.rec_len, sh.planes, nw, qc, est, err);
            n_est++;
            rq_admit(L, kk, est, ep, lane, flags);
            uint32_t node = ep, next;
            rq_pop(L, lane, node, next);
            if (lane == 0) {
                ctl->fetch_node = node;
                ctl->pred = RQ_NONE;
                ctl->pred2 = RQ_NONE;
                ctl->pred3 = RQ_NONE;
                ctl->abort = 0;
            }
        } else {
            if (layer > 0) {
                for (uint32_t i = lane; i < (1u << RABITQ_UPPER_VIS_LOG2); i += 64) sh.vis[i] = NIDX_VIS_EMPTY;
                if (lane == 0) vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, ep);
                vis_count = 1;
            } else if (lane == 0) {
                atomicOr(&gvis[ep >> 5], 1u << (ep & 31));
                for (int i = 0; i < 4; i++) ctl->cache_node[i] = RQ_NONE;
            }
        }
        meet();
        int cur = 0;
        bool need_fetch = true;
        for (;;) {
            if (need_fetch) {
                if (!w0) rq_fetch<NW>(a, sh, ctl, &buf[cur], (uint32_t)uni((int)ctl->fetch_node), RQ_NONE, RQ_NONE, layer, gvis, qc, nw, vis_count, cache_hits, lane);
                const uint64_t tw = now();
                meet();
                cyc_wait += now() - tw;
                dbg_nf += now() - tw;
            }
            // ---- the controller takes the expansion in buf[cur] and names the node the next pop will return ----
            bool fresh = false;
            float fest = 0.f;
            uint32_t faddr = 0;
            const uint64_t tc = now();
            if (w0) {
                const RqFetchBuf *b = &buf[cur];
                const uint32_t fn = (uint32_t)uni((int)b->n), bflags = (uint32_t)uni((int)b->flags);
                n_exp++;
                if (bflags) {
                    flags |= bflags;   // the upper-layer visited table is 3/4 full: this layer ends here (like the one-wave kernel)
                    if (lane == 0) {
                        ctl->abort = 1;
                        ctl->pred = RQ_NONE;
                        ctl->pred2 = RQ_NONE;
                        ctl->pred3 = RQ_NONE;
                    }
                } else {
                    n_est += fn;
                    fresh = (uint32_t)lane < fn;
                    fest = fresh ? b->est[lane] : 0.f;
                    faddr = fresh ? b->addr[lane] : 0u;
                    uint32_t p1 = RQ_NONE, p2 = RQ_NONE, p3 = RQ_NONE;
                    if (layer == 0 && speculate) {
                        L.len = uni(L.len);
                        L.worst = uni64(L.worst);
                        const float ws = rank_key_score(L.worst);
                        const bool full = L.len >= kk;
                        const uint64_t key_new = (fresh && (!full || fest > ws)) ? rq_key(fest, faddr, 1u) : 0ull;
                        const uint64_t best_new = wave_max_u64(key_new);
                        uint64_t pk1, pk2, pk3;
                        rq_peek3(L, lane, pk1, pk2, pk3);
                        // the three best of {best new neighbour, pk1 >= pk2 >= pk3}: the first is the prediction, the others the
                        // candidates whose edge records the fetcher requests ahead
                        uint64_t a1, a2, a3;
                        if (best_new > pk1) a1 = best_new, a2 = pk1, a3 = pk2;
                        else if (best_new > pk2) a1 = pk1, a2 = best_new, a3 = pk2;
                        else if (best_new > pk3) a1 = pk1, a2 = pk2, a3 = best_new;
                        else a1 = pk1, a2 = pk2, a3 = pk3;
                        if (a1) p1 = rq_addr(a1);
                        if (a2) p2 = rq_addr(a2);
                        if (a3) p3 = rq_addr(a3);
                    }
                    if (lane == 0) {
                        ctl->pred = p1;
                        ctl->pred2 = p2;
                        ctl->pred3 = p3;
                    }
                }
            }
            cyc_ctl += now() - tc;
            const uint64_t tb1 = now();
            meet();
            dbg_b1 += now() - tb1;
            const uint32_t pred = (uint32_t)uni((int)ctl->pred);
            const bool aborted = uni((int)ctl->abort) != 0;
            if (!w0) {
                if (pred != RQ_NONE) {
                    const uint64_t tf = now();
                    rq_fetch<NW>(a, sh, ctl, &buf[cur ^ 1], pred, (uint32_t)uni((int)ctl->pred2), (uint32_t)uni((int)ctl->pred3), layer, gvis, qc, nw, vis_count, cache_hits, lane);
                    cyc_fetch += now() - tf;
                }
            } else {
                uint32_t st = RQ_STATE_DONE, node = 0, next;
                if (!aborted) {
                    const uint64_t t3 = now();
                    // `if similarity.score > ws.score || len < k` replayed in edge order (search.rs:287-295)
                    unsigned long long todo = __ballot(fresh);
                    while (todo) {
                        todo = uni64(todo);
                        L.len = uni(L.len);
                        L.worst = uni64(L.worst);
                        const float ws = rank_key_score(L.worst);
                        if (L.len >= kk) {
                            todo &= __ballot(fresh && fest > ws);
                            if (!todo) break;
                        }
                        const int j = __ffsll((long long)todo) - 1;
                        todo &= ~(1ull << j);
                        const float sj = lane_f32(fest, j);
                        if (sj > ws || L.len < kk) rq_admit(L, kk, sj, lane_u32(faddr, j), lane, flags);
                    }
                    const uint64_t t4 = now();
                    cyc_ins += t4 - t3;
                    if (rq_pop(L, lane, node, next)) st = node == pred ? RQ_STATE_HIT : RQ_STATE_MISS;
                    cyc_ctl += now() - t4;
                }
                if (st == RQ_STATE_HIT) n_hit++;
                if (lane == 0) {
                    ctl->state = st;
                    ctl->fetch_node = node;
                }
            }
            const uint64_t tw2 = now();
            meet();
            cyc_wait += now() - tw2;
            dbg_b2 += now() - tw2;
            const uint32_t st = (uint32_t)uni((int)ctl->state);
            if (st == RQ_STATE_HIT) {
                cur ^= 1;
                need_fetch = false;
                continue;
            }
            if (!w0 && pred != RQ_NONE) rq_rollback(&buf[cur ^ 1], gvis, lane);   // (layer 0 only: pred is RQ_NONE above it)
            if (st == RQ_STATE_DONE) break;
            need_fetch = true;
        }
        if (w0) {
            ep = rq_addr(lane_u64(L.dir_first, 0));  // layer result (k = 1) = next entry point; layer 0 keeps the whole list
            if (lane == 0) ctl->ep = ep;
        } else if (lane == 0) {
            ctl->pad = cache_hits;
            ctl->cache_node[3] = (uint32_t)(cyc_fetch >> 8);   // (the fourth id slot is not a cache slot)
        }
        meet();
        ep = (uint32_t)uni((int)ctl->ep);
    }
    if (a.dbg && lane == 0) {   // [0..5] controller: barrier 1, barrier 2, fetch barrier, ctl, admissions, walks; [8..13] fetcher: the same barriers, fetches
        unsigned long long *d = a.dbg + (w0 ? 0 : 8);
        atomicAdd(&d[0], (unsigned long long)dbg_b1);
        atomicAdd(&d[1], (unsigned long long)dbg_b2);
        atomicAdd(&d[2], (unsigned long long)dbg_nf);
        atomicAdd(&d[3], (unsigned long long)(w0 ? cyc_ctl : cyc_fetch));
        atomicAdd(&d[4], (unsigned long long)cyc_ins);
        atomicAdd(&d[5], 1ull);
        atomicAdd(&d[6], (unsigned long long)(now() - t_start));
    }
    if (!w0) return;

    // ---- rerank_top over the ef neighbours, best estimate first (search.rs:354-363): the controller alone ----
    Reranker rr;
    rr.init(sh.best, (int)a.k, a.min_score, a.seg.vectors, a.seg.dp, sh.q);
    for (int dch = 0; dch < uni(L.n_dir); dch++) {  // the chunks in rank order = the neighbours best first
        const uint32_t meta = lane_u32(L.dir_meta, dch);
        const bool ok = lane < (int)(meta >> 8);
        uint32_t addr = 0;
        float ub = 0.f;
        if (ok) {
            const uint64_t key = L.chunk(meta & 0xffu)[lane];
            addr = rq_addr(key);
            ub = rank_key_score(key) + rabitq_error(a.quant + (size_t)addr * a.rec_len, qc);
        }
        rr.feed(ok, addr, ub, lane);
    }
    rr.write(a.out_vec + (size_t)qi * a.k, a.out_score + (size_t)qi * a.k, a.out_count + qi, lane);
    if (flags && a.flag_word && lane == 0) atomicOr(a.flag_word, flags);
    if (a.stats && lane == 0) {
        uint32_t *o = a.stats + (size_t)qi * NIDX_STAT_STRIDE;
        o[NIDX_STAT_EVALS] = n_est;
        o[NIDX_STAT_EXPANSIONS] = n_exp;
        o[NIDX_STAT_VISITED] = rr.n_eval;
        o[NIDX_STAT_FLAGS] = flags;
        // the controller's cycles: prediction + pop / expansions whose fetch was speculated and confirmed / admissions; [7] = total incl. re-rank
        // (cycles / 256, 16 bits each) the fetcher's speculative fetches | the controller's prediction + pop + waiting for the fetcher
        o[NIDX_STAT_CYC_CTL] = ((uint32_t)uni((int)ctl->cache_node[3]) & 0xffffu) | ((uint32_t)(((cyc_ctl + cyc_wait) >> 8) & 0xffffu) << 16);
        o[NIDX_STAT_EDGE_HITS] = (n_hit & 0xffffu) | ((uint32_t)uni((int)ctl->pad) << 16);   // confirmed speculations | expansions whose edge record was held
        o[NIDX_STAT_CYC_INS] = (uint32_t)cyc_ins;
        o[NIDX_STAT_CYC_TOTAL] = (uint32_t)(now() - t_start);
    }
}

// (experiment NIDX_GPU_RABITQ_WG=256: the workgroup is launched with four waves of which two leave at once — the hardware deals the
// waves of a workgroup round-robin over the CU's SIMDs, so the controller and the fetcher then surely sit on different SIMDs)
template <int NW>
__global__ __launch_bounds__(256) void rabitq_hnsw2_kernel(RabitqSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x >= 128) return;
    rabitq_hnsw2_body<NW>(a, blockIdx.x, smem);
}

// every RaBitQ segment of an index in ONE launch: block b walks query b % n_queries of segment b / n_queries, whose arguments
// come from a table in HBM (uniform address, read only: scalar loads) — like hnsw_search_segments_kernel
template <int NW>
__global__ __launch_bounds__(128) void rabitq_hnsw2_segments_kernel(const RabitqSearchArgs *table, uint32_t n_queries) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RabitqSearchArgs &a = table[blockIdx.x / n_queries];
    rabitq_hnsw2_body<NW>(a, blockIdx.x % n_queries, smem);
}
#endif  // NIDX_RABITQ_EXPERIMENTS

// ---- HNSW, RaBitQ arm, ONE wave per query with the next expansion's loads in flight under the admissions (round 5) --------------------
// What the two-wave walk above was built for — the memory round trip of expansion i + 1 hidden behind the admissions of expansion i —
// without a second wave and without meetings: a wave's own loads are asynchronous.  After the estimates of an expansion are known, the node
// the next pop will return is predicted exactly as above (best of {best unexpanded entry of the result set, best admissible new
// neighbour}); when its layer-0 edge record is already in registers (the records of the three best candidates are requested one
// expansion ahead: read-only, 84 % of the expansions find theirs), the codes of its neighbours and their visited test-and-set are ISSUED
// — and only then the admission rule is replayed for the current expansion's neighbours (LDS and scalar work, ~3 000 cycles): the
// loads land meanwhile.  The pop verifies the prediction; a mismatch clears exactly the bits the speculative test-and-set set (nobody else
// writes this query's bitset) and expands the popped node the plain way — a safety net: only nodes that were candidates one expansion ago
// have their record in registers, and when such a node outranks every neighbour about to be admitted, the admissions can neither put
// anything in front of it nor evict it (tests/test_rabitq_walk_model_cpu.py models this walk against the plain one, ties included).  Upper layers
// (a handful of expansions, an LDS hash without removal) run the plain loop.  Results are the plain kernel's bit for bit.
template <int NW>
__device__ inline void rabitq_hnsw3_body(const RabitqSearchArgs &a, const uint32_t qi, unsigned char *smem) {
    const int lane = threadIdx.x;
    const uint32_t nw = a.seg.dim / 64u;
    // LDS of this kernel (rq_smem3_bytes): planes | best | res | ties | [upper-layer table, if res cannot lend the space] | [seen cache].
    // The upper layers run with k = 1 — two chunks of `res` — so their visited table lives in the rest of `res` when that is large enough
    // (ef >= 576); the raw query stays in HBM / L2 (only the re-rank reads it, beside the rows it fetches anyway).  At D = 768, ef = 1 000:
    // 18.3 KiB + the cache (2 KiB) instead of 29.5 KiB: seven resident walks per CU instead of five.
    RqShared sh = rq_carve(smem, nw, a.seg.dp, a.k, a.ef, false);
    rq_load_query(sh, a, qi, nw, lane, false);
    uint32_t *seen = sh.vis;   // (rq_carve: the region behind `ties`)
    if ((rq_chunks(a.ef) - 2u) * 512u >= (4u << RABITQ_UPPER_VIS_LOG2)) sh.vis = reinterpret_cast<uint32_t *>(sh.res + 2 * 64);
    else seen += 1u << RABITQ_UPPER_VIS_LOG2;
    const RabitqQueryDev qc = a.qd[qi];
    uint32_t *gvis = a.visited + (size_t)qi * a.vis_words;  // layer-0 visited bitset (zeroed by the host)
    uint32_t n_est = 0, n_exp = 0, n_hit = 0, n_ask = 0, flags = 0;
    uint64_t cyc_ins = 0;
    const bool timing = a.stats != nullptr;
    auto now = [&]() -> uint64_t { return timing ? (uint64_t)clock64() : 0ull; };
    const uint64_t t_start = now();

    uint32_t ep = a.g.ep_node;
    RqLayer L;
    L.res = sh.res;
    L.ties = sh.ties;
    L.spill = a.tie_spill ? a.tie_spill + (size_t)qi * a.tie_stride : nullptr;
    L.spill_cap = (int)a.tie_stride;
    // ---- upper layers: the plain loop (k = 1) ----
    for (int layer = (int)a.g.ep_layer; layer >= 1; layer--) {
        L.init((int)rq_chunks(1u));
        for (uint32_t i = lane; i < (1u << RABITQ_UPPER_VIS_LOG2); i += 64) sh.vis[i] = NIDX_VIS_EMPTY;
        if (lane == 0) vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, ep);
        uint32_t vis_count = 1;
        {
            float est, err;
            rabitq_estimate<NW>(a.quant + (size_t)ep * a.rec_len, sh.planes, nw, qc, est, err);
            n_est++;
            rq_admit(L, 1, est, ep, lane, flags);
        }
        uint32_t node, next;
        for (;;) {
            if (!rq_pop(L, lane, node, next)) break;
            const uint32_t w = load_edge_raw(a.g, node, layer, lane);
            const uint32_t deg = lane_u32(w, 0);
            const bool is_edge = lane >= 1 && lane <= (int)deg;
            RqCode<NW> code;
            const uint8_t *rec = a.quant + (size_t)(is_edge ? w : 0u) * a.rec_len;
            if (is_edge) rq_load_code<NW>(rec, code);
            bool fresh = false;
            if (is_edge) fresh = vis_insert(sh.vis, RABITQ_UPPER_VIS_LOG2, w);
            n_exp++;
            const unsigned long long fm = __ballot(fresh);
            vis_count += (uint32_t)__popcll(fm);
            if (vis_count > (3u << RABITQ_UPPER_VIS_LOG2) / 4u) {
                flags |= NIDX_FLAG_VISITED_OVERFLOW;
                break;
            }
            float est = 0.f, err = 0.f;
            if (fresh) rq_score_code<NW>(code, rec, sh.planes, nw, qc, est, err);
            n_est += (uint32_t)__popcll(fm);
            unsigned long long todo = fm;
            while (todo) {
                todo = uni64(todo);
                L.len = uni(L.len);
                L.worst = uni64(L.worst);
                const float ws = rank_key_score(L.worst);
                if (L.len >= 1) {
                    todo &= __ballot(fresh && est > ws);
                    if (!todo) break;
                }
                const int j = __ffsll((long long)todo) - 1;
                todo &= ~(1ull << j);
                const float sj = lane_f32(est, j);
                if (sj > ws || L.len < 1) rq_admit(L, 1, sj, lane_u32(w, j), lane, flags);
            }
        }
        ep = rq_addr(lane_u64(L.dir_first, 0));
    }

    // ---- layer 0 ----
    {
        const int kk = (int)a.ef;
        L.init((int)rq_chunks((uint32_t)kk));
        if (lane == 0) atomicOr(&gvis[ep >> 5], 1u << (ep & 31));
        // Of the ~60 neighbours an expansion tests, ~55 have been visited before; their test-and-set and their codes are most of what the walk asks
        // of the memory system (1 024 bitsets of n bits do not fit the L2: three of four requests miss).  An LDS table of 2^seen_log2
        // words is a direct-mapped cache of node ids KNOWN to be visited (entered only once the bit is surely set: when an expansion is
        // no longer speculative).  A hit is exact — the lane skips the atomic and the code; a miss asks the bitset as before.
        const uint32_t seen_log2 = a.seen_log2;   // 0: no cache
        const bool use_seen = seen_log2 != 0;
        if (use_seen)
            for (uint32_t i = lane; i < (1u << seen_log2); i += 64) seen[i] = NIDX_VIS_EMPTY;
        const uint32_t seen_shift = 32u - seen_log2;
        auto seen_slot = [&](uint32_t v) -> uint32_t { return (v * 2654435761u) >> seen_shift; };
        {
            float est, err;
            rabitq_estimate<NW>(a.quant + (size_t)ep * a.rec_len, sh.planes, nw, qc, est, err);
            n_est++;
            rq_admit(L, kk, est, ep, lane, flags);
        }
        // edge records held in registers: hn* = node (wave-uniform), hw* = this lane's word of its record
        uint32_t hn0 = RQ_NONE, hn1 = RQ_NONE, hn2 = RQ_NONE, hw0 = 0, hw1 = 0, hw2 = 0;
        // the speculated expansion
        bool have_spec = false, spec_edge = false;
        uint32_t spec_node = RQ_NONE, spec_w = 0, spec_old = 0;
        RqCode<NW> spec_code;
        uint32_t node, next;
        for (;;) {
            if (!rq_pop(L, lane, node, next)) break;
            uint32_t w;
            bool is_edge, fresh = false;
            RqCode<NW> code;
            const uint8_t *rec;
            if (have_spec && spec_node == node) {
                // confirmed: its loads have been in flight since before the last admissions
                w = spec_w;
                is_edge = spec_edge;
                rec = a.quant + (size_t)(is_edge ? w : 0u) * a.rec_len;
                code = spec_code;
                fresh = is_edge && (spec_old & (1u << (w & 31))) == 0;
                n_hit++;
            } else {
                if (have_spec) {
                    // not confirmed: clear exactly the bits the speculative test-and-set set; the returned words are waited for
                    uint32_t old = 0;
                    if (spec_edge && (spec_old & (1u << (spec_w & 31))) == 0) old = atomicAnd(&gvis[spec_w >> 5], ~(1u << (spec_w & 31)));
                    asm volatile("" ::"v"(old));
                }
                const uint32_t w_held = node == hn0 ? hw0 : node == hn1 ? hw1 : hw2;
                uint32_t w_mem = 0;
                const bool held = node == hn0 || node == hn1 || node == hn2;
                if (!held) w_mem = load_edge_raw(a.g, node, 0, lane);
                w = held ? w_held : w_mem;
                const uint32_t deg = lane_u32(w, 0);
                is_edge = lane >= 1 && lane <= (int)deg;
                rec = a.quant + (size_t)(is_edge ? w : 0u) * a.rec_len;
                const bool ask = is_edge && !(use_seen && seen[seen_slot(w)] == w);
                if (ask) rq_load_code<NW>(rec, code);
                if (ask) fresh = (atomicOr(&gvis[w >> 5], 1u << (w & 31)) & (1u << (w & 31))) == 0;
                n_ask += (uint32_t)__popcll(__ballot(ask));
            }
            if (use_seen && is_edge) seen[seen_slot(w)] = w;   // every neighbour of an expanded node is visited from here on
            have_spec = false;
            n_exp++;
            const unsigned long long fm = __ballot(fresh);
            float est = 0.f, err = 0.f;
            if (fresh) rq_score_code<NW>(code, rec, sh.planes, nw, qc, est, err);
            n_est += (uint32_t)__popcll(fm);
            // ---- the next pop, predicted; its expansion's loads issued before the admissions below ----
            uint32_t p1 = RQ_NONE, p2 = RQ_NONE, p3 = RQ_NONE;
            {
                L.len = uni(L.len);
                L.worst = uni64(L.worst);
                const float ws = rank_key_score(L.worst);
                const bool full = L.len >= kk;
                const uint64_t key_new = (fresh && (!full || est > ws)) ? rq_key(est, w, 1u) : 0ull;
                const uint64_t best_new = wave_max_u64(key_new);
                uint64_t pk1, pk2, pk3;
                rq_peek3(L, lane, pk1, pk2, pk3);
                uint64_t a1, a2, a3;
                if (best_new > pk1) a1 = best_new, a2 = pk1, a3 = pk2;
                else if (best_new > pk2) a1 = pk1, a2 = best_new, a3 = pk2;
                else if (best_new > pk3) a1 = pk1, a2 = pk2, a3 = best_new;
                else a1 = pk1, a2 = pk2, a3 = pk3;
                if (a1) p1 = rq_addr(a1);
                if (a2) p2 = rq_addr(a2);
                if (a3) p3 = rq_addr(a3);
            }
            if (p1 != RQ_NONE && (p1 == hn0 || p1 == hn1 || p1 == hn2)) {
                spec_node = p1;
                spec_w = p1 == hn0 ? hw0 : p1 == hn1 ? hw1 : hw2;
                const uint32_t deg2 = lane_u32(spec_w, 0);
                spec_edge = lane >= 1 && lane <= (int)deg2;
                const uint8_t *rec2 = a.quant + (size_t)(spec_edge ? spec_w : 0u) * a.rec_len;
                const bool ask = spec_edge && !(use_seen && seen[seen_slot(spec_w)] == spec_w);
                if (ask) rq_load_code<NW>(rec2, spec_code);
                spec_old = 0xffffffffu;   // a lane that does not ask knows its node visited
                if (ask) spec_old = atomicOr(&gvis[spec_w >> 5], 1u << (spec_w & 31));
                n_ask += (uint32_t)__popcll(__ballot(ask));
                have_spec = true;
            }
            // the records of the three candidates stay / come into the held set (a slot that holds none of them is overwritten)
            {
                const bool k0 = hn0 != RQ_NONE && (hn0 == p1 || hn0 == p2 || hn0 == p3);
                const bool k1 = hn1 != RQ_NONE && (hn1 == p1 || hn1 == p2 || hn1 == p3);
                const bool k2 = hn2 != RQ_NONE && (hn2 == p1 || hn2 == p2 || hn2 == p3);
                bool f0 = !k0, f1 = !k1, f2 = !k2;
                const uint32_t want[3] = {p1, p2, p3};
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const uint32_t n = want[j];
                    if (n == RQ_NONE || n == hn0 || n == hn1 || n == hn2) continue;
                    if (j == 1 && n == want[0]) continue;
                    if (j == 2 && (n == want[0] || n == want[1])) continue;
                    if (f0) {
                        hw0 = load_edge_raw(a.g, n, 0, lane);
                        hn0 = n;
                        f0 = false;
                    } else if (f1) {
                        hw1 = load_edge_raw(a.g, n, 0, lane);
                        hn1 = n;
                        f1 = false;
                    } else if (f2) {
                        hw2 = load_edge_raw(a.g, n, 0, lane);
                        hn2 = n;
                        f2 = false;
                    }
                }
            }
            // ---- `if similarity.score > ws.score || len < k` replayed in edge order (search.rs:287-295) ----
            const uint64_t t3 = now();
            unsigned long long todo = fm;
            while (todo) {
                todo = uni64(todo);
                L.len = uni(L.len);
                L.worst = uni64(L.worst);
                const float ws = rank_key_score(L.worst);
                if (L.len >= kk) {
                    todo &= __ballot(fresh && est > ws);
                    if (!todo) break;
                }
                const int j = __ffsll((long long)todo) - 1;
                todo &= ~(1ull << j);
                const float sj = lane_f32(est, j);
                if (sj > ws || L.len < kk) rq_admit(L, kk, sj, lane_u32(w, j), lane, flags);
            }
            cyc_ins += now() - t3;
        }
    }

    // ---- rerank_top over the ef neighbours, best estimate first (search.rs:354-363) ----
    Reranker rr;
    rr.init(sh.best, (int)a.k, a.min_score, a.seg.vectors, a.seg.dp, a.queries + (size_t)qi * a.seg.dp);
    for (int dch = 0; dch < uni(L.n_dir); dch++) {
        const uint32_t meta = lane_u32(L.dir_meta, dch);
        const bool ok = lane < (int)(meta >> 8);
        uint32_t addr = 0;
        float ub = 0.f;
        if (ok) {
            const uint64_t key = L.chunk(meta & 0xffu)[lane];
            addr = rq_addr(key);
            ub = rank_key_score(key) + rabitq_error(a.quant + (size_t)addr * a.rec_len, qc);
        }
        rr.feed(ok, addr, ub, lane);
    }
    rr.write(a.out_vec + (size_t)qi * a.k, a.out_score + (size_t)qi * a.k, a.out_count + qi, lane);
    if (flags && a.flag_word && lane == 0) atomicOr(a.flag_word, flags);
    if (a.stats && lane == 0) {
        uint32_t *o = a.stats + (size_t)qi * NIDX_STAT_STRIDE;
        o[NIDX_STAT_EVALS] = n_est;
        o[NIDX_STAT_EXPANSIONS] = n_exp;
        o[NIDX_STAT_VISITED] = rr.n_eval;
        o[NIDX_STAT_FLAGS] = flags;
        o[NIDX_STAT_CYC_CTL] = n_ask;     // neighbours whose visited bit and code were asked of memory (the rest were known visited: LDS)
        o[NIDX_STAT_EDGE_HITS] = n_hit;   // expansions whose loads were in flight under the previous admissions
        o[NIDX_STAT_CYC_INS] = (uint32_t)cyc_ins;
        o[NIDX_STAT_CYC_TOTAL] = (uint32_t)(now() - t_start);
    }
}

template <int NW>
__global__ __launch_bounds__(64) void rabitq_hnsw3_kernel(RabitqSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    rabitq_hnsw3_body<NW>(a, blockIdx.x, smem);
}
template <int NW>
__global__ __launch_bounds__(64) void rabitq_hnsw3_segments_kernel(const RabitqSearchArgs *table, uint32_t n_queries) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    rabitq_hnsw3_body<NW>(table[blockIdx.x / n_queries], blockIdx.x % n_queries, smem);
}

// ---- launchers -------------------------------------------------------------------------------------------
hipError_t launch_rabitq_encode(const float *vectors, uint32_t n, uint32_t dp, uint32_t dim, uint8_t *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rabitq_encode_kernel, dim3((n + 3) / 4), dim3(256), 0, s, vectors, n, dp, dim, out, dim / 8 + 8);
    return hipGetLastError();
}
hipError_t launch_rabitq_query(const float *queries, uint32_t nq, uint32_t dp, uint32_t dim, RabitqQueryDev *qd,
                               uint64_t *planes, hipStream_t s) {
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(rabitq_query_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, queries, nq, dp, dim, qd, planes);
    return hipGetLastError();
}
// The dynamic-LDS ceiling of a kernel is an attribute of the FUNCTION, process-wide, while launches of different indexes (other k,
// ef, seen_log2: other sizes) come from different threads: it is only ever raised, under one lock, so a launch never finds it below
// what it asked for (setting it before every launch let a smaller request of another thread slip in between set and launch).
static hipError_t rq_allow_lds(const void *fn, size_t smem) {
    static std::mutex mu;
    static std::unordered_map<const void *, size_t> allowed;
    std::lock_guard<std::mutex> lk(mu);
    size_t &cur = allowed[fn];
    if (smem <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) cur = smem;
    return e;
}
template <int NW>
static hipError_t launch_bf_nw(const RabitqSearchArgs &a, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_bf_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_bf_kernel<NW>, dim3(a.n_queries), dim3(64), smem, s, a);
    return hipGetLastError();
}
template <int NW>
static hipError_t launch_hnsw_nw(const RabitqSearchArgs &a, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_hnsw_kernel<NW>, dim3(a.n_queries), dim3(64), smem, s, a);
    return hipGetLastError();
}
// word counts with a fully unrolled code fetch: 128 .. 2048 dimensions of the usual embedding models
#define RQ_DISPATCH(fn, nw, ...)                     \
    switch (nw) {                                    \
        case 2: return fn<2>(__VA_ARGS__);           \
        case 4: return fn<4>(__VA_ARGS__);           \
        case 6: return fn<6>(__VA_ARGS__);           \
        case 8: return fn<8>(__VA_ARGS__);           \
        case 12: return fn<12>(__VA_ARGS__);         \
        case 16: return fn<16>(__VA_ARGS__);         \
        case 24: return fn<24>(__VA_ARGS__);         \
        case 32: return fn<32>(__VA_ARGS__);         \
        default: return fn<0>(__VA_ARGS__);          \
    }
hipError_t launch_rabitq_bf(const RabitqSearchArgs &a, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    const size_t smem = rq_smem_bytes(a.seg.dim / 64u, a.seg.dp, a.k, 0, false);
    RQ_DISPATCH(launch_bf_nw, a.seg.dim / 64u, a, smem, s)
}
#ifdef NIDX_RABITQ_EXPERIMENTS
template <int NW>
static hipError_t launch_hnsw2_nw(const RabitqSearchArgs &a, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw2_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    const char *wg = getenv("NIDX_GPU_RABITQ_WG");
    hipLaunchKernelGGL(rabitq_hnsw2_kernel<NW>, dim3(a.n_queries), dim3(wg && atoi(wg) == 256 ? 256 : 128), smem, s, a);
    return hipGetLastError();
}
template <int NW>
static hipError_t launch_hnsw2_segments_nw(const RabitqSearchArgs *table, uint32_t n_table, uint32_t nq, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw2_segments_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_hnsw2_segments_kernel<NW>, dim3(n_table * nq), dim3(128), smem, s, table, nq);
    return hipGetLastError();
}
#endif
template <int NW>
static hipError_t launch_hnsw1_segments_nw(const RabitqSearchArgs *table, uint32_t n_table, uint32_t nq, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw_segments_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_hnsw_segments_kernel<NW>, dim3(n_table * nq), dim3(64), smem, s, table, nq);
    return hipGetLastError();
}
template <int NW>
static hipError_t launch_hnsw3_nw(const RabitqSearchArgs &a, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw3_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_hnsw3_kernel<NW>, dim3(a.n_queries), dim3(64), smem, s, a);
    return hipGetLastError();
}
template <int NW>
static hipError_t launch_hnsw3_segments_nw(const RabitqSearchArgs *table, uint32_t n_table, uint32_t nq, size_t smem, hipStream_t s) {
    hipError_t e = rq_allow_lds(reinterpret_cast<const void *>(&rabitq_hnsw3_segments_kernel<NW>), smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rabitq_hnsw3_segments_kernel<NW>, dim3(n_table * nq), dim3(64), smem, s, table, nq);
    return hipGetLastError();
}
// NIDX_GPU_RABITQ_PIPE=0: the plain one-wave walk (rounds 1-4); default: the walk with the next expansion's loads in flight under the
// admissions (dimensions whose code fetch is unrolled: every D that is a multiple of 64 up to 2 048 except the odd word counts)
static bool rabitq_pipelined(uint32_t nw) {
    const char *e = getenv("NIDX_GPU_RABITQ_PIPE");
    if (e && atoi(e) == 0) return false;
    return nw == 2 || nw == 4 || nw == 6 || nw == 8 || nw == 12 || nw == 16 || nw == 24 || nw == 32;
}
// measurement switches of the two-wave walk -> RabitqSearchArgs::no_speculation: NIDX_GPU_RABITQ_SPEC=0: no speculation (bit 0);
// =2: the waves meet through polled LDS words instead of s_barrier (bit 1); =3: both
[[maybe_unused]] static uint32_t rabitq_walk_mode() {
    const char *e = getenv("NIDX_GPU_RABITQ_SPEC");
    if (!e) return 0u;
    const int v = atoi(e);
    return v == 0 ? 1u : v == 2 ? 2u : v == 3 ? 3u : 0u;
}
// log2 of the pipelined walk's LDS cache of known-visited ids: 9 (512 ids, 2 KiB) by default; NIDX_GPU_RABITQ_SEEN=0: none (every neighbour
// asks the bitset in HBM), 8 ... 13: that size — measurement.  1 M x 768, k = 10, three batches of 1 024 in flight (scripts/r5_ab.sh rqseen /
// rqflight): no cache 50.1 k neighbours per query ask memory, 251 k queries/s; 2^9: 22.5 k, 276 k (286 k with six in flight); 2^10: 16.9 k,
// 258 k; 2^11: 12.3 k, 240 k; 2^12: 9.2 k, 188 k — a launch alone takes the same 6.4 ms with any of them (the walk is bound by its wave's
// instruction stream), what the sustained rate follows is the number of walks resident per CU (LDS: 8 / 7 / 7 / 6 / 4).
uint32_t rabitq_seen_log2() {
    const char *e = getenv("NIDX_GPU_RABITQ_SEEN");
    if (!e) return 9u;
    const int v = atoi(e);
    return v <= 0 ? 0u : v < 8 ? 8u : v > 13 ? 13u : (uint32_t)v;
}
// NIDX_GPU_RABITQ_WAVES=2: the two-wave walk (slower on MI355X as measured, kept for the comparison); default: one wave per query
bool rabitq_two_waves() {
#ifdef NIDX_RABITQ_EXPERIMENTS
    const char *e = getenv("NIDX_GPU_RABITQ_WAVES");
    return e && atoi(e) == 2;
#else
    return false;   // (the two-wave walk is only in a `make EXPERIMENTS=1` library)
#endif
}
bool rabitq_tie_spill_enabled() {
    const char *e = getenv("NIDX_GPU_RABITQ_TIE_SPILL");
    return !(e && atoi(e) == 0);
}
bool rabitq_has_experiments() {
#ifdef NIDX_RABITQ_EXPERIMENTS
    return true;
#else
    return false;
#endif
}
// the plain one-wave walk: the product runs it for dimensions whose code fetch is not unrolled (NW = 0: D / 64 read at run time) and under
// NIDX_GPU_RABITQ_PIPE=0 (parity tests); its unrolled instances are measurement material
#ifdef NIDX_RABITQ_EXPERIMENTS
#define RQ_DISPATCH_PLAIN(fn, nw, ...) RQ_DISPATCH(fn, nw, __VA_ARGS__)
#else
#define RQ_DISPATCH_PLAIN(fn, nw, ...) return fn<0>(__VA_ARGS__);
#endif
hipError_t launch_rabitq_hnsw(const RabitqSearchArgs &a, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
#ifdef NIDX_RABITQ_EXPERIMENTS
    if (rabitq_two_waves()) {
        const size_t smem2 = rq_smem2_bytes(a.seg.dim / 64u, a.seg.dp, a.k, a.ef);
        RabitqSearchArgs b = a;
        b.no_speculation = rabitq_walk_mode();
        if (getenv("NIDX_GPU_RABITQ_DEBUG")) {
            // measurement only: where the two waves of a walk spend their cycles (synchronises; prints one line per launch)
            static unsigned long long *d_dbg = nullptr;
            if (!d_dbg && hipMalloc(&d_dbg, 16 * 8) != hipSuccess) return hipErrorOutOfMemory;
            (void)hipMemsetAsync(d_dbg, 0, 16 * 8, s);
            b.dbg = d_dbg;
            hipError_t e = [&]() -> hipError_t { RQ_DISPATCH(launch_hnsw2_nw, a.seg.dim / 64u, b, smem2, s) }();
            if (e != hipSuccess) return e;
            unsigned long long h[16];
            (void)hipMemcpyAsync(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            const double n = h[5] ? (double)h[5] : 1.0;
            fprintf(stderr, "[rabitq dbg] per walk, controller: barrier1 %.0f barrier2 %.0f fetch-barrier %.0f ctl %.0f admissions %.0f walk %.0f | fetcher: barrier1 %.0f barrier2 %.0f "
                            "fetch-barrier %.0f speculative fetches %.0f walk %.0f cycles\n", h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[8] / n, h[9] / n, h[10] / n, h[11] / n, h[14] / n);
            return hipSuccess;
        }
        RQ_DISPATCH(launch_hnsw2_nw, a.seg.dim / 64u, b, smem2, s)
    }
#endif
    const size_t smem = rq_smem_bytes(a.seg.dim / 64u, a.seg.dp, a.k, a.ef, true);
    if (rabitq_pipelined(a.seg.dim / 64u)) {
        RabitqSearchArgs b = a;
        b.seen_log2 = rabitq_seen_log2();
        const size_t smem3 = rq_smem3_bytes(a.seg.dim / 64u, a.k, a.ef, b.seen_log2);
        RQ_DISPATCH(launch_hnsw3_nw, a.seg.dim / 64u, b, smem3, s)
    }
    RQ_DISPATCH_PLAIN(launch_hnsw_nw, a.seg.dim / 64u, a, smem, s)
}
// `table` (device) holds n_table argument records that agree in dim / dp / k / ef / n_queries (`shape`: one of them, host side)
hipError_t launch_rabitq_hnsw_segments(const RabitqSearchArgs *table, uint32_t n_table, const RabitqSearchArgs &shape, hipStream_t s) {
    if (n_table == 0 || shape.n_queries == 0) return hipSuccess;
#ifdef NIDX_RABITQ_EXPERIMENTS
    if (rabitq_two_waves()) {
        const size_t smem2 = rq_smem2_bytes(shape.seg.dim / 64u, shape.seg.dp, shape.k, shape.ef);
        RQ_DISPATCH(launch_hnsw2_segments_nw, shape.seg.dim / 64u, table, n_table, shape.n_queries, smem2, s)
    }
#endif
    const size_t smem = rq_smem_bytes(shape.seg.dim / 64u, shape.seg.dp, shape.k, shape.ef, true);
    if (rabitq_pipelined(shape.seg.dim / 64u)) {
        const size_t smem3 = rq_smem3_bytes(shape.seg.dim / 64u, shape.k, shape.ef, shape.seen_log2);   // (every record of the table: rabitq_seen_log2())
        RQ_DISPATCH(launch_hnsw3_segments_nw, shape.seg.dim / 64u, table, n_table, shape.n_queries, smem3, s)
    }
    RQ_DISPATCH_PLAIN(launch_hnsw1_segments_nw, shape.seg.dim / 64u, table, n_table, shape.n_queries, smem, s)
}

}  // namespace nidx
