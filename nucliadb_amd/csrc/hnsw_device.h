// hnsw_device.h — block-cooperative HNSW traversal primitives (gfx950).
//
// One workgroup of W waves serves one query.  Wave 0 is the controller: it owns the result set
// (a sorted list one-entry-per-lane in registers), pops the best candidate from an unsorted LDS
// pool (wave-parallel arg-max), reads that node's fixed-stride edge record with ONE coalesced
// 256-byte load, and tests/marks the visited set (an open-addressing hash in LDS).  All W waves
// then evaluate the similarity of the unvisited neighbours, each row as NJ coalesced 1 KiB loads
// (16 B per lane), four rows in flight per wave, reduced with the transposed butterfly (WAVE64
// order — bit-identical to the oracle).  The controller finally replays the reference's
// sequential admission rule over the scores in edge order.
//
// Restates HnswSearcher::layer_search / closest_up_nodes / search
// (nidx_vector/src/hnsw/search.rs:188-383).  Where the reference leaves tie order to BinaryHeap
// internals (score-only Ord, search.rs:89-124) we use the strict total order of rank_key():
// higher score first, then lower address — the same rule the oracle uses.
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace nidx {

#define NIDX_POOL_CAP 512
#define NIDX_VIS_EMPTY 0xffffffffu

struct SearchShared {
    uint64_t pool[NIDX_POOL_CAP];  // unexpanded candidates (rank keys), unsorted
    uint32_t nb_addr[64];          // neighbours to evaluate
    float nb_ab[64];               // <x, q>
    float nb_xx[64];               // |x|^2
    uint32_t eps[256];             // entry points for the next layer search
    int ctrl[8];                   // [0] continue, [1] n_new, [2] n_eps
};

template <int NJ>
struct QueryRegs {
    float4 qv[NJ];
    float qq;        // |q|^2 in WAVE64 order
    double sqrt_qq;
};

template <int NJ>
__device__ inline void load_query(QueryRegs<NJ> &q, const float *qrow, uint32_t dp, int lane, bool cosine) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        q.qv[j] = load_row_chunk(qrow, dp, j, lane);
        acc = fma4(q.qv[j], q.qv[j], acc);
    }
    q.qq = wave_butterfly_sum(acc);
    q.sqrt_qq = cosine ? sqrt((double)q.qq) : 0.0;
}

// similarity from the pre-reduced sums; mirrors cosine_from_sums() with sqrt(|q|^2) hoisted.
__device__ inline float score_from_sums(float ab, float xx, float qq, double sqrt_qq, bool cosine) {
    if (!cosine) return ab;
    double dab = (double)ab, dxx = (double)xx;
    double dist;
    if (dxx == 0.0 && (double)qq == 0.0) dist = 0.0;
    else if (dab == 0.0) dist = 1.0;
    else {
        double d = 1.0 - dab / (sqrt(dxx) * sqrt_qq);
        dist = d > 0.0 ? d : 0.0;
    }
    return 1.0f - (float)dist;
}

// ---- visited set ------------------------------------------------------------------------------
__device__ inline void vis_clear(uint32_t *vis, uint32_t cap) {
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) vis[i] = NIDX_VIS_EMPTY;
}
// true when v was not present (and is now)
__device__ inline bool vis_insert(uint32_t *vis, uint32_t log2cap, uint32_t v) {
    const uint32_t mask = (1u << log2cap) - 1u;
    uint32_t h = (v * 2654435761u) >> (32 - log2cap);
    for (;;) {
        uint32_t old = atomicCAS(&vis[h], NIDX_VIS_EMPTY, v);
        if (old == NIDX_VIS_EMPTY) return true;
        if (old == v) return false;
        h = (h + 1) & mask;
    }
}

// ---- candidate pool: unsorted array in LDS, arg-max pop --------------------------------------
__device__ inline uint64_t wave_max_u64(uint64_t v) { return wave_extreme_u64<true>(v); }
__device__ inline uint64_t wave_min_u64(uint64_t v) { return wave_extreme_u64<false>(v); }
// Pops the best key (EMPTY if none).  Wave-0 only; pool_len is wave-uniform.
__device__ inline uint64_t pool_pop(uint64_t *pool, int &pool_len, int lane) {
    if (pool_len == 0) return NIDX_EMPTY_KEY;
    uint64_t best = NIDX_EMPTY_KEY;
    for (int i = lane; i < pool_len; i += 64) {
        uint64_t v = pool[i];
        best = v > best ? v : best;
    }
    best = wave_max_u64(best);
    // remove one occurrence: the lowest index holding `best`
    int idx = 0x7fffffff;
    for (int i = lane; i < pool_len; i += 64)
        if (pool[i] == best && i < idx) idx = i;
    idx = wave_min_i32(idx);
    uint64_t last = pool[pool_len - 1];
    if (lane == 0) pool[idx] = last;
    pool_len--;
    return best;
}
// Removes every entry whose score is < ws (they can never be expanded once the result set is
// full: ws only grows).  Wave-0 only.
__device__ inline void pool_prune(uint64_t *pool, int &pool_len, float ws, int lane) {
    int out = 0;
    for (int base = 0; base < pool_len; base += 64) {
        int i = base + lane;
        uint64_t v = i < pool_len ? pool[i] : NIDX_EMPTY_KEY;
        bool keep = i < pool_len && !(rank_key_score(v) < ws);
        unsigned long long m = __ballot(keep);
        int pos = out + __popcll(m & ((1ull << lane) - 1ull));
        // compaction target index <= source index and every source of this round is already in
        // registers, so the in-place write is safe
        if (keep) pool[pos] = v;
        out += __popcll(m);
    }
    pool_len = out;
}

struct SearchCounters {
    uint32_t evals, expansions, visited, flags;
    // wave-0 cycle accounting (s_memtime): controller pop/edge/visited, distance phase, admission
    uint64_t cyc_ctl, cyc_eval, cyc_ins;
};

// ---- distances of sh.nb_addr[0..n) -> sh.nb_ab / sh.nb_xx (all waves) ------------------------
// EVR rows are in flight per wave at a time (EVR * NJ 16-byte loads per lane).
template <int NJ, int EVR>
__device__ inline void eval_neighbours(const SegDev &seg, const QueryRegs<NJ> &q, SearchShared &sh, int n, bool cosine) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int base = wave * EVR; base < n; base += nwaves * EVR) {
        float4 row[EVR][NJ];
#pragma unroll
        for (int i = 0; i < EVR; i++) {
            if (base + i < n) {
                const float *r = seg.vectors + (size_t)sh.nb_addr[base + i] * seg.dp;
#pragma unroll
                for (int j = 0; j < NJ; j++) row[i][j] = load_row_chunk(r, seg.dp, j, lane);
            } else {
#pragma unroll
                for (int j = 0; j < NJ; j++) row[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // reduce width: next power of two >= the number of per-lane partial sums
        constexpr int NV_COS = 2 * EVR <= 2 ? 2 : (2 * EVR <= 4 ? 4 : 8);
        constexpr int NV_DOT = EVR <= 1 ? 1 : (EVR <= 2 ? 2 : 4);
        if (cosine) {
            float v[NV_COS];
#pragma unroll
            for (int i = 0; i < NV_COS; i++) v[i] = 0.f;
#pragma unroll
            for (int i = 0; i < EVR; i++) {
                float ab = 0.f, xx = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    ab = fma4(row[i][j], q.qv[j], ab);
                    xx = fma4(row[i][j], row[i][j], xx);
                }
                v[2 * i] = ab;
                v[2 * i + 1] = xx;
            }
            float r = QReduce<NV_COS>::run(v, lane);
            int which = QReduce<NV_COS>::query_of_lane(lane);
            if ((lane & QReduce<NV_COS>::group_mask()) == 0) {
                int i = which >> 1;
                if (i < EVR && base + i < n) {
                    if ((which & 1) == 0) sh.nb_ab[base + i] = r;
                    else sh.nb_xx[base + i] = r;
                }
            }
        } else {
            float v[NV_DOT];
#pragma unroll
            for (int i = 0; i < NV_DOT; i++) v[i] = 0.f;
#pragma unroll
            for (int i = 0; i < EVR; i++) {
                float ab = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; j++) ab = fma4(row[i][j], q.qv[j], ab);
                v[i] = ab;
            }
            float r = QReduce<NV_DOT>::run(v, lane);
            int which = QReduce<NV_DOT>::query_of_lane(lane);
            if ((lane & QReduce<NV_DOT>::group_mask()) == 0 && which < EVR && base + which < n) sh.nb_ab[base + which] = r;
        }
    }
}

// ---- edge record of `node` at `layer`: lane i gets word i (wave 0) ----------------------------
__device__ inline uint32_t load_edge_word(const GraphDev &g, uint32_t node, int layer, int lane, uint32_t &deg) {
    uint32_t w = 0;
    if (layer == 0) {
        w = g.l0[(size_t)node * NIDX_L0_STRIDE + lane];
    } else {
        uint32_t base = g.upper_base[node];
        if (base != 0xffffffffu && lane < NIDX_UP_STRIDE) w = g.upper[((size_t)base + (layer - 1)) * NIDX_UP_STRIDE + lane];
    }
    deg = lane_bcast_u32(w, 0);
    return w;
}

// ---- HnswSearcher::layer_search (search.rs:242-304) ------------------------------------------
// Entry points: sh.eps[0..sh.ctrl[2]).  Result: `res` (wave 0), best first.  All threads call.
template <int NJ, int EFL, int EVR>
__device__ inline void layer_search_block(const SegDev &seg, const GraphDev &g, int layer, int k,
                                          const QueryRegs<NJ> &q, SearchShared &sh, uint32_t *vis, uint32_t vis_log2,
                                          WaveTopK<EFL> &res, SearchCounters &st) {
    const int lane = threadIdx.x & 63;
    const bool ctl = (threadIdx.x >> 6) == 0;
    const bool cosine = seg.similarity == 1;
    const uint32_t vis_cap = 1u << vis_log2;
    int pool_len = 0;
    uint32_t vis_count = 0;

    vis_clear(vis, vis_cap);
    __syncthreads();
    int n_new = sh.ctrl[2];
    if (ctl) {
        res.init();
        if (lane < n_new) {
            uint32_t ep = sh.eps[lane];
            vis_insert(vis, vis_log2, ep);
            sh.nb_addr[lane] = ep;
        }
        // entry points beyond 64 (only the ef=100 build path) are handled in a second round below
    }
    // entry points are admitted unconditionally (search.rs:256-261)
    int ep_done = 0;
    const int n_eps = n_new;
    while (ep_done < n_eps) {
        int chunk = n_eps - ep_done < 64 ? n_eps - ep_done : 64;
        if (ctl && ep_done > 0 && lane < chunk) {
            uint32_t ep = sh.eps[ep_done + lane];
            vis_insert(vis, vis_log2, ep);
            sh.nb_addr[lane] = ep;
        }
        __syncthreads();
        eval_neighbours<NJ, EVR>(seg, q, sh, chunk, cosine);
        __syncthreads();
        if (ctl) {
            float s = lane < chunk ? score_from_sums(sh.nb_ab[lane], sh.nb_xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
            uint32_t addr = sh.nb_addr[lane];
            for (int j = 0; j < chunk; j++) {
                uint64_t nk = rank_key(lane_bcast_f32(s, j), lane_bcast_u32(addr, j));
                if (lane == 0) sh.pool[pool_len] = nk;
                pool_len++;
                res.insert(nk, 64 * EFL, lane);
            }
            st.evals += chunk;
            vis_count += chunk;
        }
        ep_done += chunk;
    }

    for (;;) {
        uint64_t t_a = clock64();
        if (ctl) {
            int cont = 0;
            n_new = 0;
            uint64_t ck = pool_pop(sh.pool, pool_len, lane);
            if (ck != NIDX_EMPTY_KEY) {
                float cs = rank_key_score(ck);
                float ws = res.worst_score();
                if (!(cs < ws)) {
                    cont = 1;
                    uint32_t deg;
                    uint32_t w = load_edge_word(g, rank_key_addr(ck), layer, lane, deg);
                    bool is_edge = lane >= 1 && lane <= (int)deg;
                    bool fresh = is_edge && vis_insert(vis, vis_log2, w);
                    unsigned long long m = __ballot(fresh);
                    int pos = __popcll(m & ((1ull << lane) - 1ull));
                    if (fresh) sh.nb_addr[pos] = w;
                    n_new = __popcll(m);
                    st.expansions++;
                    vis_count += n_new;
                    if (vis_count > vis_cap - vis_cap / 4) {  // table too full: give up exactly here
                        st.flags |= NIDX_FLAG_VISITED_OVERFLOW;
                        cont = 0;
                    }
                }
            }
            if (lane == 0) {
                sh.ctrl[0] = cont;
                sh.ctrl[1] = n_new;
            }
        }
        __syncthreads();
        uint64_t t_b = clock64();
        st.cyc_ctl += t_b - t_a;
        if (!sh.ctrl[0]) break;
        n_new = sh.ctrl[1];
        eval_neighbours<NJ, EVR>(seg, q, sh, n_new, cosine);
        __syncthreads();
        uint64_t t_c = clock64();
        st.cyc_eval += t_c - t_b;
        if (ctl && n_new > 0) {
            float s = lane < n_new ? score_from_sums(sh.nb_ab[lane], sh.nb_xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
            uint32_t addr = sh.nb_addr[lane];
            st.evals += n_new;
            // `if similarity > ws || len < k` replayed in edge order (search.rs:287-295).  Once the
            // set is full ws only grows, so lanes failing against the current ws can be skipped.
            unsigned long long todo = __ballot(lane < n_new);
            while (todo) {
                float ws = res.worst_score();
                if (res.len >= k) {
                    todo &= __ballot(lane < n_new && s > ws);
                    if (!todo) break;
                }
                int j = __ffsll((long long)todo) - 1;
                todo &= ~(1ull << j);
                float sj = lane_bcast_f32(s, j);
                if (sj > ws || res.len < k) {
                    uint64_t nk = rank_key(sj, lane_bcast_u32(addr, j));
                    if (pool_len == NIDX_POOL_CAP) {
                        if (res.len >= k) pool_prune(sh.pool, pool_len, ws, lane);
                        if (pool_len == NIDX_POOL_CAP) {  // more than CAP live ties: cannot stay exact
                            st.flags |= NIDX_FLAG_POOL_INEXACT;
                            pool_len--;
                        }
                    }
                    if (lane == 0) sh.pool[pool_len] = nk;
                    pool_len++;
                    res.insert(nk, k, lane);
                }
            }
        }
        // the controller's LDS reads of nb_* above complete before it rewrites them: same wave
        st.cyc_ins += clock64() - t_c;
    }
    st.visited = st.visited > vis_count ? st.visited : vis_count;
}

}  // namespace nidx
