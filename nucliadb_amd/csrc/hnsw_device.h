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

// The neighbours of one expansion: their addresses and, after the distance phase, the two sums of each.
struct NbBuf {
    uint32_t addr[64];
    float ab[64];   // <x, q>
    float xx[64];   // |x|^2
};
struct SearchShared {
    uint64_t pool[NIDX_POOL_CAP];  // unexpanded candidates (rank keys), unsorted
    NbBuf nb[2];                   // two expansions: the one being admitted and the one being evaluated (layer_search_block)
    uint32_t eps[256];             // entry points for the next layer search
    int ctrl[12];                  // [0] continue, [1] n_new (single-buffer loops), [2] n_eps, [3] next row to evaluate,
                                   // [4] redo, [5] the speculated expansion exists, [6..7] n_new of nb[0], nb[1]
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

// Virtual thread id: the wave roles rotate with the workgroup index so that the controller waves (virtual wave 0) of the
// workgroups sharing a CU do not all sit on the same SIMD.  Lanes keep their position.
__device__ inline uint32_t nidx_tid() {
    const uint32_t nw = blockDim.x >> 6;   // 1, 2, 3 or 4 waves per workgroup
    const uint32_t t = threadIdx.x + ((blockIdx.x % nw) << 6);
    return t >= blockDim.x ? t - blockDim.x : t;
}

// ---- visited set ------------------------------------------------------------------------------
__device__ inline void vis_clear(uint32_t *vis, uint32_t cap) {
    for (uint32_t i = nidx_tid(); i < cap; i += blockDim.x) vis[i] = NIDX_VIS_EMPTY;
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

// read-only membership test (the table is not being written while this runs)
__device__ inline bool vis_contains(const uint32_t *vis, uint32_t log2cap, uint32_t v) {
    const uint32_t mask = (1u << log2cap) - 1u;
    uint32_t h = (v * 2654435761u) >> (32 - log2cap);
    for (;;) {
        const uint32_t cur = vis[h];
        if (cur == NIDX_VIS_EMPTY) return false;
        if (cur == v) return true;
        h = (h + 1) & mask;
    }
}

// ---- candidate pool: unsorted array in LDS, arg-max pop --------------------------------------
__device__ inline uint64_t wave_max_u64(uint64_t v) { return wave_extreme_u64<true>(v); }
__device__ inline uint64_t wave_min_u64(uint64_t v) { return wave_extreme_u64<false>(v); }
// The best key without removing it (EMPTY if none).  Wave-0 only.
__device__ inline uint64_t pool_peek(const uint64_t *pool, int pool_len, int lane) {
    uint64_t best = NIDX_EMPTY_KEY;
    for (int i = lane; i < pool_len; i += 64) {
        uint64_t v = pool[i];
        best = v > best ? v : best;
    }
    return wave_max_u64(best);
}
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

// ---- the result set of a layer search + which of its entries are still unexpanded --------------------------------------------
// HnswSearcher::layer_search keeps two heaps: `ms_neighbours` (the k best so far) and `candidates` (everything ever admitted,
// popped best first until the best one is worse than the worst result).  Every candidate was admitted into the result set too,
// and one that has since been evicted scores below the worst result for good — except one that left while TYING with it
// (`cs < ws` does not stop on equal scores).  So the live candidates are exactly the unexpanded entries of the result set plus
// those tied evictions: the set below is the sorted result list with one "unexpanded" bit per rank (wave-uniform masks that
// move with the insertions as scalar shifts), and the LDS pool only holds the tied evictions.  Pop = the lowest set bit — no
// LDS scan, no wave reduction on the controller's critical path.
template <int NL>
struct CandSet {
    uint64_t unexp[NL];
    __device__ inline void init() {
#pragma unroll
        for (int i = 0; i < NL; i++) unexp[i] = 0;
    }
    // Inserts an unexpanded entry, keeping the `cap` best.  out_key / out_unexp: the entry that left the set (EMPTY: none).
    __device__ inline void insert(WaveTopK<NL> &res, uint64_t nk, int cap, int lane, uint64_t &out_key, bool &out_unexp) {
        uint64_t d = nk, dflag = 1;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            if (d != NIDX_EMPTY_KEY) {
                const int pos = __popcll(__ballot(res.l[i].key > d));  // 0..64: entries of this list ranking before d
                if (pos < 64) {
                    const uint64_t last = lane_bcast_u64(res.l[i].key, 63);
                    const uint64_t lastflag = unexp[i] >> 63;
                    const uint64_t up = wave_shr1_u64(res.l[i].key);
                    if (lane > pos) res.l[i].key = up;
                    if (lane == pos) res.l[i].key = d;
                    const uint64_t low = unexp[i] & ((1ull << pos) - 1ull);
                    const uint64_t high = pos < 63 ? ((unexp[i] >> pos) << (pos + 1)) : 0ull;
                    unexp[i] = low | (dflag << pos) | high;
                    d = last;
                    dflag = lastflag;
                }
            }
        }
        res.len++;
        out_key = NIDX_EMPTY_KEY;
        out_unexp = false;
        if (res.len > cap) {
            if (cap >= 64 * NL) {  // the list itself is the bound: what fell off its end left the set
                out_key = d;
                out_unexp = dflag != 0;
            }
#pragma unroll
            for (int i = 0; i < NL; i++)
                if ((cap >> 6) == i) {
                    out_key = lane_bcast_u64(res.l[i].key, cap & 63);
                    out_unexp = ((unexp[i] >> (cap & 63)) & 1ull) != 0;
                    if (lane == (cap & 63)) res.l[i].key = NIDX_EMPTY_KEY;
                    unexp[i] &= ~(1ull << (cap & 63));
                }
            res.len = cap;
        }
    }
    // best unexpanded entry (EMPTY if none); pop marks it expanded
    __device__ inline uint64_t peek(const WaveTopK<NL> &res) const {
#pragma unroll
        for (int i = 0; i < NL; i++)
            if (unexp[i]) return lane_bcast_u64(res.l[i].key, __builtin_ctzll(unexp[i]));
        return NIDX_EMPTY_KEY;
    }
    // the two best unexpanded entries other than `skip` (EMPTY where there is none); nothing changes
    __device__ inline void peek2_except(const WaveTopK<NL> &res, uint64_t skip, uint64_t &a, uint64_t &b) const {
        a = b = NIDX_EMPTY_KEY;
        int found = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            uint64_t m = unexp[i];
            while (m && found < 2) {
                const uint64_t key = lane_bcast_u64(res.l[i].key, __builtin_ctzll(m));
                m &= m - 1ull;
                if (key == skip) continue;
                if (found == 0) a = key;
                else b = key;
                found++;
            }
        }
    }
    __device__ inline uint64_t pop(const WaveTopK<NL> &res) {
#pragma unroll
        for (int i = 0; i < NL; i++)
            if (unexp[i]) {
                const int r = __builtin_ctzll(unexp[i]);
                unexp[i] &= unexp[i] - 1ull;
                return lane_bcast_u64(res.l[i].key, r);
            }
        return NIDX_EMPTY_KEY;
    }
};

struct SearchCounters {
    uint32_t evals, expansions, visited, flags, edge_hits;
    // wave-0 cycle accounting (s_memtime): controller pop/edge/visited, distance phase, admission
    uint64_t cyc_ctl, cyc_eval, cyc_ins;
};

// ---- distances of nb.addr[base .. base + EVR) -> nb.ab / nb.xx (one wave) --------------------------------------------
// EVR rows are in flight at a time (EVR * NJ 16-byte loads per lane).
template <int NJ, int EVR>
__device__ inline void eval_row_group(const SegDev &seg, const QueryRegs<NJ> &q, NbBuf &nb, int base, int n, bool cosine, int lane) {
    float4 row[EVR][NJ];
#pragma unroll
    for (int i = 0; i < EVR; i++) {
        if (base + i < n) {
            const float *r = seg.vectors + (size_t)nb.addr[base + i] * seg.dp;
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
                if ((which & 1) == 0) nb.ab[base + i] = r;
                else nb.xx[base + i] = r;
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
        if ((lane & QReduce<NV_DOT>::group_mask()) == 0 && which < EVR && base + which < n) nb.ab[base + which] = r;
    }
}

// all waves, rows dealt out statically (wave w takes groups w, w + nwaves, ..)
template <int NJ, int EVR>
__device__ inline void eval_neighbours(const SegDev &seg, const QueryRegs<NJ> &q, NbBuf &nb, int n, bool cosine) {
    const int lane = threadIdx.x & 63;
    const int wave = nidx_tid() >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int base = wave * EVR; base < n; base += nwaves * EVR) eval_row_group<NJ, EVR>(seg, q, nb, base, n, cosine, lane);
}

// any subset of the waves, at any time: every caller claims the next EVR rows from the shared counter until none are left
// (which wave scores a row does not change the score: same loads, same fma chain, same butterfly)
template <int NJ, int EVR>
__device__ inline void eval_neighbours_dynamic(const SegDev &seg, const QueryRegs<NJ> &q, NbBuf &nb, int n, bool cosine, int *next_row) {
    const int lane = threadIdx.x & 63;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(next_row, EVR);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        eval_row_group<NJ, EVR>(seg, q, nb, base, n, cosine, lane);
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
    const bool ctl = (nidx_tid() >> 6) == 0;
    const bool cosine = seg.similarity == 1;
    const uint32_t vis_cap = 1u << vis_log2;
    int pool_len = 0;
    uint32_t vis_count = 0;

    vis_clear(vis, vis_cap);
    __syncthreads();
    int n_new = sh.ctrl[2];
    // more entry points than k (a descent kept wider than this layer's k): every entry point is admitted and each later push
    // pops ONE result (search.rs:256-261,289-292), so the set keeps the size the entry points gave it
    if (n_new > k) k = n_new < 64 * EFL ? n_new : 64 * EFL;
    CandSet<EFL> cand;
    cand.init();
    if (ctl) {
        res.init();
        if (lane < n_new) {
            uint32_t ep = sh.eps[lane];
            vis_insert(vis, vis_log2, ep);
            sh.nb[0].addr[lane] = ep;
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
            sh.nb[0].addr[lane] = ep;
        }
        __syncthreads();
        eval_neighbours<NJ, EVR>(seg, q, sh.nb[0], chunk, cosine);
        __syncthreads();
        if (ctl) {
            float s = lane < chunk ? score_from_sums(sh.nb[0].ab[lane], sh.nb[0].xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
            uint32_t addr = sh.nb[0].addr[lane];
            for (int j = 0; j < chunk; j++) {
                uint64_t ek;
                bool eu;
                cand.insert(res, rank_key(lane_bcast_f32(s, j), lane_bcast_u32(addr, j)), 64 * EFL, lane, ek, eu);
            }
            st.evals += chunk;
            vis_count += chunk;
        }
        ep_done += chunk;
    }

    // ---- the expansion loop, software-pipelined ---------------------------------------------------------------------------
    // The reference's loop is: pop the best candidate c, stop if it is worse than the worst result, score c's unvisited
    // neighbours, admit them in edge order.  One expansion is two dependent memory round trips (edge record, then rows) plus
    // the controller's serial admission — and with one query per workgroup nothing else hides them.  Here the NEXT candidate is
    // determined before the current expansion is admitted: it is max(best of the pool, best admissible new neighbour) — the best
    // new neighbour that beats the current worst result is always admitted, and nothing admitted can rank above it — so its
    // edge record is fetched, its unvisited neighbours are listed (read-only test) and the other waves start on their rows while
    // wave 0 runs the admission of the current expansion.  Nothing of the speculated expansion is committed (visited marks,
    // counters) before the real pop confirms it; a mismatch (possible only with exactly tied scores or a pool overflow) throws
    // it away and expands the popped candidate the plain way.
    // Edge records fetched ahead (wave 0, layer 0): while an expansion's rows are in flight, the records of the two best unexpanded
    // entries that were NOT chosen are loaded too — one of them is usually the next candidate but one, and its expansion then
    // starts without the edge round trip in front of its rows.  pw*: the record (word `lane`), tg*: whose it is.
    uint32_t pw0 = 0, pw1 = 0, tg0 = 0xffffffffu, tg1 = 0xffffffffu;
    uint32_t edge_hits = 0;
    auto prepare = [&](uint64_t ck, NbBuf &dst) -> int {   // wave 0: the neighbours of ck that are not visited -> dst.addr
        uint32_t deg;
        const uint32_t node = rank_key_addr(ck);
        uint32_t w;
        if (node == tg0) {
            w = pw0;
            deg = lane_bcast_u32(w, 0);
            edge_hits++;
        } else if (node == tg1) {
            w = pw1;
            deg = lane_bcast_u32(w, 0);
            edge_hits++;
        } else {
            w = load_edge_word(g, node, layer, lane, deg);
        }
        const bool fresh = lane >= 1 && lane <= (int)deg && !vis_contains(vis, vis_log2, w);
        const unsigned long long m = __ballot(fresh);
        if (fresh) dst.addr[__popcll(m & ((1ull << lane) - 1ull))] = w;
        return __popcll(m);
    };
    int p = 0;
    uint64_t cur = NIDX_EMPTY_KEY;
    if (ctl) {
        const uint64_t t_a = clock64();
        int cont = 0, n0 = 0;
        cur = cand.pop(res);
        if (cur != NIDX_EMPTY_KEY && !(rank_key_score(cur) < res.worst_score())) {
            cont = 1;
            n0 = prepare(cur, sh.nb[0]);
        }
        if (lane == 0) {
            sh.ctrl[0] = cont;
            sh.ctrl[6] = n0;
            sh.ctrl[3] = 0;
        }
        st.cyc_ctl += clock64() - t_a;
    }
    lds_barrier();
    if (sh.ctrl[0]) {
        eval_neighbours_dynamic<NJ, EVR>(seg, q, sh.nb[0], sh.ctrl[6], cosine, &sh.ctrl[3]);
        lds_barrier();
        for (;;) {
            NbBuf &nb_cur = sh.nb[p], &nb_next = sh.nb[p ^ 1];
            const int n_cur = sh.ctrl[6 + p];
            // ---- wave 0: commit the expansion whose sums are in nb_cur, pick the next candidate, list its neighbours ----
            float s = 0.f;
            uint32_t addr = 0;
            unsigned long long todo = 0;
            uint64_t nxt = NIDX_EMPTY_KEY;
            bool overflow = false;
            const uint64_t t_a = clock64();
            if (ctl) {
                const bool mine = lane < n_cur;
                addr = nb_cur.addr[lane];
                bool ins = mine && vis_insert(vis, vis_log2, addr);
                // an edge record that names a node twice (the reference can write one): the first occurrence counts
                unsigned long long failed = __ballot(mine && !ins);
                while (failed) {
                    const int f = __ffsll((long long)failed) - 1;
                    failed &= failed - 1;
                    const unsigned long long same = __ballot(mine && addr == lane_bcast_u32(addr, f));
                    if ((same >> lane) & 1ull) ins = lane == __ffsll((long long)same) - 1;
                }
                todo = __ballot(ins);
                const int n_valid = __popcll(todo);
                st.expansions++;
                st.evals += n_valid;
                vis_count += n_valid;
                overflow = vis_count > vis_cap - vis_cap / 4;  // table too full: give up exactly here
                s = ins ? score_from_sums(nb_cur.ab[lane], nb_cur.xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
                int n_next = 0;
                if (!overflow) {
                    const float ws = res.worst_score();
                    const bool adm = ins && (s > ws || res.len < k);
                    const uint64_t bn = wave_max_u64(adm ? rank_key(s, addr) : NIDX_EMPTY_KEY);
                    const uint64_t pm = cand.peek(res);
                    nxt = pm > bn ? pm : bn;
                    if (nxt == NIDX_EMPTY_KEY && pool_len > 0) nxt = pool_peek(sh.pool, pool_len, lane);  // tied evictions (rare)
                    if (nxt != NIDX_EMPTY_KEY) n_next = prepare(nxt, nb_next);
                    if (layer == 0) {
                        // the candidates after `nxt` as far as they are known now: the best unexpanded entries of the list
                        uint64_t ka, kb;
                        cand.peek2_except(res, nxt, ka, kb);
                        uint32_t wa = ka != NIDX_EMPTY_KEY ? rank_key_addr(ka) : 0xffffffffu;
                        uint32_t wb = kb != NIDX_EMPTY_KEY ? rank_key_addr(kb) : 0xffffffffu;
                        if (wa == tg1 || wb == tg0) {   // a record already held stays where it is
                            const uint32_t t = wa;
                            wa = wb;
                            wb = t;
                        }
                        if (wa != tg0) {
                            tg0 = wa;
                            if (wa != 0xffffffffu) pw0 = g.l0[(size_t)wa * NIDX_L0_STRIDE + lane];
                        }
                        if (wb != tg1) {
                            tg1 = wb;
                            if (wb != 0xffffffffu) pw1 = g.l0[(size_t)wb * NIDX_L0_STRIDE + lane];
                        }
                    }
                }
                if (lane == 0) {
                    sh.ctrl[5] = nxt != NIDX_EMPTY_KEY;
                    sh.ctrl[6 + (p ^ 1)] = n_next;
                    sh.ctrl[3] = 0;
                }
            }
            lds_barrier();
            const uint64_t t_b = clock64();
            st.cyc_ctl += t_b - t_a;
            const bool has_next = sh.ctrl[5] != 0;
            const int n_next = sh.ctrl[6 + (p ^ 1)];
            if (!ctl && has_next) eval_neighbours_dynamic<NJ, EVR>(seg, q, nb_next, n_next, cosine, &sh.ctrl[3]);
            if (ctl) {
                // `if similarity > ws || len < k` replayed in edge order (search.rs:287-295).  Once the
                // set is full ws only grows, so lanes failing against the current ws can be skipped.
                while (todo) {
                    float ws = res.worst_score();
                    if (res.len >= k) {
                        todo &= __ballot(s > ws);
                        if (!todo) break;
                    }
                    int j = __ffsll((long long)todo) - 1;
                    todo &= ~(1ull << j);
                    float sj = lane_bcast_f32(s, j);
                    if (sj > ws || res.len < k) {
                        uint64_t ek;
                        bool eu;
                        cand.insert(res, rank_key(sj, lane_bcast_u32(addr, j)), k, lane, ek, eu);
                        // an unexpanded entry evicted while tying with the new worst result is still a live candidate
                        if (eu && ek != NIDX_EMPTY_KEY && !(rank_key_score(ek) < res.worst_score())) {
                            if (pool_len == NIDX_POOL_CAP) {
                                pool_prune(sh.pool, pool_len, res.worst_score(), lane);
                                if (pool_len == NIDX_POOL_CAP) {  // more than CAP live ties: cannot stay exact
                                    st.flags |= NIDX_FLAG_POOL_INEXACT;
                                    pool_len--;
                                }
                            }
                            if (lane == 0) sh.pool[pool_len] = ek;
                            pool_len++;
                        }
                    }
                }
                int cont = 0, redo = 0;
                if (overflow) {
                    st.flags |= NIDX_FLAG_VISITED_OVERFLOW;
                } else {
                    cur = cand.pop(res);
                    if (cur == NIDX_EMPTY_KEY && pool_len > 0) cur = pool_pop(sh.pool, pool_len, lane);
                    cont = cur != NIDX_EMPTY_KEY && !(rank_key_score(cur) < res.worst_score());
                    redo = cont && cur != nxt;
                }
                if (lane == 0) {
                    sh.ctrl[0] = cont;
                    sh.ctrl[4] = redo;
                }
                const uint64_t t_c = clock64();
                st.cyc_ins += t_c - t_b;
                if (has_next) eval_neighbours_dynamic<NJ, EVR>(seg, q, nb_next, n_next, cosine, &sh.ctrl[3]);
                st.cyc_eval += clock64() - t_c;
            }
            lds_barrier();
            if (!sh.ctrl[0]) break;
            if (sh.ctrl[4]) {
                // the pop did not return the speculated candidate: expand the real one (nothing of the other was committed)
                if (ctl) {
                    const int n1 = prepare(cur, nb_next);
                    if (lane == 0) {
                        sh.ctrl[6 + (p ^ 1)] = n1;
                        sh.ctrl[3] = 0;
                    }
                }
                lds_barrier();
                eval_neighbours_dynamic<NJ, EVR>(seg, q, nb_next, sh.ctrl[6 + (p ^ 1)], cosine, &sh.ctrl[3]);
                lds_barrier();
            }
            p ^= 1;
        }
    }
    st.edge_hits += edge_hits;
    st.visited = st.visited > vis_count ? st.visited : vis_count;
}

}  // namespace nidx
