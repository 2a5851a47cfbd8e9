// hnsw_search.hip — batched HNSW k-NN query kernel (gfx950): one workgroup per query.
//
// Replaces HnswSearcher::search (nidx_vector/src/hnsw/search.rs:306-383, non-RaBitQ branch):
//   greedy descent with k=1 through the upper layers, layer-0 search with ef = max(k, EF_SEARCH),
//   closest_up_nodes (filter / de-duplication / min_score walk), final stable sort by score.
// Bound: HBM (random 4*D-byte row gathers + 256-byte edge records).  Algorithmic bytes per query =
// evals * 4*D + expansions * 256 (both counted by the kernel).
#include "hnsw_device.h"

namespace nidx {

// Vector bytes equal?  (RepCounter keys on the stored bytes, search.rs:386-412.)  Wave-0 only.
template <int NJ>
__device__ inline bool rows_equal(const SegDev &seg, uint32_t a, uint32_t b, int lane) {
    const float *ra = seg.vectors + (size_t)a * seg.dp, *rb = seg.vectors + (size_t)b * seg.dp;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
        if (e < seg.dp) {
            uint4 x = *reinterpret_cast<const uint4 *>(ra + e);
            uint4 y = *reinterpret_cast<const uint4 *>(rb + e);
            eq = eq && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
        }
    }
    return __all(eq);
}

// One query's whole search in one workgroup.  Shared by the two launch forms below: one segment per launch (arguments in the
// kernel-argument segment) and every HNSW segment of an index in ONE launch (arguments of block b's segment read from a table in HBM).
template <int NJ, int EVR, int EFL>
__device__ __forceinline__ void hnsw_search_body(const HnswSearchArgs &a, const uint32_t qi, unsigned char *smem) {
    SearchShared &sh = *reinterpret_cast<SearchShared *>(smem);
    uint32_t *vis = reinterpret_cast<uint32_t *>(smem + sizeof(SearchShared));
    __shared__ uint32_t res_addr[64 * EFL];
    __shared__ float res_score[64 * EFL];
    __shared__ uint32_t res_para[64 * EFL];

    const int lane = threadIdx.x & 63;
    const bool ctl = (nidx_tid() >> 6) == 0;
    const bool cosine = a.seg.similarity == 1;
    const int k = (int)a.k;

    QueryRegs<NJ> q;
    load_query<NJ>(q, a.queries + (size_t)qi * a.seg.dp, a.seg.dp, lane, cosine);

    SearchCounters st = {0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_start = clock64();
    WaveTopK<EFL> res;  // ef = max(k, EF_SEARCH) <= 64*EFL
    res.init();

    // ---- upper layers: k = 1 (search.rs:318-324) ----
    if (nidx_tid() == 0) {
        sh.eps[0] = a.g.ep_node;
        sh.ctrl[2] = 1;
    }
    __syncthreads();
    const bool entry_mode = a.entry_vec != nullptr;
    const int efu = a.ef_upper ? (int)a.ef_upper : 1;
    for (int layer = entry_mode ? 0 : (int)a.g.ep_layer; layer >= 1; layer--) {
        layer_search_block<NJ, EFL, EVR>(a.seg, a.g, layer, efu, q, sh, vis, a.vis_log2, res, st);
        if (ctl) {
            uint64_t key = res.l[0].key;
            if (lane < res.len) sh.eps[lane] = rank_key_addr(key);
            if (lane == 0) sh.ctrl[2] = res.len;
        }
        __syncthreads();
    }
    // ---- layer 0 with ef = max(k, EF_SEARCH) (search.rs:333-349) ----
    const int efs = a.ef_search ? (int)a.ef_search : NIDX_EF_SEARCH;
    const int ef = k > efs ? k : efs;
    if (!entry_mode) layer_search_block<NJ, EFL, EVR>(a.seg, a.g, 0, ef, q, sh, vis, a.vis_log2, res, st);
    if (a.dump_vec) {
        // spill path: the walk below runs in hnsw_closest_spill_kernel, from these candidates
        if (ctl) {
#pragma unroll
            for (int i = 0; i < EFL; i++) {
                const uint64_t key = res.mine(i);
                const int e = 64 * i + lane;
                if (e < res.len) {
                    a.dump_vec[(size_t)qi * NIDX_DUMP_STRIDE + e] = rank_key_addr(key);
                    a.dump_score[(size_t)qi * NIDX_DUMP_STRIDE + e] = rank_key_score(key);
                }
            }
            if (lane == 0) {
                a.dump_count[qi] = (uint32_t)res.len;
                if (a.stats) a.stats[(size_t)qi * NIDX_STAT_STRIDE + NIDX_STAT_FLAGS] = st.flags;
                if (st.flags && a.flag_word) atomicOr(a.flag_word, st.flags);
            }
        }
        return;
    }

    // ---- closest_up_nodes (search.rs:188-240) ----
    // candidates = the ef neighbours; visited = exactly those; pop best, accept if it passes the
    // filter, stop at k accepted, otherwise expand its unvisited layer-0 neighbours that score
    // >= min_score.
    const uint32_t vis_cap = 1u << a.vis_log2;
    vis_clear(vis, vis_cap);
    __syncthreads();
    int pool_len = 0, n_res = 0;
    uint32_t vis_count = 0;
    uint64_t dropped_best = NIDX_EMPTY_KEY;
    // The edge record of the candidate that will most likely be popped next — the best one left in the pool: a new neighbour rarely
    // beats it, the pool holds the ef best nodes the layer search found — is requested while this expansion's rows are scored, so the
    // next expansion starts without the edge round trip in front of its rows (wave 0; pf_word = word `lane` of pf_node's record).
    uint32_t pf_node = 0xffffffffu, pf_word = 0;
    if (ctl && entry_mode) {
        // RaBitQ arm: the candidates are the re-ranked neighbours, scored with the raw vectors
        const int n_entry = (int)a.entry_count[qi];
        for (int i = lane; i < n_entry; i += 64) {
            const uint32_t addr = a.entry_vec[(size_t)qi * k + i];
            vis_insert(vis, a.vis_log2, addr);
            sh.pool[i] = rank_key(a.entry_score[(size_t)qi * k + i], addr);
        }
        pool_len = n_entry;
        vis_count = n_entry;
    } else if (ctl) {
#pragma unroll
        for (int i = 0; i < EFL; i++) {
            uint64_t key = res.mine(i);
            if (64 * i + lane < res.len) {
                vis_insert(vis, a.vis_log2, rank_key_addr(key));
                sh.pool[64 * i + lane] = key;
            }
        }
        pool_len = res.len;
        vis_count = res.len;
    }
    for (;;) {
        if (ctl) {
            int cont = 0, n_new = 0;
            uint64_t ck = pool_pop(sh.pool, pool_len, lane);
            if (ck != NIDX_EMPTY_KEY) {
                float cs = rank_key_score(ck);
                uint32_t c = rank_key_addr(ck);
                if (dropped_best > ck) st.flags |= NIDX_FLAG_POOL_INEXACT;
                if (!(cs < a.min_score)) {
                    bool accept = !(cs != cs);
                    const uint32_t p = a.seg.para_of_vec ? a.seg.para_of_vec[c] : c;
                    if (accept) {
                        if (a.seg.alive && !bit_test(a.seg.alive, p)) accept = false;
                        if (accept && a.filter && !bit_test(a.filter, p)) accept = false;
                    }
                    if (accept && !a.with_duplicates) {
                        // identical bytes => identical score bits: only then compare the rows
                        for (int i = 0; i < n_res && accept; i++) {
                            if (__builtin_bit_cast(uint32_t, res_score[i]) == __builtin_bit_cast(uint32_t, cs) &&
                                rows_equal<NJ>(a.seg, res_addr[i], c, lane))
                                accept = false;
                        }
                    }
                    if (accept && a.multi) {
                        // one hit per paragraph (checked after the duplicate test, like NodeFilter::passes)
                        for (int base = 0; base < n_res && accept; base += 64)
                            if (__ballot(base + lane < n_res && res_para[base + lane] == p)) accept = false;
                    }
                    if (accept) {
                        if (lane == 0) {
                            res_addr[n_res] = c;
                            res_score[n_res] = cs;
                            res_para[n_res] = p;
                        }
                        n_res++;
                    }
                    if (n_res < k) {
                        cont = 1;
                        uint32_t deg;
                        uint32_t w;
                        if (c == pf_node) {
                            w = pf_word;
                            deg = lane_bcast_u32(w, 0);
                            st.edge_hits++;
                        } else {
                            w = load_edge_word(a.g, c, 0, lane, deg);
                        }
                        if (a.closest_prefetch) {
                            const uint64_t nk = pool_peek(sh.pool, pool_len, lane);
                            pf_node = nk != NIDX_EMPTY_KEY ? rank_key_addr(nk) : 0xffffffffu;
                            if (pf_node != 0xffffffffu) pf_word = a.g.l0[(size_t)pf_node * NIDX_L0_STRIDE + lane];
                        }
                        bool is_edge = lane >= 1 && lane <= (int)deg;
                        bool fresh = is_edge && vis_insert(vis, a.vis_log2, w);
                        unsigned long long m = __ballot(fresh);
                        int pos = __popcll(m & ((1ull << lane) - 1ull));
                        if (fresh) sh.nb[0].addr[pos] = w;
                        n_new = __popcll(m);
                        st.expansions++;
                        vis_count += n_new;
                        if (vis_count > vis_cap - vis_cap / 4) {
                            st.flags |= NIDX_FLAG_VISITED_OVERFLOW;
                            cont = 0;
                        }
                    }
                }
            }
            if (lane == 0) {
                sh.ctrl[0] = cont;
                sh.ctrl[1] = n_new;
            }
        }
        __syncthreads();
        if (!sh.ctrl[0]) break;
        int n_new = sh.ctrl[1];
        eval_neighbours<NJ, EVR>(a.seg, q, sh.nb[0], n_new, cosine);
        __syncthreads();
        if (ctl && n_new > 0) {
            float s = lane < n_new ? score_from_sums(sh.nb[0].ab[lane], sh.nb[0].xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
            uint32_t addr = sh.nb[0].addr[lane];
            st.evals += n_new;
            unsigned long long todo = __ballot(lane < n_new && s >= a.min_score);
            while (todo) {
                int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                uint64_t nk = rank_key(lane_bcast_f32(s, j), lane_bcast_u32(addr, j));
                if (pool_len == NIDX_POOL_CAP) {
                    // evict the worst candidate; remember the best one ever evicted so that popping
                    // past it is reported instead of silently diverging from the reference
                    uint64_t worst = ~0ull;
                    for (int i = lane; i < pool_len; i += 64) worst = sh.pool[i] < worst ? sh.pool[i] : worst;
                    worst = wave_min_u64(worst);
                    if (nk < worst) {
                        dropped_best = nk > dropped_best ? nk : dropped_best;
                        continue;
                    }
                    dropped_best = worst > dropped_best ? worst : dropped_best;
                    int idx = 0x7fffffff;
                    for (int i = lane; i < pool_len; i += 64)
                        if (sh.pool[i] == worst && i < idx) idx = i;
                    idx = wave_min_i32(idx);
                    if (lane == 0) sh.pool[idx] = nk;
                } else {
                    if (lane == 0) sh.pool[pool_len] = nk;
                    pool_len++;
                }
            }
        }
    }

    // ---- filtered_result.sort_by(|a, b| b.1.total_cmp(&a.1)) — stable (search.rs:381) ----
    if (ctl) {
        st.visited = st.visited > vis_count ? st.visited : vis_count;
#pragma unroll
        for (int i = 0; i < EFL; i++) {
            const int e = 64 * i + lane;
            if (e < n_res) {
                const float s = res_score[e];
                const int32_t key = total_key(s);
                int rank = 0;
                for (int j = 0; j < n_res; j++) {
                    int32_t kj = total_key(res_score[j]);
                    rank += (kj > key || (kj == key && j < e)) ? 1 : 0;
                }
                a.out_vec[(size_t)qi * k + rank] = res_addr[e];
                a.out_score[(size_t)qi * k + rank] = s;
            } else if (e < k) {
                a.out_vec[(size_t)qi * k + e] = 0xffffffffu;
                a.out_score[(size_t)qi * k + e] = 0.f;
            }
        }
        if (lane == 0) {
            a.out_count[qi] = (uint32_t)n_res;
            if (st.flags && a.flag_word) atomicOr(a.flag_word, st.flags);
            if (a.stats && entry_mode) {
                // the RaBitQ kernel's counters stay; only overflow flags are added
                a.stats[(size_t)qi * NIDX_STAT_STRIDE + NIDX_STAT_FLAGS] |= st.flags;
            } else if (a.stats) {
                uint32_t *o = a.stats + (size_t)qi * NIDX_STAT_STRIDE;
                o[NIDX_STAT_EVALS] = st.evals;
                o[NIDX_STAT_EXPANSIONS] = st.expansions;
                o[NIDX_STAT_VISITED] = st.visited;
                o[NIDX_STAT_FLAGS] = st.flags;
                o[NIDX_STAT_CYC_CTL] = (uint32_t)st.cyc_ctl;
                o[NIDX_STAT_EDGE_HITS] = st.edge_hits;
                o[NIDX_STAT_CYC_INS] = (uint32_t)st.cyc_ins;
                o[NIDX_STAT_CYC_TOTAL] = (uint32_t)(clock64() - t_start);
            }
        }
    }
}

template <int NJ, int EVR, int MINW, int EFL>
__global__ __launch_bounds__(256, MINW) void hnsw_search_kernel(HnswSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    hnsw_search_body<NJ, EVR, EFL>(a, blockIdx.x, smem);
}

// Searcher::_search's loop over the segments (nidx_vector/src/searcher.rs:270-287) as ONE grid: block b walks query b % nq of
// segment b / nq.  The reference's indexes ARE many segments (merges stop at 200 k records, nidx/src/settings.rs:258-278: 10 M vectors
// = 50 segments); a launch per segment leaves the device idle behind the longest walk of each of them 50 times per batch.  Blocks
// of one segment are neighbours, so its upper layers and hub rows stay in the L2 of the XCDs that walk it.
template <int NJ, int EVR, int MINW, int EFL>
__global__ __launch_bounds__(256, MINW) void hnsw_search_segments_kernel(const HnswSearchArgs *__restrict__ table, uint32_t nq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t s = blockIdx.x / nq;
    const HnswSearchArgs a = table[s];   // uniform address, read-only: scalar loads
    hnsw_search_body<NJ, EVR, EFL>(a, blockIdx.x - s * nq, smem);
}

template <int NJ, int EVR, int MINW, int EFL>
static hipError_t launch_v(const HnswSearchArgs &a, int waves, hipStream_t s) {
    if (a.seg_table) {
        size_t smem = sizeof(SearchShared) + ((size_t)4 << a.vis_log2);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_search_segments_kernel<NJ, EVR, MINW, EFL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((hnsw_search_segments_kernel<NJ, EVR, MINW, EFL>), dim3(a.n_queries * a.n_table), dim3(64 * waves), smem, s,
                           a.seg_table, a.n_queries);
        return hipGetLastError();
    }
    size_t smem = sizeof(SearchShared) + ((size_t)4 << a.vis_log2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_search_kernel<NJ, EVR, MINW, EFL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((hnsw_search_kernel<NJ, EVR, MINW, EFL>), dim3(a.n_queries), dim3(64 * waves), smem, s, a);
    return hipGetLastError();
}

static uint32_t layer0_ef(const HnswSearchArgs &a) {
    const uint32_t efs = a.ef_search ? a.ef_search : NIDX_EF_SEARCH;
    return a.k > efs ? a.k : efs;
}

template <int NJ>
static hipError_t launch_nj(const HnswSearchArgs &a, int waves, hipStream_t s) {
    const uint32_t ef = layer0_ef(a);
    // large result pages (ef > 64) are the rare path: one shape each
    if (ef > 256) return launch_v<NJ, 2, 2, 8>(a, waves, s);
    if (ef > 128) return launch_v<NJ, 2, 2, 4>(a, waves, s);
    if (ef > 64) return launch_v<NJ, 2, 2, 2>(a, waves, s);
    // rows in flight per wave / register budget: tuned on MI355X (profiles/r01_tune_hnsw.txt, r02_tune_hnsw.txt).
    // min_waves 5 / 6: <= 96 / 80 VGPRs, two rows in flight per wave; with vis_log2 <= 12 a CU then holds 5 / 6 walks.
    if (a.min_waves >= 6) return launch_v<NJ, 2, 6, 1>(a, waves, s);
    if (a.min_waves == 5) return launch_v<NJ, 2, 5, 1>(a, waves, s);
    if (a.eval_rows == 3) return a.min_waves >= 4 ? launch_v<NJ, 3, 4, 1>(a, waves, s) : launch_v<NJ, 3, 2, 1>(a, waves, s);
    if (a.eval_rows == 2) return a.min_waves >= 4 ? launch_v<NJ, 2, 4, 1>(a, waves, s) : launch_v<NJ, 2, 2, 1>(a, waves, s);
    return a.min_waves >= 4 ? launch_v<NJ, 4, 4, 1>(a, waves, s) : launch_v<NJ, 4, 2, 1>(a, waves, s);
}

template <int NJ>
static hipError_t launch_wide(const HnswSearchArgs &a, int waves, hipStream_t s) {
    const uint32_t ef = layer0_ef(a);
    if (ef > 256) return launch_v<NJ, 2, 1, 8>(a, waves, s);
    if (ef > 128) return launch_v<NJ, 2, 1, 4>(a, waves, s);
    if (ef > 64) return launch_v<NJ, 2, 1, 2>(a, waves, s);
    return launch_v<NJ, 2, 1, 1>(a, waves, s);
}

hipError_t launch_hnsw_search(const HnswSearchArgs &a, int waves_per_query, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    if (a.k == 0 || a.k > NIDX_K_MAX || a.ef_search > NIDX_K_MAX || a.ef_upper > 64) return hipErrorInvalidValue;
    int nj = (int)((a.seg.dp + 255u) / 256u);
    if (waves_per_query < 1) waves_per_query = 1;
    if (waves_per_query > 4) waves_per_query = 4;
    if (nj <= 1) return launch_nj<1>(a, waves_per_query, s);
    if (nj <= 2) return launch_nj<2>(a, waves_per_query, s);
    if (nj <= 3) return launch_nj<3>(a, waves_per_query, s);
    if (nj <= 4) return launch_nj<4>(a, waves_per_query, s);
    if (nj <= 6) return launch_wide<6>(a, waves_per_query, s);
    if (nj <= 8) return launch_wide<8>(a, waves_per_query, s);
    if (nj <= 12) return launch_wide<12>(a, waves_per_query, s);
    if (nj <= 16) return launch_wide<16>(a, waves_per_query, s);   // D <= 4096
    return hipErrorInvalidValue;
}

}  // namespace nidx
