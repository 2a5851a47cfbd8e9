// hnsw_build.hip — HNSW construction on the device (gfx950).
//
// Replaces HnswBuilder::{insert, layer_insert, select_neighbours_heuristic}
// (nidx_vector/src/hnsw/build.rs:57-166) and the rayon `into_par_iter().for_each(insert)` driver
// (segment.rs:165-167,254-256).  The reference inserts concurrently under per-node RwLocks and its
// graph depends on thread interleaving (segment.rs:908); here concurrency is batch-synchronous:
//   phase 1  insert_search_kernel   one workgroup per new node: greedy descent (k=1) above the
//                                   node's top layer, ef=EF_CONSTRUCTION layer_search on its layers
//                                   (the same layer_search_block as the query kernel)
//   phase 2  select_link_kernel     one wave per (node, layer): select_neighbours_heuristic(M),
//                                   write the node's edge record, emit reverse-link requests
//   phase 3  reverse_link_kernel    requests sorted by (layer, target, source); one wave per target
//                                   appends them in order and prunes with the same heuristic when the
//                                   record exceeds M_max (to prune_m = 95 %)
// Nodes of one batch do not see each other (like two rayon workers racing), everything else is the
// reference's rule.  Edge weights are kept beside the edges (hnsw.edges, disk/v2.rs:46-49).
#include <stdlib.h>

#include <hipcub/hipcub.hpp>

#include "hnsw_device.h"
#include "hnsw_graph.h"

namespace nidx {

struct BuildArgs {
    SegDev seg;
    GraphDev g;
    float *l0_w;            // [n][64]
    float *upper_w;         // [n_upper][32]
    const uint8_t *levels;  // [n]
    uint32_t batch_start, batch_size;
    const uint32_t *slot_base;  // [batch_size] first (node, layer) slot of each batch node
    uint64_t *found;            // [n_slots][128] rank keys, best first
    uint32_t *found_len;        // [n_slots]
    uint32_t *slot_node;        // [n_slots]
    uint32_t *slot_layer;       // [n_slots]
    uint64_t *req_key;          // [n_slots*32]  layer:4 | target:30 | source:30   (~0 = unused)
    float *req_val;             // [n_slots*32]
    uint32_t vis_log2;
    uint32_t ef_upper;          // results kept per layer ABOVE the node's top layer (0 = 1: the greedy descent of build.rs / search.rs)
    uint32_t *flags;            // [1] OR of NIDX_FLAG_*
    unsigned long long *dbg;    // nullptr or 5 counters: appends, prunes, prune cycles, total cycles, targets
    // The build's work, counted like the search kernel's (a distance evaluation reads one row of 4 D bytes, an expansion
    // one edge record): NIDX_BUILD_STAT_LINES cache lines of 8 counters — [0] evaluations and [1] expansions of the construction
    // searches, [2] rows read by select_neighbours_heuristic on the new nodes' own lists, [3] rows read by the reverse-link prunes,
    // [4] reverse-link appends, [5] prunes.  A workgroup / wave adds its totals once, to the line blockIdx selects (one line would
    // serialise a few hundred thousand atomics per batch on one memory channel).
    unsigned long long *stats;
};
#define STAT_LINE(blk) ((size_t)((blk) & (NIDX_BUILD_STAT_LINES - 1)) * 16)

#define FOUND_STRIDE NIDX_BUILD_FOUND_STRIDE
#define REQ_STRIDE NIDX_BUILD_REQ_STRIDE

template <int NJ>
__global__ __launch_bounds__(256, 4) void insert_search_kernel(BuildArgs a) {   // (<= 128 VGPRs: four construction searches per CU, like the query kernel; it compiled to 132 = three)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SearchShared &sh = *reinterpret_cast<SearchShared *>(smem);
    uint32_t *vis = reinterpret_cast<uint32_t *>(smem + sizeof(SearchShared));
    const int lane = threadIdx.x & 63;
    const bool ctl = (nidx_tid() >> 6) == 0;
    const uint32_t bi = blockIdx.x;
    const uint32_t x = a.batch_start + bi;
    const bool cosine = a.seg.similarity == 1;
    const int level = (int)a.levels[x];

    QueryRegs<NJ> q;  // SearchVector::Stored(x)
    load_query<NJ>(q, a.seg.vectors + (size_t)x * a.seg.dp, a.seg.dp, lane, cosine);
    SearchCounters st = {0, 0, 0, 0, 0, 0, 0};
    WaveTopK<2> res;
    res.init();
    if (nidx_tid() == 0) {
        sh.eps[0] = a.g.ep_node;
        sh.ctrl[2] = 1;
    }
    __syncthreads();
    for (int layer = (int)a.g.ep_layer; layer >= 0; layer--) {
        const bool in_layer = layer <= level;
        const int k = in_layer ? NIDX_EF_CONSTRUCTION : (a.ef_upper ? (int)a.ef_upper : 1);
        layer_search_block<NJ, 2, 4>(a.seg, a.g, layer, k, q, sh, vis, a.vis_log2, res, st);
        if (ctl) {
            // next layer's entry points = every result (build.rs:146)
            uint64_t k0 = res.l[0].key, k1 = res.l[1].key;
            if (lane < res.len) sh.eps[lane] = rank_key_addr(k0);
            if (64 + lane < res.len) sh.eps[64 + lane] = rank_key_addr(k1);
            if (lane == 0) sh.ctrl[2] = res.len;
            if (in_layer) {
                const uint32_t slot = a.slot_base[bi] + (uint32_t)layer;
                uint64_t *f = a.found + (size_t)slot * FOUND_STRIDE;
                // the node itself can only be found when it is the entry point; a self link is useless
                bool self0 = lane < res.len && rank_key_addr(k0) == x;
                bool self1 = 64 + lane < res.len && rank_key_addr(k1) == x;
                unsigned long long m0 = __ballot(lane < res.len && !self0);
                unsigned long long m1 = __ballot(64 + lane < res.len && !self1);
                int p0 = __popcll(m0 & ((1ull << lane) - 1ull));
                int p1 = __popcll(m0) + __popcll(m1 & ((1ull << lane) - 1ull));
                if (lane < res.len && !self0) f[p0] = k0;
                if (64 + lane < res.len && !self1) f[p1] = k1;
                if (lane == 0) {
                    a.found_len[slot] = (uint32_t)(__popcll(m0) + __popcll(m1));
                    a.slot_node[slot] = x;
                    a.slot_layer[slot] = (uint32_t)layer;
                }
            }
        }
        __syncthreads();
    }
    if (ctl && lane == 0 && st.flags) atomicOr(a.flags, st.flags);
    if (ctl && lane == 0 && a.stats) {
        unsigned long long *sl = a.stats + STAT_LINE(blockIdx.x);
        atomicAdd(&sl[0], (unsigned long long)st.evals);
        atomicAdd(&sl[1], (unsigned long long)st.expansions);
    }
}

// Similarities of up to 4 candidate rows (in registers) against up to 4 stored rows: 16 dot products
// reduced with one transposed butterfly (WAVE64 order, bit-identical to a per-pair butterfly).
// out[c][y] valid for c < nc, y < ny.
template <int NJ>
__device__ inline void sims4x4(const SegDev &seg, const float4 (&cv)[4][NJ], const float (&c_norm2)[4], int nc,
                               const uint32_t (&ys)[4], int ny, bool cosine, int lane, float (&out)[4][4]) {
    float4 row[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < ny) {
            const float *r = seg.vectors + (size_t)ys[i] * seg.dp;
#pragma unroll
            for (int j = 0; j < NJ; j++) row[i][j] = load_row_chunk(r, seg.dp, j, lane);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; j++) row[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float v[16];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float ab = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) ab = fma4(cv[c][j], row[i][j], ab);
            v[c * 4 + i] = ab;
        }
    float r = QReduce<16>::run(v, lane);
    // value w = 4*c + i lives in the 4-lane group whose query_of_lane == w.  That group turns its sum into the
    // similarity (ONE f64 cosine per lane instead of sixteen), then the 16 results are broadcast.
    const int w_mine = QReduce<16>::query_of_lane(lane);
    const int c_mine = w_mine >> 2, i_mine = w_mine & 3;
    float mine = r;
    if (cosine) {
        const float cn = c_mine == 0 ? c_norm2[0] : (c_mine == 1 ? c_norm2[1] : (c_mine == 2 ? c_norm2[2] : c_norm2[3]));
        const uint32_t y = i_mine == 0 ? ys[0] : (i_mine == 1 ? ys[1] : (i_mine == 2 ? ys[2] : ys[3]));
        mine = (c_mine < nc && i_mine < ny) ? cosine_from_sums(r, cn, seg.norm2[y]) : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int w = c * 4 + i;
            const int src = (((w >> 3) & 1) << 5) | (((w >> 2) & 1) << 4) | (((w >> 1) & 1) << 3) | ((w & 1) << 2);
            const float sim = __shfl(mine, src, 64);
            out[c][i] = (c < nc && i < ny) ? sim : 0.f;
        }
}

// select_neighbours_heuristic (build.rs:57-95) for one wave.
//   cand[0..n): rank keys in the order the reference iterates them
//   out[0..) in LDS: selected rank keys.  Returns the count.
// Candidates are examined four at a time: all four are compared with the set kept so far (the kept
// rows are fetched once per group instead of once per candidate), then the group is resolved in
// order, each survivor also being compared with the survivors before it in the group — the same
// decisions, in the same order, as the one-by-one loop.
template <int NJ>
__device__ inline int select_neighbours_wave(const SegDev &seg, const uint64_t *cand, int n, int k, uint64_t *out,
                                             uint64_t *discard, bool cosine, int lane, uint32_t &rows_read) {
    int n_res = 0, n_dis = 0;
    for (int i0 = 0; i0 < n && n_res < k; i0 += 4) {
        const int g = n - i0 < 4 ? n - i0 : 4;
        uint64_t ck[4];
        uint32_t caddr[4];
        float cs[4], cn[4];
        float4 cv[4][NJ];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            ck[t] = t < g ? cand[i0 + t] : 0ull;
            caddr[t] = rank_key_addr(ck[t]);
            cs[t] = rank_key_score(ck[t]);
            cn[t] = (t < g && cosine) ? seg.norm2[caddr[t]] : 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++)
                cv[t][j] = t < g ? load_row_chunk(seg.vectors + (size_t)caddr[t] * seg.dp, seg.dp, j, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        rows_read += (uint32_t)g;
        bool fail[4] = {false, false, false, false};
        const int kept_before = n_res;
        for (int b = 0; b < kept_before; b += 4) {
            uint32_t ys[4];
            const int cnt = kept_before - b < 4 ? kept_before - b : 4;
#pragma unroll
            for (int t = 0; t < 4; t++) ys[t] = t < cnt ? rank_key_addr(out[b + t]) : 0u;
            float s[4][4];
            sims4x4<NJ>(seg, cv, cn, g, ys, cnt, cosine, lane, s);
            rows_read += (uint32_t)cnt;
            bool all_failed = true;
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (c < g && t < cnt && !(cs[c] > s[c][t])) fail[c] = true;  // needs sim(x,new) > sim(x,y) for all kept y
                if (c < g && !fail[c]) all_failed = false;
            }
            if (all_failed) break;
        }
        // similarities inside the group (candidate vs earlier candidate of the group)
        float sg[4][4];
        {
            uint32_t ys[4];
#pragma unroll
            for (int t = 0; t < 4; t++) ys[t] = caddr[t];
            bool need = false;
#pragma unroll
            for (int c = 1; c < 4; c++)
                if (c < g && !fail[c]) need = true;
            if (need) {
                sims4x4<NJ>(seg, cv, cn, g, ys, g, cosine, lane, sg);
                rows_read += (uint32_t)g;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int t = 0; t < 4; t++) sg[c][t] = 0.f;
            }
        }
        bool kept[4] = {false, false, false, false};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c >= g || n_res >= k) continue;  // `if results.len() == k { break }` before looking at the candidate
            bool check = !fail[c];
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (t < c && kept[t] && !(cs[c] > sg[c][t])) check = false;
            if (check) {
                kept[c] = true;
                if (lane == 0) out[n_res] = ck[c];
                n_res++;
            } else {
                if (lane == 0) discard[n_dis] = ck[c];
                n_dis++;
            }
        }
    }
    if (n_res < k && n_dis > 0) {
        // keepPrunedConnections: best discarded first (max-heap pop order), then re-sort everything
        WaveSortedList all;
        all.init();
        WaveSortedList dl;
        dl.init();
        for (int i = 0; i < n_dis; i++) dl.insert(discard[i], lane);
        int take = k - n_res < n_dis ? k - n_res : n_dis;
        for (int i = 0; i < n_res; i++) all.insert(out[i], lane);
        for (int i = 0; i < take; i++) all.insert(dl.at(i), lane);
        n_res += take;
        if (lane < n_res) out[lane] = all.key;
    }
    return n_res;
}

__device__ inline uint32_t *edge_record(const GraphDev &g, uint32_t node, int layer) {
    if (layer == 0) return g.l0 + (size_t)node * NIDX_L0_STRIDE;
    return g.upper + ((size_t)g.upper_base[node] + (layer - 1)) * NIDX_UP_STRIDE;
}
__device__ inline float *weight_record(const BuildArgs &a, uint32_t node, int layer) {
    if (layer == 0) return a.l0_w + (size_t)node * NIDX_L0_STRIDE;
    return a.upper_w + ((size_t)a.g.upper_base[node] + (layer - 1)) * NIDX_UP_STRIDE;
}

// phase 2: one wave per slot
template <int NJ, int MINW>
__global__ __launch_bounds__(256, MINW) void select_link_kernel(BuildArgs a, uint32_t n_slots) {
    __shared__ uint64_t s_out[4][64];
    __shared__ uint64_t s_dis[4][128];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 4 + wib;
    if (slot >= n_slots) return;
    const bool cosine = a.seg.similarity == 1;
    const uint32_t x = a.slot_node[slot];
    const int layer = (int)a.slot_layer[slot];
    const int n = (int)a.found_len[slot];
    const uint64_t *cand = a.found + (size_t)slot * FOUND_STRIDE;
    uint32_t rows_read = 0;
    int m = select_neighbours_wave<NJ>(a.seg, cand, n, NIDX_M, s_out[wib], s_dis[wib], cosine, lane, rows_read);
    if (lane == 0 && a.stats) atomicAdd(&a.stats[STAT_LINE(blockIdx.x) + 2], (unsigned long long)rows_read);
    // *layer.out[x] = neighbours (build.rs:108)
    uint32_t *rec = edge_record(a.g, x, layer);
    float *wrec = weight_record(a, x, layer);
    uint64_t key = lane < m ? s_out[wib][lane] : 0ull;
    if (lane == 0) rec[0] = (uint32_t)m;
    if (lane < m) {
        rec[1 + lane] = rank_key_addr(key);
        wrec[1 + lane] = rank_key_score(key);
    }
    // reverse-link requests (build.rs:111-118)
    if (lane < REQ_STRIDE) {
        uint64_t rk = ~0ull;
        float rv = 0.f;
        if (lane < m) {
            rk = ((uint64_t)layer << 60) | ((uint64_t)rank_key_addr(key) << 30) | (uint64_t)x;
            rv = rank_key_score(key);
        }
        a.req_key[(size_t)slot * REQ_STRIDE + lane] = rk;
        a.req_val[(size_t)slot * REQ_STRIDE + lane] = rv;
    }
}

// phase 3: one wave per run of requests with the same (layer, target)
template <int NJ, int MINW>
__global__ __launch_bounds__(256, MINW) void reverse_link_kernel(BuildArgs a, const uint64_t *keys, const float *vals,
                                                           uint32_t n_req) {
    __shared__ uint64_t s_cand[4][64];
    __shared__ uint64_t s_out[4][64];
    __shared__ uint64_t s_dis[4][64];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * 4 + wib;
    if (i0 >= n_req) return;
    const uint64_t k0 = keys[i0];
    if (k0 == ~0ull) return;
    if (i0 > 0 && (keys[i0 - 1] >> 30) == (k0 >> 30)) return;  // not the head of its run
    const bool cosine = a.seg.similarity == 1;
    const int layer = (int)(k0 >> 60);
    const uint32_t y = (uint32_t)((k0 >> 30) & 0x3fffffffu);
    const int mmax = layer == 0 ? NIDX_M_MAX0 : NIDX_M_MAX;
    const int pm = mmax * 95 / 100;  // params::prune_m
    uint32_t *rec = edge_record(a.g, y, layer);
    float *wrec = weight_record(a, y, layer);
    int deg = (int)rec[0];
    // lane j holds edge j as a rank key (score = stored weight)
    uint64_t e = lane < deg ? rank_key(wrec[1 + lane], rec[1 + lane]) : 0ull;
    unsigned long long n_prune = 0, n_app = 0, cy_prune = 0;
    uint32_t rows_read = 0;
    const unsigned long long t0 = clock64();
    for (uint32_t i = i0; i < n_req; i++) {
        uint64_t ki = keys[i];
        if (ki == ~0ull || (ki >> 30) != (k0 >> 30)) break;
        uint32_t x = (uint32_t)(ki & 0x3fffffffu);
        uint64_t nk = rank_key(vals[i], x);
        // The reference would push a second copy when y already links to x (possible only for the entry
        // point, which collects reverse links before its own insertion): a duplicate edge is dead weight.
        if (__ballot(lane < deg && rank_key_addr(e) == x)) continue;
        if (lane == deg) e = nk;  // other_edges.push((x, dist))
        deg++;
        n_app++;
        if (deg > mmax) {
            const unsigned long long tp = clock64();
            s_cand[wib][lane] = e;  // stored order
            int m = select_neighbours_wave<NJ>(a.seg, s_cand[wib], deg, pm, s_out[wib], s_dis[wib], cosine, lane, rows_read);
            e = lane < m ? s_out[wib][lane] : 0ull;
            deg = m;
            n_prune++;
            cy_prune += clock64() - tp;
        }
    }
    if (a.dbg && lane == 0) {
        atomicAdd(&a.dbg[0], n_app);
        atomicAdd(&a.dbg[1], n_prune);
        atomicAdd(&a.dbg[2], cy_prune);
        atomicAdd(&a.dbg[3], clock64() - t0);
        atomicAdd(&a.dbg[4], 1ull);
    }
    if (a.stats && lane == 0) {
        unsigned long long *sl = a.stats + STAT_LINE(blockIdx.x);
        if (rows_read) atomicAdd(&sl[3], (unsigned long long)rows_read);
        atomicAdd(&sl[4], n_app);
        if (n_prune) atomicAdd(&sl[5], n_prune);
    }
    if (lane == 0) rec[0] = (uint32_t)deg;
    if (lane < deg) {
        rec[1 + lane] = rank_key_addr(e);
        wrec[1 + lane] = rank_key_score(e);
    }
}

template <int NJ>
static hipError_t launch_batch(const BuildArgs &a, uint32_t n_slots, void *sort_tmp, size_t sort_tmp_bytes,
                               uint64_t *req_key_sorted, float *req_val_sorted, hipStream_t s) {
    size_t smem = sizeof(SearchShared) + ((size_t)4 << a.vis_log2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&insert_search_kernel<NJ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((insert_search_kernel<NJ>), dim3(a.batch_size), dim3(256), smem, s, a);
    // register budget of the two heuristic kernels (they are bound by the latency of the kept-row fetches: more waves per SIMD hide
    // more of it, fewer registers spill): NIDX_GPU_BUILD_MINW = 2 | 3 (default) | 4 waves per SIMD
    static const int minw = [] { const char *e = getenv("NIDX_GPU_BUILD_MINW"); const int v = e ? atoi(e) : 3; return v < 3 ? 2 : v > 3 ? 4 : 3; }();
    if (minw == 4) hipLaunchKernelGGL((select_link_kernel<NJ, 4>), dim3((n_slots + 3) / 4), dim3(256), 0, s, a, n_slots);
    else if (minw == 2) hipLaunchKernelGGL((select_link_kernel<NJ, 2>), dim3((n_slots + 3) / 4), dim3(256), 0, s, a, n_slots);
    else hipLaunchKernelGGL((select_link_kernel<NJ, 3>), dim3((n_slots + 3) / 4), dim3(256), 0, s, a, n_slots);
    const uint32_t n_req = n_slots * REQ_STRIDE;
    e = hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_tmp_bytes, a.req_key, req_key_sorted, a.req_val,
                                           req_val_sorted, (int)n_req, 0, 64, s);
    if (e != hipSuccess) return e;
    if (minw == 4) hipLaunchKernelGGL((reverse_link_kernel<NJ, 4>), dim3((n_req + 3) / 4), dim3(256), 0, s, a, req_key_sorted, req_val_sorted, n_req);
    else if (minw == 2) hipLaunchKernelGGL((reverse_link_kernel<NJ, 2>), dim3((n_req + 3) / 4), dim3(256), 0, s, a, req_key_sorted, req_val_sorted, n_req);
    else hipLaunchKernelGGL((reverse_link_kernel<NJ, 3>), dim3((n_req + 3) / 4), dim3(256), 0, s, a, req_key_sorted, req_val_sorted, n_req);
    return hipGetLastError();
}

__global__ void flag_clear_kernel(uint32_t *word, uint32_t bits) { atomicAnd(word, ~bits); }
hipError_t launch_flag_clear(uint32_t *word, uint32_t bits, hipStream_t s) {
    hipLaunchKernelGGL(flag_clear_kernel, dim3(1), dim3(1), 0, s, word, bits);
    return hipGetLastError();
}

hipError_t build_sort_tmp_bytes(uint32_t max_req, size_t *bytes) {
    *bytes = 0;
    return hipcub::DeviceRadixSort::SortPairs(nullptr, *bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                              (float *)nullptr, (float *)nullptr, (int)max_req, 0, 64, nullptr);
}

hipError_t launch_build_batch(const BuildBatch &b, hipStream_t s) {
    BuildArgs a;
    a.seg = b.seg;
    a.g = b.g;
    a.l0_w = b.l0_w;
    a.upper_w = b.upper_w;
    a.levels = b.levels;
    a.batch_start = b.batch_start;
    a.batch_size = b.batch_size;
    a.slot_base = b.slot_base;
    a.found = b.found;
    a.found_len = b.found_len;
    a.slot_node = b.slot_node;
    a.slot_layer = b.slot_layer;
    a.req_key = b.req_key;
    a.req_val = b.req_val;
    a.vis_log2 = b.vis_log2;
    a.ef_upper = b.ef_upper;
    a.flags = b.flags;
    a.dbg = b.dbg;
    a.stats = b.stats;
    int nj = (int)((a.seg.dp + 255u) / 256u);
#define NIDX_BUILD_CASE(N) \
    return launch_batch<N>(a, b.n_slots, b.sort_tmp, b.sort_tmp_bytes, b.req_key_sorted, b.req_val_sorted, s)
    if (nj <= 1) NIDX_BUILD_CASE(1);
    if (nj <= 2) NIDX_BUILD_CASE(2);
    if (nj <= 3) NIDX_BUILD_CASE(3);
    if (nj <= 4) NIDX_BUILD_CASE(4);
    if (nj <= 6) NIDX_BUILD_CASE(6);
    if (nj <= 8) NIDX_BUILD_CASE(8);
    if (nj <= 12) NIDX_BUILD_CASE(12);
    if (nj <= 16) NIDX_BUILD_CASE(16);   // D <= 4096
#undef NIDX_BUILD_CASE
    return hipErrorInvalidValue;
}

}  // namespace nidx
