// bm25_coop.hip — bm25_stream_kernel's algorithm with a WORKGROUP per work item (gfx950).
//
// Same contract, same results bit for bit as bm25_stream_kernel (bm25_stream.hip; tantivy's Bm25Weight::score per posting, BooleanQuery
// sums in clause order, TopDocs (score desc, DocAddress asc), Count — nidx_text/src/reader.rs:433-435, nidx_paragraph/src/reader.rs:244-348),
// for the plain case: a union query of <= 8 term clauses, k <= 64, no alive set / cursor / order key / match bitset.
//
// Why.  Round 6's counters (profiles/r06_bm25_breakdown.txt): 82 % of the stream kernel's vector instructions are the candidate path and
// the 64-key merges.  With k = 20 a wave that filters n postings against its own top-k offers ~k ln(n / k) candidates — ~90 for the 1 900
// postings of an item — so nearly every row builds keys and every item merges three times; per posting that cost falls only with the
// length of the stream ONE list sees, and a wave's stream cannot grow: its 32 Kibit filter bitmap overflows the involved list beyond
// ~1 900 postings of a balanced query, and one wave walking 8 000 postings alone is a four times longer launch.  Here the four waves of
// a workgroup share one item of four times the postings: ONE filter (128 Kibit A, 8 Kibit B: the same density), ONE sorted list (in LDS,
// merged under a workgroup lock by whichever wave has 64 candidates), one threshold every wave reads per group — the memory-level
// parallelism of four 1 900-posting items with the candidates, merges, slice search and phase overheads of one.
//
// Phases (see bm25_stream.hip for the algorithm): every clause's row groups are dealt to the waves round robin (wave w takes groups
// w, w + 4, ..); phase 1 marks, | barrier |, phase 2 streams the longest clause, | barrier |, phase 3 the others, | barrier |, phase 4
// resolves the involved postings — wave 0 alone when there are <= 64 of them, all waves through an LDS hash table clause by clause
// otherwise.  Involved postings go to per-wave lists (a wave's runs stay in clause order); a list that overflows raises a workgroup
// flag, every wave sees it at the next barrier, the doc range is halved and retried with de-duplicating serial insertions, exactly like
// the one-wave kernel — exact for any input.
#include "device_common.h"
#include "kernels.h"
#include "wave_bitonic.h"

namespace nidx {

namespace {

typedef const __attribute__((address_space(4))) uint32_t *bc_cu32_t;
template <typename T>
__device__ inline bc_cu32_t bc_const_words(const T *p) { return (bc_cu32_t)(uintptr_t)p; }
__device__ inline void bc_lds_order() { asm volatile("" ::: "memory"); }
__device__ inline uint32_t bc_rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

#define BC_A_WORDS 4096u    /* 128 Kibit */
#define BC_A_MASK 0x1ffffu
#define BC_B_WORDS 256u     /* 8 Kibit */
#define BC_B_MASK 0x1fffu
#define BC_CAP 192u         /* involved postings per wave and doc range */
#define BC_QN 3
#define BC_CAND 128u
#define BC_GROUPS 8
#define BC_T_SLOTS 1024u    /* phase-4 hash table (4 x BC_CAP = 768 documents at most) */

template <int J>
__device__ inline float bc_cmpx_f32(float v, unsigned long long take_max_mask) {
    const bool tm = __builtin_amdgcn_inverse_ballot_w64(take_max_mask);
    if constexpr (J >= 16) {
        uint32_t a0 = __float_as_uint(v), a1 = a0;
        if constexpr (J == 32) swap_pair32(a0, a1);
        else swap_pair16(a0, a1);
        const float x = __uint_as_float(a0), y = __uint_as_float(a1);
        return tm ? fmaxf(x, y) : fminf(x, y);
    } else {
        const float p = __uint_as_float(xor_partner_dpp<J>(__float_as_uint(v)));
        return tm ? fmaxf(v, p) : fminf(v, p);
    }
}
template <int K, int J>
__device__ inline float bc_sort_steps_f32(float v) {
    v = bc_cmpx_f32<J>(v, bs_sort_mask(K, J));
    if constexpr (J > 1) return bc_sort_steps_f32<K, J / 2>(v);
    else return v;
}
template <int K>
__device__ inline float bc_sort_stages_f32(float v) {
    if constexpr (K > 2) v = bc_sort_stages_f32<K / 2>(v);
    return bc_sort_steps_f32<K, K / 2>(v);
}

}  // namespace

__global__ __launch_bounds__(256, 5) void bm25_coop_kernel(Bm25Args a, const uint32_t *items, uint32_t n_items) {
    __shared__ float quot[BC_QN][256];
    __shared__ float tf_cache_s[256];
    __shared__ uint32_t bm_a[BC_A_WORDS];
    __shared__ uint32_t bm_b[BC_B_WORDS];
    __shared__ uint32_t list_doc_all[4][BC_CAP];
    __shared__ uint32_t list_score_all[4][BC_CAP];
    __shared__ uint64_t cand_all[4][BC_CAND];
    __shared__ uint64_t s_list[64];       // the item's sorted list, best first
    __shared__ uint64_t s_kth;            // its k-th key (NIDX_EMPTY_KEY while it holds fewer)
    __shared__ uint32_t s_lock, s_ovf, s_total;
    __shared__ uint32_t s_run_lo[4][8], s_run_hi[4][8], s_n_long[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *list_doc = list_doc_all[wave];
    uint32_t *list_score = list_score_all[wave];
    uint64_t *cand = cand_all[wave];
    auto clear_bitmaps = [&]() {   // all four waves
        for (uint32_t i = threadIdx.x; i < BC_A_WORDS / 4; i += 256) reinterpret_cast<uint4 *>(bm_a)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (threadIdx.x < BC_B_WORDS / 4) reinterpret_cast<uint4 *>(bm_b)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    };
    {
        const float c = a.tf_cache[threadIdx.x];
        float qv[BC_QN];
#pragma unroll
        for (int t = 0; t < BC_QN; t++) qv[t] = a.tf_cache[256 * (t + 1) + threadIdx.x];
        clear_bitmaps();
#pragma unroll
        for (int t = 0; t < BC_QN; t++) quot[t][threadIdx.x] = qv[t];
        tf_cache_s[threadIdx.x] = c;
        if (threadIdx.x < 64) s_list[threadIdx.x] = NIDX_EMPTY_KEY;
        if (threadIdx.x == 0) s_kth = NIDX_EMPTY_KEY, s_lock = 0u, s_ovf = 0u, s_total = 0u;
    }
    __syncthreads();
    // ---- the item record and the query's clause table (every wave: the values are workgroup-uniform) ----
    const uint32_t item = bc_const_words(items)[blockIdx.x];
    bc_cu32_t wrec = bc_const_words(a.work) + (size_t)item * 5u;
    const uint32_t slice = wrec[1], n_slices = wrec[2], clause_first = wrec[3];
    const int C = (int)wrec[4];   // <= 8
    const int k = (int)a.k;      // <= 64
    const uint32_t *const doc_ids = a.doc_ids;
    const uint32_t *const tfs = a.tfs;

    uint32_t len_l = 0, attr_l = 0, w_bits_l = 0;
    unsigned long long b_l = 0;
    if (lane < C) {
        const Bm25UClause uc = a.uclauses[clause_first + lane];
        b_l = ((unsigned long long)uc.b_hi << 32) | uc.b_lo;
        len_l = uc.len;
        attr_l = uc.attr;
        w_bits_l = __float_as_uint(uc.weight);
    }
    const uint32_t occur_l = attr_l & 0xff;
    const uint32_t must_m = (uint32_t)__ballot(lane < C && occur_l == 1), not_m = (uint32_t)__ballot(lane < C && occur_l == 2),
                   should_m = (uint32_t)__ballot(lane < C && occur_l == 0);
    uint32_t group_l = 0;
    int n_groups = 0;
#pragma unroll
    for (int g = 0; g < BC_GROUPS; g++) {
        const uint32_t gm = (uint32_t)__ballot(lane < C && occur_l == 3u + (uint32_t)g);
        if (gm) {
            if (lane == n_groups) group_l = gm;
            n_groups++;
        }
    }
    auto mask_ok = [&](uint32_t m) -> bool {
        const bool any_required = must_m != 0 || n_groups > 0;
        bool ok = (m & must_m) == must_m && (m & not_m) == 0 && (any_required || (m & should_m) != 0);
        for (int g = 0; g < n_groups; g++)
            if ((m & bc_rl(group_l, g)) == 0) ok = false;
        return ok;
    };
    const uint32_t single_ok_m = (uint32_t)__ballot(lane < C && mask_ok(1u << lane));

    const int g_log = C <= 1 ? 6 : C <= 2 ? 5 : C <= 4 ? 4 : 3;
    auto first_ge2 = [&](uint32_t bound0, uint32_t bound1, bool want0, bool want1, uint32_t left0_l, uint32_t right0_l, uint32_t &out0_l, uint32_t &out1_l) {
        const uint32_t G = 1u << g_log;
        const int grp = lane >> g_log;
        const uint32_t li = (uint32_t)lane & (G - 1u);
        const bool g_live = grp < C;
        const unsigned long long bg = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(b_l >> 32), grp) << 32) | (uint32_t)__shfl((int)(uint32_t)b_l, grp);
        const uint32_t *ids = doc_ids + bg;
        const uint32_t l0 = g_live ? (uint32_t)__shfl((int)left0_l, grp) : 0u, r0 = g_live ? (uint32_t)__shfl((int)right0_l, grp) : 0u;
        const uint32_t bound[2] = {bound0, bound1};
        uint32_t left[2] = {l0, l0}, right[2] = {want0 ? r0 : l0, want1 ? r0 : l0};
        const unsigned long long g_mask = (G == 64u ? ~0ull : ((1ull << G) - 1ull));
        for (;;) {
            bool wide[2];
            uint32_t step[2], probe[2], v[2];
#pragma unroll
            for (int t = 0; t < 2; t++) wide[t] = right[t] - left[t] > G;
            if (!__ballot(wide[0] || wide[1])) break;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                step[t] = (right[t] - left[t] + G - 1u) >> g_log;
                probe[t] = left[t] + step[t] * li;
                v[t] = ids[wide[t] && probe[t] < right[t] ? probe[t] : 0u];
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const bool ge = (wide[t] && probe[t] < right[t]) ? v[t] >= bound[t] : true;
                const unsigned long long m = (__ballot(ge) >> (grp << g_log)) & g_mask;
                const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : G;
                if (wide[t]) {
                    const uint32_t nl = first == 0u ? left[t] : left[t] + step[t] * (first - 1u);
                    const uint32_t nr = left[t] + step[t] * first;
                    left[t] = nl;
                    right[t] = nr < right[t] ? nr : right[t];
                }
            }
        }
        uint32_t v[2], res[2];
#pragma unroll
        for (int t = 0; t < 2; t++) v[t] = ids[left[t] + li < right[t] ? left[t] + li : 0u];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const bool ge = left[t] + li < right[t] ? v[t] >= bound[t] : true;
            const unsigned long long m = (__ballot(ge) >> (grp << g_log)) & g_mask;
            const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : G;
            res[t] = left[t] + first < right[t] ? left[t] + first : right[t];
        }
        out0_l = (uint32_t)__shfl((int)res[0], (lane << g_log) & 63);
        out1_l = (uint32_t)__shfl((int)res[1], (lane << g_log) & 63);
    };

    uint32_t lo_doc = 0, hi_doc = a.n_docs;
    uint32_t s_l = 0, item_e_l = len_l;
    if (n_slices > 1) {
        lo_doc = (uint32_t)((unsigned long long)a.n_docs * slice / n_slices);
        if (slice + 1 < n_slices) hi_doc = (uint32_t)((unsigned long long)a.n_docs * (slice + 1) / n_slices);
        uint32_t p0, p1;
        first_ge2(lo_doc, hi_doc, slice > 0, slice + 1 < n_slices, 0u, len_l, p0, p1);
        if (slice > 0) s_l = p0;
        if (slice + 1 < n_slices) item_e_l = p1;
    }

    // ---- the shared list ----
    uint64_t kth = NIDX_EMPTY_KEY;   // this wave's copy of the list's k-th key (never above the real one)
    uint32_t n_cand = 0, n_flush = 0;
    bool redo = false;
    auto lock = [&]() {
        if (lane == 0) {
            uint32_t expected = 0u;
            while (!__hip_atomic_compare_exchange_strong(&s_lock, &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                expected = 0u;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto unlock = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto refresh_kth = [&]() {
        const uint64_t v = __hip_atomic_load(&s_kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        if (u > kth) kth = u;
    };
    // the first min(n_cand, 64) buffered candidates of this wave into the workgroup's list
    auto flush64 = [&]() {
        bc_lds_order();
        uint64_t v = (uint32_t)lane < n_cand ? cand[lane] : NIDX_EMPTY_KEY;
        const uint32_t rest = n_cand > 64u ? n_cand - 64u : 0u;
        bc_lds_order();
        for (uint32_t o = 0; o < rest; o += 64u) {
            const uint64_t t = cand[64u + o + (o + (uint32_t)lane < rest ? (uint32_t)lane : 0u)];
            bc_lds_order();
            if (o + (uint32_t)lane < rest) cand[o + (uint32_t)lane] = t;
            bc_lds_order();
        }
        n_cand = rest;
        lock();
        uint64_t L = s_list[lane];
        uint64_t kk = lane_bcast_u64(L, k - 1);
        if (!redo) {
            if (__ballot(v > kk)) {   // (another wave's merges may have raised the bar above everything buffered here)
                L = bs_merge64(L, v);
                kk = lane_bcast_u64(L, k - 1);
                s_list[lane] = L;
                if (lane == 0) __hip_atomic_store(&s_kth, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // a doc range was retried: keys may be offered twice from here on — one by one, skipping what the list holds
            WaveTopK<1> top;
            top.l[0].key = L;
            top.len = 64;
            unsigned long long mm = __ballot(v > kk);
            while (mm) {
                const int src = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const uint64_t nk = lane_bcast_u64(v, src);
                if (!(nk > kk)) continue;
                if (__ballot(top.l[0].key == nk)) continue;
                kk = top.insert_kth(nk, k, lane);
            }
            s_list[lane] = top.l[0].key;
            if (lane == 0) __hip_atomic_store(&s_kth, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        unlock();
        if (kk > kth) kth = kk;
        n_flush++;
    };
    auto offer = [&](uint64_t ck, bool want) {   // the candidates of one row -> this wave's buffer
        const unsigned long long mm = __ballot(want && ck > kth);
        if (!mm) return;
        const uint32_t at = n_cand + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
        if ((mm >> lane) & 1ull) cand[at] = ck;
        n_cand += (uint32_t)__popcll(mm);   // (the caller drains the buffer below 64 entries after every row)
    };

    float bar = -INFINITY;
    bool bar_set = false;
    uint32_t postings = 0, total = 0;
    uint32_t cur_lo = lo_doc, cur_hi = hi_doc;
    uint32_t e_l = item_e_l;
    bool dirty = false;
    const uint32_t wstride = 1024u, woff = 256u * (uint32_t)wave;   // this wave's groups of a clause: s + woff, + 1024, ..
    for (;;) {
        const uint32_t n_l = lane < C ? e_l - s_l : 0u;
        const uint32_t act_m = (uint32_t)__ballot(n_l > 0u);
        if (act_m) {
            if (dirty) {
                __syncthreads();   // nobody still reads the table phase 4 laid over the bitmaps
                clear_bitmaps();
                dirty = false;
                __syncthreads();
            }
            uint32_t best = n_l;
            best = wave_reduce_u32(best, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            const int L = __ffsll((long long)__ballot(lane < C && n_l == best)) - 1;
            const bool probe = (act_m & (act_m - 1u)) != 0u;
            uint32_t dn[4], wn[4];
            {
                const unsigned long long base = ((unsigned long long)bc_rl((uint32_t)(b_l >> 32), L) << 32) | bc_rl((uint32_t)b_l, L);
                const uint32_t s = bc_rl(s_l, L) + woff;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    dn[r] = (doc_ids + base + s)[64u * r + (uint32_t)lane];   // (the arrays are padded behind the last list)
                    wn[r] = (tfs + base + s)[64u * r + (uint32_t)lane];
                }
            }
            uint32_t matched = 0, posted = 0, n_short = 0, n_long = 0;
            uint32_t run_lo_l = 0, run_hi_l = 0;   // lane c: clause c's run of this wave's involved list
            bool overflow = false;
            // ---- phase 1: mark ----
            if (probe) {
                dirty = true;
                for (uint32_t cm = act_m & ~(1u << L); cm; cm &= cm - 1u) {
                    const int c = __ffs((int)cm) - 1;
                    const unsigned long long base = ((unsigned long long)bc_rl((uint32_t)(b_l >> 32), c) << 32) | bc_rl((uint32_t)b_l, c);
                    const uint32_t s = bc_rl(s_l, c) + woff, e = bc_rl(e_l, c);
                    if (s >= e) continue;
                    const uint32_t *ip = doc_ids + base;
                    uint32_t dnx[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) dnx[r] = (ip + s)[64u * r + (uint32_t)lane];
                    for (uint32_t p = s; p < e; p += wstride) {
                        uint32_t d[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) d[r] = dnx[r];
                        if (p + wstride < e) {
#pragma unroll
                            for (int r = 0; r < 4; r++) dnx[r] = (ip + p)[wstride + 64u * r + (uint32_t)lane];
                        }
                        const uint32_t rem = e - p;
                        uint32_t h[4], old[4];
                        bool in[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            in[r] = (uint32_t)lane + 64u * r < rem;
                            h[r] = (d[r] ^ (d[r] >> 15)) & BC_A_MASK;
                            old[r] = 0u;
                            if (in[r]) old[r] = __hip_atomic_fetch_or(&bm_a[h[r] >> 5], 1u << (h[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        bool hit[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) hit[r] = in[r] && __builtin_amdgcn_ubfe(old[r], h[r], 1u) != 0u;
                        if (__ballot(hit[0] || hit[1] || hit[2] || hit[3])) {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (hit[r]) __hip_atomic_fetch_or(&bm_b[(h[r] & BC_B_MASK) >> 5], 1u << (h[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                __syncthreads();   // A (and what phase 1 put into B) is complete
            }
            // ---- phases 2 and 3: the longest clause, then the others in clause order ----
            uint32_t todo_m = probe ? (act_m & ~(1u << L)) : 0u;
            bool wg_overflow = false;
            for (int step = 0;; step++) {
                int c;
                if (step == 0) c = L;
                else {
                    if (!todo_m || wg_overflow) break;
                    c = __ffs((int)todo_m) - 1;
                    todo_m &= todo_m - 1u;
                }
                const bool is_long = step == 0;
                const unsigned long long base = ((unsigned long long)bc_rl((uint32_t)(b_l >> 32), c) << 32) | bc_rl((uint32_t)b_l, c);
                const uint32_t s0 = bc_rl(s_l, c), e = bc_rl(e_l, c);
                const uint32_t s = s0 + woff;
                const uint32_t attr = bc_rl(attr_l, c), w_bits = bc_rl(w_bits_l, c);
                const float wgt = __uint_as_float(w_bits);
                const uint32_t mode = (attr >> 8) & 0xffu;
                const bool row_ok = (single_ok_m >> c) & 1u;
                const uint32_t *ip = doc_ids + base;
                const uint32_t *wp = tfs + base;
                const uint32_t run_lo = n_short;
                posted += e - s0;
                if (!is_long && s < e) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        dn[r] = (ip + s)[64u * r + (uint32_t)lane];
                        wn[r] = (wp + s)[64u * r + (uint32_t)lane];
                    }
                }
                for (uint32_t p = s; p < e && !overflow; p += wstride) {
                    uint32_t d[4], w[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) d[r] = dn[r], w[r] = wn[r];
                    if (p + wstride < e) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            dn[r] = (ip + p)[wstride + 64u * r + (uint32_t)lane];
                            wn[r] = (wp + p)[wstride + 64u * r + (uint32_t)lane];
                        }
                    }
                    const uint32_t rem = e - p;
                    uint32_t h[4], bw[4] = {0u, 0u, 0u, 0u};
                    bool in[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        in[r] = (uint32_t)lane + 64u * r < rem;
                        h[r] = (d[r] ^ (d[r] >> 15)) & BC_A_MASK;
                        if (probe) bw[r] = is_long ? bm_a[h[r] >> 5] : bm_b[(h[r] & BC_B_MASK) >> 5];
                    }
                    float sc[4];
                    if (mode == 2u) {
#pragma unroll
                        for (int r = 0; r < 4; r++) sc[r] = wgt;
                    } else {
                        uint32_t t[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) t[r] = in[r] ? (w[r] & 0xffffffu) - 1u : 0u;
                        if (mode == 1u) {
#pragma unroll
                            for (int r = 0; r < 4; r++) sc[r] = wgt * quot[0][w[r] >> 24];
                        } else if (!__ballot((t[0] | t[1] | t[2] | t[3]) > (uint32_t)(BC_QN - 1))) {
#pragma unroll
                            for (int r = 0; r < 4; r++) sc[r] = wgt * (&quot[0][0])[(t[r] << 8) + (w[r] >> 24)];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const float tf = (float)(w[r] & 0xffffffu);
                                sc[r] = wgt * (tf / (tf + tf_cache_s[w[r] >> 24]));
                            }
                        }
                    }
                    if (w_bits >> 31) {
#pragma unroll
                        for (int r = 0; r < 4; r++) sc[r] = 0.f + sc[r];
                    }
                    bool inv[4] = {false, false, false, false};
                    uint32_t n_final = rem < 256u ? rem : 256u;
                    if (probe) {
#pragma unroll
                        for (int r = 0; r < 4; r++) inv[r] = in[r] && __builtin_amdgcn_ubfe(bw[r], h[r], 1u) != 0u;
                        if (__ballot(inv[0] || inv[1] || inv[2] || inv[3])) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const unsigned long long inv_m = __ballot(inv[r]);
                                if (!inv_m || overflow) continue;
                                const uint32_t n_new = (uint32_t)__popcll(inv_m);
                                if (n_short + n_long + n_new > BC_CAP) {
                                    overflow = true;
                                    continue;
                                }
                                const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(inv_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)inv_m, 0u));
                                const uint32_t at = is_long ? BC_CAP - 1u - (n_long + rank) : n_short + rank;
                                if (inv[r]) {
                                    if (is_long) __hip_atomic_fetch_or(&bm_b[(h[r] & BC_B_MASK) >> 5], 1u << (h[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    list_doc[at] = d[r];
                                    list_score[at] = __float_as_uint(sc[r]);
                                }
                                if (is_long) n_long += n_new;
                                else n_short += n_new;
                                n_final -= n_new;
                            }
                            if (overflow) break;
                        }
                    }
                    if (row_ok) {
                        matched += n_final;
                        if (!bar_set) {
                            float mx = -INFINITY;
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (in[r] && !inv[r]) mx = fmaxf(mx, sc[r]);
                            bar = lane_bcast_f32(bc_sort_stages_f32<64>(mx), 64 - k);
                            bar_set = true;
                        }
                        refresh_kth();
                        const float kf = fmaxf(bar, rank_key_score(kth));   // (NaN while the list is not full: fmaxf keeps the bar)
                        bool cnd[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) cnd[r] = in[r] && !inv[r] && !(sc[r] < kf);
                        if (__ballot(cnd[0] || cnd[1] || cnd[2] || cnd[3])) {
                            const uint32_t flushes = n_flush;
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (r > 0 && n_flush != flushes) {
                                    const float kf2 = fmaxf(bar, rank_key_score(kth));
                                    cnd[r] = cnd[r] && !(sc[r] < kf2);
                                }
                                if (__ballot(cnd[r])) offer(rank_key(sc[r], d[r]), cnd[r]);
                                while (n_cand >= 64u) flush64();
                            }
                        }
                    }
                }
                if (overflow && lane == 0) __hip_atomic_store(&s_ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!is_long && lane == c) run_lo_l = run_lo, run_hi_l = n_short;
                if (probe) {
                    // behind the longest clause: B is complete before phase 3 reads it; behind every clause: a wave that overflowed stops the others
                    __syncthreads();
                    wg_overflow = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&s_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0;
                }
            }
            if (wg_overflow) {
                // halve the doc range (>= 1 document stays) and retry it; the candidates offered so far are all final ones, they will be
                // offered again: everything buffered goes into the list first, insertions are serial and de-duplicating from here on
                while (n_cand) flush64();
                redo = true;
                __syncthreads();   // every wave has read the flag and flushed
                if (threadIdx.x == 0) s_ovf = 0u;
                const uint32_t span = cur_hi - cur_lo;
                cur_hi = cur_lo + (span > 1u ? span / 2u : 1u);
                {
                    uint32_t p0, p1;
                    first_ge2(cur_hi, 0u, true, false, s_l, e_l, p0, p1);
                    e_l = p0;
                }
                __syncthreads();   // the flag is clear before anybody can raise it again
                continue;
            }
            // ---- phase 4: the involved postings ----
            if (probe) {
                if (lane < 8) s_run_lo[wave][lane] = run_lo_l, s_run_hi[wave][lane] = run_hi_l;
                if (lane == 0) s_n_long[wave] = n_long;
                __syncthreads();
                uint32_t n_inv = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    n_inv += s_n_long[w];
                    for (int c = 0; c < C; c++)
                        if (c != L) n_inv += s_run_hi[w][c] - s_run_lo[w][c];
                }
                n_inv = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_inv);
                if (n_inv && n_inv <= 64u) {
                    if (wave == 0) {
                        // one per lane in (clause, wave) order: a document's sum is built in lane = clause order, exactly as before
                        uint32_t my_c = 0, off = 0;
                        const uint32_t *src_doc = list_doc_all[0], *src_sc = list_score_all[0];
                        uint32_t src = 0;
                        for (int c = 0; c < C; c++) {
                            const bool is_l = c == L;
                            for (int w = 0; w < 4; w++) {
                                const uint32_t lo = is_l ? 0u : s_run_lo[w][c], len = is_l ? s_n_long[w] : s_run_hi[w][c] - lo;
                                const uint32_t i = (uint32_t)lane - off;
                                if ((uint32_t)lane >= off && i < len) {
                                    src = is_l ? BC_CAP - 1u - i : lo + i;
                                    src_doc = list_doc_all[w], src_sc = list_score_all[w];
                                    my_c = (uint32_t)c;
                                }
                                off += len;
                            }
                        }
                        const bool live = (uint32_t)lane < n_inv;
                        const uint32_t my_doc = src_doc[live ? src : 0u];
                        const uint32_t my_sc = src_sc[live ? src : 0u];
                        const uint32_t my_adds = (__builtin_amdgcn_ds_bpermute((int)(my_c << 2), (int)attr_l) & 0xff) != 2 ? 1u : 0u;
                        unsigned long long *bucket = reinterpret_cast<unsigned long long *>(bm_a);
                        bucket[lane] = 0ull;
                        bc_lds_order();
                        const uint32_t hb = (my_doc * 2654435761u) >> 26;
                        if (live) __hip_atomic_fetch_or(&bucket[hb], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        bc_lds_order();
                        unsigned long long mates = live ? bucket[hb] : 0ull;
                        const uint32_t my_ca = my_c | (my_adds << 8);
                        float acc = 0.f;
                        uint32_t mask = 0;
                        bool owner = live;
                        while (__ballot(mates != 0ull)) {
                            const bool has = mates != 0ull;
                            const uint32_t j = has ? (uint32_t)__ffsll((long long)mates) - 1u : 0u;
                            mates &= mates - 1ull;
                            const uint32_t dj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)my_doc);
                            const uint32_t caj = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)my_ca);
                            const float sj = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)my_sc));
                            const bool same = has && dj == my_doc;
                            if (same && (caj >> 8)) acc += sj;
                            if (same) mask |= 1u << (caj & 0xffu);
                            if (same && j < (uint32_t)lane) owner = false;
                        }
                        const bool ok = owner && mask_ok(mask);
                        matched += (uint32_t)__popcll(__ballot(ok));
                        refresh_kth();
                        offer(ok ? rank_key(acc, my_doc) : NIDX_EMPTY_KEY, ok);
                        while (n_cand >= 64u) flush64();
                    }
                } else if (n_inv) {
                    // all four waves: a hash table over the bitmaps' space, clause by clause in clause order — inside one clause the documents
                    // are distinct (also across the waves), so every lane owns its document's slot for that round
                    uint32_t *t_doc = bm_a, *t_acc = bm_a + BC_T_SLOTS;
                    uint8_t *t_mask = reinterpret_cast<uint8_t *>(bm_a + 2 * BC_T_SLOTS);
                    for (uint32_t i = threadIdx.x; i < BC_T_SLOTS; i += 256) t_doc[i] = ~0u;
                    reinterpret_cast<uint32_t *>(t_mask)[threadIdx.x] = 0u;   // 256 words = 1 024 mask bytes
                    __syncthreads();
                    for (int c = 0; c < C; c++) {
                        const bool is_l = c == L;
                        const uint32_t lo = is_l ? 0u : bc_rl(run_lo_l, c), hi = is_l ? n_long : bc_rl(run_hi_l, c);
                        const bool adds = (bc_rl(attr_l, c) & 0xffu) != 2u;
                        for (uint32_t b0 = lo; b0 < hi; b0 += 64) {
                            const uint32_t i = b0 + (uint32_t)lane;
                            const bool live = i < hi;
                            const uint32_t st = live ? (is_l ? BC_CAP - 1u - i : i) : 0u;
                            const uint32_t doc = list_doc[st];
                            const float sc = __uint_as_float(list_score[st]);
                            uint32_t slot = (doc * 2654435761u) >> 22;
                            bool pending = live, fresh = false;
                            while (__ballot(pending)) {
                                if (pending) {
                                    uint32_t cur = __hip_atomic_load(&t_doc[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (cur == ~0u) {
                                        uint32_t expected = ~0u;
                                        if (__hip_atomic_compare_exchange_strong(&t_doc[slot], &expected, doc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                            fresh = true;
                                            cur = doc;
                                        } else cur = expected;
                                    }
                                    if (cur == doc) pending = false;
                                    else slot = (slot + 1u) & (BC_T_SLOTS - 1u);
                                }
                            }
                            if (live) {
                                float acc = fresh ? 0.f : __uint_as_float(t_acc[slot]);
                                if (adds) acc += sc;
                                t_acc[slot] = __float_as_uint(acc);
                                t_mask[slot] = (uint8_t)(t_mask[slot] | (1u << c));
                            }
                            bc_lds_order();
                        }
                        __syncthreads();   // clause c is in the table before clause c + 1 adds to its sums
                    }
                    for (uint32_t b0 = 0; b0 < 256u; b0 += 64) {
                        const uint32_t slot = 256u * (uint32_t)wave + b0 + (uint32_t)lane;
                        const uint32_t doc = t_doc[slot];
                        const bool ok = doc != ~0u && mask_ok((uint32_t)t_mask[slot]);
                        if (!__ballot(ok)) continue;
                        matched += (uint32_t)__popcll(__ballot(ok));
                        refresh_kth();
                        offer(ok ? rank_key(__uint_as_float(t_acc[slot]), doc) : NIDX_EMPTY_KEY, ok);
                        while (n_cand >= 64u) flush64();
                    }
                }
            }
            total += matched;
            if (wave == 0) postings += posted;
        }
        if (cur_hi >= hi_doc) break;
        cur_lo = cur_hi;
        cur_hi = hi_doc;
        s_l = e_l;
        e_l = item_e_l;
    }
    // ---- what is still buffered, the totals, the output ----
    refresh_kth();
    {
        const uint64_t v = (uint32_t)lane < n_cand ? cand[lane] : NIDX_EMPTY_KEY;   // (fewer than 64)
        if (__ballot(v > kth)) flush64();
    }
    if (lane == 0 && total) atomicAdd(&s_total, total);
    __syncthreads();
    if (wave == 0) {
        const uint64_t key = s_list[lane];
        const bool valid = key != NIDX_EMPTY_KEY && lane < k;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
        if (lane < k) a.out_key[(size_t)item * k + lane] = valid ? key : NIDX_EMPTY_KEY;
        if (lane == 0) {
            a.out_count[item] = cnt;
            a.out_total[item] = s_total;
            a.out_postings[item] = postings;
        }
    }
}

// one workgroup per item; the caller sends only what the kernel covers: k <= 64 and none of alive / after / order_key / match_bits
hipError_t launch_bm25_coop(const Bm25Args &a, const uint32_t *items, uint32_t n_items, hipStream_t s) {
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(bm25_coop_kernel, dim3(n_items), dim3(256), 0, s, a, items, n_items);
    return hipGetLastError();
}

}  // namespace nidx
