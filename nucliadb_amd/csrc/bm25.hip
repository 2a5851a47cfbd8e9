// bm25.hip — term-at-a-time BM25 scoring + TopDocs for a batch of boolean term queries (gfx950).
//
// Replaces what tantivy does under TextReaderService::do_search (nidx_text/src/reader.rs:433-435)
// and ParagraphReaderService's Searcher::do_search (nidx_paragraph/src/reader.rs:244-348):
//   Bm25Weight::score(fieldnorm_id, tf) = weight * tf / (tf + K1*(1 - B + B*fieldnorm/avg)),
//   BooleanQuery of Should / Must / MustNot term clauses (clause scores summed in clause order),
//   TopDocs::with_limit(k) ordered (score desc, DocAddress asc), Count, the search-after score tweak
//   (reader.rs:350-390), deletions as an alive bitset (nidx_tantivy/src/index_reader.rs:39-74).
//
// One WAVE per work item = (query, doc-id slice of the segment): the host cuts every query into slices of roughly
// equal posting count so that a query with a 400 k-posting term does not become the tail of the launch; a slice's
// cursors are found with a wave-wide 64-ary search.  Inside a work item the postings of the query's clauses are
// consumed in lockstep doc-id windows [lo, hi) of <= 512 postings, shared out in proportion to what is left of every
// clause's list; all of a window's (doc -> partial score) pairs live in an LDS hash table, clauses are applied one
// after the other — so each doc's f32 sum is built in clause order, exactly like the oracle's term-at-a-time loop —
// and the finished window is folded into the wave's top-k list.  Postings are read once (doc ids and tfs are
// separate arrays).  bm25_merge_kernel then merges the slices of a query.
// Bound: HBM; algorithmic bytes per posting scored = 9 (u32 doc + u32 tf + u8 fieldnorm id).
#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "wave_bitonic.h"

namespace nidx {

#define BM25_EMPTY 0xffffffffu

// =====================================================================================================
// One WAVE per work item, four items per workgroup (they share nothing but the 1 KiB tf cache and never meet at a barrier
// after the start).  Clause state lives in lane registers (lane c = clause c).  A window is R ROWS of 64 postings; a row
// belongs to ONE clause, so everything a posting needs from its clause (list base, attributes, weight, hit bit) is
// wave-uniform — scalar registers, no cross-lane shuffles.  Rows are shared out in proportion to what is left of every
// clause's list; the window ends at the smallest doc id some clause could not fit (or at the slice's upper doc bound).
// Per window:
//   1. ONE round trip for the R rows' doc ids and the per-clause "first doc not loaded";
//   2. ONE round trip for tf + fieldnorm of the postings inside the window;
//   3. every posting finds or claims its document's slot in the wave's LDS hash table (all CAS issued, then the rare collisions);
//   4. scores are added with LDS float atomics and the clause's hit bit is ORed in, row by row: one wave's LDS operations
//      execute in order and rows are in clause order, so a document's f32 sum is built in clause order like the oracle's
//      term-at-a-time loop, without a read-modify-write round trip per clause;
//   5. the lane that claimed a document folds it: the boolean structure of the query is a test on the hit mask
//      (all Must bits, no MustNot bit, one bit of every required Should group), then alive / facets / order key / cursor,
//      and the rank key goes to the wave's top-k list if it beats the k-th.
// LDS: 128 R slots x 12 B per wave (R = 4: 6 KiB; 25 KiB per workgroup => 6 workgroups = 24 waves per CU).
// =====================================================================================================
#define BW_MAX_GROUPS 8

// wave reductions on the swap + DPP levels of device_common.h (no ds_bpermute round trips)
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
    return wave_reduce_u64((uint64_t)v, [](uint64_t a, uint64_t b) { return a + b; });
}
__device__ inline uint32_t wave_min_u32(uint32_t v) {
    return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
    return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ inline uint32_t rl_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ inline float rl_f32(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <typename M>
struct BwSlot {   // one document of the current window
    float acc;    // clause-ordered f32 sum of the clause scores
    M mask;       // bit c: clause c has a posting for the document
};

template <int KL, int R, typename M>
__global__ __launch_bounds__(256) void bm25_rows_kernel(Bm25Args a, const uint32_t *items, uint32_t n_work) {
    constexpr int T = 128 * R;              // slots per wave: the window's <= 64 R postings at load factor <= 0.5
    constexpr int TSHIFT = 32 - (R == 4 ? 9 : R == 8 ? 10 : 8);
    static_assert(R == 2 || R == 4 || R == 8, "rows per window");
    __shared__ float tf_cache[256];
    __shared__ uint32_t t_key_all[4][T];
    __shared__ BwSlot<M> t_val_all[4][T];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *t_key = t_key_all[wave];
    BwSlot<M> *t_val = t_val_all[wave];
    tf_cache[threadIdx.x] = a.tf_cache[threadIdx.x];
    for (int i = lane; i < T; i += 64) {
        t_key[i] = BM25_EMPTY;
        t_val[i].acc = 0.f;
        t_val[i].mask = 0;
    }
    __syncthreads();   // the only workgroup barrier: the tf cache
    const uint32_t slot_in_grid = blockIdx.x * 4u + (uint32_t)wave;
    if (slot_in_grid >= n_work) return;
    const uint32_t item = items[slot_in_grid];
    const Bm25Work work = a.work[item];
    const uint32_t q = work.query;
    const uint64_t c0 = a.clause_offsets[q], c1 = a.clause_offsets[q + 1];
    const int C = (int)(c1 - c0);
    const int k = (int)a.k;

    // ---- lane c holds clause c ----
    uint32_t attr_l = 0;  // occur | mode << 8 | aux << 16
    float weight_l = 0.f;
    unsigned long long cur_l = 0, end_l = 0;
    if (lane < C) {
        const Bm25ClauseDev cd = a.clauses[c0 + lane];
        const bool aux = (cd.term & BM25_AUX_TERM) != 0;
        const uint32_t ti = cd.term & ~BM25_AUX_TERM;
        attr_l = (uint32_t)cd.occur | ((uint32_t)cd.mode << 8) | (aux ? 1u << 16 : 0u);
        weight_l = cd.weight;
        cur_l = aux ? a.aux_offsets[2 * ti] : a.term_offsets[ti];
        end_l = aux ? a.aux_offsets[2 * ti + 1] : a.term_offsets[ti + 1];
    }
    const uint32_t occur_l = attr_l & 0xff;
    // the query's boolean structure as masks over the clause bits (BooleanQuery: every Must, no MustNot, at least one clause
    // of every required Should group; plain Shoulds are required only when nothing else is)
    const M must_m = (M)__ballot(lane < C && occur_l == 1);
    const M not_m = (M)__ballot(lane < C && occur_l == 2);
    const M should_m = (M)__ballot(lane < C && occur_l == 0);
    M group_m[BW_MAX_GROUPS];
    int n_groups = 0;
#pragma unroll
    for (int g = 0; g < BW_MAX_GROUPS; g++) {
        group_m[g] = (M)__ballot(lane < C && occur_l == 3u + (uint32_t)g);
        if (group_m[g]) n_groups = g + 1;
    }
    const bool any_required = must_m != 0 || n_groups > 0;
    const uint32_t *ids_l = (attr_l >> 16) ? a.aux_doc_ids : a.doc_ids;   // this lane's clause's doc-id array
    // doc range of this slice [lo_doc, hi_doc); the wave finds, clause by clause, the first posting >= lo_doc (64-ary search)
    uint32_t hi_doc = 0xffffffffu;
    if (work.n_slices > 1) {
        const uint32_t lo_doc = (uint32_t)((unsigned long long)a.n_docs * work.slice / work.n_slices);
        if (work.slice + 1 < work.n_slices) hi_doc = (uint32_t)((unsigned long long)a.n_docs * (work.slice + 1) / work.n_slices);
        if (work.slice > 0) {
            for (int c = 0; c < C; c++) {
                const bool aux = (rl_u32(attr_l, c) >> 16) != 0;
                const uint32_t *ids = aux ? a.aux_doc_ids : a.doc_ids;
                unsigned long long left = lane_bcast_u64(cur_l, c), right = lane_bcast_u64(end_l, c);  // first index in [left, right] whose doc >= lo_doc
                while (right - left > 64) {
                    const unsigned long long step = (right - left + 63) / 64;
                    const unsigned long long probe = left + step * (unsigned long long)lane;
                    const bool ge = probe < right ? ids[probe] >= lo_doc : true;
                    const unsigned long long m = __ballot(ge);
                    const int first = m ? __ffsll((long long)m) - 1 : 64;
                    const unsigned long long nl = first == 0 ? left : left + step * (unsigned long long)(first - 1);
                    const unsigned long long nr = left + step * (unsigned long long)first;
                    left = nl;
                    right = nr < right ? nr : right;
                }
                const unsigned long long probe = left + (unsigned long long)lane;
                const bool ge = probe < right ? ids[probe] >= lo_doc : true;
                const unsigned long long m = __ballot(ge);
                const int first = m ? __ffsll((long long)m) - 1 : 64;
                const unsigned long long res = left + (unsigned long long)first < right ? left + (unsigned long long)first : right;
                if (lane == c) cur_l = res;
            }
        }
    }
    // first doc of every clause inside the slice (0xffffffff: none)
    uint32_t cdoc_l = 0xffffffffu;
    if (lane < C && cur_l < end_l) cdoc_l = ids_l[cur_l];
    unsigned long long cy_load = 0, cy_apply = 0, cy_fold = 0, n_win = 0, postings = 0, total = 0;
    const unsigned long long cy_t0 = clock64();

    WaveTopK<KL> top;  // k <= 64*KL
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    // search-after cursor (reader.rs:379-390)
    const bool has_after = a.after != nullptr && a.after[q].has_after != 0;
    const int32_t after_key = has_after ? total_key(a.after[q].score) : 0;
    const int after_tie = has_after ? a.after[q].tie_break : 0;
    const uint64_t after_addr = has_after ? a.after[q].docaddr : 0;
    const int mslot = a.match_slot ? a.match_slot[q] : -1;
    uint32_t *mbits = mslot >= 0 ? a.match_bits + (size_t)mslot * a.match_words : nullptr;

    // One window.  FAST (at most R live clauses): whole rows per clause, everything per-clause is wave-uniform.  Otherwise the
    // PACKED form: the 64 R slots are shared out posting by posting (every live clause gets at least one — clauses tied on the
    // smallest doc id must all be inside the window), a lane's slots may belong to different clauses, their attributes travel
    // by cross-lane reads, and the apply phase walks the clauses one by one to keep the clause order of the sums.
    auto window = [&](auto fast_tag, unsigned long long act_m, int n_act) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool active = (act_m >> lane) & 1ull;
        const float left_f = active ? (float)(end_l - cur_l) : 0.f;
        const float tot = wave_butterfly_sum(left_f);   // only used for the shares: any rounding is fine
        const uint32_t units = FAST ? (uint32_t)R : 64u * R;       // rows, or posting slots
        uint32_t share_l = active ? 1u + (uint32_t)((float)(units - (uint32_t)n_act) * (left_f / tot)) : 0u;
        uint32_t end_u = share_l;  // inclusive scan over the clause lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(end_u, off, 64);
            if (lane >= off) end_u += v;
        }
        if (rl_u32(end_u, 63) > units) {   // float rounding pushed the shares past the window: one unit each
            share_l = active ? 1u : 0u;
            end_u = share_l;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(end_u, off, 64);
                if (lane >= off) end_u += v;
            }
        }
        const uint32_t start_u = end_u - share_l;
        const uint32_t n_units = rl_u32(end_u, 63);
        const unsigned long long cy_a = clock64();
        // ---- round trip 1: the window's doc ids + every live clause's first doc beyond its share ----
        const unsigned long long next_l = cur_l + (FAST ? 64ull : 1ull) * share_l;
        uint32_t p_doc[R], p_at[R];
        unsigned long long p_idx[R];
        float p_w[R];
        int p_c[R];          // the clause of slot (m, lane); FAST: wave-uniform per row
        bool p_in[R];        // the posting exists (inside the clause's list)
#pragma unroll
        for (int m = 0; m < R; m++) {
            if constexpr (FAST) {
                const int rc = __popcll(__ballot(lane < C && end_u <= (uint32_t)m));   // clauses that end before row m
                const int c = rc < C ? rc : 0;
                p_c[m] = c;
                const unsigned long long cur_c = lane_bcast_u64(cur_l, c), end_c = lane_bcast_u64(end_l, c);
                p_at[m] = rl_u32(attr_l, c);
                p_w[m] = rl_f32(weight_l, c);
                p_idx[m] = cur_c + 64ull * ((uint32_t)m - rl_u32(start_u, c)) + (unsigned long long)lane;
                p_in[m] = (uint32_t)m < n_units && p_idx[m] < end_c;
            } else {
                const uint32_t g = (uint32_t)lane + 64u * m;
                int c = 0;
                for (int j = 1; j < C; j++) c += g >= rl_u32(start_u, j) ? 1 : 0;
                p_c[m] = c;
                const unsigned long long cur_c = shfl_u64(cur_l, c), end_c = shfl_u64(end_l, c);
                p_at[m] = __shfl(attr_l, c, 64);
                p_w[m] = __shfl(weight_l, c, 64);
                p_idx[m] = cur_c + (g - __shfl(start_u, c, 64));
                p_in[m] = g < n_units && p_idx[m] < end_c;
            }
            // unconditional loads (posting 0 of the main list stands in for a slot that does not exist): a load under a branch
            // is waited for at the end of that branch, which would turn the window's one round trip into R of them
            p_doc[m] = ((p_at[m] >> 16) && p_in[m] ? a.aux_doc_ids : a.doc_ids)[p_in[m] ? p_idx[m] : 0ull];
        }
        const bool more_l = active && next_l < end_l;
        uint32_t hi_c = (more_l ? ids_l : a.doc_ids)[more_l ? next_l : 0ull];
#pragma unroll
        for (int m = 0; m < R; m++) p_doc[m] = p_in[m] ? p_doc[m] : 0xffffffffu;
        hi_c = more_l ? hi_c : 0xffffffffu;
        const uint32_t hi_w = wave_min_u32(hi_c);
        const uint32_t hi = hi_w < hi_doc ? hi_w : hi_doc;
        // ---- inside the window?  cursors advance by what was taken; a clause's next doc is its first posting that was not ----
        bool p_ok[R];
#pragma unroll
        for (int m = 0; m < R; m++) p_ok[m] = p_in[m] && p_doc[m] < hi;
        if constexpr (FAST) {
            bool upd_l = false;
            uint32_t ndoc_l = hi_c;   // every loaded row taken: the next doc is the one behind them
#pragma unroll
            for (int m = 0; m < R; m++) {
                const unsigned long long okm = __ballot(p_ok[m]);
                const unsigned long long overm = __ballot(p_in[m] && !p_ok[m]);
                const uint32_t cnt = (uint32_t)__popcll(okm);
                const uint32_t first_over = overm ? rl_u32(p_doc[m], __ffsll((long long)overm) - 1) : 0xffffffffu;
                if (lane == p_c[m]) {
                    cur_l += cnt;
                    if (overm && !upd_l) {
                        ndoc_l = first_over;
                        upd_l = true;
                    }
                }
                postings += lane == 0 ? cnt : 0u;
            }
            if (active) cdoc_l = ndoc_l;
        } else {
            for (int c = 0; c < C; c++) {
                uint32_t cnt = 0;
#pragma unroll
                for (int m = 0; m < R; m++) cnt += (uint32_t)__popcll(__ballot(p_ok[m] && p_c[m] == c));
                if (lane == c) cur_l += cnt;
                postings += lane == 0 ? cnt : 0u;
            }
            const bool left_some = active && cur_l < end_l;
            const uint32_t nd = (left_some ? ids_l : a.doc_ids)[left_some ? cur_l : 0ull];
            if (active) cdoc_l = left_some ? nd : 0xffffffffu;
        }
        // ---- round trip 2: the packed tf | fieldnorm id << 24 word of the postings that are scored (no gather by doc id) ----
        uint32_t p_tf[R], p_fn[R], p_raw[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const uint32_t occur = p_at[m] & 0xff, mode = (p_at[m] >> 8) & 0xff;
            const bool scored = p_ok[m] && occur != 2 && mode != 2;
            const uint32_t word = ((p_at[m] >> 16) && scored ? a.aux_tfs : a.tfs)[scored ? p_idx[m] : 0ull];
            p_raw[m] = word;
            p_fn[m] = word >> 24;
            p_tf[m] = (scored && mode == 0) ? (word & 0xffffffu) : 1u;
        }
        float p_score[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const uint32_t mode = (p_at[m] >> 8) & 0xff;
            if (mode == 2) p_score[m] = p_w[m];  // ConstScorer(boost)
            else if (mode == 3) p_score[m] = p_w[m] * __uint_as_float(p_raw[m]);   // a materialised sub-query: boost x its own score
            else {
                const float tf = (float)p_tf[m];   // mode 1: tf == 1
                p_score[m] = p_w[m] * (tf / (tf + tf_cache[p_fn[m]]));
            }
        }
        const unsigned long long cy_b = clock64();
        // ---- probe: every posting finds (or claims) its document's slot; the claimer OWNS the document for the fold ----
        uint32_t p_slot[R];
        uint32_t owned = 0;
        {
            uint32_t old[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                p_slot[m] = (p_doc[m] * 2654435761u) >> TSHIFT;
                old[m] = p_ok[m] ? atomicCAS(&t_key[p_slot[m]], BM25_EMPTY, p_doc[m]) : BM25_EMPTY;
            }
#pragma unroll
            for (int m = 0; m < R; m++) {
                if (!p_ok[m]) continue;
                if (old[m] == BM25_EMPTY) { owned |= 1u << m; continue; }
                uint32_t o = old[m], h = p_slot[m];
                while (o != p_doc[m]) {   // another document's slot: linear probing (the table is at most half full)
                    h = (h + 1) & (T - 1);
                    o = atomicCAS(&t_key[h], BM25_EMPTY, p_doc[m]);
                    if (o == BM25_EMPTY) { owned |= 1u << m; break; }
                }
                p_slot[m] = h;
            }
        }
        // ---- apply in clause order (LDS operations of one wave execute in order): FAST rows are in clause order already ----
        if constexpr (FAST) {
#pragma unroll
            for (int m = 0; m < R; m++)
                if (p_ok[m]) {
                    if ((p_at[m] & 0xff) != 2) __hip_atomic_fetch_add(&t_val[p_slot[m]].acc, p_score[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(&t_val[p_slot[m]].mask, (M)1 << p_c[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        } else {
            for (int c = 0; c < C; c++) {
#pragma unroll
                for (int m = 0; m < R; m++)
                    if (p_ok[m] && p_c[m] == c) {
                        if ((p_at[m] & 0xff) != 2) __hip_atomic_fetch_add(&t_val[p_slot[m]].acc, p_score[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(&t_val[p_slot[m]].mask, (M)1 << c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            }
        }
        const unsigned long long cy_c = clock64();
        // ---- fold the window into the top-k, count matches, clear the table: each lane folds the documents it owns ----
        uint32_t matched_here = 0;
#pragma unroll
        for (int m = 0; m < R; m++) {
            bool ok = false;
            uint64_t ck = NIDX_EMPTY_KEY;
            if (owned & (1u << m)) {
                const uint32_t i = p_slot[m];
                const uint32_t d = p_doc[m];
                const BwSlot<M> v = t_val[i];
                ok = (v.mask & must_m) == must_m && (v.mask & not_m) == 0 && (any_required || (v.mask & should_m) != 0);
#pragma unroll
                for (int g = 0; g < BW_MAX_GROUPS; g++)
                    if (g < n_groups && group_m[g] != 0 && (v.mask & group_m[g]) == 0) ok = false;   // (group ids need not be dense)
                if (ok && a.alive) ok = bit_test(a.alive, d);
                if (ok && mbits) atomicOr(&mbits[d >> 5], 1u << (d & 31));
                if (ok && a.order_key) {
                    // order_by_fast_field: the fast value's dense rank decides, then the lower doc id
                    const uint32_t r = a.order_key[d];
                    ck = ((uint64_t)(a.order_desc ? r : ~r) << 32) | (uint64_t)(~d);
                } else if (ok) {
                    float s = v.acc;
                    if (has_after) {
                        // tweak_score: -inf for docs not after the cursor
                        uint64_t addr = ((uint64_t)a.segment_ord << 32) | d;
                        int32_t sk = total_key(s);
                        bool after = sk < after_key || (sk == after_key && (after_tie == 0 || (after_tie == 1 && addr > after_addr)));
                        if (!after) s = -INFINITY;
                    }
                    ck = rank_key(s, d);
                }
                t_key[i] = BM25_EMPTY;
                t_val[i].acc = 0.f;
                t_val[i].mask = 0;
            }
            const unsigned long long okm = __ballot(ok);
            matched_here += (uint32_t)__popcll(okm);
            unsigned long long mm = __ballot(ok && ck > kth);
            while (mm) {
                const int src = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const uint64_t nk = lane_bcast_u64(ck, src);
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
            }
        }
        total += matched_here;
        cy_load += cy_b - cy_a;
        cy_apply += cy_c - cy_b;
        cy_fold += clock64() - cy_c;
        n_win++;
    };
    for (;;) {
        // live clauses: those whose next doc is still inside the slice
        const unsigned long long act_m = __ballot(lane < C && cdoc_l < hi_doc);
        if (!act_m) break;
        const int n_act = __popcll(act_m);
        if (n_act <= R) window(std::true_type{}, act_m, n_act);
        else window(std::false_type{}, act_m, n_act);
    }
    if (a.dbg && lane == 0) {
        atomicAdd(&a.dbg[0], cy_load);
        atomicAdd(&a.dbg[1], cy_apply);
        atomicAdd(&a.dbg[2], cy_fold);
        atomicAdd(&a.dbg[3], clock64() - cy_t0);
        atomicAdd(&a.dbg[4], n_win);
        atomicAdd(&a.dbg[5], 1ull);
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) a.out_key[(size_t)item * k + e] = valid ? key : NIDX_EMPTY_KEY;
    }
    postings = lane_bcast_u64(postings, 0);
    if (lane == 0) {
        a.out_count[item] = cnt;
        a.out_total[item] = total;
        a.out_postings[item] = postings;
    }
}

// =====================================================================================================
// The lean form for queries of at most R clauses (the usual request: a few keywords plus a few filters).  Same
// arithmetic, same window rule and same results as bm25_rows_kernel's FAST window; what differs is the instruction count:
// a clause's list is a wave-uniform base pointer + a 32-bit position (global loads with a scalar base), the row -> clause
// table and the cursor bookkeeping run on the scalar unit, in-window tests are mask arithmetic on ballots, and the table is
// 80 R slots (load factor <= 0.8, ~0.5 typical) so that five workgroups of four items fit a CU.
// =====================================================================================================
template <int KL, int R>
__global__ __launch_bounds__(256) void bm25_fast_kernel(Bm25Args a, const uint32_t *items, uint32_t n_items) {
    constexpr uint32_t T = 80 * R;
    const unsigned long long cy_entry = clock64();
    __shared__ float tf_cache[256];
    __shared__ uint32_t t_key_all[4][T];
    __shared__ uint2 t_val_all[4][T];    // x: the f32 sum's bits, y: hit mask (bit c = clause c)
    __shared__ uint32_t s_group_all[4][BW_MAX_GROUPS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *t_key = t_key_all[wave];
    uint2 *t_val = t_val_all[wave];
    uint32_t *s_group = s_group_all[wave];
    // the item's header (item -> work record -> clause records) is a chain of dependent loads: start it before the table is cleared
    const uint32_t slot_in_grid = blockIdx.x * 4u + (uint32_t)wave;
    const bool has_item = slot_in_grid < n_items;
    const uint32_t item = has_item ? items[slot_in_grid] : 0u;
    Bm25Work work = {0u, 0u, 1u, 0u, 0u};
    if (has_item) work = a.work[item];
    tf_cache[threadIdx.x] = a.tf_cache[threadIdx.x];
    for (uint32_t i = lane; i < T; i += 64) {
        t_key[i] = BM25_EMPTY;
        t_val[i] = make_uint2(0u, 0u);
    }
    __syncthreads();   // the only workgroup barrier: the tf cache
    if (!has_item) return;
    const uint32_t q = work.query;
    const uint64_t c0 = work.clause_first;
    const int C = (int)work.n_clauses;   // <= R
    const int k = (int)a.k;

    // ---- lane c holds clause c: list = base pointers + [pos, len) ----
    uint32_t attr_l = 0, pos_l = 0, len_l = 0;
    float weight_l = 0.f;
    const uint32_t *ids_l = a.doc_ids, *tfs_l = a.tfs;
    if (lane < C) {
        const Bm25ClauseDev cd = a.clauses[c0 + lane];
        const bool aux = (cd.term & BM25_AUX_TERM) != 0;
        const uint32_t ti = cd.term & ~BM25_AUX_TERM;
        attr_l = (uint32_t)cd.occur | ((uint32_t)cd.mode << 8);
        weight_l = cd.weight;
        const unsigned long long b = aux ? a.aux_offsets[2 * ti] : a.term_offsets[ti];
        const unsigned long long e = aux ? a.aux_offsets[2 * ti + 1] : a.term_offsets[ti + 1];
        ids_l = (aux ? a.aux_doc_ids : a.doc_ids) + b;
        if (cd.mode != 2) tfs_l = (aux ? a.aux_tfs : a.tfs) + b;   // tf | fieldnorm id << 24 per posting; ConstScorer clauses read neither
        len_l = (uint32_t)(e - b);
    }
    const uint32_t occur_l = attr_l & 0xff;
    const uint32_t must_m = (uint32_t)__ballot(lane < C && occur_l == 1);
    const uint32_t not_m = (uint32_t)__ballot(lane < C && occur_l == 2);
    const uint32_t should_m = (uint32_t)__ballot(lane < C && occur_l == 0);
    int n_groups = 0;
#pragma unroll
    for (int g = 0; g < BW_MAX_GROUPS; g++) {
        const uint32_t gm = (uint32_t)__ballot(lane < C && occur_l == 3u + (uint32_t)g);
        if (gm) {   // dense list of the non-empty groups
            if (lane == 0) s_group[n_groups] = gm;
            n_groups++;
        }
    }
    const bool any_required = must_m != 0 || n_groups > 0;
    uint32_t hi_doc = 0xffffffffu;
    if (work.n_slices > 1) {
        const uint32_t lo_doc = (uint32_t)((unsigned long long)a.n_docs * work.slice / work.n_slices);
        if (work.slice + 1 < work.n_slices) hi_doc = (uint32_t)((unsigned long long)a.n_docs * (work.slice + 1) / work.n_slices);
        if (work.slice > 0) {
            // first posting >= lo_doc of every clause.  The clauses search side by side: the wave is cut into groups of G = 64 / 2^ceil(log2 C)
            // lanes, group c runs a G-ary search over clause c's list (one probe per lane and step), so the number of dependent round
            // trips is that of ONE search (log_G len) instead of C searches.
            const int g_log = C <= 1 ? 6 : C <= 2 ? 5 : C <= 4 ? 4 : 3;
            const uint32_t G = 1u << g_log;
            const int grp = lane >> g_log;
            const uint32_t li = (uint32_t)lane & (G - 1u);
            const bool g_live = grp < C;
            const uint32_t *ids = reinterpret_cast<const uint32_t *>(
                ((uint64_t)(uint32_t)__shfl((int)((uint64_t)(uintptr_t)ids_l >> 32), grp) << 32) | (uint32_t)__shfl((int)(uint32_t)(uintptr_t)ids_l, grp));
            uint32_t left = 0, right = g_live ? (uint32_t)__shfl((int)len_l, grp) : 0u;   // uniform inside a group
            const unsigned long long g_mask = (G == 64u ? ~0ull : ((1ull << G) - 1ull));
            for (;;) {
                const bool wide = right - left > G;
                if (!__ballot(wide)) break;
                const uint32_t step = (right - left + G - 1u) / G;
                const uint32_t probe = left + step * li;
                const uint32_t v = ids[wide && probe < right ? probe : 0u];   // unconditional load (index 0 of a padded list is always readable)
                const bool ge = (wide && probe < right) ? v >= lo_doc : true;
                const unsigned long long m = (__ballot(ge) >> (grp << g_log)) & g_mask;
                const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : G;
                if (wide) {
                    const uint32_t nl = first == 0u ? left : left + step * (first - 1u);
                    const uint32_t nr = left + step * first;
                    left = nl;
                    right = nr < right ? nr : right;
                }
            }
            {
                const uint32_t probe = left + li;
                const uint32_t v = ids[probe < right ? probe : 0u];
                const bool ge = probe < right ? v >= lo_doc : true;
                const unsigned long long m = (__ballot(ge) >> (grp << g_log)) & g_mask;
                const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : G;
                const uint32_t res = left + first < right ? left + first : right;
                const uint32_t mine = (uint32_t)__shfl((int)res, (lane << g_log) & 63);   // lane c takes group c's answer
                if (lane < C) pos_l = mine;
            }
        }
    }
    uint32_t cdoc_l = 0xffffffffu;   // the clause's next doc inside the slice
    if (lane < C && pos_l < len_l) cdoc_l = ids_l[pos_l];
    unsigned long long cy_load = 0, cy_apply = 0, cy_fold = 0, n_win = 0, postings = 0, total = 0;
    const unsigned long long cy_t0 = clock64();

    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    const bool has_after = a.after != nullptr && a.after[q].has_after != 0;
    const int32_t after_key = has_after ? total_key(a.after[q].score) : 0;
    const int after_tie = has_after ? a.after[q].tie_break : 0;
    const uint64_t after_addr = has_after ? a.after[q].docaddr : 0;
    const int mslot = a.match_slot ? a.match_slot[q] : -1;
    uint32_t *mbits = mslot >= 0 ? a.match_bits + (size_t)mslot * a.match_words : nullptr;
    const bool extras = a.alive != nullptr || mbits != nullptr || a.order_key != nullptr || has_after;
    uint32_t matched_l = 0;   // documents this lane folded that matched (Count collector), summed once at the end

    for (;;) {
        const unsigned long long act_m = __ballot(lane < C && cdoc_l < hi_doc);
        if (!act_m) break;
        const bool active = (act_m >> lane) & 1ull;
        const int n_act = __popcll(act_m);
        // rows: one per live clause, the rest in proportion to the list remainders
        const float left_f = active ? (float)(len_l - pos_l) : 0.f;
        const float tot = wave_butterfly_sum(left_f);
        uint32_t rows_l = active ? 1u + (uint32_t)((float)(R - n_act) * (left_f / tot)) : 0u;
        // wave-uniform maps, one nibble per row / clause: the clause of row m, and the first row of clause j
        uint32_t run = 0, row_map = 0, start_map = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const uint32_t rj = rl_u32(rows_l, j);
            start_map |= run << (4 * j);
            if (rj) row_map |= ((uint32_t)j * 0x11111111u) & (((rj >= 8 ? 0u : (1u << (4 * rj))) - 1u) << (4 * run));
            run += rj;
        }
        if (run > (uint32_t)R) {   // float rounding pushed the shares past R: one row each
            rows_l = active ? 1u : 0u;
            run = 0, row_map = 0, start_map = 0;
#pragma unroll
            for (int j = 0; j < R; j++) {
                const uint32_t rj = rl_u32(rows_l, j);
                start_map |= run << (4 * j);
                if (rj) row_map |= ((uint32_t)j * 0x11111111u) & (0xfu << (4 * run));
                run += rj;
            }
        }
        const uint32_t n_rows = run;
        const unsigned long long cy_a = clock64();
        // ---- round trip 1 ----
        uint32_t p_doc[R], p_idx[R], p_tf[R];
        int row_c[R];
        unsigned long long in_m[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const int c = (int)((row_map >> (4 * m)) & 15u);   // rows past n_rows map to clause 0 and are masked out
            row_c[m] = c;
            const uint32_t first_row = (start_map >> (4 * c)) & 15u;
            const uint32_t *ids = reinterpret_cast<const uint32_t *>(lane_bcast_u64((uint64_t)(uintptr_t)ids_l, c));
            p_idx[m] = rl_u32(pos_l, c) + 64u * ((uint32_t)m - first_row) + (uint32_t)lane;
            in_m[m] = (uint32_t)m < n_rows ? __ballot(p_idx[m] < rl_u32(len_l, c)) : 0ull;
            p_doc[m] = ids[p_idx[m]];   // unconditional (the arrays are padded): every row's load is in flight before the first wait
            const uint32_t *tfs = reinterpret_cast<const uint32_t *>(lane_bcast_u64((uint64_t)(uintptr_t)tfs_l, c));
            p_tf[m] = tfs[(rl_u32(attr_l, c) >> 8) != 2u ? p_idx[m] : 0u];   // tf | fieldnorm id << 24: same round trip, no gather by doc id
        }
        const uint32_t next_l = pos_l + 64u * rows_l;
        const bool more_l = active && next_l < len_l;
        uint32_t hi_c = ids_l[more_l ? next_l : 0u];
        hi_c = more_l ? hi_c : 0xffffffffu;
        const uint32_t hi_w = wave_min_u32(hi_c);
        const uint32_t hi = hi_w < hi_doc ? hi_w : hi_doc;
        // ---- in-window masks, cursors, next docs ----
        unsigned long long ok_m[R];
        bool upd_l = false;
        uint32_t ndoc_l = hi_c;
#pragma unroll
        for (int m = 0; m < R; m++) {
            ok_m[m] = in_m[m] & __ballot(p_doc[m] < hi);
            const unsigned long long over = in_m[m] & ~ok_m[m];
            const uint32_t cnt = (uint32_t)__popcll(ok_m[m]);
            const uint32_t first_over = over ? rl_u32(p_doc[m], __ffsll((long long)over) - 1) : 0xffffffffu;
            if (lane == row_c[m]) {
                pos_l += cnt;
                if (over && !upd_l) {
                    ndoc_l = first_over;
                    upd_l = true;
                }
            }
            postings += cnt;
        }
        if (active) cdoc_l = ndoc_l;
        float p_score[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const uint32_t mode = rl_u32(attr_l, row_c[m]) >> 8;
            const float w = rl_f32(weight_l, row_c[m]);
            const float tf = mode == 0 ? (float)(p_tf[m] & 0xffffffu) : 1.0f;
            const float bm = w * (tf / (tf + tf_cache[p_tf[m] >> 24]));
            p_score[m] = mode == 2 ? w : mode == 3 ? w * __uint_as_float(p_tf[m]) : bm;   // ConstScorer(boost); a materialised sub-query
        }
        const unsigned long long cy_b = clock64();
        // ---- probe: every row's first CAS is issued before any result is looked at; collisions (another document in the slot)
        //      are the rare case and go through one loop afterwards ----
        uint32_t p_slot[R];
        uint32_t owned = 0, pend = 0;
        {
            uint32_t old[R];
#pragma unroll
            for (int m = 0; m < R; m++) {
                p_slot[m] = __umulhi(p_doc[m] * 2654435761u, T);
                const bool ok = (ok_m[m] >> lane) & 1ull;
                old[m] = ok ? atomicCAS(&t_key[p_slot[m]], BM25_EMPTY, p_doc[m]) : p_doc[m];
            }
#pragma unroll
            for (int m = 0; m < R; m++) {
                const bool ok = (ok_m[m] >> lane) & 1ull;
                owned |= (ok && old[m] == BM25_EMPTY) ? 1u << m : 0u;
                pend |= (old[m] != BM25_EMPTY && old[m] != p_doc[m]) ? 1u << m : 0u;
            }
        }
        if (__ballot(pend != 0)) {
#pragma unroll
            for (int m = 0; m < R; m++) {
                if (!(pend & (1u << m))) continue;
                uint32_t h = p_slot[m], o;
                do {   // linear probing (the table is at most 0.8 full)
                    h = h + 1 == T ? 0 : h + 1;
                    o = atomicCAS(&t_key[h], BM25_EMPTY, p_doc[m]);
                } while (o != BM25_EMPTY && o != p_doc[m]);
                if (o == BM25_EMPTY) owned |= 1u << m;
                p_slot[m] = h;
            }
        }
        // ---- apply: rows are in clause order, one wave's LDS operations execute in order ----
#pragma unroll
        for (int m = 0; m < R; m++) {
            const bool ok = (ok_m[m] >> lane) & 1ull;
            const uint32_t occur = rl_u32(attr_l, row_c[m]) & 0xff;
            if (ok) {
                if (occur != 2) __hip_atomic_fetch_add(reinterpret_cast<float *>(&t_val[p_slot[m]].x), p_score[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&t_val[p_slot[m]].y, 1u << row_c[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        const unsigned long long cy_c = clock64();
        // ---- fold: every lane finishes the documents it owns (boolean structure on the hit mask, then the optional alive /
        //      facet / order / cursor steps) and keeps its best key; matches are counted per lane and summed once per item; the
        //      top-k list is only touched when some lane holds a key above the k-th (after the list has warmed up: rarely) ----
        uint64_t p_ck[R];
        uint64_t best_l = NIDX_EMPTY_KEY;
#pragma unroll
        for (int m = 0; m < R; m++) {
            bool ok = false;
            uint64_t ck = NIDX_EMPTY_KEY;
            if (owned & (1u << m)) {
                const uint32_t i = p_slot[m];
                const uint32_t d = p_doc[m];
                const uint2 v = t_val[i];
                ok = (v.y & must_m) == must_m && (v.y & not_m) == 0 && (any_required || (v.y & should_m) != 0);
                for (int g = 0; g < n_groups; g++)
                    if ((v.y & s_group[g]) == 0) ok = false;
                if (extras) {   // wave-uniform: alive bitset, facet bitset, order by a fast field, search-after cursor
                    if (ok && a.alive) ok = bit_test(a.alive, d);
                    if (ok && mbits) atomicOr(&mbits[d >> 5], 1u << (d & 31));
                }
                if (ok) {
                    if (extras && a.order_key) {
                        const uint32_t r = a.order_key[d];
                        ck = ((uint64_t)(a.order_desc ? r : ~r) << 32) | (uint64_t)(~d);
                    } else {
                        float s = __builtin_bit_cast(float, v.x);
                        if (extras && has_after) {
                            uint64_t addr = ((uint64_t)a.segment_ord << 32) | d;
                            int32_t sk = total_key(s);
                            bool after = sk < after_key || (sk == after_key && (after_tie == 0 || (after_tie == 1 && addr > after_addr)));
                            if (!after) s = -INFINITY;
                        }
                        ck = rank_key(s, d);
                    }
                }
                t_key[i] = BM25_EMPTY;
                t_val[i] = make_uint2(0u, 0u);
            }
            matched_l += ok ? 1u : 0u;
            p_ck[m] = ck;
            best_l = ck > best_l ? ck : best_l;
        }
        if (__ballot(best_l > kth)) {
#pragma unroll
            for (int m = 0; m < R; m++) {
                unsigned long long mm = __ballot(p_ck[m] > kth);
                while (mm) {
                    const int src = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const uint64_t nk = lane_bcast_u64(p_ck[m], src);
                    if (nk > kth) kth = top.insert_kth(nk, k, lane);
                }
            }
        }
        cy_load += cy_b - cy_a;
        cy_apply += cy_c - cy_b;
        cy_fold += clock64() - cy_c;
        n_win++;
    }
    if (a.dbg && lane == 0) {
        atomicAdd(&a.dbg[0], cy_load);
        atomicAdd(&a.dbg[1], cy_apply);
        atomicAdd(&a.dbg[2], cy_fold);
        atomicAdd(&a.dbg[3], clock64() - cy_t0);
        atomicAdd(&a.dbg[4], n_win);
        atomicAdd(&a.dbg[5], 1ull);
        atomicAdd(&a.dbg[6], cy_t0 - cy_entry);
        atomicMax(&a.dbg[7], clock64() - cy_entry);
        atomicMax(&a.dbg[8], n_win);
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) a.out_key[(size_t)item * k + e] = valid ? key : NIDX_EMPTY_KEY;
    }
    total = wave_sum_u32(matched_l);
    if (lane == 0) {
        a.out_count[item] = cnt;
        a.out_total[item] = total;
        a.out_postings[item] = postings;
    }
}

// ---- per-query merge of the slices' lists (TopDocs merges its per-segment collectors the same way): one workgroup per query;
//      its four waves walk the work items' sorted key lists side by side (every fourth chunk of 64 keys each, stopping short of
//      nothing: a key below the wave's running k-th is dropped by one compare), then wave 0 folds the other three lists in ----
template <int KL>
__global__ __launch_bounds__(256) void bm25_merge_kernel(Bm25MergeArgs m) {
    __shared__ uint64_t part[3][64 * KL];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q = blockIdx.x;
    if (m.ablate == 2) return;
    const uint32_t w0 = m.item_first[q], w1 = m.item_first[q + 1];
    const int k = (int)m.k;
    const uint32_t n_items = w1 - w0;
    // Count / postings: every lane of wave 0 sums its share of the items
    unsigned long long total = 0, postings = 0;
    if (wave == 0) {
        for (uint32_t w = w0 + (uint32_t)lane; w < w1; w += 64) {
            total += m.item_total[w];
            postings += m.item_postings[w];
        }
        total = wave_sum_u64(total);
        postings = wave_sum_u64(postings);
    }
    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    auto consume = [&](uint64_t key) {   // 64 keys, one per lane
        unsigned long long mm = __ballot(key > kth);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const uint64_t nk = lane_bcast_u64(key, src);
            if (nk > kth) kth = top.insert_kth(nk, k, lane);
        }
    };
    if (n_items == 1) {
        // one slice: its list is the answer
        const uint32_t cnt = m.item_count[w0];
#pragma unroll
        for (int i = 0; i < KL; i++) {
            const int e = 64 * i + lane;
            top.l[i].key = e < (int)cnt && e < k ? m.item_key[(size_t)w0 * k + e] : NIDX_EMPTY_KEY;
        }
    } else {
        // all the slices' keys side by side, 64 at a time: the loads of a chunk do not wait for any other list (nor for the item's
        // count: the key is fetched beside it and dropped afterwards).  k <= 64: a chunk goes through the bitonic network of
        // wave_bitonic.h — 27 compare-exchange steps whatever the keys — instead of up to 64 insertions one after the other
        const uint32_t slots = n_items * (uint32_t)k;
        for (uint32_t base = 64u * (uint32_t)wave; base < slots; base += 256) {
            const uint32_t idx = base + (uint32_t)lane;
            const uint32_t it = idx < slots ? idx / (uint32_t)k : 0u, pos = idx - it * (uint32_t)k;
            const uint32_t cnt_i = m.item_count[w0 + it];
            const uint64_t key_i = m.item_key[(size_t)(w0 + it) * k + (idx < slots ? pos : 0u)];
            const uint64_t key = idx < slots && pos < cnt_i ? key_i : NIDX_EMPTY_KEY;
            if constexpr (KL == 1) top.l[0].key = bs_merge64(top.l[0].key, key);
            else consume(key);
        }
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < KL; i++) part[wave - 1][64 * i + lane] = top.l[i].key;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    if (n_items > 1) {
        const uint32_t slots = n_items * (uint32_t)k;
        for (int w = 0; w < 3; w++) {
            if (64u * (uint32_t)(w + 1) >= slots) break;   // that wave had no chunk
            if constexpr (KL == 1) {
                top.l[0].key = bs_merge_sorted(top.l[0].key, part[w][63 - lane]);
            } else {
#pragma unroll
                for (int i = 0; i < KL; i++) consume(part[w][64 * i + lane]);
            }
        }
    }
    uint32_t cnt = 0;
    if (m.ablate == 3) {
        if (top.mine(0) == 12345ull && lane == 0) m.out_count[q] = 1;   // (keeps the merge alive)
        return;
    }
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) {
            uint32_t d = valid ? rank_key_addr(key) : 0xffffffffu;
            if (m.seg_base) {
                // DocAddress of a resident doc: the last segment whose first doc is <= d (empty segments share their base with the next)
                uint32_t lo = 0, hi = m.n_seg;
                while (valid && hi - lo > 1) {
                    const uint32_t mid = lo + (hi - lo) / 2;
                    if (m.seg_base[mid] <= d) lo = mid;
                    else hi = mid;
                }
                if (valid) d -= m.seg_base[lo];
                m.out_seg[(size_t)q * k + e] = lo;
            }
            m.out_doc[(size_t)q * k + e] = d;
            m.out_score[(size_t)q * k + e] = valid ? rank_key_score(key) : 0.f;
        }
    }
    if (lane == 0) {
        m.out_count[q] = cnt;
        m.out_total[q] = total;
        m.out_postings[q] = postings;
    }
}

hipError_t launch_bm25_merge(const Bm25MergeArgs &m, uint32_t n_queries, hipStream_t s) {
    if (n_queries == 0) return hipSuccess;
    if (m.k > 256) hipLaunchKernelGGL(bm25_merge_kernel<8>, dim3(n_queries), dim3(256), 0, s, m);
    else if (m.k > 64) hipLaunchKernelGGL(bm25_merge_kernel<4>, dim3(n_queries), dim3(256), 0, s, m);
    else hipLaunchKernelGGL(bm25_merge_kernel<1>, dim3(n_queries), dim3(256), 0, s, m);
    return hipGetLastError();
}

template <int KL, typename M>
static void launch_rows(const Bm25Args &a, const uint32_t *items, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL((bm25_rows_kernel<KL, 4, M>), dim3((n + 3) / 4), dim3(256), 0, s, a, items, n);
}

// Two launches over disjoint item lists (device arrays of indices into a.work): `fast` = the items of queries with at most
// BM25_FAST_CLAUSES clauses, `wide` = the rest (max_clauses: the most clauses any of them has; <= 32: 32-bit hit masks).
hipError_t launch_bm25_search(const Bm25Args &a, const uint32_t *fast_items, uint32_t n_fast, const uint32_t *wide_items, uint32_t n_wide,
                              uint32_t max_clauses, hipStream_t s) {
    if (n_fast) {
        const dim3 grid((n_fast + 3) / 4), block(256);
        if (a.k > 256) hipLaunchKernelGGL((bm25_fast_kernel<8, BM25_FAST_CLAUSES>), grid, block, 0, s, a, fast_items, n_fast);
        else if (a.k > 64) hipLaunchKernelGGL((bm25_fast_kernel<4, BM25_FAST_CLAUSES>), grid, block, 0, s, a, fast_items, n_fast);
        else hipLaunchKernelGGL((bm25_fast_kernel<1, BM25_FAST_CLAUSES>), grid, block, 0, s, a, fast_items, n_fast);
    }
    if (n_wide) {
        if (max_clauses > 32) {
            if (a.k > 256) launch_rows<8, unsigned long long>(a, wide_items, n_wide, s);
            else if (a.k > 64) launch_rows<4, unsigned long long>(a, wide_items, n_wide, s);
            else launch_rows<1, unsigned long long>(a, wide_items, n_wide, s);
        } else {
            if (a.k > 256) launch_rows<8, uint32_t>(a, wide_items, n_wide, s);
            else if (a.k > 64) launch_rows<4, uint32_t>(a, wide_items, n_wide, s);
            else launch_rows<1, uint32_t>(a, wide_items, n_wide, s);
        }
    }
    return hipGetLastError();
}

}  // namespace nidx
