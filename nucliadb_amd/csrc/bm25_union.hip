// bm25_union.hip — the BM25 scoring kernel for term unions whose posting lists rarely meet (gfx950).
//
// Same contract as bm25_fast_kernel (bm25.hip): one WAVE per work item = (query, doc-id slice), a query of <= 8 plain term
// clauses, lockstep doc-id windows of R = 8 rows x 64 postings, a row belongs to ONE clause; it computes what tantivy computes under
// TextReaderService::do_search / the paragraph Searcher::do_search (nidx_text/src/reader.rs:433-435, nidx_paragraph/src/reader.rs:
// 244-348): Bm25Weight::score per posting, BooleanQuery sums in clause order, TopDocs (score desc, DocAddress asc), Count.
//
// What differs is how the postings of one document are brought together.  bm25_fast_kernel hashes every posting into an LDS
// table (CAS + float add + or + read + clear per posting, 12 B per slot).  Here a posting is FINAL the moment it is scored unless
// its document also occurs in another row of the window — and for keyword unions almost none does (3 lists of ~1 k postings over
// 10 M documents).  So:
//   1. every in-window posting sets one bit of a 16 Kibit LDS bitmap A (ds_or_rtn): a bit that was already set marks a POSSIBLE
//      second posting of the same document (or a hash collision: the bitmap is a filter, not the truth);
//   2. those postings set the same bit in bitmap B; then every posting reads B: a set bit = "involved";
//   3. the ~2 % involved postings are compacted into an LDS list and resolved EXACTLY by an all-pairs pass in row order (rows are in
//      clause order, so a document's f32 sum is built in clause order like the oracle's term-at-a-time loop): the first posting of
//      a document owns it, gets the sum and the clause mask, and is tested against the query's boolean structure;
//   4. everything else needs no table at all: score -> one float compare with the k-th score -> (rarely) rank key -> top-k list.
// A window whose involved postings do not fit the list (256) is cut in half (its rows stay in registers) and retried: exact for any
// input.  The host routes a query here when the expected number of shared documents (from the list lengths) is small; the same
// queries through bm25_fast_kernel give bit-identical results (tests/test_bm25_gpu.py runs both, and forces this kernel onto
// queries that make its slow path the common one).
// Bookkeeping is lane-parallel: lanes 0..7 hold the clauses' cursors, the window plan is a DPP prefix sum over them, lane m then
// builds row m's descriptor (list base, valid lanes, weight) with seven ds_bpermute gathers, and the unrolled row code reads it with
// immediate-lane v_readlane — no per-row table walks on the scalar unit.  "tf == 1" rows (nearly all) score with one LDS read and
// one multiply (w * inv1[fieldnorm]: the same two f32 operations as the general quotient).
// LDS: 4 KiB per wave + 2 KiB per workgroup.  Bound: HBM by bytes (8 per posting), VALU issue in practice.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

typedef const __attribute__((address_space(4))) uint32_t *cu32_t;
typedef uint32_t bu_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) bu_u32x4 *cu32x4_t;
template <typename T>
__device__ inline cu32_t as_const_words(const T *p) { return (cu32_t)(uintptr_t)p; }
// LDS traffic of one wave executes in order; this only keeps the COMPILER from moving plain LDS accesses across the atomics
__device__ inline void bu_lds_order() { asm volatile("" ::: "memory"); }

#define BU_NB 16384u                 /* bits per bitmap */
#define BU_WORDS (BU_NB / 32u)       /* 512 words = 2 KiB */
#define BU_CAP 256u                  /* involved postings resolved per window: 256 x 16 B = the two bitmaps' 4 KiB */
#define BU_GROUPS 8

__device__ inline uint32_t bu_rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// min over lanes 0..7 (lanes >= 8 hold don't-cares), result valid in lane 0
__device__ inline uint32_t bu_min8(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false);   // row_half_mirror: lane i <- lane 7 - i
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4e, 0xf, 0xf, false);             // quad_perm [2,3,0,1]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xb1, 0xf, 0xf, false);             // quad_perm [1,0,3,2]
    v = o < v ? o : v;
    return bu_rl(v, 0);
}
__device__ inline unsigned long long bu_lanes_below(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

template <int KL, int R, bool EXTRAS>
__global__ __launch_bounds__(256) void bm25_union_kernel(Bm25Args a, const uint32_t *items, uint32_t n_items) {
    static_assert(R == 8, "one row descriptor per lane of the first DPP half row");
    __shared__ float tf_cache[256];          // K1 * (1 - B + B * fieldnorm / avg)
    __shared__ float inv1[256];              // 1 / (1 + tf_cache): the tf == 1 quotient, the same two f32 operations as the general form
    __shared__ uint4 scratch_all[4][2 * BU_WORDS / 4];   // per wave: bitmap A | bitmap B; after the filter passes: the involved list
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long cy_entry = a.dbg ? clock64() : 0;
    uint32_t *bm_a = reinterpret_cast<uint32_t *>(scratch_all[wave]);
    uint32_t *bm_b = bm_a + BU_WORDS;
    uint4 *list = scratch_all[wave];
    {
        const float c = a.tf_cache[threadIdx.x];
        tf_cache[threadIdx.x] = c;
        inv1[threadIdx.x] = 1.0f / (1.0f + c);
        for (uint32_t i = lane; i < 2 * BU_WORDS / 4; i += 64) list[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();   // the only workgroup barrier
    const uint32_t slot_in_grid = blockIdx.x * 4u + (uint32_t)wave;
    if (slot_in_grid >= n_items) return;
    // ---- the item record and the query's clause table: scalar loads ----
    const uint32_t item = as_const_words(items)[slot_in_grid];
    cu32_t wrec = as_const_words(a.work) + (size_t)item * 5u;
    const uint32_t q = wrec[0], slice = wrec[1], n_slices = wrec[2], clause_first = wrec[3];
    const int C = (int)wrec[4];   // <= 8
    const int k = (int)a.k;
    const uint32_t *const doc_ids = a.doc_ids;
    const uint32_t *const tfs = a.tfs;

    // ---- lane c holds clause c's cursor ----
    uint32_t len_l = 0, pos_l = 0, attr_l = 0, w_bits_l = 0;
    unsigned long long b_l = 0;
    if (lane < C) {
        const Bm25UClause uc = a.uclauses[clause_first + lane];
        b_l = ((unsigned long long)uc.b_hi << 32) | uc.b_lo;
        len_l = uc.len;
        attr_l = uc.attr;
        w_bits_l = __float_as_uint(uc.weight);
    }
    const uint32_t occur_l = attr_l & 0xff;
    // the boolean structure as clause masks, parked in lanes 8..10 of a register (read back on the rare path only)
    uint32_t masks_l = 0;
    {
        const uint32_t mm = (uint32_t)__ballot(lane < C && occur_l == 1), nm = (uint32_t)__ballot(lane < C && occur_l == 2),
                       sm = (uint32_t)__ballot(lane < C && occur_l == 0);
        masks_l = lane == 8 ? mm : lane == 9 ? nm : lane == 10 ? sm : 0u;
    }
    uint32_t group_l = 0;   // lane g: the clause mask of the g-th non-empty required Should group
    int n_groups = 0;
#pragma unroll
    for (int g = 0; g < BU_GROUPS; g++) {
        const uint32_t gm = (uint32_t)__ballot(lane < C && occur_l == 3u + (uint32_t)g);
        if (gm) {
            if (lane == n_groups) group_l = gm;
            n_groups++;
        }
    }
    auto mask_ok = [&](uint32_t m) -> bool {   // BooleanQuery: every Must, no MustNot, one clause of every required Should group
        const uint32_t must_m = bu_rl(masks_l, 8), not_m = bu_rl(masks_l, 9), should_m = bu_rl(masks_l, 10);
        const bool any_required = must_m != 0 || n_groups > 0;
        bool ok = (m & must_m) == must_m && (m & not_m) == 0 && (any_required || (m & should_m) != 0);
        for (int g = 0; g < n_groups; g++)
            if ((m & bu_rl(group_l, g)) == 0) ok = false;
        return ok;
    };
    // bit c: a document that occurs in clause c ONLY matches the query
    const uint32_t single_ok_m = (uint32_t)__ballot(lane < C && mask_ok(1u << lane));

    // ---- doc range of the slice; first posting >= lo_doc of every clause (the clauses search side by side, bm25.hip) ----
    uint32_t lo_doc = 0, hi_doc = 0xffffffffu;
    if (n_slices > 1) {
        lo_doc = (uint32_t)((unsigned long long)a.n_docs * slice / n_slices);
        if (slice + 1 < n_slices) hi_doc = (uint32_t)((unsigned long long)a.n_docs * (slice + 1) / n_slices);
        if (slice > 0) {
            const int g_log = C <= 1 ? 6 : C <= 2 ? 5 : C <= 4 ? 4 : 3;
            const uint32_t G = 1u << g_log;
            const int grp = lane >> g_log;
            const uint32_t li = (uint32_t)lane & (G - 1u);
            const bool g_live = grp < C;
            const unsigned long long bg = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(b_l >> 32), grp) << 32) | (uint32_t)__shfl((int)(uint32_t)b_l, grp);
            const uint32_t *ids = doc_ids + bg;
            uint32_t left = 0, right = g_live ? (uint32_t)__shfl((int)len_l, grp) : 0u;
            const unsigned long long g_mask = (G == 64u ? ~0ull : ((1ull << G) - 1ull));
            for (;;) {
                const bool wide = right - left > G;
                if (!__ballot(wide)) break;
                const uint32_t step = (right - left + G - 1u) / G;
                const uint32_t probe = left + step * li;
                const uint32_t v = ids[wide && probe < right ? probe : 0u];
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
            const uint32_t probe = left + li;
            const uint32_t v = ids[probe < right ? probe : 0u];
            const bool ge = probe < right ? v >= lo_doc : true;
            const unsigned long long m = (__ballot(ge) >> (grp << g_log)) & g_mask;
            const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : G;
            const uint32_t res = left + first < right ? left + first : right;
            const uint32_t mine = (uint32_t)__shfl((int)res, (lane << g_log) & 63);
            if (lane < C) pos_l = mine;
        }
    }

    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    float kth_score = rank_key_score(kth);   // NaN while the list is not full: !(s < NaN) lets every score through to the exact test
    const bool has_after = EXTRAS && a.after != nullptr && a.after[q].has_after != 0;
    const int32_t after_key = has_after ? total_key(a.after[q].score) : 0;
    const int after_tie = has_after ? a.after[q].tie_break : 0;
    const uint64_t after_addr = has_after ? a.after[q].docaddr : 0;
    const int mslot = EXTRAS && a.match_slot ? a.match_slot[q] : -1;
    uint32_t *mbits = mslot >= 0 ? a.match_bits + (size_t)mslot * a.match_words : nullptr;
    // alive / facet bitset / order by a fast field / search-after cursor: the steps behind the boolean test (EXTRAS only)
    auto finish = [&](bool &ok, uint32_t d, float s) -> uint64_t {
        if constexpr (EXTRAS) {
            if (ok && a.alive) ok = bit_test(a.alive, d);
            if (ok && mbits) atomicOr(&mbits[d >> 5], 1u << (d & 31));
            if (ok && a.order_key) {
                const uint32_t r = a.order_key[d];
                return ((uint64_t)(a.order_desc ? r : ~r) << 32) | (uint64_t)(~d);
            }
            if (ok && has_after) {
                const uint64_t addr = ((uint64_t)a.segment_ord << 32) | d;
                const int32_t sk = total_key(s);
                const bool after = sk < after_key || (sk == after_key && (after_tie == 0 || (after_tie == 1 && addr > after_addr)));
                if (!after) s = -INFINITY;
            }
        }
        return ok ? rank_key(s, d) : NIDX_EMPTY_KEY;
    };
    unsigned long long n_offer = 0;   // (debug trace)
    auto offer = [&](uint64_t ck) {   // candidates of one row
#ifdef BU_EXPERIMENT_NO_OFFER
        if (ck == 1234567ull) kth = ck;
        return;
#endif
        unsigned long long mm = __ballot(ck > kth);
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const uint64_t nk = lane_bcast_u64(ck, src);
            if (nk > kth) {
                kth = top.insert_kth(nk, k, lane);
                n_offer++;
            }
        }
        kth_score = rank_key_score(kth);
    };

    uint32_t postings = 0, total = 0, n_win = 0;   // per item: well below 2^32
    const bool multi_slice = n_slices > 1;
    uint32_t dead_m = 0;          // bit c: clause c's next posting is beyond the slice
    uint32_t win_lo = lo_doc;     // lower doc bound of the current window
    const unsigned long long cy_t0 = a.dbg ? clock64() : 0;
    unsigned long long cy_load = 0, cy_filter = 0, cy_final = 0, cy_resolve = 0;
    for (;;) {
        const unsigned long long cw0 = a.dbg ? clock64() : 0;
        // ---- plan (lanes 0..7 = clauses): one row per live clause, the rest in proportion to the list remainders, never more rows
        //      than a list has left, never more than R in all ----
        const uint32_t rem_l = len_l - pos_l;
        const bool live = lane < C && rem_l > 0 && !((dead_m >> lane) & 1u);
        const uint32_t act_m = (uint32_t)__ballot(live);
        if (!act_m) break;
        const int n_act = __popc(act_m);
        const float left_f = live ? (float)rem_l : 0.f;
        float tot = left_f;   // sum over lanes 0..7 (any rounding is fine: it only shapes the window)
        tot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), 0x141, 0xf, 0xf, false));
        tot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), 0x4e, 0xf, 0xf, false));
        tot += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tot), 0xb1, 0xf, 0xf, false));
        tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 0));
        const uint32_t need_l = (rem_l + 63u) >> 6;
        uint32_t rows_l = live ? 1u + (uint32_t)((float)(R - n_act) * (left_f / tot)) : 0u;
        rows_l = rows_l < need_l ? rows_l : need_l;
        // first row of every clause: exclusive prefix sum over lanes 0..7 (row_shr 1, 2, 4 inside the DPP row; zero shifted in)
        uint32_t end_l = rows_l;
        end_l += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end_l, 0x111, 0xf, 0xf, true);
        end_l += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end_l, 0x112, 0xf, 0xf, true);
        end_l += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end_l, 0x114, 0xf, 0xf, true);
        end_l = end_l < (uint32_t)R ? end_l : (uint32_t)R;            // (float rounding can push the shares past R)
        const uint32_t start_l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)end_l, 0x111, 0xf, 0xf, true);   // lane c - 1's end (0 into lane 0)
        rows_l = end_l - start_l;
        const uint32_t n_rows = bu_rl(end_l, 7);
        // ---- row descriptors, lane m = row m: its clause = the number of clauses that end at or before m ----
        uint32_t rc_l = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) rc_l += bu_rl(end_l, c) <= (uint32_t)lane ? 1u : 0u;
        rc_l = rc_l < 8u ? rc_l : 0u;   // rows past n_rows: clause 0, masked out below
        const int src4 = (int)(rc_l << 2);
        const uint32_t r_start = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)start_l);
        const uint32_t r_pos = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)pos_l);
        const uint32_t r_len = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)len_l);
        const uint32_t r_blo = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)(uint32_t)b_l);
        const uint32_t r_bhi = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)(uint32_t)(b_l >> 32));
        const uint32_t r_attr = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)attr_l);
        const uint32_t r_w = (uint32_t)__builtin_amdgcn_ds_bpermute(src4, (int)w_bits_l);
        const bool r_valid = (uint32_t)lane < n_rows;
        const uint32_t r_off = r_pos + 64u * ((uint32_t)lane - r_start);
        const unsigned long long r_base = r_valid ? (((unsigned long long)r_bhi << 32) | r_blo) + r_off : 0ull;
        const uint32_t r_left = r_valid ? r_len - r_off : 0u;   // lanes [0, left) of the row hold postings of the list
        const uint32_t r_first = r_valid && (uint32_t)lane == r_start ? 1u : 0u;
        // ---- ONE round trip: the rows' doc ids and posting words (tf | fieldnorm id << 24), every clause's first doc not loaded ----
        uint32_t p_doc[R], p_w[R];
#pragma unroll
        for (int m = 0; m < R; m++) {
            const unsigned long long base = ((unsigned long long)bu_rl((uint32_t)(r_base >> 32), m) << 32) | bu_rl((uint32_t)r_base, m);
            const uint32_t left = bu_rl(r_left, m);
            const uint32_t *ip = doc_ids + base;   // uniform: the lane index rides in the load's offset register
            const uint32_t *wp = tfs + base;
            const uint32_t d = ip[lane];           // unconditional: the arrays are padded behind the last list
            p_w[m] = wp[lane];
            p_doc[m] = (uint32_t)lane < left ? d : 0xffffffffu;   // never below any window bound
        }
        const uint32_t next_l = pos_l + 64u * rows_l;
        const bool more_l = live && next_l < len_l;
        uint32_t hi_c = doc_ids[more_l ? b_l + next_l : 0ull];
        hi_c = more_l ? hi_c : 0xffffffffu;
        const uint32_t hi_w = bu_min8(lane < C ? hi_c : 0xffffffffu);
        uint32_t hi = hi_w < hi_doc ? hi_w : hi_doc;
        const unsigned long long cw1 = a.dbg ? clock64() + (p_doc[R - 1] & 0u) : 0;   // (the last row has landed)

        // ---- the filter passes; a window with more involved postings than the list holds is cut in half and retried.  The in-window
        //      test is one compare against the registers that hold the rows, so no mask outlives its use; "involved" lives in bit 31
        //      of the row's hash ----
        uint32_t p_h[R];   // hash: bits [4:0] = bit of the word, [13:5] = word of the bitmap; bit 31: involved
#pragma unroll
        for (int m = 0; m < R; m++) p_h[m] = (p_doc[m] ^ (p_doc[m] >> 14)) & 0x3fffu;   // doc ids inside one window differ in their low bits
        for (;;) {
            uint32_t old[R];
            bu_lds_order();
#pragma unroll
            for (int m = 0; m < R; m++)
                old[m] = p_doc[m] < hi ? __hip_atomic_fetch_or(&bm_a[p_h[m] >> 5], 1u << (p_h[m] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
#pragma unroll
            for (int m = 0; m < R; m++) {
                const unsigned long long second = __ballot(__builtin_amdgcn_ubfe(old[m], p_h[m], 1u) != 0u);
                if (second) {
                    if ((second >> lane) & 1ull) __hip_atomic_fetch_or(&bm_b[p_h[m] >> 5], 1u << (p_h[m] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            uint32_t n_flag = 0;
            bu_lds_order();
#pragma unroll
            for (int m = 0; m < R; m++) {
                const uint32_t bw = bm_b[p_h[m] >> 5];
                const uint32_t inv = p_doc[m] < hi ? __builtin_amdgcn_ubfe(bw, p_h[m], 1u) : 0u;
                n_flag += (uint32_t)__popcll(__ballot(inv != 0u));
                p_h[m] |= inv << 31;
            }
            bu_lds_order();
            if (n_flag <= BU_CAP) break;
            // too many: halve the doc range (>= 1 document stays; one document has at most 8 postings), clear, retry
            for (uint32_t i = lane; i < 2 * BU_WORDS / 4; i += 64) list[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int m = 0; m < R; m++) p_h[m] &= 0x3fffu;
            const uint32_t span = hi - win_lo;
            hi = win_lo + (span > 1u ? span / 2u : 1u);
        }
        const unsigned long long cw2 = a.dbg ? clock64() : 0;
        // ---- per row: cursor, score, and either the posting is final or it joins the involved list (which reuses the bitmaps) ----
        uint32_t matched = 0, n_inv = 0;
        bu_lds_order();
#pragma unroll
        for (int m = 0; m < R; m++) {
            const unsigned long long ok_m = __ballot(p_doc[m] < hi);
            const int c = (int)bu_rl(rc_l, m);
            const uint32_t cnt = (uint32_t)__popcll(ok_m);
            pos_l += lane == c ? cnt : 0u;
            postings += cnt;
            // a clause whose first loaded posting is beyond the slice is done with it
            if (multi_slice && bu_rl(r_first, m) && bu_rl(p_doc[m], 0) >= hi_doc) dead_m |= 1u << c;
            if (!ok_m) continue;
            const unsigned long long fl_m = __ballot((int32_t)p_h[m] < 0);
            const uint32_t attr = bu_rl(r_attr, m);
            const uint32_t w_bits = bu_rl(r_w, m);
            const float w = __uint_as_float(w_bits);
            const uint32_t mode = (attr >> 8) & 0xffu;
            const uint32_t fn = p_w[m] >> 24;
            const uint32_t tfi = p_w[m] & 0xffffffu;
            float s;
            if (mode == 2u) s = w;   // ConstScorer(boost)
            else if (mode == 1u || !(__ballot(tfi != 1u) & ok_m)) s = w * inv1[fn];
            else {
                const float tf = (float)tfi;
                s = w * (tf / (tf + tf_cache[fn]));
            }
            if (w_bits >> 31) s = 0.f + s;   // the oracle's sum starts at +0: -0 never leaves it
            if (fl_m) {
                const bool f = (fl_m >> lane) & 1ull;
                const uint32_t at = n_inv + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(fl_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fl_m, 0u));
                if (f) list[at] = make_uint4(p_doc[m], __builtin_bit_cast(uint32_t, s), (uint32_t)c | ((attr & 0xffu) == 2u ? 0x100u : 0u), 0u);
                n_inv += (uint32_t)__popcll(fl_m);
            }
            const unsigned long long clean = ok_m & ~fl_m;
            const bool row_ok = (single_ok_m >> c) & 1u;
            if (!row_ok || !clean) continue;
            if constexpr (EXTRAS) {
                bool ok = (clean >> lane) & 1ull;
                const uint64_t ck = finish(ok, p_doc[m], s);
                matched += (uint32_t)__popcll(__ballot(ok));
                offer(ck);
            } else {
                matched += (uint32_t)__popcll(clean);
                // most rows hold nothing the list wants once it is full: one float compare before any key is built
                if (__ballot(!(s < kth_score)) & clean) {
                    const uint64_t ck = ((clean >> lane) & 1ull) ? rank_key(s, p_doc[m]) : NIDX_EMPTY_KEY;
                    offer(ck);
                }
            }
        }
        const unsigned long long cw3 = a.dbg ? clock64() : 0;
        // ---- the involved postings are in row order (= clause order): resolve them against each other ----
        bu_lds_order();
        if (n_inv) {
            for (uint32_t cb = 0; cb < n_inv; cb += 64) {
                const uint32_t me = cb + (uint32_t)lane;
                const uint4 mine = list[me < n_inv ? me : 0u];
                float acc = 0.f;
                uint32_t mask = 0;
                bool owner = me < n_inv;
                for (uint32_t j = 0; j < n_inv; j++) {
                    const uint4 e = list[j];   // one address for the wave: a broadcast read
                    const bool same = e.x == mine.x;
                    if (same && !(e.z & 0x100u)) acc += __builtin_bit_cast(float, e.y);   // a MustNot clause adds nothing
                    if (same) mask |= 1u << (e.z & 0xffu);
                    if (same && j < me) owner = false;
                }
                bool ok = owner && mask_ok(mask);
                const uint64_t ck = finish(ok, mine.x, acc);
                matched += (uint32_t)__popcll(__ballot(ok));
                offer(ck);
            }
        }
        // both bitmaps (and the list that reused them) back to zero
        bu_lds_order();
        for (uint32_t i = lane; i < 2 * BU_WORDS / 4; i += 64) list[i] = make_uint4(0u, 0u, 0u, 0u);
        if (a.dbg) {
            const unsigned long long cw4 = clock64();
            cy_load += cw1 - cw0, cy_filter += cw2 - cw1, cy_final += cw3 - cw2, cy_resolve += cw4 - cw3;
        }
        total += matched;
        n_win++;
        win_lo = hi;
        if (hi >= hi_doc) break;   // every posting below the slice's upper bound was inside the loaded rows
    }
    if (a.dbg && lane == 0) {
        atomicAdd(&a.dbg[3], clock64() - cy_t0);
        atomicAdd(&a.dbg[6], cy_t0 - cy_entry);
        atomicMax(&a.dbg[7], clock64() - cy_entry);
        atomicAdd(&a.dbg[4], (unsigned long long)n_win);
        atomicAdd(&a.dbg[5], 1ull);
        atomicMax(&a.dbg[8], (unsigned long long)n_win);
        // per-item trace behind the counters: cycles entry -> exit, cycles before the first window, windows, postings
        const unsigned long long t_exit = clock64();
        a.dbg[16 + 8 * (size_t)item + 0] = t_exit - cy_entry;
        a.dbg[16 + 8 * (size_t)item + 1] = cy_t0 - cy_entry;
        a.dbg[16 + 8 * (size_t)item + 2] = n_win | (n_offer << 32);
        a.dbg[16 + 8 * (size_t)item + 3] = ((unsigned long long)postings << 32) | (unsigned long long)(uint32_t)(cy_entry & 0xffffffffu);
        a.dbg[16 + 8 * (size_t)item + 4] = cy_load;
        a.dbg[16 + 8 * (size_t)item + 5] = cy_filter;
        a.dbg[16 + 8 * (size_t)item + 6] = cy_final;
        a.dbg[16 + 8 * (size_t)item + 7] = cy_resolve;
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
    if (lane == 0) {
        a.out_count[item] = cnt;
        a.out_total[item] = total;
        a.out_postings[item] = postings;
    }
}

hipError_t launch_bm25_union(const Bm25Args &a, const uint32_t *items, uint32_t n_items, bool extras, hipStream_t s) {
    if (n_items == 0) return hipSuccess;
    const dim3 grid((n_items + 3) / 4), block(256);
#define NIDX_BU_LAUNCH(KL)                                                                                        \
    do {                                                                                                          \
        if (extras) hipLaunchKernelGGL((bm25_union_kernel<KL, 8, true>), grid, block, 0, s, a, items, n_items);   \
        else hipLaunchKernelGGL((bm25_union_kernel<KL, 8, false>), grid, block, 0, s, a, items, n_items);         \
    } while (0)
    if (a.k > 256) NIDX_BU_LAUNCH(8);
    else if (a.k > 64) NIDX_BU_LAUNCH(4);
    else NIDX_BU_LAUNCH(1);
#undef NIDX_BU_LAUNCH
    return hipGetLastError();
}

}  // namespace nidx
