// bm25_stream.hip — BM25 scoring for term unions whose posting lists rarely meet, term at a time (gfx950).
//
// Same contract as bm25_fast_kernel (bm25.hip) and bm25_union_kernel (bm25_union.hip): one WAVE per work item = (query, doc-id slice),
// a query of <= 8 plain term clauses; it computes what tantivy computes under TextReaderService::do_search / the paragraph
// Searcher::do_search (nidx_text/src/reader.rs:433-435, nidx_paragraph/src/reader.rs:244-348): Bm25Weight::score per posting,
// BooleanQuery sums in clause order, TopDocs (score desc, DocAddress asc), Count.
//
// bm25_union_kernel walks all lists in lockstep doc-id windows of 8 rows held in registers; its bookkeeping (window plan, row
// descriptors, three filter passes per row, one serial list insertion per candidate) costs ~290 instruction issues per 64-posting row,
// and instruction issue is what bounds it.  Here every clause's part of the slice is STREAMED on its own — the
// two bounds of the slice are found by the side-by-side search first, so a clause is a plain counted loop over rows with nothing to plan:
//   phase 1  every clause but the longest: doc ids only; each posting sets its bit of bitmap A (32 Kibit, ds_or_rtn).  A bit that was
//            already set marks a POSSIBLE second posting of the same document: that posting sets the document's bit in bitmap B (2 Kibit).
//   phase 2  the longest clause (it sets nothing — its documents are distinct; its first four rows have been in flight since before phase 1):
//            a posting whose A bit is set is "involved" (sets B,
//            joins the involved list with its score); any other posting is FINAL: score -> one float compare with the k-th score ->
//            (rarely) rank key -> candidate.
//   phase 3  the other clauses again, now with scores: B bit set = involved, otherwise final.
//   phase 4  the involved postings (a few per cent: true meetings + hash collisions) are resolved exactly.  Up to 64 of them sit one
//            per lane in clause order and are compared through the scalar unit (v_readlane): lane i learns its document's f32 sum, built
//            in lane = clause order, and its clause mask; the first lane of a document owns it.  More go through a 512-slot LDS hash
//            table (it reuses the bitmaps' space and the top of the candidate buffer): clause by clause in clause order — inside one clause the documents are distinct, so
//            every lane owns its document's slot for that round — a posting claims or finds its document's slot (ds_cmpst) and adds its
//            score to the slot's f32 sum, which is therefore built in clause order like the oracle's term-at-a-time loop, and ORs its
//            clause into the slot's mask; one pass over the slots then tests every document against the query's boolean structure.
// Both partners of a meeting are always involved: the later of two short-clause postings sees A set in phase 1 and sets B, which both
// read in phase 3; a long-clause posting that meets a short one sees A in phase 2 and sets B before phase 3 reads it.
// Candidates are not inserted one by one: they are appended to a 192-entry LDS buffer (ballot compaction) and merged 64 at a time
// into the sorted list by a bitonic network that runs in the VALU (v_permlane32/16_swap + DPP; compare-exchange = one v_cmp_gt_u64,
// one s_xor with a constant lane mask, two v_cndmask) — ~220 issues per 64 candidates instead of ~50 per candidate.
// An involved list that overflows (192 entries) cuts the slice's doc range in half and retries; what was already offered is offered
// again, so from then on the item inserts serially and ignores keys the list already holds: exact for any input.
// LDS: 6.75 KiB per wave (A 4 KiB, B 256 B, involved list 1.5 KiB, candidates 1 KiB) + 4 KiB per workgroup (the score tables) = 31 KiB:
// 5 workgroups per CU, and the k <= 64 kernel is compiled for five waves per SIMD (<= 96 VGPRs) so that the registers allow them too.
#include "device_common.h"
#include "kernels.h"
#include "wave_bitonic.h"

namespace nidx {

typedef const __attribute__((address_space(4))) uint32_t *bs_cu32_t;
template <typename T>
__device__ inline bs_cu32_t bs_const_words(const T *p) { return (bs_cu32_t)(uintptr_t)p; }
__device__ inline void bs_lds_order() { asm volatile("" ::: "memory"); }

#define BS_A_WORDS 1024u   /* 32 Kibit */
#define BS_B_WORDS 64u     /* 2 Kibit */
#ifndef BS_CAP
#define BS_CAP 192u        /* involved postings per doc range */
#endif
#ifndef BS_QN
#define BS_QN 3            /* rows of the quotient table: frequencies 1 .. BS_QN */
#endif
#ifndef BS_AHEAD
#define BS_AHEAD 1         /* groups of four rows in flight ahead of the one being scored: 1 or 2 */
#endif
#ifndef BS_MIN_WAVES
#define BS_MIN_WAVES 5     /* waves per SIMD the k <= 64 kernel is compiled for: <= 96 VGPRs, so that the 1 061 workgroups of the bench batch are all resident (5 per CU = 1 280 slots; with 4 the last 37 run as a second round: 73 us instead of 57) */
#endif
#ifndef BS_FAST_GROUPS
#define BS_FAST_GROUPS 1   /* four full rows at a time on the bounds-free path */
#endif
#define BS_CAND 128u       /* 63 left over + one row: the buffer is drained below 64 entries after every row */
#define BS_GROUPS 8

__device__ inline uint32_t bs_rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// Bitmap A is a blocked Bloom filter (round 6): a document owns the word h >> 5 (h = the 15-bit fold of its id, as before) and THREE bits inside
// it, taken from the top of a multiplicative hash.  A posting "finds its document in A" when all three are set.  One LDS operation per posting
// as before; at the ~650 - 1 300 marked documents of an item the false-positive rate falls from 2 - 4 % (one bit) to 0.2 - 0.6 %, and false
// positives were 99 % of the involved postings (true meetings: about two per QUERY on the bench mix) and a third of the kernel's time
// (profiles/r06_bm25_breakdown.txt section 8).
__device__ inline uint32_t bs_a_mask(uint32_t d) {
    const uint32_t g = d * 0x9E3779B1u;
    return (1u << (g >> 27)) | (1u << ((g >> 22) & 31u)) | (1u << ((g >> 17) & 31u));
}

// The same network on one f32 per lane (ascending): a compare-exchange is the partner fetch, v_max, v_min and a select.
template <int J>
__device__ inline float bs_cmpx_f32(float v, unsigned long long take_max_mask) {
    const bool tm = __builtin_amdgcn_inverse_ballot_w64(take_max_mask);
    if constexpr (J >= 16) {
        uint32_t a0 = __float_as_uint(v), a1 = a0;
        if constexpr (J == 32) swap_pair32(a0, a1);
        else swap_pair16(a0, a1);
        const float x = __uint_as_float(a0), y = __uint_as_float(a1);   // {own, partner} in some order
        return tm ? fmaxf(x, y) : fminf(x, y);
    } else {
        const float p = __uint_as_float(xor_partner_dpp<J>(__float_as_uint(v)));
        return tm ? fmaxf(v, p) : fminf(v, p);
    }
}
template <int K, int J>
__device__ inline float bs_sort_steps_f32(float v) {
    v = bs_cmpx_f32<J>(v, bs_sort_mask(K, J));
    if constexpr (J > 1) return bs_sort_steps_f32<K, J / 2>(v);
    else return v;
}
template <int K>
__device__ inline float bs_sort_stages_f32(float v) {
    if constexpr (K > 2) v = bs_sort_stages_f32<K / 2>(v);
    return bs_sort_steps_f32<K, K / 2>(v);
}

// ---- the fused merge (kernels.h: Bm25FusedMerge) ----
__device__ inline uint32_t bs_ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long bs_ld_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Cross-workgroup hand-over WITHOUT agent-scope fences.  On this part an agent-scope release is `buffer_wbl2` (the eight XCDs' L2s are not coherent
// with each other: the fence writes a whole L2's dirty lines back) and costs the scoring launch a factor of five (230 us against 43, measured).
// What the fence is for — making earlier NON-atomic stores visible, keeping later NON-atomic loads from stale lines — is not needed when every
// value that crosses workgroups is moved by agent-scope atomic accesses (sc1: written through / read past the non-coherent levels, the LLVM AMDGPU
// memory model's code for monotonic agent-scope stores and loads): the producer's atomic stores only have to be COMPLETE before its arrival is
// (s_waitcnt vmcnt(0): what the workgroup-scope release fence below compiles to, and it pins the compiler's order), the consumer's atomic loads are
// issued after its arrival returned.  bs_st_* / bs_ld_* are those accesses; nothing else crosses.
__device__ inline void bs_st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void bs_st_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one more arrival at `counter`; true for the arrival that completes `n` (it resets the counter: nobody else touches it during this launch)
__device__ inline bool bs_arrive_last(uint32_t *counter, uint32_t n, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's atomic stores have completed before its arrival is issued
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = bs_rl(old, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the others' lists are read after the arrival returned
    if (old + 1u != n) return false;
    if (lane == 0) bs_st_u32(counter, 0u);
    return true;
}
// `mine`: this item's sorted list (lane e = rank e, NIDX_EMPTY_KEY behind its end and at ranks >= k)
__device__ inline void bs_fused_merge(const Bm25Args &a, uint32_t q, uint32_t item, uint32_t slice, uint32_t n_slices, uint64_t mine, unsigned long long total,
                                      unsigned long long postings, int lane) {
    const Bm25FusedMerge &f = a.fm;
    const uint32_t k = a.k;
    const uint32_t w0 = item - slice;   // the query's first work item (its slices are consecutive work items)
    const uint32_t n_groups = (n_slices + BM25_FUSE_GROUP - 1u) / BM25_FUSE_GROUP, g = slice / BM25_FUSE_GROUP, g_first = g * BM25_FUSE_GROUP;
    const uint32_t g_n = n_slices - g_first < BM25_FUSE_GROUP ? n_slices - g_first : BM25_FUSE_GROUP;
    uint32_t *const cnt_q = f.done + (size_t)q * (BM25_FUSE_MAX_GROUPS + 1u);
    const uint32_t rev = 63u - (uint32_t)lane;   // the other lists are read worst first (bs_merge_sorted)
    uint64_t acc = mine;
    if (g_n > 1u) {
        if (!bs_arrive_last(cnt_q + 1u + g, g_n, lane)) return;
#pragma unroll
        for (uint32_t j = 0; j < BM25_FUSE_GROUP; j++) {
            const uint32_t sl = g_first + j;
            if (j >= g_n || sl == slice) continue;
            const uint32_t w = w0 + sl;
            const uint32_t c = bs_ld_u32(a.out_count + w);
            const uint64_t key = rev < (c < k ? c : k) ? bs_ld_u64(a.out_key + (size_t)w * k + rev) : NIDX_EMPTY_KEY;
            total += bs_ld_u64(a.out_total + w);
            postings += bs_ld_u64(a.out_postings + w);
            acc = bs_merge_sorted(acc, key);
        }
    }
    if (n_groups > 1u) {
        // the group's k best go to the query's table of group lists; the last group to arrive merges the table
        const size_t at = (size_t)q * BM25_FUSE_MAX_GROUPS + g;
        const bool valid = acc != NIDX_EMPTY_KEY && (uint32_t)lane < k;
        const uint32_t c_mine = (uint32_t)__popcll(__ballot(valid));
        if ((uint32_t)lane < k) bs_st_u64(f.g_key + at * k + (uint32_t)lane, acc);
        if (lane == 0) {
            bs_st_u32(f.g_count + at, c_mine);
            bs_st_u64(f.g_total + at, total);
            bs_st_u64(f.g_postings + at, postings);
        }
        if (!bs_arrive_last(cnt_q, n_groups, lane)) return;
        acc = valid ? acc : NIDX_EMPTY_KEY;
        for (uint32_t g2 = 0; g2 < n_groups; g2++) {
            if (g2 == g) continue;
            const size_t o = (size_t)q * BM25_FUSE_MAX_GROUPS + g2;
            const uint32_t c = bs_ld_u32(f.g_count + o);
            const uint64_t key = rev < c ? bs_ld_u64(f.g_key + o * k + rev) : NIDX_EMPTY_KEY;
            total += bs_ld_u64(f.g_total + o);
            postings += bs_ld_u64(f.g_postings + o);
            acc = bs_merge_sorted(acc, key);
        }
    }
    if (f.ablate == 3) {
        if (acc == 12345ull && lane == 0) f.out_count[q] = 1;   // (keeps the merge alive)
        return;
    }
    // what bm25_merge_kernel writes (bm25.hip)
    const bool valid = acc != NIDX_EMPTY_KEY && (uint32_t)lane < k;
    const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
    if ((uint32_t)lane < k) {
        uint32_t d = valid ? rank_key_addr(acc) : 0xffffffffu;
        if (f.seg_base) {
            // DocAddress of a resident doc: the last segment whose first doc is <= d (empty segments share their base with the next)
            uint32_t lo = 0, hi = f.n_seg;
            while (valid && hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (f.seg_base[mid] <= d) lo = mid;
                else hi = mid;
            }
            if (valid) d -= f.seg_base[lo];
            f.out_seg[(size_t)q * k + (uint32_t)lane] = lo;
        }
        f.out_doc[(size_t)q * k + (uint32_t)lane] = d;
        f.out_score[(size_t)q * k + (uint32_t)lane] = valid ? rank_key_score(acc) : 0.f;
    }
    if (lane == 0) {
        f.out_count[q] = cnt;
        f.out_total[q] = total;
        f.out_postings[q] = postings;
    }
}

// DBG: the per-item cycle trace of NIDX_GPU_BM25_DEBUG (a.dbg != nullptr) is a separate instantiation: none of its state in the product kernel
template <int KL, bool EXTRAS, bool DBG>
__global__ __launch_bounds__(256, (KL == 1 ? BS_MIN_WAVES : 4)) void bm25_stream_kernel(Bm25Args a, const uint32_t *items, uint32_t n_items) {
    // tf / (tf + K1 * (1 - B + B * fieldnorm / avg)) for tf = 1, 2, 3 and every fieldnorm id: the same two f32 operations as the
    // general form, done once per workgroup — a posting of a short document almost always has one of these frequencies, and the IEEE
    // division is ~16 instructions per row.  Larger frequencies take the division with the table of the index (a.tf_cache, L2-resident)
    __shared__ float quot[BS_QN][256];
    __shared__ float tf_cache_s[256];   // K1 * (1 - B + B * fieldnorm / avg) for the division (from L2 the dependent loads cost 10 us per launch)
#define BS_TFC(fn) tf_cache_s[fn]
    __shared__ uint32_t bm_a_all[4][BS_A_WORDS];
    __shared__ uint32_t bm_b_all[4][BS_B_WORDS];
    __shared__ uint32_t list_doc_all[4][BS_CAP];
    __shared__ uint32_t list_score_all[4][BS_CAP];
    __shared__ uint64_t cand_all[4][BS_CAND];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long cy_entry = DBG ? clock64() : 0;
    uint32_t *bm_a = bm_a_all[wave];
    uint32_t *bm_b = bm_b_all[wave];
    uint32_t *list_doc = list_doc_all[wave];
    uint32_t *list_score = list_score_all[wave];
    uint64_t *cand = cand_all[wave];
    auto clear_bitmaps = [&]() {
        for (uint32_t i = lane; i < BS_A_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(bm_a)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (lane < (int)(BS_B_WORDS / 4)) reinterpret_cast<uint4 *>(bm_b)[lane] = make_uint4(0u, 0u, 0u, 0u);
    };
    {
        // (the index keeps the quotient rows behind the division's table, computed once at open: bm25_index.cpp)
        static_assert(BS_QN <= 3, "the index stores the quotient rows of tf = 1 .. 3");
        const float c = a.tf_cache[threadIdx.x];
        float qv[BS_QN];
#pragma unroll
        for (int t = 0; t < BS_QN; t++) qv[t] = a.tf_cache[256 * (t + 1) + threadIdx.x];
        clear_bitmaps();
#pragma unroll
        for (int t = 0; t < BS_QN; t++) quot[t][threadIdx.x] = qv[t];
        tf_cache_s[threadIdx.x] = c;
    }
    __syncthreads();   // the only workgroup barrier
    const uint32_t slot_in_grid = blockIdx.x * 4u + (uint32_t)wave;
    if (slot_in_grid >= n_items) return;
    // ---- the item record and the query's clause table ----
    const uint32_t item = bs_const_words(items)[slot_in_grid];
    bs_cu32_t wrec = bs_const_words(a.work) + (size_t)item * 5u;
    const uint32_t q = wrec[0], slice = wrec[1], n_slices = wrec[2], clause_first = wrec[3];
    const int C = (int)wrec[4];   // <= 8
    const int k = (int)a.k;
    const uint32_t *const doc_ids = a.doc_ids;
    const uint32_t *const tfs = a.tfs;

    // ---- lane c holds clause c ----
    uint32_t len_l = 0, attr_l = 0, w_bits_l = 0, floor_bits_l = 0xff800000u;
    unsigned long long b_l = 0;
    if (lane < C) {
        const Bm25UClause uc = a.uclauses[clause_first + lane];
        b_l = ((unsigned long long)uc.b_hi << 32) | uc.b_lo;
        len_l = uc.len;
        attr_l = uc.attr;
        w_bits_l = __float_as_uint(uc.weight);
        floor_bits_l = uc.floor_bits;
    }
    // A score at least k documents of the QUERY are known to reach (bm25_index.cpp, from the per-term floors of bm25_aux.hip; -inf = none): a final
    // posting below it is not among the k best of the query, let alone of this item, so it never becomes a candidate.  Postings AT the floor stay.
    const float q_floor = EXTRAS ? -INFINITY : __uint_as_float(bs_rl(floor_bits_l, 0));
    const uint32_t occur_l = attr_l & 0xff;
    // the boolean structure as clause masks
    const uint32_t must_m = (uint32_t)__ballot(lane < C && occur_l == 1), not_m = (uint32_t)__ballot(lane < C && occur_l == 2),
                   should_m = (uint32_t)__ballot(lane < C && occur_l == 0);
    uint32_t group_l = 0;   // lane g: the clause mask of the g-th non-empty required Should group
    int n_groups = 0;
#pragma unroll
    for (int g = 0; g < BS_GROUPS; g++) {
        const uint32_t gm = (uint32_t)__ballot(lane < C && occur_l == 3u + (uint32_t)g);
        if (gm) {
            if (lane == n_groups) group_l = gm;
            n_groups++;
        }
    }
    auto mask_ok = [&](uint32_t m) -> bool {   // BooleanQuery: every Must, no MustNot, one clause of every required Should group
        const bool any_required = must_m != 0 || n_groups > 0;
        bool ok = (m & must_m) == must_m && (m & not_m) == 0 && (any_required || (m & should_m) != 0);
        for (int g = 0; g < n_groups; g++)
            if ((m & bs_rl(group_l, g)) == 0) ok = false;
        return ok;
    };
    // bit c: a document that occurs in clause c ONLY matches the query
    const uint32_t single_ok_m = (uint32_t)__ballot(lane < C && mask_ok(1u << lane));

    // ---- per clause (lane c < C): the first position in [left0, right0) whose doc id is >= bound; the clauses search side by side,
    //      a lane group each (bm25.hip) ----
    const int g_log = C <= 1 ? 6 : C <= 2 ? 5 : C <= 4 ? 4 : 3;
    // two bounds at once (the slice's lower and upper end): their probes travel together, one round trip per step for both
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
                step[t] = (right[t] - left[t] + G - 1u) >> g_log;   // (G = 1 << g_log)
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

    // ---- the slice: doc range and the clauses' posting ranges ----
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

    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
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

    // ---- candidates ----
    uint32_t n_cand = 0, n_flush = 0;
    bool redo = false;   // a doc range was retried: keys may be offered twice from here on
    auto flush64 = [&]() {   // the first min(n_cand, 64) buffered candidates into the list (KL == 1)
        bs_lds_order();
        const uint64_t v = (uint32_t)lane < n_cand ? cand[lane] : NIDX_EMPTY_KEY;
        const uint32_t rest = n_cand > 64u ? n_cand - 64u : 0u;
        bs_lds_order();
        for (uint32_t o = 0; o < rest; o += 64u) {   // (rest < 128)
            const uint64_t t = cand[64u + o + (o + (uint32_t)lane < rest ? (uint32_t)lane : 0u)];
            bs_lds_order();
            if (o + (uint32_t)lane < rest) cand[o + (uint32_t)lane] = t;
            bs_lds_order();
        }
        n_cand = rest;
        top.l[0].key = bs_merge64(top.l[0].key, v);
        kth = top.at(k - 1);
        n_flush++;
    };
    auto offer = [&](uint64_t ck, bool want) {   // the candidates of one row
        unsigned long long mm = __ballot(want && ck > kth);
        if (!mm) return;
        if (KL == 1 && !redo) {
            const uint32_t at = n_cand + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            if ((mm >> lane) & 1ull) cand[at] = ck;
            n_cand += (uint32_t)__popcll(mm);   // (the caller drains the buffer below 64 entries after every two rows)
            return;
        }
        while (mm) {
            const int src = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const uint64_t nk = lane_bcast_u64(ck, src);
            if (!(nk > kth)) continue;
            if (redo) {   // already in the list?
                bool have = false;
#pragma unroll
                for (int i = 0; i < KL; i++) have = have || top.l[i].key == nk;
                if (__ballot(have)) continue;
            }
            kth = top.insert_kth(nk, k, lane);
        }
    };

    // A bar under the candidates before the list has one: the first group of final postings of an item leaves every lane the best
    // score it saw; the k-th largest of those 64 maxima is the k-th best of 64 real documents, hence never above the k-th best of
    // the item — one f32 sort (~100 issues) instead of the three or four 64-key merges the list otherwise needs to climb there
    // (every posting passes while it is empty, and a bar that is only refreshed by a merge lags behind the stream).
    float bar = q_floor;
    bool bar_set = false;
    uint32_t postings = 0, total = 0, n_ranges = 0;
    uint32_t cur_lo = lo_doc, cur_hi = hi_doc;
    uint32_t e_l = item_e_l;
    const unsigned long long cy_t0 = DBG ? clock64() : 0;
    unsigned long long cy_p1 = 0, cy_p2 = 0, cy_p3 = 0, cy_p4 = 0;
    bool dirty = false;   // the bitmaps hold bits
    for (;;) {
        const uint32_t n_l = lane < C ? e_l - s_l : 0u;
        const uint32_t act_m = (uint32_t)__ballot(n_l > 0u);
        if (act_m) {
            if (dirty) {
                bs_lds_order();
                clear_bitmaps();
                bs_lds_order();
                dirty = false;
            }
            const unsigned long long cw0 = DBG ? clock64() : 0;
            // the longest clause (lowest index on ties)
            uint32_t best = n_l;
            best = wave_reduce_u32(best, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            const int L = __ffsll((long long)__ballot(lane < C && n_l == best)) - 1;
            const bool probe = (act_m & (act_m - 1u)) != 0u;   // more than one clause has postings here
            uint32_t dn[4], wn[4];   // the four rows in flight
            {
                const unsigned long long base = ((unsigned long long)bs_rl((uint32_t)(b_l >> 32), L) << 32) | bs_rl((uint32_t)b_l, L);
                const uint32_t s = bs_rl(s_l, L);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    dn[r] = (doc_ids + base + s)[64u * r + (uint32_t)lane];
                    wn[r] = (tfs + base + s)[64u * r + (uint32_t)lane];
                }
            }
            uint32_t matched = 0, posted = 0, n_short = 0, n_long = 0;
            uint32_t run_lo_l = 0, run_hi_l = 0;   // lane c: clause c's run of the involved list
            bool overflow = false;
            // ---- phase 1: mark ----
            if (probe) {
                dirty = true;
                for (uint32_t cm = act_m & ~(1u << L); cm; cm &= cm - 1u) {
                    const int c = __ffs((int)cm) - 1;
                    const unsigned long long base = ((unsigned long long)bs_rl((uint32_t)(b_l >> 32), c) << 32) | bs_rl((uint32_t)b_l, c);
                    const uint32_t s = bs_rl(s_l, c), e = bs_rl(e_l, c);
                    const uint32_t *ip = doc_ids + base;
                    uint32_t dnx[4];   // the next group travels while this one is marked (a wave alone waits ~3 000 cycles for every load)
#pragma unroll
                    for (int r = 0; r < 4; r++) dnx[r] = (ip + s)[64u * r + (uint32_t)lane];   // (the arrays are padded behind the last list)
                    for (uint32_t p = s; p < e; p += 256u) {
                        uint32_t d[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) d[r] = dnx[r];
                        if (p + 256u < e) {
#pragma unroll
                            for (int r = 0; r < 4; r++) dnx[r] = (ip + p)[256u + 64u * r + (uint32_t)lane];
                        }
                        if (BS_FAST_GROUPS) {
                            // a group of up to four rows, the lanes behind the clause's end switched off: the four ds_or_rtn go out back to
                            // back (LDS operations of a wave execute in order: row r still sees the bits of the rows before it) and a
                            // second posting of a document is looked for once per group
                            const uint32_t rem = e - p;
                            uint32_t h[4], old[4], am[4];
                            bool in[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                in[r] = (uint32_t)lane + 64u * r < rem;
                                h[r] = (d[r] ^ (d[r] >> 15)) & 0x7fffu;
                                old[r] = 0u;
                                am[r] = bs_a_mask(d[r]);
                                if (in[r]) old[r] = __hip_atomic_fetch_or(&bm_a[h[r] >> 5], am[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            bool hit[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) hit[r] = in[r] && (old[r] & am[r]) == am[r];
                            if (__ballot(hit[0] || hit[1] || hit[2] || hit[3])) {
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    if (hit[r]) __hip_atomic_fetch_or(&bm_b[(h[r] & 0x7ffu) >> 5], 1u << (h[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            continue;
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (p + 64u * r >= e) break;
                            const bool in = (uint32_t)lane < e - (p + 64u * r);
                            const uint32_t h = (d[r] ^ (d[r] >> 15)) & 0x7fffu;
                            uint32_t old = 0;
                            const uint32_t am = bs_a_mask(d[r]);
                            if (in) old = __hip_atomic_fetch_or(&bm_a[h >> 5], am, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const bool hit = in && (old & am) == am;
                            if (__ballot(hit)) {
                                if (hit) __hip_atomic_fetch_or(&bm_b[(h & 0x7ffu) >> 5], 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                }
                bs_lds_order();
            }
            const unsigned long long cw1 = DBG ? clock64() : 0;
            unsigned long long cw2 = cw1;
            // ---- phases 2 and 3: the longest clause, then the others in clause order ----
            uint32_t todo_m = probe ? (act_m & ~(1u << L)) : 0u;
            for (int step = 0;; step++) {
                int c;
                if (step == 0) c = L;
                else {
                    if (!todo_m || overflow) break;
                    c = __ffs((int)todo_m) - 1;
                    todo_m &= todo_m - 1u;
                }
                if (DBG && step == 1) cw2 = clock64();
                const bool is_long = step == 0;
                const unsigned long long base = ((unsigned long long)bs_rl((uint32_t)(b_l >> 32), c) << 32) | bs_rl((uint32_t)b_l, c);
                const uint32_t s = bs_rl(s_l, c), e = bs_rl(e_l, c);
                const uint32_t attr = bs_rl(attr_l, c), w_bits = bs_rl(w_bits_l, c);
                const float wgt = __uint_as_float(w_bits);
                const uint32_t mode = (attr >> 8) & 0xffu;
                const bool row_ok = (single_ok_m >> c) & 1u;
                const uint32_t *ip = doc_ids + base;
                const uint32_t *wp = tfs + base;
                const uint32_t run_lo = n_short;
                posted += e - s;
                if (!is_long) {   // (L's first rows have been in flight since before phase 1)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        dn[r] = (ip + s)[64u * r + (uint32_t)lane];   // (the arrays are padded behind the last list)
                        wn[r] = (wp + s)[64u * r + (uint32_t)lane];
                    }
                }
#if BS_AHEAD == 2
                uint32_t dnn[4] = {0u, 0u, 0u, 0u}, wnn[4] = {0u, 0u, 0u, 0u};   // the group after the next one
                if (s + 256u < e) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        dnn[r] = (ip + s)[256u + 64u * r + (uint32_t)lane];
                        wnn[r] = (wp + s)[256u + 64u * r + (uint32_t)lane];
                    }
                }
#endif
                for (uint32_t p = s; p < e && !overflow; p += 256u) {
                    uint32_t d[4], w[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) d[r] = dn[r], w[r] = wn[r];
#if BS_AHEAD == 2
                    // a single wave keeps little in flight (one group = 2 KiB): two groups travel while one is scored
#pragma unroll
                    for (int r = 0; r < 4; r++) dn[r] = dnn[r], wn[r] = wnn[r];
                    if (p + 512u < e) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            dnn[r] = (ip + p)[512u + 64u * r + (uint32_t)lane];
                            wnn[r] = (wp + p)[512u + 64u * r + (uint32_t)lane];
                        }
                    }
#else
                    if (p + 256u < e) {   // the next four rows travel while these are scored
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            dn[r] = (ip + p)[256u + 64u * r + (uint32_t)lane];
                            wn[r] = (wp + p)[256u + 64u * r + (uint32_t)lane];
                        }
                    }
#endif
                    if (BS_FAST_GROUPS && !EXTRAS) {
                        // ---- a group of up to four rows, the lanes behind the clause's end switched off: the four bitmap probes and the
                        // four score look-ups travel together, and the rare events — an involved posting, a candidate for the list — are
                        // looked for once per group instead of once per row ----
                        const uint32_t rem = e - p;
                        uint32_t h[4], bw[4] = {0u, 0u, 0u, 0u};
                        bool in[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            in[r] = (uint32_t)lane + 64u * r < rem;
                            h[r] = (d[r] ^ (d[r] >> 15)) & 0x7fffu;
                            if (probe) bw[r] = is_long ? bm_a[h[r] >> 5] : bm_b[(h[r] & 0x7ffu) >> 5];
                        }
                        float sc[4];
                        if (mode == 2u) {   // ConstScorer(boost)
#pragma unroll
                            for (int r = 0; r < 4; r++) sc[r] = wgt;
                        } else {
                            // every frequency of the group in 1 .. BS_QN (the usual case): the quotient comes from the table
                            uint32_t t[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) t[r] = in[r] ? (w[r] & 0xffffffu) - 1u : 0u;   // (tf == 0 wraps to 2^24 - 1 and takes the division)
                            if (mode == 1u) {
#pragma unroll
                                for (int r = 0; r < 4; r++) sc[r] = wgt * quot[0][w[r] >> 24];
                            } else if (!__ballot((t[0] | t[1] | t[2] | t[3]) > (uint32_t)(BS_QN - 1))) {
#pragma unroll
                                for (int r = 0; r < 4; r++) sc[r] = wgt * (&quot[0][0])[(t[r] << 8) + (w[r] >> 24)];
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const float tf = (float)(w[r] & 0xffffffu);
                                    sc[r] = wgt * (tf / (tf + BS_TFC(w[r] >> 24)));
                                }
                            }
                        }
                        if (w_bits >> 31) {   // the oracle's sum starts at +0: -0 never leaves it
#pragma unroll
                            for (int r = 0; r < 4; r++) sc[r] = 0.f + sc[r];
                        }
                        bool inv[4] = {false, false, false, false};
                        uint32_t n_final = rem < 256u ? rem : 256u;
                        if (probe) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (is_long) {
                                    const uint32_t am = bs_a_mask(d[r]);
                                    inv[r] = in[r] && (bw[r] & am) == am;
                                } else {
                                    inv[r] = in[r] && __builtin_amdgcn_ubfe(bw[r], h[r], 1u) != 0u;
                                }
                            }
#ifdef BS_ABLATE_NOINV   /* measurement only (wrong results): no posting is ever involved — the upper bound of what a sharper filter could save */
#pragma unroll
                            for (int r = 0; r < 4; r++) inv[r] = false;
#endif
                            if (__ballot(inv[0] || inv[1] || inv[2] || inv[3])) {
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const unsigned long long inv_m = __ballot(inv[r]);
                                    if (!inv_m || overflow) continue;
                                    const uint32_t n_new = (uint32_t)__popcll(inv_m);
                                    if (n_short + n_long + n_new > BS_CAP) {
                                        overflow = true;
                                        continue;
                                    }
                                    const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(inv_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)inv_m, 0u));
                                    const uint32_t at = is_long ? BS_CAP - 1u - (n_long + rank) : n_short + rank;   // L's entries from the top down
                                    if (inv[r]) {
                                        if (is_long) __hip_atomic_fetch_or(&bm_b[(h[r] & 0x7ffu) >> 5], 1u << (h[r] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        list_doc[at] = d[r];
                                        list_score[at] = __float_as_uint(sc[r]);
                                    }
                                    if (is_long) n_long += n_new;
                                    else n_short += n_new;
                                    n_final -= n_new;
                                }
                                if (overflow) break;   // (the range is cut in half and retried: nothing of this group counts)
                            }
                        }
                        if (row_ok) {
                            matched += n_final;
                            // most groups hold nothing the list wants once it is full: one float compare per posting before any key is built
                            // (the k-th score is NaN while the list is not full: !(s < NaN) lets every score through to the exact test)
                            if (KL == 1 && !bar_set) {
                                if (!(q_floor > -INFINITY)) {   // (a query with a floor starts above anything the first group could tell)
                                    float mx = -INFINITY;
#pragma unroll
                                    for (int r = 0; r < 4; r++)
                                        if (in[r] && !inv[r]) mx = fmaxf(mx, sc[r]);
                                    bar = lane_bcast_f32(bs_sort_stages_f32<64>(mx), 64 - k);   // -inf while fewer than k lanes saw a final posting
                                }
                                bar_set = true;
                            }
                            const float kf = fmaxf(bar, rank_key_score(kth));   // (NaN while the list is not full: fmaxf keeps the bar)
                            bool cnd[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) cnd[r] = in[r] && !inv[r] && !(sc[r] < kf);
                            if (__ballot(cnd[0] || cnd[1] || cnd[2] || cnd[3])) {
                                const uint32_t flushes = n_flush;
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    if (r > 0 && n_flush != flushes) {   // the bar rose under the rows before: this one faces the new one
                                        const float kf2 = fmaxf(bar, rank_key_score(kth));
                                        cnd[r] = cnd[r] && !(sc[r] < kf2);
                                    }
                                    if (__ballot(cnd[r])) offer(rank_key(sc[r], d[r]), cnd[r]);
                                    if (KL == 1) {   // the buffer holds what one row can add on top of 63 left-overs
                                        while (n_cand >= 64u) flush64();
                                    }
                                }
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (p + 64u * r >= e) break;
                        const bool in = (uint32_t)lane < e - (p + 64u * r);
                        bool inv = false;
                        const uint32_t h = (d[r] ^ (d[r] >> 15)) & 0x7fffu;
                        if (probe) {
                            const uint32_t bw = is_long ? bm_a[h >> 5] : bm_b[(h & 0x7ffu) >> 5];
                            if (is_long) {
                                const uint32_t am = bs_a_mask(d[r]);
                                inv = in && (bw & am) == am;
                            } else {
                                inv = in && __builtin_amdgcn_ubfe(bw, h, 1u) != 0u;
                            }
                        }
                        const uint32_t fn = w[r] >> 24;
                        const uint32_t tfi = w[r] & 0xffffffu;
                        float sc;
                        if (mode == 2u) sc = wgt;   // ConstScorer(boost)
                        else if (mode == 1u) sc = wgt * quot[0][fn];
                        else if (!__ballot(in && tfi - 1u > (uint32_t)(BS_QN - 1))) sc = wgt * (&quot[0][0])[(in ? (tfi - 1u) << 8 : 0u) + fn];   // tf in 1 .. BS_QN
                        else {
                            const float tf = (float)tfi;
                            sc = wgt * (tf / (tf + BS_TFC(fn)));
                        }
                        if (w_bits >> 31) sc = 0.f + sc;   // the oracle's sum starts at +0: -0 never leaves it
                        const unsigned long long inv_m = __ballot(inv);
                        if (inv_m) {
                            const uint32_t n_new = (uint32_t)__popcll(inv_m);
                            if (n_short + n_long + n_new > BS_CAP) {
                                overflow = true;
                                break;
                            }
                            const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(inv_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)inv_m, 0u));
                            const uint32_t at = is_long ? BS_CAP - 1u - (n_long + rank) : n_short + rank;   // L's entries from the top down
                            if (inv) {
                                if (is_long) __hip_atomic_fetch_or(&bm_b[(h & 0x7ffu) >> 5], 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                list_doc[at] = d[r];
                                list_score[at] = __float_as_uint(sc);
                            }
                            if (is_long) n_long += n_new;
                            else n_short += n_new;
                        }
                        if (!row_ok) continue;
                        bool ok = in && !inv;
                        if constexpr (EXTRAS) {
                            const uint64_t ck = finish(ok, d[r], sc);
                            matched += (uint32_t)__popcll(__ballot(ok));
                            offer(ck, ok);
                        } else {
                            matched += (uint32_t)__popcll(__ballot(ok));
                            // most rows hold nothing the list wants once it is full: one float compare before any key is built (the k-th
                            // score is NaN while the list is not full: !(s < NaN) lets every score through to the exact test)
                            if (__ballot(ok && !(sc < rank_key_score(kth)))) offer(rank_key(sc, d[r]), ok);
                        }
                        if (KL == 1) {   // the buffer holds what one row can add on top of 63 left-overs
                            while (n_cand >= 64u) flush64();
                        }
                    }
                    if (KL == 1) {   // (also behind a last group of one or three rows)
                        while (n_cand >= 64u) flush64();
                    }
                }
                if (!is_long && lane == c) run_lo_l = run_lo, run_hi_l = n_short;
                if (is_long) bs_lds_order();   // B is complete before phase 3 reads it
            }
            const unsigned long long cw3 = DBG ? clock64() : 0;
            if (overflow) {
                // halve the doc range (>= 1 document stays: one document has at most 8 postings) and retry it; the candidates offered so
                // far are all final ones (the involved postings were not resolved yet), they will be offered again
                if (KL == 1) {
                    while (n_cand) flush64();
                }
                redo = true;
                const uint32_t span = cur_hi - cur_lo;
                cur_hi = cur_lo + (span > 1u ? span / 2u : 1u);
                {
                    uint32_t p0, p1;
                    first_ge2(cur_hi, 0u, true, false, s_l, e_l, p0, p1);
                    e_l = p0;
                }
                n_ranges++;
                continue;
            }
            // ---- phase 4: the involved postings, clause by clause, through a hash table over the bitmaps' space ----
            bs_lds_order();
            const uint32_t n_inv = n_short + n_long;
            if (n_inv && n_inv <= 64u) {
                // few involved postings (the usual case): one per lane, in clause order, compared through the scalar unit — lane i learns
                // its document's sum (built in lane = clause order) and mask from lanes 0 .. n_inv-1; the first lane of a document owns it
                uint32_t src = 0, my_c = 0, off = 0;
                for (int c = 0; c < C; c++) {
                    const bool is_l = c == L;
                    const uint32_t lo = is_l ? 0u : bs_rl(run_lo_l, c), len = is_l ? n_long : bs_rl(run_hi_l, c) - lo;
                    const uint32_t i = (uint32_t)lane - off;
                    if ((uint32_t)lane >= off && i < len) {
                        src = is_l ? BS_CAP - 1u - i : lo + i;
                        my_c = (uint32_t)c;
                    }
                    off += len;
                }
                const bool live = (uint32_t)lane < n_inv;
                const uint32_t my_doc = list_doc[live ? src : 0u];
                const uint32_t my_sc = list_score[live ? src : 0u];
                const uint32_t my_adds = (__builtin_amdgcn_ds_bpermute((int)(my_c << 2), (int)attr_l) & 0xff) != 2 ? 1u : 0u;   // a MustNot clause adds nothing
                // Who shares my document?  Most involved postings are hash collisions of the bitmaps, alone with their document: instead of
                // comparing every lane with every other (n_inv scalar round trips), the lanes drop their bit into one of 64 buckets by a
                // hash of the document (the bitmaps' space is free now) and a lane only looks at its bucket mates — in lane order, which
                // is clause order, so the f32 sum is built exactly as before.
                unsigned long long *bucket = reinterpret_cast<unsigned long long *>(bm_a);
                dirty = true;
                bucket[lane] = 0ull;
                bs_lds_order();
                const uint32_t hb = (my_doc * 2654435761u) >> 26;
                if (live) __hip_atomic_fetch_or(&bucket[hb], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                bs_lds_order();
                unsigned long long mates = live ? bucket[hb] : 0ull;   // (my own bit included)
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
                bool ok = owner && mask_ok(mask);
                const uint64_t ck = finish(ok, my_doc, acc);
                matched += (uint32_t)__popcll(__ballot(ok));
                offer(ck, ok);
                if (KL == 1) {
                    while (n_cand >= 64u) flush64();
                }
            } else if (n_inv) {
                // 256 slots for at most BS_CAP = 192 documents: the documents, their sums and their clause masks all live in the bitmaps' space
                uint32_t *t_doc = bm_a, *t_acc = bm_a + 256;
                uint8_t *t_mask = reinterpret_cast<uint8_t *>(bm_a + 512);
                dirty = true;
                reinterpret_cast<uint4 *>(t_doc)[lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (lane < 16) reinterpret_cast<uint4 *>(t_mask)[lane] = make_uint4(0u, 0u, 0u, 0u);
                bs_lds_order();
                for (int c = 0; c < C; c++) {
                    const bool is_l = c == L;
                    const uint32_t lo = is_l ? 0u : bs_rl(run_lo_l, c), hi = is_l ? n_long : bs_rl(run_hi_l, c);
                    const bool adds = (bs_rl(attr_l, c) & 0xffu) != 2u;   // a MustNot clause adds nothing to the sum
                    for (uint32_t b0 = lo; b0 < hi; b0 += 64) {
                        const uint32_t i = b0 + (uint32_t)lane;
                        const bool live = i < hi;
                        const uint32_t st = live ? (is_l ? BS_CAP - 1u - i : i) : 0u;
                        const uint32_t doc = list_doc[st];
                        const float sc = __uint_as_float(list_score[st]);
                        uint32_t slot = (doc * 2654435761u) >> 24;
                        bool pending = live, fresh = false;
                        while (__ballot(pending)) {
                            if (pending) {
                                uint32_t cur = t_doc[slot];
                                if (cur == ~0u) {
                                    uint32_t expected = ~0u;
                                    if (__hip_atomic_compare_exchange_strong(&t_doc[slot], &expected, doc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                                        fresh = true;
                                        cur = doc;
                                    } else cur = expected;
                                }
                                if (cur == doc) pending = false;
                                else slot = (slot + 1u) & 255u;
                            }
                        }
                        if (live) {
                            float acc = fresh ? 0.f : __uint_as_float(t_acc[slot]);
                            if (adds) acc += sc;
                            t_acc[slot] = __float_as_uint(acc);
                            t_mask[slot] = (uint8_t)(t_mask[slot] | (1u << c));
                        }
                        bs_lds_order();
                    }
                }
                for (uint32_t b0 = 0; b0 < 256u; b0 += 64) {
                    const uint32_t slot = b0 + (uint32_t)lane;
                    const uint32_t doc = t_doc[slot];
                    bool ok = doc != ~0u && mask_ok((uint32_t)t_mask[slot]);
                    if (!__ballot(ok)) continue;
                    const uint64_t ck = finish(ok, doc, __uint_as_float(t_acc[slot]));
                    matched += (uint32_t)__popcll(__ballot(ok));
                    offer(ck, ok);
                    if (KL == 1) {
                        while (n_cand >= 64u) flush64();
                    }
                }
                bs_lds_order();
            }
            if constexpr (DBG) {
                const unsigned long long cw4 = clock64();
                cy_p1 += cw1 - cw0, cy_p2 += cw2 - cw1, cy_p3 += cw3 - cw2, cy_p4 += cw4 - cw3;
            }
            total += matched;
            postings += posted;
            n_ranges++;
        }
        if (cur_hi >= hi_doc) break;
        cur_lo = cur_hi;
        cur_hi = hi_doc;
        s_l = e_l;
        e_l = item_e_l;
    }
    if (KL == 1) {
        while (n_cand) flush64();
    }
    if (DBG && lane == 0) {
        // (no shared counters here: thousands of atomics on one cache line stall the memory channel that owns it for everyone)
        // per-item trace: cycles entry -> exit, cycles before the first range, ranges | flushes, postings, cycles of the four phases
        const unsigned long long t_exit = clock64();
        a.dbg[16 + 8 * (size_t)item + 0] = t_exit - cy_entry;
        a.dbg[16 + 8 * (size_t)item + 1] = cy_t0 - cy_entry;
        a.dbg[16 + 8 * (size_t)item + 2] = n_ranges | ((unsigned long long)n_flush << 32);
        a.dbg[16 + 8 * (size_t)item + 3] = ((unsigned long long)postings << 32) | (unsigned long long)(uint32_t)(cy_entry & 0xffffffffu);
        a.dbg[16 + 8 * (size_t)item + 4] = cy_p1;
        a.dbg[16 + 8 * (size_t)item + 5] = cy_p2;
        a.dbg[16 + 8 * (size_t)item + 6] = cy_p3;
        a.dbg[16 + 8 * (size_t)item + 7] = cy_p4;
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) {
            if (KL == 1 && a.fm.done) bs_st_u64(a.out_key + (size_t)item * k + e, valid ? key : NIDX_EMPTY_KEY);   // (read by another workgroup of this launch)
            else a.out_key[(size_t)item * k + e] = valid ? key : NIDX_EMPTY_KEY;
        }
    }
    if (lane == 0) {
        if (KL == 1 && a.fm.done) {
            bs_st_u32(a.out_count + item, cnt);
            bs_st_u64(a.out_total + item, total);
            bs_st_u64(a.out_postings + item, postings);
        } else {
            a.out_count[item] = cnt;
            a.out_total[item] = total;
            a.out_postings[item] = postings;
        }
    }
    if constexpr (KL == 1) {
        if (a.fm.done) {
            const uint64_t key0 = top.mine(0);
            bs_fused_merge(a, q, item, slice, n_slices, (key0 != NIDX_EMPTY_KEY && lane < k) ? key0 : NIDX_EMPTY_KEY, total, postings, lane);
        }
    }
}

hipError_t launch_bm25_stream(const Bm25Args &a, const uint32_t *items, uint32_t n_items, bool extras, hipStream_t s) {
    if (n_items == 0) return hipSuccess;
    const dim3 grid((n_items + 3) / 4), block(256);
#define NIDX_BS_LAUNCH2(KL, EX)                                                                                        \
    do {                                                                                                            \
        if (a.dbg) hipLaunchKernelGGL((bm25_stream_kernel<KL, EX, true>), grid, block, 0, s, a, items, n_items);    \
        else hipLaunchKernelGGL((bm25_stream_kernel<KL, EX, false>), grid, block, 0, s, a, items, n_items);         \
    } while (0)
#define NIDX_BS_LAUNCH(KL)                  \
    do {                                    \
        if (extras) NIDX_BS_LAUNCH2(KL, true); \
        else NIDX_BS_LAUNCH2(KL, false);    \
    } while (0)
    if (a.k > 256) NIDX_BS_LAUNCH(8);
    else if (a.k > 64) NIDX_BS_LAUNCH(4);
    else NIDX_BS_LAUNCH(1);
#undef NIDX_BS_LAUNCH
#undef NIDX_BS_LAUNCH2
    return hipGetLastError();
}

}  // namespace nidx
