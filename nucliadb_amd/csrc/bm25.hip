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
#include "device_common.h"
#include "kernels.h"

namespace nidx {

#define BM25_EMPTY 0xffffffffu

// =====================================================================================================
// One WAVE per work item, clause state in lane registers (lane c = clause c): no block barrier, no LDS traffic for
// cursors / shares / clause attributes; the window's hash table, the 1 KiB tf cache and 64 counters are all the
// LDS it uses (~11 KiB => up to 14 items per CU).
// =====================================================================================================
#define BW_TABLE 1024      /* hash slots */
#define BW_WINDOW 512      /* postings per window: 8 per lane */
#define BW_SHIFT 22        /* 32 - log2(BW_TABLE) */

// wave reductions on the swap + DPP levels of device_common.h (no ds_bpermute round trips)
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
    return wave_reduce_u64((uint64_t)v, [](uint64_t a, uint64_t b) { return a + b; });
}
__device__ inline uint32_t wave_min_u32(uint32_t v) {
    return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}
__device__ inline uint32_t rl_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

template <int KL>
__global__ __launch_bounds__(64) void bm25_wave_kernel(Bm25Args a) {
    __shared__ uint32_t t_key[BW_TABLE];
    __shared__ float t_acc[BW_TABLE];
    __shared__ uint16_t t_flags[BW_TABLE];  // bit0 should-hit, bit1 excluded, bit2 group-hit, bits 8.. must count
    __shared__ float tf_cache[256];
    __shared__ uint32_t taken[BM25_MAX_CLAUSES];
    const int lane = threadIdx.x;
    const Bm25Work work = a.work[blockIdx.x];
    const uint32_t q = work.query;
    const uint64_t c0 = a.clause_offsets[q], c1 = a.clause_offsets[q + 1];
    const int C = (int)(c1 - c0);
    const int k = (int)a.k;

    for (int i = lane; i < BW_TABLE; i += 64) {
        t_key[i] = BM25_EMPTY;
        t_acc[i] = 0.f;
        t_flags[i] = 0;
    }
    for (int i = lane; i < 256; i += 64) tf_cache[i] = a.tf_cache[i];
    taken[lane] = 0;

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
    const int n_must = __popcll(__ballot(lane < C && (attr_l & 0xff) == 1));
    const int n_group = __popcll(__ballot(lane < C && (attr_l & 0xff) == 3));
    if (work.n_slices > 1) {
        // doc range of this slice: the wave finds, clause by clause, the first posting >= lo and >= hi (64-ary search)
        const uint32_t lo_doc = (uint32_t)((unsigned long long)a.n_docs * work.slice / work.n_slices);
        const uint32_t hi_doc = (uint32_t)((unsigned long long)a.n_docs * (work.slice + 1) / work.n_slices);
        for (int c = 0; c < C; c++) {
            const bool aux = (rl_u32(attr_l, c) >> 16) != 0;
            const uint32_t *ids = aux ? a.aux_doc_ids : a.doc_ids;
            const unsigned long long b = lane_bcast_u64(cur_l, c), e = lane_bcast_u64(end_l, c);
            unsigned long long res[2];
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const uint32_t target = w == 0 ? lo_doc : hi_doc;
                unsigned long long left = b, right = e;  // first index in [left, right] whose doc >= target
                while (right - left > 64) {
                    unsigned long long step = (right - left + 63) / 64;
                    unsigned long long probe = left + step * (unsigned long long)lane;
                    bool ge = probe < right ? ids[probe] >= target : true;
                    unsigned long long m = __ballot(ge);
                    int first = m ? __ffsll((long long)m) - 1 : 64;
                    unsigned long long nl = first == 0 ? left : left + step * (unsigned long long)(first - 1);
                    unsigned long long nr = left + step * (unsigned long long)first;
                    left = nl;
                    right = nr < right ? nr : right;
                }
                unsigned long long probe = left + (unsigned long long)lane;
                bool ge = probe < right ? ids[probe] >= target : true;
                unsigned long long m = __ballot(ge);
                int first = m ? __ffsll((long long)m) - 1 : 64;
                res[w] = left + (unsigned long long)first < right ? left + (unsigned long long)first : right;
            }
            if (lane == c) {
                cur_l = res[0];
                end_l = res[1];
            }
        }
    }
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

    for (;;) {
        // ---- window: the slots are shared out in proportion to what is left of every clause's list in this slice
        //      (doc ids of one slice are spread alike, so the lists then run out at about the same doc id);
        //      the window ends at the smallest doc id that some clause could not fit ----
        const unsigned long long left_l = lane < C ? end_l - cur_l : 0ull;
        const unsigned long long total_left = wave_sum_u64(left_l);
        if (total_left == 0) break;
        uint32_t share_l = 0;
        if (lane < C)
            share_l = total_left <= (unsigned long long)BW_WINDOW ? (uint32_t)left_l
                                                                  : (uint32_t)((left_l * (unsigned long long)(BW_WINDOW - C)) / total_left) + (left_l ? 1u : 0u);
        uint32_t start_l = share_l;  // inclusive scan, then made exclusive
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(start_l, off, 64);
            if (lane >= off) start_l += v;
        }
        const uint32_t n_slots = rl_u32(start_l, 63);
        start_l -= share_l;
        uint32_t hi_c = 0xffffffffu;
        if (lane < C && cur_l + share_l < end_l) hi_c = ((attr_l >> 16) ? a.aux_doc_ids : a.doc_ids)[cur_l + share_l];
        const uint32_t hi = wave_min_u32(hi_c);
        const unsigned long long cy_a = clock64();
        // ---- load phase: 8 postings per lane; the clause of slot g is found against the C share boundaries, its
        //      cursor and attributes come from that clause's lane; then the three dependent steps (index -> doc id ->
        //      tf + fieldnorm) are each issued for all 8 slots before the first use ----
        uint32_t p_doc[8], p_attr[8];
        float p_score[8], p_weight[8];
        int p_clause[8];
        unsigned long long p_idx[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t g = (uint32_t)lane + 64u * m;
            int c = 0;
            for (int j = 1; j < C; j++) c += g >= rl_u32(start_l, j) ? 1 : 0;
            const unsigned long long cur_c = shfl_u64(cur_l, c), end_c = shfl_u64(end_l, c);
            const uint32_t start_c = __shfl(start_l, c, 64);
            p_attr[m] = __shfl(attr_l, c, 64);
            p_weight[m] = __shfl(weight_l, c, 64);
            const unsigned long long i = cur_c + (g - start_c);
            const bool valid = g < n_slots && i < end_c;
            p_clause[m] = valid ? c : -1;
            p_idx[m] = valid ? i : 0;
        }
        // unconditional loads (index 0 stands in for an unused slot), so that all eight are in flight together
#pragma unroll
        for (int m = 0; m < 8; m++) p_doc[m] = ((p_attr[m] >> 16) && p_clause[m] >= 0 ? a.aux_doc_ids : a.doc_ids)[p_idx[m]];
        uint32_t p_tf[8], p_fn[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (p_clause[m] >= 0 && p_doc[m] >= hi) p_clause[m] = -1;  // hi == 0xffffffff: every clause's remainder fits
            const bool scored = p_clause[m] >= 0 && (p_attr[m] & 0xff) != 2 && ((p_attr[m] >> 8) & 0xff) != 2;
            p_tf[m] = ((p_attr[m] >> 16) && scored ? a.aux_tfs : a.tfs)[scored && ((p_attr[m] >> 8) & 0xff) == 0 ? p_idx[m] : 0];
            p_fn[m] = (uint32_t)a.fieldnorm_ids[scored ? p_doc[m] : 0];
        }
#pragma unroll
        for (int m = 0; m < 8; m++) {
            p_score[m] = 0.f;
            if (p_clause[m] >= 0 && (p_attr[m] & 0xff) != 2) {
                const uint32_t mode = (p_attr[m] >> 8) & 0xff;
                if (mode == 2) p_score[m] = p_weight[m];  // ConstScorer(boost)
                else {
                    const float tf = mode == 1 ? 1.0f : (float)p_tf[m];
                    p_score[m] = p_weight[m] * (tf / (tf + tf_cache[p_fn[m]]));
                }
            }
        }
        const unsigned long long cy_b = clock64();
        // ---- probe phase: every posting finds (or claims) its document's slot; which posting claims a slot does
        //      not matter, so all clauses probe together.  The claimer OWNS the document for the fold. ----
        uint32_t p_slot[8];
        uint32_t owned = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            p_slot[m] = 0;
            if (p_clause[m] < 0) continue;
            atomicAdd(&taken[p_clause[m]], 1u);
            const uint32_t d = p_doc[m];
            uint32_t h = (d * 2654435761u) >> BW_SHIFT;
            for (;;) {
                uint32_t old = atomicCAS(&t_key[h], BM25_EMPTY, d);
                if (old == BM25_EMPTY) { owned |= 1u << m; break; }
                if (old == d) break;
                h = (h + 1) & (BW_TABLE - 1);
            }
            p_slot[m] = h;
        }
        // cursors advance by what was taken (lane c reads clause c's counter and clears it)
        {
            const uint32_t t = taken[lane];
            taken[lane] = 0;
            if (lane < C) cur_l += t;
            postings += t;
        }
        // ---- apply phase: clause by clause, so every doc's f32 sum is built in clause order; within a clause every
        //      posting is a different document and one wave's LDS operations execute in order ----
        for (int c = 0; c < C; c++) {
            const int occur = (int)(rl_u32(attr_l, c) & 0xff);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (p_clause[m] != c) continue;
                const uint32_t h = p_slot[m];
                if (occur == 2) {
                    t_flags[h] |= 2;  // MustNot
                } else {
                    t_acc[h] = t_acc[h] + p_score[m];
                    if (occur == 1) t_flags[h] += 0x100;
                    else if (occur == 3) t_flags[h] |= 4;
                    else t_flags[h] |= 1;
                }
            }
        }
        const unsigned long long cy_c = clock64();
        // ---- fold the window into the top-k, count matches, clear the table: each lane folds the documents it owns ----
        uint32_t matched_here = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            bool ok = false;
            uint64_t ck = NIDX_EMPTY_KEY;
            if (owned & (1u << m)) {
                const uint32_t i = p_slot[m];
                const uint32_t d = p_doc[m];
                const uint16_t f = t_flags[i];
                ok = !(f & 2) && (int)(f >> 8) == n_must && (n_group == 0 || (f & 4)) && (n_must > 0 || n_group > 0 || (f & 1));
                if (ok && a.alive) ok = bit_test(a.alive, d);
                if (ok && mbits) atomicOr(&mbits[d >> 5], 1u << (d & 31));
                if (ok && a.order_key) {
                    // order_by_fast_field: the fast value's dense rank decides, then the lower doc id
                    const uint32_t r = a.order_key[d];
                    ck = ((uint64_t)(a.order_desc ? r : ~r) << 32) | (uint64_t)(~d);
                } else if (ok) {
                    float s = t_acc[i];
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
                t_acc[i] = 0.f;
                t_flags[i] = 0;
            }
            unsigned long long okm = __ballot(ok);
            matched_here += (uint32_t)__popcll(okm);
            unsigned long long mm = __ballot(ok && ck > kth);
            while (mm) {
                int src = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                uint64_t nk = lane_bcast_u64(ck, src);
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
            }
        }
        total += matched_here;
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
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) a.out_key[(size_t)blockIdx.x * k + e] = valid ? key : NIDX_EMPTY_KEY;
    }
    postings = wave_sum_u64(postings);  // lane c counted clause c's postings
    if (lane == 0) {
        a.out_count[blockIdx.x] = cnt;
        a.out_total[blockIdx.x] = total;
        a.out_postings[blockIdx.x] = postings;
    }
}

// ---- per-query merge of the slices' lists (TopDocs merges its per-segment collectors the same way): one wave per
//      query walks its work items' sorted key lists, stopping in a list at the first key below the running k-th ----
template <int KL>
__global__ __launch_bounds__(64) void bm25_merge_kernel(Bm25MergeArgs m) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t w0 = m.item_first[q], w1 = m.item_first[q + 1];
    const int k = (int)m.k;
    WaveTopK<KL> top;
    top.init();
    uint64_t kth = NIDX_EMPTY_KEY;
    unsigned long long total = 0, postings = 0;
    for (uint32_t w = w0; w < w1; w++) {
        const uint32_t cnt = m.item_count[w];
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const uint64_t key = i < cnt ? m.item_key[(size_t)w * k + i] : NIDX_EMPTY_KEY;
            unsigned long long mm = __ballot(key > kth);
            if (!mm) break;  // sorted: nothing further down this list can enter
            while (mm) {
                const int src = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const uint64_t nk = lane_bcast_u64(key, src);
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
            }
        }
        if (lane == 0) {
            total += m.item_total[w];
            postings += m.item_postings[w];
        }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < KL; i++) {
        const int e = 64 * i + lane;
        const uint64_t key = top.mine(i);
        const bool valid = key != NIDX_EMPTY_KEY && e < k;
        cnt += (uint32_t)__popcll(__ballot(valid));
        if (e < k) {
            m.out_doc[(size_t)q * k + e] = valid ? rank_key_addr(key) : 0xffffffffu;
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
    if (m.k > 64) hipLaunchKernelGGL(bm25_merge_kernel<4>, dim3(n_queries), dim3(64), 0, s, m);
    else hipLaunchKernelGGL(bm25_merge_kernel<1>, dim3(n_queries), dim3(64), 0, s, m);
    return hipGetLastError();
}

hipError_t launch_bm25_search(const Bm25Args &a, uint32_t n_work, hipStream_t s) {
    if (n_work == 0) return hipSuccess;
    // one WAVE per work item: no block barrier anywhere on the path, four times as many independent items per CU
    if (a.k > 64) hipLaunchKernelGGL(bm25_wave_kernel<4>, dim3(n_work), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(bm25_wave_kernel<1>, dim3(n_work), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace nidx
