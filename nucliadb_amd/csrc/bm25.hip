// bm25.hip — term-at-a-time BM25 scoring + TopDocs for a batch of boolean term queries (gfx950).
//
// Replaces what tantivy does under TextReaderService::do_search (nidx_text/src/reader.rs:433-435)
// and ParagraphReaderService's Searcher::do_search (nidx_paragraph/src/reader.rs:244-348):
//   Bm25Weight::score(fieldnorm_id, tf) = weight * tf / (tf + K1*(1 - B + B*fieldnorm/avg)),
//   BooleanQuery of Should / Must / MustNot term clauses (clause scores summed in clause order),
//   TopDocs::with_limit(k) ordered (score desc, DocAddress asc), Count, the search-after score tweak
//   (reader.rs:350-390), deletions as an alive bitset (nidx_tantivy/src/index_reader.rs:39-74).
//
// One workgroup per work item = (query, doc-id slice of the segment): the host cuts every query into
// slices of roughly equal posting count so that a query with a 400 k-posting term does not become the
// tail of the launch; a slice's cursors are found with a wave-wide 64-ary search.  Inside a work item
// the postings of the query's clauses are consumed in lockstep
// doc-id windows [lo, hi): hi is chosen so that every clause contributes at most PER postings, all
// of a window's (doc -> partial score) pairs live in an LDS hash table, clauses are applied one
// after the other with a barrier in between — so each doc's f32 sum is built in clause order,
// exactly like the oracle's term-at-a-time loop — and the finished window is folded into per-wave
// top-k lists.  Postings are read once, coalesced (doc ids and tfs are separate arrays).
// Bound: HBM; algorithmic bytes per posting scored = 9 (u32 doc + u32 tf + u8 fieldnorm id).
#include "device_common.h"
#include "kernels.h"

namespace nidx {

#define BM25_EMPTY 0xffffffffu

// NT threads per work item: a window holds 8 postings per thread, the hash table is twice that.
template <int NT>
struct Bm25Shared {
    static constexpr int TABLE = NT * 16, DISTINCT = NT * 8, NW = NT / 64;
    uint32_t key[TABLE];
    float acc[TABLE];
    uint16_t flags[TABLE];  // bit0 should-hit, bit1 excluded, bit2 group-hit, bits 8.. must count
    float tf_cache[256];
    unsigned long long cursor[BM25_MAX_CLAUSES];
    unsigned long long end[BM25_MAX_CLAUSES];
    uint8_t c_aux[BM25_MAX_CLAUSES];    // the clause's cursor indexes a materialised term set (aux_doc_ids), not the postings
    uint8_t c_occur[BM25_MAX_CLAUSES];
    uint8_t c_mode[BM25_MAX_CLAUSES];
    float c_weight[BM25_MAX_CLAUSES];
    uint32_t hi;
    uint32_t slot_start[BM25_MAX_CLAUSES + 1];  // this window's slots [slot_start[c], slot_start[c + 1]) belong to clause c
    uint64_t kth[NW];  // every wave's current k-th key: the best of them is the block's admission threshold
    uint32_t taken_c[BM25_MAX_CLAUSES];
    unsigned long long total;
    unsigned long long postings;
};

template <int KL, int NT>
__global__ __launch_bounds__(NT) void bm25_search_kernel(Bm25Args a) {
    constexpr int BM25_TABLE = Bm25Shared<NT>::TABLE, BM25_MAX_DISTINCT = Bm25Shared<NT>::DISTINCT, NW = NT / 64;
    constexpr int HASH_SHIFT = NT == 256 ? 20 : (NT == 128 ? 21 : 22);  // 32 - log2(TABLE)
    __shared__ Bm25Shared<NT> sh;
    __shared__ uint64_t merge[NW > 1 ? NW - 1 : 1][64 * KL];
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const Bm25Work work = a.work[blockIdx.x];
    const uint32_t q = work.query;
    const uint64_t c0 = a.clause_offsets[q], c1 = a.clause_offsets[q + 1];
    const int C = (int)(c1 - c0);
    const Bm25ClauseDev *cl = a.clauses + c0;
    const int k = (int)a.k;

    for (int i = tid; i < BM25_TABLE; i += NT) {
        sh.key[i] = BM25_EMPTY;
        sh.acc[i] = 0.f;
        sh.flags[i] = 0;
    }
    for (int i = tid; i < 256; i += NT) sh.tf_cache[i] = a.tf_cache[i];
    int n_must = 0, n_group = 0;
    for (int c = 0; c < C; c++) {
        n_must += cl[c].occur == 1 ? 1 : 0;
        n_group += cl[c].occur == 3 ? 1 : 0;
    }
    for (int c = tid; c < C; c += NT) {
        sh.c_aux[c] = (cl[c].term & BM25_AUX_TERM) ? 1 : 0;
        sh.c_occur[c] = (uint8_t)cl[c].occur;
        sh.c_mode[c] = (uint8_t)cl[c].mode;
        sh.c_weight[c] = cl[c].weight;
    }
    if (work.n_slices <= 1) {
        if (tid < C) {
            const uint32_t t = cl[tid].term;
            const bool aux = (t & BM25_AUX_TERM) != 0;  // aux lists: [begin, end) pairs
            const uint32_t ti = t & ~BM25_AUX_TERM;
            sh.cursor[tid] = aux ? a.aux_offsets[2 * ti] : a.term_offsets[ti];
            sh.end[tid] = aux ? a.aux_offsets[2 * ti + 1] : a.term_offsets[ti + 1];
        }
    } else {
        // doc range of this slice; per clause a wave finds the first posting >= lo and >= hi
        const uint32_t lo_doc = (uint32_t)((unsigned long long)a.n_docs * work.slice / work.n_slices);
        const uint32_t hi_doc = (uint32_t)((unsigned long long)a.n_docs * (work.slice + 1) / work.n_slices);
        for (int c = wib; c < C; c += NW) {
            const uint32_t t = cl[c].term;
            const bool aux = (t & BM25_AUX_TERM) != 0;
            const uint32_t ti = t & ~BM25_AUX_TERM;
            const uint32_t *ids = aux ? a.aux_doc_ids : a.doc_ids;
            const unsigned long long b = aux ? a.aux_offsets[2 * ti] : a.term_offsets[ti], e = aux ? a.aux_offsets[2 * ti + 1] : a.term_offsets[ti + 1];
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
            if (lane == 0) {
                sh.cursor[c] = res[0];
                sh.end[c] = res[1];
            }
        }
    }
    if (tid == 0) {
        sh.total = 0;
        sh.postings = 0;
    }
    if (tid < NW) sh.kth[tid] = NIDX_EMPTY_KEY;
    __syncthreads();
    unsigned long long cy_load = 0, cy_apply = 0, cy_fold = 0, n_win = 0;
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
        if (tid == 0) sh.hi = 0xffffffffu;
        __syncthreads();
        unsigned long long total_left = 0;
        for (int c = 0; c < C; c++) total_left += sh.end[c] - sh.cursor[c];
        if (total_left == 0) break;
        if (tid == 0) {
            uint32_t at = 0;
            for (int c = 0; c < C; c++) {
                const unsigned long long left = sh.end[c] - sh.cursor[c];
                unsigned long long share = total_left <= (unsigned long long)BM25_MAX_DISTINCT
                                               ? left
                                               : (left * (unsigned long long)(BM25_MAX_DISTINCT - C)) / total_left + (left ? 1 : 0);
                sh.slot_start[c] = at;
                at += (uint32_t)share;
            }
            sh.slot_start[C] = at;
        }
        __syncthreads();
        if (tid < C) {
            unsigned long long cur = sh.cursor[tid], e = sh.end[tid];
            const uint32_t per_c = sh.slot_start[tid + 1] - sh.slot_start[tid];
            if (cur + per_c < e) atomicMin(&sh.hi, (sh.c_aux[tid] ? a.aux_doc_ids : a.doc_ids)[cur + per_c]);
        }
        __syncthreads();
        const uint32_t hi = sh.hi;
        const unsigned long long cy_a = clock64();
        // ---- load phase: the window holds <= 2048 posting slots (slot g belongs to clause g / per); every
        //      thread fetches its 8 slots for ALL clauses, each of the three dependent steps (index -> doc id ->
        //      tf + fieldnorm) issued for all 8 slots before the first use: three memory round trips per window
        uint32_t p_doc[8];
        float p_score[8];
        int p_clause[8];
        unsigned long long p_idx[8];
        uint32_t p_attr[8];   // occur | mode << 8 | aux << 16
        float p_weight[8];
        if (tid < BM25_MAX_CLAUSES) sh.taken_c[tid] = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t g = (uint32_t)tid + (uint32_t)NT * m;
            int c = 0;
            while (c < C && g >= sh.slot_start[c + 1]) c++;
            const int cc = c < C ? c : 0;
            const unsigned long long i = sh.cursor[cc] + (g - sh.slot_start[cc]);
            const bool valid = c < C && i < sh.end[cc];
            p_clause[m] = valid ? c : -1;
            p_idx[m] = valid ? i : 0;
            p_attr[m] = (uint32_t)sh.c_occur[cc] | ((uint32_t)sh.c_mode[cc] << 8) | ((uint32_t)sh.c_aux[cc] << 16);
            p_weight[m] = sh.c_weight[cc];
        }
        // unconditional loads (index 0 stands in for an unused slot), so that all eight are in flight together
#pragma unroll
        for (int m = 0; m < 8; m++) p_doc[m] = ((p_attr[m] >> 16) && p_clause[m] >= 0 ? a.aux_doc_ids : a.doc_ids)[p_idx[m]];
        uint32_t p_tf[8], p_fn[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (p_clause[m] >= 0 && p_doc[m] >= hi) p_clause[m] = -1;  // hi == 0xffffffff: every clause's remainder fits
            const bool scored = p_clause[m] >= 0 && (p_attr[m] & 0xff) != 2 && ((p_attr[m] >> 8) & 0xff) != 2;
            p_tf[m] = a.tfs[scored && ((p_attr[m] >> 8) & 0xff) == 0 ? p_idx[m] : 0];
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
                    p_score[m] = p_weight[m] * (tf / (tf + sh.tf_cache[p_fn[m]]));
                }
            }
        }
        const unsigned long long cy_b = clock64();
        // ---- probe phase: every posting finds (or claims) its document's slot; which posting claims a slot does
        //      not matter, so all clauses probe together.  The claimer OWNS the document for the fold.
        uint32_t p_slot[8];
        uint32_t owned = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            p_slot[m] = 0;
            if (p_clause[m] < 0) continue;
            const uint32_t d = p_doc[m];
            uint32_t h = (d * 2654435761u) >> HASH_SHIFT;
            for (;;) {
                uint32_t old = atomicCAS(&sh.key[h], BM25_EMPTY, d);
                if (old == BM25_EMPTY) { owned |= 1u << m; break; }
                if (old == d) break;
                h = (h + 1) & (BM25_TABLE - 1);
            }
            p_slot[m] = h;
        }
        __syncthreads();  // taken_c zeroed, every key placed
        // ---- apply phase: clause by clause (barrier in between) so every doc's f32 sum is built in clause order;
        //      within a clause every posting is a different document, so the read-modify-writes do not collide
        for (int c = 0; c < C; c++) {
            const int occur = sh.c_occur[c];
            uint32_t mine = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (p_clause[m] != c) continue;
                mine++;
                const uint32_t h = p_slot[m];
                if (occur == 2) {
                    sh.flags[h] |= 2;  // MustNot
                } else {
                    sh.acc[h] = sh.acc[h] + p_score[m];
                    if (occur == 1) sh.flags[h] += 0x100;
                    else if (occur == 3) sh.flags[h] |= 4;
                    else sh.flags[h] |= 1;
                }
            }
            if (mine) atomicAdd(&sh.taken_c[c], mine);
            __syncthreads();
        }
        if (tid < C) {
            sh.cursor[tid] += sh.taken_c[tid];
            atomicAdd(&sh.postings, (unsigned long long)sh.taken_c[tid]);
        }
        const unsigned long long cy_c = clock64();
        // ---- fold the window into the top-k, count matches, clear the table: each thread folds the documents it
        //      owns.  The admission threshold is shared by the four waves (any wave's k-th key bounds the block's).
        uint32_t matched_here = 0;
#pragma unroll
        for (int w = 0; w < NW; w++)
            if (sh.kth[w] > kth) kth = sh.kth[w];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            bool ok = false;
            uint64_t ck = NIDX_EMPTY_KEY;
            if (owned & (1u << m)) {
                const uint32_t i = p_slot[m];
                const uint32_t d = p_doc[m];
                const uint16_t f = sh.flags[i];
                ok = !(f & 2) && (int)(f >> 8) == n_must && (n_group == 0 || (f & 4)) && (n_must > 0 || n_group > 0 || (f & 1));
                if (ok && a.alive) ok = bit_test(a.alive, d);
                if (ok && mbits) atomicOr(&mbits[d >> 5], 1u << (d & 31));
                if (ok && a.order_key) {
                    // order_by_fast_field: the fast value's dense rank decides, then the lower doc id
                    const uint32_t r = a.order_key[d];
                    ck = ((uint64_t)(a.order_desc ? r : ~r) << 32) | (uint64_t)(~d);
                } else if (ok) {
                    float s = sh.acc[i];
                    if (has_after) {
                        // tweak_score: -inf for docs not after the cursor
                        uint64_t addr = ((uint64_t)a.segment_ord << 32) | d;
                        int32_t sk = total_key(s);
                        bool after = sk < after_key || (sk == after_key && (after_tie == 0 || (after_tie == 1 && addr > after_addr)));
                        if (!after) s = -INFINITY;
                    }
                    ck = rank_key(s, d);
                }
                sh.key[i] = BM25_EMPTY;
                sh.acc[i] = 0.f;
                sh.flags[i] = 0;
            }
            unsigned long long okm = __ballot(ok);
            matched_here += (uint32_t)__popcll(okm);
            unsigned long long mm = __ballot(ok && ck > kth);
            while (mm) {
                int src = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                uint64_t nk = shfl_u64(ck, src);
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
            }
        }
        if (lane == 0) sh.kth[wib] = kth;
        if (lane == 0 && matched_here) atomicAdd(&sh.total, (unsigned long long)matched_here);
        __syncthreads();
        cy_load += cy_b - cy_a;
        cy_apply += cy_c - cy_b;
        cy_fold += clock64() - cy_c;
        n_win++;
    }
    if (a.dbg && tid == 0) {
        atomicAdd(&a.dbg[0], cy_load);
        atomicAdd(&a.dbg[1], cy_apply);
        atomicAdd(&a.dbg[2], cy_fold);
        atomicAdd(&a.dbg[3], clock64() - cy_t0);
        atomicAdd(&a.dbg[4], n_win);
        atomicAdd(&a.dbg[5], 1ull);
    }

    // ---- merge the four waves' lists ----
    if (wib > 0) {
#pragma unroll
        for (int i = 0; i < KL; i++) merge[wib - 1][64 * i + lane] = top.mine(i);
    }
    __syncthreads();
    if (wib == 0) {
        kth = top.at(k - 1);  // this wave's own k-th key (the running threshold may have been another wave's)
        for (int w = 0; w < NW - 1; w++)
            for (int i = 0; i < k; i++) {
                uint64_t nk = merge[w][i];
                if (nk == NIDX_EMPTY_KEY) break;
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
                else break;
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
        if (lane == 0) {
            a.out_count[blockIdx.x] = cnt;
            a.out_total[blockIdx.x] = sh.total;
            a.out_postings[blockIdx.x] = sh.postings;
        }
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
                const uint64_t nk = shfl_u64(key, src);
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
    if (a.k > 64) hipLaunchKernelGGL((bm25_search_kernel<4, BM25_ITEM_THREADS>), dim3(n_work), dim3(BM25_ITEM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((bm25_search_kernel<1, BM25_ITEM_THREADS>), dim3(n_work), dim3(BM25_ITEM_THREADS), 0, s, a);
    return hipGetLastError();
}

}  // namespace nidx
