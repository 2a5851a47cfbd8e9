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

#define BM25_TABLE 4096
#define BM25_MAX_DISTINCT 2048
#define BM25_EMPTY 0xffffffffu

struct Bm25Shared {
    uint32_t key[BM25_TABLE];
    float acc[BM25_TABLE];
    uint16_t flags[BM25_TABLE];  // bit0 should-hit, bit1 excluded, bit2 group-hit, bits 8.. must count
    float tf_cache[256];
    unsigned long long cursor[BM25_MAX_CLAUSES];
    unsigned long long end[BM25_MAX_CLAUSES];
    const uint32_t *base[BM25_MAX_CLAUSES];  // doc-id array the clause's cursor indexes (postings, or a materialised term set)
    uint32_t hi;
    uint32_t taken_c[BM25_MAX_CLAUSES];
    unsigned long long total;
    unsigned long long postings;
};

template <int KL>
__global__ __launch_bounds__(256) void bm25_search_kernel(Bm25Args a) {
    __shared__ Bm25Shared sh;
    __shared__ uint64_t merge[3][64 * KL];
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const Bm25Work work = a.work[blockIdx.x];
    const uint32_t q = work.query;
    const uint64_t c0 = a.clause_offsets[q], c1 = a.clause_offsets[q + 1];
    const int C = (int)(c1 - c0);
    const Bm25ClauseDev *cl = a.clauses + c0;
    const int k = (int)a.k;

    for (int i = tid; i < BM25_TABLE; i += 256) {
        sh.key[i] = BM25_EMPTY;
        sh.acc[i] = 0.f;
        sh.flags[i] = 0;
    }
    if (tid < 256) sh.tf_cache[tid] = a.tf_cache[tid];
    int n_must = 0, n_group = 0;
    for (int c = 0; c < C; c++) {
        n_must += cl[c].occur == 1 ? 1 : 0;
        n_group += cl[c].occur == 3 ? 1 : 0;
    }
    if (tid < C) sh.base[tid] = (cl[tid].term & BM25_AUX_TERM) ? a.aux_doc_ids : a.doc_ids;
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
        for (int c = wib; c < C; c += 4) {
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
    __syncthreads();
    const uint32_t per = C > 0 ? (uint32_t)(BM25_MAX_DISTINCT / C) : 1u;
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
        // ---- window end: smallest doc id that some clause could not fit ----
        if (tid == 0) sh.hi = 0xffffffffu;
        __syncthreads();
        bool any_left = false;
        for (int c = 0; c < C; c++) any_left = any_left || sh.cursor[c] < sh.end[c];
        if (!any_left) break;
        if (tid < C) {
            unsigned long long cur = sh.cursor[tid], e = sh.end[tid];
            if (cur + per < e) atomicMin(&sh.hi, sh.base[tid][cur + per]);
        }
        __syncthreads();
        const uint32_t hi = sh.hi;
        const unsigned long long cy_a = clock64();
        // ---- load phase: the window holds <= 2048 posting slots (slot g belongs to clause g / per);
        //      every thread fetches its 8 slots for ALL clauses up front, so a window costs three
        //      dependent memory round trips (doc id -> tf + fieldnorm gather) instead of three per clause
        uint32_t p_doc[8];
        float p_score[8];
        int p_clause[8];
        if (tid < BM25_MAX_CLAUSES) sh.taken_c[tid] = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t g = (uint32_t)tid + 256u * m;
            const int c = (int)(g / per);
            p_clause[m] = -1;
            p_doc[m] = 0;
            p_score[m] = 0.f;
            if (c < C) {
                const unsigned long long i = sh.cursor[c] + (g - (uint32_t)c * per);
                if (i < sh.end[c]) {
                    const uint32_t d = sh.base[c][i];
                    if (d < hi) {  // hi == 0xffffffff: every clause's remainder fits
                        p_clause[m] = c;
                        p_doc[m] = d;
                        const int mode = cl[c].mode;
                        if (cl[c].occur != 2) {
                            if (mode == 2) p_score[m] = cl[c].weight;  // ConstScorer(boost)
                            else {
                                const float tf = mode == 1 ? 1.0f : (float)a.tfs[i];
                                const float norm = sh.tf_cache[a.fieldnorm_ids[d]];
                                p_score[m] = cl[c].weight * (tf / (tf + norm));
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // taken_c zeroed
        const unsigned long long cy_b = clock64();
        // ---- apply phase: clause by clause (barrier in between) so every doc's f32 sum is built in clause order
        for (int c = 0; c < C; c++) {
            const int occur = cl[c].occur;
            uint32_t mine = 0;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (p_clause[m] != c) continue;
                mine++;
                const uint32_t d = p_doc[m];
                uint32_t h = (d * 2654435761u) >> 20;
                for (;;) {
                    uint32_t old = atomicCAS(&sh.key[h], BM25_EMPTY, d);
                    if (old == BM25_EMPTY || old == d) break;
                    h = (h + 1) & (BM25_TABLE - 1);
                }
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
        // ---- fold the window into the top-k, count matches, clear the table ----
        uint32_t matched_here = 0;
        for (int base = wib * 64; base < BM25_TABLE; base += 256) {
            int i = base + lane;
            uint32_t d = sh.key[i];
            bool ok = false;
            uint64_t ck = NIDX_EMPTY_KEY;
            if (d != BM25_EMPTY) {
                uint16_t f = sh.flags[i];
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
            unsigned long long m = __ballot(ok && ck > kth);
            while (m) {
                int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                uint64_t nk = shfl_u64(ck, src);
                if (nk > kth) kth = top.insert_kth(nk, k, lane);
            }
        }
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
        for (int w = 0; w < 3; w++)
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
            if (e < k) {
                a.out_doc[(size_t)blockIdx.x * k + e] = valid ? rank_key_addr(key) : 0xffffffffu;
                a.out_score[(size_t)blockIdx.x * k + e] = valid ? rank_key_score(key) : 0.f;
            }
        }
        if (lane == 0) {
            a.out_count[blockIdx.x] = cnt;
            a.out_total[blockIdx.x] = sh.total;
            a.out_postings[blockIdx.x] = sh.postings;
        }
    }
}

hipError_t launch_bm25_search(const Bm25Args &a, uint32_t n_work, hipStream_t s) {
    if (n_work == 0) return hipSuccess;
    if (a.k > 64) hipLaunchKernelGGL(bm25_search_kernel<4>, dim3(n_work), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bm25_search_kernel<1>, dim3(n_work), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace nidx
