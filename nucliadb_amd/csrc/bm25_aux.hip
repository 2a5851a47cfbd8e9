// bm25_aux.hip — the pieces around the BM25 scorer, gfx950.
//
//   fuzzy_match_kernel    FuzzyTermQuery's Levenshtein automaton (nidx_paragraph/src/fuzzy_query.rs:127-251,
//                         query_parser/fuzzy_parser.rs:35-93: distance 1, transposition_cost_one, prefix DFA for the
//                         last literal) evaluated against EVERY term of the dictionary in parallel — tantivy walks
//                         its FST with the DFA; here the dictionary is a byte blob in HBM and each thread decides one
//                         term with the closed form of "edit distance <= 1".  HBM-bound: one pass over the blob.
//   bitset_compact_kernel AutomatonWeight::scorer (fuzzy_query.rs:90-125) inserts every posting of every accepted term
//                         into a doc BitSet and scores it with ConstScorer: the union is scattered into a bitset
//                         (filter.hip: launch_bitset_scatter) and compacted here into an ascending doc-id list that the
//                         scoring kernel consumes like any posting list (bm25.hip: BM25_AUX_TERM).
//   facet_count_kernel    FacetCollector (nidx_text/src/reader.rs:43-62,391-398; nidx_paragraph/src/reader.rs:244-348):
//                         a facet's count = |postings(facet term) ∩ matching documents|; the scoring kernel leaves the
//                         matching documents of a query as a bitset.
#include <algorithm>
#include "device_common.h"
#include "kernels.h"

namespace nidx {

#define FUZZY_MAX_CP 48  /* tantivy's RemoveLongFilter drops tokens over 40 bytes: no term is longer */

// first `cap` unicode scalar values of a UTF-8 string; returns how many were decoded, *more = bytes were left
__device__ inline int utf8_head(const uint8_t *s, uint32_t len, uint32_t *out, int cap, bool *more) {
    int n = 0;
    uint32_t i = 0;
    while (i < len && n < cap) {
        uint32_t c = s[i];
        const int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : 0;
        if (extra == 1) c &= 0x1f;
        else if (extra == 2) c &= 0x0f;
        else if (extra == 3) c &= 0x07;
        i++;
        for (int e = 0; e < extra && i < len; e++, i++) c = (c << 6) | (s[i] & 0x3f);
        out[n++] = c;
    }
    *more = i < len;
    return n;
}

// q[a .. a + len) == t[b .. b + len) ?
__device__ inline bool cp_equal(const uint32_t *q, int a, const uint32_t *t, int b, int len) {
    for (int i = 0; i < len; i++)
        if (q[a + i] != t[b + i]) return false;
    return true;
}

__global__ __launch_bounds__(256) void fuzzy_match_kernel(const uint8_t *dict_bytes, const unsigned long long *dict_offsets,
                                                          uint32_t n_terms, const uint32_t *query_cp, uint32_t n_query_cp, int prefix,
                                                          uint8_t *flags) {
    __shared__ uint32_t q[FUZZY_MAX_CP];
    if (threadIdx.x < n_query_cp) q[threadIdx.x] = query_cp[threadIdx.x];
    __syncthreads();
    const uint32_t term = blockIdx.x * blockDim.x + threadIdx.x;
    if (term >= n_terms) return;
    const int n = (int)n_query_cp;
    const unsigned long long b = dict_offsets[term], e = dict_offsets[term + 1];
    uint32_t t[FUZZY_MAX_CP + 2];
    bool more;
    // only the first n + 2 characters can matter: one more and the term is too long (exact), or past every
    // prefix that could be within one edit of the query (prefix)
    const int m = utf8_head(dict_bytes + b, (uint32_t)(e - b), t, n + 2, &more);
    const bool t_longer_than_n1 = m == n + 2;  // the term has at least n + 2 characters
    int i = 0;
    const int lim = n < m ? n : m;
    while (i < lim && q[i] == t[i]) i++;
    bool ok = false;
    if (!prefix) {
        if (!t_longer_than_n1 && m + 1 >= n) {         // |n - m| <= 1
            if (i == lim) ok = true;                   // one is a prefix of the other (or they are equal)
            else if (n == m) ok = cp_equal(q, i + 1, t, i + 1, n - i - 1) ||                                    // substitution
                                  (i + 1 < n && q[i] == t[i + 1] && q[i + 1] == t[i] && cp_equal(q, i + 2, t, i + 2, n - i - 2));  // transposition
            else if (n == m + 1) ok = cp_equal(q, i + 1, t, i, n - i - 1);                                     // the query has one extra
            else ok = cp_equal(q, i, t, i + 1, n - i);                                                           // the term has one extra
        }
    } else {
        if (i == n) ok = true;                         // the query itself is a prefix of the term
        else if (i == m) ok = n - m <= 1;              // the term is the query minus its last character
        else {
            if (m >= n) ok = cp_equal(q, i + 1, t, i + 1, n - i - 1) ||                                          // substitution, prefix of length n
                             (i + 1 < n && q[i] == t[i + 1] && q[i + 1] == t[i] && cp_equal(q, i + 2, t, i + 2, n - i - 2));
            if (!ok && m >= n - 1) ok = cp_equal(q, i + 1, t, i, n - i - 1);                                    // prefix of length n - 1
            if (!ok && m >= n + 1) ok = cp_equal(q, i, t, i + 1, n - i);                                        // prefix of length n + 1
        }
    }
    flags[term] = ok ? 1 : 0;
}

hipError_t launch_fuzzy_match(const uint8_t *dict_bytes, const unsigned long long *dict_offsets, uint32_t n_terms,
                              const uint32_t *query_cp, uint32_t n_query_cp, int prefix, uint8_t *flags, hipStream_t s) {
    if (n_terms == 0) return hipSuccess;
    if (n_query_cp == 0 || n_query_cp > FUZZY_MAX_CP) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fuzzy_match_kernel, dim3((n_terms + 255) / 256), dim3(256), 0, s, dict_bytes, dict_offsets, n_terms, query_cp,
                       n_query_cp, prefix, flags);
    return hipGetLastError();
}

// ---- bitset -> ascending id list, one block per bitset -------------------------------------------------------
__global__ __launch_bounds__(256) void bitset_compact_kernel(const uint64_t *bits, uint32_t n_words, const unsigned long long *out_offsets,
                                                             uint32_t *out, uint32_t *counts) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_s;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const uint64_t *b = bits + (size_t)blockIdx.x * n_words;
    uint32_t *o = out + out_offsets[blockIdx.x];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (uint32_t w0 = 0; w0 < n_words; w0 += 256) {
        const uint32_t w = w0 + (uint32_t)tid;
        uint64_t x = w < n_words ? b[w] : 0ull;
        const uint32_t c = (uint32_t)__popcll(x);
        // exclusive scan of c over the block: wave scan, then the four wave totals
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_sum[wib] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int i = 0; i < wib; i++) before += wave_sum[i];
        uint32_t pos = before + incl - c;
        while (x) {
            const int bit = __ffsll((long long)x) - 1;
            x &= x - 1;
            o[pos++] = w * 64u + (uint32_t)bit;
        }
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    if (tid == 0) counts[blockIdx.x] = base_s;
}

hipError_t launch_bitset_compact(const uint64_t *bits, uint32_t n_words, uint32_t n_sets, const unsigned long long *out_offsets,
                                 uint32_t *out, uint32_t *counts, hipStream_t s) {
    if (n_sets == 0) return hipSuccess;
    hipLaunchKernelGGL(bitset_compact_kernel, dim3(n_sets), dim3(256), 0, s, bits, n_words, out_offsets, out, counts);
    return hipGetLastError();
}

// ---- PhraseQuery ---------------------------------------------------------------------------------------------------
// first index in [b, e) whose doc id is >= d
__device__ inline unsigned long long lower_bound_doc(const uint32_t *doc_ids, unsigned long long b, unsigned long long e, uint32_t d) {
    while (b < e) {
        const unsigned long long mid = b + (e - b) / 2;
        if (doc_ids[mid] < d) b = mid + 1;
        else e = mid;
    }
    return b;
}

// PhraseScorer with slop over one document (tantivy phrase_scorer.rs, PhraseQuery::set_slop: "the slop is a budget between all
// terms ... works in both directions"): `left` = the matches so far as (position, budget used) pairs, ascending; against the
// positions [rb, re) of the next term shifted by `shift` a pair matches when |left - right| + used <= slop; the last left value not
// beyond `right` that still fits the budget is the one consumed ("there could be a better match"); the match moves on as
// (right, used + distance), written over the head of `left`.  -> matches
__device__ inline uint32_t slop_intersect(uint32_t *left, uint32_t n_left, const uint32_t *positions, unsigned long long rb, unsigned long long re,
                                          uint32_t shift, uint32_t slop) {
    uint32_t li = 0, count = 0;
    unsigned long long ri = rb;
    while (li < n_left && ri < re) {
        uint32_t lv = left[2 * li], used = left[2 * li + 1];
        const uint32_t rv = positions[ri] + shift;
        uint32_t dist = lv > rv ? lv - rv : rv - lv;
        if (dist + used <= slop) {
            while (li + 1 < n_left) {
                const uint32_t nv = left[2 * li + 2], nu = left[2 * li + 3];
                if (nv > rv || rv - nv + nu > slop) break;
                li++;
                lv = nv, used = nu, dist = rv - nv;
            }
            left[2 * count] = rv;
            left[2 * count + 1] = used + dist;
            count++;
            li++;
            ri++;
        } else if (lv < rv) li++;
        else ri++;
    }
    return count;
}

__global__ __launch_bounds__(256) void phrase_match_kernel(const unsigned long long *term_offsets, const uint32_t *doc_ids,
                                                           const unsigned long long *pos_offsets, const uint32_t *positions, PhraseDev ph,
                                                           uint32_t *tmp_tf, uint32_t *slop_left) {
    const unsigned long long b0 = term_offsets[ph.terms[ph.driver]], e0 = term_offsets[ph.terms[ph.driver] + 1];
    const unsigned long long i0 = b0 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= e0) return;
    const uint32_t d = doc_ids[i0];
    // the document's posting in every other term's list
    unsigned long long at[BM25_MAX_PHRASE_TERMS];
    bool all = true;
    for (uint32_t t = 0; t < ph.n_terms && all; t++) {
        if (t == ph.driver) { at[t] = i0; continue; }
        const unsigned long long b = term_offsets[ph.terms[t]], e = term_offsets[ph.terms[t] + 1];
        const unsigned long long i = lower_bound_doc(doc_ids, b, e, d);
        all = i < e && doc_ids[i] == d;
        at[t] = i;
    }
    uint32_t count = 0;
    if (all && ph.slop) {
        // left = the first term's positions + (n_terms - 1) with no budget used, in its own region of the scratch list
        const unsigned long long p0 = pos_offsets[at[0]], p1 = pos_offsets[at[0] + 1];
        uint32_t *left = slop_left + 2 * (p0 - pos_offsets[term_offsets[ph.terms[0]]]);
        uint32_t n_left = (uint32_t)(p1 - p0);
        for (uint32_t i = 0; i < n_left; i++) left[2 * i] = positions[p0 + i] + (ph.n_terms - 1), left[2 * i + 1] = 0u;
        for (uint32_t t = 1; t < ph.n_terms && n_left; t++)
            n_left = slop_intersect(left, n_left, positions, pos_offsets[at[t]], pos_offsets[at[t] + 1], ph.n_terms - 1 - t, ph.slop);
        count = n_left;
    } else if (all) {
        for (unsigned long long pi = pos_offsets[i0]; pi < pos_offsets[i0 + 1]; pi++) {
            const uint32_t p = positions[pi];
            if (p < ph.driver) continue;  // the phrase would start before position 0
            const uint32_t start = p - ph.driver;
            bool ok = true;
            for (uint32_t t = 0; t < ph.n_terms && ok; t++) {
                if (t == ph.driver) continue;
                const uint32_t want = start + t;
                unsigned long long lo = pos_offsets[at[t]], hi = pos_offsets[at[t] + 1];
                while (lo < hi) {  // positions ascend
                    const unsigned long long mid = lo + (hi - lo) / 2;
                    if (positions[mid] < want) lo = mid + 1;
                    else hi = mid;
                }
                ok = lo < pos_offsets[at[t] + 1] && positions[lo] == want;
            }
            count += ok ? 1u : 0u;
        }
    }
    tmp_tf[i0 - b0] = count;
}

hipError_t launch_phrase_match(const unsigned long long *term_offsets, const uint32_t *doc_ids, const unsigned long long *pos_offsets,
                               const uint32_t *positions, PhraseDev ph, uint32_t n_driver, uint32_t *tmp_tf, uint32_t *slop_left, hipStream_t s) {
    if (n_driver == 0) return hipSuccess;
    if (ph.slop && (!slop_left || ph.n_terms < 2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(phrase_match_kernel, dim3((n_driver + 255) / 256), dim3(256), 0, s, term_offsets, doc_ids, pos_offsets, positions, ph, tmp_tf,
                       slop_left);
    return hipGetLastError();
}

// matches (tmp_tf > 0) -> ascending (doc, tf) list; one block, running offset
__global__ __launch_bounds__(256) void phrase_compact_kernel(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph,
                                                             const uint32_t *tmp_tf, const uint8_t *fieldnorm_ids, unsigned long long out_begin,
                                                             uint32_t *out_ids, uint32_t *out_tfs, uint32_t *out_count) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_s;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const unsigned long long b0 = term_offsets[ph.terms[ph.driver]], e0 = term_offsets[ph.terms[ph.driver] + 1];
    const uint32_t n = (uint32_t)(e0 - b0);
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + (uint32_t)tid;
        const uint32_t tf = i < n ? tmp_tf[i] : 0u;
        const uint32_t c = tf ? 1u : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_sum[wib] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int w = 0; w < wib; w++) before += wave_sum[w];
        if (c) {
            const uint32_t d = doc_ids[b0 + i];
            out_ids[out_begin + before + incl - 1] = d;
            out_tfs[out_begin + before + incl - 1] = (tf & 0xffffffu) | ((uint32_t)fieldnorm_ids[d] << 24);   // the scorer's posting word
        }
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    if (tid == 0) *out_count = base_s;
}

hipError_t launch_phrase_compact(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph, const uint32_t *tmp_tf,
                                 const uint8_t *fieldnorm_ids, unsigned long long out_begin, uint32_t *out_ids, uint32_t *out_tfs,
                                 uint32_t *out_count, hipStream_t s) {
    hipLaunchKernelGGL(phrase_compact_kernel, dim3(1), dim3(256), 0, s, term_offsets, doc_ids, ph, tmp_tf, fieldnorm_ids, out_begin, out_ids, out_tfs,
                       out_count);
    return hipGetLastError();
}

// ---- nested BooleanQuery -> pre-scored posting list -----------------------------------------------------------------------------
struct SubRange {
    const uint32_t *ids, *words;
    unsigned long long b, e;
};

__device__ inline SubRange sub_leaf_range(const SubqueryLists &L, uint32_t src) {
    SubRange r;
    if (src & BM25_AUX_TERM) {
        const uint32_t a = src & ~BM25_AUX_TERM;
        r.ids = L.aux_ids;
        r.words = L.aux_words;
        r.b = L.aux_begin[a];
        r.e = r.b + L.aux_counts[a];
    } else {
        r.ids = L.doc_ids;
        r.words = L.words;
        r.b = L.term_offsets[src];
        r.e = L.term_offsets[src + 1];
    }
    return r;
}

__device__ inline SubRange sub_candidates(const SubqueryLists &L, const SubqueryDev &sq) {
    if (sq.driver == BM25_SUB_DRIVER_UNION) return SubRange{L.union_ids, nullptr, 0ull, (unsigned long long)*L.union_count};
    return sub_leaf_range(L, sq.src[sq.driver]);
}

__global__ __launch_bounds__(256) void subquery_scatter_kernel(SubqueryLists L, uint32_t src, uint64_t *bits) {
    const SubRange r = sub_leaf_range(L, src);
    for (unsigned long long i = r.b + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < r.e; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t d = r.ids[i];
        atomicOr(reinterpret_cast<unsigned long long *>(bits) + (d >> 6), 1ull << (d & 63));
    }
}

hipError_t launch_subquery_scatter(const SubqueryLists &L, uint32_t src, unsigned long long upper_bound, uint64_t *bits, hipStream_t s) {
    if (upper_bound == 0) return hipSuccess;
    const unsigned long long blocks = std::min<unsigned long long>((upper_bound + 255) / 256, 4096);
    hipLaunchKernelGGL(subquery_scatter_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, L, src, bits);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void subquery_match_kernel(SubqueryLists L, const float *tf_cache, SubqueryDev sq, uint32_t *tmp_ok, uint32_t *tmp_score) {
    const SubRange cand = sub_candidates(L, sq);
    const unsigned long long i0 = cand.b + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= cand.e) return;
    const uint32_t d = cand.ids[i0];
    uint32_t mask = 0, must_m = 0, not_m = 0, should_m = 0;
    uint32_t group_m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float acc = 0.f;   // the nested scorer's own sum, leaf by leaf
    for (uint32_t t = 0; t < sq.n; t++) {
        const uint32_t occur = sq.occur[t];
        if (occur == 0) should_m |= 1u << t;
        else if (occur == 1) must_m |= 1u << t;
        else if (occur == 2) not_m |= 1u << t;
        else group_m[(occur - 3) & 7] |= 1u << t;
        SubRange r = cand;
        unsigned long long i = i0;
        bool here = true;
        if (t != sq.driver) {
            r = sub_leaf_range(L, sq.src[t]);
            i = lower_bound_doc(r.ids, r.b, r.e, d);
            here = i < r.e && r.ids[i] == d;
        }
        if (!here) continue;
        mask |= 1u << t;
        if (occur == 2) continue;
        const uint32_t mode = sq.mode[t];
        if (mode == 2) acc += sq.weight[t];
        else {
            const uint32_t w = r.words[i];
            if (mode == 3) acc += sq.weight[t] * __uint_as_float(w);   // boost x the nested query's own score
            else {
                const float tf = mode == 0 ? (float)(w & 0xffffffu) : 1.0f;
                acc += sq.weight[t] * (tf / (tf + tf_cache[w >> 24]));
            }
        }
    }
    bool ok = (mask & must_m) == must_m && (mask & not_m) == 0;
    bool any_group = false;
    for (int g = 0; g < 8; g++) {
        any_group |= group_m[g] != 0;
        if (group_m[g] && (mask & group_m[g]) == 0) ok = false;
    }
    if (!must_m && !any_group && (mask & should_m) == 0) ok = false;   // only Should leaves: one of them has to hold the document
    tmp_ok[i0 - cand.b] = ok ? 1u : 0u;
    tmp_score[i0 - cand.b] = __float_as_uint(acc);
}

hipError_t launch_subquery_match(const SubqueryLists &L, const float *tf_cache, const SubqueryDev &sq, uint32_t n_cand_max, uint32_t *tmp_ok,
                                 uint32_t *tmp_score, hipStream_t s) {
    if (n_cand_max == 0) return hipSuccess;
    hipLaunchKernelGGL(subquery_match_kernel, dim3((n_cand_max + 255) / 256), dim3(256), 0, s, L, tf_cache, sq, tmp_ok, tmp_score);
    return hipGetLastError();
}

// matches -> ascending (doc, score bits) list; one block, running offset (the candidates are in doc order)
__global__ __launch_bounds__(256) void subquery_compact_kernel(SubqueryLists L, SubqueryDev sq, const uint32_t *tmp_ok, const uint32_t *tmp_score,
                                                               unsigned long long out_begin, uint32_t *out_ids, uint32_t *out_words, uint32_t *out_count) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_s;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const SubRange cand = sub_candidates(L, sq);
    const uint32_t n = (uint32_t)(cand.e - cand.b);
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + (uint32_t)tid;
        const uint32_t c = i < n ? tmp_ok[i] : 0u;
        const uint32_t d = c ? cand.ids[cand.b + i] : 0u, sc = c ? tmp_score[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_sum[wib] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int w = 0; w < wib; w++) before += wave_sum[w];
        if (c) {
            out_ids[out_begin + before + incl - 1] = d;
            out_words[out_begin + before + incl - 1] = sc;
        }
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    if (tid == 0) *out_count = base_s;
}

hipError_t launch_subquery_compact(const SubqueryLists &L, const SubqueryDev &sq, const uint32_t *tmp_ok, const uint32_t *tmp_score,
                                   unsigned long long out_begin, uint32_t *out_ids, uint32_t *out_words, uint32_t *out_count, hipStream_t s) {
    hipLaunchKernelGGL(subquery_compact_kernel, dim3(1), dim3(256), 0, s, L, sq, tmp_ok, tmp_score, out_begin, out_ids, out_words, out_count);
    return hipGetLastError();
}

// ---- posting words ---------------------------------------------------------------------------------------------------
// The scorer needs a posting's term frequency and its document's fieldnorm id.  Fetching the fieldnorm by doc id is one random
// cache line per posting (64-128 B moved for one byte); the resident copy therefore carries it in the posting itself:
// word = tf | fieldnorm_id << 24, written once when the segment is opened.  *flag |= 1 when a frequency does not fit 24 bits.
__global__ __launch_bounds__(256) void bm25_pack_fieldnorm_kernel(const uint32_t *__restrict__ doc_ids, uint32_t *__restrict__ tfs,
                                                                  const uint8_t *__restrict__ fieldnorm_ids, unsigned long long n, uint32_t n_docs,
                                                                  uint32_t *flag) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t tf = tfs[i], d = doc_ids[i];
        if (tf >= (1u << 24)) atomicOr(flag, 1u);
        if (d >= n_docs) {
            atomicOr(flag, 2u);
            continue;
        }
        tfs[i] = (tf & 0xffffffu) | ((uint32_t)fieldnorm_ids[d] << 24);
    }
}

hipError_t launch_bm25_pack_fieldnorm(const uint32_t *doc_ids, uint32_t *tfs, const uint8_t *fieldnorm_ids, unsigned long long n, uint32_t n_docs,
                                      uint32_t *flag, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned long long blocks = (n + 255ull) / 256ull;
    hipLaunchKernelGGL(bm25_pack_fieldnorm_kernel, dim3((uint32_t)(blocks < 65536ull ? blocks : 65536ull)), dim3(256), 0, s, doc_ids, tfs, fieldnorm_ids, n,
                       n_docs, flag);
    return hipGetLastError();
}

// ---- per-term score floors (round 6) ------------------------------------------------------------------------------------------------
// tantivy prunes a union of term scorers with block-max WAND (the reference's TopDocs collector calls for_each_pruning); the streaming
// scorer's counterpart is a STATIC bound: out[t][j] = the smallest fieldnorm id f such that at least BM25_FLOOR_RANKS[j] of the first `cap`
// postings of term t belong to documents of fieldnorm id <= f (255: fewer than that many postings).  The quotient tf / (tf + K(fieldnorm))
// does not fall with tf and does not rise with the fieldnorm id, so at least that many documents score >= weight(t) * quotient(tf = 1, f)
// on term t alone — and, sums of non-negative f32 terms never being below a term, in any query of Should clauses that holds t.  The host
// turns the ids into a per-query floor under the k-th best score (bm25_index.cpp); a prefix of a list is enough for a bound, so long
// lists cost `cap` postings.  One wave per term, a 256-bin histogram in LDS.
__global__ __launch_bounds__(256) void bm25_term_floor_kernel(const unsigned long long *__restrict__ term_offsets, const uint32_t *__restrict__ words,
                                                              uint32_t n_terms, uint32_t cap, uint8_t *__restrict__ out) {
    __shared__ uint32_t hist_all[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *hist = hist_all[wave];
    constexpr uint32_t ranks[BM25_FLOOR_NR] = BM25_FLOOR_RANKS;
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)wave; t < n_terms; t += gridDim.x * 4u) {
        const unsigned long long b = term_offsets[t], e0 = term_offsets[t + 1];
        const unsigned long long e = e0 - b > cap ? b + cap : e0;
#pragma unroll
        for (int i = 0; i < 4; i++) hist[4 * lane + i] = 0u;
        asm volatile("" ::: "memory");
        for (unsigned long long p = b; p < e; p += 256u) {
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < 4; r++) w[r] = p + 64u * r + lane < e ? words[p + 64u * r + lane] : 0xffffffffu;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (p + 64u * r + lane < e) atomicAdd(&hist[w[r] >> 24], 1u);
        }
        asm volatile("" ::: "memory");
        uint32_t c[4], incl = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[i] = hist[4 * lane + i];
            incl += c[i];
        }
        const uint32_t own = incl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t excl = incl - own;
        uint32_t mine = 255u;
#pragma unroll
        for (int j = 0; j < BM25_FLOOR_NR; j++) {
            const uint32_t r = ranks[j];
            const unsigned long long m = __ballot(incl >= r);
            uint32_t f = 255u;
            if (m) {
                const int L = __ffsll((long long)m) - 1;
                uint32_t cum = excl, first = 3u;
                bool found = false;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    cum += c[i];
                    if (!found && cum >= r) {
                        first = (uint32_t)i;
                        found = true;
                    }
                }
                f = (uint32_t)__shfl((int)(4u * (uint32_t)lane + first), L, 64);
            }
            if (lane == j) mine = f;
        }
        if (lane < BM25_FLOOR_NR) out[(size_t)t * BM25_FLOOR_NR + lane] = (uint8_t)mine;
        asm volatile("" ::: "memory");
    }
}

hipError_t launch_bm25_term_floors(const unsigned long long *term_offsets, const uint32_t *words, uint32_t n_terms, uint32_t cap, uint8_t *out,
                                   hipStream_t s) {
    if (n_terms == 0) return hipSuccess;
    const uint32_t blocks = std::min<uint32_t>((n_terms + 3u) / 4u, 256u * 8u);
    hipLaunchKernelGGL(bm25_term_floor_kernel, dim3(blocks), dim3(256), 0, s, term_offsets, words, n_terms, cap, out);
    return hipGetLastError();
}

// ---- several tantivy segments as ONE resident posting layout -------------------------------------------------------------
// An index of S segments (the log-merge policy leaves several, nidx/src/settings.rs:246-253) is searched by tantivy segment by
// segment under one searcher.search (nidx_text/src/reader.rs:433-435).  Bm25Weight's statistics are searcher-wide, so a posting's
// score does not depend on the segment it lives in, and TopDocs breaks score ties by DocAddress = (segment_ord, doc) — which is the
// order of doc + base[segment] when base is the running sum of the segments' max_doc.  The resident layout is therefore term-major
// ACROSS the segments: term t's list = its list in segment 0, then in segment 1 (+ base[1]), ...; every kernel of the scorer then
// walks a multi-segment index exactly like one segment, in one launch, and the merge across segments is the slice merge.  This
// kernel moves segment s's postings to their place: posting j of the segment lies in the list of term t = the last t with
// seg_off[t] <= j and goes to dst_start[t] + (j - seg_off[t]).
__global__ __launch_bounds__(256) void bm25_concat_postings_kernel(const unsigned long long *__restrict__ seg_off, uint32_t n_terms,
                                                                   const unsigned long long *__restrict__ dst_start, const uint32_t *__restrict__ src_doc,
                                                                   const uint32_t *__restrict__ src_tf, unsigned long long n_post, uint32_t doc_base,
                                                                   uint32_t *__restrict__ dst_doc, uint32_t *__restrict__ dst_tf) {
    for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_post; j += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n_terms;   // the last t in [0, n_terms) with seg_off[t] <= j (seg_off[0] == 0)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (seg_off[mid] <= j) lo = mid;
            else hi = mid;
        }
        const unsigned long long at = dst_start[lo] + (j - seg_off[lo]);
        dst_doc[at] = src_doc[j] + doc_base;
        dst_tf[at] = src_tf[j];
    }
}

hipError_t launch_bm25_concat_postings(const unsigned long long *seg_off, uint32_t n_terms, const unsigned long long *dst_start, const uint32_t *src_doc,
                                       const uint32_t *src_tf, unsigned long long n_post, uint32_t doc_base, uint32_t *dst_doc, uint32_t *dst_tf,
                                       hipStream_t s) {
    if (n_post == 0 || n_terms == 0) return hipSuccess;
    const unsigned long long blocks = (n_post + 255ull) / 256ull;
    hipLaunchKernelGGL(bm25_concat_postings_kernel, dim3((uint32_t)(blocks < 65536ull ? blocks : 65536ull)), dim3(256), 0, s, seg_off, n_terms, dst_start,
                       src_doc, src_tf, n_post, doc_base, dst_doc, dst_tf);
    return hipGetLastError();
}

// ---- facet counts -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void facet_count_kernel(const unsigned long long *term_offsets, const uint32_t *doc_ids,
                                                          const uint32_t *pair_term, const int *pair_slot, const uint32_t *match_bits,
                                                          uint32_t match_words, unsigned long long *counts) {
    const uint32_t p = blockIdx.x;
    const int slot = pair_slot[p];
    if (slot < 0) return;
    const uint32_t *mb = match_bits + (size_t)slot * match_words;
    const unsigned long long b = term_offsets[pair_term[p]], e = term_offsets[pair_term[p] + 1];
    uint32_t c = 0;
    for (unsigned long long i = b + threadIdx.x; i < e; i += blockDim.x) {
        const uint32_t d = doc_ids[i];
        c += (mb[d >> 5] >> (d & 31)) & 1u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[p], (unsigned long long)c);
}

hipError_t launch_facet_count(const unsigned long long *term_offsets, const uint32_t *doc_ids, const uint32_t *pair_term,
                              const int *pair_slot, uint32_t n_pairs, const uint32_t *match_bits, uint32_t match_words,
                              unsigned long long *counts, hipStream_t s) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(facet_count_kernel, dim3(n_pairs), dim3(256), 0, s, term_offsets, doc_ids, pair_term, pair_slot, match_bits,
                       match_words, counts);
    return hipGetLastError();
}


// ---- prefilter (TextReaderService::prefilter, nidx_text/src/reader.rs:148-180) --------------------------------------
// RangeQuery over a fast field, on its dense ranks: bit d = rank_lo <= order_key[d] <= rank_hi.  One lane per document, the
// wave's ballot is the output word.
__global__ __launch_bounds__(256) void rank_range_bits_kernel(const uint32_t *order_key, uint32_t n_docs, uint32_t rank_lo,
                                                              uint32_t rank_hi, uint64_t *out) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = d < n_docs ? order_key[d] : 0u;  // ranks start at 1
    const unsigned long long m = __ballot(r >= rank_lo && r <= rank_hi);
    if ((threadIdx.x & 63) == 0 && (d >> 6) < ((n_docs + 63) >> 6)) out[d >> 6] = m;
}

hipError_t launch_rank_range_bits(const uint32_t *order_key, uint32_t n_docs, uint32_t rank_lo, uint32_t rank_hi, uint64_t *out,
                                  hipStream_t s) {
    if (n_docs == 0) return hipSuccess;
    hipLaunchKernelGGL(rank_range_bits_kernel, dim3((n_docs + 255) / 256), dim3(256), 0, s, order_key, n_docs, rank_lo, rank_hi, out);
    return hipGetLastError();
}

// PhraseQuery as a filter: the driver term's documents whose phrase frequency (phrase_match_kernel) is non-zero.
__global__ __launch_bounds__(256) void phrase_bits_kernel(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph,
                                                          const uint32_t *tmp_tf, unsigned int *bits) {
    const unsigned long long b0 = term_offsets[ph.terms[ph.driver]], e0 = term_offsets[ph.terms[ph.driver] + 1];
    const unsigned long long i = b0 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e0 || tmp_tf[i - b0] == 0) return;
    const uint32_t d = doc_ids[i];
    atomicOr(&bits[d >> 5], 1u << (d & 31));
}

hipError_t launch_phrase_bits(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph, uint32_t n_driver,
                              const uint32_t *tmp_tf, uint64_t *bits, hipStream_t s) {
    if (n_driver == 0) return hipSuccess;
    hipLaunchKernelGGL(phrase_bits_kernel, dim3((n_driver + 255) / 256), dim3(256), 0, s, term_offsets, doc_ids, ph, tmp_tf,
                       reinterpret_cast<unsigned int *>(bits));
    return hipGetLastError();
}

// bitset -> ascending DocAddress list (segment << 32 | doc) in three small launches: per-block popcounts, one block scanning
// them, then every block emits its slice at its base.  A block owns 256 words = 16384 documents.
__global__ __launch_bounds__(256) void docaddr_count_kernel(const uint64_t *bits, uint32_t n_words, uint32_t *block_counts) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    uint32_t c = w < n_words ? (uint32_t)__popcll(bits[w]) : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// exclusive scan of block_counts in place (one block); total -> *total
__global__ __launch_bounds__(256) void docaddr_scan_kernel(uint32_t *block_counts, uint32_t n_blocks, unsigned long long *total) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t base_s;
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_blocks; i0 += 256) {
        const uint32_t i = i0 + (uint32_t)tid;
        const uint32_t c = i < n_blocks ? block_counts[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_sum[wib] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int w = 0; w < wib; w++) before += wave_sum[w];
        if (i < n_blocks) block_counts[i] = before + incl - c;
        __syncthreads();
        if (tid == 0) base_s += wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    if (tid == 0) *total = base_s;
}

__global__ __launch_bounds__(256) void docaddr_emit_kernel(const uint64_t *bits, uint32_t n_words, const uint32_t *block_base,
                                                           uint64_t segment_hi, unsigned long long out_begin, unsigned long long out_cap,
                                                           uint64_t *out) {
    __shared__ uint32_t wave_sum[4];
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const uint32_t w = blockIdx.x * 256u + (uint32_t)tid;
    uint64_t x = w < n_words ? bits[w] : 0ull;
    const uint32_t c = (uint32_t)__popcll(x);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_sum[wib] = incl;
    __syncthreads();
    unsigned long long at = out_begin + block_base[blockIdx.x] + incl - c;
    for (int i = 0; i < wib; i++) at += wave_sum[i];
    while (x) {
        const int b = __ffsll((long long)x) - 1;
        x &= x - 1;
        if (at < out_cap) out[at] = segment_hi | (uint64_t)(w * 64u + (uint32_t)b);
        at++;
    }
}

hipError_t launch_bitset_to_docaddr(const uint64_t *bits, uint32_t n_words, uint32_t segment, uint32_t *block_scratch,
                                    unsigned long long *total, unsigned long long out_begin, unsigned long long out_cap, uint64_t *out,
                                    hipStream_t s) {
    if (n_words == 0) return hipMemsetAsync(total, 0, 8, s);
    const uint32_t n_blocks = (n_words + 255) / 256;
    hipLaunchKernelGGL(docaddr_count_kernel, dim3(n_blocks), dim3(256), 0, s, bits, n_words, block_scratch);
    hipLaunchKernelGGL(docaddr_scan_kernel, dim3(1), dim3(256), 0, s, block_scratch, n_blocks, total);
    hipLaunchKernelGGL(docaddr_emit_kernel, dim3(n_blocks), dim3(256), 0, s, bits, n_words, block_scratch, (uint64_t)segment << 32, out_begin,
                       out_cap, out);
    return hipGetLastError();
}

}  // namespace nidx
