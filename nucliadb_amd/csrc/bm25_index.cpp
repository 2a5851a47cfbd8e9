// bm25_index.cpp — C ABI implementation of the BM25 index (include/nidx_gpu.h "BM25 index").
//
// Host side of tantivy's Bm25Weight [third party, restated; tantivy 0.26.1 query/bm25.rs,
// fieldnorm/code.rs]: K1 = 1.2, B = 0.75; idf = ln(1 + (N - n + 0.5)/(n + 0.5));
// weight = idf * (1 + K1) * boost; tf-norm cache[id] = K1 * (1 - B + B * fieldnorm(id) / avg);
// statistics are searcher-wide: N = sum of segment max_doc, n = sum of segment doc_freq,
// avg = sum of tokens / N — and are not reduced by deletions.  Scoring itself is bm25.hip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <chrono>
#include <memory>
#include <mutex>
#include <shared_mutex>

#include "device_common.h"
#include "host_common.h"
#include "kernels.h"

namespace nidx {

static const float kK1 = 1.2f, kB = 0.75f;

static uint32_t fieldnorm_from_id(uint8_t id) {
    // FIELD_NORMS_TABLE: 0..=40 exact, then eight values per octave with the step doubling
    if (id <= 40) return id;
    uint32_t v = 40;
    for (uint32_t i = 41; i <= id; i++) v += 2u << ((i - 41) / 8);
    return v;
}

static uint8_t fieldnorm_to_id(uint32_t fieldnorm) {
    // the largest id whose value is <= fieldnorm
    int lo = 0, hi = 255;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (fieldnorm_from_id((uint8_t)mid) <= fieldnorm) lo = mid;
        else hi = mid - 1;
    }
    return (uint8_t)lo;
}

static float bm25_idf(uint64_t doc_freq, uint64_t doc_count) {
    float x = ((float)(doc_count - doc_freq) + 0.5f) / ((float)doc_freq + 0.5f);
    return logf(1.0f + x);
}

struct Bm25Segment {
    uint32_t n_docs = 0, n_terms = 0;
    std::vector<uint64_t> term_offsets_host;
    DevBuf term_offsets, doc_ids, tfs, fieldnorm_ids, alive;
    DevBuf pos_offsets, positions;  // positions of every posting (phrase queries); absent when the caller gave none
    bool all_alive = true;
    // fast fields (created, modified): host values + their dense ranks in HBM (the kernel orders by rank)
    std::vector<int64_t> fast_host[2];
    std::vector<int64_t> fast_uniq[2];  // the distinct values, ascending: value -> rank for range filters
    DevBuf order_key[2];
    int64_t n_alive = -1;  // live documents, counted on first use
    uint64_t bytes() const {
        return term_offsets.bytes + doc_ids.bytes + tfs.bytes + fieldnorm_ids.bytes + alive.bytes + order_key[0].bytes + order_key[1].bytes +
               pos_offsets.bytes + positions.bytes;
    }
};

// Everything one search call writes: its stream and events, device scratch, pinned staging and the host planning vectors (kept for
// their capacity).  The index has one for the blocking entries; every pipeline slot of nidx_gpu_bm25_search_submit / _wait has its own,
// so several submitting threads prepare and launch their batches side by side (the segments they read are immutable while they do).
struct Bm25Ctx {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket the scoring kernels on `stream`
    std::vector<uint32_t> w_item_first, w_item_list;
    std::vector<Bm25Work> w_work;
    std::vector<uint8_t> w_q_union;
    std::vector<uint64_t> w_postings, w_clause_len;
    std::vector<Bm25ClauseDev> w_dev_clauses;
    std::vector<Bm25UClause> w_ucl;   // (entries of queries that are not union queries keep whatever they held: never read)
    std::vector<Bm25AfterDev> w_after;
    DevBuf s_after, s_count, s_total, s_postings, s_key;
    DevBuf s_fuse_done, s_fuse_key, s_fuse_count, s_fuse_total, s_fuse_postings;   // Bm25FusedMerge (s_fuse_done is zeroed when it grows and stays zero between launches)
    DevBuf s_phrase_tf, s_aux_tfs, s_slop_left, s_sub_bits, s_sub_union;
    DevBuf s_set_terms, s_set_bits, s_aux_off, s_aux_out_off, s_aux_ids, s_set_counts, s_match_bits, s_match_slot, s_pair_term, s_pair_slot,
        s_facet_counts;
    // packed staging: [clauses | clause offsets], [item_first | work list] in; [doc | score | count | total | postings] out
    DevBuf s_in_q, s_in_w, s_outpack;
    PinBuf h_in_q, h_in_w, h_outpack;
    float kernel_ms = 0.f;   // scoring kernels of the last call through this context
    Bm25Ctx() = default;
    Bm25Ctx(const Bm25Ctx &) = delete;
    hipError_t init() {
        // BM25 launches are short (~0.07 ms) and their callers wait for them; when they share the device with HNSW batches in flight
        // (the hybrid request: serving.cpp keeps several on their own streams) they should not queue behind those: highest priority
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        // NIDX_GPU_BM25_PRIORITY=0: default-priority streams (measurement: the runtime keeps its hardware queues per priority, and
        // streams that share a hardware queue serialise)
        if (const char *pe = getenv("NIDX_GPU_BM25_PRIORITY"))
            if (atoi(pe) == 0) hi = 0;
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi);
        if (e == hipSuccess) e = hipEventCreate(&ev0);
        if (e == hipSuccess) e = hipEventCreate(&ev1);
        return e;
    }
    // also the clean-up of an open that failed half way
    ~Bm25Ctx() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

// One batch in flight of nidx_gpu_bm25_search_submit / _wait.
struct Bm25Slot {
    bool busy = false, preparing = false, launched = false;
    uint64_t ticket = 0;
    Bm25Ctx cx;
    // layout of cx.h_outpack for the collect step
    uint32_t nq = 0, k = 0, kk = 0;
    size_t o_doc = 0, o_score = 0, o_count = 0, o_total = 0, o_post = 0, o_seg = 0;
    // a request the pipeline does not cover (term sets, phrases, nested queries, facets, order by a field) runs synchronously inside
    // submit; its results wait here
    std::vector<uint64_t> r_docaddr, r_total, r_postings;
    std::vector<float> r_score;
    std::vector<uint32_t> r_count;
};

// What is kept of every segment the caller opened when the resident layout is the term-major concatenation (Bm25Index::seg_base)
struct Bm25RealSegment {
    uint32_t n_docs = 0;
    std::vector<uint64_t> term_offsets_host;
    std::vector<int64_t> fast_host[2];
    bool has_fast[2] = {false, false};
};

struct Bm25Index {
    int device = 0;
    int n_cus = 256;   // compute units of the device (the work-item budget of one launch round follows it)
    // mu: the blocking entries (they share `main`) and the mutators; rw: searches hold it shared, mutators (deletions, fast fields)
    // exclusively; slots_mu: the slot table and tickets; ms_mu: last_kernel_ms
    std::mutex mu, slots_mu, ms_mu;
    std::shared_mutex rw;
    // The resident segments.  One per opened segment — or, for an index of several segments, ONE: the term-major concatenation over
    // doc + seg_base[segment] (bm25_aux.hip: bm25_concat_postings_kernel), which every kernel walks like a single segment.
    std::vector<Bm25Segment> segs;
    std::vector<Bm25RealSegment> real;     // non-empty <=> concatenated
    std::vector<uint32_t> seg_base;        // [n_real + 1] running sum of the real segments' n_docs
    DevBuf d_seg_base;                     // the same in HBM: bm25_merge_kernel turns resident docs into (segment, doc)
    uint32_t n_segments = 0;               // segments the caller opened
    uint64_t total_docs = 0, total_tokens = 0;
    uint32_t n_terms = 0;
    std::vector<float> idf_of_term;   // Bm25Weight's idf per term over all segments (filled at open)
    // Score floors (bm25_aux.hip: bm25_term_floor_kernel): floor_fn[t][j] = fieldnorm id under which >= BM25_FLOOR_RANKS[j] postings of term t
    // lie; quot1[id] = 1 / (1 + K(id)), the tf = 1 row of tf_cache.  Empty = no floors (several resident layouts, a quotient row that is not
    // monotone, NIDX_GPU_BM25_FLOOR=0).
    std::vector<uint8_t> floor_fn;
    float quot1[256] = {};
    Bm25Ctx main;
    DevBuf tf_cache;
    // term dictionary (fuzzy expansion) and the scratch of the collectors (under mu, on main.stream)
    DevBuf dict_bytes, dict_offsets, s_fuzzy_q, s_fuzzy_flags;
    bool has_dict = false;
    DevBuf s_pf_stack, s_pf_lists, s_pf_result, s_pf_blocks, s_pf_total, s_pf_out;  // prefilter
    float last_kernel_ms = 0.f;
    std::vector<std::unique_ptr<Bm25Slot>> slots;   // nidx_gpu_bm25_search_submit / _wait
    uint64_t next_ticket = 1;
    Bm25Index() = default;
    Bm25Index(const Bm25Index &) = delete;
    bool concatenated() const { return !real.empty(); }
    // DocAddress of resident (segment, doc)
    uint64_t docaddr(size_t resident_segment, uint32_t d) const {
        if (real.empty()) return ((uint64_t)resident_segment << 32) | d;
        const size_t s = (size_t)(std::upper_bound(seg_base.begin() + 1, seg_base.end(), d) - (seg_base.begin() + 1));
        return ((uint64_t)s << 32) | (uint64_t)(d - seg_base[s]);
    }
};

}  // namespace nidx

using namespace nidx;

extern "C" {

float nidx_gpu_bm25_idf(uint64_t doc_freq, uint64_t doc_count) { return bm25_idf(doc_freq, doc_count); }
uint32_t nidx_gpu_fieldnorm_from_id(uint8_t id) { return fieldnorm_from_id(id); }
uint8_t nidx_gpu_fieldnorm_to_id(uint32_t fieldnorm) { return fieldnorm_to_id(fieldnorm); }

// the host-side checks of one opened segment (the kernels index the fieldnorm table and the alive bitset with doc ids unchecked)
static int32_t bm25_check_segment(const nidx_gpu_bm25_segment_t &in, uint32_t s) {
    if (!in.term_offsets || (in.n_docs && !in.fieldnorm_ids)) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: NULL arrays", s);
    const uint64_t n_post = in.term_offsets[in.n_terms];
    if (n_post && (!in.doc_ids || !in.tfs)) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: NULL postings", s);
    if (in.term_offsets[0] != 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: term_offsets[0] != 0", s);
    for (uint32_t t = 0; t < in.n_terms; t++)
        if (in.term_offsets[t + 1] < in.term_offsets[t]) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: term_offsets decrease at term %u", s, t);
    uint32_t max_doc = 0;
    for (uint64_t i = 0; i < n_post; i++) max_doc = std::max(max_doc, in.doc_ids[i]);
    if (n_post && max_doc >= in.n_docs) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: posting doc id %u >= n_docs %u", s, max_doc, in.n_docs);
    if (in.pos_offsets && n_post && in.pos_offsets[n_post] && !in.positions) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: pos_offsets without positions", s);
    return NIDX_OK;
}

// resident posting word = tf | fieldnorm id << 24: the scorer reads the fieldnorm with the posting instead of one random cache line
// per posting (bm25_aux.hip)
static int32_t bm25_pack_words(Bm25Index *idx, Bm25Segment &seg, uint64_t n_post, uint32_t s) {
    if (!n_post) return NIDX_OK;
    hipStream_t st = idx->main.stream;
    DevBuf flag;
    NIDX_HIP(flag.alloc(4));
    NIDX_HIP(hipMemsetAsync(flag.p, 0, 4, st));
    NIDX_HIP(launch_bm25_pack_fieldnorm(seg.doc_ids.as<uint32_t>(), seg.tfs.as<uint32_t>(), seg.fieldnorm_ids.as<uint8_t>(), n_post, seg.n_docs,
                                        flag.as<uint32_t>(), st));
    uint32_t f = 0;
    NIDX_HIP(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, st));
    NIDX_HIP(hipStreamSynchronize(st));
    if (f & 1u) return fail(NIDX_ERR_UNSUPPORTED, "segment %u: a term frequency >= 2^24 does not fit the resident posting word", s);
    if (f & 2u) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u: a posting's doc id is >= n_docs", s);
    return NIDX_OK;
}

// one opened segment -> one resident segment
static int32_t bm25_upload_segment(Bm25Index *idx, const nidx_gpu_bm25_segment_t &in, Bm25Segment &seg, uint32_t s) {
    seg.n_docs = in.n_docs;
    seg.n_terms = in.n_terms;
    seg.term_offsets_host.assign(in.term_offsets, in.term_offsets + in.n_terms + 1);
    const uint64_t n_post = seg.term_offsets_host[in.n_terms];
    NIDX_HIP(seg.term_offsets.alloc((size_t)(in.n_terms + 1) * 8));
    NIDX_HIP(hipMemcpy(seg.term_offsets.p, in.term_offsets, (size_t)(in.n_terms + 1) * 8, hipMemcpyHostToDevice));
    NIDX_HIP(seg.doc_ids.alloc(std::max<size_t>(n_post, 1) * 4 + BM25_LIST_PAD_BYTES));
    NIDX_HIP(seg.tfs.alloc(std::max<size_t>(n_post, 1) * 4 + BM25_LIST_PAD_BYTES));
    if (n_post) {
        NIDX_HIP(hipMemcpy(seg.doc_ids.p, in.doc_ids, n_post * 4, hipMemcpyHostToDevice));
        NIDX_HIP(hipMemcpy(seg.tfs.p, in.tfs, n_post * 4, hipMemcpyHostToDevice));
    }
    NIDX_HIP(seg.fieldnorm_ids.alloc(std::max<size_t>(in.n_docs, 1)));
    if (in.n_docs) NIDX_HIP(hipMemcpy(seg.fieldnorm_ids.p, in.fieldnorm_ids, in.n_docs, hipMemcpyHostToDevice));
    if (int32_t rc = bm25_pack_words(idx, seg, n_post, s)) return rc;
    seg.all_alive = in.alive_bitset == nullptr;
    if (in.alive_bitset) {
        size_t words = ((size_t)in.n_docs + 63) / 64;
        NIDX_HIP(seg.alive.alloc(std::max<size_t>(words, 1) * 8));
        if (words) NIDX_HIP(hipMemcpy(seg.alive.p, in.alive_bitset, words * 8, hipMemcpyHostToDevice));
    }
    if (in.pos_offsets && n_post) {
        const uint64_t n_pos = in.pos_offsets[n_post];
        NIDX_HIP(seg.pos_offsets.alloc((size_t)(n_post + 1) * 8));
        NIDX_HIP(hipMemcpy(seg.pos_offsets.p, in.pos_offsets, (size_t)(n_post + 1) * 8, hipMemcpyHostToDevice));
        NIDX_HIP(seg.positions.alloc(std::max<uint64_t>(n_pos, 1) * 4));
        if (n_pos) NIDX_HIP(hipMemcpy(seg.positions.p, in.positions, n_pos * 4, hipMemcpyHostToDevice));
    }
    return NIDX_OK;
}

// S opened segments -> ONE resident segment, term-major across the segments over doc + seg_base[segment] (see
// bm25_concat_postings_kernel).  The segments' postings pass through device scratch one segment at a time.
static int32_t bm25_upload_concatenated(Bm25Index *idx, const nidx_gpu_bm25_segment_t *segments, uint32_t n_segments) {
    const uint32_t T = idx->n_terms;
    hipStream_t st = idx->main.stream;
    idx->real.resize(n_segments);
    idx->seg_base.assign(n_segments + 1, 0);
    uint64_t n_docs_all = 0, n_post_all = 0;
    bool any_dead = false, all_pos = true;
    for (uint32_t s = 0; s < n_segments; s++) {
        const nidx_gpu_bm25_segment_t &in = segments[s];
        idx->real[s].n_docs = in.n_docs;
        idx->real[s].term_offsets_host.assign(in.term_offsets, in.term_offsets + T + 1);
        n_docs_all += in.n_docs;
        if (n_docs_all > 0xffffffffull)
            return fail(NIDX_ERR_UNSUPPORTED, "the segments of one index hold more than 2^32 - 1 documents together (doc ids are 32-bit on the device)");
        idx->seg_base[s + 1] = (uint32_t)n_docs_all;
        n_post_all += in.term_offsets[T];
        any_dead |= in.alive_bitset != nullptr;
        if (in.term_offsets[T] && !in.pos_offsets) all_pos = false;   // phrases need the positions of every segment
    }
    NIDX_HIP(idx->d_seg_base.alloc((size_t)(n_segments + 1) * 4));
    NIDX_HIP(hipMemcpy(idx->d_seg_base.p, idx->seg_base.data(), (size_t)(n_segments + 1) * 4, hipMemcpyHostToDevice));
    idx->segs.resize(1);
    Bm25Segment &v = idx->segs[0];
    v.n_docs = (uint32_t)n_docs_all;
    v.n_terms = T;
    std::vector<uint64_t> &voff = v.term_offsets_host;
    voff.assign((size_t)T + 1, 0);
    for (uint32_t s = 0; s < n_segments; s++) {
        const uint64_t *o = segments[s].term_offsets;
        for (uint32_t t = 0; t <= T; t++) voff[t] += o[t];
    }
    NIDX_HIP(v.term_offsets.alloc((size_t)(T + 1) * 8));
    NIDX_HIP(hipMemcpy(v.term_offsets.p, voff.data(), (size_t)(T + 1) * 8, hipMemcpyHostToDevice));
    NIDX_HIP(v.doc_ids.alloc(std::max<size_t>(n_post_all, 1) * 4 + BM25_LIST_PAD_BYTES));
    NIDX_HIP(v.tfs.alloc(std::max<size_t>(n_post_all, 1) * 4 + BM25_LIST_PAD_BYTES));
    NIDX_HIP(v.fieldnorm_ids.alloc(std::max<size_t>(v.n_docs, 1)));
    // where segment s's part of term t's list begins: the list's start + the lengths of the earlier segments' parts
    std::vector<unsigned long long> cursor((size_t)std::max<uint32_t>(T, 1));
    for (uint32_t t = 0; t < T; t++) cursor[t] = voff[t];
    std::vector<unsigned long long> dst_start((size_t)std::max<uint32_t>(T, 1));
    DevBuf t_off, t_dst, t_doc, t_tf;
    NIDX_HIP(t_off.alloc((size_t)(T + 1) * 8));
    NIDX_HIP(t_dst.alloc((size_t)std::max<uint32_t>(T, 1) * 8));
    const bool with_pos = all_pos && n_post_all > 0;
    std::vector<uint64_t> vpos_off;
    std::vector<uint32_t> vpos;
    if (with_pos) {
        uint64_t n_pos_all = 0;
        for (uint32_t s = 0; s < n_segments; s++)
            if (segments[s].term_offsets[T]) n_pos_all += segments[s].pos_offsets[segments[s].term_offsets[T]];
        vpos_off.assign((size_t)n_post_all + 1, 0);
        vpos.resize((size_t)n_pos_all);
    }
    for (uint32_t s = 0; s < n_segments; s++) {
        const nidx_gpu_bm25_segment_t &in = segments[s];
        const uint64_t *o = in.term_offsets;
        const uint64_t n_post = o[T];
        for (uint32_t t = 0; t < T; t++) {
            dst_start[t] = cursor[t];
            cursor[t] += o[t + 1] - o[t];
        }
        if (in.n_docs) NIDX_HIP(hipMemcpy(v.fieldnorm_ids.as<uint8_t>() + idx->seg_base[s], in.fieldnorm_ids, in.n_docs, hipMemcpyHostToDevice));
        if (!n_post) continue;
        NIDX_HIP(t_doc.reserve(n_post * 4));
        NIDX_HIP(t_tf.reserve(n_post * 4));
        NIDX_HIP(hipMemcpyAsync(t_off.p, o, (size_t)(T + 1) * 8, hipMemcpyHostToDevice, st));
        NIDX_HIP(hipMemcpyAsync(t_dst.p, dst_start.data(), (size_t)T * 8, hipMemcpyHostToDevice, st));
        NIDX_HIP(hipMemcpyAsync(t_doc.p, in.doc_ids, n_post * 4, hipMemcpyHostToDevice, st));
        NIDX_HIP(hipMemcpyAsync(t_tf.p, in.tfs, n_post * 4, hipMemcpyHostToDevice, st));
        NIDX_HIP(launch_bm25_concat_postings(t_off.as<unsigned long long>(), T, t_dst.as<unsigned long long>(), t_doc.as<uint32_t>(), t_tf.as<uint32_t>(),
                                             n_post, idx->seg_base[s], v.doc_ids.as<uint32_t>(), v.tfs.as<uint32_t>(), st));
        NIDX_HIP(hipStreamSynchronize(st));   // dst_start and the scratch are rewritten for the next segment
        if (with_pos) {
            // positions follow their postings: the positions of one (term, segment) run are contiguous in the source; the lengths are
            // laid down here and summed below
            for (uint32_t t = 0; t < T; t++)
                for (uint64_t j = o[t]; j < o[t + 1]; j++) vpos_off[dst_start[t] + (j - o[t]) + 1] = in.pos_offsets[j + 1] - in.pos_offsets[j];
        }
    }
    if (with_pos) {
        for (uint64_t i = 0; i < n_post_all; i++) vpos_off[i + 1] += vpos_off[i];
        for (uint32_t t = 0; t < T; t++) cursor[t] = voff[t];
        for (uint32_t s = 0; s < n_segments; s++) {
            const nidx_gpu_bm25_segment_t &in = segments[s];
            const uint64_t *o = in.term_offsets;
            for (uint32_t t = 0; t < T; t++) {
                const uint64_t len = o[t + 1] - o[t];
                if (len) {
                    const uint64_t p0 = in.pos_offsets[o[t]], p1 = in.pos_offsets[o[t + 1]];
                    if (p1 > p0) memcpy(vpos.data() + vpos_off[cursor[t]], in.positions + p0, (size_t)(p1 - p0) * 4);
                }
                cursor[t] += len;
            }
        }
        NIDX_HIP(v.pos_offsets.alloc((size_t)(n_post_all + 1) * 8));
        NIDX_HIP(hipMemcpy(v.pos_offsets.p, vpos_off.data(), (size_t)(n_post_all + 1) * 8, hipMemcpyHostToDevice));
        NIDX_HIP(v.positions.alloc(std::max<uint64_t>(vpos.size(), 1) * 4));
        if (!vpos.empty()) NIDX_HIP(hipMemcpy(v.positions.p, vpos.data(), vpos.size() * 4, hipMemcpyHostToDevice));
    }
    if (int32_t rc = bm25_pack_words(idx, v, n_post_all, 0)) return rc;
    v.all_alive = !any_dead;
    if (any_dead) {
        // the segments' alive bitsets, each shifted to its doc base
        const size_t words = ((size_t)v.n_docs + 63) / 64;
        std::vector<uint64_t> bits(std::max<size_t>(words, 1) + 1, 0);   // + 1: the spill word of the last shift
        for (uint32_t s = 0; s < n_segments; s++) {
            const nidx_gpu_bm25_segment_t &in = segments[s];
            const uint32_t n = in.n_docs;
            for (uint32_t i = 0; i < (n + 63) / 64; i++) {
                uint64_t w = in.alive_bitset ? in.alive_bitset[i] : ~0ull;
                const uint32_t left = n - i * 64;
                if (left < 64) w &= (1ull << left) - 1ull;
                const uint64_t bit = (uint64_t)idx->seg_base[s] + (uint64_t)i * 64;
                const uint32_t sh = (uint32_t)(bit & 63);
                bits[bit >> 6] |= w << sh;
                if (sh) bits[(bit >> 6) + 1] |= w >> (64 - sh);
            }
        }
        NIDX_HIP(v.alive.alloc(std::max<size_t>(words, 1) * 8));
        if (words) NIDX_HIP(hipMemcpy(v.alive.p, bits.data(), words * 8, hipMemcpyHostToDevice));
    }
    return NIDX_OK;
}

int32_t nidx_gpu_bm25_open(const nidx_gpu_bm25_segment_t *segments, uint32_t n_segments, nidx_gpu_bm25_index_t **index_out) try {
    if (!index_out || (n_segments && !segments)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *index_out = nullptr;
    std::unique_ptr<Bm25Index> idx(new Bm25Index());
    NIDX_HIP(hipGetDevice(&idx->device));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, idx->device) == hipSuccess && cus > 0) idx->n_cus = cus;
    }
    NIDX_HIP(idx->main.init());
    idx->n_segments = n_segments;
    for (uint32_t s = 0; s < n_segments; s++) {
        const nidx_gpu_bm25_segment_t &in = segments[s];
        if (s > 0 && in.n_terms != idx->n_terms)
            return fail(NIDX_ERR_INVALID_ARGUMENT, "every segment must use the same term-id space (n_terms differs)");
        idx->n_terms = in.n_terms;
        if (int32_t rc = bm25_check_segment(in, s)) return rc;
        idx->total_docs += in.n_docs;
        idx->total_tokens += in.total_num_tokens;
    }
    // NIDX_GPU_BM25_SEGMENT_LOOP=1 keeps one resident segment per opened segment and the search a loop over them (launches, a
    // transfer and a host merge per segment): the path the one-launch layout is tested against
    const char *loop_env = getenv("NIDX_GPU_BM25_SEGMENT_LOOP");
    // Segments that disagree on positions (some indexed WithFreqsAndPositions, some not) keep their own resident layouts too: the one
    // layout would have to drop the positions of ALL of them, and a phrase over a field only the positioned segments hold would fail.
    bool some_pos = false, some_without = false;
    for (uint32_t s = 0; s < n_segments; s++) {
        if (!segments[s].term_offsets[idx->n_terms]) continue;   // (no postings: nothing to disagree about)
        if (segments[s].pos_offsets) some_pos = true;
        else some_without = true;
    }
    const bool mixed_positions = some_pos && some_without;
    if (n_segments > 1 && !(loop_env && atoi(loop_env) != 0) && !mixed_positions) {
        if (int32_t rc = bm25_upload_concatenated(idx.get(), segments, n_segments)) return rc;
    } else {
        idx->segs.resize(n_segments);
        for (uint32_t s = 0; s < n_segments; s++)
            if (int32_t rc = bm25_upload_segment(idx.get(), segments[s], idx->segs[s], s)) return rc;
    }
    // Bm25Weight's idf of every term from the searcher-wide statistics
    idx->idf_of_term.resize(idx->n_terms);
    for (uint32_t t = 0; t < idx->n_terms; t++) {
        uint64_t df = 0;
        for (const Bm25Segment &sg : idx->segs) df += sg.term_offsets_host[t + 1] - sg.term_offsets_host[t];
        idx->idf_of_term[t] = bm25_idf(df, idx->total_docs);
    }
    float avg = idx->total_docs ? (float)idx->total_tokens / (float)idx->total_docs : 0.0f;
    // [0, 256): K1 * (1 - B + B * fieldnorm / avg) per fieldnorm id; [256 t, 256 t + 256), t = 1 .. 3: the quotient tf / (tf + that) for
    // tf = t — the same two correctly rounded f32 operations the kernels would make per posting (bm25_stream.hip reads the four rows)
    float cache[4 * 256];
    for (int id = 0; id < 256; id++) {
        float fieldnorm = (float)fieldnorm_from_id((uint8_t)id);
        cache[id] = kK1 * (1.0f - kB + kB * fieldnorm / avg);
        for (int t = 1; t <= 3; t++) cache[256 * t + id] = (float)t / ((float)t + cache[id]);
    }
    NIDX_HIP(idx->tf_cache.alloc(sizeof(cache)));
    NIDX_HIP(hipMemcpy(idx->tf_cache.p, cache, sizeof(cache), hipMemcpyHostToDevice));
    // Per-term score floors for the streaming scorer (one resident layout only).  They rest on the quotient being monotone: not below
    // the tf = 1 row for larger frequencies, not rising with the fieldnorm id — checked on the very table the kernels read.
    {
        bool monotone = idx->total_docs > 0 && avg > 0.0f;
        for (int id = 0; id < 256 && monotone; id++) {
            idx->quot1[id] = cache[256 + id];
            if (!(cache[256 + id] > 0.0f) || !(cache[512 + id] >= cache[256 + id]) || !(cache[768 + id] >= cache[512 + id])) monotone = false;
            if (id > 0 && !(cache[256 + id] <= cache[256 + id - 1])) monotone = false;
            if (id > 0 && !(cache[id] >= cache[id - 1])) monotone = false;   // K(fieldnorm id): what the division of tf > 3 adds to tf
        }
        const char *fe = getenv("NIDX_GPU_BM25_FLOOR");
        if (monotone && idx->segs.size() == 1 && idx->n_terms && !(fe && atoi(fe) == 0)) {
            Bm25Segment &sg = idx->segs[0];
            DevBuf d_floor;
            const size_t bytes = (size_t)idx->n_terms * BM25_FLOOR_NR;
            NIDX_HIP(d_floor.alloc(bytes));
            NIDX_HIP(launch_bm25_term_floors(sg.term_offsets.as<unsigned long long>(), sg.tfs.as<uint32_t>(), idx->n_terms, 1u << 16, d_floor.as<uint8_t>(),
                                             idx->main.stream));
            idx->floor_fn.resize(bytes);
            NIDX_HIP(hipMemcpyAsync(idx->floor_fn.data(), d_floor.p, bytes, hipMemcpyDeviceToHost, idx->main.stream));
            NIDX_HIP(hipStreamSynchronize(idx->main.stream));
        }
    }
    *index_out = reinterpret_cast<nidx_gpu_bm25_index_t *>(idx.release());
    return NIDX_OK;
} NIDX_ABI_CATCH

void nidx_gpu_bm25_close(nidx_gpu_bm25_index_t *index) {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    delete idx;
}

int32_t nidx_gpu_bm25_last_kernel_ms(const nidx_gpu_bm25_index_t *index, float *ms_out) try {
    Bm25Index *idx = const_cast<Bm25Index *>(reinterpret_cast<const Bm25Index *>(index));
    if (!idx || !ms_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->ms_mu);
    *ms_out = idx->last_kernel_ms;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_space_usage(const nidx_gpu_bm25_index_t *index, uint64_t *bytes_out) try {
    const Bm25Index *idx = reinterpret_cast<const Bm25Index *>(index);
    if (!idx || !bytes_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    uint64_t b = idx->tf_cache.bytes;
    for (const Bm25Segment &s : idx->segs) b += s.bytes();
    *bytes_out = b;
    return NIDX_OK;
} NIDX_ABI_CATCH

// Before an exclusive operation rewrites what the scoring kernels read (alive bitsets, fast-field ranks): the batches already submitted
// finish on the state they were planned against — a reopened searcher is a snapshot (nidx_tantivy/src/index_reader.rs:39-74).  The caller
// holds idx->rw exclusively, so no submit is between its planning and its launches.
static int32_t bm25_drain_tickets(Bm25Index *idx) {
    std::lock_guard<std::mutex> lock(idx->slots_mu);
    for (auto &s : idx->slots) NIDX_HIP(hipStreamSynchronize(s->cx.stream));
    return NIDX_OK;
}

// dense ranks of a fast field (equal values <=> equal ranks, order preserved) to HBM; the kernels order by rank
static int32_t bm25_upload_fast_field(Bm25Segment &seg, uint32_t field) {
    const std::vector<int64_t> &values = seg.fast_host[field];
    std::vector<int64_t> uniq(values);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    std::vector<uint32_t> rank(seg.n_docs);
    for (uint32_t d = 0; d < seg.n_docs; d++)
        rank[d] = (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), values[d]) - uniq.begin()) + 1u;
    NIDX_HIP(seg.order_key[field].alloc(std::max<size_t>(seg.n_docs, 1) * 4));
    if (seg.n_docs) NIDX_HIP(hipMemcpy(seg.order_key[field].p, rank.data(), (size_t)seg.n_docs * 4, hipMemcpyHostToDevice));
    seg.fast_uniq[field] = std::move(uniq);
    return NIDX_OK;
}

int32_t nidx_gpu_bm25_set_fast_field(nidx_gpu_bm25_index_t *index, uint32_t segment, uint32_t field, const int64_t *values) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || segment >= idx->n_segments || field > 1 || !values) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad fast field");
    std::lock_guard<std::mutex> lock(idx->mu);
    std::unique_lock<std::shared_mutex> wlock(idx->rw);
    NIDX_HIP(hipSetDevice(idx->device));
    if (int32_t rc_drain = bm25_drain_tickets(idx)) return rc_drain;
    NIDX_HIP(hipSetDevice(idx->device));
    if (!idx->concatenated()) {
        Bm25Segment &seg = idx->segs[segment];
        seg.fast_host[field].assign(values, values + seg.n_docs);
        return bm25_upload_fast_field(seg, field);
    }
    // concatenated layout: the ranks are taken over the values of ALL segments (order_by_fast_field compares values across
    // segments), once every segment has registered the field; until then the field counts as not registered
    Bm25RealSegment &rs = idx->real[segment];
    rs.fast_host[field].assign(values, values + rs.n_docs);
    rs.has_fast[field] = true;
    Bm25Segment &v = idx->segs[0];
    for (const Bm25RealSegment &r : idx->real)
        if (!r.has_fast[field]) {
            v.order_key[field].release();
            return NIDX_OK;
        }
    v.fast_host[field].clear();
    v.fast_host[field].reserve(v.n_docs);
    for (const Bm25RealSegment &r : idx->real) v.fast_host[field].insert(v.fast_host[field].end(), r.fast_host[field].begin(), r.fast_host[field].end());
    return bm25_upload_fast_field(v, field);
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_set_dictionary(nidx_gpu_bm25_index_t *index, const uint8_t *bytes, const uint64_t *offsets) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !offsets) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    const uint64_t total = offsets[idx->n_terms];
    if (total && !bytes) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL dictionary bytes");
    NIDX_HIP(idx->dict_bytes.alloc(std::max<uint64_t>(total, 1)));
    NIDX_HIP(idx->dict_offsets.alloc((size_t)(idx->n_terms + 1) * 8));
    if (total) NIDX_HIP(hipMemcpy(idx->dict_bytes.p, bytes, total, hipMemcpyHostToDevice));
    NIDX_HIP(hipMemcpy(idx->dict_offsets.p, offsets, (size_t)(idx->n_terms + 1) * 8, hipMemcpyHostToDevice));
    idx->has_dict = true;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_fuzzy_terms(nidx_gpu_bm25_index_t *index, const uint8_t *query, uint32_t query_len, int32_t prefix,
                                  uint32_t *out_terms, uint32_t cap, uint32_t *n_out) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !n_out || (query_len && !query)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    *n_out = 0;
    if (!idx->has_dict) return fail(NIDX_ERR_INVALID_ARGUMENT, "no term dictionary: call nidx_gpu_bm25_set_dictionary first");
    // unicode scalar values of the query
    std::vector<uint32_t> cp;
    for (uint32_t i = 0; i < query_len;) {
        uint32_t c = query[i];
        int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : 0;
        if (extra == 1) c &= 0x1f;
        else if (extra == 2) c &= 0x0f;
        else if (extra == 3) c &= 0x07;
        i++;
        for (int e = 0; e < extra && i < query_len; e++, i++) c = (c << 6) | (query[i] & 0x3f);
        cp.push_back(c);
    }
    if (cp.empty() || cp.size() > 48) return NIDX_OK;  // nothing within one edit of an indexed token (<= 40 bytes)
    if (idx->n_terms == 0) return NIDX_OK;
    NIDX_HIP(idx->s_fuzzy_q.reserve(cp.size() * 4));
    NIDX_HIP(idx->s_fuzzy_flags.reserve(idx->n_terms));
    NIDX_HIP(hipMemcpyAsync(idx->s_fuzzy_q.p, cp.data(), cp.size() * 4, hipMemcpyHostToDevice, idx->main.stream));
    NIDX_HIP(launch_fuzzy_match(idx->dict_bytes.as<uint8_t>(), idx->dict_offsets.as<unsigned long long>(), idx->n_terms,
                                idx->s_fuzzy_q.as<uint32_t>(), (uint32_t)cp.size(), prefix ? 1 : 0, idx->s_fuzzy_flags.as<uint8_t>(), idx->main.stream));
    std::vector<uint8_t> flags(idx->n_terms);
    NIDX_HIP(hipMemcpyAsync(flags.data(), idx->s_fuzzy_flags.p, idx->n_terms, hipMemcpyDeviceToHost, idx->main.stream));
    NIDX_HIP(hipStreamSynchronize(idx->main.stream));
    uint32_t n = 0;
    for (uint32_t t = 0; t < idx->n_terms; t++)
        if (flags[t]) {
            if (out_terms && n < cap) out_terms[n] = t;
            n++;
        }
    *n_out = n;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_prefilter(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_prefilter_t *req, uint64_t *out_docaddr, uint64_t capacity,
                                uint64_t *n_matching, uint64_t *num_docs) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !req || !n_matching || (capacity && !out_docaddr)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    *n_matching = 0;
    if (num_docs) *num_docs = 0;
    const nidx_gpu_filter_program_t &prog = req->program;
    if (prog.n_ops && !prog.ops) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program without ops");
    // validate once (the term-id space is shared by the segments) and find the stack depth
    int depth = 0, max_depth = 1;
    for (uint32_t i = 0; i < prog.n_ops; i++) {
        const nidx_gpu_filter_op_t &op = prog.ops[i];
        switch (op.op) {
            case NIDX_FILTER_PUSH_LISTS:
                if (op.a > op.b || op.b > prog.n_lists || (op.b > op.a && !prog.lists))
                    return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: list range out of bounds");
                for (uint32_t l = op.a; l < op.b; l++)
                    if (prog.lists[l] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: term id %u out of range", prog.lists[l]);
                depth++;
                break;
            case NIDX_FILTER_PUSH_RANGE:
                if (op.a >= req->n_ranges || !req->ranges) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: range %u out of bounds", op.a);
                if (req->ranges[op.a].field > 1) return fail(NIDX_ERR_INVALID_ARGUMENT, "range %u: unknown fast field %u", op.a, req->ranges[op.a].field);
                depth++;
                break;
            case NIDX_FILTER_PUSH_PHRASE: {
                if (op.a >= req->n_phrases || !req->phrase_offsets || !req->phrase_terms)
                    return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: phrase %u out of bounds", op.a);
                const uint64_t m = req->phrase_offsets[op.a + 1] - req->phrase_offsets[op.a];
                if (m == 0 || m > BM25_MAX_PHRASE_TERMS)
                    return fail(NIDX_ERR_UNSUPPORTED, "a phrase has 1..%d terms (got %llu)", BM25_MAX_PHRASE_TERMS, (unsigned long long)m);
                for (uint64_t t = req->phrase_offsets[op.a]; t < req->phrase_offsets[op.a + 1]; t++)
                    if (req->phrase_terms[t] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrase: term id %u out of range", req->phrase_terms[t]);
                depth++;
                break;
            }
            case NIDX_FILTER_PUSH_ALL:
            case NIDX_FILTER_PUSH_NONE: depth++; break;
            case NIDX_FILTER_AND:
            case NIDX_FILTER_OR:
                if (depth < 2) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: stack underflow");
                depth--;
                break;
            case NIDX_FILTER_NOT:
                if (depth < 1) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: stack underflow");
                break;
            default: return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: unknown op %d", op.op);
        }
        max_depth = std::max(max_depth, depth);
    }
    if (prog.n_ops && depth != 1) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program must leave exactly one bitset (leaves %d)", depth);
    hipStream_t st = idx->main.stream;
    if (prog.n_lists) {
        NIDX_HIP(idx->s_pf_lists.reserve((size_t)prog.n_lists * 4));
        NIDX_HIP(hipMemcpyAsync(idx->s_pf_lists.p, prog.lists, (size_t)prog.n_lists * 4, hipMemcpyHostToDevice, st));
    }
    NIDX_HIP(idx->s_pf_total.reserve(16));
    NIDX_HIP(idx->s_pf_out.reserve(std::max<uint64_t>(capacity, 1) * 8));
    uint64_t matched = 0, live = 0;
    for (size_t si = 0; si < idx->segs.size(); si++) {
        Bm25Segment &seg = idx->segs[si];
        const uint32_t n_bits = seg.n_docs, words = (n_bits + 63) / 64;
        if (words == 0) continue;
        NIDX_HIP(idx->s_pf_stack.reserve((size_t)max_depth * words * 8));
        NIDX_HIP(idx->s_pf_result.reserve((size_t)words * 8));
        NIDX_HIP(idx->s_pf_blocks.reserve((size_t)((words + 255) / 256) * 4));
        uint64_t *stack = idx->s_pf_stack.as<uint64_t>();
        auto slot = [&](int d) { return stack + (size_t)d * words; };
        unsigned long long *d_total = idx->s_pf_total.as<unsigned long long>();
        if (seg.n_alive < 0) {  // searcher.num_docs(): documents not deleted
            if (seg.all_alive) {
                seg.n_alive = seg.n_docs;
            } else {
                NIDX_HIP(hipMemsetAsync(d_total, 0, 8, st));
                NIDX_HIP(launch_bitset_fill(slot(0), words, n_bits, 1, st));
                NIDX_HIP(launch_bitset_and_count(slot(0), seg.alive.as<uint64_t>(), idx->s_pf_result.as<uint64_t>(), words, d_total, st));
                unsigned long long c = 0;
                NIDX_HIP(hipMemcpyAsync(&c, d_total, 8, hipMemcpyDeviceToHost, st));
                NIDX_HIP(hipStreamSynchronize(st));
                seg.n_alive = (int64_t)c;
            }
        }
        live += (uint64_t)seg.n_alive;
        depth = 0;
        if (prog.n_ops == 0) NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 1, st));
        for (uint32_t i = 0; i < prog.n_ops; i++) {
            const nidx_gpu_filter_op_t &op = prog.ops[i];
            switch (op.op) {
                case NIDX_FILTER_PUSH_LISTS:
                    NIDX_HIP(launch_bitset_fill(slot(depth), words, n_bits, 0, st));
                    NIDX_HIP(launch_bitset_scatter(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(),
                                                   idx->s_pf_lists.as<uint32_t>() + op.a, op.b - op.a, n_bits, slot(depth), st));
                    depth++;
                    break;
                case NIDX_FILTER_PUSH_RANGE: {
                    const nidx_gpu_bm25_date_range_t &r = req->ranges[op.a];
                    if (!r.has_since && !r.has_until) {  // produce_date_range_query returns None -> AllQuery
                        NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 1, st));
                        break;
                    }
                    if (!seg.order_key[r.field].p) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %zu has no fast field %u (nidx_gpu_bm25_set_fast_field)", si, r.field);
                    const std::vector<int64_t> &u = seg.fast_uniq[r.field];
                    // ranks are 1-based positions in the distinct values: first value >= since .. last value <= until
                    const uint32_t lo = r.has_since ? (uint32_t)(std::lower_bound(u.begin(), u.end(), r.since) - u.begin()) + 1u : 1u;
                    const uint32_t hi = r.has_until ? (uint32_t)(std::upper_bound(u.begin(), u.end(), r.until) - u.begin()) : (uint32_t)u.size();
                    if (lo > hi) NIDX_HIP(launch_bitset_fill(slot(depth), words, n_bits, 0, st));
                    else NIDX_HIP(launch_rank_range_bits(seg.order_key[r.field].as<uint32_t>(), n_bits, lo, hi, slot(depth), st));
                    depth++;
                    break;
                }
                case NIDX_FILTER_PUSH_PHRASE: {
                    PhraseDev ph;
                    ph.slop = 0;
                    ph.n_terms = (uint32_t)(req->phrase_offsets[op.a + 1] - req->phrase_offsets[op.a]);
                    uint64_t best = ~0ull;
                    ph.driver = 0;
                    for (uint32_t t = 0; t < ph.n_terms; t++) {
                        ph.terms[t] = req->phrase_terms[req->phrase_offsets[op.a] + t];
                        const uint64_t df = seg.term_offsets_host[ph.terms[t] + 1] - seg.term_offsets_host[ph.terms[t]];
                        if (df < best) { best = df; ph.driver = t; }
                    }
                    NIDX_HIP(launch_bitset_fill(slot(depth), words, n_bits, 0, st));
                    if (best) {
                        if (!seg.pos_offsets.p) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrase filter on an index opened without positions");
                        NIDX_HIP(idx->main.s_phrase_tf.reserve(best * 4));
                        NIDX_HIP(launch_phrase_match(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(),
                                                     seg.pos_offsets.as<unsigned long long>(), seg.positions.as<uint32_t>(), ph, (uint32_t)best,
                                                     idx->main.s_phrase_tf.as<uint32_t>(), nullptr, st));
                        NIDX_HIP(launch_phrase_bits(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), ph, (uint32_t)best,
                                                    idx->main.s_phrase_tf.as<uint32_t>(), slot(depth), st));
                    }
                    depth++;
                    break;
                }
                case NIDX_FILTER_PUSH_ALL: NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 1, st)); break;
                case NIDX_FILTER_PUSH_NONE: NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 0, st)); break;
                case NIDX_FILTER_AND:
                case NIDX_FILTER_OR:
                    NIDX_HIP(launch_bitset_binop(slot(depth - 2), slot(depth - 1), words, op.op == NIDX_FILTER_AND ? 0 : 1, st));
                    depth--;
                    break;
                case NIDX_FILTER_NOT: NIDX_HIP(launch_bitset_not(slot(depth - 1), words, n_bits, st)); break;
            }
        }
        NIDX_HIP(hipMemsetAsync(d_total, 0, 16, st));
        NIDX_HIP(launch_bitset_and_count(slot(0), seg.all_alive ? nullptr : seg.alive.as<uint64_t>(), idx->s_pf_result.as<uint64_t>(), words,
                                         d_total, st));
        NIDX_HIP(launch_bitset_to_docaddr(idx->s_pf_result.as<uint64_t>(), words, (uint32_t)si, idx->s_pf_blocks.as<uint32_t>(), d_total + 1,
                                          matched, capacity, idx->s_pf_out.as<uint64_t>(), st));
        unsigned long long c[2] = {0, 0};
        NIDX_HIP(hipMemcpyAsync(c, d_total, 16, hipMemcpyDeviceToHost, st));
        NIDX_HIP(hipStreamSynchronize(st));
        if (c[0] != c[1]) return fail(NIDX_ERR_DEVICE, "prefilter: count mismatch (%llu vs %llu)", c[0], c[1]);
        matched += c[0];
    }
    const uint64_t n_copy = std::min<uint64_t>(matched, capacity);
    if (n_copy) {
        NIDX_HIP(hipMemcpyAsync(out_docaddr, idx->s_pf_out.p, n_copy * 8, hipMemcpyDeviceToHost, st));
        NIDX_HIP(hipStreamSynchronize(st));
        if (idx->concatenated())   // resident doc -> DocAddress of the segment it came from (ascending either way)
            for (uint64_t i = 0; i < n_copy; i++) out_docaddr[i] = idx->docaddr(0, (uint32_t)out_docaddr[i]);
    }
    *n_matching = matched;
    if (num_docs) *num_docs = live;
    return NIDX_OK;
} NIDX_ABI_CATCH

}  // extern "C"

// The search itself, on the context `cx` (the caller owns it for the duration and holds idx->rw shared).  With `async_slot` set and a
// request the pipeline covers it returns after the last asynchronous call (launches + the device-to-host transfer of the result block
// are queued on the slot's stream).
static int32_t bm25_search_locked(Bm25Index *idx, Bm25Ctx &cx, Bm25Slot *async_slot, const nidx_gpu_bm25_clause_t *clauses, const uint64_t *clause_offsets,
                                  uint32_t nq, const nidx_gpu_bm25_search_options_t *opt, uint64_t *out_docaddr, float *out_score,
                                  uint32_t *out_count, uint64_t *out_total, uint64_t *out_postings) {
    const double t_entry = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    NIDX_HIP(hipSetDevice(idx->device));
    // other tickets are out: this batch's launches will share the GPU with theirs (the throughput shape of the work list, below)
    bool crowded = false;
    if (async_slot) {
        std::lock_guard<std::mutex> lock(idx->slots_mu);
        for (auto &sl : idx->slots)
            if (sl.get() != async_slot && (sl->busy || sl->preparing)) crowded = true;
    }
    if (const char *e = getenv("NIDX_GPU_BM25_CROWDED")) crowded = atoi(e) != 0;   // measurement: 0 / 1 pins the shape
    const uint32_t k = opt->k;
    const nidx_gpu_bm25_search_after_t *after = opt->after;
    for (uint32_t q = 0; q < nq; q++) {
        out_count[q] = 0;
        if (out_total) out_total[q] = 0;
        if (out_postings) out_postings[q] = 0;
    }
    const uint64_t n_pairs = opt->facet_offsets ? opt->facet_offsets[nq] : 0;
    for (uint64_t p = 0; p < n_pairs && opt->out_facet_counts; p++) opt->out_facet_counts[p] = 0;
    if (nq == 0) return NIDX_OK;
    // the paragraph search asks for result_per_page + 1 with pages up to max(top_k, fusion / reranker window) = 500
    if (k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "TopDocs limit > %d is not supported (got %u)", NIDX_K_MAX, k);
    const int order_field = opt->order_field;
    if (order_field > 1) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown fast field %d", order_field);
    if (order_field >= 0) {
        if (after) return fail(NIDX_ERR_UNSUPPORTED, "search-after applies to the score order only");
        for (const Bm25Segment &sg : idx->segs)
            if (sg.n_docs && !sg.order_key[order_field].p) return fail(NIDX_ERR_INVALID_ARGUMENT, "fast field %d was not registered for every segment", order_field);
    }
    if (n_pairs && (!opt->facet_terms || !opt->out_facet_counts)) return fail(NIDX_ERR_INVALID_ARGUMENT, "facets without terms / output");
    const uint32_t n_sets = opt->n_term_sets;
    if (n_sets && (!opt->term_set_offsets || (opt->term_set_offsets[n_sets] && !opt->term_set_terms)))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "term sets without terms");
    for (uint64_t i = 0; n_sets && i < opt->term_set_offsets[n_sets]; i++)
        if (opt->term_set_terms[i] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "term set: term id %u out of range", opt->term_set_terms[i]);
    const uint32_t n_phrases = opt->n_phrases;
    if (n_phrases && (!opt->phrase_offsets || !opt->phrase_terms)) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrases without terms");
    for (uint32_t j = 0; j < n_phrases; j++) {
        const uint64_t m = opt->phrase_offsets[j + 1] - opt->phrase_offsets[j];
        if (m == 0 || m > BM25_MAX_PHRASE_TERMS)
            return fail(NIDX_ERR_UNSUPPORTED, "a phrase has 1..%d terms (got %llu)", BM25_MAX_PHRASE_TERMS, (unsigned long long)m);
        for (uint64_t i = opt->phrase_offsets[j]; i < opt->phrase_offsets[j + 1]; i++)
            if (opt->phrase_terms[i] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrase: term id %u out of range", opt->phrase_terms[i]);
    }
    if (n_phrases)
        for (const Bm25Segment &sg : idx->segs)
            if (sg.term_offsets_host[idx->n_terms] && !sg.pos_offsets.p) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrase clause on an index opened without positions");
    for (uint32_t j = 0; j < n_phrases && opt->phrase_slops; j++)
        if (opt->phrase_slops[j] && opt->phrase_offsets[j + 1] - opt->phrase_offsets[j] < 2)
            return fail(NIDX_ERR_INVALID_ARGUMENT, "a phrase with slop has at least two terms");
    // Bm25Weight::for_terms of a phrase: the idf of every term, summed in order
    auto phrase_weight = [&](uint32_t j, float boost) {
        float idf_sum = 0.0f;
        for (uint64_t i = opt->phrase_offsets[j]; i < opt->phrase_offsets[j + 1]; i++) {
            uint64_t df = 0;
            for (const Bm25Segment &sg : idx->segs) df += sg.term_offsets_host[opt->phrase_terms[i] + 1] - sg.term_offsets_host[opt->phrase_terms[i]];
            idf_sum += bm25_idf(df, idx->total_docs);
        }
        return idf_sum * (1.0f + kK1) * boost;
    };
    // nested BooleanQuerys (NIDX_BM25_SUBQUERY): up to 32 leaves each — terms of the dictionary, term sets, phrases, or nested queries
    // with a LOWER index (the caller lists a tree's queries children first), so they are materialised in index order
    const uint32_t n_sub = opt->n_subqueries;
    if (n_sub && (!opt->subquery_offsets || !opt->subquery_clauses)) return fail(NIDX_ERR_INVALID_ARGUMENT, "sub-queries without clauses");
    std::vector<SubqueryDev> subs(n_sub);
    for (uint32_t j = 0; j < n_sub; j++) {
        if (opt->subquery_offsets[j + 1] < opt->subquery_offsets[j]) return fail(NIDX_ERR_INVALID_ARGUMENT, "subquery_offsets not monotone");
        const uint64_t m = opt->subquery_offsets[j + 1] - opt->subquery_offsets[j];
        if (m == 0 || m > BM25_MAX_SUBQUERY_LEAVES)
            return fail(NIDX_ERR_UNSUPPORTED, "a nested query has 1..%d leaves (got %llu)", BM25_MAX_SUBQUERY_LEAVES, (unsigned long long)m);
        SubqueryDev &sq = subs[j];
        sq.n = (uint32_t)m;
        sq.driver = 0;
        for (uint32_t t = 0; t < sq.n; t++) {
            const nidx_gpu_bm25_clause_t &cl = opt->subquery_clauses[opt->subquery_offsets[j] + t];
            if (cl.occur < 0 || cl.occur > NIDX_OCCUR_SHOULD_GROUP + 7 || cl.mode < 0 || cl.mode > 2) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad clause in a nested query");
            sq.occur[t] = (uint8_t)cl.occur;
            if (cl.term & NIDX_BM25_TERM_SET) {   // ConstScorer(boost) over the union
                const uint32_t a = cl.term & ~NIDX_BM25_TERM_SET;
                if (a >= n_sets) return fail(NIDX_ERR_INVALID_ARGUMENT, "nested query %u: term set %u out of range", j, a);
                sq.src[t] = BM25_AUX_TERM | a, sq.mode[t] = NIDX_CONST_SCORE, sq.weight[t] = cl.boost;
            } else if (cl.term & NIDX_BM25_PHRASE) {
                const uint32_t a = cl.term & ~NIDX_BM25_PHRASE;
                if (a >= n_phrases) return fail(NIDX_ERR_INVALID_ARGUMENT, "nested query %u: phrase %u out of range", j, a);
                sq.src[t] = BM25_AUX_TERM | (n_sets + a), sq.mode[t] = NIDX_TF_FREQ, sq.weight[t] = phrase_weight(a, cl.boost);
            } else if (cl.term & NIDX_BM25_SUBQUERY) {
                const uint32_t a = cl.term & ~NIDX_BM25_SUBQUERY;
                if (a >= j) return fail(NIDX_ERR_INVALID_ARGUMENT, "nested query %u refers to nested query %u: children come first", j, a);
                sq.src[t] = BM25_AUX_TERM | (n_sets + n_phrases + a), sq.mode[t] = 3, sq.weight[t] = cl.boost;
            } else {
                if (cl.term >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "nested query %u: term id %u out of range", j, cl.term);
                uint64_t df = 0;
                for (const Bm25Segment &sg : idx->segs) df += sg.term_offsets_host[cl.term + 1] - sg.term_offsets_host[cl.term];
                sq.src[t] = cl.term, sq.mode[t] = (uint8_t)cl.mode;
                sq.weight[t] = cl.mode == NIDX_CONST_SCORE ? cl.boost : bm25_idf(df, idx->total_docs) * (1.0f + kK1) * cl.boost;
            }
        }
    }
    const uint64_t n_clauses = clause_offsets[nq];
    if (n_clauses && !clauses) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL clauses");
    // Bm25Weight per clause from searcher-wide statistics
    std::vector<Bm25ClauseDev> &dev_clauses = cx.w_dev_clauses;
    dev_clauses.resize(n_clauses);
    uint32_t max_clauses = 0;
    for (uint32_t q = 0; q < nq; q++) {
        if (clause_offsets[q + 1] < clause_offsets[q]) return fail(NIDX_ERR_INVALID_ARGUMENT, "clause_offsets not monotone");
        if (clause_offsets[q + 1] - clause_offsets[q] > BM25_MAX_CLAUSES)
            return fail(NIDX_ERR_UNSUPPORTED, "more than %d clauses in one query", BM25_MAX_CLAUSES);
        max_clauses = std::max<uint32_t>(max_clauses, (uint32_t)(clause_offsets[q + 1] - clause_offsets[q]));
    }
    // launch shape knobs (no effect on results): postings rows per window and postings per work item
    uint64_t slice_postings = BM25_SLICE_POSTINGS;
    const bool force_wide = getenv("NIDX_GPU_BM25_WIDE") != nullptr;   // every query through the general kernel (tests)
    // NIDX_GPU_BM25_UNION: 0 = never a union kernel, 1 = rare-meeting unions through bm25_stream_kernel (default), 2 = every query of
    // <= 8 plain term clauses through it (tests drive its slow paths with it); 3 / 4 = the same two with bm25_union_kernel (round 3's
    // first union kernel, kept for comparison)
    int union_mode = 1;
    if (const char *e = getenv("NIDX_GPU_BM25_UNION")) union_mode = atoi(e);
    const bool lockstep_union = union_mode >= 3;
    if (lockstep_union) union_mode -= 2;
    (void)max_clauses;
    if (const char *e = getenv("NIDX_GPU_BM25_SLICE")) slice_postings = (uint64_t)std::max(256, atoi(e));
    for (uint64_t c = 0; c < n_clauses; c++) {
        const nidx_gpu_bm25_clause_t &cl = clauses[c];
        if (cl.occur < 0 || cl.occur > NIDX_OCCUR_SHOULD_GROUP + 7 || cl.mode < 0 || cl.mode > 2) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad clause");
        if (!(cl.term & (NIDX_BM25_TERM_SET | NIDX_BM25_PHRASE)) && (cl.term & NIDX_BM25_SUBQUERY)) {
            const uint32_t j = cl.term & ~NIDX_BM25_SUBQUERY;
            if (j >= n_sub) return fail(NIDX_ERR_INVALID_ARGUMENT, "nested query %u out of range", j);
            // materialised per segment as aux list n_sets + n_phrases + j; the clause contributes boost x the nested score (mode 3)
            dev_clauses[c] = Bm25ClauseDev{BM25_AUX_TERM | (n_sets + n_phrases + j), cl.occur, 3, cl.boost};
            continue;
        }
        if (!(cl.term & NIDX_BM25_TERM_SET) && (cl.term & NIDX_BM25_PHRASE)) {
            const uint32_t j = cl.term & ~NIDX_BM25_PHRASE;
            if (j >= n_phrases) return fail(NIDX_ERR_INVALID_ARGUMENT, "phrase %u out of range", j);
            // the phrase's matches are materialised per segment as aux list n_sets + j, with their frequencies
            dev_clauses[c] = Bm25ClauseDev{BM25_AUX_TERM | (n_sets + j), cl.occur, NIDX_TF_FREQ, phrase_weight(j, cl.boost)};
            continue;
        }
        if (cl.term & NIDX_BM25_TERM_SET) {
            if ((cl.term & ~NIDX_BM25_TERM_SET) >= n_sets) return fail(NIDX_ERR_INVALID_ARGUMENT, "term set %u out of range", cl.term & ~NIDX_BM25_TERM_SET);
            dev_clauses[c] = Bm25ClauseDev{cl.term, cl.occur, NIDX_CONST_SCORE, cl.boost};  // ConstScorer(boost)
            continue;
        }
        if (cl.term >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "term id %u out of range", cl.term);
        float w = cl.boost;
        if (cl.mode != NIDX_CONST_SCORE) {
            w = idx->idf_of_term[cl.term] * (1.0f + kK1) * cl.boost;
        }
        dev_clauses[c] = Bm25ClauseDev{cl.term, cl.occur, cl.mode, w};
    }
    const uint32_t kk = std::max<uint32_t>(k, 1);
    // clauses and clause offsets travel as one pinned block
    const size_t cl_bytes = (std::max<size_t>(n_clauses, 1) * sizeof(Bm25ClauseDev) + 7) & ~(size_t)7;
    const size_t inq_bytes = cl_bytes + (size_t)(nq + 1) * 8;
    // One resident segment (the common case, several opened segments included): the clause block travels at the head of the work
    // list's transfer — every dependent operation queued on a stream costs ~10 us of hand-over between the copy engine and the compute
    // queue, and a batch's kernels are ~70 us.  The per-segment loop (NIDX_GPU_BM25_SEGMENT_LOOP) keeps two transfers: its clause block
    // outlives the work lists.
    const bool fused_in = idx->segs.size() == 1;
    const size_t inq_al = (inq_bytes + 255) & ~(size_t)255;
    const Bm25ClauseDev *d_clauses = nullptr;
    const unsigned long long *d_clause_offsets = nullptr;
    if (!fused_in) {
        NIDX_HIP(cx.s_in_q.reserve(inq_bytes));
        NIDX_HIP(cx.h_in_q.reserve(inq_bytes));
        if (n_clauses) memcpy(cx.h_in_q.p, dev_clauses.data(), n_clauses * sizeof(Bm25ClauseDev));
        memcpy(cx.h_in_q.as<unsigned char>() + cl_bytes, clause_offsets, (size_t)(nq + 1) * 8);
        NIDX_HIP(hipMemcpyAsync(cx.s_in_q.p, cx.h_in_q.p, inq_bytes, hipMemcpyHostToDevice, cx.stream));
        d_clauses = cx.s_in_q.as<Bm25ClauseDev>();
        d_clause_offsets = reinterpret_cast<const unsigned long long *>(cx.s_in_q.as<unsigned char>() + cl_bytes);
    }
    // The merged hits are written by bm25_merge_kernel straight into the pinned result block (device-visible host memory: ~190 KB of
    // fire-and-forget stores over PCIe per batch) instead of into HBM + a device-to-host transfer queued behind it: one dependent
    // operation less per batch.  NIDX_GPU_BM25_ZERO_COPY_OUT=0 keeps the transfer (comparison).
    bool zc_out = true;
    if (const char *e = getenv("NIDX_GPU_BM25_ZERO_COPY_OUT")) zc_out = atoi(e) != 0;
    static_assert(sizeof(Bm25AfterDev) == sizeof(nidx_gpu_bm25_search_after_t), "search-after layout");
    if (after) {
        NIDX_HIP(cx.s_after.reserve((size_t)nq * sizeof(Bm25AfterDev)));
        const void *src = after;
        if (idx->concatenated()) {
            // the cursor's DocAddress in the resident numbering (segment_ord 0, doc + base): (segment, doc) order is kept.  A cursor
            // beyond the end of its segment sits just before the next segment's first document.
            cx.w_after.resize(nq);
            const uint32_t S = idx->n_segments;
            for (uint32_t q = 0; q < nq; q++) {
                Bm25AfterDev c;
                memcpy(&c, &after[q], sizeof(c));
                const uint64_t seg = c.docaddr >> 32, d = c.docaddr & 0xffffffffull;
                if (seg >= S) c.docaddr = 0xffffffffull;   // behind every document (resident doc ids are < 2^32 - 1 .. and compared strictly)
                else if (d < idx->real[seg].n_docs) c.docaddr = (uint64_t)idx->seg_base[seg] + d;
                else if (idx->seg_base[seg + 1] == 0) { if (c.tie_break == 1) c.tie_break = 0; c.docaddr = 0; }   // before every document
                else c.docaddr = (uint64_t)idx->seg_base[seg + 1] - 1u;
                cx.w_after[q] = c;
            }
            src = cx.w_after.data();
        }
        NIDX_HIP(hipMemcpyAsync(cx.s_after.p, src, (size_t)nq * sizeof(Bm25AfterDev), hipMemcpyHostToDevice, cx.stream));
    }
    // facets: one matching-document bitset per query that asks for counts
    std::vector<int> match_slot(nq, -1);
    std::vector<uint32_t> pair_term(n_pairs);
    std::vector<int> pair_slot(n_pairs, -1);
    uint32_t n_slots = 0;
    for (uint32_t q = 0; q < nq && n_pairs; q++) {
        if (opt->facet_offsets[q + 1] < opt->facet_offsets[q]) return fail(NIDX_ERR_INVALID_ARGUMENT, "facet_offsets not monotone");
        if (opt->facet_offsets[q + 1] == opt->facet_offsets[q]) continue;
        match_slot[q] = (int)n_slots++;
        for (uint64_t p = opt->facet_offsets[q]; p < opt->facet_offsets[q + 1]; p++) {
            if (opt->facet_terms[p] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "facet term id %u out of range", opt->facet_terms[p]);
            pair_term[p] = opt->facet_terms[p];
            pair_slot[p] = match_slot[q];
        }
    }
    if (n_slots) {
        NIDX_HIP(cx.s_match_slot.reserve((size_t)nq * 4));
        NIDX_HIP(cx.s_pair_term.reserve(n_pairs * 4));
        NIDX_HIP(cx.s_pair_slot.reserve(n_pairs * 4));
        NIDX_HIP(cx.s_facet_counts.reserve(n_pairs * 8));
        NIDX_HIP(hipMemcpyAsync(cx.s_match_slot.p, match_slot.data(), (size_t)nq * 4, hipMemcpyHostToDevice, cx.stream));
        NIDX_HIP(hipMemcpyAsync(cx.s_pair_term.p, pair_term.data(), n_pairs * 4, hipMemcpyHostToDevice, cx.stream));
        NIDX_HIP(hipMemcpyAsync(cx.s_pair_slot.p, pair_slot.data(), n_pairs * 4, hipMemcpyHostToDevice, cx.stream));
        NIDX_HIP(hipMemsetAsync(cx.s_facet_counts.p, 0, n_pairs * 8, cx.stream));
    }
    if (n_sets) {
        const uint64_t n_set_terms = opt->term_set_offsets[n_sets];
        NIDX_HIP(cx.s_set_terms.reserve(std::max<uint64_t>(n_set_terms, 1) * 4));
        if (n_set_terms) NIDX_HIP(hipMemcpyAsync(cx.s_set_terms.p, opt->term_set_terms, n_set_terms * 4, hipMemcpyHostToDevice, cx.stream));
    }
    cx.kernel_ms = 0.f;
    const bool host_dbg = getenv("NIDX_GPU_BM25_DEBUG") != nullptr || getenv("NIDX_GPU_BM25_HOST_TRACE") != nullptr;
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_us();
    double t_work = 0, t_sync = 0, t_collect = 0;
    struct Hit { float score; uint64_t docaddr; int64_t value; };
    std::vector<std::vector<Hit>> merged(idx->segs.size() == 1 ? 0 : nq);
    std::vector<Bm25Work> &work = cx.w_work;
    for (size_t s = 0; s < idx->segs.size(); s++) {
        Bm25Segment &seg = idx->segs[s];
        // ---- the aux lists of this segment, in dependency order: term sets (union bitset -> ascending doc list,
        // AutomatonWeight::scorer), phrases, nested queries (children before parents); one host round trip for all their counts ----
        const uint32_t n_aux = n_sets + n_phrases + n_sub;
        std::vector<unsigned long long> aux_pairs(2 * (size_t)n_aux + 2, 0);  // [begin, end) per aux list into s_aux_ids
        std::vector<uint32_t> set_counts(n_aux, 0);
        // layout of the aux arrays: one region per list, sized by an upper bound of its length
        std::vector<unsigned long long> out_off(n_aux + 1, 0);
        std::vector<PhraseDev> phrases(n_phrases);
        for (uint32_t j = 0; j < n_sets; j++) {
            uint64_t df = 0;
            for (uint64_t i = opt->term_set_offsets[j]; i < opt->term_set_offsets[j + 1]; i++)
                df += seg.term_offsets_host[opt->term_set_terms[i] + 1] - seg.term_offsets_host[opt->term_set_terms[i]];
            const bool comp = opt->term_set_complement && opt->term_set_complement[j];
            out_off[j + 1] = out_off[j] + (comp ? (uint64_t)seg.n_docs : std::min<uint64_t>(df, seg.n_docs));
        }
        for (uint32_t j = 0; j < n_phrases; j++) {
            PhraseDev &ph = phrases[j];
            ph.n_terms = (uint32_t)(opt->phrase_offsets[j + 1] - opt->phrase_offsets[j]);
            ph.slop = opt->phrase_slops ? opt->phrase_slops[j] : 0u;
            ph.driver = 0;
            uint64_t best = ~0ull;
            for (uint32_t t = 0; t < ph.n_terms; t++) {
                ph.terms[t] = opt->phrase_terms[opt->phrase_offsets[j] + t];
                const uint64_t df = seg.term_offsets_host[ph.terms[t] + 1] - seg.term_offsets_host[ph.terms[t]];
                if (df < best) { best = df; ph.driver = t; }
            }
            out_off[n_sets + j + 1] = out_off[n_sets + j] + best;
        }
        // nested queries: the candidates are the shortest Must leaf (by upper bound: an aux leaf's exact length is on the device only),
        // else the union of the smallest required Should group, else the union of the Should leaves
        struct SubPlan { uint32_t union_mask = 0; uint64_t n_cand_max = 0; };
        std::vector<SubPlan> plans(n_sub);
        uint64_t max_cand = 0;
        bool any_union = false;
        for (uint32_t j = 0; j < n_sub; j++) {
            SubqueryDev &sq = subs[j];
            auto upper = [&](uint32_t t) -> uint64_t {
                if (sq.src[t] & BM25_AUX_TERM) { const uint32_t a = sq.src[t] & ~BM25_AUX_TERM; return out_off[a + 1] - out_off[a]; }
                return seg.term_offsets_host[sq.src[t] + 1] - seg.term_offsets_host[sq.src[t]];
            };
            uint64_t best = ~0ull, group_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, should_sum = 0;
            uint32_t group_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, should_mask = 0;
            bool any_must = false;
            for (uint32_t t = 0; t < sq.n; t++) {
                const uint64_t ub = upper(t);
                if (sq.occur[t] == NIDX_OCCUR_MUST) {
                    any_must = true;
                    if (ub < best) { best = ub; sq.driver = t; }
                } else if (sq.occur[t] >= NIDX_OCCUR_SHOULD_GROUP) {
                    group_sum[sq.occur[t] - NIDX_OCCUR_SHOULD_GROUP] += ub, group_mask[sq.occur[t] - NIDX_OCCUR_SHOULD_GROUP] |= 1u << t;
                } else if (sq.occur[t] == NIDX_OCCUR_SHOULD) should_sum += ub, should_mask |= 1u << t;
            }
            SubPlan &pl = plans[j];
            if (any_must) pl.n_cand_max = best;
            else {
                sq.driver = BM25_SUB_DRIVER_UNION;
                uint64_t sum = ~0ull;
                for (int g = 0; g < 8; g++)
                    if (group_mask[g] && group_sum[g] < sum) sum = group_sum[g], pl.union_mask = group_mask[g];
                if (!pl.union_mask) sum = should_sum, pl.union_mask = should_mask;   // no positive leaf at all: nothing matches
                pl.n_cand_max = pl.union_mask ? std::min<uint64_t>(sum, seg.n_docs) : 0;
                any_union |= pl.union_mask != 0;
            }
            if (pl.n_cand_max > 0xffffffffull) return fail(NIDX_ERR_UNSUPPORTED, "a posting list of one segment holds more than 2^32 - 1 postings");
            max_cand = std::max(max_cand, pl.n_cand_max);
            out_off[n_sets + n_phrases + j + 1] = out_off[n_sets + n_phrases + j] + pl.n_cand_max;
        }
        const uint32_t words = (seg.n_docs + 63) / 64;
        if (n_aux) {
            NIDX_HIP(cx.s_aux_ids.reserve(std::max<uint64_t>(out_off[n_aux], 1) * 4 + BM25_LIST_PAD_BYTES));
            NIDX_HIP(cx.s_aux_tfs.reserve(std::max<uint64_t>(out_off[n_aux], 1) * 4 + BM25_LIST_PAD_BYTES));
            NIDX_HIP(cx.s_set_counts.reserve((size_t)n_aux * 4 + 4));   // + the count of the union candidates
            NIDX_HIP(hipMemsetAsync(cx.s_set_counts.p, 0, (size_t)n_aux * 4 + 4, cx.stream));
            NIDX_HIP(cx.s_aux_out_off.reserve((size_t)(n_aux + 1) * 8));
            NIDX_HIP(hipMemcpyAsync(cx.s_aux_out_off.p, out_off.data(), (size_t)(n_aux + 1) * 8, hipMemcpyHostToDevice, cx.stream));
        }
        if (n_sets && words) {
            NIDX_HIP(cx.s_set_bits.reserve(std::max<size_t>((size_t)n_sets * words, 1) * 8));
            NIDX_HIP(launch_bitset_fill(cx.s_set_bits.as<uint64_t>(), n_sets * words, n_sets * words * 64u, 0, cx.stream));
            for (uint32_t j = 0; j < n_sets; j++) {
                const uint32_t nl = (uint32_t)(opt->term_set_offsets[j + 1] - opt->term_set_offsets[j]);
                if (nl)
                    NIDX_HIP(launch_bitset_scatter(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(),
                                                   cx.s_set_terms.as<uint32_t>() + opt->term_set_offsets[j], nl, seg.n_docs,
                                                   cx.s_set_bits.as<uint64_t>() + (size_t)j * words, cx.stream));
                if (opt->term_set_complement && opt->term_set_complement[j])
                    NIDX_HIP(launch_bitset_not(cx.s_set_bits.as<uint64_t>() + (size_t)j * words, words, seg.n_docs, cx.stream));
            }
            NIDX_HIP(launch_bitset_compact(cx.s_set_bits.as<uint64_t>(), words, n_sets, cx.s_aux_out_off.as<unsigned long long>(),
                                           cx.s_aux_ids.as<uint32_t>(), cx.s_set_counts.as<uint32_t>(), cx.stream));
        }
        for (uint32_t j = 0; j < n_phrases; j++) {
            const uint64_t n_driver = out_off[n_sets + j + 1] - out_off[n_sets + j];
            if (n_driver == 0) continue;
            NIDX_HIP(cx.s_phrase_tf.reserve(n_driver * 4));
            uint32_t *slop_left = nullptr;
            if (phrases[j].slop) {
                // PhraseScorer's `left` list per document: as many entries as the first term has positions in this segment
                const uint64_t tb = seg.term_offsets_host[phrases[j].terms[0]], te = seg.term_offsets_host[phrases[j].terms[0] + 1];
                unsigned long long pr[2] = {0, 0};
                NIDX_HIP(hipMemcpyAsync(&pr[0], seg.pos_offsets.as<unsigned long long>() + tb, 8, hipMemcpyDeviceToHost, cx.stream));
                NIDX_HIP(hipMemcpyAsync(&pr[1], seg.pos_offsets.as<unsigned long long>() + te, 8, hipMemcpyDeviceToHost, cx.stream));
                NIDX_HIP(hipStreamSynchronize(cx.stream));
                NIDX_HIP(cx.s_slop_left.reserve(std::max<uint64_t>(pr[1] - pr[0], 1) * 8));   // (position, budget used) pairs
                slop_left = cx.s_slop_left.as<uint32_t>();
            }
            NIDX_HIP(launch_phrase_match(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), seg.pos_offsets.as<unsigned long long>(),
                                         seg.positions.as<uint32_t>(), phrases[j], (uint32_t)n_driver, cx.s_phrase_tf.as<uint32_t>(), slop_left, cx.stream));
            NIDX_HIP(launch_phrase_compact(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), phrases[j], cx.s_phrase_tf.as<uint32_t>(), seg.fieldnorm_ids.as<uint8_t>(),
                                           out_off[n_sets + j], cx.s_aux_ids.as<uint32_t>(), cx.s_aux_tfs.as<uint32_t>(),
                                           cx.s_set_counts.as<uint32_t>() + n_sets + j, cx.stream));
        }
        if (n_sub) {
            NIDX_HIP(cx.s_phrase_tf.reserve(std::max<uint64_t>(max_cand, 1) * 8));   // [n] match flags | [n] score bits
            if (any_union) {
                NIDX_HIP(cx.s_sub_bits.reserve(std::max<size_t>(words, 1) * 8));
                NIDX_HIP(cx.s_sub_union.reserve(std::max<uint64_t>(max_cand, 1) * 4));
            }
            SubqueryLists L{seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), seg.tfs.as<uint32_t>(), cx.s_aux_ids.as<uint32_t>(),
                            cx.s_aux_tfs.as<uint32_t>(), cx.s_aux_out_off.as<unsigned long long>(), cx.s_set_counts.as<uint32_t>(),
                            cx.s_sub_union.as<uint32_t>(), cx.s_set_counts.as<uint32_t>() + n_aux};
            for (uint32_t j = 0; j < n_sub; j++) {
                const SubPlan &pl = plans[j];
                if (pl.n_cand_max == 0) continue;
                if (subs[j].driver == BM25_SUB_DRIVER_UNION) {
                    NIDX_HIP(launch_bitset_fill(cx.s_sub_bits.as<uint64_t>(), words, seg.n_docs, 0, cx.stream));
                    for (uint32_t t = 0; t < subs[j].n; t++)
                        if (pl.union_mask >> t & 1) NIDX_HIP(launch_subquery_scatter(L, subs[j].src[t], pl.n_cand_max, cx.s_sub_bits.as<uint64_t>(), cx.stream));
                    // out_off[0] == 0: the union list starts at the head of its own buffer
                    NIDX_HIP(launch_bitset_compact(cx.s_sub_bits.as<uint64_t>(), words, 1, cx.s_aux_out_off.as<unsigned long long>(),
                                                   cx.s_sub_union.as<uint32_t>(), cx.s_set_counts.as<uint32_t>() + n_aux, cx.stream));
                }
                uint32_t *tmp_ok = cx.s_phrase_tf.as<uint32_t>(), *tmp_score = tmp_ok + max_cand;
                NIDX_HIP(launch_subquery_match(L, idx->tf_cache.as<float>(), subs[j], (uint32_t)pl.n_cand_max, tmp_ok, tmp_score, cx.stream));
                NIDX_HIP(launch_subquery_compact(L, subs[j], tmp_ok, tmp_score, out_off[n_sets + n_phrases + j], cx.s_aux_ids.as<uint32_t>(),
                                                 cx.s_aux_tfs.as<uint32_t>(), cx.s_set_counts.as<uint32_t>() + n_sets + n_phrases + j, cx.stream));
            }
        }
        if (n_aux) {
            NIDX_HIP(hipMemcpyAsync(set_counts.data(), cx.s_set_counts.p, (size_t)n_aux * 4, hipMemcpyDeviceToHost, cx.stream));
            NIDX_HIP(hipStreamSynchronize(cx.stream));
            for (uint32_t j = 0; j < n_aux; j++) {
                aux_pairs[2 * j] = out_off[j];
                aux_pairs[2 * j + 1] = out_off[j] + set_counts[j];
            }
            NIDX_HIP(cx.s_aux_off.reserve(aux_pairs.size() * 8));
            NIDX_HIP(hipMemcpyAsync(cx.s_aux_off.p, aux_pairs.data(), aux_pairs.size() * 8, hipMemcpyHostToDevice, cx.stream));
        }
        const double t_seg0 = now_us();
        auto postings_of = [&](const nidx_gpu_bm25_clause_t &cl) -> uint64_t {
            if (cl.term & NIDX_BM25_TERM_SET) return set_counts[cl.term & ~NIDX_BM25_TERM_SET];
            if (cl.term & NIDX_BM25_PHRASE) return set_counts[n_sets + (cl.term & ~NIDX_BM25_PHRASE)];
            if (cl.term & NIDX_BM25_SUBQUERY) return set_counts[n_sets + n_phrases + (cl.term & ~NIDX_BM25_SUBQUERY)];
            return seg.term_offsets_host[cl.term + 1] - seg.term_offsets_host[cl.term];
        };
        // work list: every query cut into doc-id slices of ~BM25_SLICE_POSTINGS postings.  Three launches over disjoint item lists: term
        // unions whose lists rarely meet (bm25_stream.hip), the other narrow queries (bm25_fast_kernel), the rest (bm25_rows_kernel).  A
        // query is a "rare-meeting union" when it has at most 8 plain term clauses and the number of documents two of its lists are
        // expected to share (independent lists: len_a * len_b / n_docs, summed over the pairs) is at most 1/8 of its postings — the union
        // kernels are exact for any input, that bound only keeps their slow paths rare.
        const double t_w0 = now_us();
        // every clause's list length in this segment, looked up once (two random reads of an 8 MB table per plain term)
        std::vector<uint64_t> &clen = cx.w_clause_len;
        clen.resize(n_clauses);
        for (uint64_t c = 0; c < n_clauses; c++) {
            if (c + 16 < n_clauses && !(clauses[c + 16].term & (NIDX_BM25_TERM_SET | NIDX_BM25_PHRASE | NIDX_BM25_SUBQUERY)))
                __builtin_prefetch(&seg.term_offsets_host[clauses[c + 16].term]);
            clen[c] = postings_of(clauses[c]);
        }
        work.clear();
        std::vector<uint32_t> &item_first = cx.w_item_first;
        std::vector<uint8_t> &q_union = cx.w_q_union;
        item_first.assign(nq + 1, 0);
        q_union.assign(nq, 0);
        const double inv_docs = 1.0 / std::max<double>(1.0, (double)seg.n_docs);
        // Postings per work item.  A launch lasts as long as its slowest item while every workgroup of it is resident at once (5 per CU,
        // 4 items each); past that the tail of the grid runs as a second round.  So: the smallest slice between 1 024 and the default
        // that still fits the batch into one round (the bench batch: ~1 900 instead of 2 048, its slowest items 7 % shorter); batches
        // too large for one round keep the default.  NIDX_GPU_BM25_SLICE pins it.
        uint64_t slice_now = slice_postings;
        uint64_t short_weight = 1;   // what a posting outside the query's longest clause counts when the slices are cut (the longest clause's count 1)
        auto weigh = [&](uint64_t sum, uint64_t longest) { return longest + short_weight * (sum - longest); };
        // Other batches are resident (the pipelined entries know): the launch does not have to fill the GPU alone, and what counts is
        // the work per posting.  With k = 20 a wave that filters n postings offers ~k ln(n / k) candidates to its list and merges them
        // 64 at a time — candidates and merges are half of the kernel's vector instructions at ~1 900 postings per item and fall per
        // posting as the item grows (measured, scripts/r6_bm25_probe.py: long lists 214 -> 248 G postings/s at 4 096).  So a union query is cut
        // into the FEWEST slices of up to BM25_SLICE_CROWDED postings whose expected number of involved postings — true meetings and the
        // collisions of the kernel's two bitmaps (bm25_stream.hip: 32 Kibit A, 2 Kibit B) — stays inside the 192-entry list:
        // a balanced query keeps ~1 900 postings per slice (the bitmaps allow no more), a long list with two short ones gets ~8 000.
        // A batch alone on the GPU (the blocking entries, the first ticket) keeps the latency shape below.
        const bool crowded_shape = crowded && !getenv("NIDX_GPU_BM25_SLICE") && !lockstep_union;
        if (crowded_shape) {
            slice_now = BM25_SLICE_CROWDED;
            if (const char *e = getenv("NIDX_GPU_BM25_CROWDED_SLICE")) slice_now = (uint64_t)std::max(1024, atoi(e));   // (measurement)
        } else if (!getenv("NIDX_GPU_BM25_SLICE")) {
            const uint64_t budget = (uint64_t)idx->n_cus * 5u * 4u * 15u / 16u;
            // A launch alone lasts as long as its slowest item, so the items are cut to equal COST, not equal length: a posting of the
            // longest clause is streamed once (phase 2), one of any other clause twice (marked in phase 1, scored in phase 3).  Measured on
            // the bench batch (scripts/r6_bm25_weight.sh): with one-bit bitmaps, when a third of the kernel went into falsely involved
            // postings, weight 3 took a launch from 49.5 to 44.5 us; with the three-bit filter of bitmap A weights 1 .. 3 are within
            // 2 us of each other (43.3 - 45.7) and 2 is kept.
            std::vector<uint64_t> &pq = cx.w_postings;
            pq.assign(nq, 0);
            short_weight = 2;
            if (const char *e = getenv("NIDX_GPU_BM25_SHORT_WEIGHT")) short_weight = (uint64_t)std::max(1, atoi(e));   // (measurement)
            for (uint32_t q = 0; q < nq; q++) {
                uint64_t sum = 0, longest = 0;
                for (uint64_t c = clause_offsets[q]; c < clause_offsets[q + 1]; c++) sum += clen[c], longest = std::max<uint64_t>(longest, clen[c]);
                pq[q] = weigh(sum, longest);
            }
            // (the item count falls as the slice grows: bisection over the candidates, and a multiplication by the reciprocal instead of
            // a 64-bit division per query and candidate — the count only steers this choice, the slicing below divides exactly)
            auto fits = [&](uint64_t cand) {
                const double inv = 1.0 / (double)cand;
                uint64_t items = 0;
                for (uint32_t q = 0; q < nq; q++) items += (uint64_t)((double)pq[q] * inv) + 1u;
                return items <= budget;
            };
            uint64_t lo = 1024, hi = slice_postings * (short_weight > 1 ? 2u : 1u);   // candidates lo, lo + 128, ..; hi: the default length (in weighted postings)
            while (lo < hi) {
                const uint64_t mid = lo + ((hi - lo) / 256) * 128;
                if (fits(mid)) hi = mid;
                else lo = mid + 128;
            }
            slice_now = hi;
        }
        for (uint32_t q = 0; q < nq; q++) {
            item_first[q] = (uint32_t)work.size();
            const uint64_t c0 = clause_offsets[q], c1 = clause_offsets[q + 1];
            uint64_t p = 0, p_longest = 0;
            bool plain = true;
            double sum_sq = 0.0;
            for (uint64_t c = c0; c < c1; c++) {
                const uint64_t l = clen[c];
                p += l;
                p_longest = std::max(p_longest, l);
                sum_sq += (double)l * (double)l;
                if (clauses[c].term & (NIDX_BM25_TERM_SET | NIDX_BM25_PHRASE | NIDX_BM25_SUBQUERY)) plain = false;
            }
            uint32_t slices = (uint32_t)std::min<uint64_t>(BM25_MAX_SLICES, std::max<uint64_t>(1, (weigh(p, p_longest) + slice_now - 1) / slice_now));
            if (union_mode != 0 && plain && c1 > c0 && c1 - c0 <= BM25_FAST_CLAUSES && !force_wide) {
                const double sum = (double)p, shared = (sum * sum - sum_sq) * 0.5 * inv_docs;
                // the same term twice: those two lists meet in every document (the estimate above assumes independent lists)
                double repeated = 0.0;
                for (uint64_t c = c0; c < c1; c++)
                    for (uint64_t e = c + 1; e < c1; e++)
                        if (clauses[c].term == clauses[e].term) repeated += (double)clen[c];
                // the stream kernel resolves up to 192 involved postings per item without cutting its doc range: enough slices to keep
                // the expected number (two postings per meeting) around 96
                double want = std::ceil((shared + repeated) * 2.0 / 96.0);
                if (crowded_shape) {
                    // involved postings of one of n slices: b = postings that find their three bits of A set (bm25_stream.hip: a document
                    // owns one of 1 024 words and three of its bits; with lambda = marked documents per word a stranger passes with probability
                    // ~ (27 / 32 768) (lambda^3 + 3 lambda^2 + lambda): the third moment of a Poisson count of three-bit marks) — the longest
                    // clause's postings against the full filter, the shorter clauses' against the filter as it fills (a quarter of that) —
                    // plus two per true meeting; each sets a bit of B, and every short posting then meets B with probability b / 2 048
                    double l_max = 0.0;
                    for (uint64_t c = c0; c < c1; c++) l_max = std::max(l_max, (double)clen[c]);
                    const double s_all = sum - l_max, meet2 = (shared + repeated) * 2.0;
                    double n = std::max(1.0, (double)slices);
                    for (int it = 0; it < 24; it++) {
                        const double ss = s_all / n, ll = l_max / n, lam = ss / 1024.0;
                        const double p_a = std::min(1.0, (27.0 / 32768.0) * (lam * lam * lam + 3.0 * lam * lam + lam));
                        const double b = ll * p_a + ss * p_a * 0.25 + meet2 / n;
                        if (b * (1.0 + ss / 2048.0) <= 96.0) break;
                        n = std::ceil(n * 1.25);
                    }
                    // (never fewer postings per slice than the latency shape would give: the estimate above is the only reason to cut finer)
                    want = std::max(want, std::min(n, std::ceil(sum / 1024.0)));
                }
                if (union_mode == 2) q_union[q] = 1;
                else if (shared * 8.0 <= sum && want <= (double)BM25_MAX_SLICES) q_union[q] = 1;
                if (q_union[q] && !lockstep_union) slices = (uint32_t)std::min<double>(BM25_MAX_SLICES, std::max<double>(slices, want));

            }
            const Bm25Work w{q, 0, slices, (uint32_t)c0, (uint32_t)(c1 - c0)};
            work.resize(work.size() + slices, w);
            Bm25Work *wp = work.data() + work.size() - slices;
            for (uint32_t sl = 1; sl < slices; sl++) wp[sl].slice = sl;
        }
        const size_t nw = work.size();
        item_first[nq] = (uint32_t)nw;
        std::vector<uint32_t> &item_list = cx.w_item_list;
        item_list.resize(nw);
        uint32_t n_union = 0, n_fast = 0, n_wide = 0, wide_max_clauses = 0;
        for (size_t w = 0; w < nw; w++)
            if (q_union[work[w].query]) item_list[n_union++] = (uint32_t)w;
        if (n_union < nw) {
            for (size_t w = 0; w < nw; w++) {
                const uint32_t nc = work[w].n_clauses;
                if (!q_union[work[w].query] && nc <= BM25_FAST_CLAUSES && !force_wide) item_list[n_union + n_fast++] = (uint32_t)w;
            }
            for (size_t w = 0; w < nw; w++) {
                const uint32_t nc = work[w].n_clauses;
                if (!q_union[work[w].query] && !(nc <= BM25_FAST_CLAUSES && !force_wide)) {
                    item_list[n_union + n_fast + n_wide++] = (uint32_t)w;
                    wide_max_clauses = std::max(wide_max_clauses, nc);
                }
            }
        }
        // the union kernel's clause table: list base / length / weight / attributes per clause of this segment
        std::vector<Bm25UClause> &ucl = cx.w_ucl;
        ucl.resize(n_union ? n_clauses : 0);
        // A floor under the k-th best score of a query whose clauses are all Should terms (every document of any clause is a hit): clause c
        // alone gives >= BM25_FLOOR_RANKS[j] >= k documents a score >= weight(c) * quotient(tf = 1, floor_fn[term][j]); the best clause's bound
        // is the query's.  Only where nothing removes documents behind the scorer's back: no deletions, no cursor, no fast-field order
        // (those launches run the kernel's EXTRAS form, which never reads the floor).
        static const uint32_t floor_ranks[BM25_FLOOR_NR] = BM25_FLOOR_RANKS;
        int floor_j = -1;
        if (!idx->floor_fn.empty() && seg.all_alive && !after && order_field < 0)
            for (int j = 0; j < BM25_FLOOR_NR && floor_j < 0; j++)
                if (floor_ranks[j] >= kk) floor_j = j;
        for (uint32_t q = 0; q < nq && n_union; q++) {
            if (!q_union[q]) continue;
            float q_floor = -INFINITY;
            bool floor_ok = floor_j >= 0;
            for (uint64_t c = clause_offsets[q]; c < clause_offsets[q + 1] && floor_ok; c++) {
                const Bm25ClauseDev &dc = dev_clauses[c];
                if (dc.occur != 0 || (dc.mode != NIDX_TF_FREQ && dc.mode != 1) || !(dc.weight > 0.0f) || !(dc.weight < INFINITY)) floor_ok = false;
            }
            for (uint64_t c = clause_offsets[q]; c < clause_offsets[q + 1] && floor_ok; c++) {
                const uint8_t fn = idx->floor_fn[(size_t)clauses[c].term * BM25_FLOOR_NR + (size_t)floor_j];
                if (fn != 255u) q_floor = std::max(q_floor, dev_clauses[c].weight * idx->quot1[fn]);
            }
            uint32_t floor_bits;
            memcpy(&floor_bits, &q_floor, 4);
            for (uint64_t c = clause_offsets[q]; c < clause_offsets[q + 1]; c++) {
                const uint64_t b = seg.term_offsets_host[clauses[c].term];
                const uint64_t l = clen[c];
                if (l > 0xffffffffull) return fail(NIDX_ERR_UNSUPPORTED, "a posting list of one segment holds more than 2^32 - 1 postings");
                ucl[c] = Bm25UClause{(uint32_t)b, (uint32_t)(b >> 32), (uint32_t)l, dev_clauses[c].weight,
                                     (uint32_t)dev_clauses[c].occur | ((uint32_t)dev_clauses[c].mode << 8), floor_bits, 0u, 0u};
            }
        }
        t_work += now_us() - t_w0;
        const double t_up0 = now_us();
        NIDX_HIP(cx.s_key.reserve(nw * kk * 8));
        const size_t if_bytes = ((size_t)(nq + 1) * 4 + 31) & ~(size_t)31;   // 32-byte pieces: the clause table is read with 16-byte scalar loads
        const size_t work_bytes = (nw * sizeof(Bm25Work) + 31) & ~(size_t)31;
        const size_t items_bytes = (nw * 4 + 31) & ~(size_t)31;
        const size_t inw_bytes = if_bytes + work_bytes + items_bytes + ucl.size() * sizeof(Bm25UClause);
        const size_t w_at = fused_in ? inq_al : 0;   // the work list's place in the block: behind the clause block, or at its head
        NIDX_HIP(cx.s_in_w.reserve(w_at + inw_bytes));
        NIDX_HIP(cx.h_in_w.reserve(w_at + inw_bytes));
        unsigned char *h_w = cx.h_in_w.as<unsigned char>() + w_at, *d_w = cx.s_in_w.as<unsigned char>() + w_at;
        if (fused_in) {
            if (n_clauses) memcpy(cx.h_in_w.p, dev_clauses.data(), n_clauses * sizeof(Bm25ClauseDev));
            memcpy(cx.h_in_w.as<unsigned char>() + cl_bytes, clause_offsets, (size_t)(nq + 1) * 8);
            d_clauses = cx.s_in_w.as<Bm25ClauseDev>();
            d_clause_offsets = reinterpret_cast<const unsigned long long *>(cx.s_in_w.as<unsigned char>() + cl_bytes);
        }
        memcpy(h_w, item_first.data(), (size_t)(nq + 1) * 4);
        memcpy(h_w + if_bytes, work.data(), nw * sizeof(Bm25Work));
        memcpy(h_w + if_bytes + work_bytes, item_list.data(), nw * 4);
        if (!ucl.empty()) memcpy(h_w + if_bytes + work_bytes + items_bytes, ucl.data(), ucl.size() * sizeof(Bm25UClause));
        NIDX_HIP(hipMemcpyAsync(cx.s_in_w.p, cx.h_in_w.p, w_at + inw_bytes, hipMemcpyHostToDevice, cx.stream));
        const uint32_t *d_item_first = reinterpret_cast<const uint32_t *>(d_w);
        const Bm25Work *d_work = reinterpret_cast<const Bm25Work *>(d_w + if_bytes);
        const uint32_t *d_items = reinterpret_cast<const uint32_t *>(d_w + if_bytes + work_bytes);
        // per-query outputs of the merge kernel, one block: doc u32 [nq][kk] | score f32 [nq][kk] | count u32 [nq] | total u64 [nq] | postings u64 [nq]
        const size_t o_doc = 0, o_score = (size_t)nq * kk * 4, o_count = o_score + (size_t)nq * kk * 4;
        const size_t o_total = (o_count + (size_t)nq * 4 + 7) & ~(size_t)7, o_post = o_total + (size_t)nq * 8, o_seg = o_post + (size_t)nq * 8;
        const bool cat = idx->concatenated();
        const size_t out_bytes = o_seg + (cat ? (size_t)nq * kk * 4 : 0);   // concatenated layout: + segment u32 [nq][kk]
        NIDX_HIP(cx.h_outpack.reserve(out_bytes));
        unsigned char *d_out = nullptr;
        if (zc_out) {
            void *dp = nullptr;
            NIDX_HIP(hipHostGetDevicePointer(&dp, cx.h_outpack.p, 0));
            d_out = static_cast<unsigned char *>(dp);
        } else {
            NIDX_HIP(cx.s_outpack.reserve(out_bytes));
            d_out = cx.s_outpack.as<unsigned char>();
        }
        NIDX_HIP(cx.s_count.reserve(nw * 4));
        NIDX_HIP(cx.s_total.reserve(nw * 8));
        NIDX_HIP(cx.s_postings.reserve(nw * 8));
        const uint32_t match_words = (seg.n_docs + 31) / 32;
        if (n_slots) {
            const uint64_t bytes = (uint64_t)n_slots * std::max<uint32_t>(match_words, 1) * 4;
            if (bytes > (1ull << 30))
                return fail(NIDX_ERR_UNSUPPORTED, "%u faceted queries over a %u-document segment need %llu bytes of match bitsets (limit 1 GiB): split the batch",
                            n_slots, seg.n_docs, (unsigned long long)bytes);
            NIDX_HIP(cx.s_match_bits.reserve(bytes));
            NIDX_HIP(hipMemsetAsync(cx.s_match_bits.p, 0, bytes, cx.stream));
        }
        const double t_up1 = now_us();
        Bm25Args a;
        a.work = d_work;
        a.n_docs = seg.n_docs;
        a.term_offsets = seg.term_offsets.as<unsigned long long>();
        a.doc_ids = seg.doc_ids.as<uint32_t>();
        a.tfs = seg.tfs.as<uint32_t>();
        a.fieldnorm_ids = seg.fieldnorm_ids.as<uint8_t>();
        a.alive = seg.all_alive ? nullptr : seg.alive.as<uint64_t>();
        a.tf_cache = idx->tf_cache.as<float>();
        a.clauses = d_clauses;
        a.clause_offsets = d_clause_offsets;
        a.after = after ? cx.s_after.as<Bm25AfterDev>() : nullptr;
        a.k = kk;
        a.segment_ord = (uint32_t)s;
        a.out_key = cx.s_key.as<unsigned long long>();
        a.out_count = cx.s_count.as<uint32_t>();
        a.out_total = cx.s_total.as<unsigned long long>();
        a.out_postings = cx.s_postings.as<unsigned long long>();
        a.aux_offsets = n_aux ? cx.s_aux_off.as<unsigned long long>() : nullptr;
        a.aux_doc_ids = n_aux ? cx.s_aux_ids.as<uint32_t>() : nullptr;
        a.aux_tfs = n_aux ? cx.s_aux_tfs.as<uint32_t>() : nullptr;
        a.order_key = order_field >= 0 ? seg.order_key[order_field].as<uint32_t>() : nullptr;
        a.order_desc = opt->order_desc ? 1 : 0;
        a.match_bits = n_slots ? cx.s_match_bits.as<uint32_t>() : nullptr;
        a.match_slot = n_slots ? cx.s_match_slot.as<int>() : nullptr;
        a.match_words = match_words;
        a.uclauses = reinterpret_cast<const Bm25UClause *>(d_w + if_bytes + work_bytes + items_bytes);
        a.dbg = nullptr;
        DevBuf dbgbuf;
        if (getenv("NIDX_GPU_BM25_DEBUG")) {
            NIDX_HIP(dbgbuf.alloc((16 + 8 * nw) * 8));
            NIDX_HIP(hipMemsetAsync(dbgbuf.p, 0, (16 + 8 * nw) * 8, cx.stream));
            a.dbg = dbgbuf.as<unsigned long long>();

        }
        Bm25MergeArgs mg;
        mg.item_first = d_item_first;
        mg.item_key = cx.s_key.as<unsigned long long>();
        mg.item_count = cx.s_count.as<uint32_t>();
        mg.item_total = cx.s_total.as<unsigned long long>();
        mg.item_postings = cx.s_postings.as<unsigned long long>();
        mg.k = kk;
        mg.out_doc = reinterpret_cast<uint32_t *>(d_out + o_doc);
        mg.out_score = reinterpret_cast<float *>(d_out + o_score);
        mg.out_count = reinterpret_cast<uint32_t *>(d_out + o_count);
        mg.out_total = reinterpret_cast<unsigned long long *>(d_out + o_total);
        mg.out_postings = reinterpret_cast<unsigned long long *>(d_out + o_post);
        if (cat) {
            mg.seg_base = idx->d_seg_base.as<uint32_t>();
            mg.n_seg = idx->n_segments;
            mg.out_seg = reinterpret_cast<uint32_t *>(d_out + o_seg);
        }
        const char *ab = getenv("NIDX_GPU_BM25_ABLATE_MERGE");   // measurement (wrong hits): 1 no merge launch, 2 an empty one, 3 one that writes nothing
        mg.ablate = ab ? atoi(ab) : 0;
        // When every item of the batch goes through the stream kernel and k <= 64, the scoring launch merges as well (Bm25FusedMerge): the last wave
        // of every query's slices does what bm25_merge_kernel would do in a launch of its own.  NIDX_GPU_BM25_FUSED_MERGE=0 keeps the two launches.
        bool fused_merge = n_union == nw && nw > 0 && kk <= 64 && !lockstep_union && !a.dbg && (mg.ablate == 0 || mg.ablate == 3);
        if (const char *fe = getenv("NIDX_GPU_BM25_FUSED_MERGE")) fused_merge = fused_merge && atoi(fe) != 0;
        if (fused_merge) {
            const size_t done_bytes = (size_t)nq * (BM25_FUSE_MAX_GROUPS + 1u) * 4;
            if (cx.s_fuse_done.bytes < done_bytes) {
                NIDX_HIP(cx.s_fuse_done.reserve(done_bytes));
                NIDX_HIP(hipMemsetAsync(cx.s_fuse_done.p, 0, cx.s_fuse_done.bytes, cx.stream));
            }
            NIDX_HIP(cx.s_fuse_key.reserve((size_t)nq * BM25_FUSE_MAX_GROUPS * kk * 8));
            NIDX_HIP(cx.s_fuse_count.reserve((size_t)nq * BM25_FUSE_MAX_GROUPS * 4));
            NIDX_HIP(cx.s_fuse_total.reserve((size_t)nq * BM25_FUSE_MAX_GROUPS * 8));
            NIDX_HIP(cx.s_fuse_postings.reserve((size_t)nq * BM25_FUSE_MAX_GROUPS * 8));
            a.fm.done = cx.s_fuse_done.as<uint32_t>();
            a.fm.g_key = cx.s_fuse_key.as<unsigned long long>();
            a.fm.g_count = cx.s_fuse_count.as<uint32_t>();
            a.fm.g_total = cx.s_fuse_total.as<unsigned long long>();
            a.fm.g_postings = cx.s_fuse_postings.as<unsigned long long>();
            a.fm.out_doc = mg.out_doc;
            a.fm.out_score = mg.out_score;
            a.fm.out_count = mg.out_count;
            a.fm.out_total = mg.out_total;
            a.fm.out_postings = mg.out_postings;
            a.fm.seg_base = mg.seg_base;
            a.fm.n_seg = mg.n_seg;
            a.fm.out_seg = mg.out_seg;
            a.fm.ablate = mg.ablate;
        }
        NIDX_HIP(hipEventRecord(cx.ev0, cx.stream));
        {
            const bool extras = a.alive != nullptr || a.match_bits != nullptr || a.order_key != nullptr || a.after != nullptr;
            NIDX_HIP((lockstep_union ? launch_bm25_union : launch_bm25_stream)(a, d_items, n_union, extras, cx.stream));
        }
        NIDX_HIP(launch_bm25_search(a, d_items + n_union, n_fast, d_items + n_union + n_fast, n_wide, wide_max_clauses, cx.stream));
        NIDX_HIP(hipEventRecord(cx.ev1, cx.stream));
        if (!fused_merge && mg.ablate != 1) NIDX_HIP(launch_bm25_merge(mg, nq, cx.stream));
        if (n_slots)
            NIDX_HIP(launch_facet_count(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), cx.s_pair_term.as<uint32_t>(),
                                        cx.s_pair_slot.as<int>(), (uint32_t)n_pairs, cx.s_match_bits.as<uint32_t>(), match_words,
                                        cx.s_facet_counts.as<unsigned long long>(), cx.stream));
        if (!zc_out) NIDX_HIP(hipMemcpyAsync(cx.h_outpack.p, cx.s_outpack.p, out_bytes, hipMemcpyDeviceToHost, cx.stream));
        const unsigned char *h_out = cx.h_outpack.as<unsigned char>();
        const uint32_t *h_doc = reinterpret_cast<const uint32_t *>(h_out + o_doc);
        const float *h_score = reinterpret_cast<const float *>(h_out + o_score);
        const uint32_t *h_count = reinterpret_cast<const uint32_t *>(h_out + o_count);
        const unsigned long long *h_total = reinterpret_cast<const unsigned long long *>(h_out + o_total);
        const unsigned long long *h_post = reinterpret_cast<const unsigned long long *>(h_out + o_post);
        const uint32_t *h_seg = reinterpret_cast<const uint32_t *>(h_out + o_seg);   // (concatenated layout only)
        if (async_slot && idx->segs.size() == 1 && n_aux == 0 && n_slots == 0 && order_field < 0 && !a.dbg) {
            // pipelined: the collect step runs in nidx_gpu_bm25_search_wait (the buffers are the slot's: swapped in by submit)
            if (host_dbg)
                fprintf(stderr, "[bm25 host submit] total=%.0f us: clauses=%.0f set-up=%.0f work list=%.0f staging+upload=%.0f launches=%.0f\n", now_us() - t_entry,
                        t_begin - t_entry, t_seg0 - t_begin, t_work, t_up1 - t_up0, now_us() - t_up1);
            Bm25Slot &sl = *async_slot;
            sl.launched = true;
            sl.nq = nq, sl.k = k, sl.kk = kk;
            sl.o_doc = o_doc, sl.o_score = o_score, sl.o_count = o_count, sl.o_total = o_total, sl.o_post = o_post, sl.o_seg = o_seg;
            return NIDX_OK;
        }
        const double t_s0 = now_us();
        NIDX_HIP(hipStreamSynchronize(cx.stream));
        t_sync += now_us() - t_s0;
        const double t_c0 = now_us();
        float ms = 0.f;
        NIDX_HIP(hipEventElapsedTime(&ms, cx.ev0, cx.ev1));
        cx.kernel_ms += ms;
        if (a.dbg) {
            unsigned long long d[9];
            NIDX_HIP(hipMemcpy(d, a.dbg, 72, hipMemcpyDeviceToHost));
            if (n_union) {
                std::vector<unsigned long long> tr(8 * nw);
                NIDX_HIP(hipMemcpy(tr.data(), a.dbg + 16, tr.size() * 8, hipMemcpyDeviceToHost));
                std::vector<std::pair<unsigned long long, uint32_t>> by;
                for (uint32_t w = 0; w < nw; w++)
                    if (tr[8 * w]) by.push_back({tr[8 * w], w});
                std::sort(by.begin(), by.end());
                auto pct = [&](double p) { return by.empty() ? 0ull : by[(size_t)(p * (by.size() - 1))].first; };
                unsigned long long t_min = ~0ull, t_max = 0;
                for (auto &e : by) {
                    const unsigned long long st = tr[8 * e.second + 3] & 0xffffffffull;
                    t_min = std::min(t_min, st);
                    t_max = std::max(t_max, st);
                }
                fprintf(stderr, "[bm25 dbg] union items %zu: cycles p10 %llu p50 %llu p90 %llu p99 %llu max %llu; entry clocks (low 32 bits) span %llu\n", by.size(),
                        pct(0.1), pct(0.5), pct(0.9), pct(0.99), pct(1.0), t_max - t_min);
                {   // phase cycles per window by the number of slices of the item's query
                    const uint32_t edges[5] = {1, 2, 4, 16, 0xffffffffu};
                    for (int g = 0; g < 5; g++) {
                        unsigned long long n = 0, wins = 0, ld = 0, fi = 0, fn = 0, rs = 0, st = 0, po = 0;
                        for (auto &e : by) {
                            const uint32_t w = e.second, ns = work[w].n_slices;
                            if (ns > edges[g] || (g > 0 && ns <= edges[g - 1])) continue;
                            n++, wins += tr[8 * w + 2] & 0xffffffffull, ld += tr[8 * w + 4], fi += tr[8 * w + 5], fn += tr[8 * w + 6], rs += tr[8 * w + 7], st += tr[8 * w + 1], po += tr[8 * w + 3] >> 32;
                        }
                        if (n) fprintf(stderr, "[bm25 dbg]   queries of <= %u slices: %llu items, %.1f windows and %.0f postings per item, setup %.0f; per window: load %.0f filter %.0f final %.0f resolve %.0f cycles\n",
                                       edges[g], n, (double)wins / n, (double)po / n, (double)st / n, (double)ld / wins, (double)fi / wins, (double)fn / wins, (double)rs / wins);
                    }
                }
                for (size_t i = by.size() >= 6 ? by.size() - 6 : 0; i < by.size(); i++) {
                    const uint32_t w = by[i].second;
                    fprintf(stderr, "[bm25 dbg]   slow item %u: %llu cycles (setup %llu; load %llu filter %llu final %llu resolve %llu), %llu windows, %llu inserts, %llu postings, query %u slice %u/%u\n", w, tr[8 * w],
                            tr[8 * w + 1], tr[8 * w + 4], tr[8 * w + 5], tr[8 * w + 6], tr[8 * w + 7], tr[8 * w + 2] & 0xffffffffull, tr[8 * w + 2] >> 32, tr[8 * w + 3] >> 32, work[w].query, work[w].slice, work[w].n_slices);
                }
                for (size_t i = 0; i < std::min<size_t>(3, by.size()); i++) {
                    const uint32_t w = by[i].second;
                    fprintf(stderr, "[bm25 dbg]   fast item %u: %llu cycles (setup %llu; load %llu filter %llu final %llu resolve %llu), %llu windows, %llu inserts, %llu postings, query %u slice %u/%u\n", w, tr[8 * w],
                            tr[8 * w + 1], tr[8 * w + 4], tr[8 * w + 5], tr[8 * w + 6], tr[8 * w + 7], tr[8 * w + 2] & 0xffffffffull, tr[8 * w + 2] >> 32, tr[8 * w + 3] >> 32, work[w].query, work[w].slice, work[w].n_slices);
                }
            }
            fprintf(stderr, "[bm25 dbg] items=%llu windows=%llu (max %llu per item) cycles/item: setup=%llu load=%llu apply=%llu fold=%llu windows total=%llu; longest item (entry to exit) %llu cycles; kernel_ms=%.3f\n",
                    d[5], d[4], d[8], d[6] / (d[5] ? d[5] : 1), d[0] / (d[5] ? d[5] : 1), d[1] / (d[5] ? d[5] : 1), d[2] / (d[5] ? d[5] : 1), d[3] / (d[5] ? d[5] : 1), d[7], ms);
        }
        const bool direct = idx->segs.size() == 1;   // one segment: the device list is the answer, no host merge
        for (uint32_t q = 0; q < nq; q++) {  // per segment the device already merged the slices
            if (out_total) out_total[q] += h_total[q];
            if (out_postings) out_postings[q] += h_post[q];
            if (k == 0) continue;
            if (direct) {
                const uint32_t n = std::min<uint32_t>(h_count[q], k);
                out_count[q] = n;
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t d = h_doc[(size_t)q * kk + i];   // concatenated layout: already the doc inside its segment
                    const uint32_t sg = cat ? h_seg[(size_t)q * kk + i] : (uint32_t)s;
                    if (out_docaddr) out_docaddr[(size_t)q * k + i] = ((uint64_t)sg << 32) | d;
                    if (out_score) out_score[(size_t)q * k + i] = order_field >= 0 ? 0.f : h_score[(size_t)q * kk + i];
                    if (opt->out_order_value)
                        opt->out_order_value[(size_t)q * k + i] = order_field >= 0 ? seg.fast_host[order_field][cat ? idx->seg_base[sg] + d : d] : 0;
                }
                continue;
            }
            for (uint32_t i = 0; i < h_count[q]; i++) {
                const uint32_t d = h_doc[(size_t)q * kk + i];
                if (order_field >= 0) merged[q].push_back(Hit{0.f, ((uint64_t)s << 32) | d, seg.fast_host[order_field][d]});
                else merged[q].push_back(Hit{h_score[(size_t)q * kk + i], ((uint64_t)s << 32) | d, 0});
            }
        }
        t_collect += now_us() - t_c0;
    }
    const double t_merge0 = now_us();
    if (n_slots) {
        std::vector<unsigned long long> fc(n_pairs);
        NIDX_HIP(hipMemcpy(fc.data(), cx.s_facet_counts.p, n_pairs * 8, hipMemcpyDeviceToHost));
        for (uint64_t p = 0; p < n_pairs; p++) opt->out_facet_counts[p] = fc[p];
    }
    const bool desc = opt->order_desc != 0;
    const bool single_segment = idx->segs.size() == 1;
    for (uint32_t q = 0; q < nq && k > 0 && !single_segment; q++) {
        std::vector<Hit> &m = merged[q];
        if (single_segment) {
            // one segment: the device list is already in TopDocs order
        } else if (order_field >= 0) {
            // order_by_fast_field across segments: the fast value, then DocAddress asc
            std::sort(m.begin(), m.end(), [desc](const Hit &x, const Hit &y) {
                if (x.value != y.value) return desc ? x.value > y.value : x.value < y.value;
                return x.docaddr < y.docaddr;
            });
        } else {
            // TopDocs order across segments: score desc (total order), DocAddress asc
            std::sort(m.begin(), m.end(), [](const Hit &x, const Hit &y) {
                int32_t kx = total_key(x.score), ky = total_key(y.score);
                if (kx != ky) return kx > ky;
                return x.docaddr < y.docaddr;
            });
        }
        uint32_t n = (uint32_t)std::min<size_t>(m.size(), k);
        out_count[q] = n;
        for (uint32_t i = 0; i < n; i++) {
            if (out_docaddr) out_docaddr[(size_t)q * k + i] = m[i].docaddr;
            if (out_score) out_score[(size_t)q * k + i] = m[i].score;
            if (opt->out_order_value) opt->out_order_value[(size_t)q * k + i] = m[i].value;
        }
    }
    if (host_dbg)
        fprintf(stderr, "[bm25 host] total=%.0f us: clauses=%.0f work list=%.0f sync wait=%.0f collect=%.0f merge=%.0f\n", now_us() - t_entry, t_begin - t_entry,
                t_work, t_sync, t_collect, now_us() - t_merge0);
    return NIDX_OK;
}

extern "C" {

int32_t nidx_gpu_bm25_search_ex(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses, const uint64_t *clause_offsets,
                                uint32_t nq, const nidx_gpu_bm25_search_options_t *opt, uint64_t *out_docaddr, float *out_score,
                                uint32_t *out_count, uint64_t *out_total, uint64_t *out_postings) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !clause_offsets || !out_count || !opt) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    std::shared_lock<std::shared_mutex> rlock(idx->rw);
    const int32_t rc = bm25_search_locked(idx, idx->main, nullptr, clauses, clause_offsets, nq, opt, out_docaddr, out_score, out_count, out_total, out_postings);
    {
        std::lock_guard<std::mutex> ms(idx->ms_mu);
        idx->last_kernel_ms = idx->main.kernel_ms;
    }
    return rc;
} NIDX_ABI_CATCH

// ---- pipelined form: the host side of batch i + 1 (clause weights, work list, staging) overlaps the kernels of batch i; every slot
// has a context of its own, so two or more threads may submit at the same time (the planning of a batch costs the submitting thread
// more than its kernels cost the device) ----------
#define NIDX_BM25_PIPELINE_DEPTH NIDX_GPU_BM25_MAX_TICKETS   /* include/nidx_gpu.h */

int32_t nidx_gpu_bm25_search_submit(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses, const uint64_t *clause_offsets,
                                    uint32_t nq, const nidx_gpu_bm25_search_options_t *opt, uint64_t *ticket_out) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !clause_offsets || !opt || !ticket_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *ticket_out = 0;
    NIDX_HIP(hipSetDevice(idx->device));
    Bm25Slot *slot = nullptr;
    {
        std::lock_guard<std::mutex> lock(idx->slots_mu);
        for (auto &s : idx->slots)
            if (!s->busy && !s->preparing) { slot = s.get(); break; }
        if (!slot) {
            if (idx->slots.size() >= NIDX_BM25_PIPELINE_DEPTH)
                return fail(NIDX_ERR_BUSY, "all %d BM25 pipeline slots hold a ticket that has not been waited for", NIDX_BM25_PIPELINE_DEPTH);
            std::unique_ptr<Bm25Slot> ns(new Bm25Slot());
            NIDX_HIP(ns->cx.init());
            idx->slots.push_back(std::move(ns));
            slot = idx->slots.back().get();
        }
        slot->preparing = true;
    }
    // every way out of the preparation — an error code, a std::bad_alloc from its host vectors — gives the slot back, and a failed
    // batch leaves nothing queued on the slot's stream that could still read the staging the next submit rewrites
    struct Prepare {
        Bm25Index *idx;
        Bm25Slot *slot;
        bool ok = false;
        ~Prepare() {
            if (!ok) (void)hipStreamSynchronize(slot->cx.stream);
            std::lock_guard<std::mutex> lock(idx->slots_mu);
            slot->preparing = false;
            if (ok) {
                slot->busy = true;
                slot->ticket = idx->next_ticket++;
            }
        }
    };
    const uint32_t k = opt->k;
    int32_t rc;
    {
        Prepare prep{idx, slot};
        slot->launched = false;
        slot->nq = nq, slot->k = k;
        // outputs of a request that runs synchronously inside this call
        const size_t kk1 = std::max<uint32_t>(k, 1);
        // (resize, not assign: the search zeroes the counts itself and nothing is read beyond a query's count)
        slot->r_docaddr.resize((size_t)nq * kk1), slot->r_score.resize((size_t)nq * kk1);
        slot->r_count.resize(nq), slot->r_total.resize(nq), slot->r_postings.resize(nq);
        std::shared_lock<std::shared_mutex> rlock(idx->rw);
        rc = bm25_search_locked(idx, slot->cx, slot, clauses, clause_offsets, nq, opt, slot->r_docaddr.data(), slot->r_score.data(), slot->r_count.data(),
                                slot->r_total.data(), slot->r_postings.data());
        prep.ok = rc == NIDX_OK;
    }
    if (rc != NIDX_OK) return rc;
    *ticket_out = slot->ticket;   // (set by ~Prepare; the slot is this thread's until its ticket is handed out)
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_search_wait(nidx_gpu_bm25_index_t *index, uint64_t ticket, uint64_t *out_docaddr, float *out_score, uint32_t *out_count,
                                  uint64_t *out_total, uint64_t *out_postings) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || !out_count) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    Bm25Slot *slot = nullptr;
    {
        std::lock_guard<std::mutex> lock(idx->slots_mu);
        for (auto &s : idx->slots)
            if (s->busy && s->ticket == ticket && ticket != 0) { slot = s.get(); break; }
        if (!slot) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown ticket %llu (a ticket is waited for once)", (unsigned long long)ticket);
        slot->ticket = 0;   // nobody else finds it; it stays busy until its results are out
    }
    struct Release {
        Bm25Index *idx;
        Bm25Slot *slot;
        ~Release() {
            std::lock_guard<std::mutex> lock(idx->slots_mu);
            slot->busy = false;
        }
    } release{idx, slot};
    const uint32_t nq = slot->nq, k = slot->k;
    if (slot->launched) {
        NIDX_HIP(hipSetDevice(idx->device));
        NIDX_HIP(hipStreamSynchronize(slot->cx.stream));
        float ms = 0.f;
        NIDX_HIP(hipEventElapsedTime(&ms, slot->cx.ev0, slot->cx.ev1));
        {
            std::lock_guard<std::mutex> lock(idx->ms_mu);
            idx->last_kernel_ms = ms;
        }
        const unsigned char *h_out = slot->cx.h_outpack.as<unsigned char>();
        const uint32_t *h_doc = reinterpret_cast<const uint32_t *>(h_out + slot->o_doc);
        const float *h_score = reinterpret_cast<const float *>(h_out + slot->o_score);
        const uint32_t *h_count = reinterpret_cast<const uint32_t *>(h_out + slot->o_count);
        const unsigned long long *h_total = reinterpret_cast<const unsigned long long *>(h_out + slot->o_total);
        const unsigned long long *h_post = reinterpret_cast<const unsigned long long *>(h_out + slot->o_post);
        const uint32_t kk = slot->kk;
        const bool cat = idx->concatenated();
        const uint32_t *h_seg = reinterpret_cast<const uint32_t *>(h_out + slot->o_seg);   // (concatenated layout only)
        for (uint32_t q = 0; q < nq; q++) {   // one resident segment: the device list is the answer
            if (out_total) out_total[q] = h_total[q];
            if (out_postings) out_postings[q] = h_post[q];
            const uint32_t n = k ? std::min<uint32_t>(h_count[q], k) : 0u;
            out_count[q] = n;
            for (uint32_t i = 0; i < n; i++) {
                const uint64_t sg = cat ? h_seg[(size_t)q * kk + i] : 0u;   // (the merge kernel already split resident docs into segment, doc)
                if (out_docaddr) out_docaddr[(size_t)q * k + i] = (sg << 32) | h_doc[(size_t)q * kk + i];
                if (out_score) out_score[(size_t)q * k + i] = h_score[(size_t)q * kk + i];
            }
        }
        return NIDX_OK;
    }
    for (uint32_t q = 0; q < nq; q++) {
        out_count[q] = slot->r_count[q];
        if (out_total) out_total[q] = slot->r_total[q];
        if (out_postings) out_postings[q] = slot->r_postings[q];
        for (uint32_t i = 0; i < slot->r_count[q]; i++) {
            if (out_docaddr) out_docaddr[(size_t)q * k + i] = slot->r_docaddr[(size_t)q * k + i];
            if (out_score) out_score[(size_t)q * k + i] = slot->r_score[(size_t)q * k + i];
        }
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

// open_index_with_deletions (nidx_tantivy/src/index_reader.rs:39-74): the documents of every posting list in `terms` (the
// TermSetQuery the DeletionQueryBuilder makes of the deletion keys newer than the segment) leave the segment's alive bitset.
int32_t nidx_gpu_bm25_apply_deletions(nidx_gpu_bm25_index_t *index, uint32_t segment, const uint32_t *terms, uint32_t n_terms,
                                      uint64_t *n_alive_out) try {
    Bm25Index *idx = reinterpret_cast<Bm25Index *>(index);
    if (!idx || segment >= idx->n_segments || (n_terms && !terms)) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    std::lock_guard<std::mutex> lock(idx->mu);
    std::unique_lock<std::shared_mutex> wlock(idx->rw);
    NIDX_HIP(hipSetDevice(idx->device));
    if (int32_t rc_drain = bm25_drain_tickets(idx)) return rc_drain;
    NIDX_HIP(hipSetDevice(idx->device));
    for (uint32_t i = 0; i < n_terms; i++)
        if (terms[i] >= idx->n_terms) return fail(NIDX_ERR_INVALID_ARGUMENT, "deletion term id %u out of range", terms[i]);
    const bool cat = idx->concatenated();
    Bm25Segment &seg = idx->segs[cat ? 0 : segment];
    Bm25Ctx &cx = idx->main;
    const uint32_t words = (seg.n_docs + 63) / 64;
    hipStream_t st = cx.stream;
    // the documents the count below is taken over: the segment's own (a doc-id range of the concatenated layout)
    const uint32_t d0 = cat ? idx->seg_base[segment] : 0u, d1 = cat ? idx->seg_base[segment + 1] : seg.n_docs;
    if (n_terms && words) {
        if (seg.all_alive) {
            NIDX_HIP(seg.alive.alloc((size_t)words * 8));
            NIDX_HIP(launch_bitset_fill(seg.alive.as<uint64_t>(), words, seg.n_docs, 1, st));
            seg.all_alive = false;
        }
        NIDX_HIP(cx.s_set_bits.reserve((size_t)words * 8));
        NIDX_HIP(launch_bitset_fill(cx.s_set_bits.as<uint64_t>(), words, seg.n_docs, 0, st));
        if (!cat) {
            NIDX_HIP(cx.s_set_terms.reserve((size_t)n_terms * 4));
            NIDX_HIP(hipMemcpyAsync(cx.s_set_terms.p, terms, (size_t)n_terms * 4, hipMemcpyHostToDevice, st));
            NIDX_HIP(launch_bitset_scatter(seg.term_offsets.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), cx.s_set_terms.as<uint32_t>(),
                                           n_terms, seg.n_docs, cx.s_set_bits.as<uint64_t>(), st));
        } else {
            // only THIS segment's part of every list: list i = [ranges[2 i], ranges[2 i + 1]) of the resident postings
            std::vector<unsigned long long> ranges(2 * (size_t)n_terms + 1, 0);
            std::vector<uint32_t> lists(n_terms);
            for (uint32_t i = 0; i < n_terms; i++) {
                const uint32_t t = terms[i];
                unsigned long long at = seg.term_offsets_host[t];
                for (uint32_t s = 0; s < segment; s++) at += idx->real[s].term_offsets_host[t + 1] - idx->real[s].term_offsets_host[t];
                ranges[2 * i] = at;
                ranges[2 * i + 1] = at + (idx->real[segment].term_offsets_host[t + 1] - idx->real[segment].term_offsets_host[t]);
                lists[i] = 2 * i;
            }
            NIDX_HIP(cx.s_set_terms.reserve((size_t)n_terms * 4));
            NIDX_HIP(cx.s_aux_off.reserve(ranges.size() * 8));
            NIDX_HIP(hipMemcpyAsync(cx.s_set_terms.p, lists.data(), (size_t)n_terms * 4, hipMemcpyHostToDevice, st));
            NIDX_HIP(hipMemcpyAsync(cx.s_aux_off.p, ranges.data(), ranges.size() * 8, hipMemcpyHostToDevice, st));
            NIDX_HIP(launch_bitset_scatter(cx.s_aux_off.as<unsigned long long>(), seg.doc_ids.as<uint32_t>(), cx.s_set_terms.as<uint32_t>(), n_terms,
                                           seg.n_docs, cx.s_set_bits.as<uint64_t>(), st));
            NIDX_HIP(hipStreamSynchronize(st));   // `ranges` / `lists` are pageable host memory
        }
        NIDX_HIP(launch_bitset_binop(seg.alive.as<uint64_t>(), cx.s_set_bits.as<uint64_t>(), words, 2, st));
        seg.n_alive = -1;
    }
    if (n_alive_out) {
        if (seg.all_alive) *n_alive_out = d1 - d0;
        else {
            NIDX_HIP(cx.s_set_bits.reserve(std::max<size_t>(words, 1) * 8 + 8));
            NIDX_HIP(cx.s_set_counts.reserve(8));
            NIDX_HIP(hipMemsetAsync(cx.s_set_counts.p, 0, 8, st));
            if (!cat) {
                NIDX_HIP(launch_bitset_and_count(seg.alive.as<uint64_t>(), nullptr, cx.s_set_bits.as<uint64_t>(), words,
                                                 cx.s_set_counts.as<unsigned long long>(), st));
            } else {
                // alive AND [d0, d1): the range as a bitset (ranks of an identity key would do the same; a fill + two shifts is less)
                std::vector<uint64_t> mask(std::max<size_t>(words, 1), 0);
                for (uint64_t d = d0; d < d1;) {
                    if ((d & 63) == 0 && d + 64 <= d1) { mask[d >> 6] = ~0ull; d += 64; }
                    else { mask[d >> 6] |= 1ull << (d & 63); d++; }
                }
                NIDX_HIP(cx.s_sub_bits.reserve(std::max<size_t>(words, 1) * 8));
                NIDX_HIP(hipMemcpyAsync(cx.s_sub_bits.p, mask.data(), (size_t)words * 8, hipMemcpyHostToDevice, st));
                NIDX_HIP(launch_bitset_and_count(cx.s_sub_bits.as<uint64_t>(), seg.alive.as<uint64_t>(), cx.s_set_bits.as<uint64_t>(), words,
                                                 cx.s_set_counts.as<unsigned long long>(), st));
                NIDX_HIP(hipStreamSynchronize(st));
            }
            unsigned long long c = 0;
            NIDX_HIP(hipMemcpyAsync(&c, cx.s_set_counts.p, 8, hipMemcpyDeviceToHost, st));
            NIDX_HIP(hipStreamSynchronize(st));
            *n_alive_out = c;
            if (!cat) seg.n_alive = (int64_t)c;
        }
    }
    NIDX_HIP(hipStreamSynchronize(st));
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_bm25_search(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses, const uint64_t *clause_offsets,
                             uint32_t nq, uint32_t k, const nidx_gpu_bm25_search_after_t *after, uint64_t *out_docaddr, float *out_score,
                             uint32_t *out_count, uint64_t *out_total, uint64_t *out_postings) try {
    nidx_gpu_bm25_search_options_t opt;
    memset(&opt, 0, sizeof(opt));
    opt.k = k;
    opt.after = after;
    opt.order_field = -1;
    return nidx_gpu_bm25_search_ex(index, clauses, clause_offsets, nq, &opt, out_docaddr, out_score, out_count, out_total, out_postings);
} NIDX_ABI_CATCH

}  // extern "C"
