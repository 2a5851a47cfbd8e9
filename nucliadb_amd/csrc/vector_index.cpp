// vector_index.cpp — C ABI implementation of the vector index (include/nidx_gpu.h).
//
// Host-side mirror of nidx_vector's Searcher / OpenSegment glue around the gfx950 kernels:
//   open      VectorSearcher::open -> segment::open + apply_deletions   (lib.rs:126-200, segment.rs:39-90)
//   search    Searcher::_search: sequential segments + Fssc merge        (searcher.rs:149-199,241-290)
//             OpenSegment::_search: filter ∩ alive, use_hnsw routing      (segment.rs:496-567,626-660)
// All arithmetic on vectors happens in the kernels; this file only moves data, routes and merges.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <chrono>
#include <memory>
#include <mutex>

#include "device_common.h"
#include "hnsw_graph.h"
#include "host_common.h"
#include "kernels.h"
#include "vector_index.h"

namespace nidx {

// ---- errors -----------------------------------------------------------------------------------
// Fixed storage: recording an error never allocates, so it also works from the out-of-memory handler below.
static thread_local char g_last_error[512];

static void record_error(const char *fmt, va_list ap) { vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap); }

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    record_error(fmt, ap);
    va_end(ap);
}
int32_t fail(int32_t code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    record_error(fmt, ap);
    va_end(ap);
    return code;
}
// The catch handler of every entry point (NIDX_ABI_CATCH): the callers are Rust / cgo / ctypes frames that cannot unwind, so
// what the host containers throw (allocation failure, a length derived from a corrupt input) leaves as an error code.
int32_t abi_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(NIDX_ERR_OUT_OF_MEMORY, "host allocation failed");
    } catch (const std::length_error &e) {
        return fail(NIDX_ERR_OUT_OF_MEMORY, "host allocation failed: %s", e.what());
    } catch (const std::exception &e) {
        return fail(NIDX_ERR_INTERNAL, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(NIDX_ERR_INTERNAL, "unexpected exception");
    }
}
int32_t hip_fail(hipError_t e, const char *what) {
    return fail(NIDX_ERR_DEVICE, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

// ---- use_hnsw (segment.rs:626-660) ----------------------------------------------------------------
bool use_hnsw(uint64_t total_nodes, uint64_t matching_nodes, uint64_t top_k, bool has_rabitq) {
    const uint64_t RERANKING_FACTOR = 100;  // rabitq.rs:30-36
    uint64_t full_cost, search_mult, rerank_mult;
    if (has_rabitq) {
        full_cost = 16;
        search_mult = RERANKING_FACTOR * 3 / 4;
        rerank_mult = RERANKING_FACTOR / 2;
    } else {
        full_cost = 1;
        search_mult = 1;
        rerank_mult = 0;
    }
    float l = logf((float)total_nodes) - 2.0f;
    float hnsw_rq = (l * l) * logf((float)top_k) * (float)search_mult;
    uint64_t hnsw_full = top_k * rerank_mult + (matching_nodes ? top_k * NIDX_M * total_nodes / matching_nodes : 0);
    // Rust `as usize`: NaN -> 0, negative -> 0, saturating
    uint64_t hnsw_rq_u;
    if (!(hnsw_rq > 0.0f)) hnsw_rq_u = 0;
    else if (hnsw_rq >= 18446744073709551615.0f) hnsw_rq_u = UINT64_MAX;
    else hnsw_rq_u = (uint64_t)hnsw_rq;
    uint64_t hnsw_cost = hnsw_rq_u + hnsw_full * full_cost;
    uint64_t bf_cost = matching_nodes + top_k * rerank_mult * full_cost;
    return hnsw_cost < bf_cost;
}

// ---- utils::normalize_vector (utils.rs:20-23): f32 fold of x.powi(2), then x / sqrt ---------------
void normalize_row(const float *in, float *out, uint32_t d) {
    float acc = 0.0f;
    for (uint32_t i = 0; i < d; i++) {
        float sq = in[i] * in[i];
        acc = acc + sq;
    }
    float mag = sqrtf(acc);
    for (uint32_t i = 0; i < d; i++) out[i] = in[i] / mag;
}

static uint64_t popcount_and(const uint64_t *a, const uint64_t *b, uint32_t nbits) {
    uint64_t c = 0;
    uint32_t words = (nbits + 63) / 64;
    for (uint32_t w = 0; w < words; w++) {
        uint64_t x = a ? a[w] : ~0ull;
        if (b) x &= b[w];
        if (w == words - 1 && (nbits & 63)) x &= (1ull << (nbits & 63)) - 1ull;
        c += (uint64_t)__builtin_popcountll(x);
    }
    return c;
}

// ---- open ---------------------------------------------------------------------------------------
int32_t VectorSegment::upload_graph(const HostGraph &hg) {
    NIDX_HIP(g_l0.alloc(hg.l0.size() * 4));
    NIDX_HIP(g_upper_base.alloc(hg.upper_base.size() * 4));
    NIDX_HIP(g_upper.alloc(std::max<size_t>(hg.upper.size(), NIDX_UP_STRIDE) * 4));
    NIDX_HIP(hipMemcpy(g_l0.p, hg.l0.data(), hg.l0.size() * 4, hipMemcpyHostToDevice));
    NIDX_HIP(hipMemcpy(g_upper_base.p, hg.upper_base.data(), hg.upper_base.size() * 4, hipMemcpyHostToDevice));
    if (!hg.upper.empty()) NIDX_HIP(hipMemcpy(g_upper.p, hg.upper.data(), hg.upper.size() * 4, hipMemcpyHostToDevice));
    ep_node = hg.ep_node;
    ep_layer = hg.ep_layer;
    top_layer = hg.top_layer;
    has_graph = true;
    return NIDX_OK;
}

GraphDev VectorSegment::graph_dev() const {
    GraphDev g;
    g.l0 = g_l0.as<uint32_t>();
    g.upper_base = g_upper_base.as<uint32_t>();
    g.upper = g_upper.as<uint32_t>();
    g.ep_node = ep_node;
    g.ep_layer = ep_layer;
    g.n = n;
    return g;
}

SegDev VectorSegment::seg_dev(int similarity) const {
    SegDev s;
    s.vectors = vectors.as<float>();
    s.norm2 = norm2.as<float>();
    s.n = n;
    s.dp = dp;
    s.dim = dim;
    s.para_of_vec = identity_para ? nullptr : para_of_vec.as<uint32_t>();
    s.alive = all_alive ? nullptr : alive.as<uint64_t>();
    s.similarity = similarity;
    return s;
}

uint64_t VectorSegment::bytes() const {
    return vectors.bytes + norm2.bytes + norm2_serial.bytes + vectors16.bytes + para_of_vec.bytes + alive.bytes + g_l0.bytes + g_upper_base.bytes +
           g_upper.bytes + g_l0_w.bytes + g_upper_w.bytes + f_offsets.bytes + f_ids.bytes + f_key_bytes.bytes + f_key_offsets.bytes + quant.bytes;
}

static int32_t open_segment(const nidx_gpu_vector_config_t &cfg, const nidx_gpu_vector_segment_t &in, VectorSegment &seg,
                            hipStream_t stream) {
    const uint32_t d = cfg.dimension;
    seg.n = in.n_vectors;
    seg.dim = d;
    seg.dp = (d + 3u) & ~3u;
    seg.n_paragraphs = in.n_paragraphs;
    const uint64_t packed = (uint64_t)d * 4, trailer = packed + 4;
    if (in.n_vectors > 0 && in.vectors == nullptr) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment has no vectors pointer");
    if (in.row_stride_bytes < packed)
        // a store written for another dimension: segment::create rejects it (segment.rs:200-211)
        return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "Inconsistent dimensions. Index=%u Vector=%llu", d,
                    (unsigned long long)(in.row_stride_bytes / 4));
    // paragraph of each vector
    std::vector<uint32_t> pov;
    if (in.paragraph_of_vector) {
        pov.assign(in.paragraph_of_vector, in.paragraph_of_vector + in.n_vectors);
    } else if (in.row_stride_bytes == trailer) {
        pov.resize(in.n_vectors);
        const uint8_t *base = (const uint8_t *)in.vectors;
        for (uint32_t i = 0; i < in.n_vectors; i++) memcpy(&pov[i], base + (uint64_t)i * trailer + packed, 4);
    }
    seg.identity_para = true;
    for (uint32_t i = 0; i < pov.size(); i++)
        if (pov[i] != i) { seg.identity_para = false; break; }
    if (pov.empty() && in.n_paragraphs != in.n_vectors)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "packed vectors need n_paragraphs == n_vectors or paragraph_of_vector");
    seg.vmax = 1;
    if (!pov.empty()) {
        // StoredParagraph { first_vector, num_vectors } (data_store/v2/paragraph_store.rs:36-64): a paragraph's vectors are
        // one contiguous run; more than one only with VectorCardinality::Multi
        std::vector<uint32_t> first(in.n_paragraphs, 0), num(in.n_paragraphs, 0);
        for (uint32_t i = 0; i < in.n_vectors; i++) {
            const uint32_t p = pov[i];
            if (p >= in.n_paragraphs) return fail(NIDX_ERR_INVALID_ARGUMENT, "paragraph address out of range");
            if (num[p] == 0) first[p] = i;
            else if (pov[i - 1] != p) return fail(NIDX_ERR_INVALID_ARGUMENT, "the vectors of paragraph %u are not contiguous", p);
            num[p]++;
            if (num[p] > 1 && cfg.vector_cardinality != NIDX_CARDINALITY_MULTI)
                return fail(NIDX_ERR_INVALID_CONFIGURATION, "paragraph %u owns %u vectors but the index is VectorCardinality::Single", p, num[p]);
            seg.vmax = std::max(seg.vmax, num[p]);
        }
        if (!seg.identity_para) {
            NIDX_HIP(seg.para_first.alloc(std::max<size_t>(in.n_paragraphs, 1) * 4));
            NIDX_HIP(seg.para_num.alloc(std::max<size_t>(in.n_paragraphs, 1) * 4));
            if (in.n_paragraphs) {
                NIDX_HIP(hipMemcpy(seg.para_first.p, first.data(), (size_t)in.n_paragraphs * 4, hipMemcpyHostToDevice));
                NIDX_HIP(hipMemcpy(seg.para_num.p, num.data(), (size_t)in.n_paragraphs * 4, hipMemcpyHostToDevice));
            }
        }
    }
    seg.para_host = pov;
    if (!seg.identity_para) {
        NIDX_HIP(seg.para_of_vec.alloc((size_t)in.n_vectors * 4));
        NIDX_HIP(hipMemcpy(seg.para_of_vec.p, pov.data(), (size_t)in.n_vectors * 4, hipMemcpyHostToDevice));
    }
    // vectors -> [n][dp], zero padded; the 4-byte paragraph trailer of vectors.bin is dropped
    if (in.n_vectors > 0) {
        NIDX_HIP(seg.vectors.alloc((size_t)in.n_vectors * seg.dp * 4));
        if (seg.dp != d) NIDX_HIP(hipMemset(seg.vectors.p, 0, seg.vectors.bytes));
        // hipMemcpyDefault: a packed matrix may already live in device memory (a shard produced on the GPU); rows with the
        // 4-byte trailer are the mmap'd vectors.bin and are read by the host above
        NIDX_HIP(hipMemcpy2D(seg.vectors.p, (size_t)seg.dp * 4, in.vectors, in.row_stride_bytes, packed, in.n_vectors,
                             hipMemcpyDefault));
        // a device-to-device copy returns before it has run, and `stream` (non-blocking) does not order itself behind the null
        // stream: the norms below must not read the rows before they have landed
        NIDX_HIP(hipStreamSynchronize(nullptr));
        NIDX_HIP(seg.norm2.alloc((size_t)((in.n_vectors + 7u) & ~7u) * 4));  // padded: the shared-row scan copies 8 norms per tile
        NIDX_HIP(launch_row_norms(seg.vectors.as<float>(), seg.n, seg.dp, seg.norm2.as<float>(), stream));
    }
    // alive bitset
    const uint32_t words = (in.n_paragraphs + 63) / 64;
    seg.alive_host.assign(words, ~0ull);
    seg.all_alive = true;
    if (in.alive_bitset) {
        memcpy(seg.alive_host.data(), in.alive_bitset, (size_t)words * 8);
        seg.all_alive = popcount_and(seg.alive_host.data(), nullptr, in.n_paragraphs) == in.n_paragraphs;
    }
    if (words && (in.n_paragraphs & 63)) seg.alive_host[words - 1] &= (1ull << (in.n_paragraphs & 63)) - 1ull;
    seg.alive_count = popcount_and(seg.alive_host.data(), nullptr, in.n_paragraphs);
    if (!seg.all_alive) {
        NIDX_HIP(seg.alive.alloc((size_t)words * 8));
        NIDX_HIP(hipMemcpy(seg.alive.p, seg.alive_host.data(), (size_t)words * 8, hipMemcpyHostToDevice));
    }
    if (in.paragraph_key_ids) {
        seg.key_ids.assign(in.paragraph_key_ids, in.paragraph_key_ids + in.n_paragraphs);
        if (in.n_paragraphs) {   // the device-side Fssc (fssc_device.hip) reads them too
            NIDX_HIP(seg.key_ids_dev.alloc((size_t)in.n_paragraphs * 8));
            NIDX_HIP(hipMemcpy(seg.key_ids_dev.p, seg.key_ids.data(), (size_t)in.n_paragraphs * 8, hipMemcpyHostToDevice));
        }
    }
    // vectors.quant
    seg.has_quant = false;
    if (in.quantized && in.n_vectors > 0) {
        if (cfg.similarity != NIDX_SIMILARITY_DOT || (d % 64u) != 0)
            return fail(NIDX_ERR_INVALID_CONFIGURATION, "a quantized store needs Dot similarity and dimension %% 64 == 0");
        const uint64_t need = (uint64_t)in.n_vectors * (d / 8 + 8);
        if (in.quantized_len != need)
            return fail(NIDX_ERR_INVALID_ARGUMENT, "vectors.quant holds %llu bytes, expected %llu", (unsigned long long)in.quantized_len,
                        (unsigned long long)need);
        NIDX_HIP(seg.quant.alloc(need));
        NIDX_HIP(hipMemcpy(seg.quant.p, in.quantized, need, hipMemcpyHostToDevice));
        seg.has_quant = true;
    }
    // graph
    seg.has_graph = false;
    if (in.hnsw_graph && in.hnsw_graph_len > 0 && in.n_vectors > 0) {
        const uint32_t g_nodes = in.hnsw_graph_nodes ? in.hnsw_graph_nodes : in.n_vectors;
        if (g_nodes > in.n_vectors) return fail(NIDX_ERR_INVALID_ARGUMENT, "hnsw_graph_nodes exceeds n_vectors");
        std::unique_ptr<HostGraph> hg(new HostGraph());
        std::string err;
        int rc = parse_disk_v2(in.hnsw_graph, in.hnsw_graph_len, g_nodes, *hg, err);
        if (rc != NIDX_OK) return fail(rc, "%s", err.c_str());
        if (g_nodes < in.n_vectors) {
            // the reusable part of a merge (segment.rs:143-153): kept on the host until extend_hnsw
            rc = attach_edge_weights(*hg, in.hnsw_graph, in.hnsw_graph_len, in.hnsw_edges, in.n_hnsw_edges, err);
            if (rc != NIDX_OK) return fail(rc, "%s", err.c_str());
            fix_broken_graph(*hg);
            seg.base_nodes = g_nodes;
            seg.base_graph = std::move(hg);
        } else {
            int32_t r = seg.upload_graph(*hg);
            if (r != NIDX_OK) return r;
        }
    }
    NIDX_HIP(hipStreamSynchronize(stream));
    return NIDX_OK;
}

// ---- exact fallback: closest_up_nodes with the pool and the visited set in HBM ---------------------------------------
int32_t VectorIndex::segment_spill_search(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, bool with_duplicates,
                                          int method, const uint64_t *d_filter, uint32_t *d_out_vec, float *d_out_score,
                                          uint32_t *d_out_count, const std::vector<uint32_t> &flagged, hipStream_t st) {
    VectorSegment &seg = segs[s];
    if (flagged.empty()) return NIDX_OK;
    HnswSpillArgs sp;
    sp.seg = seg.seg_dev(cfg.similarity);
    sp.g = seg.graph_dev();
    sp.queries = d_queries;
    sp.filter = d_filter;
    sp.k = k;
    sp.min_score = min_score;
    sp.with_duplicates = with_duplicates ? 1 : 0;
    sp.multi = cfg.vector_cardinality == NIDX_CARDINALITY_MULTI ? 1 : 0;
    sp.out_vec = d_out_vec;
    sp.out_score = d_out_score;
    sp.out_count = d_out_count;
    if (method == NIDX_METHOD_RABITQ_HNSW) {
        // the re-ranked entry points of the RaBitQ arm are still in the scratch of the call that flagged
        sp.entry_vec = scratch_entry_vec.as<uint32_t>();
        sp.entry_score = scratch_entry_score.as<float>();
        sp.entry_count = scratch_entry_count.as<uint32_t>();
        sp.entry_stride = k;
    } else {
        // descent + layer-0 search again (bounded by ef), stopping before the walk
        NIDX_HIP(scratch_dump_vec.reserve((size_t)nq * NIDX_DUMP_STRIDE * 4));
        NIDX_HIP(scratch_dump_score.reserve((size_t)nq * NIDX_DUMP_STRIDE * 4));
        NIDX_HIP(scratch_dump_count.reserve((size_t)nq * 4));
        HnswSearchArgs a;
        a.seg = sp.seg;
        a.g = sp.g;
        a.queries = d_queries;
        a.n_queries = nq;
        a.filter = d_filter;
        a.k = k;
        a.min_score = min_score;
        a.with_duplicates = sp.with_duplicates;
        a.vis_log2 = 15;
        a.out_vec = d_out_vec;
        a.out_score = d_out_score;
        a.out_count = d_out_count;
        a.stats = scratch_stats.as<uint32_t>();
        a.multi = sp.multi;
        a.eval_rows = rows_for(nq);
        a.min_waves = waves_for(nq);
        a.entry_vec = nullptr;
        a.entry_score = nullptr;
        a.entry_count = nullptr;
        a.dump_vec = scratch_dump_vec.as<uint32_t>();
        a.dump_score = scratch_dump_score.as<float>();
        a.dump_count = scratch_dump_count.as<uint32_t>();
        a.ef_search = ef_search;
        a.ef_upper = ef_upper;
        NIDX_HIP(launch_hnsw_search(a, waves_per_query, st));
        std::vector<uint32_t> stats((size_t)nq * NIDX_STAT_STRIDE);
        NIDX_HIP(hipMemcpyAsync(stats.data(), scratch_stats.p, stats.size() * 4, hipMemcpyDeviceToHost, st));
        NIDX_HIP(hipStreamSynchronize(st));
        for (uint32_t q : flagged)
            if (stats[(size_t)q * NIDX_STAT_STRIDE + NIDX_STAT_FLAGS])
                return fail(NIDX_ERR_INEXACT, "HNSW layer search overflowed the 2^15-entry visited table (query %u, k=%u)", q, k);
        sp.entry_vec = a.dump_vec;
        sp.entry_score = a.dump_score;
        sp.entry_count = a.dump_count;
        sp.entry_stride = NIDX_DUMP_STRIDE;
    }
    // a node enters the pool at most once: n slots (+ one chunk of slack) and n visited bits per query
    sp.pool_chunks = seg.n / 64 + 2;
    sp.vis_words = (seg.n + 31) / 32;
    const uint64_t per_query = (uint64_t)sp.pool_chunks * 64 * 8 + (uint64_t)sp.pool_chunks * 8 + (uint64_t)sp.vis_words * 4;
    const uint64_t budget = 2ull << 30;
    const size_t chunk = (size_t)std::max<uint64_t>(1, std::min<uint64_t>(flagged.size(), budget / per_query));
    NIDX_HIP(scratch_spill_pool.reserve(chunk * sp.pool_chunks * 64 * 8));
    NIDX_HIP(scratch_spill_cmax.reserve(chunk * sp.pool_chunks * 8));
    NIDX_HIP(scratch_spill_vis.reserve(chunk * sp.vis_words * 4));
    NIDX_HIP(scratch_spill_ids.reserve(flagged.size() * 4));
    NIDX_HIP(hipMemcpyAsync(scratch_spill_ids.p, flagged.data(), flagged.size() * 4, hipMemcpyHostToDevice, st));
    sp.pool = scratch_spill_pool.as<uint64_t>();
    sp.chunk_max = scratch_spill_cmax.as<uint64_t>();
    sp.vis = scratch_spill_vis.as<uint32_t>();
    for (size_t f0 = 0; f0 < flagged.size(); f0 += chunk) {
        const size_t n = std::min(chunk, flagged.size() - f0);
        NIDX_HIP(hipMemsetAsync(scratch_spill_cmax.p, 0, n * sp.pool_chunks * 8, st));
        NIDX_HIP(hipMemsetAsync(scratch_spill_vis.p, 0, n * sp.vis_words * 4, st));
        sp.query_ids = scratch_spill_ids.as<uint32_t>() + f0;
        sp.n_queries = (uint32_t)n;
        NIDX_HIP(launch_hnsw_closest_spill(sp, st));
    }
    NIDX_HIP(hipStreamSynchronize(st));  // `flagged` is read by the copy above
    spill_queries += flagged.size();
    return NIDX_OK;
}

// ---- one segment, device resident -------------------------------------------------------------------
int32_t VectorIndex::segment_search_device(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score,
                                           bool with_duplicates, int method, const uint64_t *d_filter,
                                           uint32_t *d_out_vec, float *d_out_score, uint32_t *d_out_count,
                                           uint32_t *d_stats, uint32_t vis_log2, hipStream_t st, uint32_t *d_flag_word) {
    VectorSegment &seg = segs[s];
    if (nq == 0) return NIDX_OK;
    if ((method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW) && !seg.has_graph)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u has no HNSW graph", s);
    if (method != NIDX_METHOD_HNSW) {
        // every other method stages through index-owned scratch: order this call behind the previous user's work
        int32_t rc = scratch_acquire(st);
        if (rc != NIDX_OK) return rc;
        rc = segment_search_device_scratch(s, d_queries, nq, k, min_score, with_duplicates, method, d_filter, d_out_vec, d_out_score,
                                           d_out_count, d_stats, vis_log2, st, d_flag_word);
        const int32_t rc2 = scratch_release(st);
        return rc != NIDX_OK ? rc : rc2;
    }
    return segment_search_device_scratch(s, d_queries, nq, k, min_score, with_duplicates, method, d_filter, d_out_vec, d_out_score,
                                         d_out_count, d_stats, vis_log2, st, d_flag_word);
}

int32_t VectorIndex::scratch_acquire(hipStream_t st) {
    if (scratch_event_recorded) NIDX_HIP(hipStreamWaitEvent(st, scratch_event, 0));
    return NIDX_OK;
}
int32_t VectorIndex::scratch_release(hipStream_t st) {
    if (!scratch_event) NIDX_HIP(hipEventCreateWithFlags(&scratch_event, hipEventDisableTiming));
    NIDX_HIP(hipEventRecord(scratch_event, st));
    scratch_event_recorded = true;
    return NIDX_OK;
}

// The argument record of one segment's plain HNSW search; shape_nq = the walks of the launch it joins (the launch shape — rows in
// flight, register budget — follows the grid, and every record of one table-driven launch carries the same one).
HnswSearchArgs VectorIndex::hnsw_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t shape_nq, uint32_t k, float min_score,
                                      bool with_duplicates, const uint64_t *d_filter, uint32_t *d_out_vec, float *d_out_score,
                                      uint32_t *d_out_count, uint32_t *d_stats, uint32_t vis_log2, uint32_t *d_flag_word) const {
    const VectorSegment &seg = segs[s];
    HnswSearchArgs a;
    a.seg = seg.seg_dev(cfg.similarity);
    a.g = seg.graph_dev();
    a.queries = d_queries;
    a.n_queries = nq;
    a.filter = d_filter;
    a.k = k;
    a.min_score = min_score;
    a.with_duplicates = with_duplicates ? 1 : 0;
    a.vis_log2 = vis_log2;
    a.out_vec = d_out_vec;
    a.out_score = d_out_score;
    a.out_count = d_out_count;
    a.stats = d_stats;
    a.multi = cfg.vector_cardinality == NIDX_CARDINALITY_MULTI ? 1 : 0;
    a.eval_rows = rows_for(shape_nq);
    a.min_waves = waves_for(shape_nq);
    a.entry_vec = nullptr;
    a.entry_score = nullptr;
    a.entry_count = nullptr;
    a.dump_vec = nullptr;
    a.dump_score = nullptr;
    a.dump_count = nullptr;
    a.flag_word = d_flag_word;
    a.ef_search = ef_search;
    a.ef_upper = ef_upper;
    a.closest_prefetch = closest_prefetch ? 1 : 0;
    return a;
}

// The argument record of one segment's register-tile exact scan (single-vector paragraphs); the caller sets `partial`.
ScanArgs VectorIndex::scan_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, const uint64_t *d_filter) const {
    const VectorSegment &seg = segs[s];
    ScanArgs a;
    a.vectors = seg.vectors.as<float>();
    a.norm2 = seg.norm2.as<float>();
    a.n = seg.n;
    a.dp = seg.dp;
    a.queries = d_queries;
    a.n_queries = nq;
    a.alive = seg.all_alive ? nullptr : seg.alive.as<uint64_t>();
    a.filter = d_filter;
    a.para_of_vec = seg.identity_para ? nullptr : seg.para_of_vec.as<uint32_t>();
    a.similarity = cfg.similarity;
    a.min_score = min_score;
    a.k = k;
    a.qt = 0;
    a.partial = nullptr;
    a.row_mask = nullptr;
    return a;
}
// true when segment s's exact scan of this batch takes the register-tile kernel (the shared-row scan needs large batches over
// mostly unfiltered rows: those are long launches of their own)
bool VectorIndex::scan_takes_tile_kernel(uint32_t s, uint32_t nq, uint32_t k, uint64_t matching) const {
    const VectorSegment &seg = segs[s];
    if (seg.vmax > 1) return false;   // multi-vector paragraphs go through para_best_kernel: the per-segment path
    if (const char *e = getenv("NIDX_GPU_SCAN_SHARED"))
        if (atoi(e) == 0) return true;
    return scan_shared_stripes(seg.n, nq, seg.dp, k, matching) == 0;
}

// The argument record of one segment's RaBitQ walk (the HNSW arm, hnsw/search.rs:333-366); the caller fills in the per-launch
// buffers: qd / planes (the encoded queries), visited (zeroed bitsets), out_* (the re-ranked entry points), stats.
RabitqSearchArgs VectorIndex::rabitq_hnsw_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, uint32_t *d_flag_word) const {
    const VectorSegment &seg = segs[s];
    RabitqSearchArgs r;
    r.seg = seg.seg_dev(cfg.similarity);
    r.g = seg.graph_dev();
    r.quant = seg.quant.as<uint8_t>();
    r.rec_len = seg.dim / 8 + 8;
    r.queries = d_queries;
    r.qd = nullptr;
    r.planes = nullptr;
    r.n_queries = nq;
    r.filter = nullptr;   // (the HNSW arm filters in closest_up_nodes)
    r.para_first = nullptr;
    r.para_num = nullptr;
    r.n_paragraphs = seg.n_paragraphs;
    r.k = k;
    r.ef = std::min<uint32_t>(k * 100u, 2000u);   // last_layer_k = min(k * RERANKING_FACTOR, RERANKING_LIMIT) (hnsw/search.rs:333-340)
    r.min_score = min_score;
    r.seen_log2 = rabitq_seen_log2();
    r.visited = nullptr;
    r.vis_words = (seg.n + 31u) / 32u;
    r.out_vec = nullptr;
    r.out_score = nullptr;
    r.out_count = nullptr;
    r.stats = nullptr;
    r.flag_word = d_flag_word;
    return r;
}

int32_t VectorIndex::segment_search_device_scratch(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score,
                                                   bool with_duplicates, int method, const uint64_t *d_filter,
                                                   uint32_t *d_out_vec, float *d_out_score, uint32_t *d_out_count,
                                                   uint32_t *d_stats, uint32_t vis_log2, hipStream_t st, uint32_t *d_flag_word) {
    VectorSegment &seg = segs[s];
    if (method == NIDX_METHOD_HNSW) {
        const HnswSearchArgs a = hnsw_args(s, d_queries, nq, nq, k, min_score, with_duplicates, d_filter, d_out_vec, d_out_score, d_out_count, d_stats,
                                           vis_log2, d_flag_word);
        NIDX_HIP(launch_hnsw_search(a, waves_per_query, st));
        return NIDX_OK;
    }
    if (method == NIDX_METHOD_RABITQ_HNSW || method == NIDX_METHOD_RABITQ_BRUTE_FORCE) {
        if (!seg.has_quant) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u has no quantized store", s);
        if (k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "the RaBitQ arms keep at most %d hits per query (got k=%u)", NIDX_K_MAX, k);
        const bool hnsw = method == NIDX_METHOD_RABITQ_HNSW;
        const uint32_t nw = seg.dim / 64u;
        NIDX_HIP(scratch_rq.reserve((size_t)nq * sizeof(RabitqQueryDev)));
        NIDX_HIP(scratch_planes.reserve((size_t)nq * 4 * nw * 8));
        NIDX_HIP(launch_rabitq_query(d_queries, nq, seg.dp, seg.dim, scratch_rq.as<RabitqQueryDev>(), scratch_planes.as<uint64_t>(), st));
        RabitqSearchArgs r;
        r.seg = seg.seg_dev(cfg.similarity);
        r.g = seg.graph_dev();
        r.quant = seg.quant.as<uint8_t>();
        r.rec_len = seg.dim / 8 + 8;
        r.queries = d_queries;
        r.qd = scratch_rq.as<RabitqQueryDev>();
        r.planes = scratch_planes.as<uint64_t>();
        r.filter = d_filter;
        r.para_first = seg.identity_para ? nullptr : seg.para_first.as<uint32_t>();
        r.para_num = seg.identity_para ? nullptr : seg.para_num.as<uint32_t>();
        r.n_paragraphs = seg.n_paragraphs;
        r.k = k;
        r.ef = 0;
        r.min_score = min_score;
        r.visited = nullptr;
        r.vis_words = 0;
        r.stats = d_stats;
        r.flag_word = d_flag_word;
        if (!hnsw) {
            r.n_queries = nq;
            r.out_vec = d_out_vec;
            r.out_score = d_out_score;
            r.out_count = d_out_count;
            NIDX_HIP(launch_rabitq_bf(r, st));
            return NIDX_OK;
        }
        // last_layer_k = min(k * RERANKING_FACTOR, RERANKING_LIMIT) (hnsw/search.rs:333-340)
        r.ef = std::min<uint32_t>(k * 100u, 2000u);
        r.vis_words = (seg.n + 31u) / 32u;
        NIDX_HIP(scratch_entry_vec.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_entry_score.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_entry_count.reserve((size_t)nq * 4));
        // the visited bitsets of one launch are capped at 1 GiB: larger batches go in slices
        const uint32_t slice = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nq, (1ull << 30) / ((uint64_t)r.vis_words * 4)));
        NIDX_HIP(scratch_vis.reserve((size_t)slice * r.vis_words * 4));
        r.visited = scratch_vis.as<uint32_t>();
        // evicted candidates that still tie with a walk's worst result, beyond the 64 it keeps in LDS (thousands of identical vectors)
        r.tie_stride = rabitq_tie_stride(r.ef);
        NIDX_HIP(scratch_ties.reserve((size_t)slice * r.tie_stride * 8));
        r.tie_spill = rabitq_tie_spill_enabled() ? scratch_ties.as<uint64_t>() : nullptr;
        for (uint32_t q0 = 0; q0 < nq; q0 += slice) {
            const uint32_t cnt = std::min(slice, nq - q0);
            NIDX_HIP(hipMemsetAsync(scratch_vis.p, 0, (size_t)cnt * r.vis_words * 4, st));
            RabitqSearchArgs rs = r;
            rs.n_queries = cnt;
            rs.queries = d_queries + (size_t)q0 * seg.dp;
            rs.qd = r.qd + q0;
            rs.planes = r.planes + (size_t)q0 * 4 * nw;
            rs.out_vec = scratch_entry_vec.as<uint32_t>() + (size_t)q0 * k;
            rs.out_score = scratch_entry_score.as<float>() + (size_t)q0 * k;
            rs.out_count = scratch_entry_count.as<uint32_t>() + q0;
            rs.stats = d_stats ? d_stats + (size_t)q0 * NIDX_STAT_STRIDE : nullptr;
            NIDX_HIP(launch_rabitq_hnsw(rs, st));
        }
        // closest_up_nodes from the re-ranked entry points, on the raw query (search.rs:369-375)
        HnswSearchArgs a;
        a.seg = seg.seg_dev(cfg.similarity);
        a.g = seg.graph_dev();
        a.queries = d_queries;
        a.n_queries = nq;
        a.filter = d_filter;
        a.k = k;
        a.min_score = min_score;
        a.with_duplicates = with_duplicates ? 1 : 0;
        a.vis_log2 = vis_log2;
        a.out_vec = d_out_vec;
        a.out_score = d_out_score;
        a.out_count = d_out_count;
        a.stats = d_stats;  // entry mode only ORs its overflow flags into the RaBitQ counters
        a.multi = cfg.vector_cardinality == NIDX_CARDINALITY_MULTI ? 1 : 0;
        a.eval_rows = rows_for(nq);
        a.min_waves = waves_for(nq);
        a.entry_vec = scratch_entry_vec.as<uint32_t>();
        a.entry_score = scratch_entry_score.as<float>();
        a.entry_count = scratch_entry_count.as<uint32_t>();
        a.dump_vec = nullptr;
        a.dump_score = nullptr;
        a.dump_count = nullptr;
        a.flag_word = d_flag_word;
        NIDX_HIP(launch_hnsw_search(a, waves_per_query, st));
        return NIDX_OK;
    }
    const bool multi = seg.vmax > 1;
    // multi-vector paragraphs on the matrix-core scans: like the plain scan below, the k best paragraphs are covered by the k * vmax
    // best vectors, which para_best_kernel reduces to one hit per paragraph (segment.rs:582-593)
    const uint32_t k_page = k;
    if (multi && (method == NIDX_METHOD_BRUTE_FORCE_MFMA || method == NIDX_METHOD_BRUTE_FORCE_BF16)) {
        const uint32_t cap = method == NIDX_METHOD_BRUTE_FORCE_MFMA ? NIDX_MFMA_KMAX : NIDX_BF16_CAND;
        if ((uint64_t)k * seg.vmax > cap)
            return fail(NIDX_ERR_UNSUPPORTED, "result_per_page %u x %u vectors per paragraph exceeds the %u hits this matrix-core scan keeps", k,
                        seg.vmax, cap);
        k = k * seg.vmax;
        NIDX_HIP(scratch_multi_vec.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_multi_score.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_multi_count.reserve((size_t)nq * 4));
    }
    // where the scan leaves its k best vectors: the caller's arrays, or the input of the per-paragraph reduction
    uint32_t *const v_out_vec = multi ? scratch_multi_vec.as<uint32_t>() : d_out_vec;
    float *const v_out_score = multi ? scratch_multi_score.as<float>() : d_out_score;
    uint32_t *const v_out_count = multi ? scratch_multi_count.as<uint32_t>() : d_out_count;
    auto reduce_paragraphs = [&]() -> int {
        if (!multi) return NIDX_OK;
        NIDX_HIP(launch_para_best(v_out_vec, v_out_score, v_out_count, nq, k, seg.para_of_vec.as<uint32_t>(), k_page, d_out_vec, d_out_score,
                                  d_out_count, st));
        return NIDX_OK;
    };
    if (method == NIDX_METHOD_BRUTE_FORCE_MFMA) {
        if (k > NIDX_MFMA_KMAX) return fail(NIDX_ERR_UNSUPPORTED, "the MFMA scan keeps at most %d hits per query (got k=%u)", NIDX_MFMA_KMAX, k);
        if (cfg.similarity == NIDX_SIMILARITY_COSINE && !seg.norm2_serial.p) {
            NIDX_HIP(seg.norm2_serial.alloc((size_t)seg.n * 4));
            NIDX_HIP(launch_serial_norms(seg.vectors.as<float>(), seg.n, seg.dp, seg.norm2_serial.as<float>(), st));
        }
        NIDX_HIP(scratch_qnorm.reserve((size_t)nq * 4));
        NIDX_HIP(launch_serial_norms(d_queries, nq, seg.dp, scratch_qnorm.as<float>(), st));
        const uint32_t stripes = mfma_scan_stripes(seg.n, nq);
        NIDX_HIP(scratch_partial.reserve((size_t)nq * stripes * k * 8));
        MfmaScanArgs m;
        m.vectors = seg.vectors.as<float>();
        m.norm2 = seg.norm2_serial.as<float>();
        m.n = seg.n;
        m.dp = seg.dp;
        m.queries = d_queries;
        m.q_norm2 = scratch_qnorm.as<float>();
        m.n_queries = nq;
        m.alive = seg.all_alive ? nullptr : seg.alive.as<uint64_t>();
        m.filter = d_filter;
        m.para_of_vec = seg.identity_para ? nullptr : seg.para_of_vec.as<uint32_t>();
        m.similarity = cfg.similarity;
        m.min_score = min_score;
        m.k = k;
        m.partial = scratch_partial.as<uint64_t>();
        NIDX_HIP(launch_mfma_scan(m, stripes, st));
        NIDX_HIP(launch_merge_topk(m.partial, nq, stripes, k, v_out_vec, v_out_score, v_out_count, st));
        return reduce_paragraphs();
    }
    if (method == NIDX_METHOD_BRUTE_FORCE_BF16) {
        if (k > NIDX_BF16_CAND) return fail(NIDX_ERR_UNSUPPORTED, "the bf16 fallback re-scores %d candidates per query (got k=%u)", NIDX_BF16_CAND, k);
        const bool cos = cfg.similarity == NIDX_SIMILARITY_COSINE;
        if (!seg.vectors16.p) {
            seg.dp16 = (seg.dp + 63u) & ~63u;
            NIDX_HIP(seg.vectors16.alloc((size_t)((seg.n + 255u) & ~255u) * seg.dp16 * 2));
            NIDX_HIP(launch_to_bf16_tiled(seg.vectors.as<float>(), cos ? seg.norm2.as<float>() : nullptr, seg.n, seg.dp, seg.dp16,
                                          seg.vectors16.as<unsigned short>(), st));
        }
        NIDX_HIP(scratch_qnorm.reserve((size_t)nq * 4));
        NIDX_HIP(launch_row_norms(d_queries, nq, seg.dp, scratch_qnorm.as<float>(), st));
        NIDX_HIP(scratch_q16.reserve((size_t)((nq + 255u) & ~255u) * seg.dp16 * 2));
        NIDX_HIP(launch_to_bf16_tiled(d_queries, cos ? scratch_qnorm.as<float>() : nullptr, nq, seg.dp, seg.dp16, scratch_q16.as<unsigned short>(), st));
        NIDX_HIP(scratch_rowmask.reserve((size_t)((seg.n + 255u) / 256u) * 32));
        NIDX_HIP(launch_bf16_row_mask(seg.n, seg.identity_para ? nullptr : seg.para_of_vec.as<uint32_t>(),
                                      seg.all_alive ? nullptr : seg.alive.as<uint64_t>(), d_filter, scratch_rowmask.as<uint64_t>(), st));
        const uint32_t stripes = bf16_scan_stripes(seg.n, nq);
        NIDX_HIP(scratch_partial.reserve((size_t)nq * stripes * NIDX_BF16_CAND * 8));
        NIDX_HIP(scratch_cand_vec.reserve((size_t)nq * NIDX_BF16_CAND * 4));
        NIDX_HIP(scratch_cand_score.reserve((size_t)nq * NIDX_BF16_CAND * 4));
        NIDX_HIP(scratch_cand_count.reserve((size_t)nq * 4));
        Bf16ScanArgs b;
        b.vectors16 = seg.vectors16.as<unsigned short>();
        b.n = seg.n;
        b.dp16 = seg.dp16;
        b.queries16 = scratch_q16.as<unsigned short>();
        b.n_queries = nq;
        b.row_mask = scratch_rowmask.as<uint64_t>();
        b.partial = scratch_partial.as<uint64_t>();
        b.floor_score = nullptr;
        {
            const char *dbg = getenv("NIDX_GPU_BF16_DEBUG");
            b.debug = dbg ? atoi(dbg) : 0;
        }
        // Sample pass: the first 1/16 of the rows (at most 256 tiles) are scanned on their own; the 32nd best score each query
        // reaches there is a floor for its final 32nd best, and the full pass starts every candidate list at that floor instead
        // of at -inf — it skips the record-breaking insertions every stripe would otherwise pay while its list warms up.
        const uint32_t n_tiles = (seg.n + 255u) / 256u;
        const uint32_t sample_tiles = std::min<uint32_t>(256u, n_tiles / 16u);
        uint64_t floor_rows = 0;   // rows the current floor was taken from
        if (sample_tiles >= 8 && !(b.debug & 4)) {
            Bf16ScanArgs sm = b;
            sm.n = std::min<uint32_t>(seg.n, sample_tiles * 256u);
            const uint32_t s_stripes = bf16_scan_stripes(sm.n, nq);
            NIDX_HIP(launch_bf16_scan(sm, s_stripes, st));
            NIDX_HIP(launch_merge_topk(sm.partial, nq, s_stripes, NIDX_BF16_CAND, scratch_cand_vec.as<uint32_t>(), scratch_cand_score.as<float>(),
                                       scratch_cand_count.as<uint32_t>(), st));
            NIDX_HIP(scratch_floor.reserve((size_t)nq * 4));
            NIDX_HIP(launch_bf16_floor(scratch_cand_score.as<float>(), scratch_cand_count.as<uint32_t>(), nq, nullptr, nullptr, scratch_floor.as<float>(), st));
            b.floor_score = scratch_floor.as<float>();
            floor_rows = sm.n;
        }
        // With a floor the scan needs no candidate lists: bf16_append_kernel (BK = 64 stages, 64 x 128 wave tiles) appends the rows that
        // reach it to the 32 slots of their (query, stripe).  A floor taken from R rows lets a pass over N rows expect 32 N / (R stripes)
        // candidates per slot group; passes are sized for 8, so a floor from the 64 k-row prefix is first tightened on a sample across the
        // whole corpus (up to R stripes / 4 rows: a 16th of it with 64 stripes), then the full pass runs over the rest.  A
        // group that fills up anyway (clustered rows, a floor the filter emptied) flags its query block, and the list kernel scans
        // that block again: the candidates are the list kernel's in every case.  NIDX_GPU_BF16_APPEND=0 keeps the list kernel alone.
        const char *const append_env = getenv("NIDX_GPU_BF16_APPEND");
        const bool append_enabled = !(append_env && atoi(append_env) == 0);
        int ablate = 0;
        if (const char *e = getenv("NIDX_GPU_BF16_ABLATE")) ablate = atoi(e);   // (read by experiment builds of the ring kernel only)
        if (b.floor_score && append_enabled && stripes >= 32 && !b.debug) {
            const uint32_t qb = (nq + 255u) / 256u;
            NIDX_HIP(scratch_bf16_flags.reserve((size_t)qb * 8));
            NIDX_HIP(scratch_floor2.reserve((size_t)nq * 4));
            uint32_t *const flags_sample = scratch_bf16_flags.as<uint32_t>(), *const flags_full = flags_sample + qb;
            float *floor_next = scratch_floor2.as<float>();
            const size_t partial_bytes = (size_t)nq * stripes * NIDX_BF16_CAND * 8;
            // A sample pass scans every rs-th ROUND of the full pass's own grid (round i of stripe x = tile x + i * stripes), so the full pass can
            // skip exactly those rounds and go on from the sample's slots: its rows are not scanned twice.
            NIDX_HIP(scratch_bf16_counts.reserve((size_t)nq * stripes * 4));
            uint32_t last_rs = 0;
            for (;;) {
                const uint64_t cap_rows = floor_rows * (uint64_t)stripes / 4u;   // rows a pass may cover under the current floor
                if (cap_rows + cap_rows / 4 >= seg.n) break;   // (up to 10 expected per group of 32 slots is still far from filling one)
                // the least the full pass needs is a floor from n / (stripes / 4) rows; take more only when the current floor cannot carry that
                const uint64_t want_rows = std::min<uint64_t>(cap_rows, std::max<uint64_t>((uint64_t)seg.n * 4u / stripes, floor_rows * 2u));
                const uint32_t rs = (uint32_t)((seg.n + want_rows - 1) / want_rows);
                const uint32_t rounds = (n_tiles + stripes - 1) / stripes;
                if (rs < 2 || rounds < 2 * rs) break;   // too few rounds to take a sample of: the full pass takes what comes
                const uint64_t rows_s = (uint64_t)((rounds + rs - 1) / rs) * stripes * 256u;   // (an upper bound)
                if (rows_s > cap_rows) break;
                NIDX_HIP(hipMemsetAsync(b.partial, 0, partial_bytes, st));
                NIDX_HIP(hipMemsetAsync(flags_sample, 0, (size_t)qb * 4, st));
                Bf16ScanArgs ap = b;
                ap.debug = ablate;
                ap.round_step = rs;
                ap.overflow = flags_sample;
                ap.cnt_inout = scratch_bf16_counts.as<uint32_t>();
                NIDX_HIP(launch_bf16_append(ap, stripes, st));
                NIDX_HIP(launch_merge_topk(ap.partial, nq, stripes, NIDX_BF16_CAND, scratch_cand_vec.as<uint32_t>(), scratch_cand_score.as<float>(),
                                           scratch_cand_count.as<uint32_t>(), st));
                NIDX_HIP(launch_bf16_floor(scratch_cand_score.as<float>(), scratch_cand_count.as<uint32_t>(), nq, b.floor_score, flags_sample, floor_next, st));
                float *const used = const_cast<float *>(b.floor_score);
                b.floor_score = floor_next;
                floor_next = used;
                floor_rows = (uint64_t)(rounds / rs) * stripes * 256u;   // (a lower bound)
                last_rs = rs;
            }
            if (!last_rs) NIDX_HIP(hipMemsetAsync(b.partial, 0, partial_bytes, st));   // (else the slots hold the last sample's candidates)
            NIDX_HIP(hipMemsetAsync(flags_full, 0, (size_t)qb * 4, st));
            Bf16ScanArgs ap = b;
            ap.debug = ablate;
            ap.overflow = flags_full;
            if (last_rs) {
                ap.round_skip = last_rs;
                ap.skip_unless = flags_sample;
                ap.cnt_inout = scratch_bf16_counts.as<uint32_t>();
            }
            NIDX_HIP(launch_bf16_append(ap, stripes, st));
            b.run_if = flags_full;
        }
        NIDX_HIP(launch_bf16_scan(b, stripes, st));
        NIDX_HIP(launch_merge_topk(b.partial, nq, stripes, NIDX_BF16_CAND, scratch_cand_vec.as<uint32_t>(),
                                   scratch_cand_score.as<float>(), scratch_cand_count.as<uint32_t>(), st));
        RescoreArgs r;
        r.vectors = seg.vectors.as<float>();
        r.queries = d_queries;
        r.dp = seg.dp;
        r.n_queries = nq;
        r.cand_vec = scratch_cand_vec.as<uint32_t>();
        r.cand_count = scratch_cand_count.as<uint32_t>();
        r.n_cand_max = NIDX_BF16_CAND;
        r.similarity = cfg.similarity;
        r.min_score = min_score;
        r.k = k;
        r.out_vec = v_out_vec;
        r.out_score = v_out_score;
        r.out_count = v_out_count;
        NIDX_HIP(launch_rescore_select(r, st));
        return reduce_paragraphs();
    }
    // brute force
    uint32_t nblk = scan_num_blocks(seg.n);
    uint32_t shared_stripes = 0;  // > 0: the shared-row scan (large batches) with that many row stripes
    // multi-vector paragraphs: the k best paragraphs are covered by the k * vmax best vectors (each better paragraph
    // contributes at most vmax of them); they are reduced to one hit per paragraph afterwards
    const uint32_t k_out = k;
    if (multi) {
        if ((uint64_t)k * seg.vmax > NIDX_K_MAX)
            return fail(NIDX_ERR_UNSUPPORTED, "result_per_page %u x %u vectors per paragraph exceeds the %d-hit scan", k, seg.vmax, NIDX_K_MAX);
        k = k * seg.vmax;
        NIDX_HIP(scratch_cand_vec.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_cand_score.reserve((size_t)nq * k * 4));
        NIDX_HIP(scratch_cand_count.reserve((size_t)nq * 4));
    }
    {
        // rows the filter lets through decide which scan streams less (the register-tile scan never loads a filtered row)
        uint64_t matching = seg.n;
        if (scan_matching_hint != ~0ull) matching = scan_matching_hint;
        else if (d_filter) matching = 0;          // unknown selectivity: the scan that skips filtered rows
        else if (!seg.all_alive) matching = seg.alive_count;
        shared_stripes = scan_shared_stripes(seg.n, nq, seg.dp, k, matching);
        if (const char *e = getenv("NIDX_GPU_SCAN_SHARED")) shared_stripes = atoi(e) ? shared_stripes : 0;
        if (shared_stripes) nblk = shared_stripes;
    }
    size_t need = (size_t)nq * nblk * k * 8;
    NIDX_HIP(scratch_partial.reserve(need));
    ScanArgs a;
    a.vectors = seg.vectors.as<float>();
    a.norm2 = seg.norm2.as<float>();
    a.n = seg.n;
    a.dp = seg.dp;
    a.queries = d_queries;
    a.n_queries = nq;
    a.alive = seg.all_alive ? nullptr : seg.alive.as<uint64_t>();
    a.filter = d_filter;
    a.para_of_vec = seg.identity_para ? nullptr : seg.para_of_vec.as<uint32_t>();
    a.similarity = cfg.similarity;
    a.min_score = min_score;
    a.k = k;
    a.qt = 0;
    a.partial = scratch_partial.as<uint64_t>();
    a.row_mask = nullptr;
    if (shared_stripes) {
        NIDX_HIP(scratch_rowmask.reserve((size_t)((seg.n + 255u) / 256u) * 32));
        NIDX_HIP(launch_bf16_row_mask(seg.n, a.para_of_vec, a.alive, a.filter, scratch_rowmask.as<uint64_t>(), st));
        a.row_mask = scratch_rowmask.as<uint64_t>();
        NIDX_HIP(launch_scan_shared(a, nblk, st));
    } else {
        NIDX_HIP(launch_scan(a, nblk, st));
    }
    if (!multi) {
        NIDX_HIP(launch_merge_topk(a.partial, nq, nblk, k, d_out_vec, d_out_score, d_out_count, st));
        return NIDX_OK;
    }
    NIDX_HIP(launch_merge_topk(a.partial, nq, nblk, k, scratch_cand_vec.as<uint32_t>(), scratch_cand_score.as<float>(),
                               scratch_cand_count.as<uint32_t>(), st));
    NIDX_HIP(launch_para_best(scratch_cand_vec.as<uint32_t>(), scratch_cand_score.as<float>(), scratch_cand_count.as<uint32_t>(), nq, k,
                              seg.para_of_vec.as<uint32_t>(), k_out, d_out_vec, d_out_score, d_out_count, st));
    return NIDX_OK;
}

// ---- one segment, exactly: launch -> (only if the flag word is set) larger visited table / HBM-resident walk ----------
namespace {
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// NIDX_GPU_TRACE_SLOW_US=N: a host search that takes longer than N microseconds prints where the time went (stderr)
}  // namespace
double trace_slow_us() {
    static const double v = [] { const char *e = getenv("NIDX_GPU_TRACE_SLOW_US"); return e ? atof(e) : 0.0; }();
    return v;
}
namespace {
thread_local double t_launch_us = 0, t_sync_us = 0;
thread_local bool t_crowded_launch = false;
}  // namespace
bool VectorIndex::crowded_launch() { return t_crowded_launch; }
void VectorIndex::set_crowded_launch(bool on) { t_crowded_launch = on; }

int32_t VectorIndex::segment_search_exact(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score,
                                          bool with_duplicates, int method, const uint64_t *d_filter, uint32_t *d_out_block,
                                          uint32_t *host_block, hipStream_t st, uint32_t *n_retried) {
    uint32_t *d_vec = d_out_block;
    float *d_score = reinterpret_cast<float *>(d_out_block + (size_t)nq * k);
    uint32_t *d_count = d_out_block + (size_t)nq * k * 2;
    uint32_t *d_flag = d_count + nq;
    const size_t block_bytes = out_block_words(nq, k) * 4;
    const bool walks = method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW;
    if (n_retried) *n_retried = 0;
    if (walks) NIDX_HIP(scratch_stats.reserve((size_t)nq * NIDX_STAT_STRIDE * 4));
    uint32_t vis_log2 = default_vis_log2;
    for (;;) {
        const double t0 = now_us();
        NIDX_HIP(hipMemsetAsync(d_flag, 0, 4, st));
        int32_t rc = segment_search_device(s, d_queries, nq, k, min_score, with_duplicates, method, d_filter, d_vec, d_score, d_count,
                                           walks ? scratch_stats.as<uint32_t>() : nullptr, vis_log2, st, d_flag);
        if (rc != NIDX_OK) return rc;
        NIDX_HIP(hipMemcpyAsync(host_block, d_out_block, block_bytes, hipMemcpyDeviceToHost, st));
        const double t1 = now_us();
        NIDX_HIP(hipStreamSynchronize(st));
        t_launch_us += t1 - t0;
        t_sync_us += now_us() - t1;
        const uint32_t flags = host_block[block_bytes / 4 - 1];
        if (!walks || flags == 0) return NIDX_OK;
        // a bounded on-chip structure overflowed for some query: find which (the per-query counters) and re-run
        std::vector<uint32_t> stats((size_t)nq * NIDX_STAT_STRIDE);
        NIDX_HIP(hipMemcpyAsync(stats.data(), scratch_stats.p, stats.size() * 4, hipMemcpyDeviceToHost, st));
        NIDX_HIP(hipStreamSynchronize(st));
        if ((flags & NIDX_FLAG_POOL_INEXACT) || vis_log2 >= 15) {
            // the walk of these queries outgrew the on-chip pool / visited table: re-run them with both in HBM
            std::vector<uint32_t> flagged;
            for (uint32_t q = 0; q < nq; q++)
                if (stats[(size_t)q * NIDX_STAT_STRIDE + NIDX_STAT_FLAGS]) flagged.push_back(q);
            if (n_retried) *n_retried = (uint32_t)flagged.size();
            rc = segment_spill_search(s, d_queries, nq, k, min_score, with_duplicates, method, d_filter, d_vec, d_score, d_count, flagged, st);
            if (rc != NIDX_OK) return rc;
            NIDX_HIP(hipMemcpyAsync(host_block, d_out_block, block_bytes, hipMemcpyDeviceToHost, st));
            NIDX_HIP(hipStreamSynchronize(st));
            return NIDX_OK;
        }
        if (n_retried) *n_retried = nq;
        vis_log2 = 15;  // visited table was too small: retry once with the largest one (128 KiB of LDS)
    }
}

// ---- Fssc (searcher.rs:149-199) -----------------------------------------------------------------------
namespace {
struct Cand {
    float score;
    uint32_t seg, vec, para;
    uint64_t key;
};
}  // namespace

int32_t VectorIndex::rows_equal_host(uint32_t sa, uint32_t va, uint32_t sb, uint32_t vb, bool &eq) {
    std::vector<float> a(segs[sa].dp), b(segs[sb].dp);
    NIDX_HIP(hipMemcpy(a.data(), segs[sa].vectors.as<float>() + (size_t)va * segs[sa].dp, (size_t)segs[sa].dp * 4,
                       hipMemcpyDeviceToHost));
    NIDX_HIP(hipMemcpy(b.data(), segs[sb].vectors.as<float>() + (size_t)vb * segs[sb].dp, (size_t)segs[sb].dp * 4,
                       hipMemcpyDeviceToHost));
    eq = memcmp(a.data(), b.data(), (size_t)cfg.dimension * 4) == 0;
    return NIDX_OK;
}

// ParagraphInvertedIndexes::filter on the device (filter.hip): postfix program over posting-list unions.
int32_t VectorIndex::eval_filter_program(uint32_t si, const nidx_gpu_filter_program_t &prog, uint64_t &matching) {
    VectorSegment &seg = segs[si];
    const uint32_t n_bits = seg.n_paragraphs, words = (n_bits + 63) / 64;
    matching = 0;
    // validate + stack depth
    int depth = 0, max_depth = 0;
    for (uint32_t i = 0; i < prog.n_ops; i++) {
        const nidx_gpu_filter_op_t &op = prog.ops[i];
        switch (op.op) {
            case NIDX_FILTER_PUSH_LISTS:
                if (op.a > op.b || op.b > prog.n_lists) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: list range out of bounds");
                if (op.b > op.a && !seg.f_n_lists) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u has no filter index", si);
                for (uint32_t l = op.a; l < op.b; l++)
                    if (prog.lists[l] >= seg.f_n_lists) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program: unknown posting list %u", prog.lists[l]);
                depth++;
                break;
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
    if (depth != 1) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter program must leave exactly one bitset (leaves %d)", depth);
    NIDX_HIP(scratch_fstack.reserve((size_t)max_depth * std::max<uint32_t>(words, 1) * 8));
    NIDX_HIP(scratch_filter.reserve((size_t)std::max<uint32_t>(words, 1) * 8));
    NIDX_HIP(scratch_fcount.reserve(8));
    if (prog.n_lists) {
        NIDX_HIP(scratch_flists.reserve((size_t)prog.n_lists * 4));
        NIDX_HIP(hipMemcpyAsync(scratch_flists.p, prog.lists, (size_t)prog.n_lists * 4, hipMemcpyHostToDevice, stream));
    }
    NIDX_HIP(hipMemsetAsync(scratch_fcount.p, 0, 8, stream));
    uint64_t *stack = scratch_fstack.as<uint64_t>();
    auto slot = [&](int d) { return stack + (size_t)d * words; };
    depth = 0;
    for (uint32_t i = 0; i < prog.n_ops; i++) {
        const nidx_gpu_filter_op_t &op = prog.ops[i];
        switch (op.op) {
            case NIDX_FILTER_PUSH_LISTS:
                NIDX_HIP(launch_bitset_fill(slot(depth), words, n_bits, 0, stream));
                NIDX_HIP(launch_bitset_scatter(seg.f_offsets.as<unsigned long long>(), seg.f_ids.as<uint32_t>(),
                                               scratch_flists.as<uint32_t>() + op.a, op.b - op.a, n_bits, slot(depth), stream));
                depth++;
                break;
            case NIDX_FILTER_PUSH_ALL: NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 1, stream)); break;
            case NIDX_FILTER_PUSH_NONE: NIDX_HIP(launch_bitset_fill(slot(depth++), words, n_bits, 0, stream)); break;
            case NIDX_FILTER_AND:
            case NIDX_FILTER_OR:
                NIDX_HIP(launch_bitset_binop(slot(depth - 2), slot(depth - 1), words, op.op == NIDX_FILTER_AND ? 0 : 1, stream));
                depth--;
                break;
            case NIDX_FILTER_NOT: NIDX_HIP(launch_bitset_not(slot(depth - 1), words, n_bits, stream)); break;
        }
    }
    NIDX_HIP(launch_bitset_and_count(slot(0), seg.all_alive ? nullptr : seg.alive.as<uint64_t>(), scratch_filter.as<uint64_t>(),
                                     words, scratch_fcount.as<unsigned long long>(), stream));
    unsigned long long c = 0;
    NIDX_HIP(hipMemcpyAsync(&c, scratch_fcount.p, 8, hipMemcpyDeviceToHost, stream));
    NIDX_HIP(hipStreamSynchronize(stream));
    matching = c;
    return NIDX_OK;
}

int32_t VectorIndex::quantize(uint32_t si) {
    std::lock_guard<std::mutex> lock(mu);
    NIDX_HIP(hipSetDevice(device));
    VectorSegment &seg = segs[si];
    // VectorConfig::quantizable_vectors (config.rs:170-173)
    if (cfg.similarity != NIDX_SIMILARITY_DOT || (cfg.dimension % 64u) != 0)
        return fail(NIDX_ERR_INVALID_CONFIGURATION, "vectors are quantizable only with Dot similarity and dimension %% 64 == 0");
    seg.has_quant = false;
    if (seg.n == 0) return NIDX_OK;
    NIDX_HIP(seg.quant.alloc((size_t)seg.n * (seg.dim / 8 + 8)));
    NIDX_HIP(launch_rabitq_encode(seg.vectors.as<float>(), seg.n, seg.dp, seg.dim, seg.quant.as<uint8_t>(), stream));
    NIDX_HIP(hipStreamSynchronize(stream));
    seg.has_quant = true;
    return NIDX_OK;
}

int32_t VectorIndex::reserve_search(uint32_t nq_max, uint32_t k) {
    std::lock_guard<std::mutex> lock(mu);
    if (nq_max <= reserved_nq && k <= reserved_k) return NIDX_OK;
    NIDX_HIP(hipSetDevice(device));
    nq_max = std::max(nq_max, reserved_nq);
    k = std::max(k, reserved_k);
    const uint32_t dp = (cfg.dimension + 3u) & ~3u;
    NIDX_HIP(pin_in.reserve((size_t)nq_max * dp * 4));
    NIDX_HIP(scratch_queries.reserve((size_t)nq_max * dp * 4));
    NIDX_HIP(scratch_out_block.reserve(out_block_words(nq_max, k) * 4));
    NIDX_HIP(pin_out.reserve(out_block_words(nq_max, k) * 4));
    NIDX_HIP(scratch_stats.reserve((size_t)nq_max * NIDX_STAT_STRIDE * 4));
    reserved_nq = nq_max;
    reserved_k = k;
    return NIDX_OK;
}

int32_t VectorIndex::search_host(const float *queries, uint32_t nq, const nidx_gpu_vector_search_params_t &p,
                                 const uint64_t *const *segment_filters, const nidx_gpu_filter_program_t *programs,
                                 uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                                 uint32_t *out_count, int32_t *out_method, uint64_t *out_matching) {
    const double t_enter = now_us();
    std::lock_guard<std::mutex> lock(mu);
    const double t_locked = now_us();
    t_launch_us = t_sync_us = 0;
    struct SlowTrace {
        double t_enter, t_locked, t_staged = 0, t_searched = 0;
        uint32_t nq;
        ~SlowTrace() {
            const double lim = trace_slow_us(), t_end = now_us();
            if (lim > 0 && t_end - t_enter > lim)
                fprintf(stderr, "[nidx_gpu slow search] nq=%u total=%.0f us: lock %.0f, stage %.0f, segments %.0f (launch calls %.0f, sync %.0f), fssc %.0f\n", nq,
                        t_end - t_enter, t_locked - t_enter, t_staged - t_locked, t_searched - t_staged, t_launch_us, t_sync_us, t_end - t_searched);
        }
    } trace{t_enter, t_locked, 0, 0, nq};
    NIDX_HIP(hipSetDevice(device));
    const uint32_t k = p.k;
    const uint32_t d = cfg.dimension;
    for (uint32_t q = 0; q < nq; q++) out_count[q] = 0;
    if (out_method)
        for (size_t s = 0; s < segs.size(); s++) out_method[s] = 0;
    if (nq == 0 || k == 0 || segs.empty()) return NIDX_OK;
    // nucliadb asks for max(top_k, rank-fusion window, reranker window) results per page (unit_retrieval.py), windows <= 500
    if (k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "result_per_page > %d is not supported (got %u)", NIDX_K_MAX, k);
    if (p.method < 0 || p.method > 6) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown search method %d", p.method);

    // query batch -> HBM (normalised first when the index says so, searcher.rs:246-252), staged in pinned memory: one
    // transfer in, and per segment one transfer out (hits + counts + the launch's flag word)
    const uint32_t dp = (d + 3u) & ~3u;
    NIDX_HIP(pin_in.reserve((size_t)nq * dp * 4));
    float *qpad = pin_in.as<float>();
    stage_query_rows(queries, qpad, nq, d, dp, cfg.normalize_vectors);
    NIDX_HIP(scratch_queries.reserve((size_t)nq * dp * 4));
    NIDX_HIP(hipMemcpyAsync(scratch_queries.p, qpad, (size_t)nq * dp * 4, hipMemcpyHostToDevice, stream));
    const size_t block_words = out_block_words(nq, k);
    NIDX_HIP(scratch_out_block.reserve(block_words * 4));
    NIDX_HIP(pin_out.reserve(block_words * 4));

    trace.t_staged = now_us();
    const size_t S = segs.size();
    std::vector<std::vector<uint32_t>> hv(S), hc(S);
    std::vector<std::vector<float>> hs(S);
    for (size_t s = 0; s < S; s++) {
        VectorSegment &seg = segs[s];
        const uint64_t *filt = segment_filters ? segment_filters[s] : nullptr;
        const bool has_prog = programs && programs[s].ops && programs[s].n_ops;
        // matching = |filter ∩ alive| (segment.rs:516-531)
        uint64_t matching = filt ? popcount_and(seg.alive_host.data(), filt, seg.n_paragraphs) : seg.alive_count;
        if (has_prog) {
            int32_t rc = eval_filter_program((uint32_t)s, programs[s], matching);
            if (rc != NIDX_OK) return rc;
        }
        if (out_matching) out_matching[s] = matching;
        hc[s].assign(nq, 0);
        if (matching == 0 || seg.n == 0) continue;
        int method = p.method;
        if (method == NIDX_METHOD_AUTO) {
            // OpenSegment::_search (segment.rs:506-513,535-555): RaBitQ whenever the store has quantized vectors
            const bool rabitq = rabitq_enabled(seg);   // (pages up to NIDX_K_MAX: the lists of the RaBitQ kernels are sized by k)
            const bool hnsw = seg.has_graph && use_hnsw(seg.n_paragraphs, matching, k, rabitq);
            method = rabitq ? (hnsw ? NIDX_METHOD_RABITQ_HNSW : NIDX_METHOD_RABITQ_BRUTE_FORCE)
                            : (hnsw ? NIDX_METHOD_HNSW : NIDX_METHOD_BRUTE_FORCE);
        }
        if ((method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW) && !seg.has_graph)
            return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %zu has no HNSW graph", s);
        if (out_method) out_method[s] = method;
        const uint64_t *d_filter = nullptr;
        if (has_prog) {
            d_filter = scratch_filter.as<uint64_t>();
        } else if (filt) {
            size_t bytes = (size_t)((seg.n_paragraphs + 63) / 64) * 8;
            NIDX_HIP(scratch_filter.reserve(bytes));
            NIDX_HIP(hipMemcpyAsync(scratch_filter.p, filt, bytes, hipMemcpyHostToDevice, stream));
            d_filter = scratch_filter.as<uint64_t>();
        }
        scan_matching_hint = matching;
        const int32_t rc = segment_search_exact((uint32_t)s, scratch_queries.as<float>(), nq, k, p.min_score, p.with_duplicates != 0, method,
                                                d_filter, scratch_out_block.as<uint32_t>(), pin_out.as<uint32_t>(), stream, nullptr);
        scan_matching_hint = ~0ull;
        if (rc != NIDX_OK) return rc;
        const uint32_t *blk = pin_out.as<uint32_t>();
        hv[s].assign(blk, blk + (size_t)nq * k);
        hs[s].resize((size_t)nq * k);
        memcpy(hs[s].data(), blk + (size_t)nq * k, (size_t)nq * k * 4);
        hc[s].assign(blk + (size_t)nq * k * 2, blk + (size_t)nq * k * 2 + nq);
    }

    trace.t_searched = now_us();
    std::vector<const uint32_t *> pv(S), pc(S);
    std::vector<const float *> ps(S);
    for (size_t s = 0; s < S; s++) {
        const bool any = !hv[s].empty();
        pv[s] = any ? hv[s].data() : nullptr;
        ps[s] = any ? hs[s].data() : nullptr;
        pc[s] = any ? hc[s].data() : nullptr;
    }
    return fssc_merge(nq, p, pv.data(), ps.data(), pc.data(), out_segment, out_paragraph, out_vector, out_score, out_count);
}

// ---- Fssc (searcher.rs:149-199): the k best across segments, one hit per paragraph key --------------------------------
// seg_vec / seg_score: [n_segments] pointers to [nq][k] rows (nullptr = the segment was not searched), seg_count: [nq].
int32_t VectorIndex::fssc_merge(uint32_t nq, const nidx_gpu_vector_search_params_t &p, const uint32_t *const *seg_vec,
                                const float *const *seg_score, const uint32_t *const *seg_count, uint32_t *out_segment,
                                uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count) {
    const uint32_t k = p.k;
    const size_t S = segs.size();
    // One segment whose paragraph identities are its addresses, a query whose scores fall strictly: every hit enters the buffer
    // (<= k candidates with distinct keys: nothing is evicted or rejected, equal vector bytes would mean equal scores) and the
    // final sort leaves the kernel's order — the general loop below would copy the row unchanged.
    const bool one_plain_segment = S == 1 && seg_count[0] && segs[0].key_ids.empty();
    std::vector<Cand> buff, offered;
    std::unordered_multimap<uint32_t, uint32_t> offered_by_score;   // score bits -> index into `offered`
    for (uint32_t q = 0; q < nq; q++) {
        if (one_plain_segment) {
            const uint32_t cnt = seg_count[0][q];
            const float *sc = seg_score[0] + (size_t)q * k;
            bool falling = cnt <= k;
            for (uint32_t i = 1; i < cnt && falling; i++) falling = sc[i - 1] > sc[i];
            if (cnt == 1) falling = sc[0] == sc[0];
            if (falling) {
                const uint32_t *vv = seg_vec[0] + (size_t)q * k;
                out_count[q] = cnt;
                for (uint32_t i = 0; i < cnt; i++) {
                    if (out_segment) out_segment[(size_t)q * k + i] = 0;
                    if (out_paragraph) out_paragraph[(size_t)q * k + i] = segs[0].para_host.empty() ? vv[i] : segs[0].para_host[vv[i]];
                    if (out_vector) out_vector[(size_t)q * k + i] = vv[i];
                    if (out_score) out_score[(size_t)q * k + i] = sc[i];
                }
                continue;
            }
        }
        buff.clear();
        offered.clear();
        offered_by_score.clear();
        for (size_t s = 0; s < S; s++) {
            if (!seg_count[s]) continue;
            for (uint32_t i = 0; i < seg_count[s][q]; i++) {
                Cand c;
                c.score = seg_score[s][(size_t)q * k + i];
                c.seg = (uint32_t)s;
                c.vec = seg_vec[s][(size_t)q * k + i];
                c.para = segs[s].para_host.empty() ? c.vec : segs[s].para_host[c.vec];
                c.key = segs[s].key_ids.empty() ? (((uint64_t)s << 32) | c.para) : segs[s].key_ids[c.para];
                if (!p.with_duplicates) {
                    // Fssc.seen: vector bytes already offered.  Equal bytes imply equal score bits for
                    // one query, so rows are only fetched back on a bit-identical score.
                    // (the candidates offered so far are found by their score bits through a hash: pages of 500 hits from 50
                    // segments are 25 000 candidates per query, and a scan of all of them per candidate is 3 x 10^8 compares)
                    bool dup = false;
                    uint32_t sbits;
                    memcpy(&sbits, &c.score, 4);
                    auto range = offered_by_score.equal_range(sbits);
                    for (auto it = range.first; it != range.second && !dup; ++it) {
                        const Cand &o = offered[it->second];
                        bool eq = false;
                        int32_t rc = rows_equal_host(o.seg, o.vec, c.seg, c.vec, eq);
                        if (rc != NIDX_OK) return rc;
                        if (eq) dup = true;
                    }
                    if (dup) continue;
                    offered_by_score.emplace(sbits, (uint32_t)offered.size());
                    offered.push_back(c);
                }
                if (buff.size() == k) {
                    int victim = -1;
                    for (size_t j = 0; j < buff.size(); j++)
                        if (c.score > buff[j].score && (victim < 0 || buff[j].score < buff[victim].score)) victim = (int)j;
                    if (victim < 0) continue;
                    buff[victim] = buff.back();
                    buff.pop_back();
                }
                bool present = false;
                for (const Cand &b : buff)
                    if (b.key == c.key) { present = true; break; }
                if (!present) buff.push_back(c);
            }
        }
        // sort desc by score (partial_cmp); the HashSet order of equal scores is unspecified -> (segment, vector)
        std::stable_sort(buff.begin(), buff.end(), [](const Cand &a, const Cand &b) {
            if (a.score > b.score) return true;
            if (a.score < b.score) return false;
            if (a.seg != b.seg) return a.seg < b.seg;
            return a.vec < b.vec;
        });
        out_count[q] = (uint32_t)buff.size();
        for (size_t i = 0; i < buff.size(); i++) {
            if (out_segment) out_segment[(size_t)q * k + i] = buff[i].seg;
            if (out_paragraph) out_paragraph[(size_t)q * k + i] = buff[i].para;
            if (out_vector) out_vector[(size_t)q * k + i] = buff[i].vec;
            if (out_score) out_score[(size_t)q * k + i] = buff[i].score;
        }
    }
    return NIDX_OK;
}

uint64_t VectorIndex::popcount_filter(uint32_t s, const uint64_t *filt) const {
    return popcount_and(segs[s].alive_host.data(), filt, segs[s].n_paragraphs);
}

}  // namespace nidx

// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace nidx;

extern "C" {

int32_t nidx_gpu_last_error(char *buf, size_t len) try {
    const size_t have = strlen(g_last_error);
    if (buf && len) {
        size_t n = have < len - 1 ? have : len - 1;
        memcpy(buf, g_last_error, n);
        buf[n] = 0;
    }
    return (int32_t)have;
} NIDX_ABI_CATCH

int32_t nidx_gpu_abi_version(void) { return NIDX_GPU_ABI_VERSION; }
int32_t nidx_gpu_build_features(void) { return nidx::rabitq_has_experiments() ? NIDX_GPU_FEATURE_RABITQ_EXPERIMENTS : 0; }

int32_t nidx_gpu_device_count(int32_t *count_out) try {
    if (!count_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count_out = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count_out = n;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_set_device(int32_t device) try {
    NIDX_HIP(hipSetDevice(device));
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_open(const nidx_gpu_vector_config_t *config, const nidx_gpu_vector_segment_t *segments,
                             uint32_t n_segments, nidx_gpu_vector_index_t **index_out) try {
    if (!config || !index_out || (n_segments && !segments)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *index_out = nullptr;
    if (config->dimension == 0) return fail(NIDX_ERR_INVALID_CONFIGURATION, "Invalid configuration: Vector dimension cannot be 0");
    if (config->similarity != NIDX_SIMILARITY_DOT && config->similarity != NIDX_SIMILARITY_COSINE)
        return fail(NIDX_ERR_INVALID_CONFIGURATION, "Invalid configuration: unknown similarity %d", config->similarity);
    if (config->vector_cardinality != NIDX_CARDINALITY_SINGLE && config->vector_cardinality != NIDX_CARDINALITY_MULTI)
        return fail(NIDX_ERR_INVALID_CONFIGURATION, "unknown vector cardinality %d", config->vector_cardinality);
    // (the kernels keep a query as dimension / 256 16-byte pieces per lane: up to 16 of them)
    if (config->dimension > 4096) return fail(NIDX_ERR_UNSUPPORTED, "dimension > 4096 is not supported (got %u)", config->dimension);
    std::unique_ptr<VectorIndex> idx(new VectorIndex());
    idx->cfg = *config;
    // tuning knobs (not part of the ABI): workgroup shape and visited-table size of the HNSW kernels
    if (const char *e = getenv("NIDX_GPU_WAVES_PER_QUERY")) idx->waves_per_query = std::max(1, std::min(4, atoi(e)));
    if (const char *e = getenv("NIDX_GPU_EVAL_ROWS")) { idx->eval_rows = std::max(2, std::min(4, atoi(e))); idx->shape_pinned = true; }
    if (const char *e = getenv("NIDX_GPU_MIN_WAVES")) { idx->min_waves = atoi(e) >= 4 ? std::min(6, atoi(e)) : 2; idx->shape_pinned = true; }
    if (const char *e = getenv("NIDX_GPU_VIS_LOG2")) { idx->default_vis_log2 = (uint32_t)std::max(10, std::min(15, atoi(e))); idx->vis_pinned = true; }
    if (const char *e = getenv("NIDX_GPU_BUILD_VIS_LOG2")) {
        idx->build_vis_log2 = (uint32_t)std::max(10, std::min(15, atoi(e)));
        idx->build_vis_pinned = true;
    }
    NIDX_HIP(hipGetDevice(&idx->device));
    NIDX_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
    NIDX_HIP(idx->flag_word.alloc(64));
    NIDX_HIP(hipMemset(idx->flag_word.p, 0, 64));
    NIDX_HIP(idx->pin_flag.reserve(64));
    idx->segs.resize(n_segments);
    for (uint32_t s = 0; s < n_segments; s++) {
        int32_t rc = open_segment(*config, segments[s], idx->segs[s], idx->stream);
        if (rc != NIDX_OK) return rc;
    }
    *index_out = reinterpret_cast<nidx_gpu_vector_index_t *>(idx.release());
    return NIDX_OK;
} NIDX_ABI_CATCH

void nidx_gpu_vector_close(nidx_gpu_vector_index_t *index) {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    delete idx;
}

int32_t nidx_gpu_vector_set_tunable(nidx_gpu_vector_index_t *index, const char *name, int32_t value) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !name) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    std::string n(name);
    if (n == "waves_per_query") idx->waves_per_query = std::max(1, std::min(4, (int)value));
    else if (n == "eval_rows") { idx->eval_rows = std::max(2, std::min(4, (int)value)); idx->shape_pinned = true; }
    else if (n == "min_waves") { idx->min_waves = value >= 4 ? std::min(6, (int)value) : 2; idx->shape_pinned = true; }
    else if (n == "launch_shape") idx->launch_shape = (int)std::max<int64_t>(0, std::min<int64_t>(2, (int64_t)value));
    else if (n == "vis_log2") { idx->default_vis_log2 = (uint32_t)std::max(10, std::min(15, (int)value)); idx->vis_pinned = true; }
    else if (n == "coalesce_window_us") idx->coalescer_config(value, -1, -1);
    else if (n == "coalesce_max_batch") idx->coalescer_config(-1, value, -1);
    else if (n == "coalesce_in_flight") idx->coalescer_config(-1, -1, value);
    else if (n == "coalesce_max_callers") idx->coalescer_admission(std::max(0, (int)value), -1);   // 0 = unbounded
    else if (n == "coalesce_reject_when_full") idx->coalescer_admission(-1, value != 0);
    else if (n == "pipeline_depth") idx->pipeline_config(value);
    else if (n == "pipeline_walks") idx->pipeline_config(-1, value);   // batches whose search launches may run on the device at once (default 3); tickets beyond it upload ahead
    else if (n == "stage_threads") set_stage_threads(value);   // helper threads that share the copy of host query rows into pinned staging (process-wide; 0 = the caller alone)
    else if (n == "closest_prefetch") idx->closest_prefetch = value != 0;   // measurement knob of closest_up_nodes' edge prefetch: no result depends on it
    else if (n == "serial_segments") idx->serial_segments = value != 0;   // nidx_gpu_vector_search: one launch + transfer + wait per segment, Fssc on the host
    else if (n == "build_vis_log2") { idx->build_vis_log2 = (uint32_t)std::max(10, std::min(15, (int)value)); idx->build_vis_pinned = true; }
    else if (n == "ef_search") {   // 0 = the reference's EF_SEARCH (30)
        if (value < 0 || value > NIDX_K_MAX) return fail(NIDX_ERR_INVALID_ARGUMENT, "ef_search must be in 0..%d", NIDX_K_MAX);
        idx->ef_search = (uint32_t)value;
    }
    else if (n == "build_ef_upper") {   // 0 = the reference's greedy descent while inserting
        if (value < 0 || value > 64) return fail(NIDX_ERR_INVALID_ARGUMENT, "build_ef_upper must be in 0..64");
        idx->build_ef_upper = (uint32_t)value;
    }
    else if (n == "ef_upper") {   // 0 = the reference's greedy descent (one result per upper layer)
        if (value < 0 || value > 64) return fail(NIDX_ERR_INVALID_ARGUMENT, "ef_upper must be in 0..64");
        idx->ef_upper = (uint32_t)value;
    }
    else return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown tunable %s", name);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_space_usage(const nidx_gpu_vector_index_t *index, uint64_t *bytes_out) try {
    const VectorIndex *idx = reinterpret_cast<const VectorIndex *>(index);
    if (!idx || !bytes_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    uint64_t b = 0;
    for (const VectorSegment &s : idx->segs) b += s.bytes();
    *bytes_out = b;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_num_segments(const nidx_gpu_vector_index_t *index, uint32_t *n_out) try {
    const VectorIndex *idx = reinterpret_cast<const VectorIndex *>(index);
    if (!idx || !n_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *n_out = (uint32_t)idx->segs.size();
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_segment_records(const nidx_gpu_vector_index_t *index, uint32_t segment, uint32_t *n_out) try {
    const VectorIndex *idx = reinterpret_cast<const VectorIndex *>(index);
    if (!idx || !n_out || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad segment");
    *n_out = idx->segs[segment].n_paragraphs;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_quantize(nidx_gpu_vector_index_t *index, uint32_t segment) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad segment");
    return idx->quantize(segment);
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_serialize_quantized(nidx_gpu_vector_index_t *index, uint32_t segment, uint8_t *out, uint64_t out_cap,
                                            uint64_t *len_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size() || !len_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad segment");
    std::lock_guard<std::mutex> lock(idx->mu);
    VectorSegment &seg = idx->segs[segment];
    if (!seg.has_quant) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u has no quantized store", segment);
    *len_out = seg.quant.bytes;
    if (!out) return NIDX_OK;
    if (out_cap < seg.quant.bytes) return fail(NIDX_ERR_INVALID_ARGUMENT, "output buffer too small");
    NIDX_HIP(hipSetDevice(idx->device));
    NIDX_HIP(hipMemcpy(out, seg.quant.p, seg.quant.bytes, hipMemcpyDeviceToHost));
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_search(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                               const nidx_gpu_vector_search_params_t *params, const uint64_t *const *segment_filters,
                               uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                               uint32_t *out_count, int32_t *out_method) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || !out_count || (n_queries && !queries)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (idx->segs.size() > 1 && !out_method && !idx->serial_segments && n_queries && params->k && params->k <= NIDX_K_MAX && params->method >= 0 &&
        params->method <= 6) {
        // Searcher::_search over several segments (searcher.rs:270-287): every segment in one pass of the device and Fssc there too
        // (serving.cpp), instead of a launch, a transfer and a wait per segment
        uint64_t ticket = 0;
        const int32_t rc = idx->pipeline_submit(queries, n_queries, *params, segment_filters, true, &ticket);
        if (rc != NIDX_OK) return rc;
        return idx->pipeline_wait(ticket, out_segment, out_paragraph, out_vector, out_score, out_count, nullptr);
    }
    return idx->search_host(queries, n_queries, *params, segment_filters, nullptr, out_segment, out_paragraph, out_vector,
                            out_score, out_count, out_method, nullptr);
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_search_dim(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                                   uint32_t query_dimension, const nidx_gpu_vector_search_params_t *params,
                                   const uint64_t *const *segment_filters, uint32_t *out_segment,
                                   uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count,
                                   int32_t *out_method) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL index");
    if (query_dimension != idx->cfg.dimension)
        return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "Inconsistent dimensions. Index=%u Vector=%u", idx->cfg.dimension,
                    query_dimension);
    return nidx_gpu_vector_search(index, queries, n_queries, params, segment_filters, out_segment, out_paragraph,
                                  out_vector, out_score, out_count, out_method);
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_set_filter_index(nidx_gpu_vector_index_t *index, uint32_t segment, const nidx_gpu_filter_index_t *lists) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !lists || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    if (lists->n_lists && !lists->list_offsets) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL list_offsets");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    VectorSegment &seg = idx->segs[segment];
    const uint64_t n_ids = lists->n_lists ? lists->list_offsets[lists->n_lists] : 0;
    if (n_ids && !lists->paragraph_ids) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL paragraph_ids");
    for (uint64_t i = 0; i < n_ids; i++)
        if (lists->paragraph_ids[i] >= seg.n_paragraphs) return fail(NIDX_ERR_INVALID_ARGUMENT, "posting %llu out of range", (unsigned long long)i);
    NIDX_HIP(seg.f_offsets.alloc((size_t)(lists->n_lists + 1) * 8));
    NIDX_HIP(seg.f_ids.alloc(std::max<size_t>(n_ids, 1) * 4));
    if (lists->n_lists) NIDX_HIP(hipMemcpy(seg.f_offsets.p, lists->list_offsets, (size_t)(lists->n_lists + 1) * 8, hipMemcpyHostToDevice));
    if (n_ids) NIDX_HIP(hipMemcpy(seg.f_ids.p, lists->paragraph_ids, n_ids * 4, hipMemcpyHostToDevice));
    seg.f_n_lists = lists->n_lists;
    seg.f_n_ids = n_ids;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_set_filter_keys(nidx_gpu_vector_index_t *index, uint32_t segment, const uint8_t *key_bytes, const uint64_t *key_offsets,
                                        uint32_t n_keys) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size() || (n_keys && (!key_offsets || (key_offsets[n_keys] && !key_bytes))))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment/keys");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    VectorSegment &seg = idx->segs[segment];
    if (n_keys != seg.f_n_lists) return fail(NIDX_ERR_INVALID_ARGUMENT, "%u keys for %u posting lists", n_keys, seg.f_n_lists);
    // sorted, strictly: the lookups are binary searches
    for (uint32_t j = 1; j < n_keys; j++) {
        const uint64_t la = key_offsets[j] - key_offsets[j - 1], lb = key_offsets[j + 1] - key_offsets[j];
        const int c = memcmp(key_bytes + key_offsets[j - 1], key_bytes + key_offsets[j], (size_t)std::min(la, lb));
        if (c > 0 || (c == 0 && la >= lb)) return fail(NIDX_ERR_INVALID_ARGUMENT, "filter keys are not sorted (key %u)", j);
    }
    const uint64_t n_bytes = n_keys ? key_offsets[n_keys] : 0;
    NIDX_HIP(seg.f_key_bytes.alloc(std::max<uint64_t>(n_bytes, 1)));
    NIDX_HIP(seg.f_key_offsets.alloc((size_t)(n_keys + 1) * 8));
    if (n_bytes) NIDX_HIP(hipMemcpy(seg.f_key_bytes.p, key_bytes, n_bytes, hipMemcpyHostToDevice));
    if (n_keys) NIDX_HIP(hipMemcpy(seg.f_key_offsets.p, key_offsets, (size_t)(n_keys + 1) * 8, hipMemcpyHostToDevice));
    else NIDX_HIP(hipMemset(seg.f_key_offsets.p, 0, 8));
    seg.f_n_keys = n_keys;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_lookup_filter_keys(nidx_gpu_vector_index_t *index, uint32_t segment, const uint8_t *query_bytes,
                                           const uint64_t *query_offsets, const uint8_t *query_is_prefix, uint32_t n_queries,
                                           uint32_t *out_first, uint32_t *out_last) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    if (n_queries == 0) return NIDX_OK;
    if (!query_offsets || !query_is_prefix || !out_first || !out_last || (query_offsets[n_queries] && !query_bytes))
        return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    VectorSegment &seg = idx->segs[segment];
    if (seg.f_n_lists && !seg.f_key_offsets.p) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u has no filter key table", segment);
    const uint64_t qb = query_offsets[n_queries];
    const size_t off_bytes = (size_t)(n_queries + 1) * 8, total = off_bytes + n_queries + (size_t)qb;
    NIDX_HIP(idx->pin_in.reserve(total));
    unsigned char *h = idx->pin_in.as<unsigned char>();
    memcpy(h, query_offsets, off_bytes);
    memcpy(h + off_bytes, query_is_prefix, n_queries);
    if (qb) memcpy(h + off_bytes + n_queries, query_bytes, (size_t)qb);
    NIDX_HIP(idx->scratch_flists.reserve(total + 8));
    NIDX_HIP(idx->scratch_out_block.reserve((size_t)n_queries * 8));
    NIDX_HIP(idx->pin_out.reserve((size_t)n_queries * 8));
    NIDX_HIP(hipMemcpyAsync(idx->scratch_flists.p, h, total, hipMemcpyHostToDevice, idx->stream));
    const unsigned char *d = idx->scratch_flists.as<unsigned char>();
    uint32_t *d_first = idx->scratch_out_block.as<uint32_t>(), *d_last = d_first + n_queries;
    NIDX_HIP(launch_key_range(seg.f_key_bytes.as<uint8_t>(), seg.f_key_offsets.as<unsigned long long>(), seg.f_n_keys, d + off_bytes + n_queries,
                              reinterpret_cast<const unsigned long long *>(d), d + off_bytes, n_queries, d_first, d_last, idx->stream));
    NIDX_HIP(hipMemcpyAsync(idx->pin_out.p, d_first, (size_t)n_queries * 8, hipMemcpyDeviceToHost, idx->stream));
    NIDX_HIP(hipStreamSynchronize(idx->stream));
    memcpy(out_first, idx->pin_out.p, (size_t)n_queries * 4);
    memcpy(out_last, idx->pin_out.as<unsigned char>() + (size_t)n_queries * 4, (size_t)n_queries * 4);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_search_maxsim(nidx_gpu_vector_index_t *index, const float *queries, const uint64_t *qoff, uint32_t nq,
                                      const nidx_gpu_vector_search_params_t *params, const uint64_t *const *segment_filters,
                                      uint32_t *out_segment, uint32_t *out_paragraph, float *out_score, uint32_t *out_count) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || !out_count || !qoff || (nq && !queries)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint32_t k = params->k, d = idx->cfg.dimension;
    for (uint32_t q = 0; q < nq; q++) out_count[q] = 0;
    if (nq == 0 || k == 0) return NIDX_OK;
    const uint64_t T = qoff[nq];
    if (T == 0) return NIDX_OK;
    // first pass: every query vector on its own (searcher.rs:352-372)
    nidx_gpu_vector_search_params_t p1 = *params;
    p1.k = std::max<uint32_t>(k, 10);
    p1.min_score = -3.40282347e38f;  // f32::MIN
    p1.with_duplicates = 1;
    const uint32_t k1 = p1.k;
    std::vector<uint32_t> s1((size_t)T * k1), pa1((size_t)T * k1), c1(T);
    int32_t rc = idx->search_host(queries, (uint32_t)T, p1, segment_filters, nullptr, s1.data(), pa1.data(), nullptr, nullptr, c1.data(),
                                  nullptr, nullptr);
    if (rc != NIDX_OK) return rc;
    std::lock_guard<std::mutex> lock(idx->mu);
    NIDX_HIP(hipSetDevice(idx->device));
    // candidates: the paragraphs found by any of the query's vectors, once per ADDRESS (searcher.rs:375-377)
    struct Cand { uint32_t q, seg, para; };
    std::vector<Cand> cands;
    for (uint32_t q = 0; q < nq; q++) {
        std::vector<std::pair<uint32_t, uint32_t>> found;  // (paragraph address, segment)
        for (uint64_t t = qoff[q]; t < qoff[q + 1]; t++)
            for (uint32_t i = 0; i < c1[t]; i++) found.emplace_back(pa1[t * k1 + i], s1[t * k1 + i]);
        std::sort(found.begin(), found.end());
        for (size_t i = 0; i < found.size(); i++)
            if (i == 0 || found[i].first != found[i - 1].first) cands.push_back(Cand{q, found[i].second, found[i].first});
    }
    // raw query vectors on the device (maxsim uses the vectors as given, searcher.rs:346-350,384)
    const uint32_t dp = (d + 3u) & ~3u;
    std::vector<float> qpad((size_t)T * dp, 0.f);
    for (uint64_t t = 0; t < T; t++) memcpy(&qpad[t * dp], queries + t * d, (size_t)d * 4);
    NIDX_HIP(idx->scratch_queries.reserve(qpad.size() * 4));
    NIDX_HIP(hipMemcpyAsync(idx->scratch_queries.p, qpad.data(), qpad.size() * 4, hipMemcpyHostToDevice, idx->stream));
    std::vector<float> score(cands.size(), 0.f);
    for (size_t sgi = 0; sgi < idx->segs.size(); sgi++) {
        VectorSegment &seg = idx->segs[sgi];
        std::vector<uint32_t> meta;  // qfirst, qnum, first, num per candidate of this segment
        std::vector<size_t> where;
        std::vector<uint32_t> first, num;
        if (!seg.identity_para) {
            first.assign(seg.n_paragraphs, 0);
            num.assign(seg.n_paragraphs, 0);
            for (uint32_t v = 0; v < seg.n; v++) {
                if (num[seg.para_host[v]] == 0) first[seg.para_host[v]] = v;
                num[seg.para_host[v]]++;
            }
        }
        for (size_t i = 0; i < cands.size(); i++)
            if (cands[i].seg == sgi) where.push_back(i);
        if (where.empty()) continue;
        const size_t n = where.size();
        meta.resize(4 * n);
        for (size_t j = 0; j < n; j++) {
            const Cand &c = cands[where[j]];
            meta[j] = (uint32_t)qoff[c.q];
            meta[n + j] = (uint32_t)(qoff[c.q + 1] - qoff[c.q]);
            meta[2 * n + j] = seg.identity_para ? c.para : first[c.para];
            meta[3 * n + j] = seg.identity_para ? 1u : num[c.para];
        }
        NIDX_HIP(idx->scratch_cand_vec.reserve(meta.size() * 4));
        NIDX_HIP(idx->scratch_cand_score.reserve(n * 4));
        NIDX_HIP(hipMemcpyAsync(idx->scratch_cand_vec.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, idx->stream));
        const uint32_t *dm = idx->scratch_cand_vec.as<uint32_t>();
        NIDX_HIP(launch_maxsim(seg.vectors.as<float>(), seg.norm2.as<float>(), seg.dp, idx->cfg.similarity, idx->scratch_queries.as<float>(), dm,
                               dm + n, dm + 2 * n, dm + 3 * n, (uint32_t)n, idx->scratch_cand_score.as<float>(), idx->stream));
        std::vector<float> sc(n);
        NIDX_HIP(hipMemcpyAsync(sc.data(), idx->scratch_cand_score.p, n * 4, hipMemcpyDeviceToHost, idx->stream));
        NIDX_HIP(hipStreamSynchronize(idx->stream));
        for (size_t j = 0; j < n; j++) score[where[j]] = sc[j];
    }
    // `sp.score() > min_score`, sort by score desc, truncate (searcher.rs:381-391)
    size_t i = 0;
    for (uint32_t q = 0; q < nq; q++) {
        std::vector<size_t> mine;
        for (; i < cands.size() && cands[i].q == q; i++)
            if (score[i] > params->min_score) mine.push_back(i);
        std::sort(mine.begin(), mine.end(), [&](size_t a, size_t b) {
            if (score[a] != score[b]) return score[a] > score[b];
            if (cands[a].seg != cands[b].seg) return cands[a].seg < cands[b].seg;
            return cands[a].para < cands[b].para;
        });
        const uint32_t n = (uint32_t)std::min<size_t>(mine.size(), k);
        out_count[q] = n;
        for (uint32_t j = 0; j < n; j++) {
            if (out_segment) out_segment[(size_t)q * k + j] = cands[mine[j]].seg;
            if (out_paragraph) out_paragraph[(size_t)q * k + j] = cands[mine[j]].para;
            if (out_score) out_score[(size_t)q * k + j] = score[mine[j]];
        }
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_search_filtered(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                                        uint32_t query_dimension, const nidx_gpu_vector_search_params_t *params,
                                        const nidx_gpu_filter_program_t *segment_programs, uint32_t *out_segment,
                                        uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count,
                                        int32_t *out_method, uint64_t *out_matching) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || !out_count || (n_queries && !queries)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (query_dimension != idx->cfg.dimension)
        return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "Inconsistent dimensions. Index=%u Vector=%u", idx->cfg.dimension,
                    query_dimension);
    return idx->search_host(queries, n_queries, *params, nullptr, segment_programs, out_segment, out_paragraph, out_vector,
                            out_score, out_count, out_method, out_matching);
} NIDX_ABI_CATCH

static int32_t device_entry_method(VectorIndex *idx, uint32_t segment, const nidx_gpu_vector_search_params_t *params,
                                   const uint64_t *d_filter, int &method) {
    if (params->k == 0 || params->k > NIDX_K_MAX) return fail(NIDX_ERR_UNSUPPORTED, "k must be in 1..%d (got %u)", NIDX_K_MAX, params->k);
    if (idx->cfg.dimension & 3u)
        return fail(NIDX_ERR_UNSUPPORTED, "device-resident queries need a dimension that is a multiple of 4");
    if (params->method < 0 || params->method > 6) return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown search method %d", params->method);
    VectorSegment &seg = idx->segs[segment];
    method = params->method;
    if (method == NIDX_METHOD_AUTO) {
        if (d_filter) return fail(NIDX_ERR_INVALID_ARGUMENT, "NIDX_METHOD_AUTO with a device filter: pick the method explicitly");
        // OpenSegment::_search (segment.rs:506-513,535-555), like search_host
        const bool rabitq = idx->rabitq_enabled(seg);
        const bool hnsw = seg.has_graph && use_hnsw(seg.n_paragraphs, seg.alive_count, params->k, rabitq);
        method = rabitq ? (hnsw ? NIDX_METHOD_RABITQ_HNSW : NIDX_METHOD_RABITQ_BRUTE_FORCE)
                        : (hnsw ? NIDX_METHOD_HNSW : NIDX_METHOD_BRUTE_FORCE);
    }
    if ((method == NIDX_METHOD_HNSW || method == NIDX_METHOD_RABITQ_HNSW) && !seg.has_graph)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "segment has no HNSW graph");
    return NIDX_OK;
}

int32_t nidx_gpu_vector_segment_search_device(nidx_gpu_vector_index_t *index, uint32_t segment, const float *d_queries,
                                              uint32_t n_queries, const nidx_gpu_vector_search_params_t *params,
                                              const uint64_t *d_filter, uint32_t *d_out_vector, float *d_out_score,
                                              uint32_t *d_out_count, uint32_t *d_stats, void *stream) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    std::lock_guard<std::mutex> lock(idx->mu);
    int method = 0;
    int32_t rc = device_entry_method(idx, segment, params, d_filter, method);
    if (rc != NIDX_OK) return rc;
    return idx->segment_search_device(segment, d_queries, n_queries, params->k, params->min_score,
                                      params->with_duplicates != 0, method, d_filter, d_out_vector, d_out_score,
                                      d_out_count, d_stats, idx->vis_for(n_queries), (hipStream_t)stream, idx->flag_word.as<uint32_t>());
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_device_flags(nidx_gpu_vector_index_t *index, void *stream, uint32_t *flags_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !flags_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    hipStream_t st = (hipStream_t)stream;
    NIDX_HIP(hipMemcpyAsync(idx->pin_flag.p, idx->flag_word.p, 4, hipMemcpyDeviceToHost, st));
    NIDX_HIP(hipMemsetAsync(idx->flag_word.p, 0, 4, st));
    NIDX_HIP(hipStreamSynchronize(st));
    *flags_out = *idx->pin_flag.as<uint32_t>();
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_segment_search_device_exact(nidx_gpu_vector_index_t *index, uint32_t segment, const float *d_queries,
                                                    uint32_t n_queries, const nidx_gpu_vector_search_params_t *params,
                                                    const uint64_t *d_filter, uint32_t *d_out_block, uint32_t *host_out_block,
                                                    void *stream, uint32_t *n_retried_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !params || segment >= idx->segs.size() || !d_out_block) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    if (n_queries == 0) return NIDX_OK;
    std::lock_guard<std::mutex> lock(idx->mu);
    int method = 0;
    int32_t rc = device_entry_method(idx, segment, params, d_filter, method);
    if (rc != NIDX_OK) return rc;
    uint32_t *host = host_out_block;
    if (!host) {
        NIDX_HIP(idx->pin_out.reserve(VectorIndex::out_block_words(n_queries, params->k) * 4));
        host = idx->pin_out.as<uint32_t>();
    }
    return idx->segment_search_exact(segment, d_queries, n_queries, params->k, params->min_score, params->with_duplicates != 0, method,
                                     d_filter, d_out_block, host, (hipStream_t)stream, n_retried_out);
} NIDX_ABI_CATCH

int32_t nidx_gpu_use_hnsw(uint64_t total_nodes, uint64_t matching_nodes, uint64_t top_k, int32_t has_rabitq) try {
    return use_hnsw(total_nodes, matching_nodes, top_k, has_rabitq != 0) ? 1 : 0;
} NIDX_ABI_CATCH

int32_t nidx_gpu_similarity(const float *x, const float *y, uint32_t n_pairs, uint32_t dimension, int32_t similarity,
                            int32_t order, float *out) try {
    if (!x || !y || !out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (order != NIDX_ORDER_WAVE64) return fail(NIDX_ERR_UNSUPPORTED, "only NIDX_ORDER_WAVE64 is implemented here");
    if (n_pairs == 0) return NIDX_OK;
    const uint32_t dp = (dimension + 3u) & ~3u;
    DevBuf dx, dy, dout;
    NIDX_HIP(dx.alloc((size_t)n_pairs * dp * 4));
    NIDX_HIP(dy.alloc((size_t)n_pairs * dp * 4));
    NIDX_HIP(dout.alloc((size_t)n_pairs * 4));
    if (dp != dimension) {
        NIDX_HIP(hipMemset(dx.p, 0, dx.bytes));
        NIDX_HIP(hipMemset(dy.p, 0, dy.bytes));
    }
    NIDX_HIP(hipMemcpy2D(dx.p, (size_t)dp * 4, x, (size_t)dimension * 4, (size_t)dimension * 4, n_pairs, hipMemcpyHostToDevice));
    NIDX_HIP(hipMemcpy2D(dy.p, (size_t)dp * 4, y, (size_t)dimension * 4, (size_t)dimension * 4, n_pairs, hipMemcpyHostToDevice));
    NIDX_HIP(launch_pair_similarity(dx.as<float>(), dy.as<float>(), n_pairs, dp, similarity, dout.as<float>(), nullptr));
    NIDX_HIP(hipMemcpy(out, dout.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost));
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_normalize(const float *in, uint32_t n, uint32_t dimension, float *out) try {
    if (!in || !out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t i = 0; i < n; i++) normalize_row(in + (size_t)i * dimension, out + (size_t)i * dimension, dimension);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_vector_serialize_hnsw(const nidx_gpu_vector_index_t *index, uint32_t segment, uint8_t *graph_out,
                                       uint64_t graph_cap, uint64_t *graph_len_out, float *edges_out, uint64_t edges_cap,
                                       uint64_t *n_edges_out) try {
    const VectorIndex *idx = reinterpret_cast<const VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    const VectorSegment &seg = idx->segs[segment];
    if (!seg.has_graph) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment has no HNSW graph");
    HostGraph hg;
    hg.n = seg.n;
    hg.ep_node = seg.ep_node;
    hg.ep_layer = seg.ep_layer;
    hg.top_layer = seg.top_layer;
    hg.l0.resize(seg.g_l0.bytes / 4);
    hg.upper_base.resize(seg.g_upper_base.bytes / 4);
    NIDX_HIP(hipSetDevice(idx->device));
    NIDX_HIP(hipMemcpy(hg.l0.data(), seg.g_l0.p, seg.g_l0.bytes, hipMemcpyDeviceToHost));
    NIDX_HIP(hipMemcpy(hg.upper_base.data(), seg.g_upper_base.p, seg.g_upper_base.bytes, hipMemcpyDeviceToHost));
    uint32_t n_upper = 0;
    for (uint32_t i = 0; i < seg.n; i++) n_upper += seg.top_layer[i];
    hg.upper.resize((size_t)n_upper * NIDX_UP_STRIDE);
    if (n_upper) NIDX_HIP(hipMemcpy(hg.upper.data(), seg.g_upper.p, hg.upper.size() * 4, hipMemcpyDeviceToHost));
    if (seg.g_l0_w.p) {
        hg.l0_w.resize(hg.l0.size());
        NIDX_HIP(hipMemcpy(hg.l0_w.data(), seg.g_l0_w.p, hg.l0_w.size() * 4, hipMemcpyDeviceToHost));
        hg.upper_w.resize(hg.upper.size());
        if (n_upper) NIDX_HIP(hipMemcpy(hg.upper_w.data(), seg.g_upper_w.p, hg.upper_w.size() * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint8_t> graph;
    std::vector<float> edges;
    serialize_disk_v2(hg, graph, edges);
    if (graph_len_out) *graph_len_out = graph.size();
    if (n_edges_out) *n_edges_out = edges.size();
    if (graph_out) {
        if (graph_cap < graph.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "graph buffer too small");
        memcpy(graph_out, graph.data(), graph.size());
    }
    if (edges_out) {
        if (edges_cap < edges.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "edges buffer too small");
        memcpy(edges_out, edges.data(), edges.size() * 4);
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

}  // extern "C"
