/*
 * nidx_gpu.h — C ABI of the MI355X-native nidx search hot path (libnidx_gpu.so).
 *
 * This is the drop-in boundary: the plain-C surface a Rust `extern "C"` shim
 * (or ctypes / cgo) binds in place of the L3→L2 Rust trait calls of the reference.  Each entry
 * point cites the reference interface it replaces (paths relative to /root/reference/nidx/).
 *
 * Conventions
 *  - every function returns int32: 0 = NIDX_OK, <0 = error code; the message of the last error
 *    on the calling thread is read with nidx_gpu_last_error().  No exceptions / panics cross
 *    the ABI.
 *  - handles are opaque, immutable after open, and safe to use from many host threads
 *    (the reference's searchers are `Sync + Send` Arc's in an LRU cache,
 *    src/searcher/index_cache.rs:164-177); calls on one handle are serialised on one HIP stream.
 *  - "host" pointers are ordinary process memory, "device" pointers are HBM addresses on the
 *    device the handle was opened on (hipMalloc / a torch tensor's data_ptr()).
 *  - outputs are caller-owned buffers.
 *  - no torch / C++ types in any signature.
 */
#ifndef NIDX_GPU_H
#define NIDX_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: mirror nidx_vector::VectorErr (nidx_vector/src/lib.rs:202-234) ---- */
#define NIDX_OK 0
#define NIDX_ERR_IO (-1)
#define NIDX_ERR_INCONSISTENT_DIMENSIONS (-2) /* VectorErr::InconsistentDimensions (searcher.rs:255-262) */
#define NIDX_ERR_INVALID_CONFIGURATION (-3)   /* VectorErr::InvalidConfiguration (config.rs:190-198) */
#define NIDX_ERR_EMPTY_MERGE (-4)             /* VectorErr::EmptyMerge */
#define NIDX_ERR_INVALID_ARGUMENT (-5)
#define NIDX_ERR_UNSUPPORTED (-6)
#define NIDX_ERR_DEVICE (-7)                  /* HIP runtime error, or no gfx950 device present */
#define NIDX_ERR_INVALID_GRAPH (-8)           /* malformed hnsw.graph image */
#define NIDX_ERR_INEXACT (-9)                 /* a bounded on-chip pool overflowed: result would differ from the reference */
#define NIDX_ERR_OUT_OF_MEMORY (-10)          /* a host allocation failed inside the library */
#define NIDX_ERR_INTERNAL (-11)               /* any other C++ exception, caught at the boundary */
#define NIDX_ERR_BUSY (-12)                   /* nidx_gpu_vector_search_submit: every pipeline slot holds an unwaited ticket */

/* Copies the calling thread's last error message (NUL terminated) and returns its length. */
int32_t nidx_gpu_last_error(char *buf, size_t len);
/* ABI version of this header; bumped on any change of a signature OR of a struct the caller fills (a trailing field counts: the
 * library reads it).  A binding compares nidx_gpu_abi_version() with the NIDX_GPU_ABI_VERSION it was compiled against when it loads
 * the library and refuses a mismatch (nucliadb_amd/_lib.py does; INTEGRATION.md shows the Rust shim's check).
 * 5: nidx_gpu_bm25_search_options_t.phrase_slops, nested-query leaves of any kind (round 4); up to NIDX_GPU_BM25_MAX_TICKETS BM25
 *    tickets, several submitting threads, nidx_gpu_vector_open takes D > 3072 (round 5).
 * 6: nidx_gpu_build_features; nidx_gpu_vector_build_stats fills ten words (round 6). */
#define NIDX_GPU_ABI_VERSION 6
int32_t nidx_gpu_abi_version(void);
/* What this build of the library contains beyond the product paths: NIDX_GPU_FEATURE_RABITQ_EXPERIMENTS = the two-wave RaBitQ walk
 * and the unrolled instances of the plain one (`make EXPERIMENTS=1`; measurement material, selected by NIDX_GPU_RABITQ_WAVES=2 /
 * NIDX_GPU_RABITQ_PIPE=0 — without the feature the former is ignored and the latter runs the generic instance). */
#define NIDX_GPU_FEATURE_RABITQ_EXPERIMENTS 1
int32_t nidx_gpu_build_features(void);
/* nidx_gpu_bm25_search_submit: tickets that may be outstanding per index before it returns NIDX_ERR_BUSY */
#define NIDX_GPU_BM25_MAX_TICKETS 16
int32_t nidx_gpu_device_count(int32_t *count_out);
/* Selects the HIP device used by handles opened afterwards on this thread (one process per GPU). */
int32_t nidx_gpu_set_device(int32_t device);

/* =====================================================================================
 * Vector index — replaces nidx_vector::VectorSearcher / VectorIndexer
 * (nidx_vector/src/lib.rs:65-148) and everything under them on the hot path.
 * ===================================================================================== */

enum { NIDX_SIMILARITY_DOT = 0, NIDX_SIMILARITY_COSINE = 1 }; /* config.rs:32-37 — the only two */
enum { NIDX_CARDINALITY_SINGLE = 0, NIDX_CARDINALITY_MULTI = 1 };

/* Summation order of the f32 inner products (see DESIGN.md "numerics").  The reference's own
 * order is whatever SimSIMD dispatches to on the host CPU; ours is fixed per kernel family. */
enum {
    NIDX_ORDER_WAVE64 = 3,     /* 64 lanes x float4, fmaf chain, xor butterfly (scan + HNSW kernels) */
    NIDX_ORDER_SERIAL_FMA = 1  /* one k-ordered fmaf chain (f32 MFMA kernels) */
};

/* VectorConfig (nidx_vector/src/config.rs:102-124) — the fields the hot path reads. */
typedef struct {
    uint32_t dimension;          /* VectorType::DenseF32 { dimension } */
    int32_t similarity;          /* NIDX_SIMILARITY_* */
    int32_t normalize_vectors;   /* normalise the QUERY at search time (searcher.rs:246-252) */
    int32_t vector_cardinality;  /* NIDX_CARDINALITY_*; MULTI: a paragraph may own several (contiguous) vectors — one hit per
                                  * paragraph, its best vector (segment.rs:582-593, hnsw/search.rs:159-164) */
    uint32_t flags;              /* NIDX_CONFIG_* (VectorConfig::flags, config.rs:25-30) */
} nidx_gpu_vector_config_t;
#define NIDX_CONFIG_DISABLE_RABITQ_SEARCH 1u /* flags::DISABLE_RABITQ_SEARCH (config.rs:29) */

/* One open segment as the reference has it after segment::open + apply_deletions
 * (nidx_vector/src/segment.rs:39-90, 428-445): the caller passes the mmap'd files as they lie. */
typedef struct {
    /* vectors.bin (data_store/v2/vector_store.rs:30-40,131-147): n_vectors rows, each
     * `dimension` f32 LE followed — when row_stride_bytes == 4*dimension+4 — by the u32
     * paragraph address.  A packed [n][dimension] f32 matrix is row_stride_bytes == 4*dimension; a packed matrix
     * may also be a DEVICE pointer (the copy into the index is then device to device). */
    const void *vectors;
    uint64_t row_stride_bytes;
    uint32_t n_vectors;
    /* paragraph address of each vector; NULL => read from the row trailer, or identity when packed */
    const uint32_t *paragraph_of_vector;
    uint32_t n_paragraphs;
    /* hnsw.graph image (hnsw/disk/v2.rs:16-49); NULL/0 => no graph (brute force only) until
     * nidx_gpu_vector_build_hnsw is called */
    const uint8_t *hnsw_graph;
    uint64_t hnsw_graph_len;
    /* merge with graph reuse (segment.rs:137-167): the image may cover only the first hnsw_graph_nodes
     * vectors (0 = all); such a segment is searchable after nidx_gpu_vector_extend_hnsw inserted the rest.
     * hnsw_edges = the weights of hnsw.edges in graph order (needed by later prunes; NULL = all 0). */
    uint32_t hnsw_graph_nodes;
    const float *hnsw_edges;
    uint64_t n_hnsw_edges;
    /* alive bitset over paragraph addresses after apply_deletions (segment.rs:81,428-445);
     * bit i = word[i>>6] >> (i&63).  NULL => all alive */
    const uint64_t *alive_bitset;
    /* 64-bit identity of each paragraph key (equal key string <=> equal id) used by the
     * cross-segment merge Fssc (searcher.rs:67-96,175-198); NULL => ids unique per segment */
    const uint64_t *paragraph_key_ids;
    /* vectors.quant (data_store/v2/quant_vector_store.rs:29-64): n_vectors RaBitQ records of
     * dimension/8 + 8 bytes (rabitq.rs:38-106).  NULL => the segment has no quantized store
     * (has_quantized() == false) until nidx_gpu_vector_quantize is called.  When present — and unless the
     * config carries NIDX_CONFIG_DISABLE_RABITQ_SEARCH — NIDX_METHOD_AUTO takes the reference's RaBitQ
     * branches (segment.rs:506-513). */
    const uint8_t *quantized;
    uint64_t quantized_len;
} nidx_gpu_vector_segment_t;

typedef struct nidx_gpu_vector_index nidx_gpu_vector_index_t;

/* VectorSearcher::open (lib.rs:126-130): uploads every segment to HBM.  Segments are searched in
 * array order, like Searcher::_search's sequential loop (searcher.rs:270-287). */
int32_t nidx_gpu_vector_open(const nidx_gpu_vector_config_t *config, const nidx_gpu_vector_segment_t *segments,
                             uint32_t n_segments, nidx_gpu_vector_index_t **index_out);
void nidx_gpu_vector_close(nidx_gpu_vector_index_t *index);
/* VectorSearcher::space_usage (lib.rs:141-143): bytes of HBM held by the handle. */
int32_t nidx_gpu_vector_space_usage(const nidx_gpu_vector_index_t *index, uint64_t *bytes_out);
int32_t nidx_gpu_vector_num_segments(const nidx_gpu_vector_index_t *index, uint32_t *n_out);
int32_t nidx_gpu_vector_segment_records(const nidx_gpu_vector_index_t *index, uint32_t segment, uint32_t *n_out);

/* Launch-shape knobs of the HNSW kernels (no effect on results): "waves_per_query" 1..4,
 * "eval_rows" 2..4, "min_waves" 2|4, "vis_log2" 10..15, "build_vis_log2" 10..15, and the request coalescer's
 * "coalesce_window_us" / "coalesce_max_batch" / "coalesce_in_flight", the serving pipeline's "pipeline_depth" and "stage_threads" (0..16,
 * default 3, process-wide; NIDX_GPU_STAGE_THREADS: helper threads that share the copy of a batch's host query rows into pinned staging
 * with the submitting thread — 3 MiB per 1 024 x 768 batch, which one thread alone copies no faster than the device answers; helpers that
 * have started stay for the life of the process, so lowering the value only stops new ones from being started; a forked child starts its own).  The
 * kernel knobs are also read at open from the environment as NIDX_GPU_<NAME>.  "launch_shape": 0 (default) = a batch of more than 256
 * queries submitted through the pipeline while other batches are still on the device takes the shape that holds five walks per CU instead
 * of four (<= 96 VGPRs, 2^12-slot visited table: each walk a little slower, more of them resident), a batch that finds the device idle the
 * faster walk; 1 = every large batch takes the crowded shape (for a caller that overlaps batches on streams of its own through
 * nidx_gpu_vector_segment_search_device, which cannot see them); 2 = none does.  Setting "min_waves" / "eval_rows" / "vis_log2" pins a shape.
 * One knob DOES change results: "ef_search" (0 = the reference's constant EF_SEARCH = 30, hnsw/params.rs:46; up to 512): the
 * layer-0 search keeps max(k, ef_search) candidates.  The reference reaches its recall at 10 M vectors by searching 50 segments
 * of <= 200 k records each at ef = 30 (searcher.rs:270-287); a flat graph over the same vectors matches that recall at a larger
 * ef — bench.py reports both.  The RaBitQ arm keeps its own ef = min(100 k, 2000) (hnsw/search.rs:333-340). */
int32_t nidx_gpu_vector_set_tunable(nidx_gpu_vector_index_t *index, const char *name, int32_t value);

/* NIDX_METHOD_BRUTE_FORCE_MFMA: the same exact scan as a dense GEMM on the f32 matrix cores (one
 * pass over the corpus per batch, k <= 64; pages of k <= 16 keep two workgroups per CU).  It sums in
 * NIDX_ORDER_SERIAL_FMA, so its scores differ from the other methods' (NIDX_ORDER_WAVE64) in the last bits: never chosen
 * by AUTO.  On a multi-vector segment both matrix-core scans keep k x (most vectors of one paragraph) vectors — that product
 * is what the 64 / 32 bound applies to — and return each paragraph's best vector (segment.rs:582-593).
 * NIDX_METHOD_BRUTE_FORCE_BF16: the batched fallback on the bf16 matrix cores (k <= 32): candidates are
 * ranked with bf16 operands, the 32 best per query are re-scored from the f32 rows in
 * NIDX_ORDER_WAVE64 — returned scores are exact, the id set is the exact top-k up to bf16 ranking
 * error (recall measured in DESIGN.md).  Explicit only.
 * NIDX_METHOD_RABITQ_HNSW / NIDX_METHOD_RABITQ_BRUTE_FORCE: the two RaBitQ arms (hnsw/search.rs:333-366,
 * segment.rs:602-611): estimates from the 1-bit codes, error-bounded re-rank with the raw vectors.  AUTO
 * picks them, like the reference, whenever the segment has a quantized store. */
enum { NIDX_METHOD_AUTO = 0, NIDX_METHOD_HNSW = 1, NIDX_METHOD_BRUTE_FORCE = 2, NIDX_METHOD_BRUTE_FORCE_MFMA = 3,
       NIDX_METHOD_BRUTE_FORCE_BF16 = 4, NIDX_METHOD_RABITQ_HNSW = 5, NIDX_METHOD_RABITQ_BRUTE_FORCE = 6 };

/* The request fields the hot path reads (nidx_vector/src/request_types.rs:19-35). */
typedef struct {
    uint32_t k;               /* result_per_page (no_results) */
    float min_score;
    int32_t with_duplicates;  /* proto default false => dedupe by vector bytes */
    int32_t method;           /* NIDX_METHOD_AUTO = the use_hnsw cost model (segment.rs:626-660) */
} nidx_gpu_vector_search_params_t;

/* VectorSearcher::search for a BATCH of queries (the reference takes one query per call,
 * searcher.rs:241-290; a batch is n independent calls).  Host buffers, synchronous.
 *   queries           [n_queries][dimension] f32
 *   segment_filters   NULL, or n_segments pointers (each NULL or a bitset over that segment's
 *                     paragraph addresses) = inverted_indexes.filter(formula), already ANDed
 *                     with nothing: the alive bitset is applied here (segment.rs:516-529)
 *   out_segment/out_paragraph/out_vector/out_score   [n_queries][k] (k <= 512: nucliadb's result_per_page = max(top_k, rank-fusion window, reranker window) <= 500;
 *                     every arm, the RaBitQ ones included); rows hold out_count[q] hits,
 *                     score descending (Fssc -> Vec, searcher.rs:156-161)
 *   out_method        NULL or [n_segments]: NIDX_METHOD_* chosen per segment (same for every query)
 * Errors: NIDX_ERR_INCONSISTENT_DIMENSIONS is raised by nidx_gpu_vector_search_dim. */
int32_t nidx_gpu_vector_search(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                               const nidx_gpu_vector_search_params_t *params,
                               const uint64_t *const *segment_filters, uint32_t *out_segment,
                               uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                               uint32_t *out_count, int32_t *out_method);
/* Same, checking the query length like Searcher::_search (searcher.rs:255-262). */
int32_t nidx_gpu_vector_search_dim(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                                   uint32_t query_dimension, const nidx_gpu_vector_search_params_t *params,
                                   const uint64_t *const *segment_filters, uint32_t *out_segment,
                                   uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                                   uint32_t *out_count, int32_t *out_method);

/* ---- filter formulas evaluated on the device -------------------------------------------------------
 * Replaces ParagraphInvertedIndexes::filter (inverted_index/paragraph.rs:124-184).  A segment's label and
 * field-key posting lists (what index.map holds behind label.fst / field.fst) are uploaded once; a
 * request's Formula arrives as a postfix program whose atoms name posting lists by id (the host keeps the
 * string -> list-id lookup the FSTs do). */
typedef struct {
    uint32_t n_lists;
    const uint64_t *list_offsets; /* [n_lists+1] into paragraph_ids */
    const uint32_t *paragraph_ids; /* paragraph addresses */
} nidx_gpu_filter_index_t;
int32_t nidx_gpu_vector_set_filter_index(nidx_gpu_vector_index_t *index, uint32_t segment,
                                         const nidx_gpu_filter_index_t *lists);

/* The string -> posting-list lookups label.fst / field.fst serve (inverted_index/fst_index.rs:70-83, map.rs:78-86), on the
 * device for batches: the keys of the segment's posting lists — key j names list j of nidx_gpu_vector_set_filter_index — as
 * bytewise-sorted strings in HBM (labels_key / FieldKey bytes, inverted_index/paragraph.rs:63-103, however the caller encodes
 * them), and a batched lookup: for every query string the range [first, last) of lists whose key EQUALS it (query_is_prefix[q]
 * == 0: FstIndexReader::get) or STARTS WITH it (!= 0: get_prefix).  A PrefilterResult::Some of thousands of field ids
 * (searcher.rs:300-313) is one call. */
int32_t nidx_gpu_vector_set_filter_keys(nidx_gpu_vector_index_t *index, uint32_t segment, const uint8_t *key_bytes, const uint64_t *key_offsets,
                                        uint32_t n_keys);
int32_t nidx_gpu_vector_lookup_filter_keys(nidx_gpu_vector_index_t *index, uint32_t segment, const uint8_t *query_bytes,
                                           const uint64_t *query_offsets, const uint8_t *query_is_prefix, uint32_t n_queries,
                                           uint32_t *out_first, uint32_t *out_last);

enum {
    NIDX_FILTER_PUSH_LISTS = 0, /* push the union of posting lists lists[a .. b) (an AtomClause) */
    NIDX_FILTER_AND = 1,        /* pop 2, push their intersection */
    NIDX_FILTER_OR = 2,         /* pop 2, push their union */
    NIDX_FILTER_NOT = 3,        /* complement the top (BooleanOperator::Not = complement of the AND of its operands) */
    NIDX_FILTER_PUSH_ALL = 4,
    NIDX_FILTER_PUSH_NONE = 5
};
typedef struct { int32_t op; uint32_t a, b; } nidx_gpu_filter_op_t;
typedef struct {
    const nidx_gpu_filter_op_t *ops; /* NULL / 0 => no filter on this segment */
    uint32_t n_ops;
    const uint32_t *lists;           /* list-id table referenced by PUSH_LISTS */
    uint32_t n_lists;
} nidx_gpu_filter_program_t;

/* Searcher::search_multi_vector (searcher.rs:345-394) for VectorCardinality::Multi indexes: query q is the
 * query_vec_offsets[q+1] - query_vec_offsets[q] vectors starting at queries[query_vec_offsets[q] * dimension].  Every
 * query vector is searched on its own (max(k, 10) hits, duplicates kept, no min_score), the paragraphs found are
 * scored with maxsim_similarity (multivector.rs:33-46: sum over the query vectors of the best similarity among the
 * paragraph's vectors), those with score > min_score are ranked (score desc; ties by segment, paragraph) and cut to
 * k.  Paragraphs are de-duplicated by their address alone, like the reference (searcher.rs:375-377).
 * out_segment / out_paragraph / out_score: [n_queries][k]. */
int32_t nidx_gpu_vector_search_maxsim(nidx_gpu_vector_index_t *index, const float *queries, const uint64_t *query_vec_offsets,
                                      uint32_t n_queries, const nidx_gpu_vector_search_params_t *params,
                                      const uint64_t *const *segment_filters, uint32_t *out_segment, uint32_t *out_paragraph,
                                      float *out_score, uint32_t *out_count);

/* nidx_gpu_vector_search_dim with the filter of every segment given as a program (segment_programs:
 * NULL or [n_segments]).  out_matching: NULL or [n_segments] = |filter ∩ alive| per segment. */
int32_t nidx_gpu_vector_search_filtered(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries,
                                        uint32_t query_dimension, const nidx_gpu_vector_search_params_t *params,
                                        const nidx_gpu_filter_program_t *segment_programs, uint32_t *out_segment,
                                        uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                                        uint32_t *out_count, int32_t *out_method, uint64_t *out_matching);

/* OpenSegment::search on ONE segment with everything resident in HBM (segment.rs:477-567):
 * device pointers, asynchronous on `stream` (a hipStream_t; NULL = the null stream).
 *   d_queries [n_queries][dimension] f32 (already normalised if the index wants that)
 *   d_filter  NULL or device bitset over paragraph addrs (ANDed with alive on device)
 *   d_out_vector/d_out_score [n_queries][k], d_out_count [n_queries]
 *   d_stats   NULL or [n_queries][8] u32: distance evals, expansions, visited-set peak, flags,
 *             then wave-0 cycle counts: control / distance / admission / total */
int32_t nidx_gpu_vector_segment_search_device(nidx_gpu_vector_index_t *index, uint32_t segment,
                                              const float *d_queries, uint32_t n_queries,
                                              const nidx_gpu_vector_search_params_t *params,
                                              const uint64_t *d_filter, uint32_t *d_out_vector,
                                              float *d_out_score, uint32_t *d_out_count, uint32_t *d_stats,
                                              void *stream);

/* Every launch made through nidx_gpu_vector_segment_search_device ORs the NIDX_FLAG_* its queries raised (a bounded on-chip
 * structure overflowed: the answer of that query needs the exact fallback) into one device word.  This call reads and clears it
 * (it synchronises `stream`): 0 = every launch since the previous call already produced the final, exact hits, which is what a
 * serving loop polls once per delivery instead of reading n_queries x 8 counters back.  Flagged launches are repeated through
 * nidx_gpu_vector_segment_search_device_exact. */
int32_t nidx_gpu_vector_device_flags(nidx_gpu_vector_index_t *index, void *stream, uint32_t *flags_out);

/* OpenSegment::search on one segment, device-resident queries, complete: the launch, ONE device-to-host transfer of the result
 * block, and — only when the block's flag word is set — the re-run of the flagged queries with the 2^15 visited table /
 * the HBM-resident closest_up_nodes walk (the reference's heap and visited set are unbounded, hnsw/search.rs:188-304).
 *   d_out_block     device, n_queries*k u32 vectors | n_queries*k f32 scores | n_queries u32 counts | 1 u32 flag word
 *   host_out_block  NULL or pinned host memory of the same size: receives the block (the call returns with it filled)
 *   n_retried_out   NULL or the number of queries that took the fallback
 * Synchronous on `stream`. */
int32_t nidx_gpu_vector_segment_search_device_exact(nidx_gpu_vector_index_t *index, uint32_t segment, const float *d_queries,
                                                    uint32_t n_queries, const nidx_gpu_vector_search_params_t *params,
                                                    const uint64_t *d_filter, uint32_t *d_out_block, uint32_t *host_out_block,
                                                    void *stream, uint32_t *n_retried_out);

/* Measurement probe (no reference counterpart): read-only random gathers of whole `dimension`-float rows from a device matrix —
 * the row access of the HNSW kernels without any traversal — `waves` wavefronts x `gathers_per_wave` rows, `rows_in_flight`
 * (1, 2, 4, 8) rows requested per wave before the first is consumed.  ms_out = average launch time over `repeats` launches after
 * one warm-up; rows x 4*dimension / time is the measured gather ceiling bench.py quotes beside the roofline fraction. */
int32_t nidx_gpu_diag_gather(const float *d_rows, uint32_t n_rows, uint32_t dimension, uint32_t waves, uint32_t gathers_per_wave,
                             int32_t rows_in_flight, uint32_t repeats, float *ms_out);

/* Measurement probe: `threads` native threads issue `calls` nidx_gpu_vector_search_one requests in total, back to back (query c
 * = queries[c % n_queries]); latencies_us_out[calls] = wall time of every call, *elapsed_s_out = the whole run.  The reference's
 * serving shape (one blocking thread per request, src/searcher/shard_search.rs:139-153) without an interpreter in the way. */
int32_t nidx_gpu_diag_single_query_latency(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries, uint32_t dimension,
                                           const nidx_gpu_vector_search_params_t *params, uint32_t threads, uint32_t calls,
                                           float *latencies_us_out, double *elapsed_s_out);

/* ---- pipelined serving -----------------------------------------------------------------------------------
 * Replaces the loop the reference runs around VectorSearcher::search (lib.rs:120-148; one call per request from
 * src/searcher/shard_search.rs:139-153) for callers that have batches: the library owns "pipeline_depth" slots (tunable, default
 * 4, at most 16) — a HIP stream, pinned staging and a device result block each — so that several batches are in flight and the
 * walk-length tail of one launch overlaps the body of the next (a launch lasts as long as its longest walk).  The library sets
 * GPU_MAX_HW_QUEUES=8 at load time unless the variable is already set (streams beyond the runtime's 4 hardware queues share a
 * queue and serialise).
 *
 * submit: `queries` = [n_queries][dimension] f32 rows in host memory (copied before the call returns; normalised when the index
 * says so) or in DEVICE memory (searched where they lie: dimension % 4 == 0, an index that does not normalise, rows complete —
 * their producing stream synchronised — and untouched until the ticket has been waited for).  segment_filters as in
 * nidx_gpu_vector_search (host bitsets, copied before the call returns).  Every segment's search is launched on the slot's
 * stream and ONE device-to-host transfer of the result block is queued behind them; the call returns without waiting for
 * either.  NIDX_ERR_BUSY when every slot holds a ticket that has not been waited for (nothing was launched).
 *
 * wait: blocks until the block of `ticket` has landed in pinned host memory, re-runs a segment whose flag word says a bounded
 * on-chip structure overflowed through the exact fallback (as nidx_gpu_vector_segment_search_device_exact does; *n_retried_out
 * = queries that took it), merges the segments with Fssc and fills the caller's [n_queries][k] arrays exactly as
 * nidx_gpu_vector_search would have.  A ticket is waited for once; waits may come in any order and from any thread. */
int32_t nidx_gpu_vector_search_submit(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries, uint32_t query_dimension,
                                      const nidx_gpu_vector_search_params_t *params, const uint64_t *const *segment_filters,
                                      uint64_t *ticket_out);
int32_t nidx_gpu_vector_search_wait(nidx_gpu_vector_index_t *index, uint64_t ticket, uint32_t *out_segment, uint32_t *out_paragraph,
                                    uint32_t *out_vector, float *out_score, uint32_t *out_count, uint32_t *n_retried_out);

/* One query per call, the shape of the reference's request path (one blocking thread per Search request,
 * src/searcher/shard_search.rs:139-153; one vector per request, nodereader.proto:402).  Thread safe:
 * concurrent callers with equal params are coalesced into batched launches that run through the pipeline above, up to
 * "coalesce_in_flight" (default 4) of them at a time: while batches are running the next one gathers, and it is closed when its
 * window ("coalesce_window_us", default 50, counted from the moment it starts gathering) has passed or it is full
 * ("coalesce_max_batch", default 1024) AND a slot is free — so under load the batch size adapts to the arrival rate.
 * Admission: at most "coalesce_max_callers" (default 256, 0 = unbounded) requests are inside at once; further callers wait at
 * the door and are let in as requests leave, or — tunable "coalesce_reject_when_full" = 1 — get NIDX_ERR_BUSY at once (what a
 * gRPC front end turns into RESOURCE_EXHAUSTED): 1 024 blocked callers then see the latencies of 256, not a scheduler convoy.
 * Unfiltered; outputs are [k] rows.  Blocks until this query's hits are ready. */
int32_t nidx_gpu_vector_search_one(nidx_gpu_vector_index_t *index, const float *query, uint32_t query_dimension,
                                   const nidx_gpu_vector_search_params_t *params, uint32_t *out_segment,
                                   uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count);
/* Launches and queries served through nidx_gpu_vector_search_one so far. */
int32_t nidx_gpu_vector_coalescer_stats(nidx_gpu_vector_index_t *index, uint64_t *batches_out, uint64_t *queries_out);
/* Queries whose closest_up_nodes walk (hnsw/search.rs:188-240) outgrew the on-chip candidate pool / visited table
 * and were re-run by the exact HBM-resident fallback (the reference's heap and visited set are unbounded), since
 * open.  Results are the same either way; the counter tells how often the slow path ran. */
int32_t nidx_gpu_vector_spill_stats(nidx_gpu_vector_index_t *index, uint64_t *queries_out);

/* DataStoreV2::create's quantized writer (data_store/v2.rs:57-76) for one segment of an open index:
 * EncodedVector::encode (rabitq.rs:75-106) of every vector, on the device; the segment then has a
 * quantized store.  NIDX_ERR_INVALID_CONFIGURATION unless the config is quantizable (Dot similarity and
 * dimension % 64 == 0, config.rs:170-173). */
int32_t nidx_gpu_vector_quantize(nidx_gpu_vector_index_t *index, uint32_t segment);
/* The bytes of vectors.quant.  Call with out == NULL to get the length. */
int32_t nidx_gpu_vector_serialize_quantized(nidx_gpu_vector_index_t *index, uint32_t segment, uint8_t *out,
                                            uint64_t out_cap, uint64_t *len_out);

/* use_hnsw (segment.rs:626-660) — exposed so callers can route exactly like the reference. */
int32_t nidx_gpu_use_hnsw(uint64_t total_nodes, uint64_t matching_nodes, uint64_t top_k, int32_t has_rabitq);

/* Pairwise similarity of rows (host in/out), the numerics of dense_f32::{dot,cosine}_similarity
 * (vector_types/dense_f32.rs:29-39) in the given summation order: out[i] = sim(x[i], y[i]). */
int32_t nidx_gpu_similarity(const float *x, const float *y, uint32_t n_pairs, uint32_t dimension,
                            int32_t similarity, int32_t order, float *out);
/* utils::normalize_vector (utils.rs:20-23) for n rows, host in/out. */
int32_t nidx_gpu_normalize(const float *in, uint32_t n, uint32_t dimension, float *out);

/* Host-only check of an hnsw.graph image (and, when given, its hnsw.edges weights) for `n_nodes` vectors — the validation
 * nidx_gpu_vector_open applies before the graph goes to HBM (node offsets, layer offsets, degrees <= M_max, edge targets,
 * entry point; hnsw/disk/v2.rs:16-49), without a device.  NIDX_ERR_INVALID_GRAPH + message when malformed.  Outputs (each may be
 * NULL): the entry point (node, layer), the number of edges, and the number of links fix_broken_graph would drop
 * (ram_hnsw.rs:109-143: links into a layer the target does not live on). */
int32_t nidx_gpu_hnsw_graph_check(const uint8_t *graph, uint64_t graph_len, const float *edges, uint64_t n_edges, uint32_t n_nodes,
                                  uint32_t *entry_node_out, uint32_t *entry_layer_out, uint64_t *n_links_out,
                                  uint64_t *n_broken_links_out);

/* HnswBuilder (hnsw/build.rs:28-167) on the device for one segment of an open index: level
 * draw from SmallRng::seed_from_u64(level_seed) (build.rs:36-55; the reference uses 2), batched
 * concurrent inserts with M=30/M0=60/efC=100 (hnsw/params.rs:20-46).  Replaces any graph the
 * segment had.  Like the reference's rayon build the graph is not unique; parity is recall. */
int32_t nidx_gpu_vector_build_hnsw(nidx_gpu_vector_index_t *index, uint32_t segment, uint64_t level_seed);
/* The work of the last nidx_gpu_vector_build_hnsw / _extend_hnsw of this index, counted by the build kernels the way the search
 * kernel counts its own (measurement; HnswBuilder::insert, hnsw/build.rs:97-167): stats_out[10] (8 before ABI version 6) =
 *   [0] nodes inserted, [1] microseconds from the first batch's launch to the last one's completion,
 *   [2] distance evaluations and [3] expansions of the construction searches (layer_search with ef = 100, build.rs:137-166),
 *   [4] rows read by select_neighbours_heuristic over the new nodes' own candidate lists (build.rs:57-95, 104-108),
 *   [5] rows read by the reverse-link prunes (build.rs:111-119), [6] reverse-link appends, [7] prunes.
 *   [8] the first node that was inserted with the large visited table of the construction searches (the build starts with a 2^12
 *       table and moves to the configured one when a batch reports it three quarters full; 0 = it never did),
 *   [9] the NIDX_FLAG_* the build ended with (bit 0: a construction search filled the LARGE table and ended early).
 * Algorithmic bytes of the build = ([2] + [4] + [5]) x 4 x dimension + [3] x 256.  [2] = ~0 when the build ran with
 * NIDX_GPU_BUILD_STATS=0 (no counters). */
int32_t nidx_gpu_vector_build_stats(nidx_gpu_vector_index_t *index, uint64_t *stats_out);
/* segment::merge's graph reuse (segment.rs:137-167): the segment was opened with an hnsw.graph image of
 * its first hnsw_graph_nodes vectors (the largest, deletion-free operand of the merge); draw levels
 * for the remaining nodes from a fresh SmallRng::seed_from_u64(level_seed) (build.rs:50-55,
 * initialize_graph(skip_nodes, total)) and insert only those. */
int32_t nidx_gpu_vector_extend_hnsw(nidx_gpu_vector_index_t *index, uint32_t segment, uint64_t level_seed);
/* DiskHnswV2::serialize_to (hnsw/disk/v2.rs:109-218): writes the segment's graph as hnsw.graph /
 * hnsw.edges bytes.  Call with NULL buffers to get the sizes. */
int32_t nidx_gpu_vector_serialize_hnsw(const nidx_gpu_vector_index_t *index, uint32_t segment, uint8_t *graph_out,
                                       uint64_t graph_cap, uint64_t *graph_len_out, float *edges_out,
                                       uint64_t edges_cap, uint64_t *n_edges_out);

/* ---- segment directories -------------------------------------------------------------
 * A vector segment as the reference lays it out on disk: vectors.bin (data_store/v2/vector_store.rs:30-40,
 * 131-147), paragraphs.bin / paragraphs.pos (data_store/v2/paragraph_store.rs:37-44,100-106,132-150; the
 * StoredParagraph records in the bincode-2 "standard" layout of utils.rs:25-28), vectors.quant
 * (data_store/v2/quant_vector_store.rs:29-64), hnsw.graph / hnsw.edges (hnsw/disk/v2.rs).  Host side only (no
 * device call): the files are mmap'd and the views below feed nidx_gpu_vector_open and
 * nidx_gpu_vector_set_filter_index without a copy; everything stays valid until _close.
 * field.fst / label.fst / index.map (inverted_index/{fst_index.rs:26-87, map.rs:27-86, paragraph.rs:68-121}: two
 * fst::Map images key -> offset into index.map, whose records are a u64 count + the stream-vbyte encoded paragraph
 * addresses) are read when index.map exists (InvertedIndexes::exists, inverted_index.rs:57-60) and every list in
 * them is a well-formed list of this paragraph store; otherwise — like segment::open when they are missing
 * (segment.rs:49-67) — the posting lists are rebuilt from the paragraph store (ParagraphInvertedIndexes::build,
 * paragraph.rs:68-103).  nidx_gpu_segment_dir_write / _merge write the three files when NIDX_GPU_SEGMENT_DIR_FST=1 is in the
 * environment; by default they leave them out — the reference regenerates them on open — because the two containers are
 * third-party formats (fst 0.4.7, stream-vbyte 0.4.1) restated without the crates at hand (see fst_index.cpp): until the
 * one-time check of INTEGRATION.md section 3 has been run against the crates, a directory without them is the safe one to
 * hand to the stock searcher.  NIDX_GPU_SEGMENT_DIR_FST=0: neither written nor read.
 * A pre-migration directory — nodes.kv (DataStoreV1, data_store/v1.rs) and index.hnsw (DiskHnswV1, hnsw/disk/v1.rs), what
 * segment::open takes first when nodes.kv exists (segment.rs:41-57) and open_disk_hnsw falls back to (hnsw/disk.rs:25-32) —
 * is migrated in memory at open: its records and graph are re-laid out as the vectors.bin / paragraphs.bin / paragraphs.pos /
 * hnsw.graph / hnsw.edges images of the same segment (one vector per paragraph), and everything below behaves as if those
 * files had been mapped; nidx_gpu_segment_dir_merge therefore writes the current formats from it (segment.rs:117-128). */
typedef struct nidx_gpu_segment_dir nidx_gpu_segment_dir_t;
int32_t nidx_gpu_segment_dir_open(const char *path, uint32_t dimension, nidx_gpu_segment_dir_t **dir_out);
void nidx_gpu_segment_dir_close(nidx_gpu_segment_dir_t *dir);
/* 1: the posting lists came from field.fst / label.fst / index.map; 0: rebuilt from the paragraph store; -1: NULL */
int32_t nidx_gpu_segment_dir_index_source(const nidx_gpu_segment_dir_t *dir);
/* The nidx_gpu_vector_segment_t of the directory (alive_bitset NULL: apply_deletions is the caller's,
 * segment.rs:428-445; paragraph_key_ids = a 64-bit hash of every key). */
int32_t nidx_gpu_segment_dir_segment(const nidx_gpu_segment_dir_t *dir, nidx_gpu_vector_segment_t *segment_out);
/* The rebuilt inverted indexes as posting lists, the field-key lists first, then the label lists, each group in
 * key order. */
int32_t nidx_gpu_segment_dir_filter_index(const nidx_gpu_segment_dir_t *dir, nidx_gpu_filter_index_t *lists_out);
/* The lookups the FSTs serve: the ids [first, first + count) of the posting lists selected by
 *   NIDX_LIST_LABEL  a label such as "/l/set/label": label_index.get_prefix(labels_key(label)) — the label and
 *                    its children (inverted_index/paragraph.rs:63-66,144-146); `prefix` is ignored;
 *   NIDX_LIST_FIELD  a field id "uuid/type/name" (or a bare "uuid"), keyed as FieldKey::from_field_id does
 *                    (utils.rs:84-115): prefix == 0 is field_index.get (AtomClause::KeyPrefixSet,
 *                    paragraph.rs:147-151), prefix != 0 is field_index.get_prefix (ids_for_deletion_key, :118-120).
 * An id FieldKey::from_field_id rejects selects nothing (count 0). */
enum { NIDX_LIST_LABEL = 0, NIDX_LIST_FIELD = 1 };
int32_t nidx_gpu_segment_dir_lists(const nidx_gpu_segment_dir_t *dir, int32_t kind, const uint8_t *key, uint32_t key_len,
                                   int32_t prefix, uint32_t *first_out, uint32_t *count_out);
/* StoredParagraph of one address (DataStore::get_paragraph): what try_to_document_scored (searcher.rs:126-146)
 * reads to assemble a hit.  Pointers into the mapped paragraphs.bin, not NUL terminated. */
typedef struct {
    const char *key;
    uint32_t key_len;
    const uint8_t *metadata;
    uint32_t metadata_len;
    uint32_t n_labels;
    uint32_t first_vector, num_vectors;
} nidx_gpu_paragraph_t;
int32_t nidx_gpu_segment_dir_paragraph(const nidx_gpu_segment_dir_t *dir, uint32_t addr, nidx_gpu_paragraph_t *out);
int32_t nidx_gpu_segment_dir_paragraph_label(const nidx_gpu_segment_dir_t *dir, uint32_t addr, uint32_t i,
                                             const char **label_out, uint32_t *len_out);
/* segment::create's file output (segment.rs:199-239) for a segment built or merged on the device: vectors and
 * paragraphs in address order; the graph image / edge weights as nidx_gpu_vector_serialize_hnsw returns them, the
 * quantized store as nidx_gpu_vector_serialize_quantized does (NULL = file not written). */
typedef struct {
    uint32_t dimension, n_vectors, n_paragraphs;
    const float *vectors;                    /* [n_vectors][dimension] */
    const uint32_t *paragraph_of_vector;     /* NULL = one vector per paragraph, in order */
    const uint8_t *keys;
    const uint64_t *key_offsets;             /* [n_paragraphs + 1] */
    const uint8_t *labels;
    const uint64_t *label_offsets;           /* [n_labels_total + 1] */
    const uint64_t *paragraph_label_offsets; /* [n_paragraphs + 1] into the label table; NULL = no labels */
    const uint8_t *metadata;
    const uint64_t *metadata_offsets;        /* [n_paragraphs + 1]; NULL = no metadata */
    const uint8_t *hnsw_graph;
    uint64_t hnsw_graph_len;
    const float *hnsw_edges;
    uint64_t n_hnsw_edges;
    const uint8_t *quantized;
    uint64_t quantized_len;
} nidx_gpu_segment_dir_contents_t;
int32_t nidx_gpu_segment_dir_write(const char *path, const nidx_gpu_segment_dir_contents_t *contents);

/* The file side of segment::merge (segment.rs:92-135) / DataStoreV2::merge (data_store/v2.rs:82-128): the operands sorted
 * largest first (stored paragraphs, stable), the alive paragraphs of each copied in address order with their vectors (new
 * trailers), their StoredParagraph records (new first_vector) and — when every operand has a vectors.quant — their RaBitQ
 * records.  Writes vectors.bin, paragraphs.bin/.pos, vectors.quant under `path` (an existing directory).  When no paragraph
 * of the largest operand is deleted and it has a graph, hnsw.graph / hnsw.edges are copied and *graph_nodes_out = its vector
 * count (merge_indexes, segment.rs:143-158): open the result with hnsw_graph_nodes = that value and finish the graph with
 * nidx_gpu_vector_extend_hnsw; otherwise *graph_nodes_out = 0 and the graph is built from scratch.  *has_quantized_out = 0 on
 * a quantizable index means some operand had no codes: nidx_gpu_vector_quantize re-encodes the merged segment on the device. */
/* OpenSegment::apply_deletions (segment.rs:428-445): clears in `alive_bitset` (one bit per stored paragraph, initialised by
 * the caller) the paragraphs of every deletion key — a resource uuid or `uuid/type/name` — found by
 * field_index.get_prefix(FieldKey::from_field_id(key)) (inverted_index/paragraph.rs:118-120; a BYTE prefix: `…/t/title` also
 * reaches `…/t/title2`).  Keys that are no field id delete nothing.  *n_cleared_out = paragraphs that were alive. */
int32_t nidx_gpu_segment_dir_apply_deletions(const nidx_gpu_segment_dir_t *dir, const char *const *keys, const uint32_t *key_lens,
                                             uint32_t n_keys, uint64_t *alive_bitset, uint32_t *n_cleared_out);

typedef struct {
    const nidx_gpu_segment_dir_t *dir;
    const uint64_t *alive_bitset; /* one bit per stored paragraph (apply_deletions, segment.rs:428-445); NULL = all alive */
} nidx_gpu_merge_operand_t;
int32_t nidx_gpu_segment_dir_merge(const char *path, uint32_t dimension, const nidx_gpu_merge_operand_t *operands,
                                   uint32_t n_operands, uint32_t *records_out, uint32_t *vectors_out, uint32_t *graph_nodes_out,
                                   int32_t *has_quantized_out);

/* The containers on their own (tooling and tests).  _fst_map_build: FstIndexWriter::write's fst::MapBuilder (fst_index.rs:
 * 40-50) over n strictly ascending keys; call with out = NULL for the size.  _fst_map_get: FstIndexReader::get's lookup
 * (:66-68).  _fst_map_entries: the stream of get_prefix (:76-86) with an empty prefix — every key in order (sizes first with
 * NULL buffers).  _index_map_read: InvertedMapReader::get (map.rs:63-70). */
int32_t nidx_gpu_fst_map_build(const uint8_t *keys, const uint64_t *key_offsets, const uint64_t *values, uint32_t n, uint8_t *out,
                               uint64_t cap, uint64_t *len_out);
int32_t nidx_gpu_fst_map_get(const uint8_t *image, uint64_t len, const uint8_t *key, uint32_t key_len, uint64_t *value_out,
                             int32_t *found_out);
int32_t nidx_gpu_fst_map_entries(const uint8_t *image, uint64_t len, uint8_t *keys_out, uint64_t keys_cap, uint64_t *key_offsets_out,
                                 uint64_t *values_out, uint32_t cap, uint32_t *n_out, uint64_t *keys_len_out);
int32_t nidx_gpu_index_map_read(const uint8_t *map, uint64_t len, uint64_t pos, uint32_t *ids_out, uint32_t cap, uint32_t *n_out);

/* =====================================================================================
 * BM25 index — replaces the tantivy scoring under TextSearcher::search
 * (nidx_text/src/reader.rs:367-451) and ParagraphSearcher::search
 * (nidx_paragraph/src/reader.rs:244-348): term-at-a-time BM25 + TopDocs.
 * ===================================================================================== */

/* One tantivy segment's postings for the scored text field, already term-id resolved. */
typedef struct {
    uint32_t n_docs;              /* max_doc (deleted docs included: statistics do not shrink) */
    uint64_t total_num_tokens;    /* Σ fieldnorm over docs (tantivy FieldNormReader totals) */
    uint32_t n_terms;
    const uint64_t *term_offsets; /* [n_terms+1] into doc_ids / tfs */
    const uint32_t *doc_ids;      /* ascending within a term */
    const uint32_t *tfs;          /* term frequency per posting; must be < 2^24 (the resident posting word is tf | fieldnorm id << 24:
                                   * NIDX_ERR_UNSUPPORTED at open otherwise) */
    const uint8_t *fieldnorm_ids; /* [n_docs] 1-byte fieldnorm ids */
    const uint64_t *alive_bitset; /* open_index_with_deletions (nidx_tantivy/src/index_reader.rs:39-74); NULL = all */
    /* positions of every posting (the text field is indexed WithFreqsAndPositions, schema.rs:59-115): posting i owns
     * positions[pos_offsets[i] .. pos_offsets[i+1]), ascending.  NULL = no positions: phrase clauses are refused */
    const uint64_t *pos_offsets;  /* [n_postings + 1] */
    const uint32_t *positions;
} nidx_gpu_bm25_segment_t;

typedef struct nidx_gpu_bm25_index nidx_gpu_bm25_index_t;

/* Opens the segments of one index (open_index_with_deletions, nidx_tantivy/src/index_reader.rs:39-74).  tantivy searches them one
 * after the other under one searcher.search with searcher-wide Bm25Weight statistics and merges by (score, DocAddress)
 * (nidx_text/src/reader.rs:433-435, nidx_paragraph/src/reader.rs:244-348); the log-merge policy leaves several
 * (nidx/src/settings.rs:246-253).  Here an index of several segments is resident as ONE posting layout, term-major across the
 * segments over doc + base[segment] (base = running sum of n_docs): every search is one launch sequence whatever the number of
 * segments, and the merge across segments is the scorer's own slice merge on the device.  DocAddresses in and out (hits, search-after
 * cursors, prefilter results) and the `segment` arguments of nidx_gpu_bm25_set_fast_field / _apply_deletions keep naming the opened
 * segments.  The segments together hold at most 2^32 - 1 documents.  NIDX_GPU_BM25_SEGMENT_LOOP=1 in the environment keeps one resident
 * segment per opened segment and searches them in a host loop (round 4's path; the tests compare the two). */
int32_t nidx_gpu_bm25_open(const nidx_gpu_bm25_segment_t *segments, uint32_t n_segments,
                           nidx_gpu_bm25_index_t **index_out);
void nidx_gpu_bm25_close(nidx_gpu_bm25_index_t *index);
int32_t nidx_gpu_bm25_space_usage(const nidx_gpu_bm25_index_t *index, uint64_t *bytes_out);

/* NIDX_OCCUR_SHOULD_GROUP: a Should clause of a nested Must(BooleanQuery[Should ...]) — the paragraph
 * keyword query under its Must filters (nidx_paragraph/src/search_query.rs:185-243): it scores like a
 * Should, and a document has to match at least one clause of the group. */
/* NIDX_OCCUR_SHOULD_GROUP + g, g < 8: further required Should groups (a document needs a clause of EVERY group): the
 * prefilter's BooleanQuery[Should SetQuery(field_uuid), Should SetQuery(uuid)] and an Or formula beside the keyword group
 * (nidx_paragraph/src/search_query.rs:88-143). */
enum { NIDX_OCCUR_SHOULD = 0, NIDX_OCCUR_MUST = 1, NIDX_OCCUR_MUST_NOT = 2, NIDX_OCCUR_SHOULD_GROUP = 3 };
/* NIDX_TF_FREQ: BM25 with the stored tf (nidx_text, IndexRecordOption::WithFreqs);
 * NIDX_TF_BASIC: tf == 1 (nidx_paragraph keyword terms, query_parser/keyword_parser.rs:62-67);
 * NIDX_CONST_SCORE: ConstScorer(boost) (prefilter SetQuery, nidx_paragraph/src/search_query.rs:105-139) */
enum { NIDX_TF_FREQ = 0, NIDX_TF_BASIC = 1, NIDX_CONST_SCORE = 2 };

typedef struct {
    uint32_t term;   /* term id (per-index dictionary order; the same id in every segment) */
    int32_t occur;   /* NIDX_OCCUR_* */
    int32_t mode;    /* NIDX_TF_* / NIDX_CONST_SCORE */
    float boost;
} nidx_gpu_bm25_clause_t;

/* search-after cursor (nidx_paragraph/src/reader.rs:350-390) */
typedef struct {
    int32_t has_after;
    float score;
    int32_t tie_break;   /* 0: keep all ties, 1: keep docaddr > cursor, 2: drop ties */
    uint64_t docaddr;
} nidx_gpu_bm25_search_after_t;

/* A batch of BooleanQuery's over term clauses, TopDocs::with_limit(k).order_by_score() + Count
 * (nidx_text/src/reader.rs:433-435).  Query q owns clauses[clause_offsets[q] .. clause_offsets[q+1]).
 *   after            NULL or [n_queries]
 *   out_docaddr      [n_queries][k]  (segment_ord << 32) | doc_id  (nidx_text/src/reader.rs:310)
 *   out_score        [n_queries][k]  score desc, docaddr asc on ties
 *   out_count        [n_queries]     hits written
 *   out_total        [n_queries]     matching alive docs (Count collector)
 *   out_postings     NULL or [n_queries]: postings scored (the BASELINE "docs scored" unit) */
int32_t nidx_gpu_bm25_search(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses,
                             const uint64_t *clause_offsets, uint32_t n_queries, uint32_t k,
                             const nidx_gpu_bm25_search_after_t *after, uint64_t *out_docaddr,
                             float *out_score, uint32_t *out_count, uint64_t *out_total,
                             uint64_t *out_postings);

/* ---- the collectors and query shapes around the scorer ---- */

/* A clause whose `term` is NIDX_BM25_TERM_SET | j matches the UNION of the posting lists of term set j
 * (options.term_set_terms[term_set_offsets[j] .. term_set_offsets[j+1])), each document once, scored
 * ConstScorer(boost): FuzzyTermQuery / AutomatonWeight (nidx_paragraph/src/fuzzy_query.rs:55-125). */
#define NIDX_BM25_TERM_SET 0x80000000u
/* A clause whose `term` is NIDX_BM25_PHRASE | j is PhraseQuery(options.phrase_terms[phrase_offsets[j] ..
 * phrase_offsets[j+1])) (keyword_parser.rs:69-91 for multi-word quotes; tantivy's QueryParser for nidx_text): with
 * slop 0 the terms at consecutive positions, tf = number of occurrences, Bm25Weight::for_terms (idf summed over the
 * terms).  options.phrase_slops[j] > 0 (`"a b"~2` in the QueryParser's grammar) = PhraseQuery::set_slop: tantivy's
 * PhraseScorer with slop — each next term may trail the match so far by up to `slop` extra positions, in order.
 * `mode` is ignored. */
#define NIDX_BM25_PHRASE 0x40000000u
/* A clause whose `term` is NIDX_BM25_SUBQUERY | j is a nested BooleanQuery: the leaves options.subquery_clauses[subquery_offsets[j]
 * .. subquery_offsets[j+1]), at most 32; occur / mode / boost as for top-level clauses, required Should groups included.  A leaf is
 * a term of the dictionary, a term set (NIDX_BM25_TERM_SET | s), a phrase (NIDX_BM25_PHRASE | p) or ANOTHER nested query
 * (NIDX_BM25_SUBQUERY | i with i < j: a tree's queries are listed children first), so boolean trees of any depth are expressible.
 * It matches a document when its own boolean structure does — every Must leaf, no MustNot leaf, a member of every required Should
 * group, and when it has neither Must leaves nor groups at least one Should leaf (a nested query with only MustNot leaves matches
 * nothing, like tantivy's) — its score there is the f32 sum of its scoring leaves that hold the document (leaf order), and the outer
 * clause contributes boost x that score (tantivy: BoostQuery over the nested BooleanQuery; `mode` of the outer clause is ignored).
 * This is what tantivy's QueryParser builds for parenthesised boolean expressions (nidx_text/src/reader.rs:357-376) and what nested
 * filtering formulas become (nidx_paragraph/src/search_query.rs:88-143). */
#define NIDX_BM25_SUBQUERY 0x20000000u

typedef struct {
    uint32_t k;                                  /* TopDocs limit (0 with facets = only_faceted) */
    const nidx_gpu_bm25_search_after_t *after;   /* NULL or [n_queries] (order by score only) */
    const uint32_t *term_set_terms;              /* term ids of every set, concatenated */
    const uint64_t *term_set_offsets;            /* [n_term_sets + 1] */
    uint32_t n_term_sets;
    /* NULL or [n_term_sets]: != 0 = the clause matches every document OUTSIDE the union — parse_excluded's
     * BooleanQuery[Must AllQuery, MustNot term] (nidx_paragraph/src/query_parser/keyword_parser.rs:93-105) */
    const uint8_t *term_set_complement;
    const uint32_t *phrase_terms;                /* term ids of every phrase, concatenated */
    const uint64_t *phrase_offsets;              /* [n_phrases + 1] */
    uint32_t n_phrases;
    /* TopDocs::order_by_fast_field (nidx_text/src/reader.rs:210-224, custom_order_collector): -1 = by score,
     * else the fast field registered with nidx_gpu_bm25_set_fast_field (0 = created, 1 = modified) */
    int32_t order_field;
    int32_t order_desc;
    /* FacetCollector (nidx_text/src/reader.rs:391-398): query q wants the number of matching documents that
     * carry each of facet_terms[facet_offsets[q] .. facet_offsets[q+1]) (the term ids of the children of the
     * requested facets); NULL = no facets */
    const uint32_t *facet_terms;
    const uint64_t *facet_offsets;               /* [n_queries + 1] */
    uint64_t *out_facet_counts;                  /* [facet_offsets[n_queries]], summed over segments */
    int64_t *out_order_value;                    /* NULL or [n_queries][k]: the fast value of every hit (order_field >= 0) */
    const nidx_gpu_bm25_clause_t *subquery_clauses;   /* leaves of every nested BooleanQuery, concatenated (NIDX_BM25_SUBQUERY) */
    const uint64_t *subquery_offsets;                 /* [n_subqueries + 1] */
    uint32_t n_subqueries;
    const uint32_t *phrase_slops;                     /* NULL (every phrase exact) or [n_phrases]: PhraseQuery::set_slop */
} nidx_gpu_bm25_search_options_t;

/* nidx_gpu_bm25_search with the collectors above; out_score is the BM25 score when ordering by score and 0
 * when ordering by a fast field (the reference returns the sort value instead, reader.rs:262-270). */
int32_t nidx_gpu_bm25_search_ex(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses,
                                const uint64_t *clause_offsets, uint32_t n_queries,
                                const nidx_gpu_bm25_search_options_t *options, uint64_t *out_docaddr,
                                float *out_score, uint32_t *out_count, uint64_t *out_total, uint64_t *out_postings);

/* The same search, pipelined (the loop the reference runs around TextSearcher / ParagraphSearcher::search, nidx_text/src/lib.rs:178-237,
 * nidx_paragraph/src/lib.rs:117-169, one call per request from src/searcher/shard_search.rs:176-248, for a caller that has batches):
 * submit prepares the batch (clause weights, work list, staging), queues the launches and ONE device-to-host transfer of the result
 * block on a stream of its own and returns; wait blocks until that block has landed and fills the caller's arrays exactly as
 * nidx_gpu_bm25_search_ex would have.  With two tickets outstanding the host side of batch i + 1 overlaps the kernels of batch i.  At most 16
 * tickets may be outstanding (NIDX_ERR_BUSY otherwise); a ticket is waited for once, from any thread.  Several threads may submit at the
 * same time: every ticket's batch is planned and launched on a context of its own (the planning of a batch costs the submitting thread
 * more than its kernels cost the device).  An index of several segments goes through the pipeline like one of a single segment (it is
 * resident as one term-major posting layout, see nidx_gpu_bm25_open).  A request the pipeline does not cover — term sets, phrases,
 * nested queries, facets, order by a fast field — runs to completion inside submit (options.out_facet_counts / out_order_value are
 * filled there) and wait only hands its hits over.  clauses / clause_offsets / options
 * need not outlive submit. */
int32_t nidx_gpu_bm25_search_submit(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_clause_t *clauses, const uint64_t *clause_offsets,
                                    uint32_t n_queries, const nidx_gpu_bm25_search_options_t *options, uint64_t *ticket_out);
int32_t nidx_gpu_bm25_search_wait(nidx_gpu_bm25_index_t *index, uint64_t ticket, uint64_t *out_docaddr, float *out_score, uint32_t *out_count,
                                  uint64_t *out_total, uint64_t *out_postings);

/* The `created` / `modified` fast fields of one segment (nidx_text/src/schema.rs:59-115): values[doc]. */
int32_t nidx_gpu_bm25_set_fast_field(nidx_gpu_bm25_index_t *index, uint32_t segment, uint32_t field, const int64_t *values);

/* The term dictionary of the scored text field: term id t = bytes[offsets[t] .. offsets[t+1]) (UTF-8). */
int32_t nidx_gpu_bm25_set_dictionary(nidx_gpu_bm25_index_t *index, const uint8_t *bytes, const uint64_t *offsets);
/* FuzzyTermQuery's automaton over the dictionary (fuzzy_query.rs:127-251; Levenshtein distance 1, a
 * transposition costs one, fuzzy_parser.rs:38-74; prefix != 0 = build_prefix_dfa): the ids of the accepted
 * terms, ascending.  n_out receives the full count even when it exceeds cap. */
int32_t nidx_gpu_bm25_fuzzy_terms(nidx_gpu_bm25_index_t *index, const uint8_t *query_utf8, uint32_t query_len,
                                  int32_t prefix, uint32_t *out_terms, uint32_t cap, uint32_t *n_out);

/* TextReaderService::prefilter (nidx_text/src/reader.rs:148-180): the documents (fields) that satisfy a boolean
 * filter expression, evaluated as bitset algebra on the device.  The expression is what filter_to_query
 * (nidx_text/src/search_query.rs:156-223) and security_query (ibid. 66-90) build, flattened by the caller to a
 * postfix nidx_gpu_filter_program_t whose posting lists are TERM IDS of this index (facet, field, uuid, group and
 * text terms alike live in one term-id space here):
 *   Facet / Field / Resource / single-token Keyword / security groups -> PUSH_LISTS over the term(s)
 *   BoolAnd / BoolOr / BoolNot                                        -> AND / OR / NOT (NOT = AllQuery minus operand)
 *   Date {field, since, until}                                        -> NIDX_FILTER_PUSH_RANGE a = index into `ranges`
 *                                                                        (both bounds INCLUSIVE, search_query.rs:30-49;
 *                                                                        neither present = AllQuery)
 *   multi-token Keyword (query_io.rs:22-42, a PhraseQuery)            -> NIDX_FILTER_PUSH_PHRASE a = index into phrases
 * Deleted documents never match.  out_docaddr receives the matches as (segment << 32) | doc, ascending, at most
 * `capacity` of them; *n_matching is the full count (call again with a larger buffer when it exceeds capacity) and
 * *num_docs the live documents of the index (searcher.num_docs()), so the caller derives PrefilterResult::
 * None (0) / All (== num_docs) / Some exactly as reader.rs:166-179 does. */
#define NIDX_FILTER_PUSH_RANGE 6
#define NIDX_FILTER_PUSH_PHRASE 7
typedef struct {
    uint32_t field;      /* 0 = created, 1 = modified (nidx_gpu_bm25_set_fast_field) */
    int32_t has_since;   /* value >= since */
    int32_t has_until;   /* value <= until */
    int32_t reserved;
    int64_t since, until;
} nidx_gpu_bm25_date_range_t;
typedef struct {
    nidx_gpu_filter_program_t program;
    const nidx_gpu_bm25_date_range_t *ranges;
    uint32_t n_ranges;
    uint32_t n_phrases;
    const uint32_t *phrase_terms;   /* term ids of phrase j: phrase_terms[phrase_offsets[j] .. phrase_offsets[j+1]) */
    const uint64_t *phrase_offsets; /* [n_phrases + 1] */
} nidx_gpu_bm25_prefilter_t;
int32_t nidx_gpu_bm25_prefilter(nidx_gpu_bm25_index_t *index, const nidx_gpu_bm25_prefilter_t *request,
                                uint64_t *out_docaddr, uint64_t capacity, uint64_t *n_matching, uint64_t *num_docs);

/* open_index_with_deletions (nidx_tantivy/src/index_reader.rs:39-74), the device half: the caller maps the deletion keys that
 * are newer than the segment (`Seq(segment) < del_seq`) to term ids the way the DeletionQueryBuilders do (a key longer than 32
 * bytes is a field id, else a resource uuid: nidx_text/src/lib.rs:95-128, nidx_paragraph/src/lib.rs:50-71); every document in
 * any of those posting lists leaves the segment's alive bitset (union of the lists scattered into a bitset, and-not).
 * n_alive_out: NULL or the segment's live documents afterwards. */
int32_t nidx_gpu_bm25_apply_deletions(nidx_gpu_bm25_index_t *index, uint32_t segment, const uint32_t *terms, uint32_t n_terms,
                                      uint64_t *n_alive_out);

/* Device time (HIP events on the handle's stream) spent in the scoring kernel(s) of the last
 * nidx_gpu_bm25_search call, summed over segments.  For batches of term unions with k <= 64 the scoring launch
 * also merges the doc-id slices of every query (no merge launch of its own behind it): that time is included. */
int32_t nidx_gpu_bm25_last_kernel_ms(const nidx_gpu_bm25_index_t *index, float *ms_out);

/* tantivy Bm25Weight pieces, exposed for the host query layer and for tests. */
float nidx_gpu_bm25_idf(uint64_t doc_freq, uint64_t doc_count);
uint32_t nidx_gpu_fieldnorm_from_id(uint8_t id);
uint8_t nidx_gpu_fieldnorm_to_id(uint32_t fieldnorm);

/* =====================================================================================
 * Shard merge — replaces src/searcher/shard_merge.rs:197-414: host merges, device merges for whole batches, and the
 * multi-GPU exchange (an RCCL all-gather inside the library) that feeds them.
 * ===================================================================================== */

/* merge_vector_responses (shard_merge.rs:332-348): kmerge_by(a.score >= b.score).take(limit).
 * lists[i] = scores of shard i (sorted desc), ids[i] = their payloads. Returns count in *n_out. */
int32_t nidx_gpu_merge_vector(const float *const *scores, const uint64_t *const *ids, const uint32_t *lens,
                              uint32_t n_lists, uint32_t limit, float *out_score, uint64_t *out_id,
                              uint32_t *out_list, uint32_t *n_out);
/* The same merge for a whole batch with everything in HBM (what each GPU runs after the RCCL
 * all-gather of the per-shard top-k): d_scores/d_ids [n_lists][n_queries][k], d_counts
 * [n_lists][n_queries]; outputs [n_queries][limit] / [n_queries].  Asynchronous on `stream`. */
int32_t nidx_gpu_merge_vector_device(const float *d_scores, const uint64_t *d_ids, const uint32_t *d_counts,
                                     uint32_t n_lists, uint32_t n_queries, uint32_t k, uint32_t limit,
                                     float *d_out_score, uint64_t *d_out_id, uint32_t *d_out_count, void *stream);
/* sort_documents_fn / sort_paragraphs_fn (shard_merge.rs:211-234,289-312): a before b iff
 * bm25 greater (total_cmp), else shard_id greater (bytes), else docaddr smaller. */
int32_t nidx_gpu_merge_bm25(const float *const *scores, const uint64_t *const *docaddrs, const uint32_t *lens,
                            const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists,
                            uint32_t limit, float *out_score, uint64_t *out_docaddr, uint32_t *out_list,
                            uint32_t *n_out);

/* The BM25 merges for a whole batch with everything in HBM: d_scores [n_lists][n_queries][k] f32, d_docaddrs [..] u64,
 * d_order_values [..] i64 or NULL (the sort value of every hit when the request orders by a date field — seconds * 10^9 + nanos,
 * or the fast value itself), d_counts [n_lists][n_queries].  `order`: NIDX_MERGE_ORDER_SCORE = sort_documents_fn /
 * sort_paragraphs_fn with SortExpr::Score (bm25 greater by total_cmp, else shard id greater as bytes, else docaddr smaller);
 * NIDX_MERGE_ORDER_VALUE_DESC / _ASC = SortExpr::Date descending / ascending (shard_merge.rs:236-250,314-329: strictly greater /
 * smaller value first; ties keep kmerge's heap order, the same as nidx_gpu_merge_bm25's).  shard_ids (host): the id of every list's
 * shard.  Outputs [n_queries][limit] (each may be NULL) / [n_queries]; d_out_list = the list every hit came from. */
enum { NIDX_MERGE_ORDER_SCORE = 0, NIDX_MERGE_ORDER_VALUE_DESC = 1, NIDX_MERGE_ORDER_VALUE_ASC = 2 };
int32_t nidx_gpu_merge_bm25_device(const float *d_scores, const uint64_t *d_docaddrs, const int64_t *d_order_values, const uint32_t *d_counts,
                                   const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists, uint32_t n_queries,
                                   uint32_t k, uint32_t limit, int32_t order, float *d_out_score, uint64_t *d_out_docaddr,
                                   int64_t *d_out_order_value, uint32_t *d_out_list, uint32_t *d_out_count, void *stream);
/* The two host merges for a whole batch (arrays laid out like the device forms, in host memory; order as above). */
int32_t nidx_gpu_merge_vector_batch(const float *scores, const uint64_t *ids, const uint32_t *counts, uint32_t n_lists, uint32_t n_queries,
                                    uint32_t k, uint32_t limit, float *out_score, uint64_t *out_id, uint32_t *out_count);
int32_t nidx_gpu_merge_bm25_batch(const float *scores, const uint64_t *docaddrs, const int64_t *order_values, const uint32_t *counts,
                                  const uint8_t *const *shard_ids, const uint32_t *shard_id_lens, uint32_t n_lists, uint32_t n_queries,
                                  uint32_t k, uint32_t limit, int32_t order, float *out_score, uint64_t *out_docaddr,
                                  int64_t *out_order_value, uint32_t *out_list, uint32_t *out_count);

/* merge_facets (shard_merge.rs:380-414): the facet counts of several shards summed per (group, tag).  The reference returns a
 * HashMap (iteration order unspecified); the output here is sorted by (group, tag) bytes.  Output entries point into the
 * inputs' strings.  *n_out receives the number of distinct (group, tag) pairs even when it exceeds `capacity`. */
typedef struct {
    const uint8_t *group;   /* the facet root, e.g. "/l" */
    uint32_t group_len;
    const uint8_t *tag;     /* FacetResult.tag, e.g. "/l/mylabel" */
    uint32_t tag_len;
    int32_t total;          /* FacetResult.total */
} nidx_gpu_facet_count_t;
int32_t nidx_gpu_merge_facets(const nidx_gpu_facet_count_t *const *shard_facets, const uint32_t *shard_lens, uint32_t n_shards,
                              nidx_gpu_facet_count_t *out, uint32_t capacity, uint32_t *n_out);

/* ---- the multi-GPU exchange: one index shard per GPU, one process per GPU, RCCL over xGMI ------------------------------------
 * Replaces the gRPC scatter / gather around merge_search (shard_merge.rs:54-99; src/searcher/shard_search.rs) inside one node:
 * every rank searches its own shard, then ONE ncclAllGather moves every rank's per-query top-k (a packed block: scores | ids |
 * [sort values] | counts — 120 KiB per rank at 1 024 queries x 10 hits) and every rank runs the reference's k-way merge on the
 * device.  librccl is bound at the first call (dlopen); no torch, no host staging.
 *   rank 0:    nidx_gpu_shard_comm_unique_id(id)            (ncclGetUniqueId; the host ships the 128 bytes to the other ranks)
 *   all ranks: nidx_gpu_set_device(local gpu); nidx_gpu_shard_comm_init(id, rank, world, this shard's id, ..)   (collective)
 *   per batch: nidx_gpu_shard_exchange_merge_vector / _bm25  (collective: same order on every rank; asynchronous on `stream`,
 *              inputs and outputs are device arrays of this rank; the merged lists are identical on every rank)
 * d_out_rank = the rank (= shard) every merged hit came from.  At most 64 ranks. */
#define NIDX_SHARD_COMM_ID_BYTES 128
typedef struct nidx_gpu_shard_comm nidx_gpu_shard_comm_t;
int32_t nidx_gpu_shard_comm_unique_id(uint8_t *id_out /* [NIDX_SHARD_COMM_ID_BYTES] */);
/* The same communicator over a second transport, for processes of ONE node that share a GPU (a test box has one): an id that
 * names a POSIX shared-memory segment; nidx_gpu_shard_comm_init with it gathers through that segment (device -> segment, barrier,
 * segment -> device) instead of ncclAllGather.  Packing, gather offsets, shard order and the merge kernels are the same code as
 * over RCCL; every exchange first checks that all ranks brought the same block shape (over RCCL: NIDX_GPU_SHARD_COMM_CHECK=1).
 * Not a product path: xGMI does not carry it. */
int32_t nidx_gpu_shard_comm_unique_id_shm(uint8_t *id_out /* [NIDX_SHARD_COMM_ID_BYTES] */);
int32_t nidx_gpu_shard_comm_init(const uint8_t *unique_id, int32_t rank, int32_t world, const uint8_t *shard_id, uint32_t shard_id_len,
                                 nidx_gpu_shard_comm_t **comm_out);
void nidx_gpu_shard_comm_destroy(nidx_gpu_shard_comm_t *comm);
int32_t nidx_gpu_shard_exchange_merge_vector(nidx_gpu_shard_comm_t *comm, const float *d_scores, const uint64_t *d_ids, const uint32_t *d_counts,
                                             uint32_t n_queries, uint32_t k, uint32_t limit, float *d_out_score, uint64_t *d_out_id,
                                             uint32_t *d_out_count, void *stream);
int32_t nidx_gpu_shard_exchange_merge_bm25(nidx_gpu_shard_comm_t *comm, const float *d_scores, const uint64_t *d_docaddrs,
                                           const int64_t *d_order_values, const uint32_t *d_counts, uint32_t n_queries, uint32_t k, uint32_t limit,
                                           int32_t order, float *d_out_score, uint64_t *d_out_docaddr, int64_t *d_out_order_value,
                                           uint32_t *d_out_rank, uint32_t *d_out_count, void *stream);

/* Rank fusion of the ranked lists a hybrid request gets back (nucliadb/src/nucliadb/search/search/rank_fusion.py; the
 * reference does this per request in Python).  Batched host code, no device work.
 * ReciprocalRankFusion._fuse + RankFusionAlgorithm.fuse (:60-181): for every query, walk the lists in the order given
 * (list 0 first), hit by hit in rank order (a list that carries scores is first ranked by them, descending, with a stable sort,
 * like _fuse's sorted(); a list without scores is taken as ranked); a hit's id is first seen => it enters the result with score
 * weight / (k + rank) — evaluated as (1.0 / (k + rank)) * weight in f64, like the Python expression — else the term is
 * added to its score; then a stable sort by score descending (ties keep first-seen order) and the first `window` hits.
 * A query with exactly one non-empty list returns that list unchanged with its own scores (fuse(): "one non-empty
 * source => its hits unchanged"), which is why lists may carry scores. */
typedef struct {
    const uint64_t *ids;     /* [n_queries][stride], ranked best first */
    const float *scores;     /* [n_queries][stride] or NULL (only read for the single-source case; NULL => 0) */
    const uint32_t *counts;  /* [n_queries] valid entries per row */
    uint32_t stride;
    double weight;           /* ReciprocalRankFusion.weights[source] (default_weight when absent) */
} nidx_gpu_ranked_list_t;
int32_t nidx_gpu_rank_fusion_rrf(const nidx_gpu_ranked_list_t *lists, uint32_t n_lists, uint32_t n_queries, double k,
                                 uint32_t window, uint64_t *out_ids, double *out_scores, uint32_t *out_counts);
/* WeightedCombSum._fuse + RankFusionAlgorithm.fuse (rank_fusion.py:184-254): the lists in the order given, their hits in the order
 * given (no re-ranking); a hit's id is first seen => it enters with score (f64)score * weight, else the term is added; then the
 * same stable sort by score descending, the first `window` hits, and the same single-source shortcut.  Every list carries scores;
 * `weight` = WeightedCombSum.weights[source]. */
int32_t nidx_gpu_rank_fusion_wcombsum(const nidx_gpu_ranked_list_t *lists, uint32_t n_lists, uint32_t n_queries, uint32_t window,
                                      uint64_t *out_ids, double *out_scores, uint32_t *out_counts);

#ifdef __cplusplus
}
#endif
#endif /* NIDX_GPU_H */
