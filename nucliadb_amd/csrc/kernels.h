// kernels.h — host-visible launch interfaces of the gfx950 kernels (internal to libnidx_gpu).
#pragma once
#define NIDX_K_MAX 512   /* largest result page: nucliadb asks for max(top_k, rank-fusion window, reranker window) <= 500 */
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nidx {

// ---- exact scan (vector_scan.hip) ----
struct ScanArgs {
    const float *vectors;         // [n][dp]
    const float *norm2;           // [n] WAVE64-order |x|^2 (cosine) or nullptr
    uint32_t n, dp;
    const float *queries;         // [n_queries][dp], zero padded
    uint32_t n_queries;
    const uint64_t *alive;        // bitset over paragraph addrs or nullptr
    const uint64_t *filter;       // bitset over paragraph addrs or nullptr
    const uint32_t *para_of_vec;  // nullptr = identity
    int similarity;               // 0 dot, 1 cosine
    float min_score;
    uint32_t k;                   // <= 256
    uint32_t qt;                  // query tile (filled by launch_scan)
    uint64_t *partial;            // [n_queries][nblk][k] rank keys
    const uint64_t *row_mask;     // shared-row scan only: bit r = row r is scanned (launch_bf16_row_mask); nullptr otherwise
};
uint32_t scan_query_tile(uint32_t n_queries, uint32_t dp, uint32_t k);
uint32_t scan_num_blocks(uint32_t n);
hipError_t launch_scan(ScanArgs a, uint32_t nblk, hipStream_t s);
// the same scan for large batches (vector_scan_shared.hip): row stripes to use, 0 = not applicable (use launch_scan);
// `matching` = rows passing the filter.  Needs a.row_mask and norm2 padded to a multiple of 8 rows.
uint32_t scan_shared_stripes(uint32_t n, uint32_t n_queries, uint32_t dp, uint32_t k, uint64_t matching);
hipError_t launch_scan_shared(const ScanArgs &a, uint32_t stripes, hipStream_t s);
hipError_t launch_merge_topk(const uint64_t *partial, uint32_t n_queries, uint32_t lists_per_query, uint32_t k,
                             uint32_t *out_vec, float *out_score, uint32_t *out_count, hipStream_t s);
// the register-tile scans / merges of several segments in one launch each (table-driven: blockIdx.z / .y = segment)
struct ScanMergeTab {
    const uint64_t *partial;
    uint32_t *out_vec;
    float *out_score;
    uint32_t *out_count;
};
hipError_t launch_scan_segments(const ScanArgs *table, uint32_t n_seg, ScanArgs shape, uint32_t nblk, hipStream_t s);
hipError_t launch_merge_topk_segments(const ScanMergeTab *table, uint32_t n_seg, uint32_t n_queries, uint32_t lists_per_query, uint32_t k, hipStream_t s);
hipError_t launch_maxsim(const float *vectors, const float *norm2, uint32_t dp, int similarity, const float *queries,
                         const uint32_t *cand_qfirst, const uint32_t *cand_qnum, const uint32_t *cand_first, const uint32_t *cand_num,
                         uint32_t n_cand, float *out, hipStream_t s);
hipError_t launch_para_best(const uint32_t *in_vec, const float *in_score, const uint32_t *in_count, uint32_t n_queries, uint32_t k_in,
                            const uint32_t *para_of_vec, uint32_t k, uint32_t *out_vec, float *out_score, uint32_t *out_count,
                            hipStream_t s);
hipError_t launch_row_norms(const float *vectors, uint32_t n, uint32_t dp, float *norm2, hipStream_t s);
hipError_t launch_pair_similarity(const float *x, const float *y, uint32_t n, uint32_t dp, int similarity, float *out,
                                  hipStream_t s);

// ---- batched exact scan on the f32 matrix cores (vector_mfma.hip) ----
struct MfmaScanArgs {
    const float *vectors;   // [n][dp]
    const float *norm2;     // [n] SERIAL_FMA-order |x|^2 (cosine) or nullptr
    uint32_t n, dp;
    const float *queries;   // [n_queries][dp]
    const float *q_norm2;   // [n_queries] SERIAL_FMA-order |q|^2
    uint32_t n_queries;
    const uint64_t *alive, *filter;
    const uint32_t *para_of_vec;
    int similarity;
    float min_score;
    uint32_t k;             // <= NIDX_MFMA_KMAX
    uint64_t *partial;      // [n_queries][stripes][k]
};
#define NIDX_MFMA_KMAX 64
hipError_t launch_serial_norms(const float *rows, uint32_t n, uint32_t dp, float *out, hipStream_t s);
uint32_t mfma_scan_stripes(uint32_t n, uint32_t n_queries);
hipError_t launch_mfma_scan(const MfmaScanArgs &a, uint32_t stripes, hipStream_t s);

// ---- bf16-MFMA brute-force fallback with exact re-scoring (vector_bf16.hip) ----
#define NIDX_BF16_CAND 32  /* approximate candidates kept per query before the exact re-score */
struct Bf16ScanArgs {
    const unsigned short *vectors16;  // tiled bf16 blocks [ceil(n / 256)][dp16 / 16][256][16] (vector_bf16.hip "Operand layout")
    uint32_t n, dp16;
    const unsigned short *queries16;  // tiled the same way, [ceil(n_queries / 256)][dp16 / 16][256][16]
    uint32_t n_queries;
    const uint64_t *row_mask;         // [ceil(n / 256) * 4] bit r = row r is scanned (launch_bf16_row_mask)
    uint64_t *partial;                // [n_queries][stripes][NIDX_BF16_CAND]
    int debug;                        // diagnostics (env NIDX_GPU_BF16_DEBUG): 1 = no candidate admitted (GEMM time only)
    const float *floor_score;         // nullptr or [n_queries]: a score at least NIDX_BF16_CAND rows are known to reach (sample pass)
    const uint32_t *run_if = nullptr; // bf16_scan_kernel: nullptr or [query blocks of 256]: a block whose word is 0 is skipped
    uint32_t *overflow = nullptr;     // bf16_append_kernel: [query blocks of 256], ORed with 1 when a stripe of the block ran out of slots
    // bf16_append_kernel: round i of a workgroup = corpus tile blockIdx.x + i * gridDim.x
    uint32_t round_step = 1;          // scan every round_step-th round (a sample pass)
    uint32_t round_skip = 0;          // > 1: the full pass behind a sample pass of that round_step on the same grid: skip its rounds, go on from its slots
    const uint32_t *skip_unless = nullptr;  // with round_skip: [query blocks]: non-zero = that sample ran out of slots: empty the slots, scan every round
    uint32_t *cnt_inout = nullptr;    // [n_queries][stripes] candidates per slot group: written by a sample pass, read by the full pass behind it
};
struct RescoreArgs {
    const float *vectors;   // [n][dp] f32
    const float *queries;   // [n_queries][dp] f32
    uint32_t dp, n_queries;
    const uint32_t *cand_vec;    // [n_queries][n_cand_max]
    const uint32_t *cand_count;  // [n_queries]
    uint32_t n_cand_max;
    int similarity;
    float min_score;
    uint32_t k;             // <= 64
    uint32_t *out_vec;
    float *out_score;
    uint32_t *out_count;
};
// rows -> the tiled bf16 operand layout; norm2 != nullptr scales every row by 1 / sqrt(norm2[row]) (cosine)
hipError_t launch_to_bf16_tiled(const float *in, const float *norm2, uint32_t n, uint32_t dp, uint32_t dp16, unsigned short *out, hipStream_t s);
// floor[q] = the NIDX_BF16_CAND-th best score of query q's merged sample candidates (count < NIDX_BF16_CAND: -inf)
hipError_t launch_bf16_floor(const float *cand_score, const uint32_t *cand_count, uint32_t n_queries, const float *prev, const uint32_t *overflow,
                             float *floor, hipStream_t s);
// the list-free scan (needs a floor per query): candidates appended to the [query][stripe][NIDX_BF16_CAND] block, which the caller zeroes first
hipError_t launch_bf16_append(const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s);
hipError_t launch_bf16_row_mask(uint32_t n, const uint32_t *para_of_vec, const uint64_t *alive, const uint64_t *filter, uint64_t *out,
                                hipStream_t s);
uint32_t bf16_scan_stripes(uint32_t n, uint32_t n_queries);
hipError_t launch_bf16_scan(const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s);
hipError_t launch_rescore_select(const RescoreArgs &a, hipStream_t s);

// ---- HNSW graph in HBM ----
// layer 0: fixed 256-byte records [deg, e0..e59, pad x3]; upper layers: 128-byte records
// [deg, e0..e29, pad]; a node with top layer L >= 1 owns L consecutive upper records starting at
// upper_base[node] (record of layer l is upper_base[node] + l - 1).
#define NIDX_L0_STRIDE 64
#define NIDX_UP_STRIDE 32
#define NIDX_M 30          /* hnsw/params.rs:40 */
#define NIDX_M_MAX 30      /* hnsw/params.rs:37 */
#define NIDX_M_MAX0 60     /* hnsw/params.rs:34 */
#define NIDX_EF_CONSTRUCTION 100 /* hnsw/params.rs:43 */
#define NIDX_EF_SEARCH 30  /* hnsw/params.rs:46 */

struct GraphDev {
    uint32_t *l0;          // [n][64]
    uint32_t *upper_base;  // [n] or 0xffffffff
    uint32_t *upper;       // [n_upper][32]
    uint32_t ep_node, ep_layer;
    uint32_t n;
};

struct SegDev {
    const float *vectors;  // [n][dp]
    const float *norm2;    // [n]
    uint32_t n, dp, dim;
    const uint32_t *para_of_vec;  // nullptr = identity
    const uint64_t *alive;        // nullptr = all
    int similarity;
};

// per-query counters written by the search kernel
#define NIDX_STAT_EVALS 0
#define NIDX_STAT_EXPANSIONS 1
#define NIDX_STAT_VISITED 2
#define NIDX_STAT_FLAGS 3
#define NIDX_STAT_CYC_CTL 4    /* wave-0 cycles: pop + edge record + visited test */
#define NIDX_STAT_EDGE_HITS 5  /* expansions whose edge record had been fetched ahead (no round trip in front of the rows) */
#define NIDX_STAT_CYC_INS 6    /* cycles replaying the admission rule */
#define NIDX_STAT_CYC_TOTAL 7
#define NIDX_STAT_STRIDE 8
#define NIDX_FLAG_VISITED_OVERFLOW 1u
#define NIDX_FLAG_POOL_INEXACT 2u

struct HnswSearchArgs {
    SegDev seg;
    GraphDev g;
    const float *queries;   // [n_queries][dp]
    uint32_t n_queries;
    const uint64_t *filter; // nullptr or bitset over paragraph addrs
    uint32_t k;             // <= 256
    float min_score;
    int with_duplicates;
    uint32_t vis_log2;      // visited table = 1<<vis_log2 u32 slots in LDS
    uint32_t *out_vec;      // [n_queries][k]
    float *out_score;       // [n_queries][k]
    uint32_t *out_count;    // [n_queries]
    uint32_t *stats;        // nullptr or [n_queries][NIDX_STAT_STRIDE]
    int multi;              // VectorCardinality::Multi: one hit per paragraph (NodeFilter::paragraphs, search.rs:159-164)
    int eval_rows;          // rows in flight per wave in the distance phase: 2 or 4
    int min_waves;          // register budget: 2 (<=256 VGPR) or 4 (<=128 VGPR) waves per SIMD
    // entry mode (RaBitQ arm): skip the descent and the layer-0 search, run closest_up_nodes from these
    // re-ranked entry points (search.rs:354-375).  nullptr = the normal search.
    const uint32_t *entry_vec;    // [n_queries][k]
    const float *entry_score;     // [n_queries][k]
    const uint32_t *entry_count;  // [n_queries]
    // dump mode (spill path): stop after the layer-0 search and hand its ef results over as entry points for
    // hnsw_closest_spill_kernel; nullptr = the normal search.  Rows of NIDX_DUMP_STRIDE entries.
    uint32_t *dump_vec;
    float *dump_score;
    uint32_t *dump_count;
    // nullptr or [1]: every query that raises a NIDX_FLAG_* ORs it in here too (one atomic, only when a flag was raised), so
    // a caller that did not ask for per-query counters can tell with one word whether the launch needs the exact fallback
    uint32_t *flag_word = nullptr;
    // 0 = the reference's EF_SEARCH (hnsw/params.rs:46); the layer-0 search keeps ef = max(k, ef_search) results.  Only the
    // "ef_search" tunable sets it: a flat 10 M graph trades ef against the recall the reference gets from merging 50 segments
    uint32_t ef_search = 0;
    // 0 = 1, the reference's greedy descent (search.rs:318-324: k = 1 on the layers above 0).  Only the "ef_upper" tunable sets it: the
    // descent then keeps ef_upper results per upper layer and hands all of them to the next layer as entry points (<= 64).
    uint32_t ef_upper = 0;
    // 1 (default): closest_up_nodes requests the edge record of the best candidate left in the pool while an expansion's rows are
    // scored; 0 switches that off (measurement only: no result depends on it)
    int closest_prefetch = 1;
    // launch_hnsw_search only: nullptr, or n_table argument records in HBM — one per segment, every one with the launch shape
    // (dp, k, vis_log2, ef_search, ef_upper, n_queries, eval_rows, min_waves) of this record: ONE grid of n_queries x n_table
    // walks (hnsw_search_segments_kernel); the kernels never read these two fields
    const HnswSearchArgs *seg_table = nullptr;
    uint32_t n_table = 0;
};
#define NIDX_DUMP_STRIDE 512
hipError_t launch_hnsw_search(const HnswSearchArgs &a, int waves_per_query, hipStream_t s);

// Fssc on the device (fssc_device.hip): the merge of every segment's hits of a query (searcher.rs:149-199).
struct FsscSegDev {
    const float *vectors;               // [n][dp] rows (the `seen` set compares vector bytes)
    const uint32_t *para_of_vec;        // nullptr = identity
    const unsigned long long *key_ids;  // [n_paragraphs] paragraph identity across segments; nullptr = (segment, paragraph)
    const uint32_t *vec, *count;        // this segment's result rows [nq][k] and [nq]; count == nullptr: the segment was not searched
    const float *score;
    uint32_t dp, pad;
};
struct FsscArgs {
    const FsscSegDev *segs;   // [n_segs] in HBM, index order
    uint32_t n_segs, nq, k, dim;
    int with_duplicates;
    uint32_t *offered;        // with_duplicates == 0: scratch [nq][offered_stride][3] (score bits, segment, vector)
    uint32_t offered_stride;  // >= the most candidates one query can be offered (searched segments x k)
    uint32_t *out_seg, *out_para, *out_vec;   // [nq][k]
    float *out_score;
    uint32_t *out_count;      // [nq]
};
hipError_t launch_fssc_merge(const FsscArgs &a, hipStream_t s);

// closest_up_nodes with the candidate pool and the visited set in HBM (hnsw_spill.hip): the exact fallback for the
// queries whose walk outgrew the LDS structures of hnsw_search_kernel.
struct HnswSpillArgs {
    SegDev seg;
    GraphDev g;
    const float *queries;        // [*][dp], indexed by query id
    const uint32_t *query_ids;   // [n_queries] ids of the queries to run; outputs / entries are indexed by id
    uint32_t n_queries;
    const uint64_t *filter;
    uint32_t k;
    float min_score;
    int with_duplicates;
    int multi;
    const uint32_t *entry_vec;   // [*][entry_stride]
    const float *entry_score;
    const uint32_t *entry_count; // [*]
    uint32_t entry_stride;
    uint64_t *pool;              // [n_queries][pool_chunks * 64]
    uint64_t *chunk_max;         // [n_queries][pool_chunks], zeroed
    uint32_t *vis;               // [n_queries][vis_words], zeroed
    uint32_t pool_chunks, vis_words;
    uint32_t *out_vec;           // [*][k]
    float *out_score;
    uint32_t *out_count;
};
hipError_t launch_hnsw_closest_spill(const HnswSpillArgs &a, hipStream_t s);

// ---- RaBitQ (rabitq.hip) ----
struct RabitqQueryDev {   // QueryVector (rabitq.rs:109-122) with the similarity() constants folded
    float c_dot;          // 2.0 * delta / root_dim
    float two_low;        // 2.0 * low
    float c_sumq;         // delta * sum_quantized / root_dim
    float c_low;          // low * root_dim
    float root_dim, low, delta;
    uint32_t sum_quantized;
};
struct RabitqSearchArgs {
    SegDev seg;
    GraphDev g;               // hnsw only
    const uint8_t *quant;     // [n][rec_len] vectors.quant records
    uint32_t rec_len;         // dim / 8 + 8
    const float *queries;     // [n_queries][dp] raw queries (re-rank)
    const RabitqQueryDev *qd; // [n_queries]
    const uint64_t *planes;   // [n_queries][4][dim / 64]
    uint32_t n_queries;
    const uint64_t *filter;   // brute force only (the HNSW arm filters in closest_up_nodes)
    const uint32_t *para_first, *para_num;  // brute force, multi-vector paragraphs: the vectors of paragraph p (nullptr = one each)
    uint32_t n_paragraphs;
    uint32_t k;               // <= 256
    uint32_t ef;              // hnsw: min(k * 100, 2000) (search.rs:333-340)
    float min_score;
    uint32_t *visited;        // hnsw: [n_queries][vis_words] zeroed bitsets over vector addrs
    uint32_t vis_words;
    uint32_t *out_vec;        // [n_queries][k]   brute force: the hits; hnsw: the re-ranked entry points
    float *out_score;
    uint32_t *out_count;
    uint32_t *stats;          // nullptr or [n_queries][NIDX_STAT_STRIDE]: estimates, expansions, rows re-ranked, flags
    uint32_t *flag_word = nullptr;  // nullptr or [1]: OR of the flags any query raised (see HnswSearchArgs)
    uint32_t no_speculation = 0;    // two-wave walk, measurement: 1 = the fetcher never runs ahead of the controller
    unsigned long long *dbg = nullptr;   // two-wave walk, measurement: [16] cycle totals of the two waves (NIDX_GPU_RABITQ_DEBUG)
    uint32_t seen_log2 = 0;         // pipelined walk: log2 words of the LDS cache of known-visited ids, 0 = none (rabitq_seen_log2())
    // hnsw: [n_queries][tie_stride] — where a walk keeps the evicted candidates that still tie with its worst result once the 64 in
    // LDS are full (the reference's BinaryHeap is unbounded, hnsw/search.rs:252-299); rabitq_tie_stride(ef) entries per query always
    // suffice (see RqLayer).  Not initialised by the host.  nullptr: such a walk raises NIDX_FLAG_POOL_INEXACT instead.
    uint64_t *tie_spill = nullptr;
    uint32_t tie_stride = 0;
};
inline uint32_t rabitq_tie_stride(uint32_t ef) { return (ef + 63u) / 64u * 64u + 64u; }
bool rabitq_tie_spill_enabled();   // false with NIDX_GPU_RABITQ_TIE_SPILL=0 (tests: shows that a scenario does overflow the 64 ties in LDS)
hipError_t launch_rabitq_encode(const float *vectors, uint32_t n, uint32_t dp, uint32_t dim, uint8_t *out, hipStream_t s);
hipError_t launch_rabitq_query(const float *queries, uint32_t nq, uint32_t dp, uint32_t dim, RabitqQueryDev *qd,
                               uint64_t *planes, hipStream_t s);
hipError_t launch_rabitq_bf(const RabitqSearchArgs &a, hipStream_t s);
hipError_t launch_rabitq_hnsw(const RabitqSearchArgs &a, hipStream_t s);
// several segments' walks in one launch: `table` (device, n_table records agreeing in shape with `shape`); block b = query b % nq of record b / nq
bool rabitq_two_waves();   // true with NIDX_GPU_RABITQ_WAVES=2 in a `make EXPERIMENTS=1` library (the two-wave walk: measured slower, kept for comparison)
bool rabitq_has_experiments();
uint32_t rabitq_seen_log2();   // 9; NIDX_GPU_RABITQ_SEEN=0 / 8...13 (measurement)
hipError_t launch_rabitq_hnsw_segments(const RabitqSearchArgs *table, uint32_t n_table, const RabitqSearchArgs &shape, hipStream_t s);

// ---- HNSW build (hnsw_build.hip): one batch of concurrent inserts ----
#define NIDX_BUILD_FOUND_STRIDE 128
#define NIDX_BUILD_REQ_STRIDE 32
struct BuildBatch {
    SegDev seg;
    GraphDev g;
    float *l0_w, *upper_w;      // edge weights, same geometry as g.l0 / g.upper
    const uint8_t *levels;      // [n] top layer of each node
    uint32_t batch_start, batch_size;
    const uint32_t *slot_base;  // [batch_size] first (node, layer) slot of each node of the batch
    uint32_t n_slots;
    uint64_t *found;            // [n_slots][128]
    uint32_t *found_len, *slot_node, *slot_layer;  // [n_slots]
    uint64_t *req_key, *req_key_sorted;            // [n_slots*32]
    float *req_val, *req_val_sorted;               // [n_slots*32]
    void *sort_tmp;
    size_t sort_tmp_bytes;
    uint32_t vis_log2;
    uint32_t ef_upper = 0;      // results kept per layer above the node's top layer (0 = 1: the reference's greedy descent)
    uint32_t *flags;            // [1]
    unsigned long long *dbg;    // nullptr or [5] (NIDX_GPU_BUILD_DEBUG)
    unsigned long long *stats;  // nullptr or [NIDX_BUILD_STAT_LINES][16]: the build's work counters (hnsw_build.hip: BuildArgs::stats)
};
#define NIDX_BUILD_STAT_LINES 256
hipError_t launch_flag_clear(uint32_t *word, uint32_t bits, hipStream_t s);   // *word &= ~bits, in stream order
hipError_t build_sort_tmp_bytes(uint32_t max_req, size_t *bytes);
hipError_t launch_build_batch(const BuildBatch &b, hipStream_t s);

// ---- filter formulas on the device (filter.hip) ----
hipError_t launch_bitset_fill(uint64_t *out, uint32_t n_words, uint32_t n_bits, int ones, hipStream_t s);
hipError_t launch_bitset_scatter(const unsigned long long *list_offsets, const uint32_t *ids, const uint32_t *lists,
                                 uint32_t n_lists, uint32_t n_bits, uint64_t *out, hipStream_t s);
hipError_t launch_bitset_binop(uint64_t *a, const uint64_t *b, uint32_t n_words, int op, hipStream_t s);
hipError_t launch_bitset_not(uint64_t *a, uint32_t n_words, uint32_t n_bits, hipStream_t s);
// sorted key table -> [first, last) of the entries equal to / prefixed by each query (filter.hip)
hipError_t launch_key_range(const uint8_t *tbl, const unsigned long long *tbl_off, uint32_t n_keys, const uint8_t *qb, const unsigned long long *q_off,
                            const uint8_t *q_prefix, uint32_t n_q, uint32_t *first, uint32_t *last, hipStream_t s);
hipError_t launch_bitset_and_count(const uint64_t *a, const uint64_t *alive, uint64_t *out, uint32_t n_words,
                                   unsigned long long *count, hipStream_t s);

// ---- BM25 (bm25.hip) ----
#define BM25_MAX_CLAUSES 64
struct Bm25ClauseDev {
    uint32_t term;
    int occur;     // 0 should, 1 must, 2 must-not, 3 + g: should of required group g (g < 8)
    int mode;      // 0 stored tf, 1 tf == 1, 2 constant score, 3 pre-scored (the posting word is an f32: a materialised sub-query)
    float weight;  // idf * (1 + K1) * boost, or the constant score
};
struct Bm25AfterDev {  // same layout as nidx_gpu_bm25_search_after_t
    int has_after;
    float score;
    int tie_break;
    unsigned long long docaddr;
};
struct Bm25Work {  // one work item: query `query`, doc-id slice `slice` of `n_slices`; its clause records [clause_first, + n_clauses)
    uint32_t query, slice, n_slices, clause_first, n_clauses;
};
// bm25_union_kernel's clause table (one record per query clause, per segment): the list's first posting, its length, the
// Bm25Weight (or constant score) and occur | mode << 8 — everything the kernel needs of a clause without touching term_offsets
struct Bm25UClause {
    uint32_t b_lo, b_hi, len;
    float weight;
    uint32_t attr;
    uint32_t floor_bits;   // f32: a score at least k documents of the QUERY reach (the same value in every clause of a query; -inf = none): bm25_stream_kernel
    uint32_t pad1, pad2;
};
#define BM25_ITEM_THREADS 64      /* threads per work item (64 = one wave: no block barriers) */
#define BM25_SLICE_POSTINGS 2048  /* target postings per work item */
#define BM25_SLICE_CROWDED 8192   /* ... of a union query when other batches are resident (bm25_index.cpp: crowded_shape) */
#define BM25_MAX_SLICES 256
// bm25_stream_kernel's fused merge (k <= 64, every item of the launch a stream item): the LAST wave to finish among a group of eight slices of
// a query merges the group's lists, the last group to finish merges the groups' lists and writes what bm25_merge_kernel would have written —
// no second launch behind the scoring kernel (3.4 us of every 30 us batch in the pipelined bench; scripts/r6_ab_merge.sh).  A wave merges at
// most 7 + 31 lists whatever the query's length.  `done` holds the arrival counters ([n_queries][33]: the query's, then one per group); whoever
// completes a count resets it, so the array is zero between launches.
#define BM25_FUSE_GROUP 8u
#define BM25_FUSE_MAX_GROUPS 32u   /* BM25_MAX_SLICES / BM25_FUSE_GROUP */
struct Bm25FusedMerge {
    uint32_t *done = nullptr;       // nullptr: bm25_merge_kernel merges
    unsigned long long *g_key;      // [n_queries][32][k] the groups' lists
    uint32_t *g_count;              // [n_queries][32]
    unsigned long long *g_total, *g_postings;   // [n_queries][32]
    uint32_t *out_doc;              // the merged hits, exactly as Bm25MergeArgs
    float *out_score;
    uint32_t *out_count;
    unsigned long long *out_total, *out_postings;
    const uint32_t *seg_base = nullptr;
    uint32_t n_seg = 0;
    uint32_t *out_seg = nullptr;
    int ablate = 0;                 // measurement (NIDX_GPU_BM25_ABLATE_MERGE=3 with the fused merge): merge, write nothing
};
struct Bm25Args {
    const Bm25Work *work;
    uint32_t n_docs;
    const unsigned long long *term_offsets;
    const uint32_t *doc_ids;
    const uint32_t *tfs;              // posting words: tf | fieldnorm id << 24 (packed at open)
    const uint8_t *fieldnorm_ids;
    const uint64_t *alive;   // nullptr = all
    const float *tf_cache;   // [4][256]: K1*(1-B+B*fieldnorm/avg), then tf / (tf + that) for tf = 1, 2, 3
    const Bm25ClauseDev *clauses;
    const unsigned long long *clause_offsets;
    const Bm25AfterDev *after;  // nullptr or [n_queries]
    uint32_t k;                 // <= 256
    uint32_t segment_ord;
    unsigned long long *out_key;  // [n_work][k] rank keys, best first (score bits or order rank << 32 | ~doc)
    uint32_t *out_count;        // [n_work]
    unsigned long long *out_total;
    unsigned long long *out_postings;
    unsigned long long *dbg;  // nullptr, or 6 counters: cycles load/apply/fold/total, windows, work items
    // term sets (FuzzyTermQuery): clause.term = BM25_AUX_TERM | j reads the materialised union list j
    const unsigned long long *aux_offsets;  // [n_sets][2]: begin, end into aux_doc_ids
    const uint32_t *aux_doc_ids;            // ascending doc ids of each union / phrase
    const uint32_t *aux_tfs;                // phrase frequencies (parallel to aux_doc_ids; read for phrase lists only)
    // TopDocs::order_by_fast_field: per-doc dense rank of the fast value (nullptr = order by score)
    const uint32_t *order_key;
    int order_desc;
    // matching-document bitsets for the FacetCollector: query q writes slot match_slot[q] (-1 = none)
    uint32_t *match_bits;                   // [n_slots][match_words]
    const int *match_slot;                  // [n_queries] or nullptr
    uint32_t match_words;
    const Bm25UClause *uclauses;            // [n_clauses of the batch] (bm25_union_kernel), parallel to `clauses`
    Bm25FusedMerge fm;                      // bm25_stream_kernel only
};
#define BM25_AUX_TERM 0x80000000u
struct Bm25MergeArgs {  // per query: merge the key lists of its work items [item_first[q], item_first[q + 1])
    const uint32_t *item_first;             // [n_queries + 1]
    const unsigned long long *item_key;     // [n_work][k]
    const uint32_t *item_count;             // [n_work]
    const unsigned long long *item_total, *item_postings;  // [n_work]
    uint32_t k;
    uint32_t *out_doc;                      // [n_queries][k]
    float *out_score;                       // [n_queries][k] (meaningless when ordering by a fast field)
    uint32_t *out_count;
    unsigned long long *out_total, *out_postings;  // [n_queries]
    // an index of several segments resident as one (bm25_index.cpp: bm25_upload_concatenated): seg_base[n_seg] = first resident doc of
    // every opened segment; out_doc then is the doc id INSIDE its segment and out_seg [n_queries][k] the segment.  nullptr = one segment
    const uint32_t *seg_base = nullptr;
    uint32_t n_seg = 0;
    uint32_t *out_seg = nullptr;
    int ablate = 0;   // measurement (NIDX_GPU_BM25_ABLATE_MERGE = 2 / 3): return at once / merge but write nothing
};
hipError_t launch_bm25_merge(const Bm25MergeArgs &m, uint32_t n_queries, hipStream_t s);
#define BM25_FAST_CLAUSES 8   /* queries of at most this many clauses take bm25_fast_kernel */
#define BM25_LIST_PAD_BYTES 8192  /* slack behind the posting arrays: the kernels load whole 64-posting rows unconditionally */
hipError_t launch_bm25_search(const Bm25Args &a, const uint32_t *fast_items, uint32_t n_fast, const uint32_t *wide_items, uint32_t n_wide,
                              uint32_t max_clauses, hipStream_t s);
// term unions whose lists rarely meet (bm25_union.hip); extras = alive bitset / facet bitsets / order by a fast field / search-after
hipError_t launch_bm25_union(const Bm25Args &a, const uint32_t *items, uint32_t n_items, bool extras, hipStream_t s);
// the same queries term at a time (bm25_stream.hip): the default; launch_bm25_union stays selectable for comparison
hipError_t launch_bm25_stream(const Bm25Args &a, const uint32_t *items, uint32_t n_items, bool extras, hipStream_t s);

// ---- BM25 surroundings (bm25_aux.hip) ----
// FuzzyTermQuery's automaton over the whole term dictionary: flags[t] = 1 when term t is accepted
// (distance <= 1 in unicode scalar values, transposition = 1 edit; prefix: some prefix of the term)
hipError_t launch_fuzzy_match(const uint8_t *dict_bytes, const unsigned long long *dict_offsets, uint32_t n_terms,
                              const uint32_t *query_cp, uint32_t n_query_cp, int prefix, uint8_t *flags, hipStream_t s);
// set bits -> ascending ids (one block per bitset): out[out_offsets[b] ..), counts[b] = number written
hipError_t launch_bitset_compact(const uint64_t *bits, uint32_t n_words, uint32_t n_sets, const unsigned long long *out_offsets,
                                 uint32_t *out, uint32_t *counts, hipStream_t s);
// PhraseQuery: for every posting i of the driver term (the rarest of the phrase) tmp_tf[i] = number of matches of the phrase in
// that document (0 = the document does not match); then the matches are compacted in document order into (out_ids, out_tfs)[out_begin ..)
// and *out_count is set.  slop 0: the number of start positions at which all terms follow each other in order.  slop > 0: tantivy's
// PhraseScorer with slop (PhraseQuery::set_slop: a budget of position moves shared by all terms, in both directions) — the positions
// of term i shifted by n_terms - 1 - i, the running `left` list of (position, budget used) pairs intersected term after term;
// `left` lives in slop_left, one region of 2 words per position of terms[0].
struct PhraseDev {
    uint32_t terms[8];     // phrase terms in order
    uint32_t n_terms;      // <= 8
    uint32_t driver;       // index into terms[] of the term whose postings are walked
    uint32_t slop;
};
#define BM25_MAX_PHRASE_TERMS 8
hipError_t launch_phrase_match(const unsigned long long *term_offsets, const uint32_t *doc_ids, const unsigned long long *pos_offsets,
                               const uint32_t *positions, PhraseDev ph, uint32_t n_driver, uint32_t *tmp_tf, uint32_t *slop_left, hipStream_t s);
hipError_t launch_phrase_compact(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph, const uint32_t *tmp_tf,
                                 const uint8_t *fieldnorm_ids, unsigned long long out_begin, uint32_t *out_ids, uint32_t *out_tfs,
                                 uint32_t *out_count, hipStream_t s);
// A nested BooleanQuery (tantivy: a BooleanQuery inside a BooleanQuery — an AND inside an OR, a negated conjunction, a disjunction
// inside a conjunction inside a disjunction ..., a conjunction inside an `Or` formula): its matches are materialised as one more aux
// list whose posting word is the sub-score's f32 bits (clause mode 3 = pre-scored).  A leaf is a posting list of the dictionary or an
// aux list materialised before it — a term set, a phrase, or ANOTHER nested query (so trees of any depth are lists of lists).  The
// candidates are the postings of one leaf (the shortest Must leaf) or, for a query without a Must leaf, the union of the members of
// one required Should group / of all its Should leaves (scattered into a bitset and compacted); one lane per candidate probes every
// leaf by binary search.  A document matches when every Must leaf holds it, no MustNot leaf does, every required Should group has a
// member that does and — without Must leaves and groups — some Should leaf does; its score is the f32 sum of the scoring leaves that
// hold it, in leaf order.  Aux list a lies at aux_ids[aux_begin[a] .. + aux_counts[a]) — both read on the device, so a chain of
// nested queries is materialised without a host round trip between its levels.
#define BM25_MAX_SUBQUERY_LEAVES 32
#define BM25_SUB_DRIVER_UNION 0xffffffffu
struct SubqueryDev {
    uint32_t n;        // leaves
    uint32_t driver;   // index of the leaf whose postings are the candidates, or BM25_SUB_DRIVER_UNION (the union list)
    uint32_t src[BM25_MAX_SUBQUERY_LEAVES];    // term id, or BM25_AUX_TERM | aux list
    uint8_t occur[BM25_MAX_SUBQUERY_LEAVES];   // 0 should, 1 must, 2 must-not, 3 + g required Should group g
    uint8_t mode[BM25_MAX_SUBQUERY_LEAVES];    // 0 stored tf, 1 tf == 1, 2 constant score, 3 pre-scored (word = f32 bits)
    float weight[BM25_MAX_SUBQUERY_LEAVES];
};
struct SubqueryLists {   // what a leaf's `src` resolves against
    const unsigned long long *term_offsets;
    const uint32_t *doc_ids, *words;          // the segment's postings (word = tf | fieldnorm id << 24)
    const uint32_t *aux_ids, *aux_words;      // the aux lists materialised so far
    const unsigned long long *aux_begin;      // [n_aux] first entry of every aux list
    const uint32_t *aux_counts;               // [n_aux] entries of every aux list
    const uint32_t *union_ids, *union_count;  // the union candidates (driver == BM25_SUB_DRIVER_UNION)
};
// bits |= the documents of leaf `src` (one launch per member of the union)
hipError_t launch_subquery_scatter(const SubqueryLists &L, uint32_t src, unsigned long long upper_bound, uint64_t *bits, hipStream_t s);
// n_cand_max: a host-side upper bound of the candidates (the launch shape); the kernels read the exact number on the device
hipError_t launch_subquery_match(const SubqueryLists &L, const float *tf_cache, const SubqueryDev &sq, uint32_t n_cand_max, uint32_t *tmp_ok,
                                 uint32_t *tmp_score, hipStream_t s);
hipError_t launch_subquery_compact(const SubqueryLists &L, const SubqueryDev &sq, const uint32_t *tmp_ok, const uint32_t *tmp_score,
                                   unsigned long long out_begin, uint32_t *out_ids, uint32_t *out_words, uint32_t *out_count, hipStream_t s);
// resident posting word: tfs[i] = tf | fieldnorm_ids[doc_ids[i]] << 24 (in place); *flag: bit 0 = a tf >= 2^24, bit 1 = a doc id >= n_docs
hipError_t launch_bm25_pack_fieldnorm(const uint32_t *doc_ids, uint32_t *tfs, const uint8_t *fieldnorm_ids, unsigned long long n, uint32_t n_docs,
                                      uint32_t *flag, hipStream_t s);
// per-term score floors (bm25_aux.hip): out[t][j] = smallest fieldnorm id f with >= BM25_FLOOR_RANKS[j] of the first `cap` resident posting words of
// term t at fieldnorm id <= f, 255 = none
#define BM25_FLOOR_NR 18
#define BM25_FLOOR_RANKS {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512}
hipError_t launch_bm25_term_floors(const unsigned long long *term_offsets, const uint32_t *words, uint32_t n_terms, uint32_t cap, uint8_t *out,
                                   hipStream_t s);
// several segments -> one term-major resident layout (bm25_aux.hip): segment postings [0, n_post) to dst_start[t] + (j - seg_off[t])
hipError_t launch_bm25_concat_postings(const unsigned long long *seg_off, uint32_t n_terms, const unsigned long long *dst_start, const uint32_t *src_doc,
                                       const uint32_t *src_tf, unsigned long long n_post, uint32_t doc_base, uint32_t *dst_doc, uint32_t *dst_tf,
                                       hipStream_t s);
// FacetCollector: counts[p] += |postings(term[p]) ∩ match bitset slot[p]|
hipError_t launch_facet_count(const unsigned long long *term_offsets, const uint32_t *doc_ids, const uint32_t *pair_term,
                              const int *pair_slot, uint32_t n_pairs, const uint32_t *match_bits, uint32_t match_words,
                              unsigned long long *counts, hipStream_t s);

// Prefilter (nidx_text reader.rs:148-180): range over a fast field's dense ranks, phrase matches as a bitset, and the
// bitset -> ascending DocAddress list (block_scratch: ceil(n_words / 256) u32; *total = number of set bits; entries beyond
// out_cap are counted but not written).
hipError_t launch_rank_range_bits(const uint32_t *order_key, uint32_t n_docs, uint32_t rank_lo, uint32_t rank_hi, uint64_t *out,
                                  hipStream_t s);
hipError_t launch_phrase_bits(const unsigned long long *term_offsets, const uint32_t *doc_ids, PhraseDev ph, uint32_t n_driver,
                              const uint32_t *tmp_tf, uint64_t *bits, hipStream_t s);
hipError_t launch_bitset_to_docaddr(const uint64_t *bits, uint32_t n_words, uint32_t segment, uint32_t *block_scratch,
                                    unsigned long long *total, unsigned long long out_begin, unsigned long long out_cap, uint64_t *out,
                                    hipStream_t s);

}  // namespace nidx
