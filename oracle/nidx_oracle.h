/*
 * nidx_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic on nucliadb's nidx search hot path
 * (nidx_vector HNSW / brute-force k-NN, tantivy-style BM25 scoring, shard merge),
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker for the HIP implementation under nucliadb_amd/.
 *
 * Every function cites the reference file:line (relative to /root/reference) that
 * it follows.  The reference is Rust + two crates.io libraries that are NOT in
 * /root/reference (simsimd 6.5.16, tantivy 0.26.1; nidx/Cargo.lock:4552-4555,
 * 4894-4897) and cannot be compiled here (no cargo/rustc, no network), so the
 * library arithmetic is restated from its published algorithm.
 *
 * PARITY PINNING: the oracle is pinned against every golden the reference's own
 * tests hold for this path (tests/test_oracle_golden.py).  Those
 * goldens pin ordering, loose score thresholds, recall floors, the HNSW disk
 * byte format and the shard-merge comparators.  They do NOT pin an exact cosine
 * bit pattern, any BM25 score value, or HNSW graph identity: for those the
 * oracle DEFINES the golden value ("parity unpinned" — stated in DESIGN.md).
 */
#ifndef NIDX_ORACLE_H
#define NIDX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- similarity (nidx_vector/src/config.rs:32-37: only Dot and Cosine exist) ---- */
enum { ORC_SIM_DOT = 0, ORC_SIM_COSINE = 1 };

/* Summation orders for the f32 inner products.  SimSIMD dispatches on the host
 * CPU at run time, so the reference's own bit pattern is machine dependent; the
 * orders below are the ones we restate / define:
 *   SERIAL     SimSIMD's portable macro: f32 accumulators, mul then add, i ascending
 *   SERIAL_FMA one fmaf chain, i ascending  (== v_mfma_f32_* numerics on gfx950)
 *   HASWELL    8 f32 FMA lanes, lanes reduced in f64, scalar tail (AVX2 kernel shape)
 *   WAVE64     the canonical order of the HIP kernels: 64 lanes x float4, 6-step xor butterfly
 */
enum { ORC_ORDER_SERIAL = 0, ORC_ORDER_SERIAL_FMA = 1, ORC_ORDER_HASWELL = 2, ORC_ORDER_WAVE64 = 3 };

void orc_sums(const float *x, const float *y, size_t n, int order, float *ab, float *xx, float *yy);
float orc_dot(const float *x, const float *y, size_t n, int order);
float orc_cosine(const float *x, const float *y, size_t n, int order);
float orc_cosine_from_sums(float ab, float xx, float yy);
float orc_similarity(const float *x, const float *y, size_t n, int similarity, int order);
void orc_normalize(const float *in, float *out, size_t n);

/* f32 total order used by every sort on the path (Rust f32::total_cmp). */
int orc_total_cmp(float a, float b);

/* ---- segment: brute force, cost model, HNSW ---- */
typedef struct orc_hnsw orc_hnsw;

typedef struct {
    const float *vectors;      /* [n_vectors][dim] contiguous f32 */
    uint32_t n_vectors;
    uint32_t dim;
    int similarity;            /* ORC_SIM_* */
    int order;                 /* ORC_ORDER_* */
    const uint32_t *vec_paragraph;   /* [n_vectors] paragraph addr of each vector (NULL => identity) */
    uint32_t n_paragraphs;
    const uint32_t *para_first_vec;  /* [n_paragraphs] (NULL => identity, 1 vector each) */
    const uint32_t *para_num_vec;    /* [n_paragraphs] */
    const uint64_t *alive;     /* bitset over paragraph addrs, NULL => all alive */
    const orc_hnsw *graph;     /* may be NULL (brute force only) */
    const uint8_t *quantized;  /* vectors.quant: [n_vectors][dim/8 + 8] RaBitQ records, or NULL.  When set,
                                * searches take the RaBitQ branches (has_quantized && !DISABLE_RABITQ_SEARCH,
                                * segment.rs:506-513) */
    uint32_t ef_search;        /* 0 = the reference's constant EF_SEARCH = 30 (hnsw/params.rs:46); another value only for the
                                * bench's iso-recall leg (the flat graph at the ef whose recall matches the segmented regime) */
    uint32_t ef_upper;         /* 0 = 1: the reference's greedy descent, k = 1 on the layers above 0 (hnsw/search.rs:318-324); another
                                * value keeps that many results per upper layer, all of them entry points of the next one — the
                                * product's "ef_upper" tunable, restated here so that its results are checked too */
} orc_segment;

typedef struct {
    uint64_t distance_evals;   /* similarity() calls */
    uint64_t expansions;       /* nodes whose edge list was walked */
    uint64_t edges_read;       /* u32 edge words read */
} orc_stats;

int orc_use_hnsw(size_t total_nodes, size_t matching_nodes, size_t top_k, int has_rabitq);

/* Returns number of results (<= k). filter: bitset over paragraph addrs or NULL (=> seg->alive). */
int orc_brute_force_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                           size_t k, float min_score, uint32_t *out_vec, float *out_score);
int orc_hnsw_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                    size_t k, float min_score, int with_duplicates, int multi_vector,
                    uint32_t *out_vec, float *out_score, orc_stats *stats);
/* OpenSegment::_search: routes with orc_use_hnsw. method_out: 0 none, 1 hnsw, 2 brute force */
int orc_segment_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                       size_t k, float min_score, int with_duplicates,
                       uint32_t *out_vec, float *out_score, int *method_out);

/* maxsim_similarity (nidx_vector/src/multivector.rs:33-46): sum over the query vectors of the best similarity among the
 * document's vectors, 0.0 when none is positive */
float orc_maxsim(const float *query_vectors, size_t n_query, const float *doc_vectors, size_t n_doc, size_t dim, int similarity, int order);

/* layer_search exposed for unit tests: returns count, results sorted (score desc, addr asc) */
int orc_layer_search(const orc_segment *seg, const float *query, int query_is_stored, uint32_t stored_addr,
                     int layer, size_t k, const uint32_t *entry_points, size_t n_ep,
                     uint32_t *out_vec, float *out_score, orc_stats *stats);

/* ---- RaBitQ (nidx_vector/src/vector_types/rabitq.rs) ---- */
#define ORC_RABITQ_RERANKING_FACTOR 100 /* rabitq.rs:34 */
#define ORC_RABITQ_RERANKING_LIMIT 2000 /* rabitq.rs:36 */
size_t orc_rabitq_encoded_len(size_t dim);                     /* EncodedVector::encoded_len (:70-73) */
/* EncodedVector::encode (:75-106): [f32 dot_quant_original][u32 sum_bits][dim/64 x u64 sign bits].
 * `order` = summation order of the SimSIMD dot inside (ORC_ORDER_*). */
void orc_rabitq_encode(const float *v, size_t dim, int order, uint8_t *out);
typedef struct {
    float low, delta, root_dim;
    uint32_t sum_quantized;
    uint32_t n_words;          /* dim / 64 */
    uint64_t *planes;          /* [4][n_words]: bit p of every 4-bit code */
} orc_rabitq_query;
/* QueryVector::from_vector (:124-157).  Free with orc_rabitq_query_free. */
void orc_rabitq_query_init(orc_rabitq_query *q, const float *v, size_t dim);
void orc_rabitq_query_free(orc_rabitq_query *q);
/* QueryVector::similarity (:202-218) -> (estimate, error) */
void orc_rabitq_similarity(const orc_rabitq_query *q, const uint8_t *encoded, float *estimate, float *error);
/* rerank_top (:221-244): candidates in the given order; returns count (<= k), sorted score desc */
size_t orc_rabitq_rerank_top(const orc_segment *seg, const float *query, float min_score, const uint32_t *cand,
                             const float *cand_upper_bound, size_t n_cand, size_t k, uint32_t *out_vec, float *out_score,
                             uint64_t *n_evaluated);

/* ---- HNSW graph ---- */
orc_hnsw *orc_hnsw_new(void);
void orc_hnsw_free(orc_hnsw *g);
/* Sequential (deterministic) build over seg->vectors, insertion order 0..n-1. */
orc_hnsw *orc_hnsw_build(const orc_segment *seg, uint64_t level_seed);
/* Level draw of HnswBuilder::get_random_layer for the first `n` nodes */
void orc_hnsw_levels(uint64_t seed, uint32_t n, uint32_t *levels_out);
uint32_t orc_hnsw_num_layers(const orc_hnsw *g);
uint32_t orc_hnsw_num_nodes(const orc_hnsw *g);
void orc_hnsw_entry_point(const orc_hnsw *g, uint32_t *node, uint32_t *layer);
void orc_hnsw_set_entry_point(orc_hnsw *g, uint32_t node, uint32_t layer);
/* edges of `node` at `layer`: returns degree, copies up to cap entries */
uint32_t orc_hnsw_edges(const orc_hnsw *g, uint32_t layer, uint32_t node, uint32_t *out, float *w_out, uint32_t cap);
int orc_hnsw_contains(const orc_hnsw *g, uint32_t layer, uint32_t node);
void orc_hnsw_add_node(orc_hnsw *g, uint32_t node, uint32_t top_layer);
void orc_hnsw_set_edges(orc_hnsw *g, uint32_t layer, uint32_t node, const uint32_t *edges, const float *w, uint32_t n);
void orc_hnsw_update_entry_point(orc_hnsw *g);
void orc_hnsw_fix_broken_graph(orc_hnsw *g);
/* DiskHnswV2 byte format. Returns bytes needed; writes when buffers are large enough. */
size_t orc_hnsw_serialize_v2(const orc_hnsw *g, uint32_t num_nodes, uint8_t *graph, size_t graph_cap,
                             float *edges, size_t edges_cap, size_t *n_edges_out);
orc_hnsw *orc_hnsw_deserialize_v2(const uint8_t *graph, size_t len, const float *edges, size_t n_edges);
/* Reads edges straight from the serialized image (DiskHnswV2::get_out_edges) */
uint32_t orc_disk_v2_edges(const uint8_t *graph, size_t len, uint32_t layer, uint32_t node, uint32_t *out, uint32_t cap);
void orc_disk_v2_entry_point(const uint8_t *graph, size_t len, uint32_t *node, uint32_t *layer);

/* ---- Searcher: multi segment merge (Fssc) ---- */
typedef struct {
    uint64_t paragraph_key;  /* stands for the paragraph id string (equal string <=> equal key) */
    float score;
    uint32_t segment;
    uint32_t vector;
} orc_scored_paragraph;

/* segments searched in array order; para_keys[s][paragraph addr] = key id.
 * Returns count, out sorted by score desc. */
int orc_searcher_search(const orc_segment *segs, const uint64_t *const *para_keys, size_t n_segs,
                        const float *query, const uint64_t *const *filters, size_t k, float min_score,
                        int with_duplicates, int normalize_query, orc_scored_paragraph *out);

/* ---- BM25 (tantivy 0.26.1 restated) ---- */
uint32_t orc_fieldnorm_from_id(uint8_t id);
uint8_t orc_fieldnorm_to_id(uint32_t fieldnorm);
float orc_bm25_idf(uint64_t doc_freq, uint64_t doc_count);
void orc_bm25_tf_cache(float average_fieldnorm, float cache[256]);

typedef struct {
    uint32_t n_docs;               /* max_doc: includes deleted docs (statistics do not shrink) */
    uint64_t total_num_tokens;
    uint32_t n_terms;
    const uint64_t *term_offsets;  /* [n_terms+1] into doc_ids/tfs */
    const uint32_t *doc_ids;       /* ascending per term */
    const uint32_t *tfs;
    const uint8_t *fieldnorm_ids;  /* [n_docs] */
    const uint64_t *alive;         /* bitset or NULL */
    /* positions of every posting (IndexRecordOption::WithFreqsAndPositions): posting i owns
     * positions[pos_offsets[i] .. pos_offsets[i+1]) ascending; NULL = no positions (no phrase clauses) */
    const uint64_t *pos_offsets;
    const uint32_t *positions;
} orc_bm25_index;

/* ORC_OCCUR_SHOULD_GROUP: a Should clause of a nested Must(BooleanQuery[Should..]) — the shape of
 * nidx_paragraph's keyword query under its Must filters (search_query.rs:185-243): scored like Should,
 * and a document must match at least one clause of the group. */
/* ORC_OCCUR_SHOULD_GROUP + g (g < 8): the same for a g-th nested group — the prefilter's BooleanQuery[Should SetQuery(field_uuid),
 * Should SetQuery(uuid)] and an Or formula are further required groups beside the keyword group (search_query.rs:88-143,218-223). */
enum { ORC_OCCUR_SHOULD = 0, ORC_OCCUR_MUST = 1, ORC_OCCUR_MUST_NOT = 2, ORC_OCCUR_SHOULD_GROUP = 3 };
enum { ORC_TF_FREQ = 0, ORC_TF_BASIC = 1, ORC_CONST_SCORE = 2 };

typedef struct {
    uint32_t term;     /* term id in the index */
    int occur;         /* ORC_OCCUR_* */
    int mode;          /* ORC_TF_FREQ: BM25 with stored tf; ORC_TF_BASIC: tf == 1; ORC_CONST_SCORE: score = boost */
    float boost;
    /* term set (FuzzyTermQuery / AutomatonWeight, nidx_paragraph/src/fuzzy_query.rs:55-125): when n_set_terms > 0
     * the clause matches the UNION of these terms' posting lists, each document once, ConstScorer(boost);
     * `term` and `mode` are ignored */
    const uint32_t *set_terms;
    uint32_t n_set_terms;
    /* set_complement != 0: the clause matches every document NOT in the union — parse_excluded's
     * BooleanQuery[Must AllQuery, MustNot term] (query_parser/keyword_parser.rs:93-105), scored AllQuery's 1.0 * boost */
    int set_complement;
    /* set_phrase != 0: PhraseQuery(set_terms) with slop 0 (tantivy PhraseWeight / PhraseScorer restated): a document
     * matches when the terms occur at consecutive positions in this order; tf = the number of such occurrences,
     * weight = (sum of the terms' idf, in term order) * (1 + K1) * boost (Bm25Weight::for_terms) */
    int set_phrase;
} orc_bm25_clause;

typedef struct {
    int has_after;
    float score;
    int tie_break;     /* 0 keep, 1 keep-after(docaddr), 2 other */
    uint64_t docaddr;
} orc_search_after;

/* Returns number of hits written (<= k); total_out = matching alive docs (Count collector). */
int orc_bm25_search(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                    size_t k, const orc_search_after *after, uint32_t segment_ord,
                    uint64_t *out_docaddr, float *out_score, uint64_t *total_out);

/* The collectors around the same scoring (nidx_text/src/reader.rs:367-451): TopDocs ordered by a fast field
 * (order_values[doc], TopDocs::order_by_fast_field; NULL = by score), and the set of matching alive documents as a
 * bitset (match_bits_out, NULL = not wanted) from which FacetCollector counts are taken.  out_order_value: NULL
 * or [k]. */
int orc_bm25_search_ex(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                       size_t k, const orc_search_after *after, uint32_t segment_ord,
                       const int64_t *order_values, int order_desc, uint64_t *match_bits_out,
                       uint64_t *out_docaddr, float *out_score, int64_t *out_order_value, uint64_t *total_out);

/* ---- a searcher over SEVERAL segments of one index (tantivy Searcher::search; nidx_tantivy/src/index_reader.rs:39-74 opens every
 * segment of the index under one searcher, nidx_text/src/reader.rs:433-435 and nidx_paragraph/src/reader.rs:290-292,330-332 search
 * it once).  Term ids name the same term in every segment (a term a segment does not hold has an empty list there).
 * Bm25Weight's statistics are searcher-wide: total_num_docs = sum of max_doc, total_num_tokens = sum over segments,
 * doc_freq(term) = sum of the segments' doc_freq; every segment is scored with that weight and fieldnorm cache, its own
 * fieldnorms and alive set; DocAddress = (segment_ord << 32) | doc with segment_ord = the position in `segs`; the
 * search-after cursor is tested per document against that DocAddress (nidx_paragraph/src/reader.rs:350-390);
 * merge_fruits keeps the k best by (score desc, DocAddress asc) or (fast value, DocAddress asc); Count adds up.
 * order_values / match_bits_out: NULL or one pointer per segment. ---- */
void orc_bm25_searcher_stats(const orc_bm25_index *segs, size_t n_segs, uint64_t *total_docs, uint64_t *total_tokens, float *avg_fieldnorm);
uint64_t orc_bm25_searcher_doc_freq(const orc_bm25_index *segs, size_t n_segs, uint32_t term);
int orc_bm25_searcher_search_ex(const orc_bm25_index *segs, size_t n_segs, const orc_bm25_clause *clauses, size_t n_clauses,
                                size_t k, const orc_search_after *after, const int64_t *const *order_values, int order_desc,
                                uint64_t *const *match_bits_out, uint64_t *out_docaddr, float *out_score,
                                int64_t *out_order_value, uint64_t *total_out);
/* the same for plain term clauses ordered by score, document at a time (no dense accumulator: benchmark-scale checks) */
int orc_bm25_searcher_search_daat(const orc_bm25_index *segs, size_t n_segs, const orc_bm25_clause *clauses, size_t n_clauses,
                                  size_t k, const orc_search_after *after, uint64_t *out_docaddr, float *out_score, uint64_t *total_out);

/* Levenshtein automaton of FuzzyTermQuery (levenshtein_automata 0.2.1, nidx/Cargo.lock:2313; restated):
 * distance in unicode scalar values with a transposition of two adjacent characters costing one
 * (`transposition_cost_one = true`, fuzzy_parser.rs:73); prefix = build_prefix_dfa: SOME prefix of `term` is
 * within `distance` of `query`.  Returns 1 when `term` is accepted. */
int orc_fuzzy_match(const uint8_t *query, size_t qlen, const uint8_t *term, size_t tlen, int distance, int prefix);
/* every accepted term of a dictionary (term i = bytes[offsets[i] .. offsets[i+1])); returns the count */
size_t orc_fuzzy_terms(const uint8_t *bytes, const uint64_t *offsets, size_t n_terms, const uint8_t *query, size_t qlen,
                       int distance, int prefix, uint32_t *out, size_t cap);

/* TextReaderService::prefilter (nidx_text/src/reader.rs:148-180) restated document-at-a-time: the boolean expression
 * of filter_to_query (nidx_text/src/search_query.rs:156-223) in postfix form, evaluated per live document.  Leaves:
 * ORC_FILTER_TERMS a..b (the document is in the posting list of ANY of lists[a..b), a TermQuery / union of TermQuery),
 * ORC_FILTER_RANGE a (RangeQuery with INCLUSIVE bounds on a date fast field, search_query.rs:30-49; no bound at all
 * = AllQuery), ORC_FILTER_PHRASE a (PhraseQuery of a tokenised keyword, query_io.rs:22-42), ALL and NONE; NOT is
 * BooleanQuery[Must AllQuery, MustNot e].  out_docs: matching live doc ids ascending (at most cap written); returns
 * the full count; *live_out = searcher.num_docs(). */
enum { ORC_FILTER_TERMS = 0, ORC_FILTER_AND = 1, ORC_FILTER_OR = 2, ORC_FILTER_NOT = 3, ORC_FILTER_ALL = 4, ORC_FILTER_NONE = 5,
       ORC_FILTER_RANGE = 6, ORC_FILTER_PHRASE = 7 };
typedef struct { int op; uint32_t a, b; } orc_filter_op;
typedef struct { uint32_t field; int has_since, has_until; int64_t since, until; } orc_date_range;
size_t orc_bm25_prefilter(const orc_bm25_index *idx, const orc_filter_op *ops, size_t n_ops, const uint32_t *lists,
                          const orc_date_range *ranges, const int64_t *created, const int64_t *modified,
                          const uint32_t *phrase_terms, const uint64_t *phrase_offsets, uint32_t *out_docs, size_t cap,
                          uint64_t *live_out);

/* Same results, document-at-a-time (no dense accumulator): the CPU baseline of bench.py. */
int orc_bm25_search_daat(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                         size_t k, const orc_search_after *after, uint32_t segment_ord,
                         uint64_t *out_docaddr, float *out_score, uint64_t *total_out);

/* ---- shard merge (nidx/src/searcher/shard_merge.rs) ---- */
typedef struct {
    float score;
    uint64_t id;       /* opaque payload */
} orc_vec_hit;
size_t orc_merge_vector(const orc_vec_hit *const *lists, const size_t *lens, size_t n_lists, size_t limit, orc_vec_hit *out);

typedef struct {
    float bm25;
    uint64_t docaddr;
    const uint8_t *shard_id;
    size_t shard_id_len;
    uint64_t payload;
} orc_bm25_hit;
size_t orc_merge_bm25(const orc_bm25_hit *const *lists, const size_t *lens, size_t n_lists, size_t limit, orc_bm25_hit *out);

/* ---- vector-index filter formulas, document at a time (ParagraphInvertedIndexes::filter, inverted_index/paragraph.rs:124-184;
 * see the definition for the program format) ---- */
int orc_field_key(const uint8_t *id, size_t len, uint8_t *out, size_t cap);   /* FieldKey::from_field_id (utils.rs:80-111) */
long orc_formula_filter(const uint8_t *key_bytes, const uint64_t *key_offsets, size_t n_paragraphs,
                        const uint8_t *label_bytes, const uint64_t *label_offsets, const uint64_t *para_label_offsets, const uint32_t *para_labels,
                        const uint8_t *atom_bytes, const uint64_t *atom_offsets, const orc_filter_op *ops, size_t n_ops, int resource_prefix,
                        uint64_t *out);

/* ---- batch runners (cpu_baseline leg of bench.py, benchmark-scale parity checks): the single-query functions above,
 * one query per work item on `threads` POSIX threads (one blocking thread per request, src/searcher/shard_search.rs:139-153).
 * out_* are [n_queries][k]; stats NULL or [n_queries]. */
void orc_hnsw_search_batch(const orc_segment *seg, const float *queries, size_t n_queries, size_t k, float min_score,
                           int with_duplicates, unsigned threads, uint32_t *out_vec, float *out_score, uint32_t *out_count,
                           orc_stats *stats);
void orc_brute_force_batch(const orc_segment *seg, const float *queries, size_t n_queries, size_t k, float min_score,
                           unsigned threads, uint32_t *out_vec, float *out_score, uint32_t *out_count);
void orc_searcher_search_batch(const orc_segment *segs, const uint64_t *const *para_keys, size_t n_segs, const float *queries,
                               size_t n_queries, size_t k, float min_score, int with_duplicates, unsigned threads,
                               orc_scored_paragraph *out, uint32_t *out_count);
void orc_bm25_search_daat_batch(const orc_bm25_index *idx, const orc_bm25_clause *clauses, const uint64_t *clause_offsets,
                                size_t n_queries, size_t k, unsigned threads, uint64_t *out_docaddr, float *out_score,
                                uint32_t *out_count, uint64_t *out_total);

#ifdef __cplusplus
}
#endif
#endif
