// vector_index.h — the opaque nidx_gpu_vector_index_t (one process-local, device-resident Searcher).
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "hnsw_graph.h"
#include "host_common.h"
#include "kernels.h"

namespace nidx {

bool use_hnsw(uint64_t total_nodes, uint64_t matching_nodes, uint64_t top_k, bool has_rabitq);
void normalize_row(const float *in, float *out, uint32_t d);
// serving.cpp: host query rows -> pinned staging (normalised / zero padded), shared with a few helper threads for large batches
void stage_query_rows(const float *src, float *dst, uint32_t nq, uint32_t d, uint32_t dp, bool normalize);
void set_stage_threads(int32_t n);

// One OpenSegment in HBM (nidx_vector/src/segment.rs:288-357 Retriever + graph).
struct VectorSegment {
    uint32_t n = 0, dim = 0, dp = 0, n_paragraphs = 0;
    DevBuf vectors;      // [n][dp] f32, zero padded
    DevBuf norm2;        // [n] f32, WAVE64-order |x|^2
    DevBuf norm2_serial; // [n] f32, SERIAL_FMA-order |x|^2 (filled on the first MFMA scan)
    DevBuf vectors16;    // tiled bf16 copy (vector_bf16.hip "Operand layout"), filled on the first bf16 scan
    uint32_t dp16 = 0;
    DevBuf para_of_vec;  // [n] u32 (absent when identity)
    DevBuf para_first, para_num;  // [n_paragraphs] u32: the contiguous vectors of every paragraph (absent when identity)
    uint32_t vmax = 1;   // most vectors owned by one paragraph (> 1 only with VectorCardinality::Multi)
    DevBuf alive;        // bitset over paragraph addrs (absent when all alive)
    bool identity_para = true, all_alive = true;
    std::vector<uint32_t> para_host;   // empty when identity
    std::vector<uint64_t> alive_host;  // always present
    uint64_t alive_count = 0;
    std::vector<uint64_t> key_ids;     // Fssc identity of each paragraph (optional)
    DevBuf key_ids_dev;                // the same in HBM (the device-side Fssc)
    // label / field-key posting lists for device-side filter formulas (optional)
    DevBuf f_offsets, f_ids;
    DevBuf f_key_bytes, f_key_offsets;   // the posting lists' keys, sorted bytewise (what label.fst / field.fst resolve)
    uint32_t f_n_keys = 0;
    uint32_t f_n_lists = 0;
    uint64_t f_n_ids = 0;
    // HNSW graph
    bool has_graph = false;
    DevBuf g_l0, g_upper_base, g_upper;  // kernels.h GraphDev geometry
    DevBuf g_l0_w, g_upper_w;            // edge weights (built graphs only)
    uint32_t ep_node = 0, ep_layer = 0;
    std::vector<uint8_t> top_layer;
    // vectors.quant: RaBitQ records [n][dim/8 + 8] (absent => has_quantized() == false)
    DevBuf quant;
    bool has_quant = false;
    // merge with graph reuse: the graph of the first base_nodes vectors, waiting for extend_hnsw
    uint32_t base_nodes = 0;
    std::unique_ptr<HostGraph> base_graph;

    int32_t upload_graph(const HostGraph &hg);
    GraphDev graph_dev() const;
    SegDev seg_dev(int similarity) const;
    uint64_t bytes() const;
};

struct Coalescer;
std::shared_ptr<Coalescer> make_coalescer();
struct Pipeline;
std::shared_ptr<Pipeline> make_pipeline();
double trace_slow_us();  // NIDX_GPU_TRACE_SLOW_US  // coalescer.cpp

struct VectorIndex {
    nidx_gpu_vector_config_t cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    VectorIndex() = default;
    VectorIndex(const VectorIndex &) = delete;
    // also the clean-up of an open that failed half way
    ~VectorIndex() {
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        if (scratch_event) (void)hipEventDestroy(scratch_event);
        pipe.reset();   // synchronises and destroys the slot streams before the segments they read go away
    }
    std::mutex mu;
    std::vector<VectorSegment> segs;
    // tunables
    int waves_per_query = 4;
    int eval_rows = 4;   // rows in flight per wave (HNSW distance phase); tuned on MI355X (profiles/r02_tune_hnsw.txt: 4 rows fit the 128-VGPR budget since the pipelined loop)
    int min_waves = 4;   // register budget class of the HNSW kernel (4 => <=128 VGPR, 16 waves per CU)
    bool shape_pinned = false;  // eval_rows / min_waves were set by the caller (tunable or environment): no per-batch choice
    // launch shape of the HNSW kernels for a batch: a batch that leaves most CUs with at most one workgroup (<= 256 queries) is
    // latency-bound, not occupancy-bound — four rows in flight per wave and the 256-VGPR budget measured 10-11 % faster there
    // (batch 1: 0.48 -> 0.43 ms, batch 64: 0.65 -> 0.57 ms at 1 M x 768); large batches keep 2 rows / 128 VGPRs (16 waves per CU)
    int rows_for(uint32_t nq) const { return (!shape_pinned && nq <= 256) ? 4 : eval_rows; }
    int waves_for(uint32_t nq) const { return (!shape_pinned && nq <= 256) ? 2 : (!shape_pinned && crowded_now()) ? 5 : min_waves; }
    // A large batch submitted while other batches are still on the device (serving.cpp sets this for the submitting thread): together
    // they oversubscribe the CUs' workgroup slots, and what counts then is how many walks a CU holds, not how fast one walk runs —
    // <= 96 VGPRs and a 2^12 visited table (16 KiB of LDS less) make it FIVE workgroups per CU instead of four.  Measured, 10 M x 768,
    // three batches of 1 024 in flight: 4.02 M queries/s against 3.58 M; one launch alone in that shape takes 0.52 ms against 0.36 ms,
    // which is why a batch that finds the device idle keeps the faster walk (scripts/r5_shape.sh).  Same hits either way; a table that
    // fills up flags its query, which is then re-run with the larger one as ever.
    static bool crowded_launch();
    static void set_crowded_launch(bool on);
    // tunable "launch_shape": 0 = as above (the pipeline decides per batch), 1 = every large batch takes the crowded shape, 2 = none does — for
    // a caller that keeps several batches in flight on streams of its own through nidx_gpu_vector_segment_search_device (which cannot see them)
    int launch_shape = 0;
    bool crowded_now() const { return launch_shape == 1 || (launch_shape == 0 && crowded_launch()); }
    bool vis_pinned = false;    // the tunable "vis_log2" / NIDX_GPU_VIS_LOG2 was set
    uint32_t vis_for(uint32_t nq) const { return (!vis_pinned && !shape_pinned && nq > 256 && crowded_now()) ? std::min<uint32_t>(default_vis_log2, 12u) : default_vis_log2; }
    uint32_t ef_search = 0;   // 0 = EF_SEARCH (hnsw/params.rs:46); tunable "ef_search"
    uint32_t ef_upper = 0;    // 0 = 1: the greedy descent of hnsw/search.rs:318-324; tunable "ef_upper"
    bool closest_prefetch = true;   // tunable "closest_prefetch" (measurement)
    bool serial_segments = false;   // tunable "serial_segments": the blocking search of a multi-segment index one segment at a time
    uint32_t default_vis_log2 = 13;
    uint32_t build_vis_log2 = 14;
    bool build_vis_pinned = false;   // the tunable / NIDX_GPU_BUILD_VIS_LOG2 was set: no adaptive start at 2^12 (hnsw_build_host.cpp)
    uint32_t build_ef_upper = 0;   // 0 = 1 (reference); tunable "build_ef_upper": a wider descent when inserting into very large flat graphs
    uint32_t last_build_flags = 0;
    uint32_t last_build_escalated_at = 0;   // first node inserted with the large visited table (0: the 2^12 table held for the whole build)
    uint64_t last_build_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // nidx_gpu_vector_build_stats
    // grow-only scratch, guarded by mu
    DevBuf scratch_fstack, scratch_flists, scratch_fcount, scratch_q16, scratch_cand_vec, scratch_cand_score, scratch_cand_count, scratch_multi_vec, scratch_multi_score, scratch_multi_count, scratch_qnorm, scratch_partial, scratch_queries, scratch_filter, scratch_out_vec, scratch_out_score, scratch_out_count,
        scratch_stats, scratch_rq, scratch_planes, scratch_vis, scratch_ties, scratch_entry_vec, scratch_entry_score, scratch_entry_count,
        scratch_dump_vec, scratch_dump_score, scratch_dump_count, scratch_spill_pool, scratch_spill_cmax, scratch_spill_vis, scratch_spill_ids, scratch_rowmask, scratch_floor, scratch_floor2, scratch_bf16_flags, scratch_bf16_counts;
    uint64_t scan_matching_hint = ~0ull;  // paragraphs passing the current filter when the caller knows (search_host), ~0 = unknown
    uint64_t spill_queries = 0;  // queries re-run by the exact fallback since open (tunable "spill_queries" reads it)
    bool rabitq_enabled(const VectorSegment &seg) const { return seg.has_quant && !(cfg.flags & NIDX_CONFIG_DISABLE_RABITQ_SEARCH); }
    int32_t quantize(uint32_t segment);

    int32_t segment_search_device(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score,
                                  bool with_duplicates, int method, const uint64_t *d_filter, uint32_t *d_out_vec,
                                  float *d_out_score, uint32_t *d_out_count, uint32_t *d_stats, uint32_t vis_log2,
                                  hipStream_t st, uint32_t *d_flag_word = nullptr);
    ScanArgs scan_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, const uint64_t *d_filter) const;
    bool scan_takes_tile_kernel(uint32_t s, uint32_t nq, uint32_t k, uint64_t matching) const;
    RabitqSearchArgs rabitq_hnsw_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, uint32_t *d_flag_word) const;
    HnswSearchArgs hnsw_args(uint32_t s, const float *d_queries, uint32_t nq, uint32_t shape_nq, uint32_t k, float min_score, bool with_duplicates,
                             const uint64_t *d_filter, uint32_t *d_out_vec, float *d_out_score, uint32_t *d_out_count, uint32_t *d_stats,
                             uint32_t vis_log2, uint32_t *d_flag_word) const;
    int32_t segment_search_device_scratch(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score,
                                          bool with_duplicates, int method, const uint64_t *d_filter, uint32_t *d_out_vec,
                                          float *d_out_score, uint32_t *d_out_count, uint32_t *d_stats, uint32_t vis_log2,
                                          hipStream_t st, uint32_t *d_flag_word);
    // The whole OpenSegment::search of one segment: launch, then — only when the launch's flag word says a bounded on-chip
    // structure overflowed — the 2^15 visited table and the HBM-resident closest_up_nodes for exactly the flagged queries.
    // d_out_block = [nq*k vec][nq*k score][nq count][1 flag word]; it is copied to `host_block` (pinned) in ONE transfer.
    int32_t segment_search_exact(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, bool with_duplicates,
                                 int method, const uint64_t *d_filter, uint32_t *d_out_block, uint32_t *host_block, hipStream_t st,
                                 uint32_t *n_retried);
    static size_t out_block_words(uint32_t nq, uint32_t k) { return (size_t)nq * k * 2 + nq + 1; }
    // scratch buffers are index-owned but the device entry point runs on the caller's stream and returns with the work in
    // flight: the next user of the scratch (any stream) first waits for this event
    hipEvent_t scratch_event = nullptr;
    bool scratch_event_recorded = false;
    int32_t scratch_acquire(hipStream_t st);
    int32_t scratch_release(hipStream_t st);
    DevBuf flag_word;        // [16] u32: [0] = flags raised by device-entry launches since the last nidx_gpu_vector_device_flags
    DevBuf scratch_out_block;
    PinBuf pin_in, pin_out, pin_flag;
    // exact fallback for the queries whose closest_up_nodes walk outgrew the LDS pool / visited table (hnsw_spill.hip)
    int32_t segment_spill_search(uint32_t s, const float *d_queries, uint32_t nq, uint32_t k, float min_score, bool with_duplicates,
                                 int method, const uint64_t *d_filter, uint32_t *d_out_vec, float *d_out_score, uint32_t *d_out_count,
                                 const std::vector<uint32_t> &flagged, hipStream_t st);
    int32_t rows_equal_host(uint32_t sa, uint32_t va, uint32_t sb, uint32_t vb, bool &eq);
    int32_t search_host(const float *queries, uint32_t nq, const nidx_gpu_vector_search_params_t &p,
                        const uint64_t *const *segment_filters, const nidx_gpu_filter_program_t *programs,
                        uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                        uint32_t *out_count, int32_t *out_method, uint64_t *out_matching);
    // Fssc over per-segment result rows (nullptr = segment not searched); shared by search_host and the pipeline's wait
    int32_t fssc_merge(uint32_t nq, const nidx_gpu_vector_search_params_t &p, const uint32_t *const *seg_vec, const float *const *seg_score,
                       const uint32_t *const *seg_count, uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector,
                       float *out_score, uint32_t *out_count);
    uint64_t popcount_filter(uint32_t s, const uint64_t *filt) const;   // |filt ∩ alive| of segment s
    // pipelined serving (serving.cpp): batches in flight on their own streams, results delivered to pinned host memory
    std::shared_ptr<Pipeline> pipe = make_pipeline();
    int32_t pipeline_submit(const float *queries, uint32_t nq, const nidx_gpu_vector_search_params_t &p,
                            const uint64_t *const *segment_filters, bool blocking, uint64_t *ticket_out);
    int32_t pipeline_wait(uint64_t ticket, uint32_t *out_segment, uint32_t *out_paragraph, uint32_t *out_vector, float *out_score,
                          uint32_t *out_count, uint32_t *n_retried_out);
    void pipeline_config(int32_t depth, int32_t walks = -1);
    // evaluates `prog` for segment s into scratch_filter (already intersected with alive); returns |filter ∩ alive|
    int32_t eval_filter_program(uint32_t s, const nidx_gpu_filter_program_t &prog, uint64_t &matching);
    int32_t build_hnsw(uint32_t segment, uint64_t level_seed, bool extend = false);
    // request coalescing for single-query callers (coalescer.cpp)
    std::shared_ptr<Coalescer> coalescer = make_coalescer();
    // staging for batches of up to nq_max queries with pages of k hits, taken once: pinning memory costs tens of ms, which a
    // serving loop must not meet the first time a larger batch than any before comes together
    int32_t reserve_search(uint32_t nq_max, uint32_t k);
    uint32_t reserved_nq = 0, reserved_k = 0;
    int32_t search_one(const float *query, const nidx_gpu_vector_search_params_t &p, uint32_t *out_segment,
                       uint32_t *out_paragraph, uint32_t *out_vector, float *out_score, uint32_t *out_count);
    void coalescer_stats(uint64_t &batches, uint64_t &queries);
    void coalescer_config(int32_t window_us, int32_t max_batch, int32_t in_flight);
    void coalescer_admission(int32_t max_callers, int32_t reject_when_full);
};

}  // namespace nidx
