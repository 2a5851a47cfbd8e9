// hnsw_build_host.cpp — host driver of the device HNSW build (nidx_gpu_vector_build_hnsw).
//
// HnswBuilder::initialize_graph (hnsw/build.rs:50-55): every node draws its top layer from
// SmallRng::seed_from_u64(seed) and is added, edge-less, to layers 0..=top; the entry point is a
// node of the top layer (ram_hnsw.rs:99-107 takes the first key of an FxHashMap, i.e. an arbitrary
// one; we take the lowest address).  Then the nodes are inserted in address order in batches
// (hnsw_build.hip) whose size grows with the graph: a batch never exceeds 1/16 of the nodes
// already inserted, so at most ~6 % of a node's potential neighbours are invisible to it — the
// same kind of race the reference's rayon workers have.
#include <stdlib.h>

#include <algorithm>
#include <chrono>

#include "host_common.h"
#include "vector_index.h"

namespace nidx {

// A batch is at most 1/16 of the graph it searches (the ramp below) and at most this many nodes: 0.3 % of a 10 M graph.  Measured at 10 M x 768
// clustered (scripts/r5_build2.sh): 8 192 -> 12.27 s of kernels, recall@10 0.9949; 16 384 -> 11.62 s, 0.9965; 32 768 -> 11.26 s, 0.9965 (the tail of
// a launch and the sort / launch gaps are paid per batch).
static const uint32_t kMaxBatch = 32768;

int32_t VectorIndex::build_hnsw(uint32_t si, uint64_t level_seed, bool extend) {
    std::lock_guard<std::mutex> lock(mu);
    NIDX_HIP(hipSetDevice(device));
    VectorSegment &seg = segs[si];
    const uint32_t n = seg.n;
    if (extend && !seg.base_graph) return fail(NIDX_ERR_INVALID_ARGUMENT, "segment %u was not opened with a partial hnsw graph", si);
    const uint32_t n0 = extend ? seg.base_nodes : 0;  // nodes [0, n0) keep their graph (segment.rs:143-153)
    const HostGraph *base = extend ? seg.base_graph.get() : nullptr;
    seg.has_graph = false;
    if (n == 0) return NIDX_OK;
    if (n >= (1u << 30)) return fail(NIDX_ERR_UNSUPPORTED, "segments of 2^30 or more vectors are not supported");
    // initialize_graph(skip_nodes = n0, total = n): a fresh RNG draws the levels of the NEW nodes only
    std::vector<uint8_t> drawn, levels(n);
    draw_levels(level_seed, n - n0, drawn);
    for (uint32_t i = 0; i < n0; i++) {
        // the request keys of the build carry 4 bits of layer and the records of a node are sized from its level: a reused
        // graph with a node above layer 15 (P ~ 30^-15 for a real one: a corrupt or crafted image) is refused, not clamped
        if (base->top_layer[i] > 15)
            return fail(NIDX_ERR_UNSUPPORTED, "hnsw.graph: node %u lives on layer %u; graphs above layer 15 cannot be extended", i, (unsigned)base->top_layer[i]);
        levels[i] = base->top_layer[i];
    }
    for (uint32_t i = n0; i < n; i++) levels[i] = drawn[i - n0];
    uint32_t max_level = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (levels[i] > 15) levels[i] = 15;  // 4 bits of layer in the request key; P(level > 15) ~ 30^-15
        max_level = std::max<uint32_t>(max_level, levels[i]);
    }
    HostGraph hg;
    hg.n = n;
    hg.ep_layer = max_level;
    hg.ep_node = 0;
    if (base && base->ep_layer >= max_level) {
        // update_entry_point (ram_hnsw.rs:99-107) only moves the entry point when a higher layer appeared
        hg.ep_node = base->ep_node;
        hg.ep_layer = base->ep_layer;
    } else {
        for (uint32_t i = n0; i < n; i++)
            if (levels[i] == max_level) { hg.ep_node = i; break; }
        if (!base)
            for (uint32_t i = 0; i < n; i++)
                if (levels[i] == max_level) { hg.ep_node = i; break; }
    }
    hg.top_layer = levels;
    hg.upper_base.assign(n, 0xffffffffu);
    uint32_t n_upper = 0;
    for (uint32_t i = 0; i < n; i++)
        if (levels[i] > 0) {
            hg.upper_base[i] = n_upper;
            n_upper += levels[i];
        }
    // edge-less graph in HBM
    NIDX_HIP(seg.g_l0.alloc((size_t)n * NIDX_L0_STRIDE * 4));
    NIDX_HIP(seg.g_l0_w.alloc((size_t)n * NIDX_L0_STRIDE * 4));
    NIDX_HIP(seg.g_upper_base.alloc((size_t)n * 4));
    NIDX_HIP(seg.g_upper.alloc((size_t)std::max<uint32_t>(n_upper, 1) * NIDX_UP_STRIDE * 4));
    NIDX_HIP(seg.g_upper_w.alloc((size_t)std::max<uint32_t>(n_upper, 1) * NIDX_UP_STRIDE * 4));
    NIDX_HIP(hipMemsetAsync(seg.g_l0.p, 0, seg.g_l0.bytes, stream));
    NIDX_HIP(hipMemsetAsync(seg.g_l0_w.p, 0, seg.g_l0_w.bytes, stream));
    NIDX_HIP(hipMemsetAsync(seg.g_upper.p, 0, seg.g_upper.bytes, stream));
    NIDX_HIP(hipMemsetAsync(seg.g_upper_w.p, 0, seg.g_upper_w.bytes, stream));
    NIDX_HIP(hipMemcpyAsync(seg.g_upper_base.p, hg.upper_base.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    if (base && n0 > 0) {
        // the reused graph: layer-0 records keep their place; upper records move to the new numbering
        NIDX_HIP(hipMemcpyAsync(seg.g_l0.p, base->l0.data(), (size_t)n0 * NIDX_L0_STRIDE * 4, hipMemcpyHostToDevice, stream));
        NIDX_HIP(hipMemcpyAsync(seg.g_l0_w.p, base->l0_w.data(), (size_t)n0 * NIDX_L0_STRIDE * 4, hipMemcpyHostToDevice, stream));
        std::vector<uint32_t> up(hg.upper_base.empty() ? 0 : (size_t)n_upper * NIDX_UP_STRIDE, 0u);
        std::vector<float> upw(up.size(), 0.f);
        for (uint32_t i = 0; i < n0; i++)
            for (uint32_t l = 1; l <= base->top_layer[i] && base->upper_base[i] != 0xffffffffu; l++) {
                const size_t src = ((size_t)base->upper_base[i] + (l - 1)) * NIDX_UP_STRIDE;
                const size_t dst = ((size_t)hg.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE;
                std::copy(base->upper.begin() + src, base->upper.begin() + src + NIDX_UP_STRIDE, up.begin() + dst);
                std::copy(base->upper_w.begin() + src, base->upper_w.begin() + src + NIDX_UP_STRIDE, upw.begin() + dst);
            }
        if (!up.empty()) {
            NIDX_HIP(hipMemcpyAsync(seg.g_upper.p, up.data(), up.size() * 4, hipMemcpyHostToDevice, stream));
            NIDX_HIP(hipMemcpyAsync(seg.g_upper_w.p, upw.data(), upw.size() * 4, hipMemcpyHostToDevice, stream));
        }
        NIDX_HIP(hipStreamSynchronize(stream));  // `up` / `upw` go out of scope
    }
    seg.ep_node = hg.ep_node;
    seg.ep_layer = hg.ep_layer;
    seg.top_layer = levels;

    // batch plan + per-node slot index inside its batch
    struct Batch { uint32_t start, size, n_slots; };
    std::vector<Batch> batches;
    std::vector<uint32_t> slot_base(n);
    uint32_t max_slots = 0;
    uint32_t max_batch = kMaxBatch;
    if (const char *e = getenv("NIDX_GPU_BUILD_MAX_BATCH")) max_batch = std::max<uint32_t>(32, (uint32_t)atoi(e));   // diagnostics
    for (uint32_t start = n0; start < n;) {
        // a batch's nodes do not see each other, so it stays a small fraction of the graph they search.
        // When extending, the appended rows may be a distribution of their own (another segment's
        // clusters): ramp on the number of NEW nodes already linked, not on the reused graph's size.
        uint32_t ramp = start / 16;
        if (extend) ramp = std::min<uint32_t>(ramp, std::max<uint32_t>(32, (start - n0) / 8));
        uint32_t size = std::min<uint32_t>(std::min<uint32_t>(max_batch, std::max<uint32_t>(1, ramp)), n - start);
        uint32_t slots = 0;
        for (uint32_t i = start; i < start + size; i++) {
            slot_base[i] = slots;
            slots += (uint32_t)levels[i] + 1;
        }
        batches.push_back(Batch{start, size, slots});
        max_slots = std::max(max_slots, slots);
        start += size;
    }
    DevBuf d_levels, d_slot_base, d_found, d_found_len, d_slot_node, d_slot_layer, d_req_key, d_req_key2, d_req_val,
        d_req_val2, d_tmp, d_flags;
    NIDX_HIP(d_levels.alloc(n));
    NIDX_HIP(d_slot_base.alloc((size_t)n * 4));
    NIDX_HIP(hipMemcpyAsync(d_levels.p, levels.data(), n, hipMemcpyHostToDevice, stream));
    NIDX_HIP(hipMemcpyAsync(d_slot_base.p, slot_base.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    NIDX_HIP(d_found.alloc((size_t)max_slots * NIDX_BUILD_FOUND_STRIDE * 8));
    NIDX_HIP(d_found_len.alloc((size_t)max_slots * 4));
    NIDX_HIP(d_slot_node.alloc((size_t)max_slots * 4));
    NIDX_HIP(d_slot_layer.alloc((size_t)max_slots * 4));
    const size_t max_req = (size_t)max_slots * NIDX_BUILD_REQ_STRIDE;
    NIDX_HIP(d_req_key.alloc(max_req * 8));
    NIDX_HIP(d_req_key2.alloc(max_req * 8));
    NIDX_HIP(d_req_val.alloc(max_req * 4));
    NIDX_HIP(d_req_val2.alloc(max_req * 4));
    size_t tmp_bytes = 0;
    NIDX_HIP(build_sort_tmp_bytes((uint32_t)max_req, &tmp_bytes));
    NIDX_HIP(d_tmp.alloc(std::max<size_t>(tmp_bytes, 16)));
    NIDX_HIP(d_flags.alloc(4));
    NIDX_HIP(hipMemsetAsync(d_flags.p, 0, 4, stream));

    BuildBatch b;
    b.seg = seg.seg_dev(cfg.similarity);
    b.seg.alive = nullptr;  // deleted paragraphs stay in the graph (the alive bitset is a search-time filter)
    b.g = seg.graph_dev();
    b.l0_w = seg.g_l0_w.as<float>();
    b.upper_w = seg.g_upper_w.as<float>();
    b.levels = d_levels.as<uint8_t>();
    b.found = d_found.as<uint64_t>();
    b.found_len = d_found_len.as<uint32_t>();
    b.slot_node = d_slot_node.as<uint32_t>();
    b.slot_layer = d_slot_layer.as<uint32_t>();
    b.req_key = d_req_key.as<uint64_t>();
    b.req_key_sorted = d_req_key2.as<uint64_t>();
    b.req_val = d_req_val.as<float>();
    b.req_val_sorted = d_req_val2.as<float>();
    b.sort_tmp = d_tmp.p;
    b.sort_tmp_bytes = tmp_bytes;
    b.vis_log2 = build_vis_log2;
    b.ef_upper = build_ef_upper;
    b.flags = d_flags.as<uint32_t>();
    DevBuf d_dbg;
    b.dbg = nullptr;
    if (getenv("NIDX_GPU_BUILD_DEBUG")) {
        NIDX_HIP(d_dbg.alloc(40));
        NIDX_HIP(hipMemsetAsync(d_dbg.p, 0, 40, stream));
        b.dbg = d_dbg.as<unsigned long long>();
    }
    // work counters of this build (nidx_gpu_vector_build_stats); NIDX_GPU_BUILD_STATS=0 builds without them
    DevBuf d_stats;
    b.stats = nullptr;
    {
        const char *e = getenv("NIDX_GPU_BUILD_STATS");
        if (!(e && atoi(e) == 0)) {
            NIDX_HIP(d_stats.alloc((size_t)NIDX_BUILD_STAT_LINES * 16 * 8));
            NIDX_HIP(hipMemsetAsync(d_stats.p, 0, d_stats.bytes, stream));
            b.stats = d_stats.as<unsigned long long>();
        }
    }
    const auto t_build0 = std::chrono::steady_clock::now();
    // The construction searches' visited table.  A search visits a few hundred to a few thousand nodes (ef = 100); the table used to be
    // 2^14 slots for every build — 64 KiB of LDS per workgroup, TWO construction searches per CU where the registers allow four.  Builds
    // now start at 2^12 (16 KiB: four per CU) unless the tunable says otherwise; the flag word is read back one batch behind the
    // launches (no stall), and the first batch that reports a table three quarters full moves the rest of the build to 2^14 — a search
    // that filled its table only ended early (the graph is approximate by nature, and the reference's own build races).
    PinBuf h_flags;
    NIDX_HIP(h_flags.reserve(2 * 4));
    uint32_t *hf = h_flags.as<uint32_t>();
    hf[0] = hf[1] = 0;
    hipEvent_t ev_flags[2] = {nullptr, nullptr};
    struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{ev_flags};
    for (int i = 0; i < 2; i++) NIDX_HIP(hipEventCreateWithFlags(&ev_flags[i], hipEventDisableTiming));
    const bool adaptive_vis = !build_vis_pinned && b.vis_log2 > 12;
    if (adaptive_vis) b.vis_log2 = 12;
    uint32_t n_escalated_at = 0;
    size_t bi = 0;
    for (const Batch &bt : batches) {
        b.batch_start = bt.start;
        b.batch_size = bt.size;
        b.slot_base = d_slot_base.as<uint32_t>() + bt.start;
        b.n_slots = bt.n_slots;
        NIDX_HIP(launch_build_batch(b, stream));
        if (adaptive_vis && b.vis_log2 == 12) {
            NIDX_HIP(hipMemcpyAsync(&hf[bi & 1], d_flags.p, 4, hipMemcpyDeviceToHost, stream));
            NIDX_HIP(hipEventRecord(ev_flags[bi & 1], stream));
            if (bi > 0 && hipEventQuery(ev_flags[(bi - 1) & 1]) == hipSuccess && (hf[(bi - 1) & 1] & NIDX_FLAG_VISITED_OVERFLOW)) {
                b.vis_log2 = build_vis_log2;
                n_escalated_at = bt.start + bt.size;   // the first node inserted with the large table
                // from here on the flag word describes the large table: the bit the small one raised is taken out (stream order: behind
                // the batches launched so far, before the next one) — a build that still carries it at the end overflowed 2^vis_log2
                NIDX_HIP(launch_flag_clear(d_flags.as<uint32_t>(), NIDX_FLAG_VISITED_OVERFLOW, stream));
            }
        }
        bi++;
    }
    last_build_escalated_at = n_escalated_at;
    uint32_t flags = 0;
    NIDX_HIP(hipMemcpyAsync(&flags, d_flags.p, 4, hipMemcpyDeviceToHost, stream));
    NIDX_HIP(hipStreamSynchronize(stream));
    if (b.dbg) {
        unsigned long long d5[5];
        NIDX_HIP(hipMemcpy(d5, b.dbg, 40, hipMemcpyDeviceToHost));
        fprintf(stderr, "[build dbg] reverse links: appends=%llu prunes=%llu targets=%llu cycles/prune=%llu prune share of wave cycles=%.2f\n", d5[0], d5[1], d5[4],
                d5[1] ? d5[2] / d5[1] : 0, d5[3] ? (double)d5[2] / (double)d5[3] : 0.0);
    }
    // a visited-table overflow only ends one construction search early (the graph is approximate by
    // nature); a candidate-pool overflow cannot happen below 412 exact ties and is reported
    if (flags & NIDX_FLAG_POOL_INEXACT) return fail(NIDX_ERR_INEXACT, "HNSW build: candidate pool overflow");
    last_build_flags = flags;
    {
        std::vector<unsigned long long> lines((size_t)NIDX_BUILD_STAT_LINES * 16, 0ull);
        if (b.stats) NIDX_HIP(hipMemcpy(lines.data(), b.stats, lines.size() * 8, hipMemcpyDeviceToHost));
        for (int c = 0; c < 8; c++) last_build_stats[c] = 0;
        for (size_t l = 0; l < NIDX_BUILD_STAT_LINES; l++)
            for (int c = 0; c < 6; c++) last_build_stats[2 + c] += lines[l * 16 + c];
        last_build_stats[0] = n - n0;   // nodes inserted
        last_build_stats[1] = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_build0).count();   // batches, launch to completion
        if (!b.stats) last_build_stats[2] = ~0ull;   // built without counters
    }
    seg.has_graph = true;
    seg.base_graph.reset();
    seg.base_nodes = 0;
    return NIDX_OK;
}

}  // namespace nidx

using namespace nidx;

extern "C" int32_t nidx_gpu_vector_build_hnsw(nidx_gpu_vector_index_t *index, uint32_t segment, uint64_t level_seed) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    return idx->build_hnsw(segment, level_seed, false);
} NIDX_ABI_CATCH

extern "C" int32_t nidx_gpu_vector_build_stats(nidx_gpu_vector_index_t *index, uint64_t *stats_out) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || !stats_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    for (int c = 0; c < 8; c++) stats_out[c] = idx->last_build_stats[c];
    stats_out[8] = idx->last_build_escalated_at;
    stats_out[9] = idx->last_build_flags;
    return NIDX_OK;
} NIDX_ABI_CATCH

extern "C" int32_t nidx_gpu_vector_extend_hnsw(nidx_gpu_vector_index_t *index, uint32_t segment, uint64_t level_seed) try {
    VectorIndex *idx = reinterpret_cast<VectorIndex *>(index);
    if (!idx || segment >= idx->segs.size()) return fail(NIDX_ERR_INVALID_ARGUMENT, "bad index/segment");
    return idx->build_hnsw(segment, level_seed, true);
} NIDX_ABI_CATCH
