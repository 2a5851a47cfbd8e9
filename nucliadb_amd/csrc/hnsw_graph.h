// hnsw_graph.h — host-side image of the fixed-stride HBM graph layout and its conversion from/to
// the reference's DiskHnswV2 byte format (nidx_vector/src/hnsw/disk/v2.rs:16-49).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "kernels.h"

namespace nidx {

struct HostGraph {
    uint32_t n = 0;
    uint32_t ep_node = 0, ep_layer = 0;
    std::vector<uint32_t> l0;          // [n][64]: deg, edges...
    std::vector<uint32_t> upper_base;  // [n]
    std::vector<uint32_t> upper;       // [n_upper][32]
    std::vector<uint8_t> top_layer;    // [n] highest layer the node owns a record for
    // optional edge weights in the same geometry (built graphs only)
    std::vector<float> l0_w, upper_w;

    uint32_t n_upper_records() const { return (uint32_t)(upper.size() / NIDX_UP_STRIDE); }
};

// Parses an hnsw.graph image for `n_nodes` nodes.  Returns 0 or a NIDX_ERR_* code (message in err).
int parse_disk_v2(const uint8_t *buf, uint64_t len, uint32_t n_nodes, HostGraph &out, std::string &err);
// Fills l0_w / upper_w from an hnsw.edges stream (weights in graph order, disk/v2.rs:46-49).
int attach_edge_weights(HostGraph &g, const uint8_t *graph_buf, uint64_t graph_len, const float *edges, uint64_t n_edges,
                        std::string &err);
// RAMHnsw::fix_broken_graph (ram_hnsw.rs:109-143): drop links to nodes that are not in the layer.
void fix_broken_graph(HostGraph &g);
// DiskHnswV2::serialize_into: returns the image; edges (weights) in graph order.
void serialize_disk_v2(const HostGraph &g, std::vector<uint8_t> &graph, std::vector<float> &edges);

// HnswBuilder::get_random_layer for nodes 0..n-1 (hnsw/build.rs:36-55,97-101): SmallRng::seed_from_u64
// (Xoshiro256++ seeded via SplitMix64), Uniform f64 in [0,1), round(-ln(u) / ln(M)).
void draw_levels(uint64_t seed, uint32_t n, std::vector<uint8_t> &levels);

}  // namespace nidx
