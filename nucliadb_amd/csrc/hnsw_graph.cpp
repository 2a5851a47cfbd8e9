// hnsw_graph.cpp — DiskHnswV2 <-> fixed-stride HBM graph image, and the level draw.
// Reference: nidx_vector/src/hnsw/disk/v2.rs:16-49 (format), :109-245 (writer / accessors);
// nidx_vector/src/hnsw/build.rs:36-55,97-101 and hnsw/params.rs:20-22 (levels).
#include "hnsw_graph.h"

#include <math.h>
#include <string.h>

#include "../../include/nidx_gpu.h"
#include "host_common.h"

namespace nidx {

static inline uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void wr32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)x);
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 24));
}

int parse_disk_v2(const uint8_t *buf, uint64_t len, uint32_t n_nodes, HostGraph &g, std::string &err) {
    g = HostGraph();
    g.n = n_nodes;
    g.l0.assign((size_t)n_nodes * NIDX_L0_STRIDE, 0u);
    g.upper_base.assign(n_nodes, 0xffffffffu);
    g.top_layer.assign(n_nodes, 0);
    if (n_nodes == 0 || len == 0) return NIDX_OK;  // serialize_into writes nothing for an empty graph
    // trailer: [node end offsets, reversed][ep layer][ep node]
    if (len < 8 + (uint64_t)n_nodes * 4) { err = "hnsw.graph shorter than its node index"; return NIDX_ERR_INVALID_GRAPH; }
    g.ep_node = rd32(buf + len - 4);
    g.ep_layer = rd32(buf + len - 8);
    if (g.ep_node >= n_nodes) { err = "hnsw.graph entry point out of range"; return NIDX_ERR_INVALID_GRAPH; }
    if (g.ep_layer >= 64) { err = "hnsw.graph entry point layer out of range"; return NIDX_ERR_INVALID_GRAPH; }
    const uint32_t n_layers = g.ep_layer + 1;  // the entry point lives on the top layer (ram_hnsw.rs:99-107)
    const uint64_t indexing_end = len - 8;
    const uint64_t nodes_limit = indexing_end - (uint64_t)n_nodes * 4;

    // pass 1: find each node's top layer to lay out the upper records
    std::vector<uint32_t> node_end(n_nodes);
    uint64_t n_upper = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint64_t pos = indexing_end - ((uint64_t)i + 1) * 4;
        uint32_t end = rd32(buf + pos);
        if (end > nodes_limit || end < n_layers * 4) { err = "hnsw.graph node offset out of range"; return NIDX_ERR_INVALID_GRAPH; }
        node_end[i] = end;
        uint32_t top = 0;
        for (uint32_t l = 0; l < n_layers; l++) {
            uint32_t off = rd32(buf + end - (l + 1) * 4);
            if (off > end || off < 4) { err = "hnsw.graph layer offset out of range"; return NIDX_ERR_INVALID_GRAPH; }
            uint32_t start = end - off;
            uint32_t deg = rd32(buf + start);
            if ((uint64_t)start + 4 + (uint64_t)deg * 4 > end) { err = "hnsw.graph edge list overruns its node"; return NIDX_ERR_INVALID_GRAPH; }
            if (deg > (l == 0 ? (uint32_t)NIDX_M_MAX0 : (uint32_t)NIDX_M_MAX)) {
                err = "hnsw.graph degree exceeds M_max";
                return NIDX_ERR_INVALID_GRAPH;
            }
            if (deg > 0) top = l;
        }
        if (i == g.ep_node && top < g.ep_layer) top = g.ep_layer;
        g.top_layer[i] = (uint8_t)top;
        if (top > 0) {
            g.upper_base[i] = (uint32_t)n_upper;
            n_upper += top;
            if (n_upper >= 0xffffffffull) { err = "hnsw.graph holds more upper-layer records than a u32 indexes"; return NIDX_ERR_INVALID_GRAPH; }
        }
    }
    g.upper.assign((size_t)n_upper * NIDX_UP_STRIDE, 0u);
    // pass 2: copy edge lists
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint32_t end = node_end[i];
        for (uint32_t l = 0; l <= g.top_layer[i]; l++) {
            uint32_t off = rd32(buf + end - (l + 1) * 4);
            uint32_t start = end - off;
            uint32_t deg = rd32(buf + start);
            uint32_t *rec = l == 0 ? &g.l0[(size_t)i * NIDX_L0_STRIDE]
                                   : &g.upper[((size_t)g.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE];
            rec[0] = deg;
            for (uint32_t e = 0; e < deg; e++) {
                uint32_t to = rd32(buf + start + 4 + e * 4);
                if (to >= n_nodes) { err = "hnsw.graph edge target out of range"; return NIDX_ERR_INVALID_GRAPH; }
                rec[1 + e] = to;
            }
        }
    }
    return NIDX_OK;
}

int attach_edge_weights(HostGraph &g, const uint8_t *buf, uint64_t len, const float *edges, uint64_t n_edges, std::string &err) {
    g.l0_w.assign(g.l0.size(), 0.f);
    g.upper_w.assign(g.upper.size(), 0.f);
    if (!edges || g.n == 0 || len == 0) return NIDX_OK;
    // the weights were written node by node, layer by layer, edge by edge: walk the image the same way
    const uint32_t n_layers = g.ep_layer + 1;
    const uint64_t indexing_end = len - 8;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < g.n; i++) {
        uint32_t end = rd32(buf + indexing_end - ((uint64_t)i + 1) * 4);
        for (uint32_t l = 0; l < n_layers; l++) {
            uint32_t off = rd32(buf + end - (l + 1) * 4);
            uint32_t deg = rd32(buf + (end - off));
            float *w = nullptr;
            if (l == 0) w = &g.l0_w[(size_t)i * NIDX_L0_STRIDE];
            else if (l <= g.top_layer[i] && g.upper_base[i] != 0xffffffffu)
                w = &g.upper_w[((size_t)g.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE];
            if (pos + deg > n_edges) { err = "hnsw.edges shorter than the graph"; return NIDX_ERR_INVALID_GRAPH; }
            for (uint32_t e = 0; e < deg; e++)
                if (w) w[1 + e] = edges[pos + e];
            pos += deg;
        }
    }
    return NIDX_OK;
}

void fix_broken_graph(HostGraph &g) {
    for (uint32_t i = 0; i < g.n; i++) {
        for (uint32_t l = 1; l <= g.top_layer[i]; l++) {
            if (g.upper_base[i] == 0xffffffffu) break;
            const size_t r = ((size_t)g.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE;
            uint32_t *rec = &g.upper[r];
            float *w = g.upper_w.empty() ? nullptr : &g.upper_w[r];
            uint32_t out = 0;
            for (uint32_t e = 0; e < rec[0]; e++) {
                uint32_t to = rec[1 + e];
                if (g.top_layer[to] >= l) {
                    rec[1 + out] = to;
                    if (w) w[1 + out] = w[1 + e];
                    out++;
                }
            }
            rec[0] = out;
        }
    }
    if (g.n && g.top_layer[g.ep_node] < g.ep_layer) {  // entry point not on its layer: take another node that is
        for (uint32_t i = 0; i < g.n; i++)
            if (g.top_layer[i] >= g.ep_layer) { g.ep_node = i; break; }
    }
}

void serialize_disk_v2(const HostGraph &g, std::vector<uint8_t> &graph, std::vector<float> &edges) {
    graph.clear();
    edges.clear();
    if (g.n == 0) return;
    const uint32_t n_layers = g.ep_layer + 1;
    const bool have_w = !g.l0_w.empty();
    std::vector<uint32_t> ends;
    ends.reserve(g.n);
    uint32_t pos = 0;
    for (uint32_t i = 0; i < g.n; i++) {
        std::vector<uint32_t> layer_start(n_layers);
        uint32_t node_pos = 0;
        for (uint32_t l = 0; l < n_layers; l++) {
            layer_start[l] = node_pos;
            const uint32_t *rec = nullptr;
            const float *w = nullptr;
            if (l == 0) {
                rec = &g.l0[(size_t)i * NIDX_L0_STRIDE];
                if (have_w) w = &g.l0_w[(size_t)i * NIDX_L0_STRIDE];
            } else if (l <= g.top_layer[i] && g.upper_base[i] != 0xffffffffu) {
                size_t r = ((size_t)g.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE;
                rec = &g.upper[r];
                if (have_w) w = &g.upper_w[r];
            }
            uint32_t deg = rec ? rec[0] : 0;
            wr32(graph, deg);
            for (uint32_t e = 0; e < deg; e++) {
                wr32(graph, rec[1 + e]);
                edges.push_back(w ? w[1 + e] : 0.f);
            }
            node_pos += (1 + deg) * 4;
        }
        node_pos += n_layers * 4;
        for (uint32_t l = n_layers; l-- > 0;) wr32(graph, node_pos - layer_start[l]);
        pos += node_pos;
        ends.push_back(pos);
    }
    for (size_t i = ends.size(); i-- > 0;) wr32(graph, ends[i]);
    wr32(graph, g.ep_layer);
    wr32(graph, g.ep_node);
}

// ---- level draw ---------------------------------------------------------------------------------
namespace {
struct Xoshiro256pp {
    uint64_t s[4];
    explicit Xoshiro256pp(uint64_t seed) {
        // rand::SeedableRng::seed_from_u64 for xoshiro: four SplitMix64 outputs
        for (int i = 0; i < 4; i++) {
            seed += 0x9e3779b97f4a7c15ULL;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
};
}  // namespace

void draw_levels(uint64_t seed, uint32_t n, std::vector<uint8_t> &levels) {
    levels.resize(n);
    Xoshiro256pp rng(seed);
    const double factor = 1.0 / log((double)NIDX_M);
    for (uint32_t i = 0; i < n; i++) {
        // Uniform<f64>::new(0,1): 52 mantissa bits -> [1,2) - 1
        uint64_t bits = (rng.next() >> 12) | 0x3ff0000000000000ULL;
        double u;
        memcpy(&u, &bits, 8);
        u -= 1.0;
        double picked = round(-log(u) * factor);
        int lvl = picked > 0.0 ? (picked > 63.0 ? 63 : (int)picked) : 0;
        levels[i] = (uint8_t)lvl;
    }
}

}  // namespace nidx

extern "C" int32_t nidx_gpu_hnsw_graph_check(const uint8_t *graph, uint64_t graph_len, const float *edges, uint64_t n_edges, uint32_t n_nodes,
                                             uint32_t *entry_node_out, uint32_t *entry_layer_out, uint64_t *n_links_out,
                                             uint64_t *n_broken_links_out) try {
    using namespace nidx;
    if (graph_len && !graph) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL graph");
    HostGraph g;
    std::string err;
    int rc = parse_disk_v2(graph, graph_len, n_nodes, g, err);
    if (rc == NIDX_OK) rc = attach_edge_weights(g, graph, graph_len, edges, n_edges, err);
    if (rc != NIDX_OK) return fail(rc, "%s", err.c_str());
    auto links = [&]() {
        uint64_t n = 0;
        for (uint32_t i = 0; i < g.n; i++) {
            n += g.l0[(size_t)i * NIDX_L0_STRIDE];
            for (uint32_t l = 1; l <= g.top_layer[i] && g.upper_base[i] != 0xffffffffu; l++) n += g.upper[((size_t)g.upper_base[i] + (l - 1)) * NIDX_UP_STRIDE];
        }
        return n;
    };
    const uint64_t before = links();
    if (entry_node_out) *entry_node_out = g.ep_node;
    if (entry_layer_out) *entry_layer_out = g.ep_layer;
    if (n_links_out) *n_links_out = before;
    if (n_broken_links_out) {
        fix_broken_graph(g);
        *n_broken_links_out = before - links();
    }
    return NIDX_OK;
} NIDX_ABI_CATCH
