// segment_v1.cpp — the pre-migration segment files, read once and handed on in the current layouts, host side only.
//
//   nodes.kv    DataStoreV1 (nidx_vector/src/data_store/v1.rs:33-87, v1/store.rs:23-72, v1/node.rs:19-133, v1/trie.rs:29-105):
//               [n: u64 LE][n slot addresses: u64 LE][slots].  A slot is a Node: header of four u64 (len, vector_start, key_start,
//               label_start), the metadata bytes, the vector segment [len: u32][pad: u32][pad zero bytes][len vector bytes], the key
//               segment [len: u64][utf-8], and the label trie: [len: u64] then per trie node [is_final: u8][n_edges: u64]{[byte: u8]
//               [target node: u64]} and, at the end of the `len` bytes, one u64 per trie node in reverse order = where that node's
//               record starts.  One vector per paragraph: VectorAddr == ParagraphAddr.
//   index.hnsw  DiskHnswV1 (hnsw/disk/v1.rs:15-36,128-200,229-289): per node, for every layer of the graph, [n_edges: u64]
//               {[node: u64][weight: f32]}, then one u64 per layer in reverse order = where that layer's record starts (absolute); behind
//               the nodes one u64 per node in reverse order = where the node ends, then the entry point [layer: u64][node: u64].
//
// segment::open reads such a directory through DataStoreV1 / DiskHnswV1 (segment.rs:39-57, hnsw/disk.rs:25-32) and segment::merge writes
// the current formats from it (segment.rs:117-128).  Here the migration happens in memory at open: the functions below produce the
// vectors.bin / paragraphs.bin / paragraphs.pos and hnsw.graph / hnsw.edges images of the same segment, and segment_dir.cpp carries on
// as if it had mapped those files.  Nothing in the reference tree holds bytes of either format (their writers are test-only code now):
// the layouts are restated from the serialisers, **parity unpinned** beyond that.
#include <cstring>
#include <string>
#include <vector>

#include "hnsw_graph.h"
#include "segment_v1.h"

namespace nidx {
namespace {

bool rd64(const uint8_t *d, size_t len, uint64_t at, uint64_t &out) {
    if (at > len || len - at < 8) return false;
    out = 0;
    for (int i = 0; i < 8; i++) out |= (uint64_t)d[at + i] << (8 * i);
    return true;
}
bool rd32(const uint8_t *d, size_t len, uint64_t at, uint32_t &out) {
    if (at > len || len - at < 4) return false;
    out = 0;
    for (int i = 0; i < 4; i++) out |= (uint32_t)d[at + i] << (8 * i);
    return true;
}
void put_varint(std::vector<uint8_t> &o, uint64_t v) {   // bincode-2 `standard()` integer (segment_dir.cpp)
    int n;
    if (v < 251) { o.push_back((uint8_t)v); return; }
    if (v <= 0xffffu) { o.push_back(251); n = 2; }
    else if (v <= 0xffffffffu) { o.push_back(252); n = 4; }
    else { o.push_back(253); n = 8; }
    for (int i = 0; i < n; i++) o.push_back((uint8_t)(v >> (8 * i)));
}

// trie::decompress (v1/trie.rs:71-105): every label of a node's trie, depth first in the order the edges are stored
bool trie_labels(const uint8_t *t, size_t avail, std::vector<std::string> &out) {
    uint64_t tlen;
    if (!rd64(t, avail, 0, tlen) || tlen > avail || tlen < 8) return false;
    struct Frame { uint64_t node_ptr, n_edges, next; };
    std::vector<Frame> st;
    std::string cur;
    auto node_ptr_of = [&](uint64_t node, uint64_t &ptr) {   // get_node_ptr: the index sits at the end, node 0 last
        if (node + 1 > tlen / 8) return false;
        return rd64(t, tlen, tlen - (node + 1) * 8, ptr) && ptr <= tlen;
    };
    auto enter = [&](uint64_t node) {
        uint64_t ptr, n_edges;
        if (!node_ptr_of(node, ptr) || ptr + 9 > tlen) return false;
        if (t[ptr] == 1) out.push_back(cur);
        if (!rd64(t, tlen, ptr + 1, n_edges) || n_edges > (tlen - ptr - 9) / 9) return false;
        st.push_back({ptr, n_edges, 0});
        return true;
    };
    if (!enter(0)) return false;
    uint64_t visited = 0;
    while (!st.empty()) {
        Frame &f = st.back();
        if (f.next == f.n_edges) {
            st.pop_back();
            if (!cur.empty()) cur.pop_back();
            continue;
        }
        const uint64_t e = f.node_ptr + 9 + 9 * f.next++;
        uint64_t target;
        if (!rd64(t, tlen, e + 1, target)) return false;
        if (st.size() > 4096 || ++visited > tlen) return false;   // (a trie has fewer nodes than bytes: anything else is a cycle)
        cur.push_back((char)t[e]);
        if (!enter(target)) return false;
    }
    return true;
}

}  // namespace

int migrate_nodes_kv(const uint8_t *kv, size_t len, uint32_t dimension, std::vector<uint8_t> &vectors_bin, std::vector<uint8_t> &paragraphs_bin,
                     std::vector<uint8_t> &paragraphs_pos, std::string &err) {
    vectors_bin.clear(), paragraphs_bin.clear(), paragraphs_pos.clear();
    uint64_t n;
    if (!rd64(kv, len, 0, n) || n > 0xffffffffull || n > (len - 8) / 8) { err = "nodes.kv: bad element count"; return -1; }
    const size_t row = (size_t)dimension * 4;
    vectors_bin.reserve((size_t)n * (row + 4));
    std::vector<std::string> labels;
    for (uint64_t id = 0; id < n; id++) {
        uint64_t ptr, nlen, vstart, kstart, lstart;
        auto bad = [&](const char *what) { err = "nodes.kv: node " + std::to_string(id) + ": " + what; return -1; };
        if (!rd64(kv, len, 8 + 8 * id, ptr) || ptr > len) return bad("slot address beyond the file");
        const uint8_t *nd = kv + ptr;
        const size_t avail = len - ptr;
        if (!rd64(nd, avail, 0, nlen) || nlen > avail || nlen < 32) return bad("bad length");
        if (!rd64(nd, nlen, 8, vstart) || !rd64(nd, nlen, 16, kstart) || !rd64(nd, nlen, 24, lstart)) return bad("truncated header");
        if (vstart < 32 || vstart > nlen || kstart > nlen || lstart > nlen) return bad("segment offsets beyond the node");
        // vector segment
        uint32_t vlen, vpad;
        if (!rd32(nd, nlen, vstart, vlen) || !rd32(nd, nlen, vstart + 4, vpad)) return bad("truncated vector segment");
        const uint64_t vdata = vstart + 8 + vpad;
        if (vlen != row) return bad("vector length does not match the index dimension");
        if (vdata > nlen || nlen - vdata < vlen) return bad("vector beyond the node");
        const uint32_t trailer = (uint32_t)id;   // one vector per paragraph
        vectors_bin.insert(vectors_bin.end(), nd + vdata, nd + vdata + vlen);
        for (int i = 0; i < 4; i++) vectors_bin.push_back((uint8_t)(trailer >> (8 * i)));
        // key segment
        uint64_t klen;
        if (!rd64(nd, nlen, kstart, klen) || klen > nlen - kstart - 8) return bad("key beyond the node");
        // labels
        labels.clear();
        if (!trie_labels(nd + lstart, nlen - lstart, labels)) return bad("malformed label trie");
        // StoredParagraph {key, labels, metadata, first_vector, num_vectors} in the bincode layout of paragraphs.bin
        if (paragraphs_bin.size() > 0xffffffffull) { err = "nodes.kv: the paragraph records exceed the 4 GiB paragraphs.pos addresses"; return -1; }
        const uint32_t pos = (uint32_t)paragraphs_bin.size();
        for (int i = 0; i < 4; i++) paragraphs_pos.push_back((uint8_t)(pos >> (8 * i)));
        put_varint(paragraphs_bin, klen);
        paragraphs_bin.insert(paragraphs_bin.end(), nd + kstart + 8, nd + kstart + 8 + klen);
        put_varint(paragraphs_bin, labels.size());
        for (const std::string &l : labels) {
            put_varint(paragraphs_bin, l.size());
            paragraphs_bin.insert(paragraphs_bin.end(), l.begin(), l.end());
        }
        put_varint(paragraphs_bin, vstart - 32);   // metadata: everything between the header and the vector segment
        paragraphs_bin.insert(paragraphs_bin.end(), nd + 32, nd + vstart);
        put_varint(paragraphs_bin, id);
        put_varint(paragraphs_bin, 1);
    }
    return 0;
}

int migrate_index_hnsw(const uint8_t *buf, size_t len, uint32_t n_nodes, std::vector<uint8_t> &graph, std::vector<float> &edges, std::string &err) {
    graph.clear(), edges.clear();
    if (len == 0) return 0;   // an empty graph is an empty file (v1.rs:178-181)
    if (len < 24 || n_nodes == 0 || (uint64_t)n_nodes > (len - 16) / 8) { err = "index.hnsw: too short for its nodes"; return -1; }
    uint64_t ep_layer, ep_node;
    if (!rd64(buf, len, len - 16, ep_layer) || !rd64(buf, len, len - 8, ep_node)) { err = "index.hnsw: truncated entry point"; return -1; }
    if (ep_node >= n_nodes || ep_layer > 255) { err = "index.hnsw: entry point out of range"; return -1; }
    // DiskHnswV1::deserialize (v1.rs:229-289): nodes until a node ends where the node index starts, layers until a layer's connexions
    // end where the node's layer index starts; a node is in layer l > 0 when it has edges there
    struct Rec { std::vector<uint32_t> to; std::vector<float> w; };
    std::vector<std::vector<Rec>> nodes;   // [node][layer]
    const uint64_t index_end = len - 16;
    uint64_t total_edges = 0;   // every edge owns 12 bytes of its own in a well-formed file
    for (uint64_t i = 0;; i++) {
        if (i >= n_nodes) { err = "index.hnsw: more nodes than the segment has vectors"; return -1; }
        const uint64_t indexing_pos = index_end - (i + 1) * 8;
        uint64_t node_end;
        if (!rd64(buf, len, indexing_pos, node_end) || node_end > indexing_pos || node_end < 8) { err = "index.hnsw: bad node end"; return -1; }
        nodes.emplace_back();
        for (uint64_t l = 0;; l++) {
            if ((l + 1) * 8 > node_end || l > 255) { err = "index.hnsw: bad layer index"; return -1; }
            const uint64_t layer_pos = node_end - (l + 1) * 8;
            uint64_t start, n_edges;
            if (!rd64(buf, len, layer_pos, start) || start > layer_pos || !rd64(buf, layer_pos, start, n_edges) || n_edges > (layer_pos - start - 8) / 12) {
                err = "index.hnsw: bad connexion list";
                return -1;
            }
            // the device layout's bounds, enforced WHILE reading: offsets of different layers and nodes may point at the same connexion
            // bytes, so a small crafted file could otherwise expand to n_nodes x 256 x (len / 12) edges in host memory before the
            // stride checks below ever ran (an out-of-memory denial of service from an on-disk file)
            if (n_edges >= (l == 0 ? (uint64_t)NIDX_L0_STRIDE : (uint64_t)NIDX_UP_STRIDE)) {
                err = l == 0 ? "index.hnsw: more layer-0 edges than the device layout holds" : "index.hnsw: more upper-layer edges than the device layout holds";
                return -1;
            }
            total_edges += n_edges;
            if (total_edges > len / 12) { err = "index.hnsw: connexion lists overlap (more edges than the file has bytes for)"; return -1; }
            Rec r;
            for (uint64_t e = 0; e < n_edges; e++) {
                uint64_t to;
                uint32_t wbits;
                rd64(buf, len, start + 8 + 12 * e, to);
                rd32(buf, len, start + 16 + 12 * e, wbits);
                if (to >= n_nodes) { err = "index.hnsw: edge to a node the segment does not have"; return -1; }
                float w;
                memcpy(&w, &wbits, 4);
                r.to.push_back((uint32_t)to);
                r.w.push_back(w);
            }
            nodes.back().push_back(std::move(r));
            if (start + 8 + 12 * n_edges == layer_pos) break;
        }
        if (node_end == indexing_pos) break;
    }
    if (nodes.size() != n_nodes) { err = "index.hnsw: " + std::to_string(nodes.size()) + " nodes for " + std::to_string(n_nodes) + " vectors"; return -1; }
    HostGraph g;
    g.n = n_nodes;
    g.ep_node = (uint32_t)ep_node, g.ep_layer = (uint32_t)ep_layer;
    g.l0.assign((size_t)n_nodes * NIDX_L0_STRIDE, 0);
    g.l0_w.assign((size_t)n_nodes * NIDX_L0_STRIDE, 0.f);
    g.upper_base.assign(n_nodes, 0xffffffffu);
    g.top_layer.assign(n_nodes, 0);
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint32_t top = 0;
        for (uint32_t l = 1; l < nodes[i].size(); l++)
            if (!nodes[i][l].to.empty()) top = l;
        if (top > ep_layer) { err = "index.hnsw: a node reaches above the entry point's layer"; return -1; }
        g.top_layer[i] = (uint8_t)top;
        const Rec &r0 = nodes[i][0];
        if (r0.to.size() >= NIDX_L0_STRIDE) { err = "index.hnsw: more layer-0 edges than the device layout holds"; return -1; }
        g.l0[(size_t)i * NIDX_L0_STRIDE] = (uint32_t)r0.to.size();
        for (size_t e = 0; e < r0.to.size(); e++) {
            g.l0[(size_t)i * NIDX_L0_STRIDE + 1 + e] = r0.to[e];
            g.l0_w[(size_t)i * NIDX_L0_STRIDE + 1 + e] = r0.w[e];
        }
        if (top) {
            g.upper_base[i] = g.n_upper_records();
            for (uint32_t l = 1; l <= top; l++) {
                const Rec &r = nodes[i][l];
                if (r.to.size() >= NIDX_UP_STRIDE) { err = "index.hnsw: more upper-layer edges than the device layout holds"; return -1; }
                const size_t at = g.upper.size();
                g.upper.resize(at + NIDX_UP_STRIDE, 0);
                g.upper_w.resize(at + NIDX_UP_STRIDE, 0.f);
                g.upper[at] = (uint32_t)r.to.size();
                for (size_t e = 0; e < r.to.size(); e++) {
                    g.upper[at + 1 + e] = r.to[e];
                    g.upper_w[at + 1 + e] = r.w[e];
                }
            }
        }
    }
    serialize_disk_v2(g, graph, edges);
    return 0;
}

}  // namespace nidx
