// fst_index.cpp — the three inverted-index files of a vector segment: field.fst, label.fst, index.map.
//
// Reference: nidx_vector/src/inverted_index/{fst_index.rs:26-87, map.rs:27-86, paragraph.rs:68-121}, inverted_index.rs:31-61.
//   index.map   records back to back: [n: u64 LE][stream-vbyte (scalar) encoding of n u32 paragraph addresses]; a record's byte
//               offset is the value its key maps to.  stream-vbyte 0.4.1: ceil(n / 4) control bytes (2 bits per number, number
//               4 i + j in bits 2 j of control byte i, code = its length in bytes - 1), then the numbers' 1-4 little-endian bytes.
//   *.fst       an `fst::Map` (fst 0.4.7, format version 3): key bytes -> u64 (the record's offset in index.map).
// Both are third-party containers that are not vendored in the reference tree and cannot be built here (no Rust toolchain): the
// layouts below restate the crates' published formats — **unpinned** against the crates themselves; what the
// tests pin is that the reader reads what the writer writes, hand-assembled images of every node kind, and that a directory
// opened through these files gives the same posting lists as ParagraphInvertedIndexes::build's logic on the paragraph store.
//
// fst image: [version u64 = 3][type u64 = 0][nodes ...][len u64][root address u64][checksum u32: masked CRC32C of all before].
// A node is read BACKWARDS from its address (the position of its last byte, the state byte).  Three node kinds:
//   OneTransNext  0b11cccccc: one transition to the node right below (address = this node's first byte - 1), output 0;
//   OneTrans      0b10cccccc: one transition: [output][delta address][pack sizes][input?][state];
//   AnyTrans      0b0fnnnnnn: f = final, n = transitions (0: the count sits in the byte below; 256 is stored as 1):
//                 [final output][outputs, last transition first][delta addresses][inputs][256-byte index if n > 32][pack sizes][n?][state].
// cccccc = index + 1 of the input in the crate's table of common bytes (0: the input byte is stored).  pack sizes = transition
// bytes << 4 | output bytes; a delta address = (this node's first byte) - (target's address), 0 = the empty final node (address 0,
// never written).  A key's value is the sum of the outputs on its path plus the final output of the last node.
// The writer builds a trie (no suffix sharing: the reader does not need minimality), every node as AnyTrans, the value of a key in
// the final output of its last node.
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "fst_index.h"

namespace nidx {

// ---- index.map ------------------------------------------------------------------------------------------------------------------
uint64_t map_append(std::vector<uint8_t> &out, const uint32_t *ids, size_t n) {
    const uint64_t pos = out.size();
    const uint64_t n64 = n;
    for (int i = 0; i < 8; i++) out.push_back((uint8_t)(n64 >> (8 * i)));
    const size_t ctrl_at = out.size();
    out.resize(out.size() + (n + 3) / 4, 0);
    for (size_t i = 0; i < n; i++) {
        const uint32_t v = ids[i];
        const uint32_t len = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
        out[ctrl_at + i / 4] |= (uint8_t)((len - 1) << (2 * (i % 4)));
        for (uint32_t b = 0; b < len; b++) out.push_back((uint8_t)(v >> (8 * b)));
    }
    return pos;
}

bool map_read(const uint8_t *data, size_t len, uint64_t pos, std::vector<uint32_t> &out) {
    out.clear();
    if (pos > len || len - pos < 8) return false;
    uint64_t n = 0;
    for (int i = 0; i < 8; i++) n |= (uint64_t)data[pos + i] << (8 * i);
    if (n > 0xffffffffull) return false;
    const size_t ctrl = pos + 8, n_ctrl = (size_t)((n + 3) / 4);
    if (n_ctrl > len - ctrl) return false;
    size_t at = ctrl + n_ctrl;
    out.reserve((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t l = ((data[ctrl + i / 4] >> (2 * (i % 4))) & 3u) + 1u;
        if (l > len - at) return false;
        uint32_t v = 0;
        for (uint32_t b = 0; b < l; b++) v |= (uint32_t)data[at + b] << (8 * b);
        at += l;
        out.push_back(v);
    }
    return true;
}

// ---- fst ------------------------------------------------------------------------------------------------------------------------
namespace {

// the crate's table of common input bytes, most common first (raw/common_inputs.rs: COMMON_INPUTS_INV); a state byte can name
// the first 63 of them
const char kCommonInputs[] = "te/oasripcnw.hlm-du012g=:bf3y5&_4v9678k%?xCDASFIBEjPTzRNM+LOqHGWUV,YKJZXQ;)(~[]$!'*@";

uint32_t crc32c(const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = ~0u;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return ~c;
}
uint32_t masked(uint32_t sum) { return ((sum >> 15) | (sum << 17)) + 0xA282EAD8u; }

uint32_t pack_size(uint64_t v) {
    uint32_t n = 1;
    while (n < 8 && (v >> (8 * n)) != 0) n++;
    return n;
}
void pack_uint(std::vector<uint8_t> &out, uint64_t v, uint32_t nbytes) {
    for (uint32_t i = 0; i < nbytes; i++) out.push_back((uint8_t)(v >> (8 * i)));
}
uint64_t unpack_uint(const uint8_t *p, uint32_t nbytes) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < nbytes; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

struct BuildNode {
    bool is_final = false;
    uint64_t final_output = 0;
    std::vector<std::pair<uint8_t, uint64_t>> trans;   // (input, target address), inputs ascending
};

// one node as StateAnyTrans (raw/node.rs: StateAnyTrans::compile) -> its address
uint64_t compile_node(std::vector<uint8_t> &out, const BuildNode &nd) {
    if (nd.is_final && nd.trans.empty() && nd.final_output == 0) return 0;   // the empty final node
    const uint64_t start = out.size();
    uint32_t tsize = 0;
    for (const auto &t : nd.trans) tsize = std::max(tsize, pack_size(t.second == 0 ? 0 : start - t.second));
    const bool any_outs = nd.final_output != 0;
    const uint32_t osize = any_outs ? pack_size(nd.final_output) : 0;
    if (any_outs) {
        if (nd.is_final) pack_uint(out, nd.final_output, osize);
        for (size_t i = nd.trans.size(); i-- > 0;) pack_uint(out, 0, osize);
    }
    for (size_t i = nd.trans.size(); i-- > 0;) pack_uint(out, nd.trans[i].second == 0 ? 0 : start - nd.trans[i].second, tsize);
    for (size_t i = nd.trans.size(); i-- > 0;) out.push_back(nd.trans[i].first);
    if (nd.trans.size() > 32) {
        uint8_t index[256];
        memset(index, 255, sizeof(index));
        for (size_t i = 0; i < nd.trans.size(); i++) index[nd.trans[i].first] = (uint8_t)i;
        out.insert(out.end(), index, index + 256);
    }
    out.push_back((uint8_t)((tsize << 4) | osize));
    const size_t n = nd.trans.size();
    if (n == 0 || n > 63) out.push_back(n == 256 ? 1 : (uint8_t)n);
    out.push_back((uint8_t)((nd.is_final ? 0x40 : 0x00) | (n >= 1 && n <= 63 ? n : 0)));
    return out.size() - 1;
}

// a node of an image, decoded
struct Node {
    bool is_final = false;
    uint64_t final_output = 0;
    uint32_t n = 0;
    // transition i
    std::vector<uint8_t> input;
    std::vector<uint64_t> output, addr;
};

bool decode_node(const uint8_t *d, size_t len, uint64_t version, uint64_t addr, Node &nd) {
    nd = Node();
    if (addr == 0) {   // the empty final node
        nd.is_final = true;
        return true;
    }
    if (addr >= len) return false;
    const uint8_t st = d[addr];
    auto need = [&](uint64_t bytes) { return bytes <= addr; };   // bytes below the state byte
    if ((st >> 6) == 3 || (st >> 6) == 2) {
        const uint32_t c = st & 0x3f;
        if (c > sizeof(kCommonInputs) - 1) return false;
        const uint32_t input_len = c ? 0 : 1;
        uint8_t inp;
        if (c) inp = (uint8_t)kCommonInputs[c - 1];
        else {
            if (!need(1)) return false;
            inp = d[addr - 1];
        }
        nd.n = 1;
        nd.input.push_back(inp);
        if ((st >> 6) == 3) {   // OneTransNext
            if (!need(input_len + 1)) return false;
            const uint64_t end = addr - input_len;
            nd.output.push_back(0);
            nd.addr.push_back(end - 1);
            return true;
        }
        if (!need(input_len + 1)) return false;
        const uint8_t sizes = d[addr - input_len - 1];
        const uint32_t tsize = sizes >> 4, osize = sizes & 15;
        if (tsize == 0 || tsize > 8 || osize > 8 || !need(input_len + 1 + tsize + osize)) return false;
        const uint64_t end = addr - input_len - 1 - tsize - osize;
        const uint64_t delta = unpack_uint(d + addr - input_len - 1 - tsize, tsize);
        if (delta > end) return false;
        nd.addr.push_back(delta == 0 ? 0 : end - delta);
        nd.output.push_back(osize ? unpack_uint(d + end, osize) : 0);
        return true;
    }
    nd.is_final = (st & 0x40) != 0;
    uint32_t n = st & 0x3f, ntrans_len = 0;
    if (n == 0) {
        if (!need(1)) return false;
        ntrans_len = 1;
        n = d[addr - 1];
        if (n == 1) n = 256;
    }
    if (!need(ntrans_len + 1)) return false;
    const uint8_t sizes = d[addr - ntrans_len - 1];
    const uint32_t tsize = sizes >> 4, osize = sizes & 15;
    if (tsize > 8 || osize > 8 || (n && tsize == 0)) return false;
    const uint64_t index_size = (version >= 2 && n > 32) ? 256 : 0;
    const uint64_t total_trans = index_size + (uint64_t)n * (1 + tsize);
    const uint64_t below = ntrans_len + 1 + total_trans + (uint64_t)n * osize + (nd.is_final ? osize : 0);
    if (!need(below)) return false;
    const uint64_t end = addr - below;
    nd.n = n;
    nd.input.resize(n), nd.output.resize(n), nd.addr.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        nd.input[i] = d[addr - ntrans_len - 1 - index_size - i - 1];
        const uint64_t at = addr - ntrans_len - 1 - index_size - n - (uint64_t)i * tsize - tsize;
        const uint64_t delta = unpack_uint(d + at, tsize);
        if (delta > end) return false;
        nd.addr[i] = delta == 0 ? 0 : end - delta;
        nd.output[i] = osize ? unpack_uint(d + addr - ntrans_len - 1 - total_trans - (uint64_t)i * osize - osize, osize) : 0;
    }
    if (nd.is_final && osize) nd.final_output = unpack_uint(d + addr - ntrans_len - 1 - total_trans - (uint64_t)n * osize - osize, osize);
    return true;
}

struct Image {
    const uint8_t *d;
    size_t len;        // bytes in front of the footer: node addresses are < len
    uint64_t version, n_keys, root;
};
bool open_image(const uint8_t *data, size_t len, Image &im) {
    if (len < 32) return false;
    im.d = data;
    im.version = unpack_uint(data, 8);
    if (im.version < 1 || im.version > 3) return false;
    size_t end = len;
    if (im.version >= 3) {
        if (len < 36) return false;
        end = len - 4;   // the checksum is not verified at open by the crate either (Fst::verify does); fst_check_sum() below does
    }
    im.root = unpack_uint(data + end - 8, 8);
    im.n_keys = unpack_uint(data + end - 16, 8);
    im.len = end - 16;
    return im.root < im.len || im.root == 0;
}

}  // namespace

bool fst_build(const std::vector<std::pair<std::string, uint64_t>> &entries, std::vector<uint8_t> &out) {
    out.clear();
    pack_uint(out, 3, 8);   // version
    pack_uint(out, 0, 8);   // type
    // the path of the previous key: stack[d] = the unfinished node at depth d
    std::vector<BuildNode> stack(1);
    std::string prev;
    bool first = true;
    auto freeze_below = [&](size_t keep) {   // compile the nodes deeper than `keep` and hang them on their parents
        while (stack.size() > keep + 1) {
            const uint64_t addr = compile_node(out, stack.back());
            stack.pop_back();
            stack.back().trans.back().second = addr;
        }
    };
    for (const auto &e : entries) {
        const std::string &key = e.first;
        if (!first && !(prev < key)) return false;   // keys strictly ascending (fst::MapBuilder::insert's rule)
        size_t common = 0;
        while (!first && common < key.size() && common < prev.size() && key[common] == prev[common]) common++;
        freeze_below(common);
        for (size_t i = common; i < key.size(); i++) {
            stack.back().trans.push_back({(uint8_t)key[i], 0});
            stack.push_back(BuildNode());
        }
        stack.back().is_final = true;
        stack.back().final_output = e.second;
        prev = key;
        first = false;
    }
    freeze_below(0);
    const uint64_t root = compile_node(out, stack[0]);
    pack_uint(out, entries.size(), 8);
    pack_uint(out, root, 8);
    pack_uint(out, masked(crc32c(out.data(), out.size())), 4);
    return true;
}

bool fst_check_sum(const uint8_t *data, size_t len) {
    if (len < 36 || unpack_uint(data, 8) < 3) return true;   // older versions carry none
    return (uint32_t)unpack_uint(data + len - 4, 4) == masked(crc32c(data, len - 4));
}

bool fst_enumerate(const uint8_t *data, size_t len, std::vector<std::pair<std::string, uint64_t>> &out, uint64_t max_keys) {
    out.clear();
    Image im;
    if (!open_image(data, len, im) || im.n_keys > max_keys) return false;
    // an image with shared suffixes is a DAG: the walk is bounded by the keys it may produce, not by the file size
    const uint64_t max_visits = 4096 * (im.n_keys + 16);
    uint64_t visits = 0;
    struct Frame {
        Node nd;
        uint32_t next;
        uint64_t sum;
    };
    std::vector<Frame> st;
    std::string key;
    st.push_back(Frame());
    if (!decode_node(im.d, im.len, im.version, im.root, st.back().nd)) return false;
    st.back().next = 0, st.back().sum = 0;
    if (st.back().nd.is_final) out.push_back({key, st.back().nd.final_output});
    while (!st.empty()) {
        Frame &f = st.back();
        if (f.next == f.nd.n) {
            st.pop_back();
            if (!key.empty()) key.pop_back();
            continue;
        }
        const uint32_t i = f.next++;
        if (i > 0 && f.nd.input[i] <= f.nd.input[i - 1]) return false;   // transitions ascend
        if (st.size() > 65536 || out.size() > im.n_keys || ++visits > max_visits) return false;   // a cycle, or more keys than the footer says
        key.push_back((char)f.nd.input[i]);
        const uint64_t sum = f.sum + f.nd.output[i];
        const uint64_t target = f.nd.addr[i];
        st.push_back(Frame());
        Frame &c = st.back();
        if (!decode_node(im.d, im.len, im.version, target, c.nd)) return false;
        c.next = 0, c.sum = sum;
        if (c.nd.is_final) out.push_back({key, sum + c.nd.final_output});
    }
    return out.size() == im.n_keys;
}

bool fst_get(const uint8_t *data, size_t len, const uint8_t *key, size_t key_len, uint64_t *value_out) {
    Image im;
    if (!open_image(data, len, im)) return false;
    Node nd;
    if (!decode_node(im.d, im.len, im.version, im.root, nd)) return false;
    uint64_t sum = 0;
    for (size_t p = 0; p < key_len; p++) {
        uint32_t i = 0;
        while (i < nd.n && nd.input[i] != key[p]) i++;
        if (i == nd.n) return false;
        sum += nd.output[i];
        const uint64_t target = nd.addr[i];
        if (!decode_node(im.d, im.len, im.version, target, nd)) return false;
    }
    if (!nd.is_final) return false;
    *value_out = sum + nd.final_output;
    return true;
}

}  // namespace nidx
