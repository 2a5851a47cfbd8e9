// segment_dir.cpp — a vector segment directory as the reference writes it, host side only.
//
//   vectors.bin      data_store/v2/vector_store.rs:30-40,131-147   record = dimension f32 LE + u32 LE paragraph address
//   paragraphs.bin   data_store/v2/paragraph_store.rs:37-44,132-150 StoredParagraph records, bincode-2 "standard" layout
//   paragraphs.pos   ibid. :100-106,141                             u32 LE start of every record in paragraphs.bin
//   vectors.quant    data_store/v2/quant_vector_store.rs:29-64      dimension/8 + 8 bytes per vector (rabitq.rs:38-106)
//   hnsw.graph/.edges hnsw/disk/v2.rs:16-49,214-252                 opaque here: handed to nidx_gpu_vector_open as they lie
//
//   field.fst / label.fst / index.map   inverted_index/{fst_index.rs,map.rs,paragraph.rs}   fst_index.cpp
//
//   nodes.kv / index.hnsw   data_store/v1.rs, hnsw/disk/v1.rs (pre-migration segments)   segment_v1.cpp: migrated in memory at open
//
// The files are mmap'd and handed to nidx_gpu_vector_open without a copy.  The inverted indexes are read from their three
// files when index.map is there (InvertedIndexes::exists, inverted_index.rs:57-60) and every list they hold passes the
// checks below; otherwise — like segment::open when they are missing (segment.rs:49-67) — the posting lists are rebuilt
// from the paragraph store, keyed the same way (ParagraphInvertedIndexes::build, inverted_index/paragraph.rs:68-103:
// labels_key / FieldKey).  Writers emit the three files next to the stores when NIDX_GPU_SEGMENT_DIR_FST=1 (the default only
// reads them; 0: neither written nor read).
//
// StoredParagraph is serialised with wincode configured to match bincode::config::standard() (utils.rs:25-28): little
// endian, variable-length integers (u < 251: one byte; 251 + u16; 252 + u32; 253 + u64), a sequence or string = its
// length as such an integer followed by the elements / UTF-8 bytes; struct fields in declaration order:
// key, labels, metadata, first_vector, num_vectors.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/nidx_gpu.h"
#include "fst_index.h"
#include "host_common.h"
#include "segment_v1.h"

namespace nidx {
namespace {

struct MappedFile {
    const uint8_t *p = nullptr;
    size_t len = 0;
    bool present = false;
    std::vector<uint8_t> owned;   // an image produced in memory instead of a mapped file (segment_v1.cpp)
    ~MappedFile() {
        if (p && len && owned.empty()) munmap(const_cast<uint8_t *>(p), len);
    }
    void adopt(std::vector<uint8_t> &&bytes) {
        owned = std::move(bytes);
        if (owned.empty()) owned.push_back(0), len = 0;   // (keeps `owned` non-empty: nothing to unmap)
        else len = owned.size();
        p = owned.data();
        present = true;
    }
    // 0 = ok, 1 = missing, -1 = error
    int open(const std::string &path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return errno == ENOENT ? 1 : -1;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return -1; }
        len = (size_t)st.st_size;
        present = true;
        if (len) {
            void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); p = nullptr; len = 0; return -1; }
            p = static_cast<const uint8_t *>(m);
        }
        ::close(fd);
        return 0;
    }
};

// bincode-2 varint
bool read_varint(const uint8_t *d, size_t len, size_t &at, uint64_t &out) {
    if (at >= len) return false;
    const uint8_t b = d[at++];
    int n;
    if (b < 251) { out = b; return true; }
    if (b == 251) n = 2;
    else if (b == 252) n = 4;
    else if (b == 253) n = 8;
    else return false;  // u128 never occurs in these records
    if (at + (size_t)n > len) return false;
    out = 0;
    for (int i = 0; i < n; i++) out |= (uint64_t)d[at + i] << (8 * i);
    at += (size_t)n;
    return true;
}

void write_varint(std::vector<uint8_t> &o, uint64_t v) {
    int n;
    if (v < 251) { o.push_back((uint8_t)v); return; }
    if (v <= 0xffffu) { o.push_back(251); n = 2; }
    else if (v <= 0xffffffffu) { o.push_back(252); n = 4; }
    else { o.push_back(253); n = 8; }
    for (int i = 0; i < n; i++) o.push_back((uint8_t)(v >> (8 * i)));
}

struct Span { uint64_t off; uint32_t len; };

struct Paragraph {
    Span key, metadata;
    uint32_t first_label, n_labels;  // into SegmentDir::labels
    uint32_t first_vector, num_vectors;
};

// 64-bit identity of a key string (FNV-1a folded through a finaliser); equal strings <=> equal ids up to 2^-64
uint64_t key_id(const uint8_t *s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ull; }
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

int hex_val(uint8_t c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

// uuid::Uuid::parse_str on the simple (32 hex) and hyphenated (8-4-4-4-12) forms -> 16 bytes
bool parse_uuid(const uint8_t *s, size_t n, uint8_t out[16]) {
    if (n == 36) {
        if (s[8] != '-' || s[13] != '-' || s[18] != '-' || s[23] != '-') return false;
    } else if (n != 32) {
        return false;
    }
    int k = 0;
    for (size_t i = 0; i < n;) {
        if (n == 36 && (i == 8 || i == 13 || i == 18 || i == 23)) { i++; continue; }
        const int hi = hex_val(s[i]), lo = hex_val(s[i + 1]);
        if (hi < 0 || lo < 0) return false;
        out[k++] = (uint8_t)(hi * 16 + lo);
        i += 2;
    }
    return k == 16;
}

// FieldKey::from_field_id (utils.rs:84-115): uuid bytes, then "type/name" when both are present; a lone uuid is the
// resource key; "uuid/type" alone is rejected.
bool field_key(const uint8_t *s, size_t n, std::string &out) {
    size_t cut[3], nc = 0;
    for (size_t i = 0; i < n && nc < 3; i++)
        if (s[i] == '/') cut[nc++] = i;
    const size_t uuid_end = nc >= 1 ? cut[0] : n;
    uint8_t rid[16];
    if (!parse_uuid(s, uuid_end, rid)) return false;
    out.assign(reinterpret_cast<const char *>(rid), 16);
    if (nc == 0) return true;
    if (nc == 1) return false;  // a field type without a name
    const size_t name_end = nc >= 3 ? cut[2] : n;
    out.append(reinterpret_cast<const char *>(s + cut[0] + 1), cut[1] - cut[0] - 1);
    out.push_back('/');
    out.append(reinterpret_cast<const char *>(s + cut[1] + 1), name_end - cut[1] - 1);
    return true;
}

// ParagraphInvertedIndexes::build (inverted_index/paragraph.rs:72-85): "F" + FieldKey bytes | "L" + labels_key -> addresses
struct ListBuilder {
    std::map<std::string, std::vector<uint32_t>> lists;
    std::string fk;
    void key(const uint8_t *s, size_t n, uint32_t a) {
        if (field_key(s, n, fk)) lists["F" + fk].push_back(a);
    }
    void label(const uint8_t *s, size_t n, uint32_t a) {
        if (n == 0) return;
        // labels_key: the label without its leading '/', plus a trailing '/'
        std::vector<uint32_t> &pl = lists["L" + std::string(reinterpret_cast<const char *>(s + 1), n - 1) + "/"];
        if (pl.empty() || pl.back() != a) pl.push_back(a);
    }
};

// NIDX_GPU_SEGMENT_DIR_FST: unset = READ the three index files when they are there and well formed, do not WRITE them; 1 = write
// them too; 0 = neither.  Writing is opt-in because the `fst` / `stream-vbyte` byte layouts are restated without the crates at hand
// (parity unpinned): a directory written without them is one the stock searcher completes itself on open
// (segment.rs:49-67), a directory written with a wrong one would be trusted.
bool fst_files_readable() {
    const char *e = getenv("NIDX_GPU_SEGMENT_DIR_FST");
    return !(e && e[0] == '0');
}
bool fst_files_written() {
    const char *e = getenv("NIDX_GPU_SEGMENT_DIR_FST");
    return e && e[0] == '1';
}

int write_bytes(const std::string &path, const void *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return -1;
    const bool ok = n == 0 || fwrite(p, 1, n, f) == n;
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

// ParagraphInvertedIndexes::build's file output: the field index first, then the label index, both into one index.map
// (paragraph.rs:87-100; IndexBuilder sorts every list, fst_index.rs:36-38 — they already ascend here).  index.map is written
// last: it is the file whose presence says "the indexes exist".
int write_index_files(const std::string &base, const std::map<std::string, std::vector<uint32_t>> &lists) {
    std::vector<uint8_t> map_bytes, image;
    for (const char kind : {'F', 'L'}) {
        std::vector<std::pair<std::string, uint64_t>> entries;
        for (const auto &kv : lists)
            if (kv.first[0] == kind) entries.push_back({kv.first.substr(1), map_append(map_bytes, kv.second.data(), kv.second.size())});
        if (!fst_build(entries, image)) return -1;
        if (write_bytes(base + (kind == 'F' ? "field.fst" : "label.fst"), image.data(), image.size())) return -1;
    }
    return write_bytes(base + "index.map", map_bytes.data(), map_bytes.size());
}

}  // namespace

struct SegmentDir {
    uint32_t dimension = 0;
    MappedFile vectors, para_data, para_pos, quant, graph, edges;
    uint32_t n_vectors = 0, n_paragraphs = 0;
    uint64_t row_stride = 0;
    std::vector<Paragraph> paragraphs;
    std::vector<Span> labels;
    std::vector<uint64_t> key_ids;
    // the inverted indexes: sorted keys ("L" + labels_key | "F" + FieldKey bytes) -> ascending paragraph lists
    std::vector<std::string> list_keys;
    std::vector<uint64_t> list_offsets;
    std::vector<uint32_t> list_ids;
    uint32_t n_label_lists = 0;  // the "F" lists come first ('F' < 'L')
    bool lists_from_files = false;
    bool from_v1 = false;   // nodes.kv: the stores above are images made at open, not mapped files
};

namespace {
// The posting lists out of field.fst / label.fst / index.map.  false (nothing kept): a file is missing or anything in them
// is not what a well-formed index of THIS paragraph store looks like — the caller rebuilds from the paragraphs instead.
bool load_index_files(SegmentDir *d, const std::string &base) {
    MappedFile map, fst[2];
    if (map.open(base + "index.map") != 0 || fst[0].open(base + "field.fst") != 0 || fst[1].open(base + "label.fst") != 0) return false;
    std::vector<std::string> keys;
    std::vector<uint64_t> offsets(1, 0);
    std::vector<uint32_t> ids, list;
    std::vector<std::pair<std::string, uint64_t>> entries;
    uint32_t n_label = 0;
    for (int kind = 0; kind < 2; kind++) {
        if (!fst_check_sum(fst[kind].p, fst[kind].len) || !fst_enumerate(fst[kind].p, fst[kind].len, entries, map.len / 9 + 1)) return false;   // (a record is >= 9 bytes)
        for (const auto &e : entries) {
            if (!map_read(map.p, map.len, e.second, list) || list.empty()) return false;
            for (size_t i = 0; i < list.size(); i++)
                if (list[i] >= d->n_paragraphs || (i && list[i] <= list[i - 1])) return false;
            keys.push_back((kind == 0 ? "F" : "L") + e.first);
            ids.insert(ids.end(), list.begin(), list.end());
            offsets.push_back(ids.size());
        }
        if (kind == 1) n_label = (uint32_t)entries.size();
    }
    d->list_keys.swap(keys), d->list_offsets.swap(offsets), d->list_ids.swap(ids);
    d->n_label_lists = n_label;
    return true;
}
}  // namespace

}  // namespace nidx

using namespace nidx;

extern "C" {

int32_t nidx_gpu_segment_dir_open(const char *path, uint32_t dimension, nidx_gpu_segment_dir_t **dir_out) try {
    if (!path || !dir_out || dimension == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *dir_out = nullptr;
    std::unique_ptr<SegmentDir> d(new SegmentDir());
    d->dimension = dimension;
    const std::string base = std::string(path) + "/";
    MappedFile v1_nodes;
    const int v1_state = v1_nodes.open(base + "nodes.kv");
    if (v1_state < 0) return fail(NIDX_ERR_IO, "cannot read %snodes.kv", base.c_str());
    if (v1_state == 0) {
        // DataStoreV1::exists (data_store/v1.rs:89-91, segment.rs:41-57): a pre-migration segment.  Its records are re-laid out in
        // memory as vectors.bin / paragraphs.bin / paragraphs.pos (what segment::merge would write from it, segment.rs:117-128)
        std::vector<uint8_t> vb, pb, pp;
        std::string err;
        if (migrate_nodes_kv(v1_nodes.p, v1_nodes.len, dimension, vb, pb, pp, err)) return fail(NIDX_ERR_IO, "%s%s", base.c_str(), err.c_str());
        d->vectors.adopt(std::move(vb)), d->para_data.adopt(std::move(pb)), d->para_pos.adopt(std::move(pp));
        d->from_v1 = true;
    } else {
        if (d->vectors.open(base + "vectors.bin") != 0) return fail(NIDX_ERR_IO, "cannot open %svectors.bin", base.c_str());
        if (d->para_data.open(base + "paragraphs.bin") != 0) return fail(NIDX_ERR_IO, "cannot open %sparagraphs.bin", base.c_str());
        if (d->para_pos.open(base + "paragraphs.pos") != 0) return fail(NIDX_ERR_IO, "cannot open %sparagraphs.pos", base.c_str());
        if (d->quant.open(base + "vectors.quant") < 0) return fail(NIDX_ERR_IO, "cannot read %svectors.quant", base.c_str());
    }
    // open_disk_hnsw (hnsw/disk.rs:25-32): the current format first, then DiskHnswV1's index.hnsw
    if (d->graph.open(base + "hnsw.graph") < 0) return fail(NIDX_ERR_IO, "cannot read %shnsw.graph", base.c_str());
    if (d->edges.open(base + "hnsw.edges") < 0) return fail(NIDX_ERR_IO, "cannot read %shnsw.edges", base.c_str());
    if (!d->graph.present) {
        MappedFile v1_graph;
        const int gs = v1_graph.open(base + "index.hnsw");
        if (gs < 0) return fail(NIDX_ERR_IO, "cannot read %sindex.hnsw", base.c_str());
        if (gs == 0) {
            const uint64_t stride = (uint64_t)dimension * 4 + 4;
            if (d->vectors.len % stride) return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "vectors.bin (%zu bytes) is not a multiple of %llu-byte records", d->vectors.len, (unsigned long long)stride);
            std::vector<uint8_t> gb;
            std::vector<float> ew;
            std::string err;
            if (migrate_index_hnsw(v1_graph.p, v1_graph.len, (uint32_t)(d->vectors.len / stride), gb, ew, err)) return fail(NIDX_ERR_IO, "%s%s", base.c_str(), err.c_str());
            std::vector<uint8_t> eb(ew.size() * 4);
            if (!ew.empty()) memcpy(eb.data(), ew.data(), eb.size());
            d->graph.adopt(std::move(gb)), d->edges.adopt(std::move(eb));
        }
    }
    // vector_alignment(DenseF32) == 4 == U32_LEN: no padding after the trailer (vector_store.rs:35-41)
    d->row_stride = (uint64_t)dimension * 4 + 4;
    if (d->vectors.len % d->row_stride) return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "vectors.bin (%zu bytes) is not a multiple of %llu-byte records", d->vectors.len, (unsigned long long)d->row_stride);
    if (d->vectors.len / d->row_stride > 0xffffffffull || d->para_pos.len % 4) return fail(NIDX_ERR_IO, "malformed segment files");
    d->n_vectors = (uint32_t)(d->vectors.len / d->row_stride);
    d->n_paragraphs = (uint32_t)(d->para_pos.len / 4);
    if (d->quant.present && d->quant.len != (uint64_t)d->n_vectors * (dimension / 8 + 8))
        return fail(NIDX_ERR_IO, "vectors.quant holds %zu bytes, expected %llu", d->quant.len, (unsigned long long)d->n_vectors * (dimension / 8 + 8));
    if (d->edges.len % 4) return fail(NIDX_ERR_IO, "hnsw.edges is not a whole number of f32");
    // decode every StoredParagraph
    d->paragraphs.resize(d->n_paragraphs);
    d->key_ids.resize(d->n_paragraphs);
    const uint8_t *data = d->para_data.p;
    const size_t dlen = d->para_data.len;
    d->lists_from_files = fst_files_readable() && load_index_files(d.get(), base);
    const bool rebuild = !d->lists_from_files;
    ListBuilder lb;
    for (uint32_t a = 0; a < d->n_paragraphs; a++) {
        uint32_t start;
        memcpy(&start, d->para_pos.p + (size_t)a * 4, 4);
        size_t at = start;
        uint64_t v, n;
        Paragraph &pg = d->paragraphs[a];
        auto bad = [&]() { return fail(NIDX_ERR_IO, "paragraphs.bin: truncated record %u", a); };
        if (!read_varint(data, dlen, at, n) || n > dlen - at || n > 0xffffffffull) return bad();
        pg.key = {at, (uint32_t)n};
        at += n;
        if (!read_varint(data, dlen, at, n) || n > dlen - at) return bad();  // every label takes at least its length byte
        pg.first_label = (uint32_t)d->labels.size();
        pg.n_labels = (uint32_t)n;
        for (uint64_t i = 0; i < n; i++) {
            if (!read_varint(data, dlen, at, v) || v > dlen - at || v > 0xffffffffull) return bad();
            d->labels.push_back({at, (uint32_t)v});
            at += v;
        }
        if (!read_varint(data, dlen, at, n) || n > dlen - at || n > 0xffffffffull) return bad();
        pg.metadata = {at, (uint32_t)n};
        at += n;
        uint64_t fv, nv;
        if (!read_varint(data, dlen, at, fv) || !read_varint(data, dlen, at, nv)) return bad();
        pg.first_vector = (uint32_t)fv;
        pg.num_vectors = (uint32_t)nv;
        if (fv > d->n_vectors || nv > d->n_vectors - fv) return fail(NIDX_ERR_IO, "paragraph %u owns vectors beyond vectors.bin", a);
        d->key_ids[a] = key_id(data + pg.key.off, pg.key.len);
        if (!rebuild) continue;
        lb.key(data + pg.key.off, pg.key.len, a);
        for (uint32_t i = 0; i < pg.n_labels; i++) lb.label(data + d->labels[pg.first_label + i].off, d->labels[pg.first_label + i].len, a);
    }
    if (rebuild) {
        d->list_offsets.push_back(0);
        for (auto &kv : lb.lists) {
            d->list_keys.push_back(kv.first);
            d->list_ids.insert(d->list_ids.end(), kv.second.begin(), kv.second.end());
            d->list_offsets.push_back(d->list_ids.size());
            if (kv.first[0] == 'L') d->n_label_lists++;
        }
    }
    *dir_out = reinterpret_cast<nidx_gpu_segment_dir_t *>(d.release());
    return NIDX_OK;
} NIDX_ABI_CATCH

void nidx_gpu_segment_dir_close(nidx_gpu_segment_dir_t *dir) { delete reinterpret_cast<SegmentDir *>(dir); }

int32_t nidx_gpu_segment_dir_index_source(const nidx_gpu_segment_dir_t *dir) {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    return d ? (d->lists_from_files ? 1 : 0) : -1;
}

int32_t nidx_gpu_segment_dir_segment(const nidx_gpu_segment_dir_t *dir, nidx_gpu_vector_segment_t *out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    memset(out, 0, sizeof(*out));
    out->vectors = d->vectors.p;
    out->row_stride_bytes = d->row_stride;
    out->n_vectors = d->n_vectors;
    out->paragraph_of_vector = nullptr;  // the row trailers
    out->n_paragraphs = d->n_paragraphs;
    out->hnsw_graph = d->graph.len ? d->graph.p : nullptr;
    out->hnsw_graph_len = d->graph.len;
    out->hnsw_edges = d->edges.len ? reinterpret_cast<const float *>(d->edges.p) : nullptr;
    out->n_hnsw_edges = d->edges.len / 4;
    out->paragraph_key_ids = d->n_paragraphs ? d->key_ids.data() : nullptr;
    out->quantized = d->quant.len ? d->quant.p : nullptr;
    out->quantized_len = d->quant.len;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_segment_dir_filter_index(const nidx_gpu_segment_dir_t *dir, nidx_gpu_filter_index_t *out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    out->n_lists = (uint32_t)d->list_keys.size();
    out->list_offsets = d->list_offsets.data();
    out->paragraph_ids = d->list_ids.data();
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_segment_dir_lists(const nidx_gpu_segment_dir_t *dir, int32_t kind, const uint8_t *key, uint32_t key_len, int32_t prefix,
                                   uint32_t *first_out, uint32_t *count_out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !first_out || !count_out || (key_len && !key)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *first_out = *count_out = 0;
    std::string k;
    if (kind == NIDX_LIST_LABEL) {
        // labels_key (inverted_index/paragraph.rs:63-66); the lookup is always a prefix search (:144-146)
        if (key_len == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "empty label");
        k = "L" + std::string(reinterpret_cast<const char *>(key) + 1, key_len - 1) + "/";
        prefix = 1;
    } else if (kind == NIDX_LIST_FIELD) {
        std::string fk;
        if (!field_key(key, key_len, fk)) return NIDX_OK;  // from_field_id -> None: the id selects nothing
        k = "F" + fk;
    } else {
        return fail(NIDX_ERR_INVALID_ARGUMENT, "unknown list kind %d", kind);
    }
    auto lo = std::lower_bound(d->list_keys.begin(), d->list_keys.end(), k);
    auto hi = lo;
    if (prefix) {
        while (hi != d->list_keys.end() && hi->compare(0, k.size(), k) == 0) ++hi;
    } else if (hi != d->list_keys.end() && *hi == k) {
        ++hi;
    }
    *first_out = (uint32_t)(lo - d->list_keys.begin());
    *count_out = (uint32_t)(hi - lo);
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_segment_dir_paragraph(const nidx_gpu_segment_dir_t *dir, uint32_t addr, nidx_gpu_paragraph_t *out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (addr >= d->n_paragraphs) return fail(NIDX_ERR_INVALID_ARGUMENT, "paragraph %u out of range (%u stored)", addr, d->n_paragraphs);
    const Paragraph &pg = d->paragraphs[addr];
    out->key = reinterpret_cast<const char *>(d->para_data.p + pg.key.off);
    out->key_len = pg.key.len;
    out->metadata = d->para_data.p + pg.metadata.off;
    out->metadata_len = pg.metadata.len;
    out->n_labels = pg.n_labels;
    out->first_vector = pg.first_vector;
    out->num_vectors = pg.num_vectors;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_segment_dir_paragraph_label(const nidx_gpu_segment_dir_t *dir, uint32_t addr, uint32_t i, const char **label_out,
                                             uint32_t *len_out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !label_out || !len_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (addr >= d->n_paragraphs || i >= d->paragraphs[addr].n_labels) return fail(NIDX_ERR_INVALID_ARGUMENT, "label %u of paragraph %u out of range", i, addr);
    const Span &l = d->labels[d->paragraphs[addr].first_label + i];
    *label_out = reinterpret_cast<const char *>(d->para_data.p + l.off);
    *len_out = l.len;
    return NIDX_OK;
} NIDX_ABI_CATCH

static int write_file(const std::string &path, const void *p, size_t n) { return write_bytes(path, p, n); }

int32_t nidx_gpu_segment_dir_write(const char *path, const nidx_gpu_segment_dir_contents_t *c) try {
    if (!path || !c || c->dimension == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (c->n_vectors && !c->vectors) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL vectors");
    if (c->n_paragraphs && (!c->key_offsets || !c->keys)) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL keys");
    const std::string base = std::string(path) + "/";
    const uint32_t D = c->dimension;
    // first vector / count of every paragraph: vectors of one paragraph are contiguous (segment.rs:216-229)
    std::vector<uint32_t> first(c->n_paragraphs, 0), num(c->n_paragraphs, 0);
    for (uint32_t v = 0; v < c->n_vectors; v++) {
        const uint32_t p = c->paragraph_of_vector ? c->paragraph_of_vector[v] : v;
        if (p >= c->n_paragraphs) return fail(NIDX_ERR_INVALID_ARGUMENT, "vector %u belongs to paragraph %u of %u", v, p, c->n_paragraphs);
        if (num[p] == 0) first[p] = v;
        else if (first[p] + num[p] != v) return fail(NIDX_ERR_INVALID_ARGUMENT, "the vectors of paragraph %u are not contiguous", p);
        num[p]++;
    }
    {   // vectors.bin, streamed a few MiB at a time (a 10 M x 768 segment is 30 GB: no second copy in host memory)
        const size_t stride = (size_t)D * 4 + 4;
        const uint32_t chunk = (uint32_t)std::max<size_t>(1, (4u << 20) / stride);
        std::vector<uint8_t> rows((size_t)std::min(chunk, std::max(c->n_vectors, 1u)) * stride);
        FILE *f = fopen((base + "vectors.bin").c_str(), "wb");
        bool ok = f != nullptr;
        for (uint32_t v0 = 0; ok && v0 < c->n_vectors; v0 += chunk) {
            const uint32_t nv = std::min(chunk, c->n_vectors - v0);
            for (uint32_t i = 0; i < nv; i++) {
                const uint32_t v = v0 + i, p = c->paragraph_of_vector ? c->paragraph_of_vector[v] : v;
                memcpy(rows.data() + (size_t)i * stride, c->vectors + (size_t)v * D, (size_t)D * 4);
                memcpy(rows.data() + (size_t)i * stride + (size_t)D * 4, &p, 4);
            }
            ok = fwrite(rows.data(), stride, nv, f) == nv;
        }
        if (f && fclose(f) != 0) ok = false;
        if (!ok) return fail(NIDX_ERR_IO, "cannot write %svectors.bin", base.c_str());
    }
    {   // paragraphs.bin + paragraphs.pos
        std::vector<uint8_t> data;
        std::vector<uint32_t> pos(c->n_paragraphs);
        ListBuilder built;
        for (uint32_t a = 0; a < c->n_paragraphs; a++) {
            if (data.size() > 0xffffffffull) return fail(NIDX_ERR_UNSUPPORTED, "paragraphs.bin would exceed the 4 GiB its u32 offsets address");
            pos[a] = (uint32_t)data.size();
            const uint64_t kb = c->key_offsets[a], ke = c->key_offsets[a + 1];
            write_varint(data, ke - kb);
            data.insert(data.end(), c->keys + kb, c->keys + ke);
            built.key(c->keys + kb, ke - kb, a);
            const uint64_t lb = c->paragraph_label_offsets ? c->paragraph_label_offsets[a] : 0, le = c->paragraph_label_offsets ? c->paragraph_label_offsets[a + 1] : 0;
            write_varint(data, le - lb);
            for (uint64_t l = lb; l < le; l++) {
                write_varint(data, c->label_offsets[l + 1] - c->label_offsets[l]);
                data.insert(data.end(), c->labels + c->label_offsets[l], c->labels + c->label_offsets[l + 1]);
                built.label(c->labels + c->label_offsets[l], c->label_offsets[l + 1] - c->label_offsets[l], a);
            }
            const uint64_t mb = c->metadata_offsets ? c->metadata_offsets[a] : 0, me = c->metadata_offsets ? c->metadata_offsets[a + 1] : 0;
            write_varint(data, me - mb);
            if (me > mb) data.insert(data.end(), c->metadata + mb, c->metadata + me);
            write_varint(data, first[a]);
            write_varint(data, num[a]);
        }
        if (write_file(base + "paragraphs.bin", data.data(), data.size())) return fail(NIDX_ERR_IO, "cannot write %sparagraphs.bin", base.c_str());
        if (write_file(base + "paragraphs.pos", pos.data(), pos.size() * 4)) return fail(NIDX_ERR_IO, "cannot write %sparagraphs.pos", base.c_str());
        if (fst_files_written() && write_index_files(base, built.lists)) return fail(NIDX_ERR_IO, "cannot write the inverted indexes under %s", base.c_str());
    }
    if (c->quantized && c->quantized_len) {
        if (c->quantized_len != (uint64_t)c->n_vectors * (D / 8 + 8)) return fail(NIDX_ERR_INVALID_ARGUMENT, "quantized store has the wrong size");
        if (write_file(base + "vectors.quant", c->quantized, c->quantized_len)) return fail(NIDX_ERR_IO, "cannot write %svectors.quant", base.c_str());
    }
    if (c->hnsw_graph && c->hnsw_graph_len) {
        if (write_file(base + "hnsw.graph", c->hnsw_graph, c->hnsw_graph_len)) return fail(NIDX_ERR_IO, "cannot write %shnsw.graph", base.c_str());
        if (write_file(base + "hnsw.edges", c->hnsw_edges, (size_t)c->n_hnsw_edges * 4)) return fail(NIDX_ERR_IO, "cannot write %shnsw.edges", base.c_str());
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

namespace {
struct OutFile {
    FILE *f = nullptr;
    bool ok = true;
    explicit OutFile(const std::string &path) : f(fopen(path.c_str(), "wb")) { ok = f != nullptr; }
    OutFile(const OutFile &) = delete;
    ~OutFile() { if (f) fclose(f); }
    void put(const void *p, size_t n) { if (ok && n) ok = fwrite(p, 1, n, f) == n; }
    bool finish() {
        if (f && fclose(f) != 0) ok = false;
        f = nullptr;
        return ok;
    }
};
}  // namespace

int32_t nidx_gpu_segment_dir_merge(const char *path, uint32_t dimension, const nidx_gpu_merge_operand_t *operands, uint32_t n_operands,
                                   uint32_t *records_out, uint32_t *vectors_out, uint32_t *graph_nodes_out, int32_t *has_quantized_out) try {
    if (!path || !operands || dimension == 0) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_operands == 0) return fail(NIDX_ERR_EMPTY_MERGE, "Can not merge zero segments");
    // segment::merge (segment.rs:92-94): largest operand first, so that as much of its graph as possible is reused
    std::vector<uint32_t> order(n_operands);
    for (uint32_t i = 0; i < n_operands; i++) {
        if (!operands[i].dir) return fail(NIDX_ERR_INVALID_ARGUMENT, "operand %u: NULL directory", i);
        if (reinterpret_cast<const SegmentDir *>(operands[i].dir)->dimension != dimension)
            return fail(NIDX_ERR_INCONSISTENT_DIMENSIONS, "operand %u was opened with another dimension", i);
        order[i] = i;
    }
    auto dir_of = [&](uint32_t i) { return reinterpret_cast<const SegmentDir *>(operands[i].dir); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return dir_of(a)->n_paragraphs > dir_of(b)->n_paragraphs; });
    auto alive = [&](uint32_t i, uint32_t a) {
        const uint64_t *w = operands[i].alive_bitset;
        return !w || ((w[a >> 6] >> (a & 63)) & 1);
    };
    bool all_quant = true;
    for (uint32_t i = 0; i < n_operands; i++) all_quant = all_quant && (dir_of(i)->quant.present || dir_of(i)->n_vectors == 0);
    const bool quantizable = dimension % 64 == 0;  // the caller only passes stores with codes for Dot indexes (config.rs:170-173)
    const bool write_quant = quantizable && all_quant && [&] { for (uint32_t i = 0; i < n_operands; i++) if (dir_of(i)->quant.present) return true; return false; }();
    const std::string base = std::string(path) + "/";
    const size_t row_bytes = (size_t)dimension * 4, qrec = (size_t)dimension / 8 + 8;
    OutFile vec(base + "vectors.bin"), pdata(base + "paragraphs.bin"), ppos(base + "paragraphs.pos");
    std::unique_ptr<OutFile> quant;
    if (write_quant) quant.reset(new OutFile(base + "vectors.quant"));
    // DataStoreV2::merge (data_store/v2.rs:82-128): the alive paragraphs of every operand in address order, each with its
    // vectors (their trailer = the paragraph's new address) and, when stored, their RaBitQ records
    uint64_t p_idx = 0, v_idx = 0, data_len = 0;
    std::vector<uint8_t> rec;
    ListBuilder lb;   // segment::create_indexes builds the inverted indexes of the merged store (segment.rs:231-236)
    for (uint32_t oi : order) {
        const SegmentDir *d = dir_of(oi);
        const uint8_t *data = d->para_data.p;
        for (uint32_t a = 0; a < d->n_paragraphs; a++) {
            if (!alive(oi, a)) continue;
            const Paragraph &pg = d->paragraphs[a];
            if (p_idx > 0xfffffffeull || v_idx + pg.num_vectors > 0xffffffffull) return fail(NIDX_ERR_UNSUPPORTED, "merged segment exceeds 2^32 records");
            if (data_len > 0xffffffffull) return fail(NIDX_ERR_UNSUPPORTED, "paragraphs.bin would exceed the 4 GiB its u32 offsets address");
            const uint32_t trailer = (uint32_t)p_idx, pos = (uint32_t)data_len;
            for (uint32_t v = 0; v < pg.num_vectors; v++) {
                vec.put(d->vectors.p + (size_t)(pg.first_vector + v) * d->row_stride, row_bytes);
                vec.put(&trailer, 4);
                if (quant) quant->put(d->quant.p + (size_t)(pg.first_vector + v) * qrec, qrec);
            }
            rec.clear();
            write_varint(rec, pg.key.len);
            rec.insert(rec.end(), data + pg.key.off, data + pg.key.off + pg.key.len);
            lb.key(data + pg.key.off, pg.key.len, trailer);
            write_varint(rec, pg.n_labels);
            for (uint32_t l = 0; l < pg.n_labels; l++) {
                const Span &sp = d->labels[pg.first_label + l];
                write_varint(rec, sp.len);
                rec.insert(rec.end(), data + sp.off, data + sp.off + sp.len);
                lb.label(data + sp.off, sp.len, trailer);
            }
            write_varint(rec, pg.metadata.len);
            rec.insert(rec.end(), data + pg.metadata.off, data + pg.metadata.off + pg.metadata.len);
            write_varint(rec, v_idx);
            write_varint(rec, pg.num_vectors);
            pdata.put(rec.data(), rec.size());
            ppos.put(&pos, 4);
            data_len += rec.size();
            v_idx += pg.num_vectors;
            p_idx++;
        }
    }
    if (!vec.finish() || !pdata.finish() || !ppos.finish() || (quant && !quant->finish())) return fail(NIDX_ERR_IO, "cannot write the merged segment under %s", base.c_str());
    if (fst_files_written() && write_index_files(base, lb.lists)) return fail(NIDX_ERR_IO, "cannot write the inverted indexes under %s", base.c_str());
    // merge_indexes (segment.rs:137-167): the largest operand's graph is reused when none of its paragraphs is deleted — its
    // vectors are then the first rows of the merged store; the caller extends it (nidx_gpu_vector_extend_hnsw)
    const SegmentDir *first = dir_of(order[0]);
    uint32_t first_alive = 0;
    for (uint32_t a = 0; a < first->n_paragraphs; a++) first_alive += alive(order[0], a) ? 1 : 0;
    uint32_t graph_nodes = 0;
    if (first_alive == first->n_paragraphs && first->graph.len) {
        if (write_file(base + "hnsw.graph", first->graph.p, first->graph.len) || write_file(base + "hnsw.edges", first->edges.p, first->edges.len))
            return fail(NIDX_ERR_IO, "cannot write %shnsw.graph", base.c_str());
        graph_nodes = first->n_vectors;
    }
    if (records_out) *records_out = (uint32_t)p_idx;
    if (vectors_out) *vectors_out = (uint32_t)v_idx;
    if (graph_nodes_out) *graph_nodes_out = graph_nodes;
    if (has_quantized_out) *has_quantized_out = write_quant ? 1 : 0;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_segment_dir_apply_deletions(const nidx_gpu_segment_dir_t *dir, const char *const *keys, const uint32_t *key_lens, uint32_t n_keys,
                                             uint64_t *alive_bitset, uint32_t *n_cleared_out) try {
    const SegmentDir *d = reinterpret_cast<const SegmentDir *>(dir);
    if (!d || !alive_bitset || (n_keys && (!keys || !key_lens))) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    uint32_t cleared = 0;
    std::string fk;
    for (uint32_t i = 0; i < n_keys; i++) {
        if (!keys[i] && key_lens[i]) return fail(NIDX_ERR_INVALID_ARGUMENT, "key %u is NULL", i);
        // FieldKey::from_field_id -> None: not a field id, nothing to delete (lib.rs:193-195)
        if (!field_key(reinterpret_cast<const uint8_t *>(keys[i]), key_lens[i], fk)) continue;
        const std::string k = "F" + fk;
        // field_index.get_prefix (inverted_index/paragraph.rs:118-120): every indexed key the FieldKey bytes are a prefix of
        for (auto it = std::lower_bound(d->list_keys.begin(), d->list_keys.end(), k); it != d->list_keys.end() && it->compare(0, k.size(), k) == 0; ++it) {
            const size_t l = (size_t)(it - d->list_keys.begin());
            for (uint64_t j = d->list_offsets[l]; j < d->list_offsets[l + 1]; j++) {
                const uint32_t a = d->list_ids[j];
                const uint64_t bit = 1ull << (a & 63);
                if (alive_bitset[a >> 6] & bit) { alive_bitset[a >> 6] &= ~bit; cleared++; }
            }
        }
    }
    if (n_cleared_out) *n_cleared_out = cleared;
    return NIDX_OK;
} NIDX_ABI_CATCH

// ---- the containers themselves (tooling / tests; fst_index.cpp) ----
int32_t nidx_gpu_fst_map_build(const uint8_t *keys, const uint64_t *key_offsets, const uint64_t *values, uint32_t n, uint8_t *out, uint64_t cap, uint64_t *len_out) try {
    if (!len_out || (n && (!keys || !key_offsets || !values))) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<std::pair<std::string, uint64_t>> entries(n);
    for (uint32_t i = 0; i < n; i++) entries[i] = {std::string(reinterpret_cast<const char *>(keys + key_offsets[i]), key_offsets[i + 1] - key_offsets[i]), values[i]};
    std::vector<uint8_t> image;
    if (!fst_build(entries, image)) return fail(NIDX_ERR_INVALID_ARGUMENT, "keys must be strictly ascending");
    *len_out = image.size();
    if (out && cap >= image.size()) memcpy(out, image.data(), image.size());
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_fst_map_get(const uint8_t *image, uint64_t len, const uint8_t *key, uint32_t key_len, uint64_t *value_out, int32_t *found_out) try {
    if (!image || !value_out || !found_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    *found_out = fst_get(image, len, key, key_len, value_out) ? 1 : 0;
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_fst_map_entries(const uint8_t *image, uint64_t len, uint8_t *keys_out, uint64_t keys_cap, uint64_t *key_offsets_out, uint64_t *values_out,
                                 uint32_t cap, uint32_t *n_out, uint64_t *keys_len_out) try {
    if (!image || !n_out || !keys_len_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<std::pair<std::string, uint64_t>> entries;
    if (!fst_enumerate(image, len, entries)) return fail(NIDX_ERR_IO, "not a well-formed fst image");
    uint64_t total = 0;
    for (const auto &e : entries) total += e.first.size();
    *n_out = (uint32_t)entries.size();
    *keys_len_out = total;
    if (keys_out && key_offsets_out && values_out && cap >= entries.size() && keys_cap >= total) {
        uint64_t at = 0;
        for (size_t i = 0; i < entries.size(); i++) {
            key_offsets_out[i] = at;
            memcpy(keys_out + at, entries[i].first.data(), entries[i].first.size());
            at += entries[i].first.size();
            values_out[i] = entries[i].second;
        }
        key_offsets_out[entries.size()] = at;
    }
    return NIDX_OK;
} NIDX_ABI_CATCH

int32_t nidx_gpu_index_map_read(const uint8_t *map, uint64_t len, uint64_t pos, uint32_t *ids_out, uint32_t cap, uint32_t *n_out) try {
    if (!map || !n_out) return fail(NIDX_ERR_INVALID_ARGUMENT, "NULL argument");
    std::vector<uint32_t> ids;
    if (!map_read(map, len, pos, ids)) return fail(NIDX_ERR_IO, "no record at offset %llu", (unsigned long long)pos);
    *n_out = (uint32_t)ids.size();
    if (ids_out && cap >= ids.size() && !ids.empty()) memcpy(ids_out, ids.data(), ids.size() * 4);
    return NIDX_OK;
} NIDX_ABI_CATCH

}  // extern "C"
