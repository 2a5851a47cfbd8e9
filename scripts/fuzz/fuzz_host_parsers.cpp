// fuzz_host_parsers.cpp — AddressSanitizer / UBSan run over the parsers of untrusted files: the segment-directory reader
// (csrc/segment_dir.cpp) and the hnsw.graph validator (csrc/hnsw_graph.cpp), host code only.  Built and driven by
// scripts/fuzz/run.sh: a seed directory written by the Python mirror is mutated (bytes, words, truncations) and every mutant
// is opened and — when accepted — walked through every accessor and merged with itself.  Any out-of-bounds read aborts.
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <random>
#include <string>
#include <vector>

#include "../../include/nidx_gpu.h"

static std::vector<uint8_t> slurp(const std::string &p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}
static void spit(const std::string &p, const std::vector<uint8_t> &v) {
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) { perror(p.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
}

static std::mt19937_64 rng(getenv("NIDX_FUZZ_SEED") ? strtoull(getenv("NIDX_FUZZ_SEED"), nullptr, 10) : 12345);
static void mutate(std::vector<uint8_t> &v) {
    if (v.empty()) return;
    const int kind = (int)(rng() % 5);
    const size_t at = rng() % v.size();
    static const uint32_t magic[] = {0, 1, 4, 250, 251, 252, 253, 254, 255, 0x7fffffffu, 0xffffffffu, 1u << 16};
    if (kind == 0) v[at] = (uint8_t)rng();
    else if (kind == 1) v[at] ^= (uint8_t)(1u << (rng() % 8));
    else if (kind == 2 && v.size() >= 4) { uint32_t w = magic[rng() % 12]; memcpy(&v[(at & ~(size_t)3) % (v.size() - 3)], &w, 4); }
    else if (kind == 3) v.resize(at);
    else { v[at] = (uint8_t)magic[rng() % 9]; }
}

static uint64_t walk(nidx_gpu_segment_dir_t *d, const std::string &scratch, uint32_t dim) {
    uint64_t acc = 0;
    nidx_gpu_vector_segment_t seg;
    nidx_gpu_filter_index_t fi;
    if (nidx_gpu_segment_dir_segment(d, &seg) != 0 || nidx_gpu_segment_dir_filter_index(d, &fi) != 0) abort();
    for (uint32_t l = 0; l < fi.n_lists; l++)
        for (uint64_t j = fi.list_offsets[l]; j < fi.list_offsets[l + 1]; j++) acc += fi.paragraph_ids[j];
    std::vector<uint64_t> alive((seg.n_paragraphs + 63) / 64 + 1, ~0ull);
    for (uint32_t a = 0; a < seg.n_paragraphs; a++) {
        nidx_gpu_paragraph_t p;
        if (nidx_gpu_segment_dir_paragraph(d, a, &p) != 0) abort();
        for (uint32_t i = 0; i < p.key_len; i++) acc += (uint8_t)p.key[i];
        for (uint32_t i = 0; i < p.metadata_len; i++) acc += p.metadata[i];
        for (uint32_t i = 0; i < p.n_labels; i++) {
            const char *lab;
            uint32_t len, first, count;
            if (nidx_gpu_segment_dir_paragraph_label(d, a, i, &lab, &len) != 0) abort();
            for (uint32_t j = 0; j < len; j++) acc += (uint8_t)lab[j];
            if (len) nidx_gpu_segment_dir_lists(d, NIDX_LIST_LABEL, (const uint8_t *)lab, len, 1, &first, &count);
        }
        uint32_t first, count, cleared;
        nidx_gpu_segment_dir_lists(d, NIDX_LIST_FIELD, (const uint8_t *)p.key, p.key_len, a & 1, &first, &count);
        const char *k = p.key;
        if ((a % 7) == 0) nidx_gpu_segment_dir_apply_deletions(d, &k, &p.key_len, 1, alive.data(), &cleared);
        // every vector row and its trailer
        for (uint32_t v = 0; v < p.num_vectors; v++) {
            const uint8_t *row = (const uint8_t *)seg.vectors + (size_t)(p.first_vector + v) * seg.row_stride_bytes;
            acc += row[0] + row[seg.row_stride_bytes - 1];
        }
    }
    nidx_gpu_merge_operand_t ops[2] = {{d, alive.data()}, {d, nullptr}};
    uint32_t rec, vec, gn;
    int32_t hq;
    nidx_gpu_segment_dir_merge(scratch.c_str(), dim, ops, 2, &rec, &vec, &gn, &hq);
    return acc + rec;
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s seed_dir scratch_dir dimension iterations\n", argv[0]); return 2; }
    const std::string seed = argv[1], scratch = argv[2];
    const uint32_t dim = (uint32_t)atoi(argv[3]);
    const int iters = atoi(argv[4]);
    const char *names[] = {"vectors.bin", "paragraphs.bin", "paragraphs.pos", "vectors.quant", "hnsw.graph", "hnsw.edges", "field.fst", "label.fst", "index.map"};
    const size_t n_names = sizeof(names) / sizeof(names[0]);
    std::vector<std::vector<uint8_t>> good;
    for (const char *n : names) good.push_back(slurp(seed + "/" + n));
    const std::string mdir = scratch + "/mutant", odir = scratch + "/merged";
    mkdir(mdir.c_str(), 0755);
    mkdir(odir.c_str(), 0755);
    uint64_t accepted = 0, refused = 0, acc = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<std::vector<uint8_t>> files = good;
        if (it) {
            const int n_mut = 1 + (int)(rng() % 3);
            for (int m = 0; m < n_mut; m++) mutate(files[1 + rng() % 2 + (rng() % 4 == 0 ? 2 : 0)]);  // mostly the paragraph store
            if (rng() % 8 == 0) mutate(files[0]);
            if (rng() % 3 == 0) mutate(files[6 + rng() % 3]);   // the inverted-index files (an fst that fails its checksum => rebuild)
        }
        for (size_t i = 0; i < n_names; i++) spit(mdir + "/" + names[i], files[i]);
        nidx_gpu_segment_dir_t *d = nullptr;
        const int32_t rc = nidx_gpu_segment_dir_open(mdir.c_str(), dim, &d);
        if (rc == 0) { accepted++; acc += walk(d, odir, dim); nidx_gpu_segment_dir_close(d); }
        else { refused++; if (it == 0) { fprintf(stderr, "the seed directory was refused (%d)\n", rc); return 1; } }
        // the graph validator on an exact-size heap copy: an over-read is an ASan report
        std::vector<uint8_t> g = good[4];
        if (it) { mutate(g); if (rng() % 2) mutate(g); }
        uint8_t *exact = g.empty() ? nullptr : (uint8_t *)malloc(g.size());
        if (exact) memcpy(exact, g.data(), g.size());
        const uint32_t n_nodes = (uint32_t)(good[0].size() / ((size_t)dim * 4 + 4));
        uint32_t en, el;
        uint64_t nl, nb;
        const int32_t grc = nidx_gpu_hnsw_graph_check(exact, g.size(), (const float *)good[5].data(), good[5].size() / 4, n_nodes, &en, &el, &nl, &nb);
        if (it == 0 && grc != 0) { fprintf(stderr, "the seed graph was refused (%d)\n", grc); return 1; }
        free(exact);
        // the fst reader (no checksum gate here) and the index.map record reader on exact-size heap copies
        std::vector<uint8_t> f = good[6 + it % 2];
        if (it) { mutate(f); if (rng() % 2) mutate(f); }
        uint8_t *fx = (uint8_t *)malloc(f.size() ? f.size() : 1);
        memcpy(fx, f.data(), f.size());
        uint32_t n_keys = 0;
        uint64_t keys_len = 0;
        const int32_t frc = nidx_gpu_fst_map_entries(fx, f.size(), nullptr, 0, nullptr, nullptr, 0, &n_keys, &keys_len);
        if (it < 2 && (frc != 0 || n_keys == 0)) { fprintf(stderr, "the seed fst was refused (%d)\n", frc); return 1; }
        uint64_t v = 0;
        int32_t found = 0;
        nidx_gpu_fst_map_get(fx, f.size(), (const uint8_t *)"set/1/", 6, &v, &found);
        acc += n_keys + (uint64_t)found;
        free(fx);
        std::vector<uint8_t> m = good[8];
        if (it) mutate(m);
        uint8_t *mx = (uint8_t *)malloc(m.size() ? m.size() : 1);
        memcpy(mx, m.data(), m.size());
        uint32_t n_ids = 0;
        std::vector<uint32_t> ids(64);
        if (nidx_gpu_index_map_read(mx, m.size(), it ? rng() % (m.size() + 8) : 0, ids.data(), 64, &n_ids) == 0) acc += n_ids;
        free(mx);
    }
    // pre-migration directories: nodes.kv + index.hnsw mutants through the same entry point
    uint64_t v1_accepted = 0, v1_refused = 0;
    if (argc > 5) {
        const std::string seed1 = argv[5], vdir = scratch + "/mutant_v1";
        mkdir(vdir.c_str(), 0755);
        const char *vnames[] = {"nodes.kv", "index.hnsw"};
        std::vector<std::vector<uint8_t>> vgood;
        for (const char *n : vnames) vgood.push_back(slurp(seed1 + "/" + n));
        for (int it = 0; it < iters; it++) {
            std::vector<std::vector<uint8_t>> files = vgood;
            if (it) {
                const int n_mut = 1 + (int)(rng() % 3);
                for (int m = 0; m < n_mut; m++) mutate(files[rng() % 2]);
            }
            for (size_t i = 0; i < 2; i++) spit(vdir + "/" + vnames[i], files[i]);
            nidx_gpu_segment_dir_t *d = nullptr;
            const int32_t rc = nidx_gpu_segment_dir_open(vdir.c_str(), dim, &d);
            if (rc == 0) { v1_accepted++; acc += walk(d, odir, dim); nidx_gpu_segment_dir_close(d); }
            else { v1_refused++; if (it == 0) { fprintf(stderr, "the pre-migration seed directory was refused (%d)\n", rc); return 1; } }
        }
        printf("pre-migration directories: accepted %llu refused %llu\n", (unsigned long long)v1_accepted, (unsigned long long)v1_refused);
    }
    printf("iterations %d: directories accepted %llu refused %llu (checksum %llu)\n", iters, (unsigned long long)accepted, (unsigned long long)refused,
           (unsigned long long)acc);
    return 0;
}
