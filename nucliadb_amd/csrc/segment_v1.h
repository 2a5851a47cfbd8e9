// segment_v1.h — DataStoreV1 (nodes.kv) / DiskHnswV1 (index.hnsw) images migrated in memory to the current layouts (segment_v1.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nidx {

// nodes.kv -> the vectors.bin (dimension f32 + u32 paragraph address per row), paragraphs.bin and paragraphs.pos images of the same
// records (one vector per paragraph, in address order).  0, or -1 with a message.
int migrate_nodes_kv(const uint8_t *kv, size_t len, uint32_t dimension, std::vector<uint8_t> &vectors_bin, std::vector<uint8_t> &paragraphs_bin,
                     std::vector<uint8_t> &paragraphs_pos, std::string &err);
// index.hnsw -> the hnsw.graph image and the hnsw.edges weights of the same graph (DiskHnswV1::deserialize, then DiskHnswV2::serialize_into)
int migrate_index_hnsw(const uint8_t *buf, size_t len, uint32_t n_nodes, std::vector<uint8_t> &graph, std::vector<float> &edges, std::string &err);

}  // namespace nidx
