// fst_index.h — readers/writers of a vector segment's field.fst / label.fst / index.map (see fst_index.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace nidx {

// index.map (inverted_index/map.rs:48-86): append one record, returns its offset / read the record at `pos`
uint64_t map_append(std::vector<uint8_t> &out, const uint32_t *ids, size_t n);
bool map_read(const uint8_t *data, size_t len, uint64_t pos, std::vector<uint32_t> &out);

// fst::Map image from (key, value) pairs with strictly ascending keys (fst_index.rs:40-50); false: keys out of order
bool fst_build(const std::vector<std::pair<std::string, uint64_t>> &entries, std::vector<uint8_t> &out);
// every (key, value) of an image in key order (fst_index.rs:76-86 with an empty prefix); false: not a well-formed image, or
// one that claims more than max_keys keys
bool fst_enumerate(const uint8_t *data, size_t len, std::vector<std::pair<std::string, uint64_t>> &out, uint64_t max_keys = 1ull << 28);
// fst_index.rs:66-74; false: key absent (or image malformed)
bool fst_get(const uint8_t *data, size_t len, const uint8_t *key, size_t key_len, uint64_t *value_out);
bool fst_check_sum(const uint8_t *data, size_t len);

}  // namespace nidx
