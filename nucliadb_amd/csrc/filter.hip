// filter.hip — label / key-prefix filter formulas evaluated on the device (gfx950).
//
// Replaces ParagraphInvertedIndexes::filter / filter_clause (nidx_vector/src/inverted_index/paragraph.rs:
// 124-184): every atom of a formula is a union of posting lists (paragraph addresses), compounds are
// AND / OR / NOT over bitsets of n_paragraphs bits, the result is intersected with the alive bitset
// and its popcount ("matching", segment.rs:516-531) routes the search (use_hnsw).  The posting lists
// of a segment live in HBM (CSR); the host only resolves strings to list ids (the FST lookups of
// inverted_index/fst_index.rs stay host side) and sends a postfix program.  Pure HBM-bound bit work:
// n_paragraphs/8 bytes per operator, 4 bytes per posting.
#include "device_common.h"
#include "kernels.h"

namespace nidx {

__global__ void bitset_fill_kernel(uint64_t *out, uint32_t n_words, uint32_t n_bits, int ones) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = ones ? ~0ull : 0ull;
    if (ones && w == n_words - 1 && (n_bits & 63)) v = (1ull << (n_bits & 63)) - 1ull;
    out[w] = v;
}

// out |= bits of every id in the given lists.  One block per (list, chunk of 1024 postings).
__global__ void bitset_scatter_kernel(const unsigned long long *__restrict__ list_offsets, const uint32_t *__restrict__ ids,
                                      const uint32_t *__restrict__ lists, uint32_t n_lists, uint32_t n_bits,
                                      unsigned int *__restrict__ out32) {
    for (uint32_t l = blockIdx.y; l < n_lists; l += gridDim.y) {
        const unsigned long long b = list_offsets[lists[l]], e = list_offsets[lists[l] + 1];
        for (unsigned long long i = b + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < e;
             i += (unsigned long long)gridDim.x * blockDim.x) {
            uint32_t id = ids[i];
            if (id < n_bits) atomicOr(&out32[id >> 5], 1u << (id & 31));
        }
    }
}

// op: 0 and, 1 or, 2 and-not (a &= ~b)
__global__ void bitset_binop_kernel(uint64_t *a, const uint64_t *b, uint32_t n_words, int op) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    a[w] = op == 0 ? (a[w] & b[w]) : op == 1 ? (a[w] | b[w]) : (a[w] & ~b[w]);
}

__global__ void bitset_not_kernel(uint64_t *a, uint32_t n_words, uint32_t n_bits) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t v = ~a[w];
    if (w == n_words - 1 && (n_bits & 63)) v &= (1ull << (n_bits & 63)) - 1ull;
    a[w] = v;
}

// out = a & alive (alive may be null); *count += popcount(out)
__global__ void bitset_and_count_kernel(const uint64_t *a, const uint64_t *alive, uint64_t *out, uint32_t n_words,
                                        unsigned long long *count) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t v = 0;
    if (w < n_words) {
        v = a[w];
        if (alive) v &= alive[w];
        out[w] = v;
    }
    unsigned long long c = (unsigned long long)__popcll(v);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

static inline dim3 words_grid(uint32_t n_words) { return dim3((n_words + 255) / 256); }

hipError_t launch_bitset_fill(uint64_t *out, uint32_t n_words, uint32_t n_bits, int ones, hipStream_t s) {
    if (!n_words) return hipSuccess;
    hipLaunchKernelGGL(bitset_fill_kernel, words_grid(n_words), dim3(256), 0, s, out, n_words, n_bits, ones);
    return hipGetLastError();
}
hipError_t launch_bitset_scatter(const unsigned long long *list_offsets, const uint32_t *ids, const uint32_t *lists,
                                 uint32_t n_lists, uint32_t n_bits, uint64_t *out, hipStream_t s) {
    if (!n_lists) return hipSuccess;
    dim3 grid(64, n_lists < 1024 ? n_lists : 1024);
    hipLaunchKernelGGL(bitset_scatter_kernel, grid, dim3(256), 0, s, list_offsets, ids, lists, n_lists, n_bits,
                       reinterpret_cast<unsigned int *>(out));
    return hipGetLastError();
}
hipError_t launch_bitset_binop(uint64_t *a, const uint64_t *b, uint32_t n_words, int op, hipStream_t s) {
    if (!n_words) return hipSuccess;
    hipLaunchKernelGGL(bitset_binop_kernel, words_grid(n_words), dim3(256), 0, s, a, b, n_words, op);
    return hipGetLastError();
}
hipError_t launch_bitset_not(uint64_t *a, uint32_t n_words, uint32_t n_bits, hipStream_t s) {
    if (!n_words) return hipSuccess;
    hipLaunchKernelGGL(bitset_not_kernel, words_grid(n_words), dim3(256), 0, s, a, n_words, n_bits);
    return hipGetLastError();
}
hipError_t launch_bitset_and_count(const uint64_t *a, const uint64_t *alive, uint64_t *out, uint32_t n_words,
                                   unsigned long long *count, hipStream_t s) {
    if (!n_words) return hipSuccess;
    hipLaunchKernelGGL(bitset_and_count_kernel, words_grid(n_words), dim3(256), 0, s, a, alive, out, n_words, count);
    return hipGetLastError();
}

}  // namespace nidx
