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


// ---- the posting lists' KEY TABLE (what label.fst / field.fst resolve): sorted byte strings, key j names posting list j.
// One thread per query: [first, last) = the table entries equal to the query (exact) or starting with it (prefix): two
// binary searches over byte strings in HBM.  A prefilter hands over thousands of field ids at once; their lookups run side
// by side here instead of one FST walk after the other on the host.
__device__ inline int key_cmp(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb) {   // bytewise, shorter first on a tie
    const uint32_t n = la < lb ? la : lb;
    for (uint32_t i = 0; i < n; i++)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return la == lb ? 0 : (la < lb ? -1 : 1);
}
__device__ inline bool key_starts_with(const uint8_t *k, uint32_t lk, const uint8_t *p, uint32_t lp) {
    if (lk < lp) return false;
    for (uint32_t i = 0; i < lp; i++)
        if (k[i] != p[i]) return false;
    return true;
}
__global__ void key_range_kernel(const uint8_t *tbl, const unsigned long long *tbl_off, uint32_t n_keys, const uint8_t *qb,
                                 const unsigned long long *q_off, const uint8_t *q_prefix, uint32_t n_q, uint32_t *first, uint32_t *last) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    const uint8_t *p = qb + q_off[q];
    const uint32_t lp = (uint32_t)(q_off[q + 1] - q_off[q]);
    uint32_t lo = 0, hi = n_keys;
    while (lo < hi) {   // lower bound: first key >= query
        const uint32_t mid = lo + (hi - lo) / 2;
        if (key_cmp(tbl + tbl_off[mid], (uint32_t)(tbl_off[mid + 1] - tbl_off[mid]), p, lp) < 0) lo = mid + 1;
        else hi = mid;
    }
    const uint32_t f = lo;
    uint32_t l = f;
    if (q_prefix[q]) {
        hi = n_keys;      // keys with the prefix are contiguous from f: first key beyond them
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (key_starts_with(tbl + tbl_off[mid], (uint32_t)(tbl_off[mid + 1] - tbl_off[mid]), p, lp)) lo = mid + 1;
            else hi = mid;
        }
        l = lo;
    } else if (f < n_keys && key_cmp(tbl + tbl_off[f], (uint32_t)(tbl_off[f + 1] - tbl_off[f]), p, lp) == 0) {
        l = f + 1;
    }
    first[q] = f;
    last[q] = l;
}
hipError_t launch_key_range(const uint8_t *tbl, const unsigned long long *tbl_off, uint32_t n_keys, const uint8_t *qb, const unsigned long long *q_off,
                            const uint8_t *q_prefix, uint32_t n_q, uint32_t *first, uint32_t *last, hipStream_t s) {
    if (!n_q) return hipSuccess;
    hipLaunchKernelGGL(key_range_kernel, dim3((n_q + 127) / 128), dim3(128), 0, s, tbl, tbl_off, n_keys, qb, q_off, q_prefix, n_q, first, last);
    return hipGetLastError();
}

}  // namespace nidx
