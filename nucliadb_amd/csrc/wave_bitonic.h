// wave_bitonic.h — the bitonic network on one 64-bit rank key per lane of a wave (gfx950): compare-exchange partners come through
// v_permlane32/16_swap and DPP row moves, never through LDS.  Used by bm25_stream_kernel (candidate buffer -> top-k list) and
// bm25_merge_kernel (the slices of a query).  Keys are the rank keys of device_common.h: larger = better, NIDX_EMPTY_KEY smallest.
#pragma once
#include "device_common.h"

namespace nidx {

// compare-exchange with lane ^ J: lanes whose bit of TAKE_MAX is set keep the larger key, the others the smaller one
template <int J>
__device__ inline uint64_t bs_cmpx(uint64_t v, unsigned long long take_max_mask) {
    const bool tm = __builtin_amdgcn_inverse_ballot_w64(take_max_mask);
    if constexpr (J >= 16) {
        uint32_t a0 = (uint32_t)v, a1 = a0, b0 = (uint32_t)(v >> 32), b1 = b0;
        if constexpr (J == 32) {
            swap_pair32(a0, a1);
            swap_pair32(b0, b1);
        } else {
            swap_pair16(a0, a1);
            swap_pair16(b0, b1);
        }
        const uint64_t x = ((uint64_t)b0 << 32) | a0, y = ((uint64_t)b1 << 32) | a1;   // {own, partner} in some order
        return ((x > y) == tm) ? x : y;
    } else {
        const uint64_t p = ((uint64_t)xor_partner_dpp<J>((uint32_t)(v >> 32)) << 32) | xor_partner_dpp<J>((uint32_t)v);
        return ((v > p) == tm) ? v : p;
    }
}
constexpr unsigned long long bs_sort_mask(int K, int J) {   // ascending sort, stage K, substep J: who keeps the larger key
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++)
        if (((l & J) != 0) != ((l & K) != 0)) m |= 1ull << l;
    return m;
}
constexpr unsigned long long bs_merge_mask(int J) {   // descending merge: the lane with the J bit clear keeps the larger key
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++)
        if ((l & J) == 0) m |= 1ull << l;
    return m;
}
template <int K, int J>
__device__ inline uint64_t bs_sort_steps(uint64_t v) {
    v = bs_cmpx<J>(v, bs_sort_mask(K, J));
    if constexpr (J > 1) return bs_sort_steps<K, J / 2>(v);
    else return v;
}
template <int K>
__device__ inline uint64_t bs_sort_stages(uint64_t v) {   // stages 2 .. K
    if constexpr (K > 2) v = bs_sort_stages<K / 2>(v);
    return bs_sort_steps<K, K / 2>(v);
}
template <int J>
__device__ inline uint64_t bs_merge_steps(uint64_t v) {
    v = bs_cmpx<J>(v, bs_merge_mask(J));
    if constexpr (J > 1) return bs_merge_steps<J / 2>(v);
    else return v;
}
// sorted (best first) list `top` of 64 keys and 64 unsorted keys `v` -> the 64 best of the 128, sorted
__device__ inline uint64_t bs_merge64(uint64_t top, uint64_t v) {
    v = bs_sort_stages<64>(v);            // ascending
    const uint64_t m = top > v ? top : v; // a descending and an ascending run, element by element: bitonic, holds the 64 best
    return bs_merge_steps<32>(m);
}

// two sorted (best first) lists, the second one handed over REVERSED (worst first; e.g. read from LDS at 63 - lane) -> the 64 best, sorted
__device__ inline uint64_t bs_merge_sorted(uint64_t top, uint64_t other_reversed) {
    const uint64_t m = top > other_reversed ? top : other_reversed;
    return bs_merge_steps<32>(m);
}

}  // namespace nidx
