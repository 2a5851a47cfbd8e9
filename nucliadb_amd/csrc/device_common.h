// device_common.h — shared device helpers for the gfx950 kernels of libnidx_gpu.
//
// Numerics contract (DESIGN.md "numerics"): every inner product on the scan/HNSW path is summed in
// the WAVE64 order — lane l owns elements j*256+4l..+3 (one 16-byte load per lane per 1 KiB row
// chunk), an fmaf chain in (j, component) order per lane, then a 6-step xor butterfly (offsets
// 32,16,8,4,2,1) so that every lane holds the same f32 sum.  The file is compiled with
// -ffp-contract=off: every FMA below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NIDX_WAVE 64

namespace nidx {

// f32::total_cmp key (Rust): monotone int32 image of an f32 bit pattern.
__host__ __device__ inline int32_t total_key(float f) {
    int32_t b = __builtin_bit_cast(int32_t, f);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

// Strict total order used wherever the reference leaves ties to container internals
// (BinaryHeap on a score-only Ord, sort_unstable): higher score first, then LOWER address.
// Packed as one u64 so that a > b  <=>  a ranks before b.
__host__ __device__ inline uint64_t rank_key(float score, uint32_t addr) {
    uint32_t k = (uint32_t)total_key(score) ^ 0x80000000u;  // unsigned-monotone
    return ((uint64_t)k << 32) | (uint64_t)(~addr);
}
__host__ __device__ inline float rank_key_score(uint64_t key) {
    uint32_t k = (uint32_t)(key >> 32) ^ 0x80000000u;
    int32_t b = (int32_t)k;
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return __builtin_bit_cast(float, b);
}
__host__ __device__ inline uint32_t rank_key_addr(uint64_t key) { return ~(uint32_t)key; }
// The empty slot: ranks after every real entry (real keys have addr != 0xffffffff or score bits > 0).
#define NIDX_EMPTY_KEY 0ull

// ---- x[l] + x[l ^ OFF] without the LDS crossbar -------------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (an address register, an LDS-pipe round trip of ~100 cycles, six of them in a row for
// one wave sum).  gfx950 can do every level of the xor butterfly in the VALU: 32 and 16 with v_permlane32_swap /
// v_permlane16_swap (swap(x, y) leaves [x.lower, y.lower] in x and [x.upper, y.upper] in y — halves of 32 lanes, resp. 16-lane
// rows), 8 with the DPP row rotation by 8, 4 with row_half_mirror followed by the quad reversal (l ^ 7 ^ 3 = l ^ 4), 2 and 1 with
// quad permutations.  The operand pairs are exactly those of the shuffle form (float add is commutative), so results are
// bit-identical.
// (inline asm for the swaps: with ROCm 7.2's hipcc the __builtin_amdgcn_permlane32_swap / 16_swap builtins use the first result
// for both elements of the returned pair, i.e. x + x; the s_nop 1 is the two wait states hipcc itself puts between a VALU write
// of an operand and the swap — it cannot see inside the asm; results may be read right away.)
__device__ inline float swap_add32(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ inline float swap_add16(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
template <int CTRL>
__device__ inline float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int OFF>
__device__ inline float xor_add(float x) {
    static_assert(OFF == 32 || OFF == 16 || OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "xor_add: power of two below 64");
    if constexpr (OFF == 32) return swap_add32(x, x);
    else if constexpr (OFF == 16) return swap_add16(x, x);
    else if constexpr (OFF == 8) return x + dpp_move<0x128>(x);                   // row_ror:8
    else if constexpr (OFF == 4) return x + dpp_move<0x1B>(dpp_move<0x141>(x));   // row_half_mirror, then quad_perm:[3,2,1,0]
    else if constexpr (OFF == 2) return x + dpp_move<0x4E>(x);                    // quad_perm:[2,3,0,1]
    else return x + dpp_move<0xB1>(x);                                            // quad_perm:[1,0,3,2]
}
// levels OFF, OFF / 2, .., 1 of the butterfly
template <int OFF>
__device__ inline float xor_tail(float x) {
    x = xor_add<OFF>(x);
    if constexpr (OFF > 1) return xor_tail<OFF / 2>(x);
    else return x;
}

// xor butterfly: every lane ends with the same value, pairing a[l] + a[l^off] at each level.
__device__ inline float wave_butterfly_sum(float v) { return xor_tail<32>(v); }

// The same VALU-only butterfly for order statistics.  combine(x[l], x[l ^ OFF]) for a commutative combine: levels 32 and 16 hand
// both operands over as the two results of a permlane swap of the value with itself, the levels below as a DPP move.
__device__ inline void swap_pair32(uint32_t &x, uint32_t &y) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
__device__ inline void swap_pair16(uint32_t &x, uint32_t &y) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
template <int CTRL>
__device__ inline uint32_t dpp_move_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false); }
template <int OFF>
__device__ inline uint32_t xor_partner_dpp(uint32_t x) {   // x[l ^ OFF] for OFF in {8, 4, 2, 1}
    if constexpr (OFF == 8) return dpp_move_u32<0x128>(x);
    else if constexpr (OFF == 4) return dpp_move_u32<0x1B>(dpp_move_u32<0x141>(x));
    else if constexpr (OFF == 2) return dpp_move_u32<0x4E>(x);
    else return dpp_move_u32<0xB1>(x);
}
// all-lanes reduction of a 64-bit value with a commutative, associative combine
template <typename F>
__device__ inline uint64_t wave_reduce_u64(uint64_t v, F pick) {
    {
        uint32_t a0 = (uint32_t)v, a1 = a0, b0 = (uint32_t)(v >> 32), b1 = b0;
        swap_pair32(a0, a1);
        swap_pair32(b0, b1);
        v = pick(((uint64_t)b0 << 32) | a0, ((uint64_t)b1 << 32) | a1);
    }
    {
        uint32_t a0 = (uint32_t)v, a1 = a0, b0 = (uint32_t)(v >> 32), b1 = b0;
        swap_pair16(a0, a1);
        swap_pair16(b0, b1);
        v = pick(((uint64_t)b0 << 32) | a0, ((uint64_t)b1 << 32) | a1);
    }
    v = pick(v, ((uint64_t)xor_partner_dpp<8>((uint32_t)(v >> 32)) << 32) | xor_partner_dpp<8>((uint32_t)v));
    v = pick(v, ((uint64_t)xor_partner_dpp<4>((uint32_t)(v >> 32)) << 32) | xor_partner_dpp<4>((uint32_t)v));
    v = pick(v, ((uint64_t)xor_partner_dpp<2>((uint32_t)(v >> 32)) << 32) | xor_partner_dpp<2>((uint32_t)v));
    v = pick(v, ((uint64_t)xor_partner_dpp<1>((uint32_t)(v >> 32)) << 32) | xor_partner_dpp<1>((uint32_t)v));
    return v;
}
template <bool MAX>
__device__ inline uint64_t wave_extreme_u64(uint64_t v) {
    return wave_reduce_u64(v, [](uint64_t a, uint64_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); });
}
template <typename F>
__device__ inline uint32_t wave_reduce_u32(uint32_t x, F pick) {
    uint32_t a0 = x, a1 = x;
    swap_pair32(a0, a1);
    x = pick(a0, a1);
    a0 = a1 = x;
    swap_pair16(a0, a1);
    x = pick(a0, a1);
    x = pick(x, xor_partner_dpp<8>(x));
    x = pick(x, xor_partner_dpp<4>(x));
    x = pick(x, xor_partner_dpp<2>(x));
    x = pick(x, xor_partner_dpp<1>(x));
    return x;
}
__device__ inline int wave_min_i32(int x) {
    uint32_t a0 = (uint32_t)x, a1 = a0;
    swap_pair32(a0, a1);
    x = (int)a0 < (int)a1 ? (int)a0 : (int)a1;
    a0 = a1 = (uint32_t)x;
    swap_pair16(a0, a1);
    x = (int)a0 < (int)a1 ? (int)a0 : (int)a1;
    int o;
    o = (int)xor_partner_dpp<8>((uint32_t)x); x = o < x ? o : x;
    o = (int)xor_partner_dpp<4>((uint32_t)x); x = o < x ? o : x;
    o = (int)xor_partner_dpp<2>((uint32_t)x); x = o < x ? o : x;
    o = (int)xor_partner_dpp<1>((uint32_t)x); x = o < x ? o : x;
    return x;
}
// value of lane `src` (wave-uniform index) through the scalar unit
__device__ inline uint32_t lane_bcast_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ inline uint64_t lane_bcast_u64(uint64_t v, int src) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
}
__device__ inline float lane_bcast_f32(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); }

__device__ inline uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint64_t shfl_up_u64(uint64_t v, int delta) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return ((uint64_t)hi << 32) | lo;
}
// lane l gets lane l - 1's value (lane 0 keeps its own): the wavefront shift of the gfx9 DPP unit (wave_shr:1) instead of two
// ds_bpermute_b32 round trips — the move every sorted-list insertion makes
__device__ inline uint64_t wave_shr1_u64(uint64_t v) {
    const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return ((uint64_t)h2 << 32) | l2;
}

// dense_f32::cosine_similarity (vector_types/dense_f32.rs:29-34) on pre-summed f32 terms:
// SimSIMD distance semantics (zero-norm cases, clamp >= 0) evaluated in f64, then `1.0 - (d as f32)`.
__host__ __device__ inline float cosine_from_sums(float ab, float xx, float yy) {
    double dab = (double)ab, dxx = (double)xx, dyy = (double)yy;
    double dist;
    if (dxx == 0.0 && dyy == 0.0) {
        dist = 0.0;
    } else if (dab == 0.0) {
        dist = 1.0;
    } else {
        double d = 1.0 - dab / (sqrt(dxx) * sqrt(dyy));
        dist = d > 0.0 ? d : 0.0;
    }
    return 1.0f - (float)dist;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a full workgroup fence: it also waits for every
// GLOBAL load in flight (s_waitcnt vmcnt(0)), which serialises a software pipeline whose prefetched rows are meant to stay
// in flight across the barrier.  Here only the LDS queue is drained; the compiler still waits for a register's own load
// before its first use.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline bool bit_test(const uint64_t *bits, uint32_t i) { return (bits[i >> 6] >> (i & 63)) & 1ull; }

// A sorted (best first) list of up to 64 (score, addr) entries held one per lane as rank keys.
// insert() keeps the 64 best; returns the key that fell off the end (NIDX_EMPTY_KEY if none).
struct WaveSortedList {
    uint64_t key;  // lane i holds rank i
    __device__ inline void init() { key = NIDX_EMPTY_KEY; }
    __device__ inline uint64_t insert(uint64_t nk, int lane) {
        // position = number of entries ranking before nk
        unsigned long long before = __ballot(key > nk);
        int pos = __popcll(before);
        // lane 63's key through the scalar unit (v_readlane) rather than the LDS crossbar
        uint64_t dropped = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), 63) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, 63);
        uint64_t up = wave_shr1_u64(key);
        if (lane > pos) key = up;
        if (lane == pos) key = nk;
        return pos >= 64 ? nk : dropped;
    }
    __device__ inline uint64_t at(int rank) const { return lane_bcast_u64(key, rank); }  // rank is wave-uniform at every call site
};

// Up to 64*NL best entries: NL sorted lists chained (what falls off list i is inserted into list i+1).
// `len` is the number of live entries; insert() with cap < 64*NL evicts the worst when full.
template <int NL>
struct WaveTopK {
    WaveSortedList l[NL];
    int len;
    __device__ inline void init() {
#pragma unroll
        for (int i = 0; i < NL; i++) l[i].init();
        len = 0;
    }
    __device__ inline uint64_t at(int rank) const {
        uint64_t v = l[0].at(rank & 63);
#pragma unroll
        for (int i = 1; i < NL; i++) {
            uint64_t t = l[i].at(rank & 63);
            if ((rank >> 6) == i) v = t;
        }
        return v;
    }
    // entry of rank 64*i + lane, for every i (this lane's slice of the lists)
    __device__ inline uint64_t mine(int i) const { return l[i].key; }
    __device__ inline void insert(uint64_t nk, int cap, int lane) {
        uint64_t d = l[0].insert(nk, lane);
#pragma unroll
        for (int i = 1; i < NL; i++)
            if (d != NIDX_EMPTY_KEY) d = l[i].insert(d, lane);
        len++;
        if (len > cap) {  // drop rank `cap`
#pragma unroll
            for (int i = 0; i < NL; i++)
                if ((cap >> 6) == i && lane == (cap & 63)) l[i].key = NIDX_EMPTY_KEY;
            len = cap;
        }
    }
    // insert while tracking only the k best (no len bookkeeping): returns the new k-th key
    __device__ inline uint64_t insert_kth(uint64_t nk, int k, int lane) {
        uint64_t d = l[0].insert(nk, lane);
#pragma unroll
        for (int i = 1; i < NL; i++)
            if (d != NIDX_EMPTY_KEY) d = l[i].insert(d, lane);
        return at(k - 1);
    }
    __device__ inline float worst_score() const { return rank_key_score(at(len - 1)); }
};

// ---- transposed multi-value butterfly -------------------------------------------------------
// Reduces QT per-lane partial sums across the wave with one shuffle per PAIR of values at the
// first log2(QT) levels.  Value v ends up (identically) in every lane l with query_of_lane(l) == v,
// and equals bit for bit what wave_butterfly_sum would give for it (same pairs a[l] + a[l^off]).
template <int QT>
struct QReduce;

template <>
struct QReduce<1> {
    static __device__ inline float run(float (&a)[1], int) { return wave_butterfly_sum(a[0]); }
    static __device__ inline int query_of_lane(int) { return 0; }
    static __device__ inline int group_mask() { return 63; }
};

template <int QT>
struct QReduce {
    // Halving levels: at offset `off` lanes with (lane & off) keep the upper half of the query set.  Levels 32 and 16: one
    // swap_add per PAIR of values — x + y after the swap is "x[l] + x[l ^ off]" in the lanes that keep x and "y[l] + y[l ^ off]"
    // in the lanes that keep y: no selects at all.  Below that: both butterflies in the VALU (xor_add) and one select.  (A
    // generic "keep / send / shuffle" formulation compiled to ~40 compare + select instructions per value.)
    template <int N, int OFF>
    static __device__ inline void level(float (&v)[QT], int lane) {
        if constexpr (N > 1) {
            if constexpr (OFF == 32) {
#pragma unroll
                for (int i = 0; i < N / 2; i++) v[i] = swap_add32(v[i], v[i + N / 2]);
            } else if constexpr (OFF == 16) {
#pragma unroll
                for (int i = 0; i < N / 2; i++) v[i] = swap_add16(v[i], v[i + N / 2]);
            } else {
                const bool up = (lane & OFF) != 0;
#pragma unroll
                for (int i = 0; i < N / 2; i++) {
                    const float lo = xor_add<OFF>(v[i]), hi = xor_add<OFF>(v[i + N / 2]);
                    v[i] = up ? hi : lo;   // x[l] + x[l ^ off] is the same value in l and l ^ off
                }
            }
            level<N / 2, OFF / 2>(v, lane);
        } else {
            if constexpr (OFF >= 1) v[0] = xor_tail<OFF>(v[0]);
        }
    }
    static __device__ inline float run(float (&a)[QT], int lane) {
        float v[QT];
#pragma unroll
        for (int i = 0; i < QT; i++) v[i] = a[i];
        level<QT, 32>(v, lane);
        return v[0];
    }
    static __device__ inline int query_of_lane(int lane) {
        int q = 0, off = 32;
#pragma unroll
        for (int n = QT; n > 1; n >>= 1, off >>= 1)
            if (lane & off) q += n / 2;
        return q;
    }
    // lanes with (lane & group_mask()) == 0 lead a group that shares one query's value
    static __device__ inline int group_mask() { return (64 / QT) - 1; }
};

__device__ inline float4 load_row_chunk(const float *row, uint32_t dp, int j, int lane) {
    uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
    if (e < dp) return *reinterpret_cast<const float4 *>(row + e);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ inline float fma4(const float4 &x, const float4 &y, float acc) {
    acc = fmaf(x.x, y.x, acc);
    acc = fmaf(x.y, y.y, acc);
    acc = fmaf(x.z, y.z, acc);
    acc = fmaf(x.w, y.w, acc);
    return acc;
}

}  // namespace nidx
