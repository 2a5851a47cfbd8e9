// vector_bf16.hip — batched brute-force FALLBACK on the bf16 matrix cores with exact f32 re-scoring
// (gfx950).  BASELINE.json configs[4]; SURVEY.md §7 step 5.
//
// Stage 1 (bf16_scan_kernel): S~ = Q~·X~ᵀ with bf16 copies of queries and corpus on
// v_mfma_f32_32x32x16_bf16 (f32 accumulate), fused per-query top-K' by the APPROXIMATE score
// (K' = 32 candidates per query and stripe).  The bf16 corpus copy is half the bytes of the f32 one
// and the matrix cores run 16x the f32 rate, so the corpus streams once per batch at ~HBM speed.
// Stage 2 (merge_topk_kernel, K' per query) + Stage 3 (rescore_select_kernel): the K' survivors of a
// query are re-scored from the f32 rows in the WAVE64 order of the scan/HNSW kernels and the final
// top-k is selected by those exact scores.  Returned scores are therefore bit-identical to
// brute_force_search's for the returned ids; the id set equals the exact top-k unless a true top-k
// row fell outside the approximate top-K' (bf16 rounding: relative 2^-9 per operand) — recall is
// measured, not assumed (tests/test_vector_gpu.py).  This method is explicit
// (NIDX_METHOD_BRUTE_FORCE_BF16), never chosen by the cost model.
// Stage 1 has two forms with the same candidates: bf16_scan_kernel keeps sorted lists in LDS (any input, no floor needed);
// bf16_append_kernel (round 4, below) needs a floor per query, keeps no lists and runs the wider pipeline — the host
// (vector_index.cpp) seeds floors with the former on a prefix, tightens them with the latter on a sample, and falls back to the
// former per query block.
#include "device_common.h"
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace nidx {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BF_BM 256                /* queries per workgroup */
#define BF_BN 256                /* corpus rows per tile */
#define BF_BK 16                 /* bf16 elements per K step = one 32-byte LDS row */
#define BF_BLOCK_BYTES (256 * BF_BK * 2)     /* one operand block: 256 rows x 32 B = 8 KiB */
#define BF_STAGE_BYTES (2 * BF_BLOCK_BYTES)  /* [Q block][X block] */
#define BF_STAGES 5
#define BF_KP NIDX_BF16_CAND     /* candidates kept per query and list */
#define BF_THREADS 512

// Operand layout.  Both bf16 operands are stored TILED in HBM: for a tile of 256 rows and K step kc (16 elements) the 8-KiB
// block [256 rows][2 chunks of 16 B], blocks ordered [tile][kc].  A block is exactly the LDS image of that K step, bank swizzle
// included (chunk position p of row r holds K chunk p ^ ((r >> 3) & 1): the 16 lanes of a ds_read_b128 phase then hit 16
// distinct 16-byte bank groups), so staging is a linear copy: global_load_lds_dwordx4, 1 KiB per wave instruction, every
// 128-byte line of the stream used in full.  Cosine indexes store x / |x| (and q / |q|), so the accumulator IS the approximate
// score; rows outside the filter / alive set / past n are masked by a per-call bitset (bf16_row_mask_kernel).
//
// Tiling.  One workgroup of 8 waves per CU owns 256 queries and walks its stripe of 256-row corpus tiles; wave w holds the
// 32 x 256 block of scores of queries 32 w.. against the whole tile as 8 accumulators of v_mfma_f32_32x32x16_bf16 (128
// VGPRs): a K step costs 1 + 8 fragment reads (ds_read_b128) for 8 MFMAs, and — the point of giving a wave its own queries —
// the candidate lists of a query are touched by exactly one wave, so they stay in LDS without locks.
//
// Pipeline.  The K steps of all the tiles of a stripe form one stream; step g lives in stage g % 5 and its two 8-KiB blocks
// are requested FOUR steps ahead (2 DMA pieces per wave and step).  Per step: s_waitcnt vmcnt(6) — this wave's pieces of step g
// have landed, three later steps stay in flight — then one raw s_barrier (everybody's pieces landed; everybody is done reading
// step g - 1), then the request for step g + 4 into the stage step g - 1 used, then the fragment reads and MFMAs.  Nothing in
// the loop waits for vmcnt(0), so HBM / L2 latency is covered by four steps of MFMA work; the epilogue between two tiles uses
// LDS and scalar loads only and leaves the DMA queue alone.
// LDS: 80 KiB stages + 64 KiB lists + 3 KiB thresholds.
struct Bf16Shared {
    __attribute__((aligned(16))) unsigned char stage[BF_STAGES][BF_STAGE_BYTES];
    uint64_t lists[BF_BM][BF_KP];
    uint64_t thr_key[BF_BM];
    float thr_score[BF_BM];
};

__device__ inline unsigned short f32_to_bf16_rne(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// rows [n][dp] f32 -> tiled bf16 blocks [ceil(n / 256)][dp16 / 16][256][16] (see "Operand layout"); rows past n and columns past
// dp are zero; norm2 != nullptr: rows are scaled by 1 / sqrt(norm2[row]) (cosine).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void to_bf16_tiled_kernel(const float *__restrict__ in, const float *__restrict__ norm2, uint32_t n, uint32_t dp,
                                                            uint32_t dp16, unsigned short *__restrict__ out) {
    const uint32_t nk = dp16 / BF_BK;
    const size_t n_chunks = (size_t)((n + 255u) / 256u) * nk * 512u;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (size_t)gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)(c & 1u), rr = (uint32_t)((c >> 1) & 255u);
        const size_t blk = c >> 9;
        const uint32_t kc = (uint32_t)(blk % nk);
        const uint32_t row = (uint32_t)(blk / nk) * 256u + rr;
        const uint32_t k0 = kc * BF_BK + 8u * (p ^ ((rr >> 3) & 1u));
        unsigned short v[8];
        float scale = 1.0f;
        if (row < n && norm2) {
            const float nn = norm2[row];
            scale = nn > 0.f ? 1.0f / sqrtf(nn) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (row < n && k0 + e < dp) ? f32_to_bf16_rne(in[(size_t)row * dp + k0 + e] * scale) : (unsigned short)0;
        uint4 o;
        o.x = v[0] | ((uint32_t)v[1] << 16);
        o.y = v[2] | ((uint32_t)v[3] << 16);
        o.z = v[4] | ((uint32_t)v[5] << 16);
        o.w = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4 *>(out + c * 8) = o;
    }
}

// bit r of the mask = row r takes part: r < n, its paragraph alive and inside the filter
__global__ __launch_bounds__(256) void bf16_row_mask_kernel(uint32_t n, uint32_t n_pad, const uint32_t *__restrict__ para_of_vec,
                                                            const uint64_t *__restrict__ alive, const uint64_t *__restrict__ filter,
                                                            uint64_t *__restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = r < n;
    if (ok) {
        const uint32_t p = para_of_vec ? para_of_vec[r] : r;
        if (alive && !bit_test(alive, p)) ok = false;
        if (ok && filter && !bit_test(filter, p)) ok = false;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && r < n_pad) out[r >> 6] = m;
}

template <int AUX>
__device__ inline void bf_glds16(const unsigned char *src, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, AUX);
}

// One 32 x 32 accumulator (the wave's 32 queries qb.., tile rows jb..) -> the candidate lists of its queries.  A lane holds
// tile row jb + li against the 16 queries qb + (r & 3) + 8 (r >> 2) + 4 half.  ok_word: the 32 row-mask bits of jb.. .
__device__ __forceinline__ void bf_epilogue_tile(const floatx16 &acc, float (&thr)[16], Bf16Shared &sh, uint32_t ok_word, uint32_t r0, int qb, int jb,
                                                 int lane) {
    const int li = lane & 31, half = lane >> 5;
    const uint32_t row = r0 + (uint32_t)(jb + li);
    const bool row_ok = (ok_word >> li) & 1u;
    // thr[r]: this lane's copy of the threshold of query qb + (r & 3) + 8 (r >> 2) + 4 half, loaded once per tile; a stale value
    // only lets a few extra candidates through to the exact key test below; padding queries carry +inf
    uint32_t mask = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) mask |= (row_ok && acc[r] > thr[r]) ? (1u << r) : 0u;
    if (!__ballot(mask != 0)) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const unsigned long long m = __ballot((mask >> r) & 1u);
        if (!m) continue;
        const uint64_t key = rank_key(acc[r], row);
        // the lanes of one half share the query: its list is loaded once, takes every candidate of the group in registers, and
        // goes back once
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned long long mh = m & (h ? 0xffffffff00000000ull : 0x00000000ffffffffull);
            if (!mh) continue;
            const int sq = qb + (r & 3) + 8 * (r >> 2) + 4 * h;
            WaveSortedList l;
            l.key = lane < BF_KP ? sh.lists[sq][lane] : NIDX_EMPTY_KEY;
            uint64_t kth = sh.thr_key[sq];
            bool changed = false;
            while (mh) {
                const int src_lane = __ffsll((long long)mh) - 1;
                mh &= mh - 1;
                const uint64_t nk_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), src_lane) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, src_lane);
                if (!(nk_ > kth)) continue;
                l.insert(nk_, lane);
                const uint64_t last = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(l.key >> 32), BF_KP - 1) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)l.key, BF_KP - 1);
                if (last != NIDX_EMPTY_KEY) kth = last;
                changed = true;
            }
            if (changed) {
                if (lane < BF_KP) sh.lists[sq][lane] = l.key;
                if (lane == 0 && kth != NIDX_EMPTY_KEY) {
                    sh.thr_key[sq] = kth;
                    sh.thr_score[sq] = fmaxf(sh.thr_score[sq], rank_key_score(kth));   // never below the sample floor
                }
            }
        }
        thr[r] = sh.thr_score[qb + (r & 3) + 8 * (r >> 2) + 4 * half];   // what this group may just have raised
    }
}

__global__ __launch_bounds__(BF_THREADS, 1) void bf16_scan_kernel(Bf16ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bf_smem[];
    Bf16Shared &sh = *reinterpret_cast<Bf16Shared *>(bf_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const uint32_t q0 = blockIdx.y * BF_BM;
    const uint32_t n_tiles = (a.n + BF_BN - 1) / BF_BN;
    const uint32_t nk = a.dp16 / BF_BK;
    // the exact second opinion behind bf16_append_kernel: only the query blocks whose append pass ran out of slots are scanned again
    if (a.run_if && a.run_if[blockIdx.y] == 0) return;

    if (tid < BF_BM) {
        const bool real = q0 + tid < a.n_queries && !(a.debug & 1);
        sh.thr_key[tid] = real ? NIDX_EMPTY_KEY : ~0ull;     // padding queries admit nothing
        // sample pass: BF_KP rows reach floor_score, so a row strictly below it is not among the query's BF_KP best; the test
        // below is `>`, hence the largest float under the floor
        float t0 = -INFINITY;
        if (real && a.floor_score) {
            const float f = a.floor_score[q0 + tid];
            if (f > -INFINITY) {
                const int32_t k = total_key(f) - 1;                                               // the float just below f in total order
                t0 = __builtin_bit_cast(float, k ^ (int32_t)(((uint32_t)(k >> 31)) >> 1));   // total_key is its own inverse
            }
        }
        sh.thr_score[tid] = real ? t0 : INFINITY;
    }
    for (int i = tid; i < BF_BM * BF_KP; i += BF_THREADS) (&sh.lists[0][0])[i] = NIDX_EMPTY_KEY;

    // the stream of K steps of this workgroup's tiles: tile ordinal i -> tile blockIdx.x + i * gridDim.x
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t G = my_tiles * nk;
    // DMA role of this wave: waves 0-3 copy the query block, 4-7 the corpus block; two 1-KiB pieces each.  The wave keeps one
    // running source pointer: + 8 KiB per step; when a tile's last step has been requested the query waves rewind to the first
    // block, the corpus waves jump to the stripe's next tile (or rewind on the last one: the requests run ahead of the work and
    // simply re-read valid blocks into stages nobody looks at again — that keeps "four steps in flight" true to the very end, so
    // the loop needs a single s_waitcnt vmcnt(6) and no tail cases).
    const bool q_role = wave < 4;
    const long long blk_tile = (long long)nk * BF_BLOCK_BYTES;                 // bytes of one tile's blocks
    const long long wrap_back = -blk_tile;
    const long long wrap_fwd = q_role ? -blk_tile : (long long)(gridDim.x - 1) * blk_tile;
    const unsigned char *ptr = (q_role ? reinterpret_cast<const unsigned char *>(a.queries16) + (size_t)blockIdx.y * blk_tile
                                       : reinterpret_cast<const unsigned char *>(a.vectors16) + (size_t)blockIdx.x * blk_tile) +
                               (uint32_t)((wave & 3) * 2) * 1024u + (uint32_t)lane * 16u;
    const uint32_t dst_wave = (q_role ? 0u : (uint32_t)BF_BLOCK_BYTES) + (uint32_t)(wave & 3) * 2048u;
    uint32_t is_kc = 0, is_tile = blockIdx.x;   // the next step to request
    uint32_t wr_stage = 0, rd_stage = 0;        // byte offsets of the stage written next / read next
    auto issue = [&]() __attribute__((always_inline)) {
        unsigned char *dst = &sh.stage[0][0] + wr_stage + dst_wave;
        bf_glds16<0>(ptr, dst);   // (the non-temporal hint, aux = 2, on the corpus stream measured 9 % slower)
        bf_glds16<0>(ptr + 1024, dst + 1024);
        ptr += BF_BLOCK_BYTES;
        wr_stage = wr_stage + BF_STAGE_BYTES == BF_STAGES * BF_STAGE_BYTES ? 0 : wr_stage + BF_STAGE_BYTES;
        if (++is_kc == nk) {
            is_kc = 0;
            if (is_tile + gridDim.x < n_tiles) {
                is_tile += gridDim.x;
                ptr += wrap_fwd;
            } else {
                ptr += wrap_back;
            }
        }
    };
    __syncthreads();  // lists / thresholds initialised
    if (G)
        for (int i = 0; i < BF_STAGES - 1; i++) issue();

    // fragment addresses inside a stage: row r = 32 B, chunk position = half ^ ((r >> 3) & 1)
    const int frag = (half ^ ((li >> 3) & 1)) * 16;
    const int a_off = (32 * wave + li) * 32 + frag, b_off = BF_BLOCK_BYTES + li * 32 + frag;

    // Next step: wait until its blocks are in LDS for everybody, request the step four ahead, read its nine fragments.  The
    // fragment reads of the previous step (issued one call earlier, into the other register set) are drained first: the request
    // below reuses the stage they came from.
    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    auto mma = [&](const bf16x8 &av, const bf16x8 (&bv)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[t], acc[t], 0, 0, 0);
    };
    // The two waves of a SIMD (w and w + 4) take the two halves of a step in opposite order, so that one of them feeds the
    // matrix core while the other one is busy with the barrier, the DMA requests and its LDS reads: waves 0-3 run the previous
    // step's MFMAs first and read afterwards, waves 4-7 read first.
    auto step = [&](auto mma_first_tag, bf16x8 &av, bf16x8 (&bv)[8], bf16x8 &pav, bf16x8 (&pbv)[8], bool with_mma) __attribute__((always_inline)) {
        constexpr bool mma_first = decltype(mma_first_tag)::value;
        __builtin_amdgcn_sched_barrier(0);  // the MFMAs issued before this call stay before it
        // s_waitcnt vmcnt(6) lgkmcnt(0) — this wave's two pieces of the step have landed, three later steps stay in flight — as
        // the builtin (gfx9 encoding: vmcnt [3:0], expcnt [6:4] = 7 "no wait", lgkmcnt [11:8]), so that the compiler's own
        // wait-count bookkeeping knows the LDS queue is empty here
        __builtin_amdgcn_s_waitcnt(0x0076);
        // the previous step's fragments (the other register set) are complete now; re-defining them through an empty asm keeps
        // the compiler from waiting for them again — with the new reads already queued behind — in front of their MFMAs
        asm volatile("" : "+v"(pav), "+v"(pbv[0]), "+v"(pbv[1]), "+v"(pbv[2]), "+v"(pbv[3]), "+v"(pbv[4]), "+v"(pbv[5]), "+v"(pbv[6]), "+v"(pbv[7]));
        __builtin_amdgcn_s_barrier();  // everybody's pieces landed; everybody is done with the stage of the previous step
        issue();                       // four steps ahead -> the stage the previous step used
        const unsigned char *base = &sh.stage[0][0] + rd_stage;
        rd_stage = rd_stage + BF_STAGE_BYTES == BF_STAGES * BF_STAGE_BYTES ? 0 : rd_stage + BF_STAGE_BYTES;
        if constexpr (mma_first) {
            if (with_mma) mma(pav, pbv);
            __builtin_amdgcn_sched_barrier(0);
            av = *reinterpret_cast<const bf16x8 *>(base + a_off);
#pragma unroll
            for (int t = 0; t < 8; t++) bv[t] = *reinterpret_cast<const bf16x8 *>(base + b_off + 32 * t * 32);
        } else {
            av = *reinterpret_cast<const bf16x8 *>(base + a_off);
#pragma unroll
            for (int t = 0; t < 8; t++) bv[t] = *reinterpret_cast<const bf16x8 *>(base + b_off + 32 * t * 32);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (they belong to the other register set)
            if (with_mma) mma(pav, pbv);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    bf16x8 av0 = {}, bv0[8] = {}, av1 = {}, bv1[8] = {};
    // approximate scores of a finished tile -> per-query candidate lists (one accumulator at a time); LDS + scalar loads only
    auto epilogue = [&](uint32_t tile) __attribute__((always_inline)) {
        const uint32_t r0 = tile * BF_BN;
        // the tile's 256 row-mask bits through the scalar cache (a vector load here would make the compiler drain the DMA queue)
        typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
        u32x8 okw;
        const uint64_t *mp = a.row_mask + (r0 >> 6);
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(okw) : "s"(mp) : "memory");
        float thr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) thr[r] = sh.thr_score[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half];
#define BF_EPI(T) bf_epilogue_tile(acc[T], thr, sh, okw[T], r0, 32 * wave, 32 * (T), lane)
        BF_EPI(0); BF_EPI(1); BF_EPI(2); BF_EPI(3); BF_EPI(4); BF_EPI(5); BF_EPI(6); BF_EPI(7);
#undef BF_EPI
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    };

    // Two K steps per iteration (nk and therefore G are even): the fragments of the next step are on their way from LDS while the
    // matrix cores work on the current one.  The last pair is peeled so that the loop body has no conditional fetch.
    auto run = [&](auto tag) __attribute__((always_inline)) {
        uint32_t tile = blockIdx.x, kc = 0, g = 0;
        step(tag, av0, bv0, av1, bv1, false);
        for (; g + 2 < G; g += 2) {
            step(tag, av1, bv1, av0, bv0, true);
            step(tag, av0, bv0, av1, bv1, true);
            kc += 2;
            if (kc == nk) {
                epilogue(tile);
                kc = 0;
                tile += gridDim.x;
            }
        }
        step(tag, av1, bv1, av0, bv0, true);
        __builtin_amdgcn_s_waitcnt(0x0070);  // the last fragments, and the requests that ran ahead of the last step
        mma(av1, bv1);
        epilogue(tile);
    };
    if (G) {
        if (wave < 4) run(std::true_type{});
        else run(std::false_type{});
    }
    __syncthreads();
    for (int i = tid; i < BF_BM * BF_KP; i += BF_THREADS) {
        int q = i / BF_KP, e = i % BF_KP;
        if (q0 + q < a.n_queries) a.partial[((size_t)(q0 + q) * gridDim.x + blockIdx.x) * BF_KP + e] = sh.lists[q][e];
    }
}

// ---- the same scan without candidate lists in LDS (round 4; the mainloop is round 6's) -----------------------------------------------
// With a floor per query (32 rows of a sample are known to reach it) a stripe holds only a handful of rows at or above it, so the
// sorted lists — 64 KiB of LDS — are not needed: a row that reaches its query's floor is APPENDED to the query's 32 slots of this
// stripe (an LDS counter gives the position, the key goes straight to HBM), and merge_topk_kernel sees the same [query][stripe][32]
// block as before.  A stripe that would need a 33rd slot raises its query block's flag and bf16_scan_kernel (run_if) scans that block
// again with its lists: the result is the list kernel's in every case.  No list belongs to a wave any more, so the wave tiles are
// 64 x 128 (wave (w & 3, w >> 2): 2 + 4 fragment reads per 8 MFMAs instead of 1 + 8).  A workgroup's tiles come in rounds (round i =
// tile blockIdx.x + i gridDim.x).  round_step > 1: only every round_step-th round is scanned — the sample across the corpus the floor
// of the full pass comes from; the groups' counts go to cnt_inout.  round_skip > 1 (the full pass after such a sample, same grid):
// those rounds are skipped and the sample's entries stay in their slots — the workgroup first drops the ones under the tightened floor
// and goes on appending behind the rest; a query block whose sample ran out of slots (skip_unless) empties its slots and scans every
// round.
//
// Mainloop: a ring of 16-element K chunks.  Round 4's form (two 64-KiB stages of four chunks, `s_waitcnt vmcnt(0)` + barrier per stage)
// requested a stage ONE stage ahead — 2 048 matrix-pipe cycles per SIMD, about a microsecond, less than a loaded HBM round trip — and its
// waves were parked at that wait for a third of their cycles (profiles/r06_sq_counters_bf16_12m5x1024.txt).  Here the unit of staging is
// the 16-KiB image of ONE chunk ([Q block 8 KiB][X block 8 KiB]; the HBM layout is unchanged), R of them form a ring, and a chunk is
// requested R - KC chunks (6 x 512 pipe cycles) before it is read; nothing in the loop waits for vmcnt(0).  A barrier every KC chunks
// publishes the next KC chunks; the fragment reads of chunk c and the two LDS-DMA pieces a wave requests per chunk sit in the issue gaps
// between the 8 MFMAs of chunk c - 1 (two register sets), so the first MFMA after a barrier never waits for LDS.  Measured (12.5 M x 1024,
// batch 1 024, three interleaved runs on one box): 24.24 ms per batch against 24.95 ms — the gain is small because the wait was not
// latency: the ablations of profiles/r06_bf16_ablation.txt (library built with -DNIDX_BF16_ABLATE) show the requests alone (no MFMAs,
// no fragment reads) take 0.6 of the batch and the MFMAs alone 0.65 — the LDS port (16 KiB of DMA writes at 64 B/clk + 48 KiB of
// fragment reads at 256 B/clk = 448 of the 512 cycles a chunk's MFMAs take per SIMD pair) and the L2 -> LDS path at the clock the matrix
// pipe's power leaves (1.6 GHz with every MFMA slot used) are what the 256 x 256 tile saturates.  Two shorter epilogues (eight v_max3 and
// one compare per accumulator instead of sixteen compares; a zero C operand instead of clearing the accumulators) measured 5 % SLOWER.
template <int R>
struct Bf16RingShared {
    __attribute__((aligned(16))) unsigned char slot[R][2 * BF_BLOCK_BYTES];
    float thr[BF_BM];
    uint32_t cnt[BF_BM];
    uint64_t keep[BF_THREADS / 64][BF_KP];
    uint32_t overflow;
};

template <int KC, int R>
__global__ __launch_bounds__(BF_THREADS, 1) void bf16_append_kernel(Bf16ScanArgs a) {
    static_assert(R % KC == 0 && 2 * (R - 2 * KC) <= 15, "ring shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char bf_smem[];
    Bf16RingShared<R> &sh = *reinterpret_cast<Bf16RingShared<R> *>(bf_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const int wq = wave & 3, wr = wave >> 2;          // this wave: queries 64 wq.., tile rows 128 wr..
    const uint32_t q0 = blockIdx.y * BF_BM;
    const uint32_t rs = a.round_step ? a.round_step : 1u;
    const bool after_sample = a.round_skip > 1;
    const bool keep_sample = after_sample && !(a.skip_unless && a.skip_unless[blockIdx.y]);
    const uint32_t skip = keep_sample ? a.round_skip : 0u;
    const uint32_t n_tiles = (a.n + BF_BN - 1) / BF_BN;
    const uint32_t nk = a.dp16 / BF_BK;               // chunks per tile (a multiple of 4)
#ifdef NIDX_BF16_ABLATE
    const int abl = a.debug;   // experiment builds only (wrong results): 1 no requests, 2 no MFMAs, 4 no epilogue, 8 corpus from eight tiles, 16 no fragment reads, 32 no barriers
#else
    constexpr int abl = 0;
#endif

    if (tid < BF_BM) {
        const bool real = q0 + tid < a.n_queries;
        float t0 = INFINITY;
        if (real) {
            const float f = a.floor_score[q0 + tid];
            t0 = -INFINITY;
            if (f > -INFINITY) {
                const int32_t k = total_key(f) - 1;                                           // the float just below f in total order
                t0 = __builtin_bit_cast(float, k ^ (int32_t)(((uint32_t)(k >> 31)) >> 1));
            }
        }
        sh.thr[tid] = t0;
        sh.cnt[tid] = 0;
    }
    if (tid == 0) sh.overflow = 0;
    __syncthreads();
    if (tid < BF_BM && sh.thr[tid] == -INFINITY) sh.overflow = 1u;   // a query without a floor: the block is the list kernel's
    __syncthreads();
    if (sh.overflow) {
        if (tid == 0) atomicOr(&a.overflow[blockIdx.y], 1u);
        return;
    }

    const uint32_t rounds = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t my_tiles = skip ? rounds - (rounds + skip - 1) / skip : (rounds + rs - 1) / rs;
    const uint32_t C = my_tiles * nk;                 // chunks of this workgroup's stream (even)
    const uint32_t round0 = skip ? 1u : 0u;
    auto next_round = [&](uint32_t r) __attribute__((always_inline)) {
        r += rs;
        if (skip && r % skip == 0) r++;
        return r;
    };
    if (after_sample) {
        __syncthreads();
        for (int i = 0; i < 32; i++) {
            const int ql = 32 * wave + i;
            const uint32_t q = q0 + (uint32_t)ql;
            if (q >= a.n_queries) break;
            uint64_t *slots = a.partial + ((size_t)q * gridDim.x + blockIdx.x) * BF_KP;
            uint32_t c = keep_sample ? a.cnt_inout[(size_t)q * gridDim.x + blockIdx.x] : 0u;
            c = c < (uint32_t)BF_KP ? c : (uint32_t)BF_KP;
            const uint64_t key = (uint32_t)lane < c ? slots[lane] : NIDX_EMPTY_KEY;
            const bool kept = (uint32_t)lane < c && rank_key_score(key) > sh.thr[ql];
            const unsigned long long m = __ballot(kept);
            if (kept) sh.keep[wave][__popcll(m & ((1ull << lane) - 1ull))] = key;
            const uint32_t n_kept = (uint32_t)__popcll(m);
            if (lane < BF_KP) slots[lane] = (uint32_t)lane < n_kept ? sh.keep[wave][lane] : NIDX_EMPTY_KEY;
            if (lane == 0) sh.cnt[ql] = n_kept;
        }
        __syncthreads();
    }

    // ---- requests: every wave copies piece `wave` (1 KiB) of the chunk's query block and of its corpus block ----
    const size_t tile_bytes = (size_t)nk * BF_BLOCK_BYTES;
    const unsigned char *qp = reinterpret_cast<const unsigned char *>(a.queries16) + (size_t)blockIdx.y * tile_bytes + (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
    const unsigned char *xp = reinterpret_cast<const unsigned char *>(a.vectors16) + ((size_t)blockIdx.x + (size_t)round0 * gridDim.x) * tile_bytes +
                              (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
    uint32_t is_kc = 0, is_round = round0, is_left = C;   // the next chunk to request; chunks not requested yet
    uint32_t wr_off = 0, rd_off = 0;
    constexpr uint32_t SLOT = 2 * BF_BLOCK_BYTES, RING = (uint32_t)R * SLOT;
    auto issue_q = [&]() __attribute__((always_inline)) { if (!(abl & 1)) bf_glds16<0>(qp, &sh.slot[0][0] + wr_off + (uint32_t)wave * 1024u); };
    auto issue_x = [&]() __attribute__((always_inline)) { if (!(abl & 1)) bf_glds16<0>(xp, &sh.slot[0][0] + wr_off + BF_BLOCK_BYTES + (uint32_t)wave * 1024u); };
    // past the stream's last chunk the same chunk is requested again into a slot nobody reads: the counts behind `s_waitcnt vmcnt(N)` stay true to the end
    auto advance = [&]() __attribute__((always_inline)) {
        wr_off = wr_off + SLOT == RING ? 0u : wr_off + SLOT;
        if (is_left > 1) {
            is_left--;
            qp += BF_BLOCK_BYTES;
            xp += BF_BLOCK_BYTES;
            if (++is_kc == nk) {
                is_kc = 0;
                qp -= tile_bytes;
                const uint32_t nr = next_round(is_round);
                if ((abl & 8) && (nr & 7u) == 0) xp -= (size_t)8 * gridDim.x * tile_bytes;
                xp += ((size_t)(nr - is_round) * gridDim.x - 1u) * tile_bytes;
                is_round = nr;
            }
        }
    };

    floatx16 acc[8];   // [a = query half 0..1][t = row quarter 0..3]
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int frag = (half ^ ((li >> 3) & 1)) * 16;
    const int a_off = (64 * wq + li) * 32 + frag, b_off = BF_BLOCK_BYTES + (128 * wr + li) * 32 + frag;

    auto epilogue = [&](uint32_t tile) __attribute__((always_inline)) {   // `tile`: the corpus tile
        if (abl & 4) return;
        const uint32_t r0 = tile * BF_BN;
        int half = lane >> 5, li = lane & 31;
        asm volatile("" : "+v"(half), "+v"(li));   // (keeps hipcc from hoisting this rare path's addresses into the mainloop's registers)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 okw;
        const uint64_t *mp = a.row_mask + (r0 >> 6) + 2 * wr;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(okw) : "s"(mp) : "memory");
#pragma unroll
        for (int qa = 0; qa < 2; qa++) {
            const int qb = 64 * wq + 32 * qa;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const floatx16 &c = acc[qa * 4 + t];
                const bool row_ok = (okw[t] >> li) & 1u;
                float thr[16];
#pragma unroll
                for (int r = 0; r < 16; r++) thr[r] = sh.thr[qb + (r & 3) + 8 * (r >> 2) + 4 * half];
                // the common case — no lane holds a candidate — costs one compare per value: the lane masks are ORed on the scalar side
                unsigned long long any = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) any |= __ballot(c[r] > thr[r]);
                if (!(any & __ballot(row_ok))) continue;
                uint32_t mask = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) mask |= (row_ok && c[r] > thr[r]) ? (1u << r) : 0u;
                const uint32_t row = r0 + (uint32_t)(128 * wr + 32 * t + li);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (!((mask >> r) & 1u)) continue;
                    const int ql = qb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const uint32_t pos = atomicAdd(&sh.cnt[ql], 1u);
                    if (pos < (uint32_t)BF_KP) a.partial[((size_t)(q0 + ql) * gridDim.x + blockIdx.x) * BF_KP + pos] = rank_key(c[r], row);
                    else sh.overflow = 1u;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    };

    __syncthreads();   // thresholds and counters
    if (C) {
        // s_waitcnt vmcnt(N) lgkmcnt(0) (gfx9 encoding: vmcnt [3:0], expcnt [6:4] = 7, lgkmcnt [11:8]): the pieces of the chunks this barrier
        // publishes have landed, R - 2 KC later chunks stay in flight
        constexpr int WAIT_BAR = 0x0070 | (2 * (R - 2 * KC));
        constexpr int WAIT_LDS = 0xC07F;   // lgkmcnt(0) alone
        for (int i = 0; i < R - KC; i++) {
            issue_q();
            issue_x();
            advance();
        }
        bf16x8 av0[2] = {}, bv0[4] = {}, av1[2] = {}, bv1[4] = {};
        uint32_t ep_round = round0, done_kc = 0;
        // One chunk: [wait + barrier] then the 8 MFMAs of the PREVIOUS chunk (fragments pav / pbv) with this chunk's 6 fragment reads and 2 requests
        // in their issue gaps.
        auto chunk = [&](auto bar_tag, bf16x8 (&av)[2], bf16x8 (&bv)[4], bf16x8 (&pav)[2], bf16x8 (&pbv)[4], bool with_mma) __attribute__((always_inline)) {
            constexpr bool bar = decltype(bar_tag)::value;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(bar ? WAIT_BAR : WAIT_LDS);
            // the previous chunk's fragments are complete now; re-defining them keeps the compiler from waiting for them again behind the new reads
            asm volatile("" : "+v"(pav[0]), "+v"(pav[1]), "+v"(pbv[0]), "+v"(pbv[1]), "+v"(pbv[2]), "+v"(pbv[3]));
            if constexpr (bar) if (!(abl & 32)) __builtin_amdgcn_s_barrier();
            const unsigned char *base = &sh.slot[0][0] + rd_off;
            rd_off = rd_off + SLOT == RING ? 0u : rd_off + SLOT;
            const bool with_reads = !(abl & 16);
            with_mma = with_mma && !(abl & 2);
#define BF_MMA(I) if (with_mma) acc[I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pav[(I) >> 2], pbv[(I) & 3], acc[I], 0, 0, 0)
#define BF_GAP() __builtin_amdgcn_sched_barrier(0)
            BF_MMA(0); if (with_reads) av[0] = *reinterpret_cast<const bf16x8 *>(base + a_off); BF_GAP();
            BF_MMA(1); if (with_reads) av[1] = *reinterpret_cast<const bf16x8 *>(base + a_off + 32 * 32); BF_GAP();
            BF_MMA(2); if (with_reads) bv[0] = *reinterpret_cast<const bf16x8 *>(base + b_off); BF_GAP();
            BF_MMA(3); if (with_reads) bv[1] = *reinterpret_cast<const bf16x8 *>(base + b_off + 32 * 32); BF_GAP();
            BF_MMA(4); if (with_reads) bv[2] = *reinterpret_cast<const bf16x8 *>(base + b_off + 2 * 32 * 32); BF_GAP();
            BF_MMA(5); if (with_reads) bv[3] = *reinterpret_cast<const bf16x8 *>(base + b_off + 3 * 32 * 32); BF_GAP();
            BF_MMA(6); issue_q(); BF_GAP();
            BF_MMA(7); issue_x(); advance(); BF_GAP();
#undef BF_MMA
#undef BF_GAP
            if (with_mma && ++done_kc == nk) {
                epilogue(blockIdx.x + ep_round * gridDim.x);
                done_kc = 0;
                ep_round = next_round(ep_round);
            }
        };
        using Bar = std::true_type;
        using Mid = std::integral_constant<bool, KC == 1>;   // an odd chunk has a barrier of its own only when every chunk is published alone
        chunk(Bar{}, av0, bv0, av1, bv1, false);
        for (uint32_t c = 1; c + 1 < C; c += 2) {
            chunk(Mid{}, av1, bv1, av0, bv0, true);
            chunk(Bar{}, av0, bv0, av1, bv1, true);
        }
        chunk(Mid{}, av1, bv1, av0, bv0, true);
        __builtin_amdgcn_s_waitcnt(0x0070);   // the last fragments; the requests that ran past the end of the stream
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av1[i >> 2], bv1[i & 3], acc[i], 0, 0, 0);
        epilogue(blockIdx.x + ep_round * gridDim.x);
        __syncthreads();
    }
    if (tid == 0 && sh.overflow) atomicOr(&a.overflow[blockIdx.y], 1u);
    if (a.cnt_inout && !after_sample && tid < BF_BM && q0 + tid < a.n_queries)
        a.cnt_inout[(size_t)(q0 + tid) * gridDim.x + blockIdx.x] = sh.cnt[tid] < (uint32_t)BF_KP ? sh.cnt[tid] : (uint32_t)BF_KP;
}

// Stage 3: one wave per query re-scores its candidates exactly (WAVE64 order) and keeps the best k.
template <int NJ>
__global__ __launch_bounds__(256) void rescore_select_kernel(RescoreArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= a.n_queries) return;
    const bool cosine = a.similarity == 1;
    float4 qv[NJ];
    float qacc = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        qv[j] = load_row_chunk(a.queries + (size_t)q * a.dp, a.dp, j, lane);
        qacc = fma4(qv[j], qv[j], qacc);
    }
    const float qq = wave_butterfly_sum(qacc);
    const uint32_t n_cand = a.cand_count[q];
    WaveSortedList top;
    top.init();
    for (uint32_t c = 0; c < n_cand; c++) {
        const uint32_t row = a.cand_vec[(size_t)q * a.n_cand_max + c];
        const float *r = a.vectors + (size_t)row * a.dp;
        float ab = 0.f, xx = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            float4 x = load_row_chunk(r, a.dp, j, lane);
            ab = fma4(x, qv[j], ab);
            xx = fma4(x, x, xx);
        }
        ab = wave_butterfly_sum(ab);
        float score = ab;
        if (cosine) {
            xx = wave_butterfly_sum(xx);
            score = cosine_from_sums(ab, xx, qq);
        }
        if (score >= a.min_score) top.insert(rank_key(score, row), lane);
    }
    const int k = (int)a.k;
    const bool v = top.key != NIDX_EMPTY_KEY && lane < k;
    unsigned long long vm = __ballot(v);
    if (lane < k) {
        a.out_vec[(size_t)q * k + lane] = v ? rank_key_addr(top.key) : 0xffffffffu;
        a.out_score[(size_t)q * k + lane] = v ? rank_key_score(top.key) : 0.f;
    }
    if (lane == 0) a.out_count[q] = (uint32_t)__popcll(vm);
}

hipError_t launch_to_bf16_tiled(const float *in, const float *norm2, uint32_t n, uint32_t dp, uint32_t dp16, unsigned short *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t chunks = (size_t)((n + 255u) / 256u) * (dp16 / BF_BK) * 512u;
    const uint32_t blocks = (uint32_t)std::min<size_t>((chunks + 255) / 256, 16384);
    hipLaunchKernelGGL(to_bf16_tiled_kernel, dim3(blocks), dim3(256), 0, s, in, norm2, n, dp, dp16, out);
    return hipGetLastError();
}

// floor[q] = the BF_KP-th best score of query q's merged candidates; prev != nullptr: never below prev[q], and prev[q] itself where
// the query has fewer candidates or its block's pass ran out of slots (overflow[q / 256] != 0: that pass's lists are incomplete)
__global__ __launch_bounds__(256) void bf16_floor_kernel(const float *__restrict__ cand_score, const uint32_t *__restrict__ cand_count, uint32_t n_queries,
                                                         const float *__restrict__ prev, const uint32_t *__restrict__ overflow, float *__restrict__ floor) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const float p = prev ? prev[q] : -INFINITY;
    float f = cand_count[q] >= BF_KP ? cand_score[(size_t)q * BF_KP + BF_KP - 1] : -INFINITY;
    if (overflow && overflow[q / BF_BM]) f = -INFINITY;
    floor[q] = fmaxf(f, p);
}

hipError_t launch_bf16_floor(const float *cand_score, const uint32_t *cand_count, uint32_t n_queries, const float *prev, const uint32_t *overflow,
                             float *floor, hipStream_t s) {
    if (n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(bf16_floor_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, cand_score, cand_count, n_queries, prev, overflow, floor);
    return hipGetLastError();
}

hipError_t launch_bf16_row_mask(uint32_t n, const uint32_t *para_of_vec, const uint64_t *alive, const uint64_t *filter, uint64_t *out,
                                hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t n_pad = (n + 255u) & ~255u;
    hipLaunchKernelGGL(bf16_row_mask_kernel, dim3(n_pad / 256), dim3(256), 0, s, n, n_pad, para_of_vec, alive, filter, out);
    return hipGetLastError();
}

uint32_t bf16_scan_stripes(uint32_t n, uint32_t n_queries) {
    uint32_t tiles = (n + BF_BN - 1) / BF_BN, qb = (n_queries + BF_BM - 1) / BF_BM;
    uint32_t s = 256 / (qb ? qb : 1);  // one 8-wave workgroup per CU
    if (s < 1) s = 1;
    if (s > tiles) s = tiles;
    return s ? s : 1;
}

hipError_t launch_bf16_scan(const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    if (a.n == 0 || (a.dp16 % 64u)) return hipErrorInvalidValue;
    const size_t smem = sizeof(Bf16Shared);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bf16_scan_kernel, dim3(stripes, (a.n_queries + BF_BM - 1) / BF_BM), dim3(BF_THREADS), smem, s, a);
    return hipGetLastError();
}

template <typename K, typename S>
static hipError_t bf16_append_launch_as(K kernel, const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s) {
    const size_t smem = sizeof(S);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(stripes, (a.n_queries + BF_BM - 1) / BF_BM), dim3(BF_THREADS), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_bf16_append(const Bf16ScanArgs &a, uint32_t stripes, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    if (a.n == 0 || (a.dp16 % 64u) || !a.floor_score || !a.overflow) return hipErrorInvalidValue;
    // NIDX_GPU_BF16_MAINLOOP=1: a ring of nine chunks with a barrier per chunk (measured 1 % behind the shipped ring of eight with a barrier per two)
    const char *fe = getenv("NIDX_GPU_BF16_MAINLOOP");
    const int form = fe ? atoi(fe) : 2;
    if (form == 1) return bf16_append_launch_as<decltype(&bf16_append_kernel<1, 9>), Bf16RingShared<9>>(&bf16_append_kernel<1, 9>, a, stripes, s);
    return bf16_append_launch_as<decltype(&bf16_append_kernel<2, 8>), Bf16RingShared<8>>(&bf16_append_kernel<2, 8>, a, stripes, s);
}

hipError_t launch_rescore_select(const RescoreArgs &a, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    const dim3 grid((a.n_queries + 3) / 4), block(256);
    int nj = (int)((a.dp + 255u) / 256u);
#define NIDX_RS_CASE(N) hipLaunchKernelGGL((rescore_select_kernel<N>), grid, block, 0, s, a); return hipGetLastError()
    if (nj <= 1) { NIDX_RS_CASE(1); }
    if (nj <= 2) { NIDX_RS_CASE(2); }
    if (nj <= 3) { NIDX_RS_CASE(3); }
    if (nj <= 4) { NIDX_RS_CASE(4); }
    if (nj <= 6) { NIDX_RS_CASE(6); }
    if (nj <= 8) { NIDX_RS_CASE(8); }
    if (nj <= 12) { NIDX_RS_CASE(12); }
    if (nj <= 16) { NIDX_RS_CASE(16); }
#undef NIDX_RS_CASE
    return hipErrorInvalidValue;
}

}  // namespace nidx
