// vector_scan_shared.hip — the exact WAVE64-order scan for LARGE query batches (gfx950).
//
// Same arithmetic as scan_topk_kernel (vector_scan.hip; OpenSegment::brute_force_search, nidx_vector/src/segment.rs:569-623):
// per (row, query) an fmaf chain per lane over the row's 1-KiB chunks, the transposed xor butterfly over the eight queries of a
// wave, cosine from the f32 sums in f64, rank keys — scores and ranks are bit-identical to the register-tile scan and to the
// HNSW kernels.  What changes is who fetches the rows.  In scan_topk_kernel every workgroup owns ONE tile of 8 queries and
// streams the corpus for it: 1 024 queries = 128 passes over HBM (measured 159 ms per 1 M x 768 batch, 2.4 TB/s).  Here a
// workgroup of 8 waves owns 64 queries — wave w keeps queries 8 w.. in registers, exactly like the other kernel — and the eight
// waves share the rows: tiles of 8 consecutive rows travel HBM -> LDS once per workgroup (global_load_lds_dwordx4, wave w copies
// row w of the tile; rows are contiguous in HBM so the copy is linear) through three stages with a counted s_waitcnt vmcnt — one
// tile always in flight, never vmcnt(0) — and every wave reads every row of the tile from LDS (each lane its own 16 bytes of a
// 1-KiB chunk: conflict-free).  1 024 queries = 16 passes over HBM.  Nothing else in the loop is a vector memory load (those
// return in order behind the DMA requests and would drain the queue): the tile's eight |x|^2 ride along as a ninth, 32-byte DMA
// piece per wave, and alive / filter / paragraph indirection are folded into a row bitset beforehand (launch_bf16_row_mask) that
// is read through the scalar cache.
//
// Bound: VALU (64 lanes x D fma per (row, query)) once the passes are down to 16; algorithmic HBM bytes = 4 D per (row, 64 queries).
// Used for n_queries >= 64, D a multiple of 256, k <= 64, and unless the filter leaves so few rows that skipping their loads
// (which the register-tile scan does, and this one cannot) pays more — see scan_shared_stripes().
#include "device_common.h"
#include "kernels.h"

namespace nidx {

#define SS_WAVES 8
#define SS_QT 8
#define SS_ROWS 8   /* rows per tile = one per wave */
#define SS_STAGES 3

template <int NJ>
__global__ __launch_bounds__(64 * SS_WAVES, 1) void scan_shared_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t q0 = blockIdx.y * (SS_WAVES * SS_QT) + (uint32_t)wave * SS_QT;
    const int myq = QReduce<SS_QT>::query_of_lane(lane);
    const bool cosine = a.similarity == 1;
    const uint32_t row_bytes = a.dp * 4u;
    const uint32_t stage_bytes = SS_ROWS * row_bytes + SS_WAVES * 1024u;   // rows, then one private 1-KiB slot per wave whose first 32 bytes are the tile's 8 norms
    const bool live = q0 < a.n_queries;   // a wave whose queries are all padding still copies its row and meets the barriers

    // query tile -> registers; queries past n_queries replicate the last one (their lists are dropped)
    float4 qv[SS_QT][NJ];
    double sqrt_qq = 0.0;
    float qq_mine = 0.f;
    {
        float qq[SS_QT];
#pragma unroll
        for (int q = 0; q < SS_QT; q++) {
            const uint32_t qi = q0 + q < a.n_queries ? q0 + q : a.n_queries - 1;
            const float *qrow = a.queries + (size_t)qi * a.dp;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                qv[q][j] = load_row_chunk(qrow, a.dp, j, lane);
                acc = fma4(qv[q][j], qv[q][j], acc);
            }
            qq[q] = acc;
        }
        if (cosine) {
            qq_mine = QReduce<SS_QT>::run(qq, lane);
            sqrt_qq = sqrt((double)qq_mine);
        }
    }
    // the queries are in registers before the first DMA request: the compiler must not find a pending vector load behind the
    // requests later (it would wait vmcnt(0) for it inside the loop and drain the queue)
    __builtin_amdgcn_s_waitcnt(0x0070);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 qp[SS_QT / 2][NJ][4];  // [query pair][chunk][component] = (element of query 2h, element of query 2h + 1)
#pragma unroll
    for (int h = 0; h < SS_QT / 2; h++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            qp[h][j][0] = f32x2{qv[2 * h][j].x, qv[2 * h + 1][j].x};
            qp[h][j][1] = f32x2{qv[2 * h][j].y, qv[2 * h + 1][j].y};
            qp[h][j][2] = f32x2{qv[2 * h][j].z, qv[2 * h + 1][j].z};
            qp[h][j][3] = f32x2{qv[2 * h][j].w, qv[2 * h + 1][j].w};
#pragma unroll
            for (int c = 0; c < 4; c++) asm volatile("" : "+v"(qp[h][j][c]));
        }
    asm volatile("" : "+v"(qq_mine));
    WaveTopK<1> top[SS_QT];
#pragma unroll
    for (int q = 0; q < SS_QT; q++) top[q].init();
    uint64_t thr = NIDX_EMPTY_KEY;  // k-th key of this lane's query (EMPTY while the list is short)
    const int k = (int)a.k;

    // the stream of this workgroup's row tiles; requests run two tiles ahead and, past the end, re-read the last tile into a
    // stage nobody looks at again (constant queue depth: one s_waitcnt immediate, no tail cases)
    const uint32_t n_tiles = (a.n + SS_ROWS - 1) / SS_ROWS;
    uint32_t is_tile = blockIdx.x, wr_stage = 0, rd_stage = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        const uint32_t t = is_tile < n_tiles ? is_tile : n_tiles - 1;
        uint32_t r = t * SS_ROWS + (uint32_t)wave;
        r = r < a.n ? r : a.n - 1;  // clamped rows are never scored
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.vectors + (size_t)r * a.dp) + lane * 16;
        unsigned char *dst = ss_smem + wr_stage + (uint32_t)wave * row_bytes;
#pragma unroll
        for (int j = 0; j < NJ; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
        // norm2[8 t .. 8 t + 7] into this wave's own slot: a full-wave piece (no divergent branch around a DMA request — the
        // compiler's wait-count pass then treats every later LDS read as dependent on it); lanes >= 2 re-read the same 32 bytes
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(a.norm2 + (size_t)t * SS_ROWS) + (lane & 1) * 16),
                                         (__attribute__((address_space(3))) void *)(ss_smem + wr_stage + SS_ROWS * row_bytes + (uint32_t)wave * 1024u), 16, 0, 0);
        is_tile += gridDim.x;
        wr_stage = wr_stage + stage_bytes == SS_STAGES * stage_bytes ? 0 : wr_stage + stage_bytes;
    };
    if (blockIdx.x < n_tiles) {
        issue();
        issue();
    }
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // this wave's row of the tile has landed (the NJ pieces of the next tile stay in flight); gfx9 s_waitcnt encoding: vmcnt
        // [3:0] + [15:14], expcnt [6:4] = 7 "no wait", lgkmcnt [11:8] = 0
        __builtin_amdgcn_s_waitcnt(0x0070 | (NJ + 1));
        __builtin_amdgcn_s_barrier();  // everybody's rows landed; everybody is done with the previous tile's stage
        issue();                       // two tiles ahead -> the stage the previous tile used
        const unsigned char *stage = ss_smem + rd_stage;
        rd_stage = rd_stage + stage_bytes == SS_STAGES * stage_bytes ? 0 : rd_stage + stage_bytes;
        if (!live) continue;
        const uint32_t r0 = tile * SS_ROWS;
        // the tile's 8 row-mask bits through the scalar cache
        uint32_t mask_word;
        {
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(a.row_mask) + (r0 >> 5);
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mask_word) : "s"(mp) : "memory");
        }
        const uint32_t tile_mask = (mask_word >> (r0 & 31u)) & 0xffu;
        // the tile's 8 norms (this wave's own DMA slot), same explicit read
        typedef float nf32x4 __attribute__((ext_vector_type(4)));
        nf32x4 n_lo, n_hi;
        {
            const uint32_t naddr = (uint32_t)(uintptr_t)(stage + SS_ROWS * row_bytes + (uint32_t)wave * 1024u);
            // early-clobber outputs: the first result must not land in the address register the second read still needs
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(n_lo), "=&v"(n_hi) : "v"(naddr) : "memory");
        }
        // SEVERAL rows per iteration: a wave's work on one row is a single dependent chain (reads -> fma chains -> 6 reduction levels
        // -> f64 cosine -> key -> ballot) and only two waves share a SIMD, so the other rows' chains are what fills the
        // latency of the first.  A row the mask excludes is computed along and dropped at the admission test.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        auto read_row = [&](uint32_t rr, f32x4 (&raw)[NJ]) __attribute__((always_inline)) {
            // explicit ds_read_b128: behind a C++ load the compiler's wait-count pass sees an LDS read that may alias the DMA
            // requests in flight and puts s_waitcnt vmcnt(0) in front of it — which would drain the queue every row.  The stage
            // being read was waited for above; the ones being written are other stages.
            const uint32_t addr = (uint32_t)(uintptr_t)(stage + rr * row_bytes + lane * 16);
            asm volatile("ds_read_b128 %0, %1" : "=v"(raw[0]) : "v"(addr) : "memory");
            if constexpr (NJ > 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(raw[1]) : "v"(addr) : "memory");
            if constexpr (NJ > 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(raw[2]) : "v"(addr) : "memory");
            if constexpr (NJ > 3) asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(raw[3]) : "v"(addr) : "memory");
        };
        auto wait_row = [&](f32x4 (&ra)[NJ]) __attribute__((always_inline)) {   // after the LAST read: every row of the group is in
            if constexpr (NJ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0])::"memory");
            if constexpr (NJ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1])::"memory");
            if constexpr (NJ == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2])::"memory");
            if constexpr (NJ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3])::"memory");
        };
        // score of the row against this lane's query, from the per-lane partial sums of the wave's 8 queries
        auto finish = [&](float (&acc)[SS_QT], float xx) __attribute__((always_inline)) -> float {
            const float ab = QReduce<SS_QT>::run(acc, lane);
            if (!cosine) return ab;
            // cosine_from_sums with sqrt(|q|^2) hoisted (same f64 operations, same order)
            const double dab = (double)ab, dxx = (double)xx;
            double dist;
            if (dxx == 0.0 && (double)qq_mine == 0.0) dist = 0.0;
            else if (dab == 0.0) dist = 1.0;
            else {
                const double d = 1.0 - dab / (sqrt(dxx) * sqrt_qq);
                dist = d > 0.0 ? d : 0.0;
            }
            return 1.0f - (float)dist;
        };
        auto admit = [&](float score, uint32_t r, bool row_ok) __attribute__((always_inline)) {
            const uint64_t ck = rank_key(score, r);
            const bool ok = row_ok && (score >= a.min_score) && (ck > thr) && ((lane & QReduce<SS_QT>::group_mask()) == 0);
            unsigned long long m = __ballot(ok);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint64_t nk = lane_bcast_u64(ck, src);
                const int q = QReduce<SS_QT>::query_of_lane(src);
#pragma unroll
                for (int qq = 0; qq < SS_QT; qq++) {
                    if (qq == q) {
                        const uint64_t kth = top[qq].insert_kth(nk, k, lane);
                        if (myq == qq) thr = kth;
                    }
                }
            }
        };
        const float xs[8] = {n_lo.x, n_lo.y, n_lo.z, n_lo.w, n_hi.x, n_hi.y, n_hi.z, n_hi.w};
        constexpr int RG = NJ <= 3 ? 4 : 2;   // rows per group: what the register file holds next to the 8 x NJ x 4 query registers
#pragma unroll
        for (uint32_t rr = 0; rr < SS_ROWS; rr += RG) {
            const uint32_t bits = (tile_mask >> rr) & ((1u << RG) - 1u);
            if (!bits) continue;
            f32x4 raw[RG][NJ];
#pragma unroll
            for (int g = 0; g < RG; g++) read_row(rr + g, raw[g]);
#pragma unroll
            for (int g = 0; g < RG; g++) wait_row(raw[g]);   // one real wait (the first), the rest only tie the registers to it
            // two queries per v_pk_fma_f32: the row element is broadcast to both halves, the pair of query elements sits in one
            // 64-bit register pair (qp[]), each half is its query's own fmaf chain in the usual (chunk, component) order
            f32x2 acc2[RG][SS_QT / 2];
#pragma unroll
            for (int g = 0; g < RG; g++)
#pragma unroll
                for (int h = 0; h < SS_QT / 2; h++) acc2[g][h] = f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int g = 0; g < RG; g++) {
                        const float x = c == 0 ? raw[g][j].x : c == 1 ? raw[g][j].y : c == 2 ? raw[g][j].z : raw[g][j].w;
#pragma unroll
                        for (int h = 0; h < SS_QT / 2; h++) acc2[g][h] = __builtin_elementwise_fma(f32x2{x, x}, qp[h][j][c], acc2[g][h]);
                    }
            float score[RG];
#pragma unroll
            for (int g = 0; g < RG; g++) {
                float acc[SS_QT];
#pragma unroll
                for (int h = 0; h < SS_QT / 2; h++) {
                    acc[2 * h] = acc2[g][h].x;
                    acc[2 * h + 1] = acc2[g][h].y;
                }
                score[g] = finish(acc, xs[rr + g]);
            }
#pragma unroll
            for (int g = 0; g < RG; g++) admit(score[g], r0 + rr + g, (bits >> g) & 1u);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // the requests that ran ahead of the last tile
    // a query's list lives in exactly one wave: no merge inside the workgroup
#pragma unroll
    for (int q = 0; q < SS_QT; q++)
        if (q0 + q < a.n_queries && lane < k) a.partial[((size_t)(q0 + q) * gridDim.x + blockIdx.x) * k + lane] = top[q].mine(0);
}

// Row stripes (= candidate lists per query) the shared-row scan would use; 0 = keep the register-tile scan.
uint32_t scan_shared_stripes(uint32_t n, uint32_t n_queries, uint32_t dp, uint32_t k, uint64_t matching) {
    if (n_queries < 64 || (dp % 256u) != 0 || dp > 1024 || k == 0 || k > 64 || n < 4096) return 0;   // 8 queries x dp / 64 registers per lane
    // the register-tile scan never loads a filtered row; this one streams all of them: only when a good part of the rows count
    if (matching * 8 < n) return 0;
    const uint32_t groups = (n_queries + SS_WAVES * SS_QT - 1) / (SS_WAVES * SS_QT);
    uint32_t s = 256 / groups;  // one 8-wave workgroup per CU
    if (s < 1) s = 1;
    const uint32_t tiles = (n + SS_ROWS - 1) / SS_ROWS;
    if (s > tiles) s = tiles;
    return s;
}

template <int NJ>
static hipError_t launch_shared_nj(const ScanArgs &a, uint32_t stripes, hipStream_t s) {
    const size_t smem = (size_t)SS_STAGES * (SS_ROWS * a.dp * 4 + SS_WAVES * 1024);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_shared_kernel<NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_shared_kernel<NJ>), dim3(stripes, (a.n_queries + SS_WAVES * SS_QT - 1) / (SS_WAVES * SS_QT)), dim3(64 * SS_WAVES), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_scan_shared(const ScanArgs &a, uint32_t stripes, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    if (a.k == 0 || a.k > 64 || (a.dp % 256u) != 0 || stripes == 0 || a.n == 0) return hipErrorInvalidValue;
    if (!a.row_mask) return hipErrorInvalidValue;
    switch (a.dp / 256u) {
        case 1: return launch_shared_nj<1>(a, stripes, s);
        case 2: return launch_shared_nj<2>(a, stripes, s);
        case 3: return launch_shared_nj<3>(a, stripes, s);
        case 4: return launch_shared_nj<4>(a, stripes, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nidx
