// hnsw_spill.hip — closest_up_nodes (nidx_vector/src/hnsw/search.rs:188-240) with UNBOUNDED candidate and visited sets.
//
// The batched search kernel (hnsw_search.hip) keeps the walk's candidate heap in a 512-entry LDS pool and its visited set in
// an LDS hash table.  The reference's BinaryHeap / visited set are unbounded: under a very selective filter (or a low
// min_score with many duplicates) the walk pops hundreds of nodes before k of them are accepted and admits every unvisited
// neighbour on the way.  The fast kernel detects the moment its bounded structures would change the result
// (NIDX_FLAG_POOL_INEXACT / NIDX_FLAG_VISITED_OVERFLOW) and the host re-runs just those queries here: same walk, same
// arithmetic (eval_neighbours, score_from_sums, rank keys), but the pool lives in HBM — n slots per query, a node enters
// at most once — and the visited set is a bitset over the segment's vectors.  One workgroup per flagged query.
//
// Pool layout: append-only slots grouped in chunks of 64 (slot = chunk * 64 + lane) with the maximum key of every chunk in
// chunk_max[]; pop = arg-max over chunk_max (one pass, 64 chunks per step), then the arg-max inside that chunk; the popped
// slot is tombstoned.  Keys are unique (the vector address is part of the key), so the arg-max is unambiguous.
#include "hnsw_device.h"

namespace nidx {

namespace {

// the controller wave re-reads slots its other lanes wrote a few instructions earlier: workgroup-scope accesses keep that
// coherent through the CU's L1 (agent scope would bypass the per-XCD L2 on every access)
__device__ inline uint64_t ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void st64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int NJ>
__device__ inline bool spill_rows_equal(const SegDev &seg, uint32_t a, uint32_t b, int lane) {
    const float *ra = seg.vectors + (size_t)a * seg.dp, *rb = seg.vectors + (size_t)b * seg.dp;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        uint32_t e = (uint32_t)j * 256u + (uint32_t)lane * 4u;
        if (e < seg.dp) {
            uint4 x = *reinterpret_cast<const uint4 *>(ra + e);
            uint4 y = *reinterpret_cast<const uint4 *>(rb + e);
            eq = eq && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
        }
    }
    return __all(eq);
}

struct SpillPool {
    uint64_t *slots;      // [chunks * 64]
    uint64_t *chunk_max;  // [chunks], zeroed by the host
    uint32_t end;         // slots appended so far (wave-uniform)

    // appends the keys of the lanes with `ins` set (wave 0, all lanes call)
    __device__ inline void push(bool ins, uint64_t key, int lane) {
        const unsigned long long m = __ballot(ins);
        if (!m) return;
        const uint32_t pos = end + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (ins) st64(slots + pos, key);
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t c0 = end >> 6, c1 = (end + cnt - 1) >> 6;
        for (uint32_t c = c0; c <= c1; c++) {
            uint64_t v = (ins && (pos >> 6) == c) ? key : NIDX_EMPTY_KEY;
            v = wave_max_u64(v);
            if (lane == 0) {
                const uint64_t old = ld64(chunk_max + c);
                if (v > old) st64(chunk_max + c, v);
            }
        }
        end += cnt;
    }

    // removes and returns the best key (EMPTY when the pool is empty)
    __device__ inline uint64_t pop(int lane) {
        const uint32_t n_chunks = (end + 63) >> 6;
        uint64_t best = NIDX_EMPTY_KEY;
        uint32_t best_c = 0;
        for (uint32_t c = (uint32_t)lane; c < n_chunks; c += 64) {
            const uint64_t v = ld64(chunk_max + c);
            if (v > best) { best = v; best_c = c; }
        }
        const uint64_t top = wave_max_u64(best);
        if (top == NIDX_EMPTY_KEY) return NIDX_EMPTY_KEY;
        const unsigned long long who = __ballot(best == top);
        const uint32_t c = (uint32_t)__shfl((int)best_c, __ffsll((long long)who) - 1, 64);
        const uint32_t slot = c * 64u + (uint32_t)lane;
        uint64_t v = slot < end ? ld64(slots + slot) : NIDX_EMPTY_KEY;
        if (v == top) {
            st64(slots + slot, NIDX_EMPTY_KEY);
            v = NIDX_EMPTY_KEY;
        }
        const uint64_t rest = wave_max_u64(v);
        if (lane == 0) st64(chunk_max + c, rest);
        return top;
    }
};

}  // namespace

template <int NJ>
__global__ __launch_bounds__(256) void hnsw_closest_spill_kernel(HnswSpillArgs a) {
    __shared__ SearchShared sh;
    __shared__ uint32_t res_addr[NIDX_K_MAX];
    __shared__ float res_score[NIDX_K_MAX];
    __shared__ uint32_t res_para[NIDX_K_MAX];

    const int lane = threadIdx.x & 63;
    const bool ctl = (nidx_tid() >> 6) == 0;
    const uint32_t slot = blockIdx.x;
    const uint32_t qi = a.query_ids[slot];
    const bool cosine = a.seg.similarity == 1;
    const int k = (int)a.k;

    QueryRegs<NJ> q;
    load_query<NJ>(q, a.queries + (size_t)qi * a.seg.dp, a.seg.dp, lane, cosine);

    uint32_t *vis = a.vis + (size_t)slot * a.vis_words;
    SpillPool pool;
    pool.slots = a.pool + (size_t)slot * a.pool_chunks * 64u;
    pool.chunk_max = a.chunk_max + (size_t)slot * a.pool_chunks;
    pool.end = 0;
    int n_res = 0;

    if (ctl) {
        // candidates = the entry points, visited = exactly those (search.rs:196-203)
        const int n_entry = (int)a.entry_count[qi];
        for (int base = 0; base < n_entry; base += 64) {
            const int i = base + lane;
            const bool in = i < n_entry;
            uint32_t addr = 0;
            float s = 0.f;
            if (in) {
                addr = a.entry_vec[(size_t)qi * a.entry_stride + i];
                s = a.entry_score[(size_t)qi * a.entry_stride + i];
                atomicOr(&vis[addr >> 5], 1u << (addr & 31));
            }
            pool.push(in, rank_key(s, addr), lane);
        }
    }
    for (;;) {
        if (ctl) {
            int cont = 0, n_new = 0;
            const uint64_t ck = pool.pop(lane);
            if (ck != NIDX_EMPTY_KEY) {
                const float cs = rank_key_score(ck);
                const uint32_t c = rank_key_addr(ck);
                if (!(cs < a.min_score)) {
                    bool accept = !(cs != cs);
                    const uint32_t p = a.seg.para_of_vec ? a.seg.para_of_vec[c] : c;
                    if (accept) {
                        if (a.seg.alive && !bit_test(a.seg.alive, p)) accept = false;
                        if (accept && a.filter && !bit_test(a.filter, p)) accept = false;
                    }
                    if (accept && !a.with_duplicates) {
                        for (int i = 0; i < n_res && accept; i++) {
                            if (__builtin_bit_cast(uint32_t, res_score[i]) == __builtin_bit_cast(uint32_t, cs) &&
                                spill_rows_equal<NJ>(a.seg, res_addr[i], c, lane))
                                accept = false;
                        }
                    }
                    if (accept && a.multi) {
                        for (int base = 0; base < n_res && accept; base += 64)
                            if (__ballot(base + lane < n_res && res_para[base + lane] == p)) accept = false;
                    }
                    if (accept) {
                        if (lane == 0) {
                            res_addr[n_res] = c;
                            res_score[n_res] = cs;
                            res_para[n_res] = p;
                        }
                        n_res++;
                    }
                    if (n_res < k) {
                        cont = 1;
                        uint32_t deg;
                        const uint32_t w = load_edge_word(a.g, c, 0, lane, deg);
                        const bool is_edge = lane >= 1 && lane <= (int)deg;
                        bool fresh = false;
                        if (is_edge) {
                            const uint32_t bit = 1u << (w & 31);
                            fresh = !(atomicOr(&vis[w >> 5], bit) & bit);
                        }
                        const unsigned long long m = __ballot(fresh);
                        const int pos = __popcll(m & ((1ull << lane) - 1ull));
                        if (fresh) sh.nb[0].addr[pos] = w;
                        n_new = __popcll(m);
                    }
                }
            }
            if (lane == 0) {
                sh.ctrl[0] = cont;
                sh.ctrl[1] = n_new;
            }
        }
        __syncthreads();
        if (!sh.ctrl[0]) break;
        const int n_new = sh.ctrl[1];
        eval_neighbours<NJ, 2>(a.seg, q, sh.nb[0], n_new, cosine);
        __syncthreads();
        if (ctl && n_new > 0) {
            const float s = lane < n_new ? score_from_sums(sh.nb[0].ab[lane], sh.nb[0].xx[lane], q.qq, q.sqrt_qq, cosine) : 0.f;
            const uint32_t addr = sh.nb[0].addr[lane];
            pool.push(lane < n_new && s >= a.min_score, rank_key(s, addr), lane);
        }
    }

    // filtered_result.sort_by(|a, b| b.1.total_cmp(&a.1)) — stable (search.rs:381)
    if (ctl) {
        for (int e = lane; e < NIDX_K_MAX; e += 64) {
            if (e < n_res) {
                const float s = res_score[e];
                const int32_t key = total_key(s);
                int rank = 0;
                for (int j = 0; j < n_res; j++) {
                    const int32_t kj = total_key(res_score[j]);
                    rank += (kj > key || (kj == key && j < e)) ? 1 : 0;
                }
                a.out_vec[(size_t)qi * k + rank] = res_addr[e];
                a.out_score[(size_t)qi * k + rank] = s;
            } else if (e < k) {
                a.out_vec[(size_t)qi * k + e] = 0xffffffffu;
                a.out_score[(size_t)qi * k + e] = 0.f;
            }
        }
        if (lane == 0) a.out_count[qi] = (uint32_t)n_res;
    }
}

template <int NJ>
static hipError_t launch_spill_nj(const HnswSpillArgs &a, hipStream_t s) {
    hipLaunchKernelGGL((hnsw_closest_spill_kernel<NJ>), dim3(a.n_queries), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_hnsw_closest_spill(const HnswSpillArgs &a, hipStream_t s) {
    if (a.n_queries == 0) return hipSuccess;
    if (a.k == 0 || a.k > NIDX_K_MAX) return hipErrorInvalidValue;
    const int nj = (int)((a.seg.dp + 255u) / 256u);
    if (nj <= 1) return launch_spill_nj<1>(a, s);
    if (nj <= 2) return launch_spill_nj<2>(a, s);
    if (nj <= 3) return launch_spill_nj<3>(a, s);
    if (nj <= 4) return launch_spill_nj<4>(a, s);
    if (nj <= 6) return launch_spill_nj<6>(a, s);
    if (nj <= 8) return launch_spill_nj<8>(a, s);
    if (nj <= 12) return launch_spill_nj<12>(a, s);
    if (nj <= 16) return launch_spill_nj<16>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace nidx
