// diag.hip — measurement probes (no reference counterpart): what the box can do on the access pattern of the HNSW kernels.
//
// gather_probe_kernel: read-only random gathers of whole rows (NJ x 1 KiB, 16 B per lane — the row access of eval_neighbours,
// hnsw_device.h) from a matrix in HBM, independent of any traversal: every wave draws its row numbers from a counter hash, keeps
// R rows in flight, folds what it read into one word so the loads cannot be dropped.  rows/s x row bytes is the ceiling the
// HBM-bound roofline fraction of hnsw_search_kernel should be read against (a float4 COPY reaches 6.3 TB/s on this
// part because half of its traffic is writes; a pure gather has no write stream).
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <thread>
#include <vector>

#include "device_common.h"
#include "host_common.h"
#include "kernels.h"

namespace nidx {

__device__ inline uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int NJ, int R>
__global__ __launch_bounds__(256) void gather_probe_kernel(const float *rows, uint32_t n, uint32_t dp, uint32_t gathers_per_wave,
                                                           uint32_t seed, float *sink) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float acc = 0.f;
    for (uint32_t g = 0; g < gathers_per_wave; g += R) {
        float4 v[R][NJ];
#pragma unroll
        for (int r = 0; r < R; r++) {
            // wave-uniform row number (readfirstlane keeps the address arithmetic scalar, as in the search kernel)
            const uint32_t id = __builtin_amdgcn_readfirstlane(probe_hash(seed ^ (wave * 0x9e3779b9u + g + (uint32_t)r)) % n);
            const float *row = rows + (size_t)id * dp;
#pragma unroll
            for (int j = 0; j < NJ; j++) v[r][j] = load_row_chunk(row, dp, j, lane);
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc += v[r][j].x + v[r][j].y + v[r][j].z + v[r][j].w;
    }
    if (acc == 12345.678f) sink[wave] = acc;  // never true for the probe's data; keeps the loads alive
}

template <int NJ>
static hipError_t launch_probe(const float *rows, uint32_t n, uint32_t dp, uint32_t waves, uint32_t gathers_per_wave, int in_flight,
                               uint32_t seed, float *sink, hipStream_t s) {
    const dim3 grid((waves + 3) / 4), block(256);
    switch (in_flight) {
        case 1: hipLaunchKernelGGL((gather_probe_kernel<NJ, 1>), grid, block, 0, s, rows, n, dp, gathers_per_wave, seed, sink); break;
        case 2: hipLaunchKernelGGL((gather_probe_kernel<NJ, 2>), grid, block, 0, s, rows, n, dp, gathers_per_wave, seed, sink); break;
        case 4: hipLaunchKernelGGL((gather_probe_kernel<NJ, 4>), grid, block, 0, s, rows, n, dp, gathers_per_wave, seed, sink); break;
        default: hipLaunchKernelGGL((gather_probe_kernel<NJ, 8>), grid, block, 0, s, rows, n, dp, gathers_per_wave, seed, sink); break;
    }
    return hipGetLastError();
}

}  // namespace nidx

using namespace nidx;

extern "C" int32_t nidx_gpu_diag_gather(const float *d_rows, uint32_t n_rows, uint32_t dimension, uint32_t waves, uint32_t gathers_per_wave,
                                        int32_t rows_in_flight, uint32_t repeats, float *ms_out) try {
    if (!d_rows || !ms_out || n_rows == 0 || waves == 0 || gathers_per_wave == 0 || repeats == 0)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "bad probe arguments");
    if (dimension & 3u) return fail(NIDX_ERR_UNSUPPORTED, "the probe reads 16 B per lane: dimension must be a multiple of 4");
    if (rows_in_flight != 1 && rows_in_flight != 2 && rows_in_flight != 4 && rows_in_flight != 8)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "rows_in_flight must be 1, 2, 4 or 8");
    const uint32_t nj = (dimension + 255u) / 256u;
    if (nj != 3 && nj != 4) return fail(NIDX_ERR_UNSUPPORTED, "the probe is built for 513..1024-dimensional rows");
    gathers_per_wave = (gathers_per_wave + 7u) & ~7u;
    DevBuf sink;
    NIDX_HIP(sink.alloc((size_t)((waves + 3) / 4) * 4 * 4));
    hipEvent_t e0, e1;
    NIDX_HIP(hipEventCreate(&e0));
    NIDX_HIP(hipEventCreate(&e1));
    hipError_t err = hipSuccess;
    for (uint32_t r = 0; r < repeats + 1 && err == hipSuccess; r++) {
        if (r == 1) err = hipEventRecord(e0, nullptr);  // first launch = warm-up
        if (err == hipSuccess)
            err = nj == 3 ? launch_probe<3>(d_rows, n_rows, dimension, waves, gathers_per_wave, rows_in_flight, 1000u + r, sink.as<float>(), nullptr)
                          : launch_probe<4>(d_rows, n_rows, dimension, waves, gathers_per_wave, rows_in_flight, 1000u + r, sink.as<float>(), nullptr);
    }
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    NIDX_HIP(err);
    *ms_out = ms / (float)repeats;
    return NIDX_OK;
} NIDX_ABI_CATCH

// The reference's request shape measured without an interpreter in the way: `threads` native threads each issue
// nidx_gpu_vector_search_one calls back to back (one blocking thread per Search request, src/searcher/shard_search.rs:139-153);
// the coalescer (coalescer.cpp) merges concurrent callers into batched launches.  latencies_us_out[c] = wall time of call c.
extern "C" int32_t nidx_gpu_diag_single_query_latency(nidx_gpu_vector_index_t *index, const float *queries, uint32_t n_queries, uint32_t dimension,
                                                      const nidx_gpu_vector_search_params_t *params, uint32_t threads, uint32_t calls,
                                                      float *latencies_us_out, double *elapsed_s_out) try {
    if (!index || !queries || !params || !latencies_us_out || n_queries == 0 || threads == 0 || calls == 0 || params->k == 0)
        return fail(NIDX_ERR_INVALID_ARGUMENT, "bad latency probe arguments");
    std::vector<int32_t> rc(threads, NIDX_OK);
    int dev = 0;
    NIDX_HIP(hipGetDevice(&dev));
    // a serving thread lives long: its first HIP call (per-thread runtime state, tens of ms when 64 threads start at once) is not
    // part of a request's latency — every thread makes one untimed call, then all start the timed calls together
    std::atomic<uint32_t> ready{0};
    std::chrono::steady_clock::time_point t0;
    std::vector<std::thread> pool;
    pool.reserve(threads);
    // nothing may leave a worker thread as an exception (std::terminate, not an error code), and a thread that cannot be started
    // must not strand the ones already spinning at the start line
    std::atomic<uint32_t> started{0};
    std::atomic<bool> abort_run{false};
    std::mutex start_mu;
    std::condition_variable start_cv;
    auto worker = [&](uint32_t t) {
        try {
            (void)hipSetDevice(dev);
            const uint32_t k = params->k;
            std::vector<uint32_t> seg(k), par(k), vec(k);
            std::vector<float> score(k);
            uint32_t count = 0;
            int32_t r = nidx_gpu_vector_search_one(index, queries + (size_t)(t % n_queries) * dimension, dimension, params, seg.data(), par.data(),
                                                   vec.data(), score.data(), &count);
            {   // the start line: parked, not spinning (1 024 yielding threads on 64 cores would be the first thing measured)
                std::unique_lock<std::mutex> lk(start_mu);
                if (ready.fetch_add(1) + 1 == threads) {
                    t0 = std::chrono::steady_clock::now();
                    start_cv.notify_all();
                } else {
                    start_cv.wait(lk, [&] { return ready.load() >= threads || abort_run.load(); });
                }
            }
            if (abort_run.load()) return;
            if (r != NIDX_OK) { rc[t] = r; return; }
            for (uint32_t c = t; c < calls; c += threads) {
                const auto a = std::chrono::steady_clock::now();
                r = nidx_gpu_vector_search_one(index, queries + (size_t)(c % n_queries) * dimension, dimension, params, seg.data(), par.data(),
                                               vec.data(), score.data(), &count);
                latencies_us_out[c] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - a).count();
                if (r != NIDX_OK) { rc[t] = r; return; }
            }
        } catch (const std::bad_alloc &) {
            rc[t] = NIDX_ERR_OUT_OF_MEMORY;
            std::lock_guard<std::mutex> lk(start_mu);
            ready.fetch_add(1);
            start_cv.notify_all();
        } catch (...) {
            rc[t] = NIDX_ERR_INTERNAL;
            std::lock_guard<std::mutex> lk(start_mu);
            ready.fetch_add(1);
            start_cv.notify_all();
        }
    };
    int32_t spawn_rc = NIDX_OK;
    for (uint32_t t = 0; t < threads; t++) {
        try {
            pool.emplace_back(worker, t);
            started.fetch_add(1);
        } catch (...) {
            spawn_rc = fail(NIDX_ERR_INTERNAL, "could not start probe thread %u of %u", t, threads);
            {
                std::lock_guard<std::mutex> lk(start_mu);
                abort_run.store(true);
            }
            start_cv.notify_all();
            break;
        }
    }
    for (auto &th : pool) th.join();
    if (spawn_rc != NIDX_OK) return spawn_rc;
    if (elapsed_s_out) *elapsed_s_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (uint32_t t = 0; t < threads; t++)
        if (rc[t] != NIDX_OK) return rc[t];
    return NIDX_OK;
} NIDX_ABI_CATCH
