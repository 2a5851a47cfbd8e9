// host_common.h — error plumbing and small RAII helpers for the C ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/nidx_gpu.h"

namespace nidx {

void set_error(const char *fmt, ...);
int32_t fail(int32_t code, const char *fmt, ...);
int32_t hip_fail(hipError_t e, const char *what);
int32_t abi_exception() noexcept;

// Every extern "C" entry point is a function-try-block closed by this handler: no C++ exception crosses the ABI.
#define NIDX_ABI_CATCH \
    catch (...) { return ::nidx::abi_exception(); }

#define NIDX_HIP(expr)                                         \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return ::nidx::hip_fail(_e, #expr); \
    } while (0)

// Device buffer that frees itself; tracks bytes for space_usage().
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    // grow-only scratch; small requests are rounded up to a power of two so that a serving loop whose batch sizes wander (the
    // coalescer's batches of 1..64 queries) reallocates a handful of times, not on every new maximum
    static size_t grow_size(size_t n) {
        if (n > ((size_t)256 << 20)) return n;
        size_t p = 4096;
        while (p < n) p <<= 1;
        return p;
    }
    hipError_t reserve(size_t n) { return n <= bytes ? hipSuccess : alloc(grow_size(n)); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// grow-only pinned host staging buffer: one hipMemcpyAsync moves a whole call's inputs (or outputs) without the runtime's own
// bounce through pageable memory
struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    PinBuf(PinBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    PinBuf &operator=(PinBuf &&o) noexcept {
        if (this != &o) {
            if (p) (void)hipHostFree(p);
            p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0;
        }
        return *this;
    }
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        n = DevBuf::grow_size(n);   // pinning is slow (tens of ms): grow in powers of two
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace nidx
