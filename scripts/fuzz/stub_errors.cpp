// Error plumbing for the sanitizer build of the host-only sources (the product's lives in vector_index.cpp, next to the HIP runtime).
#include <stdarg.h>
#include <stdio.h>

#include <new>

#include "host_common.h"
namespace nidx {
static thread_local char g_err[512];
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int32_t fail(int32_t code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
int32_t hip_fail(hipError_t, const char *) { return NIDX_ERR_DEVICE; }
int32_t abi_exception() noexcept {
    try { throw; } catch (const std::bad_alloc &) { return NIDX_ERR_OUT_OF_MEMORY; } catch (...) { return NIDX_ERR_INTERNAL; }
}
}  // namespace nidx
