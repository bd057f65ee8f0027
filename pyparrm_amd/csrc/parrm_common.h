// Shared helpers for the gfx950 PARRM kernels (internal; the public ABI is include/parrm_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "parrm_hip.h"

namespace parrm {

// Records the message returned by parrm_hip_last_error() on this thread.
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t err, const char *what);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// 64-lane wavefront sum, result valid in every lane (fixed butterfly order -> deterministic).
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace parrm

#define PARRM_HIP_CHECK(expr)                                          \
    do {                                                               \
        hipError_t parrm_e_ = (expr);                                  \
        if (parrm_e_ != hipSuccess) return ::parrm::hip_fail(parrm_e_, #expr); \
    } while (0)

#define PARRM_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            ::parrm::set_error(__VA_ARGS__);  \
            return PARRM_ERR_INVALID;         \
        }                                     \
    } while (0)
