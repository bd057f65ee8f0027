// Shared helpers for the gfx950 PARRM kernels (internal; the public ABI is include/parrm_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "parrm_hip.h"

namespace parrm {

// Records the message returned by parrm_hip_last_error() on this thread.
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t err, const char *what);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the CURRENT device (hipDeviceProp_t::multiProcessorCount, cached per device): 256 on a whole MI355X,
// fewer on a partitioned one (CPX / NPS modes expose a slice of the chip per device).  What the launch planners size
// their rounds of resident workgroups by.  256 when no device can be queried (the planners also run on GPU-less hosts:
// workspace sizes, the build-time kernel generator).
int device_cu_count();

// 64-lane wavefront sum, result valid in every lane (fixed butterfly order -> deterministic).
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace parrm

namespace parrm_nmcore {
struct Core;
}
namespace parrm {
// The Nelder-Mead refinement as a device-side chain (parrm_period.hip; state machine: parrm_nm_core.h).
size_t nm_chain_workspace_bytes(int64_t n_idx, int64_t n_chans, int bw);
int nm_chain_run(const parrm_nmcore::Core &init, const double *d_y, int64_t ldy, const int64_t *d_idx, int64_t n_idx,
                 int64_t n_chans, int bw, double lambda, void *d_workspace, size_t workspace_bytes, void *stream, double *hist_x,
                 double *hist_f, int hist_capacity, int *batch_sizes, int batch_capacity, int *n_batches, int *hist_used,
                 double *res_x, double *res_f, int *res_its, int *res_calls, bool *handled);
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
