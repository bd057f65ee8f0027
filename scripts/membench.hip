// Micro-benchmark: what does the memory system deliver for the stretch-streaming access pattern
// of the filter kernels (one workgroup walks one channel-stretch in rows of q doubles)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench scripts/membench.hip && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void copy_linear(const double2 *x, double2 *y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = x[i];
}

// mode bit0: barrier per iteration; bit1: prefetch distance 2 iterations
template <int R, int MODE>
__global__ void __launch_bounds__(1024) copy_stretch(const double *x, double *y, long ld, int q, int qp, int ng,
                                                      long stretch_rows, long n_stretch, long n_samples) {
    extern __shared__ double lds_pad[];  // only to limit workgroups per CU like the filter kernel
    if (threadIdx.x == 1025) lds_pad[0] = 0.0;
    const int tid = threadIdx.x;
    const int g = tid / qp, r = tid - g * qp;
    const long c = blockIdx.x / n_stretch, st = blockIdx.x - c * n_stretch;
    const long s0 = st * stretch_rows * q;
    const double *xr = x + c * ld;
    double *yr = y + c * ld;
    const int ngr = ng * R;
    double cur[R], nxt[R];
    auto ld_rows = [&](long mk, double (&v)[R]) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long n = s0 + (mk + g * R + i) * q + r;
            v[i] = (r < q && n < n_samples) ? xr[n] : 0.0;
        }
    };
    ld_rows(0, cur);
    for (long mk = 0; mk < stretch_rows; mk += ngr) {
        if (mk + ngr < stretch_rows) ld_rows(mk + ngr, nxt);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long n = s0 + (mk + g * R + i) * q + r;
            if (r < q && n < n_samples) yr[n] = cur[i] * 1.0000001;
        }
        if (MODE & 1) __syncthreads();
#pragma unroll
        for (int i = 0; i < R; ++i) cur[i] = nxt[i];
    }
}

int main(int argc, char **argv) {
    const long C = 256, N = 10000000;
    double *x, *y;
    CK(hipMalloc(&x, C * N * 8));
    CK(hipMalloc(&y, C * N * 8));
    CK(hipMemset(x, 0, C * N * 8));
    CK(hipMemset(y, 0, C * N * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int k = 0; k < 3; ++k) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, 2.0 * C * N * 8 / best / 1e6);
    };
    time("linear double2 copy, 2048x256", [&] { copy_linear<<<2048, 256>>>((const double2 *)x, (double2 *)y, C * N / 2); });
    time("linear double2 copy, 16384x256", [&] { copy_linear<<<16384, 256>>>((const double2 *)x, (double2 *)y, C * N / 2); });
    for (int q : {169, 192}) {
        for (long rows : {768L, 6144L}) {
            const int qp = 192, ng = 2;
            const long n_stretch = (N + rows * q - 1) / (rows * q);
            char name[128];
            snprintf(name, sizeof name, "stretch q=%d rows=%ld R=4 ng=2 nobarrier", q, rows);
            time(name, [&] { copy_stretch<4, 0><<<C * n_stretch, ng * qp>>>(x, y, N, q, qp, ng, rows, n_stretch, N); });
            snprintf(name, sizeof name, "stretch q=%d rows=%ld R=4 ng=2 barrier", q, rows);
            time(name, [&] { copy_stretch<4, 1><<<C * n_stretch, ng * qp>>>(x, y, N, q, qp, ng, rows, n_stretch, N); });
        }
    }
    for (size_t lds : {size_t(0), size_t(40 * 1024), size_t(80 * 1024)}) {
        const int q = 169, qp = 192, ng = 2;
        const long rows = 768, n_stretch = (N + rows * q - 1) / (rows * q);
        char name[128];
        snprintf(name, sizeof name, "stretch q=169 R=4 ng=2 barrier, LDS %zu KB/block", lds / 1024);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(copy_stretch<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        time(name, [&] { copy_stretch<4, 1><<<C * n_stretch, ng * qp, lds>>>(x, y, N, q, qp, ng, rows, n_stretch, N); });
    }
    {
        const int q = 169, qp = 192, ng = 4;
        const long rows = 768, n_stretch = (N + rows * q - 1) / (rows * q);
        time("stretch q=169 rows=768 R=4 ng=4 barrier", [&] { copy_stretch<4, 1><<<C * n_stretch, ng * qp>>>(x, y, N, q, qp, ng, rows, n_stretch, N); });
    }
    return 0;
}
