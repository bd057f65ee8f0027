// Does a launch with a large dynamic LDS allocation start later?  An empty kernel behind a 100 us spinner, launch-to-done
// latency by dynamic LDS size (hipcc --offload-arch=gfx950 -O2 exp_launch_lds.hip -o /tmp/exp_launch_lds)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
extern __shared__ double lds[];
__global__ void touch(double *out, int n) {
    if (n > 0) lds[threadIdx.x] = 1.0;
    __syncthreads();
    if (n > 0 && threadIdx.x == 0) out[blockIdx.x] = lds[n - 1];
}
int main() {
    double *d;
    hipMalloc(&d, 1 << 20);
    hipStream_t s;
    hipStreamCreate(&s);
    const int sizes[] = {0, 16384, 49152, 65536, 68608, 81920, 131072};
    for (int bytes : sizes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(touch), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            printf("attribute %d failed\n", bytes);
        for (int grid : {8, 512}) {
            double best = 1e9, sum = 0;
            for (int rep = 0; rep < 200; ++rep) {
                hipStreamSynchronize(s);
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(touch, dim3(grid), dim3(256), bytes, s, d, bytes ? 64 : 0);
                hipStreamSynchronize(s);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep >= 20) { best = us < best ? us : best; sum += us; }
            }
            printf("dynamic LDS %6d B, grid %3d: launch+sync best %.1f us, mean %.1f us\n", bytes, grid, best, sum / 180);
        }
    }
    return 0;
}
