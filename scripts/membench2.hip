// Micro-benchmark 2: which global-memory access shape should the phase kernel's load / store legs
// use?  One workgroup walks one (channel, stretch) in iterations of NGR rows of q doubles, as the
// filter kernel does (per-iteration barrier, loads requested D iterations ahead, 2 workgroups per CU
// forced through an LDS reservation).  Only the lane -> address mapping differs between the modes:
//   0  row mapping, 8 B per lane: lane r of row group g touches element (row, r)       (round 1)
//   1  chunk mapping, 16 B per lane: the NGR*q contiguous doubles of an iteration as double2
//   2  pair mapping, 16 B per lane: even lanes take (row i, r..r+1), odd lanes (row i+1, r-1..r)
//   3  chunk mapping, 8 B per lane (elements t, t+T of the chunk)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench2 scripts/membench2.hip && /tmp/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ void copy_linear(const double2_t *x, double2_t *y, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) y[i] = x[i] * 1.0000001;
}

template <int MODE_, int D, bool NT>
__global__ void __launch_bounds__(1024) walk(const double *x, double *y, long ld, int q, int qp, int ng,
                                              long stretch_rows, long n_stretch, long n_samples, int misalign) {
    extern __shared__ double lds_pad[];
    if (threadIdx.x == 1025) lds_pad[0] = 0.0;
    constexpr int R = 2;
    constexpr int MODE = MODE_ >= 4 ? 0 : MODE_;
    constexpr bool kLoads = MODE_ != 5, kStores = MODE_ != 4, kBarrier = MODE_ != 6;
    const int tid = threadIdx.x;
    const int g = tid / qp, r = tid - g * qp;
    const long c = blockIdx.x / n_stretch, st = blockIdx.x - c * n_stretch;
    const long s0 = st * stretch_rows * q + misalign;
    const double *xr = x + c * ld;
    double *yr = y + c * ld;
    const int ngr = ng * R;
    const long chunk = (long)ngr * q;
    const long lim = n_samples - 2 * chunk;  // keep every access in range without per-lane tests near the end
    double2_t buf[D + 1][R] = {};  // MODE 0/3 use .x only... (two 8-byte values for R rows, or one double2)

    auto ld8 = [&](const double *p) -> double { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto st8 = [&](double *p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    auto ld16 = [&](const double *p) -> double2_t {
        const double2_t *pp = reinterpret_cast<const double2_t *>(p);
        return NT ? __builtin_nontemporal_load(pp) : *pp;
    };
    auto st16 = [&](double *p, double2_t v) {
        double2_t *pp = reinterpret_cast<double2_t *>(p);
        if (NT) __builtin_nontemporal_store(v, pp); else *pp = v;
    };

    auto request = [&](long mk, double2_t (&v)[R]) {
        const long base = s0 + mk * q;
        if (base > lim || !kLoads) return;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (r < q) v[i].x = ld8(xr + base + (long)(g * R + i) * q + r);
        } else if constexpr (MODE == 1) {
            const long e = 2L * tid;
            if (e < chunk) v[0] = ld16(xr + base + e);
        } else if constexpr (MODE == 2) {
            // even lane: row g*R, columns r, r+1; odd lane: row g*R+1, columns r-1, r
            const int col = r & ~1;
            if (col + 1 < q) v[0] = ld16(xr + base + (long)(g * R + (r & 1)) * q + col);
            else if (col < q) v[0].x = ld8(xr + base + (long)(g * R + (r & 1)) * q + col);
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const long e = tid + (long)i * blockDim.x;
                if (e < chunk) v[i].x = ld8(xr + base + e);
            }
        }
    };
    auto emit = [&](long mk, const double2_t (&v)[R]) {
        const long base = s0 + mk * q;
        if (base > lim) return;
        if (!kStores) { if (v[0].x == 12345.678) st8(yr, v[1].x); return; }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (r < q) st8(yr + base + (long)(g * R + i) * q + r, v[i].x * 1.0000001);
        } else if constexpr (MODE == 1) {
            const long e = 2L * tid;
            if (e < chunk) st16(yr + base + e, v[0] * 1.0000001);
        } else if constexpr (MODE == 2) {
            const int col = r & ~1;
            if (col + 1 < q) st16(yr + base + (long)(g * R + (r & 1)) * q + col, v[0] * 1.0000001);
            else if (col < q) st8(yr + base + (long)(g * R + (r & 1)) * q + col, v[0].x * 1.0000001);
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const long e = tid + (long)i * blockDim.x;
                if (e < chunk) st8(yr + base + e, v[i].x * 1.0000001);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) request((long)d * ngr, buf[d]);
    long mk = 0;
    while (mk < stretch_rows) {
#pragma unroll
        for (int d = 0; d <= D; ++d) {  // rotate statically through the D+1 register sets
            if (mk < stretch_rows) {
                request(mk + (long)D * ngr, buf[(d + D) % (D + 1)]);
                emit(mk, buf[d]);
                if (kBarrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                mk += ngr;
            }
        }
    }
}

int main(int argc, char **argv) {
    const long C = 256, N = 10000000;
    double *x, *y;
    CK(hipMalloc(&x, C * N * 8));
    CK(hipMalloc(&y, C * N * 8));
    CK(hipMemset(x, 1, C * N * 8));
    CK(hipMemset(y, 0, C * N * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int k = 0; k < 5; ++k) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-64s best %7.3f ms  mean %7.3f  %7.1f GB/s\n", name, best, sum / 5, 2.0 * C * N * 8 / best / 1e6);
        fflush(stdout);
    };
    const int q = 169, qp = 192;
    const size_t lds = 80 * 1024;
#define RUN(MODE, D, NT, NG, MIS, ROWS)                                                                         \
    do {                                                                                                        \
        const long rows = ROWS, n_stretch = (N + rows * q - 1) / (rows * q);                                     \
        char name[160];                                                                                          \
        snprintf(name, sizeof name, "mode %d D=%d nt=%d ng=%d misalign=%d rows=%ld", MODE, D, (int)NT, NG, MIS, rows); \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(walk<MODE, D, NT>),                                \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        time(name, [&] { walk<MODE, D, NT><<<C * n_stretch, NG * qp, lds>>>(x, y, N, q, qp, NG, rows, n_stretch, N, MIS); }); \
    } while (0)
    time("linear double2 copy 16384x256", [&] { copy_linear<<<16384, 256>>>((const double2_t *)x, (double2_t *)y, C * N / 2); });
    time("linear double2 copy 2048x1024", [&] { copy_linear<<<2048, 1024>>>((const double2_t *)x, (double2_t *)y, C * N / 2); });
    long rows_list[] = {1552, 400, 200, 96, 48, 24};
    for (long rw : rows_list) {
        RUN(0, 2, false, 4, 0, rw);
        RUN(1, 2, false, 4, 0, rw);
    }
    RUN(4, 2, false, 4, 0, 1552);  // loads only
    RUN(5, 2, false, 4, 0, 1552);  // stores only
    RUN(6, 2, false, 4, 0, 1552);  // no barrier
    time("linear double2 copy 16384x256", [&] { copy_linear<<<16384, 256>>>((const double2_t *)x, (double2_t *)y, C * N / 2); });
    return 0;
}
