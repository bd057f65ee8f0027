// CAUTION (round 2): only the `walk` modes 0/1/2/3 of this file produced usable numbers (8 B vs 16 B per
// lane: within 2 % of each other).  The later kernels here (walk3, walk4: per-lane bounds tests inside the
// loop) compile to vmcnt(0) waits, branches and scratch spills -- their 12-28 ms measure hipcc, not the
// memory system.  scripts/membench3.hip is the hand-shaped replacement (buffer descriptors, counted waits)
// and the one DESIGN.md 4.2 quotes.
//
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

// The three-residues-per-lane kernel's memory side: one-wave row groups (64 lanes), NG groups per
// workgroup, R = 2 rows per group and iteration.  Loads: contiguous, lane l takes columns l, l+64, l+128.
// Stores: STRIDED = lane j stores columns 3j, 3j+1, 3j+2 (24-byte lane stride, what the compute mapping
// holds); otherwise contiguous like the loads.
template <int LMODE, int SMODE, int D, bool BARRIER>
__global__ void __launch_bounds__(1024) walk3(const double *x, double *y, long ld, int q, int ng, long stretch_rows,
                                               long n_stretch, long n_samples) {
    extern __shared__ double lds_pad[];
    if (threadIdx.x == 1025) lds_pad[0] = 0.0;
    constexpr int R = 2;
    const int tid = threadIdx.x, g = tid >> 6, l = tid & 63;
    const long c = blockIdx.x / n_stretch, st = blockIdx.x - c * n_stretch;
    const long s0 = st * stretch_rows * q;
    const double *xr = x + c * ld;
    double *yr = y + c * ld;
    const int ngr = ng * R;
    const long lim = n_samples - 4L * ngr * q;
    double2_t buf[D + 1][3] = {};  // R*3 doubles
    auto request = [&](long mk, double2_t (&v)[3]) {
        const long base = s0 + (mk + g * R) * q;
        if (base > lim) return;
        if constexpr (LMODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (l + 64 * k < q) {
                    v[k].x = xr[base + l + 64 * k];
                    v[k].y = xr[base + q + l + 64 * k];
                }
        } else if constexpr (LMODE == 2) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (2 * (l + 64 * k) < R * q) v[k] = *reinterpret_cast<const double2_t *>(xr + base + 2 * (l + 64 * k));
        }
    };
    auto emit = [&](long mk, const double2_t (&v)[3]) {
        const long base = s0 + (mk + g * R) * q;
        if (base > lim) return;
        if constexpr (SMODE == 0) {
            if (v[0].x == 12345.678) yr[0] = v[1].y + v[2].x + v[2].y + v[0].y + v[1].x;
        } else if constexpr (SMODE == 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (2 * (l + 64 * k) < R * q) *reinterpret_cast<double2_t *>(yr + base + 2 * (l + 64 * k)) = v[k] * 1.0000001;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int col = SMODE == 2 ? 3 * l + k : l + 64 * k;
                if (col < q) {
                    yr[base + col] = v[k].x * 1.0000001;
                    yr[base + q + col] = v[k].y * 1.0000001;
                }
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) request((long)d * ngr, buf[d]);
    long mk = 0;
    while (mk < stretch_rows) {
#pragma unroll
        for (int d = 0; d <= D; ++d) {
            if (mk < stretch_rows) {
                request(mk + (long)D * ngr, buf[(d + D) % (D + 1)]);
                emit(mk, buf[d]);
                if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                mk += ngr;
            }
        }
    }
}

// Specialised skeleton: NG store-only waves + NL load-only waves per workgroup.  Loader waves pull the
// NGR*q contiguous doubles of iteration k+D as 16-byte pieces and drop iteration k+1's into an LDS staging
// ring; after the barrier each store wave reads its 2 rows x 3 columns (compute mapping: lane j owns
// columns 3j..3j+2) and stores them (24-byte lane stride).  No wave has loads AND stores in its vmcnt queue.
template <int D, int NL>
__global__ void __launch_bounds__(1024) walk4(const double *x, double *y, long ld, int q, int ng, long stretch_rows,
                                               long n_stretch, long n_samples) {
    extern __shared__ double lds[];
    constexpr int R = 2;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const long c = blockIdx.x / n_stretch, st = blockIdx.x - c * n_stretch;
    const long s0 = st * stretch_rows * q;
    const double *xr = x + c * ld;
    double *yr = y + c * ld;
    const int ngr = ng * R;
    const int chunk = ngr * q;            // doubles per iteration
    const int pieces = (chunk + 1) / 2;   // double2 pieces
    const long lim = n_samples - 4L * chunk;
    const bool loader = w >= ng;
    const int lw = w - ng;                // loader index
    constexpr int PMAX = 12;              // pieces per loader lane (NL*64*PMAX >= pieces)
    double2_t buf[D][PMAX] = {};
    auto request = [&](long mk, double2_t (&v)[PMAX]) {
        const long base = s0 + mk * q;
        if (base > lim) return;
#pragma unroll
        for (int t = 0; t < PMAX; ++t) {
            const int pc = (t * NL + lw) * 64 + l;
            if (pc < pieces) v[t] = *reinterpret_cast<const double2_t *>(xr + base + 2 * pc);
        }
    };
    auto publish = [&](int slot, const double2_t (&v)[PMAX]) {
#pragma unroll
        for (int t = 0; t < PMAX; ++t) {
            const int pc = (t * NL + lw) * 64 + l;
            if (pc < pieces) *reinterpret_cast<double2_t *>(lds + (size_t)slot * (chunk + 2) + 2 * pc) = v[t];
        }
    };
    if (loader) {
#pragma unroll
        for (int d = 0; d < D; ++d) request((long)d * ngr, buf[d]);
    }
    long mk = 0;
    int it = 0;
    while (mk < stretch_rows) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (mk < stretch_rows) {
                if (loader) {
                    publish(it & 1, buf[d]);                 // rows of iteration `it` (requested D iterations ago)
                    request(mk + (long)D * ngr, buf[d]);     // refill the same register set
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!loader) {
                    const long base = s0 + mk * q;
                    if (base <= lim) {
                        const double *src = lds + (size_t)(it & 1) * (chunk + 2) + (w * R) * q;
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const int col = 3 * l + k;
                                if (col < q) yr[base + (long)(w * R + i) * q + col] = src[i * q + col] * 1.0000001;
                            }
                    }
                }
                mk += ngr;
                ++it;
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
    RUN(0, 2, false, 4, 0, 1552);
#define RUN3(LM, SM, D, BAR, NG, LDSKB)                                                                        \
    do {                                                                                                       \
        const long rows = 1552, n_stretch = (N + rows * q - 1) / (rows * q);                                     \
        char name[160];                                                                                          \
        snprintf(name, sizeof name, "walk3 load=%d store=%d D=%d barrier=%d ng=%d LDS %d KB", LM, SM, D, (int)BAR, NG, LDSKB); \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(walk3<LM, SM, D, BAR>),                            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024));                       \
        time(name, [&] { walk3<LM, SM, D, BAR><<<C * n_stretch, NG * 64, LDSKB * 1024>>>(x, y, N, q, NG, rows, n_stretch, N); }); \
    } while (0)
    RUN3(1, 1, 2, true, 4, 80);
#define RUN4(D, NL, NG, LDSKB)                                                                                  \
    do {                                                                                                       \
        const long rows = 1552, n_stretch = (N + rows * q - 1) / (rows * q);                                     \
        char name[160];                                                                                          \
        snprintf(name, sizeof name, "walk4 D=%d loaders=%d ng=%d LDS %d KB", D, NL, NG, LDSKB);                  \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(walk4<D, NL>),                                     \
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024));                       \
        time(name, [&] { walk4<D, NL><<<C * n_stretch, (NG + NL) * 64, LDSKB * 1024>>>(x, y, N, q, NG, rows, n_stretch, N); }); \
    } while (0)
    RUN4(2, 1, 4, 80);
    RUN4(3, 1, 4, 80);
    RUN4(2, 2, 4, 80);
    RUN4(3, 2, 4, 80);
    RUN4(4, 2, 4, 80);
    RUN4(2, 1, 4, 52);
    RUN4(2, 2, 4, 52);
    RUN4(2, 2, 8, 80);
    return 0;
}
