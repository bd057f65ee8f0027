// Micro-benchmark 3: how many waves per CU does the stretch walk need on its memory side?
// Hand-shaped like the phase kernel's interior loop (buffer descriptors, wave-uniform scalar offsets,
// two alternating register sets, LDS-only barrier), so that hipcc's vmcnt bookkeeping is the same.
// A workgroup of NW waves walks a (channel, stretch); per iteration every wave requests LPW rows-worth
// of 64 x 8 B (for the iteration after next), drops the previous request into LDS, reads it back and
// stores LPW x 64 x 8 B of outputs.  STRIDE3 stores with a 24-byte lane stride (3 residues per lane).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/membench3 scripts/membench3.hip && /tmp/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int LPW, bool STRIDE3>
__global__ void __launch_bounds__(1024) walk5(const double *x, double *y, long ld, long chunk, long iters, long n_stretch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int nw = blockDim.x >> 6;
    const long c = blockIdx.x / n_stretch, st = blockIdx.x - c * n_stretch;
    const long s0 = st * iters * chunk;  // doubles
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x + c * ld), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + c * ld, 0, 0x7ffffff0, 0x00020000);
    // per iteration the workgroup covers `chunk` = nw * LPW * 64 doubles; wave w takes LPW consecutive 64-double rows
    const unsigned voff = l * 8u;
    const unsigned voff_st = STRIDE3 ? (unsigned)((l * 3) * 8u) : l * 8u;
    unsigned soff_x = (unsigned)((s0 + 2 * chunk + (long)w * LPW * 64) * 8);
    unsigned soff_y = (unsigned)((s0 + (long)w * LPW * 64) * 8);
    const unsigned step = (unsigned)(chunk * 8);
    double *cell = reinterpret_cast<double *>(lds_raw) + (w * LPW * 64 + l);
    double a[LPW], b[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        a[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, voff, soff_x - step + i * 512u, 0));
    }
    auto iteration = [&](double (&pub)[LPW], double (&req)[LPW]) {
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            req[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, voff, soff_x + i * 512u, 0));
        soff_x += step;
        double out[LPW];
#pragma unroll
        for (int i = 0; i < LPW; ++i) out[i] = cell[i * 64] * 1.0000001;
#pragma unroll
        for (int i = 0; i < LPW; ++i) cell[i * 64] = pub[i];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            if (STRIDE3) {  // lane j stores "residues" 3j+k of a 192-wide row: 24-byte lane stride
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, out[i]), ry, voff_st + (i % 3) * 8u,
                                                      soff_y + (i / 3) * 1536u, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, out[i]), ry, voff_st, soff_y + i * 512u, 0);
            }
        }
        soff_y += step;
    };
    for (long k = 0; k + 1 < iters; k += 2) {
        iteration(a, b);
        iteration(b, a);
    }
    (void)nw;
}

int main() {
    const long C = 256, N = 10000000;
    double *x, *y;
    CK(hipMalloc(&x, C * N * 8 + (1 << 24)));
    CK(hipMalloc(&y, C * N * 8 + (1 << 24)));
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
        printf("%-60s best %7.3f ms  mean %7.3f  %7.1f GB/s\n", name, best, sum / 5, 2.0 * C * N * 8 / best / 1e6);
        fflush(stdout);
    };
#define RUN5(LPW, S3, NW, LDSKB)                                                                                  \
    do {                                                                                                          \
        const long chunk = (long)NW * LPW * 64, iters = 262144 / chunk / 2 * 2;                                     \
        const long n_stretch = N / (iters * chunk);                                                                 \
        char name[160];                                                                                             \
        snprintf(name, sizeof name, "walk5 waves/WG=%d loads/wave=%d stride3=%d LDS %d KB (%ld WGs)", NW, LPW, (int)S3, LDSKB, C * n_stretch); \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(walk5<LPW, S3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024)); \
        time(name, [&] { walk5<LPW, S3><<<C * n_stretch, NW * 64, LDSKB * 1024>>>(x, y, N, chunk, iters, n_stretch); }); \
    } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN5(2, false, 12, 80);   // round-1 kernel: 24 waves per CU, 2 rows per wave
        RUN5(6, false, 4, 80);    // 3 residues per lane, R = 2: 8 waves per CU
        RUN5(6, true, 4, 80);
        RUN5(3, false, 8, 80);    // 3 residues per lane, R = 1: 16 waves per CU
        RUN5(3, true, 8, 80);
        RUN5(6, true, 8, 80);     // 16 waves per CU, R = 2
        RUN5(6, true, 6, 52);     // 18 waves per CU
        RUN5(12, true, 4, 80);    // R = 4
        RUN5(4, false, 6, 80);    // 12 waves per CU
    }
    return 0;
}
