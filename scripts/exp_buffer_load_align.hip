// What do buffer loads of 32 / 64 bits return at addresses that are 0 / 4 mod 8?  (raw buffer, offen)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, int n, float *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, n * 4, 0x00020000);
    const int t = threadIdx.x;  // lane t reads the pair starting at float t
    const u32 a = __builtin_amdgcn_raw_buffer_load_b32(r, t * 4, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b32(r, t * 4 + 4, 0, 0);
    const u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(r, t * 4, 0, 0);
    u32x2 q;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(q) : "v"((u32)(t * 4)), "s"(r) : "memory");
    u32 c, d;
    const u32 soff = 8;
    asm volatile("buffer_load_dword %0, %2, %3, %4 offen\n\tbuffer_load_dword %1, %2, %3, %4 offen offset:4\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(c), "=&v"(d) : "v"((u32)(t * 4)), "s"(r), "s"(soff) : "memory");
    out[8 * t + 0] = __builtin_bit_cast(float, a);
    out[8 * t + 1] = __builtin_bit_cast(float, b);
    out[8 * t + 2] = __builtin_bit_cast(float, p.x);
    out[8 * t + 3] = __builtin_bit_cast(float, p.y);
    out[8 * t + 4] = __builtin_bit_cast(float, q.x);
    out[8 * t + 5] = __builtin_bit_cast(float, q.y);
    out[8 * t + 6] = __builtin_bit_cast(float, c);
    out[8 * t + 7] = __builtin_bit_cast(float, d);
}
int main() {
    const int n = 64;
    float h[n], *dx, *dout, o[8 * 8];
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 8 * 8 * 4);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, dx, n, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t)
        printf("lane %d (floats %d,%d): b32+b32 (%g,%g)  b64 (%g,%g)  asm dwordx2 (%g,%g)  asm dword+dword soff 8 (%g,%g)\n", t, t, t + 1,
               o[8 * t], o[8 * t + 1], o[8 * t + 2], o[8 * t + 3], o[8 * t + 4], o[8 * t + 5], o[8 * t + 6], o[8 * t + 7]);
    return 0;
}
