// valu_peak.hip -- what is the integer VALU issue rate of gfx950 for the ops the pair kernel uses?
// Pure register loops, no memory: v_and_b32 / v_and_or_b32 (SGPR operand) / v_bcnt_u32_b32 / v_or_b32.
// Build: hipcc -O3 --offload-arch=gfx950 valu_peak.hip -o valu_peak ; run: ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned s0, unsigned s1, unsigned s2, unsigned s3, int iters)
{
    unsigned a[8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; acc[i] = i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) {            // the pair kernel's mix: and, 3 x and_or(sgpr), bcnt, or(sgpr), bcnt
                    unsigned m;
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(m) : "s"(s0), "v"(a[i]));
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(m) : "s"(s1), "v"(a[(i + 1) & 7]));
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(m) : "s"(s2), "v"(a[(i + 2) & 7]));
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(m) : "s"(s3), "v"(a[(i + 3) & 7]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(m));
                    asm volatile("v_or_b32 %0, %1, %2" : "=v"(m) : "s"(s0), "v"(a[(i + 4) & 7]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[(i + 5) & 7]) : "v"(m));
                } else if (MODE == 1) {     // 7 x v_and_b32 (VOP2, sgpr src0)
#pragma unroll
                    for (int q = 0; q < 7; q++) asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[i]) : "s"(s0));
                } else if (MODE == 2) {     // 7 x v_bcnt
#pragma unroll
                    for (int q = 0; q < 7; q++) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
                } else if (MODE == 3) {     // 7 x v_and_or (all VGPR)
#pragma unroll
                    for (int q = 0; q < 7; q++) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                } else if (MODE == 4) {     // 7 x v_fma_f32
#pragma unroll
                    for (int q = 0; q < 7; q++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                } else if (MODE == 5) {     // 7 x v_pk_fma_f32 (2 regs)
#pragma unroll
                    for (int q = 0; q < 7; q++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
                }
            }
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char *name, int blocks_per_cu)
{
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    unsigned *d;
    CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x0f0f0f0fu, 0x33333333u, 0x55555555u, 0xff00ff00u, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x0f0f0f0fu, 0x33333333u, 0x55555555u, 0xff00ff00u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 /*waves*/ * iters * 8 * 8 * 7;
    const double laneops = instr * 64;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f Tlane-op/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", name, blocks_per_cu, ms,
           laneops / ms / 1e9, (ms * 1e-3 * 2.4e9) / (instr / 1024.0));
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    for (int b = 1; b <= 8; b *= 2) {
        run<0>("pair mix (and,3and_or,bcnt,or,bcnt)", b);
        run<1>("v_and_b32 (sgpr)", b);
        run<2>("v_bcnt_u32_b32", b);
        run<3>("v_and_or_b32", b);
        run<4>("v_fma_f32", b);
        run<5>("v_add_u32", b);
    }
    return 0;
}
