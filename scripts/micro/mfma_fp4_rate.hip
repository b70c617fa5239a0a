// mfma_fp4_rate.hip -- sustained issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands (registers only),
// by operand content (zeros / the +-1 patterns of the pair kernel) and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, unsigned pattern, int scale, int iters)
{
    v16f acc[8];
    for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    v8i opa[4], opb[4];
    for (int i = 0; i < 4; i++) {
        unsigned w = pattern ? (pattern ^ (threadIdx.x * 0x9E3779B1u * (i + 1))) & 0xAAAAAAAAu | 0x22222222u : 0u;   // nibbles 0x2 / 0xA = +-1.0
        opa[i] = v8i{(int)w, (int)(w * 3u & 0xAAAAAAAAu | (pattern ? 0x22222222u : 0u)), (int)w, (int)w, 0, 0, 0, 0};
        opb[i] = v8i{(int)(w ^ (pattern ? 0x80808080u : 0u)), (int)w, (int)w, (int)w, 0, 0, 0, 0};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 8; a++) {
            acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(opa[a & 3], opb[(a + 1) & 3], acc[a], 4, 4, 0, scale, 0, scale);
            acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(opa[(a + 2) & 3], opb[a & 3], acc[a], 4, 4, 0, scale, 0, scale);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 8; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k16(float *out, unsigned pattern, int scale, int iters)
{
    v4f acc[16];
    for (int a = 0; a < 16; a++) for (int r = 0; r < 4; r++) acc[a][r] = 0.f;
    v8i opa[4], opb[4];
    for (int i = 0; i < 4; i++) {
        unsigned w = pattern ? (pattern ^ (threadIdx.x * 0x9E3779B1u * (i + 1))) & 0xAAAAAAAAu | 0x22222222u : 0u;
        opa[i] = v8i{(int)w, (int)(w * 3u & 0xAAAAAAAAu | (pattern ? 0x22222222u : 0u)), (int)w, (int)w, 0, 0, 0, 0};
        opb[i] = v8i{(int)(w ^ (pattern ? 0x80808080u : 0u)), (int)w, (int)w, (int)w, 0, 0, 0, 0};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 16; a++) {
            acc[a] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opa[a & 3], opb[(a + 1) & 3], acc[a], 4, 4, 0, scale, 0, scale);
            acc[a] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opa[(a + 2) & 3], opb[a & 3], acc[a], 4, 4, 0, scale, 0, scale);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 16; a++) for (int r = 0; r < 4; r++) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int run16(const char *name, unsigned pattern, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000;
    float *d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, pattern, 127, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, d, pattern, 127, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)blocks * 4 * iters * 32;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  %.1f nominal cycles/MFMA/SIMD @2.4GHz\n", name, ms, mfma * 65536.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
    CHECK(hipFree(d)); return 0;
}
int run(const char *name, unsigned pattern, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000;
    float *d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pattern, 127, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, pattern, 127, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)blocks * 4 * iters * 16;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  %.1f nominal cycles/MFMA/SIMD @2.4GHz\n", name, ms, mfma * 131072.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
    CHECK(hipFree(d)); return 0;
}
int main()
{
    run("fp4 32x32x64, zero operands, 1 wave/SIMD", 0u, 1);
    run("fp4 32x32x64, +-1 operands, 1 wave/SIMD", 0x5A5A1234u, 1);
    run("fp4 32x32x64, +-1 operands, 2 waves/SIMD", 0x5A5A1234u, 2);
    run("fp4 32x32x64, zero operands, 2 waves/SIMD", 0u, 2);
    run16("fp4 16x16x128, +-1 operands, 1 wave/SIMD", 0x5A5A1234u, 1);
    run16("fp4 16x16x128, +-1 operands, 2 waves/SIMD", 0x5A5A1234u, 2);
    run16("fp4 16x16x128, zero operands, 2 waves/SIMD", 0u, 2);
    return 0;
}
