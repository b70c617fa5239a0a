// valu_ops.hip -- per-opcode issue cost on gfx950 with INDEPENDENT chains (24 accumulators per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NA = 24;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned s0, unsigned s1, int iters)
{
    unsigned a[NA], b[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) { a[i] = threadIdx.x * 2654435761u + i; b[i] = a[i] ^ (s0 + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (MODE == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 1) asm volatile("v_and_b32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 2) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
            if (MODE == 3) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 5) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 6) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 7) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 8) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s1), "v"(b[i]));
            if (MODE == 9) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(a[i]) : "s"(s1), "v"(b[i]));
            if (MODE == 10) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 11) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 12) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % NA]));
            if (MODE == 13) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 14) {   // general pair-word: and, 3 x bitop3, bcnt, or, bcnt (all VGPR)
                unsigned m;
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(m) : "v"(b[i]), "v"(b[(i + 1) % NA]));
                asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xea" : "+v"(m) : "v"(b[(i + 2) % NA]), "v"(b[(i + 3) % NA]));
                asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xea" : "+v"(m) : "v"(b[(i + 4) % NA]), "v"(b[(i + 5) % NA]));
                asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xea" : "+v"(m) : "v"(b[(i + 6) % NA]), "v"(b[(i + 7) % NA]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                asm volatile("v_or_b32 %0, %1, %2" : "=v"(m) : "v"(b[(i + 8) % NA]), "v"(b[(i + 9) % NA]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(i + 1) % NA]) : "v"(m));
            }
            if (MODE == 16) asm volatile("v_xor_b32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            if (MODE == 17) {   // consensus pair-word with the row operand taken through DPP row_newbcast (no broadcast LDS reads)
                unsigned v, t, u;
                asm volatile("v_and_b32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(b[i]), "v"(b[(i + 1) % NA]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(v));
                asm volatile("v_xor_b32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[(i + 2) % NA]), "v"(b[(i + 3) % NA]));
                asm volatile("v_xor_b32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(u) : "v"(b[(i + 4) % NA]), "v"(b[(i + 5) % NA]));
                asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xa8" : "+v"(v) : "v"(t), "v"(u));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(i + 1) % NA]) : "v"(v));
            }
            if (MODE == 15) {   // consensus pair-word: and, bcnt, xor, xor, bitop3, bcnt
                unsigned v, t, u;
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(v) : "v"(b[i]), "v"(b[(i + 1) % NA]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(v));
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "v"(b[(i + 2) % NA]), "v"(b[(i + 3) % NA]));
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u) : "v"(b[(i + 4) % NA]), "v"(b[(i + 5) % NA]));
                asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xa8" : "+v"(v) : "v"(t), "v"(u));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(i + 1) % NA]) : "v"(v));
            }
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < NA; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> int run(const char *name, int waves_per_simd = 4)
{
    const int blocks = 256 * waves_per_simd, iters = 40000;     // 4 waves per SIMD by default, ~5 ms per run
    unsigned *d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 * iters * NA * (MODE == 14 ? 7 : (MODE == 15 || MODE == 17) ? 6 : 1);
    printf("%-34s %.3f ms  %.2f Tlane-op/s  %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, instr * 64 / ms / 1e9, (ms * 1e-3 * 2.4e9) / (instr / 1024.0));
    CHECK(hipFree(d)); return 0;
}
int main()
{
    run<0>("v_and_b32 e32 (VOP2)"); run<1>("v_and_b32 e64 (VOP3 encoding)"); run<2>("v_xor_b32 sgpr"); run<3>("v_and_or_b32 vvv");
    run<8>("v_and_or_b32 svv"); run<4>("v_bcnt_u32_b32"); run<5>("v_bitop3_b32 vvv"); run<9>("v_bitop3_b32 svv"); run<6>("v_or3_b32");
    run<7>("v_add_u32"); run<10>("v_mad_u32_u24"); run<11>("v_bfi_b32"); run<12>("v_dot4_u32_u8"); run<13>("v_pk_add_u16");
    run<14>("general pair-word mix (7 ops)"); run<15>("consensus pair-word mix (6 ops)");
    run<16>("v_xor_b32_dpp row_newbcast"); run<17>("consensus mix, rows via DPP row_newbcast");
    run<15>("consensus mix, 2 waves/SIMD", 2); run<15>("consensus mix, 3 waves/SIMD", 3); run<15>("consensus mix, 5 waves/SIMD", 5);
    run<15>("consensus mix, 6 waves/SIMD", 6); run<15>("consensus mix, 7 waves/SIMD", 7); run<15>("consensus mix, 8 waves/SIMD", 8); run<17>("consensus mix via DPP, 5 waves/SIMD", 5); run<14>("general mix, 2 waves/SIMD", 2);
    return 0;
}
