// Microbenchmark: random whole 128-byte lines against random 64-byte half lines out of a region the Infinity Cache can hold
// (what nn_rows_kernel's walks fetch: one n8 line per (row, site); would a walk that needs half of the line cost half?).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/rand_lines.hip -o /tmp/rand_lines && /tmp/rand_lines [region MiB] [lines per wave]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// LANES lanes of 16 bytes per line: 8 = the whole 128-byte line, 4 = its first 64 bytes, 2 = 32 bytes
template <int LANES>
__global__ __launch_bounds__(1024) void walk(const uint4 *__restrict__ buf, unsigned n_lines, unsigned rounds, unsigned seed, unsigned *__restrict__ sink)
{
    const unsigned lane = threadIdx.x & 63u, grp = lane / LANES, l = lane % LANES;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    unsigned h = mix(wave * 0x9E3779B9u + seed);
    for (unsigned r = 0; r < rounds; r++) {
        h = mix(h + r);
        const unsigned line = mix(h + grp * 0x85EBCA6Bu) % n_lines;
        const uint4 v = buf[(size_t)line * 8 + l];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int LANES>
static int run(const uint4 *buf, unsigned n_lines, unsigned *sink, const char *what)
{
    const unsigned rounds = 2048, blocks = 256 * 2 * 8;                 // 16 waves per workgroup, two workgroups per CU, eight waves of them
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(walk<LANES>, dim3(blocks), dim3(1024), 0, 0, buf, n_lines, rounds / 8, 1u, sink);   // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(walk<LANES>, dim3(blocks), dim3(1024), 0, 0, buf, n_lines, rounds, 7u, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double walks = (double)blocks * 16.0 * rounds * (64 / LANES);
    printf("%-28s %8.3f ms  %7.2f G walks/s  %7.2f TB/s requested\n", what, ms, walks / ms / 1e6, walks * LANES * 16.0 / ms / 1e9);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 256;
    const size_t bytes = mib << 20;
    const unsigned n_lines = (unsigned)(bytes / 128);
    uint4 *buf; unsigned *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    printf("region %zu MiB (%u lines)\n", mib, n_lines);
    if (run<8>(buf, n_lines, sink, "128-byte lines (8 lanes)")) return 1;
    if (run<4>(buf, n_lines, sink, "64-byte halves (4 lanes)")) return 1;
    if (run<2>(buf, n_lines, sink, "32-byte quarters (2 lanes)")) return 1;
    return 0;
}
