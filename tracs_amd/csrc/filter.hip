// filter.hip -- recombination filter on emitted pairs (gfx950).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   filter_recomb :251-318, range_count :223-248, cached_binomial_cdf :41-58
// For an emitted pair the reference flips the match set into the SNP set, and for every SNP site i counts
// the SNPs inside [i-w, i+w+1) (w = clamp(int(1/p/2+1), 50, 5000), p = d/L) and the span first..last of
// those; the SNP survives iff 1 - BinomCDF(count; n = span, p) >= 0.05/d (or it is alone in its window).
//
// Two kernels, one wave per pair:
//   filter_extract_kernel  re-derives the pair's SNP bits from the packed planes (L/32 words per pair,
//                          HBM-bound; sample-minor layout => 16 B segments) and writes the sorted site list;
//   filter_test_kernel     lanes stride over the SNPs: two binary searches give count and span, the
//                          binomial tail is summed directly (integer a, b: I_p(k+1, n-k) is a finite sum).
// PARITY UNPINNED: Boost's ibetac is replaced by the exact finite sum.
#include "common.h"
#include "filter_math.h"

#include <algorithm>
#include <cstdlib>

namespace tracs {

__global__ __launch_bounds__(64) void filter_extract_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned L,
                                                            const unsigned *__restrict__ rows,
                                                            const unsigned *__restrict__ cols, size_t n_pairs,
                                                            const long long *__restrict__ pos_off,
                                                            unsigned *__restrict__ positions, unsigned *__restrict__ found)
{
    const int lane = threadIdx.x;
    const unsigned W = (L + 31) / 32;
    for (size_t t = blockIdx.x; t < n_pairs; t += gridDim.x) {
        const size_t si = rows[t], sj = cols[t];
        long long o = pos_off[t];
        const long long cap = pos_off[t + 1];
        unsigned total = 0;
        for (unsigned base = 0; base < W; base += 64) {
            const unsigned w = base + lane;
            unsigned snp = 0;
            if (w < W) {
                const size_t g = w >> 2;
                const unsigned comp = w & 3;
                const unsigned *pi = reinterpret_cast<const unsigned *>(P + g * NPLANES * n_pad + si) + comp;
                const unsigned *pj = reinterpret_cast<const unsigned *>(P + g * NPLANES * n_pad + sj) + comp;
                const size_t ps = n_pad * 4;       // plane stride in dwords
                const unsigned m = (pi[0] & pj[0]) | (pi[ps] & pj[ps]) | (pi[2 * ps] & pj[2 * ps]) | (pi[3 * ps] & pj[3 * ps]);
                snp = ~m;                                                  // res.flip(), :254
                const unsigned rem = L - w * 32;                           // only the L real bits are flipped
                if (rem < 32) snp &= (1u << rem) - 1u;
            }
            // exclusive prefix of popcounts across the wave
            unsigned c = __popc(snp), incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            long long dst = o + (incl - c);
            while (snp) {
                const int b = __ffs(snp) - 1;
                snp &= snp - 1;
                if (dst < cap) positions[dst] = w * 32 + b;
                dst++;
            }
            const unsigned chunk = __shfl(incl, 63, 64);
            o += chunk;
            total += chunk;
        }
        if (lane == 0) found[t] = total;
    }
}

// Large emitted sets: one PAIR per lane, every lane walks the whole alignment.  The COO list is row-major, so the 64 pairs of
// a wave mostly share their row (one broadcast 16 B load per plane) and have consecutive columns (64 x 16 B contiguous): full
// sectors instead of the per-pair kernel's isolated 16 B segments.  Positions come out in site order per pair by construction.
__global__ __launch_bounds__(256) void filter_extract_lanes_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned L,
                                                                  unsigned groups, const unsigned *__restrict__ rows,
                                                                  const unsigned *__restrict__ cols, size_t n_pairs,
                                                                  const long long *__restrict__ pos_off,
                                                                  unsigned *__restrict__ positions, unsigned *__restrict__ found)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = t < n_pairs;
    const size_t si = act ? rows[t] : 0, sj = act ? cols[t] : 0;
    long long o = act ? pos_off[t] : 0;
    const long long start = o, cap = act ? pos_off[t + 1] : 0;
    auto emit = [&](unsigned snp, unsigned w) {
        while (snp) {
            const int b = __ffs(snp) - 1;
            snp &= snp - 1;
            if (o < cap) positions[o] = w * 32 + b;
            o++;
        }
    };
#pragma unroll 2
    for (unsigned g = 0; g < groups; g++) {
        const uint4 *base = P + (size_t)g * NPLANES * n_pad;
        const uint4 a0 = base[si], a1 = base[n_pad + si], a2 = base[2 * n_pad + si], a3 = base[3 * n_pad + si];
        const uint4 b0 = base[sj], b1 = base[n_pad + sj], b2 = base[2 * n_pad + sj], b3 = base[3 * n_pad + sj];
        uint4 snp;                                                            // res.flip(), src/pairsnp.hpp:254
        snp.x = ~((a0.x & b0.x) | (a1.x & b1.x) | (a2.x & b2.x) | (a3.x & b3.x));
        snp.y = ~((a0.y & b0.y) | (a1.y & b1.y) | (a2.y & b2.y) | (a3.y & b3.y));
        snp.z = ~((a0.z & b0.z) | (a1.z & b1.z) | (a2.z & b2.z) | (a3.z & b3.z));
        snp.w = ~((a0.w & b0.w) | (a1.w & b1.w) | (a2.w & b2.w) | (a3.w & b3.w));
        if (g + 1 == groups) {                                                // only the L real bits are flipped
            unsigned *sw = &snp.x;
            for (int c = 0; c < 4; c++) {
                const unsigned long long first = ((unsigned long long)g * 4 + c) * 32;
                if (first >= L) sw[c] = 0;
                else if (L - first < 32) sw[c] &= (1u << (unsigned)(L - first)) - 1u;
            }
        }
        if (!act) continue;
        if (snp.x | snp.y | snp.z | snp.w) {
            emit(snp.x, g * 4 + 0); emit(snp.y, g * 4 + 1); emit(snp.z, g * 4 + 2); emit(snp.w, g * 4 + 3);
        }
    }
    if (act) found[t] = (unsigned)(o - start);
}

__device__ __forceinline__ long long lower_bound_u32(const unsigned *a, long long n, long long key)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// tbl (may be NULL): per d the smallest surviving span for every count < FLT_KT (filter_lists.hip builds the rows; tbl_state[d] == 1
// where a row exists); without a row the tail is summed per SNP
__global__ __launch_bounds__(64) void filter_test_kernel(const unsigned *__restrict__ positions,
                                                         const long long *__restrict__ pos_off, size_t n_pairs, unsigned L,
                                                         const double *__restrict__ lg, unsigned *__restrict__ filt,
                                                         const unsigned *__restrict__ tbl, const unsigned char *__restrict__ tbl_state)
{
    const int lane = threadIdx.x;
    for (size_t t = blockIdx.x; t < n_pairs; t += gridDim.x) {
        const unsigned *pos = positions + pos_off[t];
        const long long dn = pos_off[t + 1] - pos_off[t];
        if (dn <= 1) { if (lane == 0) filt[t] = (unsigned)dn; continue; }      // :259-261
        const int aln = (int)L;
        const FilterWindow fw = filter_window(dn, L);                          // :265-271
        const double p = fw.p, thr = fw.thr;
        const int wh = fw.wh;
        const unsigned *row = (tbl && dn <= (long long)FLT_DCAP && tbl_state[dn] == 1) ? tbl + (size_t)dn * FLT_KT : nullptr;
        unsigned kept = 0;
        for (long long u = lane; u < dn; u += 64) {
            const int i = (int)pos[u];
            const long long left = max(0, i - wh);                             // :284
            const long long right = min((long long)aln, (long long)i + wh + 1);   // :285
            const long long first = lower_bound_u32(pos, dn, left);
            const long long last = lower_bound_u32(pos, dn, right) - 1;
            const long long count = last - first + 1;
            if (count > 1) {                                                   // :294
                const long long length = (long long)pos[last] - (long long)pos[first] + 1;   // :242
                if (row && count < (long long)FLT_KT) kept += (unsigned long long)length >= row[count] ? 1u : 0u;
                else if (filter_keep(length, count, p, thr, lg)) kept++;       // :294-309
            } else {
                kept++;                                                        // :311
            }
        }
        for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
        if (lane == 0) filt[t] = kept;
    }
}

int get_lgamma_table_for_filter(hipStream_t stream, const double **out);   // transcluster.hip

// extract + test for the pairs of one batch (pos_off: exclusive scan of their distances)
static int filter_scan_launch(const tracs_alignment *a, const unsigned *rows, const unsigned *cols, size_t n_pairs, const long long *pos_off,
                              unsigned *positions, unsigned *found, unsigned *filt, const unsigned *tbl, const unsigned char *tbl_state,
                              const double *lg, hipStream_t stream)
{
    const unsigned blocks = (unsigned)std::min<size_t>(n_pairs, 256 * 64);
    // a pair per lane needs enough pairs to fill the chip (every lane walks all groups); below that, a wave per pair
    size_t lanes_min = 16384;
    if (const char *e = std::getenv("TRACS_FILTER_LANES_MIN")) lanes_min = (size_t)std::strtoull(e, nullptr, 10);
    if (n_pairs >= lanes_min)
        hipLaunchKernelGGL(filter_extract_lanes_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, a->planes,
                           a->n_pad, (unsigned)a->L, (unsigned)a->groups, rows, cols, n_pairs, pos_off, positions, found);
    else
        hipLaunchKernelGGL(filter_extract_kernel, dim3(blocks), dim3(64), 0, stream, a->planes, a->n_pad, (unsigned)a->L, rows, cols,
                           n_pairs, pos_off, positions, found);
    hipLaunchKernelGGL(filter_test_kernel, dim3(blocks), dim3(64), 0, stream, positions, pos_off, n_pairs, (unsigned)a->L, lg, filt, tbl,
                       tbl_state);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// pos_off of a batch on the device: exclusive scan of d[0 .. n) in three launches (tiles of 1 024)
__global__ __launch_bounds__(256) void filter_tile_sums_kernel(const unsigned *__restrict__ d, size_t n, unsigned long long *__restrict__ sums)
{
    __shared__ unsigned long long part[4];
    const size_t base = (size_t)blockIdx.x * 1024;
    unsigned long long s = 0;
    for (int k = 0; k < 4; k++) { const size_t t = base + (size_t)k * 256 + threadIdx.x; if (t < n) s += d[t]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void filter_scan_sums_kernel(unsigned long long *__restrict__ sums, size_t tiles)
{
    __shared__ unsigned long long part[1024];
    const size_t per = (tiles + 1023) / 1024, b = std::min(tiles, (size_t)threadIdx.x * per), e = std::min(tiles, b + per);
    unsigned long long s = 0;
    for (size_t k = b; k < e; k++) s += sums[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (unsigned st = 1; st < 1024; st <<= 1) {
        const unsigned long long v = threadIdx.x >= st ? part[threadIdx.x - st] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;
    for (size_t k = b; k < e; k++) { const unsigned long long v = sums[k]; sums[k] = run; run += v; }
}
__global__ __launch_bounds__(256) void filter_offsets_kernel(const unsigned *__restrict__ d, size_t n, const unsigned long long *__restrict__ sums,
                                                             long long *__restrict__ off)
{
    __shared__ unsigned long long wsum[4];
    const size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
    unsigned v[4];
    unsigned long long s = 0;
    for (int k = 0; k < 4; k++) { v[k] = base + k < n ? d[base + k] : 0u; s += v[k]; }
    unsigned long long incl = s;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long x = __shfl_up(incl, o, 64); if (lane >= o) incl += x; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long run = sums[blockIdx.x] + incl - s;
    for (unsigned w = 0; w < (threadIdx.x >> 6); w++) run += wsum[w];
    // (n + 1 offsets: index n takes the total)
    for (int k = 0; k < 4; k++) { if (base + k <= n) off[base + k] = (long long)run; run += v[k]; }
}
__global__ __launch_bounds__(256) void filter_found_check_kernel(const unsigned *__restrict__ found, const unsigned *__restrict__ d, size_t n,
                                                                 unsigned *__restrict__ bad)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256)
        if (found[t] != d[t]) atomicAdd(bad, 1u);
}

struct ScanWs { enum { POS = 76, FOUND, OFF, SUMS }; };

// The scan of the planes for pairs whose SNP sites do not come from lists (filter_lists.hip): batches whose SNP-site lists fit
// 2^28 entries, offsets scanned on the device, the two extraction kernels above, the window test with the callers' threshold rows.
// *bad (device) counts the pairs whose SNP bits do not add up to d.
int filter_scan_route(const tracs_alignment *a, const unsigned *rows, const unsigned *cols, const unsigned *d, size_t n_pairs, unsigned max_d,
                      unsigned *filt, const unsigned *tbl, const unsigned char *tbl_state, const double *lg, unsigned *bad,
                      hipStream_t stream)
{
    if (!n_pairs) return TRACS_OK;
    size_t kMaxPos = 1ull << 28;
    if (const char *e = std::getenv("TRACS_FILTER_SCAN_MAXPOS")) kMaxPos = std::max<size_t>(1, (size_t)std::strtoull(e, nullptr, 10));   // (tests: several batches)
    const size_t per = std::max<size_t>(1, std::min<size_t>(n_pairs, kMaxPos / std::max<size_t>(max_d, 1)));
    unsigned *pos, *found;
    long long *off;
    unsigned long long *sums;
    int rc;
    if ((rc = workspace_get(ScanWs::POS, (std::min<size_t>(per * std::max<size_t>(max_d, 1), kMaxPos) + 64) * 4, reinterpret_cast<void **>(&pos))) ||
        (rc = workspace_get(ScanWs::FOUND, per * 4, reinterpret_cast<void **>(&found))) ||
        (rc = workspace_get(ScanWs::OFF, (per + 1) * 8, reinterpret_cast<void **>(&off))) ||
        (rc = workspace_get(ScanWs::SUMS, ((per + 1023) / 1024 + 1) * 8, reinterpret_cast<void **>(&sums)))) return rc;
    for (size_t t0 = 0; t0 < n_pairs; t0 += per) {
        const size_t np = std::min(per, n_pairs - t0), tiles = (np + 1023) / 1024;
        // (np + 1 offsets: the tile that holds index np writes off[np]; when np is a multiple of 1 024 that is one tile more)
        const size_t otiles = np / 1024 + 1;
        TRACS_HIP_CHECK(hipMemsetAsync(sums, 0, (otiles + 1) * 8, stream));
        hipLaunchKernelGGL(filter_tile_sums_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, d + t0, np, sums);
        hipLaunchKernelGGL(filter_scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, otiles);
        hipLaunchKernelGGL(filter_offsets_kernel, dim3((unsigned)otiles), dim3(256), 0, stream, d + t0, np, sums, off);
        if ((rc = filter_scan_launch(a, rows + t0, cols + t0, np, off, pos, found, filt + t0, tbl, tbl_state, lg, stream))) return rc;
        hipLaunchKernelGGL(filter_found_check_kernel, dim3((unsigned)std::min<size_t>((np + 255) / 256, 4096)), dim3(256), 0, stream, found,
                           d + t0, np, bad);
        TRACS_HIP_CHECK(hipGetLastError());
    }
    return TRACS_OK;
}

}  // namespace tracs

using namespace tracs;

extern "C" {

// rows/cols/pos_off/positions/found/filt: device.  pos_off = exclusive scan of the pairs' SNP distances
// (n_pairs + 1 entries); positions has pos_off[n_pairs] entries.  found[t] receives the number of SNP
// bits actually seen (must equal the distance; checked by the caller).
int tracs_filter_recomb_device(const tracs_alignment *a, const uint32_t *rows, const uint32_t *cols, size_t n_pairs,
                               const int64_t *pos_off, uint32_t *positions, uint32_t *found, uint32_t *filt, void *stream_)
{
    if (n_pairs == 0) return TRACS_OK;
    if (!a || !rows || !cols || !pos_off || !positions || !found || !filt) { set_error("tracs_filter_recomb_device: NULL argument"); return TRACS_E_ARG; }
    if (a->L >= (1ull << 31)) { set_error("filter: alignment longer than 2^31 sites"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    const double *lg = nullptr;
    int rc = get_lgamma_table_for_filter(stream, &lg);
    if (rc) return rc;
    return filter_scan_launch(a, rows, cols, n_pairs, reinterpret_cast<const long long *>(pos_off), positions, found, filt, nullptr, nullptr, lg,
                              stream);
}

}  // extern "C"
