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

__global__ __launch_bounds__(64) void filter_test_kernel(const unsigned *__restrict__ positions,
                                                         const long long *__restrict__ pos_off, size_t n_pairs, unsigned L,
                                                         const double *__restrict__ lg, unsigned *__restrict__ filt)
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
                if (filter_keep(length, count, p, thr, lg)) kept++;            // :294-309
            } else {
                kept++;                                                        // :311
            }
        }
        for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
        if (lane == 0) filt[t] = kept;
    }
}

int get_lgamma_table_for_filter(hipStream_t stream, const double **out);   // transcluster.hip

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
    const unsigned blocks = (unsigned)std::min<size_t>(n_pairs, 256 * 64);
    // a pair per lane needs enough pairs to fill the chip (every lane walks all groups); below that, a wave per pair
    size_t lanes_min = 16384;
    if (const char *e = std::getenv("TRACS_FILTER_LANES_MIN")) lanes_min = (size_t)std::strtoull(e, nullptr, 10);
    if (n_pairs >= lanes_min)
        hipLaunchKernelGGL(filter_extract_lanes_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, a->planes,
                           a->n_pad, (unsigned)a->L, (unsigned)a->groups, rows, cols, n_pairs,
                           reinterpret_cast<const long long *>(pos_off), positions, found);
    else
        hipLaunchKernelGGL(filter_extract_kernel, dim3(blocks), dim3(64), 0, stream, a->planes, a->n_pad, (unsigned)a->L, rows, cols,
                           n_pairs, reinterpret_cast<const long long *>(pos_off), positions, found);
    hipLaunchKernelGGL(filter_test_kernel, dim3(blocks), dim3(64), 0, stream, positions, reinterpret_cast<const long long *>(pos_off),
                       n_pairs, (unsigned)a->L, lg, filt);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

}  // extern "C"
