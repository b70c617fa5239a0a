// dmultinomial.hip -- per-site Dirichlet-multinomial posterior filter for gfx950 (f64, HBM-bound).
//
// Reference behaviour restated (never copied): /root/reference/src/dmultinomial.hpp:8-86
//   alphas sorted descending (:13), a0 = sum (:14), a_min = alphas[0]/a0 (:15);
//   per row: denom = sum of counts in column order (:38-42); stable descending argsort (:45-47);
//   denom <= 0 -> every cell a_min (:53-56); else cell idx[j] = (c + alpha[rank]) / (denom + a0),
//   the rank advancing only when the next sorted count differs (:59-64);
//   then cells <= threshold become `threshold` if keep && count > 0 else 0 (:69-82).
// One thread per site; a row is 32 B in / 32 B out for K = 4, read and written as 2 x 16 B.
#include "common.h"

namespace tracs {

constexpr int KMAX = 8;

struct Alphas { double a[KMAX]; double a0, a_min; };

// The reference walks the stably sorted row and advances the alpha rank whenever the next sorted
// count differs (:59-64).  Equal counts are contiguous after the sort, so the rank a cell gets is
// the number of DISTINCT count values strictly greater than its own -- computed here directly,
// with static indices only (runtime-indexed per-thread arrays would live in scratch memory).
template <int K>
__device__ __forceinline__ void posterior_row(const double (&row)[K], const Alphas &A, int keep, double expected,
                                              double (&res)[K])
{
    double denom = 0;
#pragma unroll
    for (int j = 0; j < K; j++) denom += row[j];
    if (denom <= 0) {
#pragma unroll
        for (int j = 0; j < K; j++) res[j] = A.a_min;
    } else {
        const double den = denom + A.a0;
        bool first[K];       // first occurrence of its value
#pragma unroll
        for (int i = 0; i < K; i++) {
            bool f = true;
#pragma unroll
            for (int h = 0; h < i; h++) f = f && (row[h] != row[i]);
            first[i] = f;
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            int rank = 0;
#pragma unroll
            for (int i = 0; i < K; i++) rank += (first[i] && row[i] > row[j]) ? 1 : 0;
            double al = A.a[0];
#pragma unroll
            for (int r = 1; r < K; r++) al = rank == r ? A.a[r] : al;
            res[j] = (row[j] + al) / den;
        }
    }
#pragma unroll
    for (int j = 0; j < K; j++)
        if (res[j] <= expected) res[j] = (keep && (row[j] > 0)) ? expected : 0.0;
}

// K = 4 fast path: double2 x 2 per row
__global__ __launch_bounds__(256) void posteriors4_kernel(const double2 *__restrict__ counts, size_t L, Alphas A, int keep,
                                                          double expected, double2 *__restrict__ post)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (size_t)gridDim.x * blockDim.x) {
        const double2 lo = counts[2 * i], hi = counts[2 * i + 1];
        const double row[4] = {lo.x, lo.y, hi.x, hi.y};
        double res[4];
        posterior_row<4>(row, A, keep, expected, res);
        post[2 * i] = make_double2(res[0], res[1]);
        post[2 * i + 1] = make_double2(res[2], res[3]);
    }
}

template <int K>
__global__ __launch_bounds__(256) void posteriorsK_kernel(const double *__restrict__ counts, size_t L, Alphas A, int keep,
                                                          double expected, double *__restrict__ post)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (size_t)gridDim.x * blockDim.x) {
        double row[K], res[K];
#pragma unroll
        for (int j = 0; j < K; j++) row[j] = counts[i * K + j];
        posterior_row<K>(row, A, keep, expected, res);
#pragma unroll
        for (int j = 0; j < K; j++) post[i * K + j] = res[j];
    }
}

// production form: uint16 counts (A,C,G,T) -> 4-bit allele mask, two sites per byte.
// Only "is the thresholded posterior > 0" is needed, so the divide of :59 is replaced by a comparison of the
// numerator with threshold * denominator; the exact quotient is formed only inside a 4-ulp guard band around the
// threshold, which keeps the decision identical to the reference's `post <= threshold` on the divided value.
__device__ __forceinline__ void cmpx_desc(unsigned &a, unsigned &b)
{
    const unsigned hi = max(a, b), lo = min(a, b);
    a = hi; b = lo;
}

// Integer route to the ranks: sort the four (count << 2 | allele) keys descending with a 5-exchange network; equal counts
// are adjacent, so the alpha rank (number of distinct larger counts, see posterior_row) grows by one exactly where the
// sorted count changes (:59-64).  Only the final "posterior > threshold" comparison is floating point.
__device__ __forceinline__ unsigned posterior_mask4(const unsigned (&c)[4], const Alphas &A, int keep, double expected)
{
    const unsigned tot = c[0] + c[1] + c[2] + c[3];                      // exact: the f64 sum of :38-42 is exact for integers < 2^53
    if (tot == 0) {                                                      // :53-56: every cell a_min; counts are zero: `keep` cannot apply
        return (!(A.a_min <= expected) && A.a_min > 0.0) ? 15u : 0u;
    }
    unsigned k0 = (c[0] << 2) | 0u, k1 = (c[1] << 2) | 1u, k2 = (c[2] << 2) | 2u, k3 = (c[3] << 2) | 3u;
    cmpx_desc(k0, k1); cmpx_desc(k2, k3); cmpx_desc(k0, k2); cmpx_desc(k1, k3); cmpx_desc(k1, k2);
    const unsigned s0 = k0 >> 2, s1 = k1 >> 2, s2 = k2 >> 2, s3 = k3 >> 2;
    const bool n1 = s1 != s0, n2 = s2 != s1, n3 = s3 != s2;              // rank steps
    const double al0 = A.a[0];
    const double al1 = n1 ? A.a[1] : al0;
    const double al2 = n2 ? (n1 ? A.a[2] : A.a[1]) : al1;
    const double al3 = n3 ? ((n1 && n2) ? A.a[3] : ((n1 || n2) ? A.a[2] : A.a[1])) : al2;
    const double den = (double)tot + A.a0;
    const double lim = expected * den, hi = lim * (1.0 + 1e-15), lo = lim * (1.0 - 1e-15);
    unsigned m = 0;
#define TRACS_CELL(S, AL, K)                                                                                       \
    {                                                                                                              \
        const double num = (double)(S) + (AL);                                                                     \
        bool above;                                   /* post > expected ? */                                      \
        if (num > hi) above = true;                                                                                \
        else if (num < lo) above = false;                                                                          \
        else above = !((num / den) <= expected);      /* guard band: the reference's own arithmetic */             \
        const bool bit = above ? (num > 0.0) : (keep && (S) > 0 && expected > 0.0);                                \
        m |= (bit ? 1u : 0u) << ((K) & 3u);                                                                        \
    }
    TRACS_CELL(s0, al0, k0) TRACS_CELL(s1, al1, k1) TRACS_CELL(s2, al2, k2) TRACS_CELL(s3, al3, k3)
#undef TRACS_CELL
    return m;
}

// one thread = two sites: 16 B in, 1 B out (an 8-site / 4-byte-store form measured 13 % slower: the kernel is bound by
// the f64 compare chain, not by its stores -- bench_aux.py)
// Coverage rules of the align stage applied after the posterior filter (tracs/align.py:599-613): sites whose total count
// is below min_cov, or inside the outlier band [cov_lo, cov_hi] (disabled when cov_lo > cov_hi), become fully ambiguous.
struct CovRule { unsigned min_cov; double cov_lo, cov_hi; };

// WIDE = false: uint16 counts, 16 B per site pair; WIDE = true: uint32 counts (depth above 65535: deep amplicon / viral data)
template <bool WIDE>
__global__ __launch_bounds__(256) void posterior_codes_kernel(const void *__restrict__ counts_, size_t L, Alphas A, int keep,
                                                              double expected, CovRule cov, uint8_t *__restrict__ codes)
{
    const size_t npairs = (L + 1) / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += (size_t)gridDim.x * blockDim.x) {
        unsigned rows[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if (WIDE) {
            const uint4 *c = reinterpret_cast<const uint4 *>(counts_);
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (2 * t + s < L) { const uint4 v = c[2 * t + s]; rows[s][0] = v.x; rows[s][1] = v.y; rows[s][2] = v.z; rows[s][3] = v.w; }
        } else {
            uint4 v;
            if (2 * t + 1 < L) v = reinterpret_cast<const uint4 *>(counts_)[t];
            else {   // odd tail: only 8 valid bytes
                const uint2 h = reinterpret_cast<const uint2 *>(counts_)[2 * t];
                v = make_uint4(h.x, h.y, 0u, 0u);
            }
            rows[0][0] = v.x & 0xFFFFu; rows[0][1] = v.x >> 16; rows[0][2] = v.y & 0xFFFFu; rows[0][3] = v.y >> 16;
            rows[1][0] = v.z & 0xFFFFu; rows[1][1] = v.z >> 16; rows[1][2] = v.w & 0xFFFFu; rows[1][3] = v.w >> 16;
        }
        unsigned out = 0;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (2 * t + s < L) {
                const unsigned rs = rows[s][0] + rows[s][1] + rows[s][2] + rows[s][3];
                const bool masked = rs < cov.min_cov || ((double)rs >= cov.cov_lo && (double)rs <= cov.cov_hi);
                out |= (masked ? 15u : posterior_mask4(rows[s], A, keep, expected)) << (4 * s);
            }
        }
        codes[t] = (uint8_t)out;
    }
}

// ---- align stage, before the posterior filter (tracs/align.py:473-516) ------------------------------------------------
// One pass over the f64 counts the pileup parser produced: narrow to uint16 (what posterior_codes_kernel reads), and
// histogram the per-site coverage rs = sum of the four counts -- every statistic the stage needs (fraction covered,
// fraction >= min_cov, np.median / np.quantile of the non-zero coverages) is an order statistic of that histogram.
// A count that is not an integer in [0, 65535] raises the `bad` flag (the host then refuses: no silent narrowing).
constexpr int COV_LDS_BINS = 8192;
constexpr double COUNT_MAX_WIDE = 1073741823.0;      // 2^30 - 1: posterior_mask4 sorts (count << 2 | allele) keys
template <bool WIDE>
__global__ __launch_bounds__(256) void coverage_profile_kernel(const double *__restrict__ counts, size_t L, unsigned nbins,
                                                               unsigned long long *__restrict__ hist,
                                                               void *__restrict__ counts_out, unsigned *__restrict__ bad)
{
    __shared__ unsigned local[COV_LDS_BINS];
    for (int b = threadIdx.x; b < COV_LDS_BINS; b += blockDim.x) local[b] = 0;
    __syncthreads();
    bool any_bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (size_t)gridDim.x * blockDim.x) {
        const double2 lo = reinterpret_cast<const double2 *>(counts)[2 * i], hi = reinterpret_cast<const double2 *>(counts)[2 * i + 1];
        const double c[4] = {lo.x, lo.y, hi.x, hi.y};
        unsigned v[4], rs = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = c[k] >= 0.0 && c[k] <= (WIDE ? COUNT_MAX_WIDE : 65535.0) && c[k] == (double)(unsigned)c[k];
            any_bad |= !ok;
            v[k] = ok ? (unsigned)c[k] : 0u;
            rs += v[k];
        }
        if (counts_out) {
            if (WIDE) reinterpret_cast<uint4 *>(counts_out)[i] = make_uint4(v[0], v[1], v[2], v[3]);
            else reinterpret_cast<uint2 *>(counts_out)[i] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        }
        if (rs < (unsigned)COV_LDS_BINS) atomicAdd(&local[rs], 1u);
        else atomicAdd(&hist[min(rs, nbins - 1)], 1ull);     // the last bin collects every coverage >= nbins - 1
    }
    __syncthreads();
    for (int b = threadIdx.x; b < COV_LDS_BINS; b += blockDim.x)
        if (local[b]) atomicAdd(&hist[min((unsigned)b, nbins - 1)], (unsigned long long)local[b]);
    if (any_bad) atomicOr(bad, 1u);
}

// --consensus (tracs/align.py:482-493): the first allele with the largest count, or every allele (N) below min_cov.
template <bool WIDE>
__global__ __launch_bounds__(256) void consensus_codes_kernel(const void *__restrict__ counts_, size_t L, unsigned min_cov,
                                                              uint8_t *__restrict__ codes)
{
    const size_t npairs = (L + 1) / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += (size_t)gridDim.x * blockDim.x) {
        unsigned out = 0;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (2 * t + s >= L) continue;
            unsigned c[4];
            if (WIDE) { const uint4 w = reinterpret_cast<const uint4 *>(counts_)[2 * t + s]; c[0] = w.x; c[1] = w.y; c[2] = w.z; c[3] = w.w; }
            else { const uint2 w = reinterpret_cast<const uint2 *>(counts_)[2 * t + s]; c[0] = w.x & 0xFFFFu; c[1] = w.x >> 16; c[2] = w.y & 0xFFFFu; c[3] = w.y >> 16; }
            unsigned best = 0;
#pragma unroll
            for (int k = 1; k < 4; k++) best = c[k] > c[best] ? k : best;           // np.argmax: first maximum
            const unsigned rs = c[0] + c[1] + c[2] + c[3];
            out |= (rs < min_cov ? 15u : (1u << best)) << (4 * s);
        }
        codes[t] = (uint8_t)out;
    }
}

// 4-bit allele mask -> IUPAC letter exactly as tracs/align.py:285-323 maps np.packbits(..., bitorder="little"):
// 0 -> 'X' (no allele survived), 15 -> 'N'.
__global__ void codes_to_iupac_kernel(const uint8_t *__restrict__ codes, size_t L, uint8_t *__restrict__ ascii)
{
    //                          0    1    2    3    4    5    6    7    8    9    10   11   12   13   14   15
    const char lut[17] = "XACMGRSVTWYHKDBN";
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (size_t)gridDim.x * blockDim.x)
        ascii[i] = (uint8_t)lut[(codes[i >> 1] >> (4 * (i & 1))) & 15u];
}

static int make_alphas(const double *alphas, size_t K, Alphas &A)
{
    if (K < 1 || K > KMAX) { set_error("calculate_posteriors: 1 <= K <= 8 alleles supported"); return TRACS_E_ARG; }
    for (size_t j = 0; j < KMAX; j++) A.a[j] = 0.0;
    for (size_t j = 0; j < K; j++) A.a[j] = alphas[j];
    for (size_t a = 1; a < K; a++) {       // descending (:13)
        const double v = A.a[a];
        size_t b = a;
        while (b > 0 && A.a[b - 1] < v) { A.a[b] = A.a[b - 1]; b--; }
        A.a[b] = v;
    }
    A.a0 = 0.0;
    for (size_t j = 0; j < K; j++) A.a0 += A.a[j];     // :14, same order as std::accumulate
    A.a_min = A.a[0] / A.a0;                           // :15
    return TRACS_OK;
}

}  // namespace tracs

using namespace tracs;

extern "C" {

int tracs_calculate_posteriors_device(const double *counts, size_t L, size_t K, const double *alphas_host, int keep,
                                      double threshold, double *posterior, void *stream_)
{
    if (L == 0) return TRACS_OK;
    if (!counts || !alphas_host || !posterior) { set_error("tracs_calculate_posteriors_device: NULL argument"); return TRACS_E_ARG; }
    Alphas A;
    int rc = make_alphas(alphas_host, K, A);
    if (rc) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned blocks = (unsigned)std::min<size_t>((L + 255) / 256, 256 * 16);
    switch (K) {
    case 4:
        hipLaunchKernelGGL(posteriors4_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const double2 *>(counts), L, A,
                           keep, threshold, reinterpret_cast<double2 *>(posterior));
        break;
#define TRACS_K_CASE(KK) case KK: hipLaunchKernelGGL((posteriorsK_kernel<KK>), dim3(blocks), dim3(256), 0, stream, counts, L, A, keep, threshold, posterior); break;
        TRACS_K_CASE(1) TRACS_K_CASE(2) TRACS_K_CASE(3) TRACS_K_CASE(5) TRACS_K_CASE(6) TRACS_K_CASE(7) TRACS_K_CASE(8)
#undef TRACS_K_CASE
    default: set_error("calculate_posteriors: unsupported K"); return TRACS_E_ARG;
    }
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_codes_to_iupac_device(const uint8_t *codes, size_t L, uint8_t *ascii, void *stream_)
{
    if (L == 0) return TRACS_OK;
    if (!codes || !ascii) { set_error("tracs_codes_to_iupac_device: NULL argument"); return TRACS_E_ARG; }
    hipLaunchKernelGGL(codes_to_iupac_kernel, dim3((unsigned)std::min<size_t>((L + 255) / 256, 256 * 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), codes, L, ascii);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_posterior_codes_cov_device(const uint16_t *counts, size_t L, const double *alphas_host, int keep, double threshold,
                                     uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes, void *stream_);

int tracs_posterior_codes_device(const uint16_t *counts, size_t L, const double *alphas_host, int keep, double threshold,
                                 uint8_t *codes, void *stream_)
{
    return tracs_posterior_codes_cov_device(counts, L, alphas_host, keep, threshold, 0u, 1.0, 0.0, codes, stream_);
}

static int posterior_codes_impl(bool wide, const void *counts, size_t L, const double *alphas_host, int keep, double threshold,
                                uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes, void *stream_)
{
    if (L == 0) return TRACS_OK;
    if (!counts || !alphas_host || !codes) { set_error("tracs_posterior_codes_device: NULL argument"); return TRACS_E_ARG; }
    Alphas A;
    int rc = make_alphas(alphas_host, 4, A);
    if (rc) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t npairs = (L + 1) / 2;
    const unsigned blocks = (unsigned)std::min<size_t>((npairs + 255) / 256, 256 * 16);
    const CovRule cov{min_cov, cov_lo, cov_hi};
    if (wide) hipLaunchKernelGGL(posterior_codes_kernel<true>, dim3(blocks), dim3(256), 0, stream, counts, L, A, keep, threshold, cov, codes);
    else hipLaunchKernelGGL(posterior_codes_kernel<false>, dim3(blocks), dim3(256), 0, stream, counts, L, A, keep, threshold, cov, codes);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_posterior_codes_cov_device(const uint16_t *counts, size_t L, const double *alphas_host, int keep, double threshold,
                                     uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes, void *stream_)
{
    return posterior_codes_impl(false, counts, L, alphas_host, keep, threshold, min_cov, cov_lo, cov_hi, codes, stream_);
}

int tracs_posterior_codes_cov_device32(const uint32_t *counts, size_t L, const double *alphas_host, int keep, double threshold,
                                       uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes, void *stream_)
{
    return posterior_codes_impl(true, counts, L, alphas_host, keep, threshold, min_cov, cov_lo, cov_hi, codes, stream_);
}

static int coverage_profile_impl(bool wide, const double *counts, size_t L, uint64_t *hist, size_t nbins, void *counts_out,
                                 uint32_t *bad, void *stream_)
{
    if (!hist || !bad || nbins < 2 || nbins > 0x7FFFFFFFu || (!counts && L)) { set_error("tracs_coverage_profile_device: bad argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TRACS_HIP_CHECK(hipMemsetAsync(hist, 0, nbins * sizeof(uint64_t), stream));
    TRACS_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(uint32_t), stream));
    if (L == 0) return TRACS_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((L + 255) / 256, 256 * 4);
    if (wide) hipLaunchKernelGGL(coverage_profile_kernel<true>, dim3(blocks), dim3(256), 0, stream, counts, L, (unsigned)nbins,
                                 reinterpret_cast<unsigned long long *>(hist), counts_out, bad);
    else hipLaunchKernelGGL(coverage_profile_kernel<false>, dim3(blocks), dim3(256), 0, stream, counts, L, (unsigned)nbins,
                            reinterpret_cast<unsigned long long *>(hist), counts_out, bad);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_coverage_profile_device(const double *counts, size_t L, uint64_t *hist, size_t nbins, uint16_t *counts16,
                                  uint32_t *bad, void *stream_)
{
    return coverage_profile_impl(false, counts, L, hist, nbins, counts16, bad, stream_);
}

int tracs_coverage_profile_device32(const double *counts, size_t L, uint64_t *hist, size_t nbins, uint32_t *counts32,
                                    uint32_t *bad, void *stream_)
{
    return coverage_profile_impl(true, counts, L, hist, nbins, counts32, bad, stream_);
}

static int consensus_codes_impl(bool wide, const void *counts, size_t L, uint32_t min_cov, uint8_t *codes, void *stream_)
{
    if (L == 0) return TRACS_OK;
    if (!counts || !codes) { set_error("tracs_consensus_codes_device: NULL argument"); return TRACS_E_ARG; }
    const size_t npairs = (L + 1) / 2;
    const dim3 grid((unsigned)std::min<size_t>((npairs + 255) / 256, 256 * 16));
    if (wide) hipLaunchKernelGGL(consensus_codes_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), counts, L, min_cov, codes);
    else hipLaunchKernelGGL(consensus_codes_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream_), counts, L, min_cov, codes);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_consensus_codes_device(const uint16_t *counts16, size_t L, uint32_t min_cov, uint8_t *codes, void *stream_)
{
    return consensus_codes_impl(false, counts16, L, min_cov, codes, stream_);
}

int tracs_consensus_codes_device32(const uint32_t *counts32, size_t L, uint32_t min_cov, uint8_t *codes, void *stream_)
{
    return consensus_codes_impl(true, counts32, L, min_cov, codes, stream_);
}

int tracs_calculate_posteriors(const double *counts, size_t L, size_t K, const double *alphas, int keep, double threshold,
                               double *posterior)
{
    if (L == 0) return TRACS_OK;
    if (!counts || !alphas || !posterior) { set_error("tracs_calculate_posteriors: NULL argument"); return TRACS_E_ARG; }
    double *dC = nullptr, *dP = nullptr;
    const size_t bytes = L * K * sizeof(double);
    auto cleanup = [&]() { if (dC) (void)hipFree(dC); if (dP) (void)hipFree(dP); };
#define CP_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
    CP_CHECK(hipMalloc(reinterpret_cast<void **>(&dC), bytes));
    CP_CHECK(hipMalloc(reinterpret_cast<void **>(&dP), bytes));
    CP_CHECK(hipMemcpy(dC, counts, bytes, hipMemcpyHostToDevice));
    int rc = tracs_calculate_posteriors_device(dC, L, K, alphas, keep, threshold, dP, nullptr);
    if (rc) { cleanup(); return rc; }
    CP_CHECK(hipMemcpy(posterior, dP, bytes, hipMemcpyDeviceToHost));
#undef CP_CHECK
    cleanup();
    return TRACS_OK;
}

}  // extern "C"
