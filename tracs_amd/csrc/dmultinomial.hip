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

#include <algorithm>
#include <cmath>

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

// Coverage rules of the align stage applied after the posterior filter (tracs/align.py:599-613): sites whose total count
// is below min_cov, or inside the outlier band [cov_lo, cov_hi] (disabled when cov_lo > cov_hi), become fully ambiguous.
struct CovRule { unsigned min_cov; double cov_lo, cov_hi; };

// Single-precision screen of the same decision.  Everything uniform is prepared on the host; a cell whose numerator lies within
// 2^-14 (relative) of threshold * denominator -- far outside f32 rounding (~4e-7 here) -- is "uncertain" and sends its site to
// posterior_mask4, the exact f64 route with the reference's own divide inside its own, narrower band.  So the mask is the
// reference's bit for bit, and on real counts the f64 route runs for a handful of sites per million.
struct FastPost {
    float a[4], a0, expected, up, down;      // alphas (descending), their sum, threshold, 1 + 2^-14, 1 - 2^-14
    unsigned zero_mask;                      // mask of a site without any count (:53-56)
    unsigned keep_bit;                       // keep (threshold > 0 on this route): a cell at or below the threshold with a count keeps its allele
    unsigned min_cov, band_lo, band_span;    // coverage rules on integers: masked when rs < min_cov or band_lo <= rs <= band_lo + band_span
    unsigned always_exact;                   // parameters outside what the screen was derived for: every site takes the f64 route
};

// SHIFT = 4: the sort keys carry the allele as a one-hot nibble (count << 4 | 1 << allele; counts below 2^28), so a passing
// cell ORs its key's low nibble into the mask; SHIFT = 2 (uint32 counts up to 2^30 - 1): count << 2 | allele.
// Every alpha is positive on this route (the host sends anything else to the exact one), so a cell above the threshold always
// has a positive posterior and the bit is  above | (keep_bit & count > 0)  -- condition-mask arithmetic on the scalar unit.
template <int SHIFT>
__device__ __forceinline__ unsigned posterior_mask4_fast(const unsigned (&c)[4], const FastPost &F, bool &uncertain)
{
    const unsigned tot = c[0] + c[1] + c[2] + c[3];
    unsigned k0 = (c[0] << SHIFT) | (SHIFT == 4 ? 1u : 0u), k1 = (c[1] << SHIFT) | (SHIFT == 4 ? 2u : 1u);
    unsigned k2 = (c[2] << SHIFT) | (SHIFT == 4 ? 4u : 2u), k3 = (c[3] << SHIFT) | (SHIFT == 4 ? 8u : 3u);
    cmpx_desc(k0, k1); cmpx_desc(k2, k3); cmpx_desc(k0, k2); cmpx_desc(k1, k3); cmpx_desc(k1, k2);
    const unsigned s0 = k0 >> SHIFT, s1 = k1 >> SHIFT, s2 = k2 >> SHIFT, s3 = k3 >> SHIFT;
    const bool n1 = s1 != s0, n2 = s2 != s1, n3 = s3 != s2;              // rank steps (:59-64)
    // alpha of the next rank along the sorted row: three selects for al2's and al3's candidates, one each to take the step
    const float al1 = n1 ? F.a[1] : F.a[0];
    const float nx1 = n1 ? F.a[2] : F.a[1];                              // the rank after al1's
    const float al2 = n2 ? nx1 : al1;
    const float nx2 = n2 ? (n1 ? F.a[3] : F.a[2]) : nx1;                 // the rank after al2's
    const float al3 = n3 ? nx2 : al2;
    const float lim = F.expected * ((float)tot + F.a0), hi = lim * F.up, lo = lim * F.down;
    const float u0 = (float)s0 + F.a[0], u1 = (float)s1 + al1, u2 = (float)s2 + al2, u3 = (float)s3 + al3;
    const bool a0 = u0 > hi, a1 = u1 > hi, a2 = u2 > hi, a3 = u3 > hi;
    const bool kb = F.keep_bit != 0u;
    const bool b0 = a0 || (kb && s0 > 0u), b1 = a1 || (kb && s1 > 0u), b2 = a2 || (kb && s2 > 0u), b3 = a3 || (kb && s3 > 0u);
    unsigned m;
    if (SHIFT == 4) {
        m = (k0 & (b0 ? 15u : 0u)) | (k1 & (b1 ? 15u : 0u));
        m |= (k2 & (b2 ? 15u : 0u)) | (k3 & (b3 ? 15u : 0u));
    } else {
        m = ((b0 ? 1u : 0u) << (k0 & 3u)) | ((b1 ? 1u : 0u) << (k1 & 3u)) | ((b2 ? 1u : 0u) << (k2 & 3u)) | ((b3 ? 1u : 0u) << (k3 & 3u));
    }
    const bool empty = tot == 0u;
    uncertain = !empty && ((!a0 && !(u0 < lo)) || (!a1 && !(u1 < lo)) || (!a2 && !(u2 < lo)) || (!a3 && !(u3 < lo)));
    return empty ? F.zero_mask : m;
}

// ---- the table route -----------------------------------------------------------------------------------------------------
// For a site with total count tot, the cell with count s and alpha rank r keeps its allele iff
//     post = (s + alpha_r) / (tot + a0) > threshold   (or the `keep` rule applies),
// and post grows with s: for every (tot, r) there is ONE smallest passing count Y_r(tot).  posterior_table_kernel finds it with
// the reference's own arithmetic (the f64 divide and `<=` of src/dmultinomial.hpp:59-82, by bisection over s), for every total
// below POST_TABLE_TOT, and folds in everything else that only depends on the total: the coverage rules (a masked site passes
// every cell: Y = 0) and the site without any count (:53-56 is the same formula with four tied cells).  The streaming kernel then
// decides a site with integer compares only -- sort the four (count << 4 | one-hot allele) keys, one 16-byte LDS read of
// Y_0..Y_3 (stored << 4: key >= Y << 4 iff count >= Y), pick by rank, compare -- exact by construction, no floating point, no
// guard band.  Totals of POST_TABLE_TOT and more (depth in the thousands) take the screen + exact route below.
constexpr int POST_TABLE_TOT = 4096;                        // x 4 ranks x 4 B = 64 KiB of LDS per 1024-thread workgroup

__device__ __forceinline__ bool posterior_cell_bit(unsigned s, unsigned tot, double alpha, double a0, int keep, double expected)
{
    const double num = (double)s + alpha, den = (double)tot + a0;
    const bool above = !((num / den) <= expected);           // :69: `post <= expected` on the divided value
    return above ? (num > 0.0) : (keep && s > 0u && expected > 0.0);
}

__global__ __launch_bounds__(256) void posterior_table_kernel(Alphas A, int keep, double expected, unsigned min_cov, unsigned band_lo,
                                                              unsigned band_span, unsigned *__restrict__ table)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= POST_TABLE_TOT * 4) return;
    const unsigned tot = e >> 2, r = e & 3u;
    const double alpha = r == 0 ? A.a[0] : r == 1 ? A.a[1] : r == 2 ? A.a[2] : A.a[3];
    unsigned y;
    if (tot < min_cov || (tot - band_lo) <= band_span) y = 0;                       // masked: every allele
    else if (posterior_cell_bit(0u, tot, alpha, A.a0, keep, expected)) y = 0;
    else if (!posterior_cell_bit(tot, tot, alpha, A.a0, keep, expected)) y = tot + 1;   // no count up to the total passes
    else {
        unsigned lo = 0, hi = tot;                           // bit(lo) = 0, bit(hi) = 1
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (posterior_cell_bit(mid, tot, alpha, A.a0, keep, expected)) hi = mid; else lo = mid;
        }
        y = hi;
    }
    table[e] = y << 4;
}

// one site: table route below POST_TABLE_TOT, else the single-precision screen with the exact route where it is uncertain
template <bool WIDE>
__device__ __forceinline__ unsigned posterior_code(const unsigned (&c)[4], const uint4 *__restrict__ lds_table, const Alphas &A, const FastPost &F,
                                                   int keep, double expected)
{
    const unsigned tot = c[0] + c[1] + c[2] + c[3];
    if (tot < (unsigned)POST_TABLE_TOT) {
        unsigned k0 = (c[0] << 4) | 1u, k1 = (c[1] << 4) | 2u, k2 = (c[2] << 4) | 4u, k3 = (c[3] << 4) | 8u;
        cmpx_desc(k0, k1); cmpx_desc(k2, k3); cmpx_desc(k0, k2); cmpx_desc(k1, k3); cmpx_desc(k1, k2);
        const uint4 Y = lds_table[tot];
        const bool n1 = (k0 ^ k1) > 15u, n2 = (k1 ^ k2) > 15u, n3 = (k2 ^ k3) > 15u;      // the sorted count steps down (:59-64)
        const unsigned y1 = n1 ? Y.y : Y.x;
        const unsigned x1 = n1 ? Y.z : Y.y;                  // the rank after y1's
        const unsigned y2 = n2 ? x1 : y1;
        const unsigned x2 = n2 ? (n1 ? Y.w : Y.z) : x1;      // the rank after y2's
        const unsigned y3 = n3 ? x2 : y2;
        unsigned m = (k0 & (k0 >= Y.x ? 15u : 0u)) | (k1 & (k1 >= y1 ? 15u : 0u));
        m |= (k2 & (k2 >= y2 ? 15u : 0u)) | (k3 & (k3 >= y3 ? 15u : 0u));
        return m;
    }
    bool unc;
    unsigned m = posterior_mask4_fast<WIDE ? 2 : 4>(c, F, unc);
    if (unc || F.always_exact) m = posterior_mask4(c, A, keep, expected);      // the exact f64 route
    const bool masked = tot < F.min_cov || (tot - F.band_lo) <= F.band_span;   // unsigned: band_lo <= tot <= band_lo + band_span
    return masked ? 15u : m;
}

// uint16 counts (A, C, G, T; 8 B per site) -> 4-bit allele masks, two sites per byte: 8.5 B per site, HBM-bound by design.
// Main kernel: whole rounds only, no bounds checks anywhere.  A wave takes 512 consecutive sites per round: four fully
// coalesced 16-byte loads per lane (lane l: site pairs l, 64 + l, 128 + l, 192 + l of the round), one byte stored per pair
// (64 contiguous bytes per wave and store).  WIDE = true: uint32 counts (depth above 65535: deep amplicon / viral data),
// 16 B per site, a pair = two loads.  The last partial round goes to posterior_codes_tail_kernel.
template <bool WIDE>
__global__ __launch_bounds__(1024) void posterior_codes_kernel(const void *__restrict__ counts_, size_t rounds, const uint4 *__restrict__ table,
                                                               Alphas A, FastPost F, int keep, double expected, uint8_t *__restrict__ codes)
{
    __shared__ uint4 lds_table[POST_TABLE_TOT];
    for (int e = threadIdx.x; e < POST_TABLE_TOT; e += blockDim.x) lds_table[e] = table[e];
    __syncthreads();
    constexpr int PAIRS = 4;                                 // site pairs per lane and round
    const size_t lane = threadIdx.x & 63, wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave_global; r < rounds; r += nwaves) {
        const size_t base = r * (64 * PAIRS) + lane;
        unsigned rows[PAIRS][2][4];
#pragma unroll
        for (int k = 0; k < PAIRS; k++) {
            const size_t t = base + (size_t)k * 64;
            if (WIDE) {
                const uint4 *c = reinterpret_cast<const uint4 *>(counts_);
                const uint4 v0 = c[2 * t], v1 = c[2 * t + 1];
                rows[k][0][0] = v0.x; rows[k][0][1] = v0.y; rows[k][0][2] = v0.z; rows[k][0][3] = v0.w;
                rows[k][1][0] = v1.x; rows[k][1][1] = v1.y; rows[k][1][2] = v1.z; rows[k][1][3] = v1.w;
            } else {
                const uint4 v = reinterpret_cast<const uint4 *>(counts_)[t];
                rows[k][0][0] = v.x & 0xFFFFu; rows[k][0][1] = v.x >> 16; rows[k][0][2] = v.y & 0xFFFFu; rows[k][0][3] = v.y >> 16;
                rows[k][1][0] = v.z & 0xFFFFu; rows[k][1][1] = v.z >> 16; rows[k][1][2] = v.w & 0xFFFFu; rows[k][1][3] = v.w >> 16;
            }
        }
#pragma unroll
        for (int k = 0; k < PAIRS; k++)
            codes[base + (size_t)k * 64] = (uint8_t)(posterior_code<WIDE>(rows[k][0], lds_table, A, F, keep, expected) |
                                                     (posterior_code<WIDE>(rows[k][1], lds_table, A, F, keep, expected) << 4));
    }
}

// the sites behind the last whole round: one thread = one pair, every access guarded (the table is read from global memory)
template <bool WIDE>
__global__ __launch_bounds__(256) void posterior_codes_tail_kernel(const void *__restrict__ counts_, size_t first_pair, size_t L,
                                                                   const uint4 *__restrict__ table, Alphas A, FastPost F, int keep,
                                                                   double expected, uint8_t *__restrict__ codes)
{
    const size_t t = first_pair + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t >= L) return;
    unsigned out = 0;
    for (int s = 0; s < 2; s++) {
        const size_t site = 2 * t + s;
        if (site >= L) break;
        unsigned c[4];
        if (WIDE) { const uint4 v = reinterpret_cast<const uint4 *>(counts_)[site]; c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
        else { const uint2 v = reinterpret_cast<const uint2 *>(counts_)[site]; c[0] = v.x & 0xFFFFu; c[1] = v.x >> 16; c[2] = v.y & 0xFFFFu; c[3] = v.y >> 16; }
        out |= posterior_code<WIDE>(c, table, A, F, keep, expected) << (4 * s);
    }
    codes[t] = (uint8_t)out;
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
    // the single-precision screen's uniform parameters (posterior_mask4_fast)
    FastPost F;
    bool plain = std::isfinite(threshold) && threshold > 0.0 && threshold < 1e30;
    for (int j = 0; j < 4; j++) {
        F.a[j] = (float)A.a[j];
        plain = plain && std::isfinite(A.a[j]) && A.a[j] > 1e-30 && A.a[j] < 1e30;      // positive: see posterior_mask4_fast
    }
    F.a0 = (float)A.a0; F.expected = (float)threshold;
    F.up = 1.0f + 1.0f / 16384.0f; F.down = 1.0f - 1.0f / 16384.0f;
    F.zero_mask = (!(A.a_min <= threshold) && A.a_min > 0.0) ? 15u : 0u;
    F.keep_bit = (keep && threshold > 0.0) ? 1u : 0u;
    F.always_exact = plain ? 0u : 1u;
    // rs >= cov_lo && rs <= cov_hi on the integer total (the band is disabled when cov_lo > cov_hi)
    F.min_cov = min_cov;
    // the band is disabled when cov_lo > cov_hi: band_lo = 0xFFFFFFFF with span 0 only matches a total no uint16 row reaches,
    // and a uint32 row's total of 2^32 - 1 is beyond COUNT_MAX_WIDE
    F.band_lo = 0xFFFFFFFFu; F.band_span = 0u;
    if (cov_lo <= cov_hi && cov_hi >= 0.0) {
        const double lo = std::ceil(std::max(cov_lo, 0.0)), hi = std::min(std::floor(cov_hi), 4294967294.0);
        if (lo <= hi) { F.band_lo = (unsigned)lo; F.band_span = (unsigned)(hi - lo); }
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    uint4 *table = nullptr;
    if ((rc = workspace_get(24, POST_TABLE_TOT * sizeof(uint4), reinterpret_cast<void **>(&table)))) return rc;
    hipLaunchKernelGGL(posterior_table_kernel, dim3(POST_TABLE_TOT * 4 / 256), dim3(256), 0, stream, A, keep, threshold, F.min_cov, F.band_lo,
                       F.band_span, reinterpret_cast<unsigned *>(table));
    const size_t rounds = (L / 2) / 256;                     // whole rounds of 256 site pairs
    if (rounds) {
        const unsigned blocks = (unsigned)std::min<size_t>((rounds + 15) / 16, 256 * 4);
        if (wide) hipLaunchKernelGGL(posterior_codes_kernel<true>, dim3(blocks), dim3(1024), 0, stream, counts, rounds, table, A, F, keep, threshold, codes);
        else hipLaunchKernelGGL(posterior_codes_kernel<false>, dim3(blocks), dim3(1024), 0, stream, counts, rounds, table, A, F, keep, threshold, codes);
    }
    const size_t first_pair = rounds * 256, tail_pairs = (L + 1) / 2 - first_pair;
    if (tail_pairs) {
        const unsigned blocks = (unsigned)((tail_pairs + 255) / 256);
        if (wide) hipLaunchKernelGGL(posterior_codes_tail_kernel<true>, dim3(blocks), dim3(256), 0, stream, counts, first_pair, L, table, A, F, keep, threshold, codes);
        else hipLaunchKernelGGL(posterior_codes_tail_kernel<false>, dim3(blocks), dim3(256), 0, stream, counts, first_pair, L, table, A, F, keep, threshold, codes);
    }
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
