// filter_math.h -- arithmetic shared by the two routes of the recombination filter (filter.hip: positions from a scan of the
// planes; filter_lists.hip: positions from the samples' departure lists).  Reference behaviour restated (never copied):
// /root/reference/src/pairsnp.hpp filter_recomb :251-318, range_count :223-248, cached_binomial_cdf :41-58.
// PARITY UNPINNED: Boost's ibetac is replaced by the exact finite sum.
#pragma once
#include "common.h"

namespace tracs {

constexpr unsigned FLT_KT = 64;               // threshold table (filter_lists.hip): counts 2 .. 63 per row
constexpr unsigned FLT_DCAP = 65536;          // ... rows for d <= this (beyond: the tail is summed per SNP)

// what filter_recomb derives from a pair's SNP count d and the alignment length (:265-271)
struct FilterWindow {
    double p, thr;
    int wh;
};
__host__ __device__ inline FilterWindow filter_window(long long dn, unsigned L)
{
    FilterWindow w;
    const double d = (double)dn;
    w.p = d / (double)(int)L;                                           // :265
    w.thr = 0.05 / d;                                                   // :266
    int wh = (int)(1.0 / w.p / 2.0 + 1);                                // :269
    wh = wh < 5000 ? wh : 5000;                                         // :270
    wh = wh > 50 ? wh : 50;                                             // :271
    w.wh = wh;
    return w;
}

#ifdef __HIPCC__
// P(X <= k), X ~ Binomial(n, p), 0 <= k < n, summed on the shorter side of the mean in log space.
__device__ inline double binom_cdf(int n, double p, int k, const double *__restrict__ lg)
{
    const double lp = log(p), lq = log1p(-p), ln1 = lg[n + 1];
    const double mean = (double)n * p;
    if ((double)k + 1.0 > mean) {
        // upper tail sum_{j=k+1}^{n}: terms fall off geometrically past the mean
        double term = exp(ln1 - lg[k + 2] - lg[n - k] + (double)(k + 1) * lp + (double)(n - k - 1) * lq);
        double sum = term;
        const double odds = p / (1.0 - p);
        for (int j = k + 1; j < n; j++) {
            term *= (double)(n - j) / (double)(j + 1) * odds;
            sum += term;
            if (term < sum * 1e-18) break;
        }
        return 1.0 - sum;
    }
    double sum = 0.0;
    for (int j = 0; j <= k; j++) sum += exp(ln1 - lg[j + 1] - lg[n - j + 1] + (double)j * lp + (double)(n - j) * lq);
    return sum;
}

// does a SNP whose window holds `count` (> 1) SNPs over `length` sites (first to last) survive?  (:294-309)
__device__ inline bool filter_keep(long long length, long long count, double p, double thr, const double *__restrict__ lg)
{
    const double cdf = count >= length ? 1.0 : binom_cdf((int)length, p, (int)count, lg);
    const double p_value = 1.0 - cdf;                                   // :302
    return p_value >= thr;                                              // :305
}
#endif

}  // namespace tracs
