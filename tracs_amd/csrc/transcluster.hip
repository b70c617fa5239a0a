// transcluster.hip -- transmission-distance kernels (f64) for gfx950.
//
// Reference behaviour restated (never copied): /root/reference/src/transcluster.hpp
//   logaddexpd :62-75, lprob_k_given_N :90-129, lprob_k_given_N_2 :131-170,
//   upper_bound_E :173-188, expected_k :191-238, trans_dist :240-287
// and tracs/transcluster.py:5,26-39 (date difference in years, exp(p0)).
//
// Structure (DESIGN.md "transcluster kernels"):
//   1. dedup: the per-pair answer is a pure function of the key (N, delta), which the reference
//      memoises in two hash maps (:245-246).  Here a device open-addressing table assigns every
//      element the slot of its key (atomicCAS claims a slot for the first element of a key).
//   2. key kernel: one thread per distinct key evaluates p0 and E(K).  The reference's O(N+k)
//      inner sums per k are replaced by running sums (exact algebra, see tc_eval) so a key costs
//      O(N + k_stop) logaddexp instead of O(k_stop * (N + k_stop)); the stopping test uses the
//      same quantities in the same order, so the truncation point k_stop is the reference's.
//   3. gather: each element copies the result of its slot.
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace tracs {

#ifndef TRACS_TC_TPL
#define TRACS_TC_TPL 4       // E(K) wave loop: consecutive k per lane and step
#endif
#ifndef TRACS_TC_WAVES
#define TRACS_TC_WAVES 3     // ... its register budget, as waves per SIMD
#endif
#ifndef TRACS_TC_FT
#define TRACS_TC_FT 4        // term-ratio loop: consecutive k per lane and step
#endif
#ifndef TRACS_TC_RW
#define TRACS_TC_RW 4        // ... its register budget, as waves per SIMD
#endif
constexpr int LG_TABLE = 32768;     // lgamma(n) table, n < LG_TABLE; beyond: lgamma() inline
constexpr int LK_TABLE = 10240;     // log(k) table behind it (the E(K) loop stops at k = 10 000): lg[LG_TABLE + k] = log(k),
                                    // and 1 / k behind that: lg[LG_TABLE + LK_TABLE + k] = 1 / k (the term-ratio recurrence of tc_eval_wave)

struct TcParams {
    double lamb, beta, thr;

    double ln_lamb, ln_beta, ln_lb;   // log(lamb), log(beta), log(lamb+beta): filled on the device
};

__device__ __forceinline__ double lae(double x, double y)   // logaddexpd, transcluster.hpp:62-75
{
    const double tmp = x - y;
    if (x == y) return x + 0.693147180559945309417232121458176568;
    if (tmp > 0) return x + log1p(exp(-tmp));
    else if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp;
}

// (beyond the table -- SNP distances of 22 000 and more -- and out of line: inlined at every use, lgamma's registers cost the E(K)
// kernels their occupancy: 284 VGPRs against 128)
__device__ __noinline__ double lgamma_beyond(double x) { return lgamma(x); }
__device__ __forceinline__ double lg_at(const double *__restrict__ lg, long long n)
{
    return n < LG_TABLE ? lg[n] : lgamma_beyond((double)n);
}

// i * log(x) as the shipped -ffast-math build evaluates it: the i == 0 term is 0 even when
// log(x) = -inf (delta == 0) -- SURVEY.md 8c; pinned by tests/golden (delta = 0 rows).
__device__ __forceinline__ double imul(long long i, double l) { return i == 0 ? 0.0 : (double)i * l; }

// p0 = lprob_k_given_N_2(N, 0, delta)[0]  (trans_dist :276-282)
// eK = expected_k(N, delta, ...)          (:191-238)
//
// For delta > 0 the reference evaluates, for every k,
//   lhs_k      = (N+1)ln(lamb) + k ln(beta) + lgG(N+k+1) - lgG(N+1) - lgG(k+1) - delta*beta - pois      (:140-149)
//   integral_k = ln sum_{i=0}^{N+k} delta^(N+k-i)/(N+k-i)! (lamb+beta)^-(i+1)                            (:152-158)
// with pois = ln sum_{i<=N} (lamb*delta)^i / i!  independent of k.  Substituting j = N+k-i,
//   integral_k = -(N+k+1) ln(lamb+beta) + ln S_{N+k},   S_M = sum_{j<=M} (delta(lamb+beta))^j / j!
// so S is a running sum in k.  All terms are positive: no cancellation is introduced.
// k_cap: evaluate at most k_cap-1 terms here; returns false if the reference's loop would still be running
// (the key is then finished by tc_long_keys_kernel, one wave per key).
// state (4 doubles, may be NULL): when the cap is hit, the running sums {pois, lnS, lprob, elprob} at k = k_stop, so that
// tc_eval_wave resumes there instead of starting over (the O(N) prefix and the first k_cap terms are not summed twice).
__device__ bool tc_eval(int N, double delta, const TcParams &P, const double *__restrict__ lg, double &p0, double &eK,
                        int &k_stop, int k_cap, double *__restrict__ state = nullptr)
{
    const double n1 = (double)(N + 1);
    const double lg_n1 = lg_at(lg, (long long)N + 1);
    double lprob = -INFINITY, elprob = -INFINITY;
    int k = 1;
    if (delta > 0) {
        const double lx = log(P.lamb * delta);
        double pois = -INFINITY;                                       // :144-148, same fold order
        for (long long i = 0; i <= N; i++) pois = lae(imul(i, lx) - lg_at(lg, i + 1), pois);
        const double upper = exp(P.ln_beta + delta * P.lamb + log(n1) - (P.ln_lamb + pois));   // :185
        const double ld = log(delta);
        double lnS = -INFINITY;
        for (long long j = 0; j <= N; j++) lnS = lae(lnS, imul(j, ld) + (double)j * P.ln_lb - lg_at(lg, j + 1));
        {
            double l0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1);
            l0 = l0 - lg_n1 - lg_at(lg, 1) - delta * P.beta;
            l0 -= pois;
            p0 = l0 + (lnS - n1 * P.ln_lb);
        }
        double diff = P.thr + 1;
        while ((diff > P.thr) && (k < 10000)) {                        // :207
            if (k >= k_cap) {
                k_stop = k;
                if (state) { state[0] = pois; state[1] = lnS; state[2] = lprob; state[3] = elprob; }
                return false;
            }
            const long long M = (long long)N + k;
            lnS = lae(lnS, imul(M, ld) + (double)M * P.ln_lb - lg_at(lg, M + 1));
            double lhs = (n1 * P.ln_lamb + (double)k * P.ln_beta + lg_at(lg, M + 1));   // :140
            lhs = lhs - lg_n1 - lg_at(lg, (long long)k + 1) - delta * P.beta;            // :141
            lhs -= pois;                                                                // :149
            const double m1 = (double)(M + 1) * P.ln_lb;
            const double lk = log((double)k);
            lprob = lae(lprob, (lhs + (lnS - m1)) + lk);                                // :227
            elprob = lae(elprob, lhs + lk + delta * (P.lamb + P.beta) - m1);            // :231
            diff = upper - exp(elprob);                                                 // :232
            k++;
        }
    } else {
        // closed form (:163-167); upper bound with pois = 0 (fast-math build, see imul)
        const double upper = exp(P.ln_beta + log(n1) - P.ln_lamb);
        p0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1 - lg_n1 - lg_at(lg, 1) - n1 * P.ln_lb);
        double diff = P.thr + 1;
        while ((diff > P.thr) && (k < 10000)) {
            if (k >= k_cap) {
                k_stop = k;
                if (state) { state[0] = 0.0; state[1] = -INFINITY; state[2] = lprob; state[3] = elprob; }
                return false;
            }
            const long long M = (long long)N + k;
            const double m1 = (double)(M + 1) * P.ln_lb;
            const double lhs = (n1 * P.ln_lamb + (double)k * P.ln_beta + lg_at(lg, M + 1) - lg_n1 -
                                lg_at(lg, (long long)k + 1) - m1);
            const double lk = log((double)k);
            lprob = lae(lprob, lhs + lk);
            elprob = lae(elprob, lhs + lk + delta * (P.lamb + P.beta) - m1);
            diff = upper - exp(elprob);
            k++;
        }
    }
    eK = exp(lprob);                                                   // :237
    k_stop = k;
    return true;
}

// ---- wave-parallel E(K) for long series -------------------------------------------------------------
// One wave per key, 64 consecutive k per step.  The three running quantities of the loop (S, lprob, elprob)
// are log-space prefix sums, so a step is: per-lane terms, three inclusive wave scans with logaddexp, the
// reference's stopping test on every lane's prefix, and a ballot for the first lane that satisfies it.
// Summation order differs from the serial loop by rounding only (all terms positive).
// Resumes the serial loop of tc_eval at k = k_start from its running sums `state` = {pois, lnS, lprob, elprob}.
// The three running sums (S, lprob's and elprob's) are kept in LINEAR space, each in units of exp(its running maximum): a step
// costs one exp per lane and sum (exp(term - maximum)) and one log per lane for ln S; when a step's largest term exceeds the
// running maximum the sum is rescaled once (wave-uniform).  All terms are positive, the scaled sums stay within [1, 64 x terms),
// and only rounding differs from the reference's log-space fold (:207-232).  The stopping test upper - elprob_linear > thr is
// evaluated in the same units: elprob_scaled < (upper - thr) exp(-maximum), the right-hand side refreshed when the maximum moves.
// Tables over (day gap, M) for sources whose delta is a whole number of days (DenseSource): both O(N) prefix sums of a key and
// the running sum S of its loop depend on (delta, index) only --
//     lnS[gap][M]  = ln sum_{j <= M} (delta (lamb + beta))^j / j!        (M <= largest N + 10 000)
//     pois[gap][N] = ln sum_{i <= N} (lamb delta)^i / i!                  (:144-148)
// -- so they are summed once per distinct gap (tc_tables_kernel, one wave per gap) instead of once per key, and a key's loop
// reads S instead of carrying a third running sum.  Built per call when the table fits TC_TABLE_ELEMS (decided on the device:
// no host round trip); `ok` = 0 otherwise, and for sources without day gaps.
struct TcTables {
    double *lnS, *pois;
    double *lnS_n;                  // linear tables: ln S_N for N <= n_max (rows of ldn), what p0 needs in log space
    unsigned n_max, gap_max;        // bounds of the keys of this call
    unsigned ldm, ldn;              // row lengths: n_max + 10 001, n_max + 1
    unsigned ok;
    unsigned linear;                // lnS[gap][M] holds F = S_M exp(-x), x = delta (lamb + beta): the Poisson(x) distribution function at M,
                                    // in LINEAR space (x <= TC_LINEAR_X_MAX over the call's gaps: F >= exp(-x) stays far from underflow)
};
constexpr double TC_LINEAR_X_MAX = 600.0;
constexpr unsigned long long TC_TABLE_ELEMS = 48ull << 20;      // doubles in the lnS table (384 MB)

// A key handed over without any term summed (state[0] is NaN: tc_keys_kernel does that for N >= TC_WAVE_PREFIX_MIN)
// starts here with the two O(N) prefix sums of the loop -- pois (:144-148) and S_N -- as wave sums of 64 terms per step instead of a
// serial fold, and p0 (the k = 0 value, :276-282) comes back through `p0_out`.
__device__ __forceinline__ void tc_eval_wave(int N, double delta, const TcParams &P, const double *__restrict__ lg, double &eK,
                             const double *__restrict__ state, int k_start, double *p0_out, const TcTables *__restrict__ tab, long long gap)
{
    // this key's rows of the (gap, M) tables, when they exist and hold it
    const bool tabled = tab && tab->ok && gap >= 1 && gap <= (long long)tab->gap_max && (unsigned)N <= tab->n_max;
    const double *__restrict__ rowS = tabled ? tab->lnS + (size_t)gap * tab->ldm : nullptr;
    const int lane = threadIdx.x & 63;
    const double n1 = (double)(N + 1);
    const double lg_n1 = lg_at(lg, (long long)N + 1);
    const double *__restrict__ lkt = lg + LG_TABLE;                    // log(k), k < LK_TABLE
    const bool pos = delta > 0;
    const bool fresh = state[0] != state[0];
    auto wave_max = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
        return v;
    };
    auto wave_prefix = [&](double e) {                                // inclusive prefix sums across the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(e, off, 64);
            if (lane >= off) e += o;
        }
        return e;
    };
    // sum = scaled * exp(mx); an empty sum is (0, -inf)
    double pois, ld = 0.0, upper;
    double Ms, Ss, Mp, Lps, Me, Els;
    const bool linear = tabled && tab->linear;
    if (fresh && tabled) {
        pois = tab->pois[(size_t)gap * tab->ldn + N];
        const double lnS = linear ? tab->lnS_n[(size_t)gap * tab->ldn + N] : rowS[N];
        double l0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1);
        l0 = l0 - lg_n1 - lg_at(lg, 1) - delta * P.beta;
        l0 -= pois;
        *p0_out = l0 + (lnS - n1 * P.ln_lb);
        Ms = -INFINITY; Ss = 0.0;                                      // (unused: S comes from the table)
        Mp = -INFINITY; Lps = 0.0; Me = -INFINITY; Els = 0.0;
        k_start = 1;
    } else if (fresh && !pos) {                                        // delta = 0: closed forms, no prefix (:163-167)
        pois = 0.0;
        *p0_out = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1 - lg_n1 - lg_at(lg, 1) - n1 * P.ln_lb);
        Ms = -INFINITY; Ss = 0.0; Mp = -INFINITY; Lps = 0.0; Me = -INFINITY; Els = 0.0;
        k_start = 1;
    } else if (fresh) {                                                // (delta > 0)
        const double lx = log(P.lamb * delta);
        ld = log(delta);
        // two sweeps over the lane's own terms (i = lane, lane + 64, ..): the largest term first, then sum exp(term - largest) --
        // one exp per term and no cross-lane traffic until the two wave reductions at the end
        auto term_q = [&](long long i) { return imul(i, lx) - lg_at(lg, i + 1); };
        auto term_a = [&](long long i) { return imul(i, ld) + (double)i * P.ln_lb - lg_at(lg, i + 1); };
        double Mq = -INFINITY;
        Ms = -INFINITY;
        for (long long i = lane; i <= N; i += 64) { Mq = fmax(Mq, term_q(i)); Ms = fmax(Ms, term_a(i)); }
        Mq = wave_max(Mq); Ms = wave_max(Ms);
        double Qs = 0.0;
        Ss = 0.0;
        for (long long i = lane; i <= N; i += 64) { Qs += exp(term_q(i) - Mq); Ss += exp(term_a(i) - Ms); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { Qs += __shfl_xor(Qs, off, 64); Ss += __shfl_xor(Ss, off, 64); }
        pois = Mq + log(Qs);
        const double lnS = Ms + log(Ss);
        double l0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1);
        l0 = l0 - lg_n1 - lg_at(lg, 1) - delta * P.beta;
        l0 -= pois;
        *p0_out = l0 + (lnS - n1 * P.ln_lb);
        Mp = -INFINITY; Lps = 0.0; Me = -INFINITY; Els = 0.0;
        k_start = 1;
    } else {
        pois = state[0];
        Ms = state[1]; Ss = Ms == -INFINITY ? 0.0 : 1.0;
        Mp = state[2]; Lps = Mp == -INFINITY ? 0.0 : 1.0;
        Me = state[3]; Els = Me == -INFINITY ? 0.0 : 1.0;
        if (pos) ld = log(delta);
    }
    if (pos) upper = exp(P.ln_beta + delta * P.lamb + log(n1) - (P.ln_lamb + pois));
    else upper = exp(P.ln_beta + log(n1) - P.ln_lamb);
    double lim = Me == -INFINITY ? INFINITY : (upper - P.thr) * exp(-Me);
    // TPL consecutive k per lane and step (lane-major: the terms of a step stay in k order): the TPL x 2 exponentials of a lane are
    // independent, and the two wave scans and two wave maxima of a step are paid once per 64 x TPL terms
    constexpr int TPL = TRACS_TC_TPL;
    // Where the loop starts.  The elprob terms t2(k) = C + k ln(beta / (lamb + beta)) + lgG(N + k + 1) - lgG(k + 1) + ln k grow while
    // k < k* = beta (N + 1) / lamb (t2(k + 1) - t2(k) = ln(beta (N + k + 1) / ((lamb + beta) k))) and fall behind it, and the lprob
    // terms are t1(k) = t2(k) + c(k) with c(k) = ln S_{N+k} - delta (lamb + beta) <= 0 growing with k.  Every term before
    // k_lo <= min(k*, 9999) with t2(k_lo) <= t2(min(k*, 9999)) - 60 is therefore below e^-60 of the largest term of its sum: all of them
    // together change neither sum by an ulp, and they are skipped -- at N = 1 000 (k* = 2 440, the loop stops near 3 000) the first
    // 1 500 terms, at N = 10 000 (k* beyond the loop's 10 000 terms) all but the last ~ 200.  The stopping test cannot fire among the
    // skipped terms when their sum, at most k_lo exp(t2(k_lo)), is below upper - thr: checked.  Needs S from the table (the running sum
    // S cannot skip), so tabled keys with delta > 0 only; k_lo is looked for at 64 probes between the start and the peak.
    const double c2 = n1 * P.ln_lamb - lg_n1 - delta * P.beta - pois + delta * (P.lamb + P.beta);
    auto t2_at = [&](int k) {
        const long long M = (long long)N + k;
        return c2 + (double)k * P.ln_beta + lg_at(lg, M + 1) - lg_at(lg, (long long)k + 1) + lkt[k] - (double)(M + 1) * P.ln_lb;
    };
    const int k_peak = (int)fmin(9999.0, fmax(1.0, floor(P.beta * n1 / P.lamb)));
    if (pos && tabled && upper - P.thr > 0.0) {
        if (k_peak > k_start + 64 * TPL) {
            const double t_peak = t2_at(k_peak);
            const int kj = k_start + (int)(((long long)(k_peak - k_start) * lane) >> 6);
            const double tj = t2_at(kj);
            const unsigned long long low = __ballot(tj <= t_peak - 60.0);
            if (low) {
                const int j = 63 - __clzll((long long)low);
                const int k_lo = __shfl(kj, j, 64);
                const double t_lo = __shfl(tj, j, 64);
                if (log((double)k_lo) + t_lo < log(upper - P.thr)) k_start = k_lo;
            }
        }
    }
    auto accumulate_n = [&](const double (&t)[TPL], double &scaled, double &mx, bool &moved, double (&out)[TPL]) {
        double m = t[0];
#pragma unroll
        for (int q = 1; q < TPL; q++) m = fmax(m, t[q]);
        m = wave_max(m);
        moved = m > mx;
        if (moved) { scaled = mx == -INFINITY ? 0.0 : scaled * exp(mx - m); mx = m; }
        double run = 0.0;
#pragma unroll
        for (int q = 0; q < TPL; q++) { run += t[q] == -INFINITY ? 0.0 : exp(t[q] - mx); out[q] = run; }
        const double incl = wave_prefix(run);
        const double before = __shfl_up(incl, 1, 64);
        const double base = scaled + (lane ? before : 0.0);
#pragma unroll
        for (int q = 0; q < TPL; q++) out[q] += base;
        scaled = __shfl(out[TPL - 1], 63, 64);                         // (every term of the step included)
    };
    for (int k0 = k_start; k0 < 10000; k0 += 64 * TPL) {
        const int kb = k0 + lane * TPL;
        double t1[TPL], t2[TPL], lp[TPL], el[TPL];
        bool live[TPL];
        bool moved;
        if (pos) {
            double Sk[TPL];
            if (tabled) {
#pragma unroll
                for (int q = 0; q < TPL; q++) Sk[q] = kb + q < 10000 ? (linear ? log(rowS[(long long)N + kb + q]) + delta * (P.lamb + P.beta) : rowS[(long long)N + kb + q]) : 0.0;
            } else {
                double a[TPL], Sl[TPL];
#pragma unroll
                for (int q = 0; q < TPL; q++) {
                    const long long M = (long long)N + kb + q;
                    a[q] = kb + q < 10000 ? imul(M, ld) + (double)M * P.ln_lb - lg_at(lg, M + 1) : -INFINITY;
                }
                accumulate_n(a, Ss, Ms, moved, Sl);
#pragma unroll
                for (int q = 0; q < TPL; q++) Sk[q] = Ms + log(Sl[q]);
            }
#pragma unroll
            for (int q = 0; q < TPL; q++) {
                const int k = kb + q;
                live[q] = k < 10000;
                const long long M = (long long)N + k;
                const double m1 = (double)(M + 1) * P.ln_lb;
                const double lk = lkt[live[q] ? k : 0];
                double lhs = (n1 * P.ln_lamb + (double)k * P.ln_beta + lg_at(lg, M + 1));
                lhs = lhs - lg_n1 - lg_at(lg, (long long)k + 1) - delta * P.beta;
                lhs -= pois;
                t1[q] = live[q] ? (lhs + (Sk[q] - m1)) + lk : -INFINITY;
                t2[q] = live[q] ? lhs + lk + delta * (P.lamb + P.beta) - m1 : -INFINITY;
            }
        } else {
#pragma unroll
            for (int q = 0; q < TPL; q++) {
                const int k = kb + q;
                live[q] = k < 10000;
                const long long M = (long long)N + k;
                const double m1 = (double)(M + 1) * P.ln_lb;
                const double lk = lkt[live[q] ? k : 0];
                const double lhs = (n1 * P.ln_lamb + (double)k * P.ln_beta + lg_at(lg, M + 1) - lg_n1 -
                                    lg_at(lg, (long long)k + 1) - m1);
                t1[q] = live[q] ? lhs + lk : -INFINITY;
                t2[q] = live[q] ? lhs + lk + delta * (P.lamb + P.beta) - m1 : -INFINITY;
            }
        }
        accumulate_n(t1, Lps, Mp, moved, lp);
        accumulate_n(t2, Els, Me, moved, el);
        if (moved) lim = (upper - P.thr) * exp(-Me);
        // the while condition (upper - elprob > thr) fails after the first k whose prefix reaches lim (the prefixes grow with k)
        bool stop = false;
        double val = lp[TPL - 1];
#pragma unroll
        for (int q = TPL - 1; q >= 0; q--)
            if (live[q] && !(el[q] < lim)) { stop = true; val = lp[q]; }
        const unsigned long long m = __ballot(stop);
        if (m) {
            const int f = __ffsll((long long)m) - 1;
            eK = __shfl(val, f, 64) * exp(Mp);
            return;
        }
    }
    eK = Lps * exp(Mp);                                                // ran to k = 9999
}

// ---- the term-ratio loop: keys of a dense block with N >= TC_WAVE_PREFIX_MIN, a day gap >= 1 and LINEAR tables ----------------------
// exp(t2(k + 1) - t2(k)) = beta (N + k + 1) / ((lamb + beta) k) is rational in k, and t1(k) = t2(k) + ln F(N + k) with F the table's
// Poisson distribution function in linear space: a lane evaluates ONE exponential per step -- its first term, in units of exp(ref),
// ref = t2 at its peak k* = beta (N + 1) / lamb: every term is <= ~1, nothing overflows, terms 700 below the peak vanish as they do in
// the reference's log-space fold -- and walks its FT consecutive k with two multiplications per term (1 / k from a table), where the
// log-space form took two exponentials per term.  Per step the lanes' two run sums are prefix-summed over the wave and the reference's
// stopping test (upper - exp(elprob) > thr, :207,232) is made on every term's prefix.  Rounding only.  The head of the series is
// skipped as in tc_eval_wave (terms e^-60 below the peak, their sum checked against the stopping bound).  One wave per key; a kernel
// of its own so that its small register budget buys the occupancy that hides the loads' latency (a key is a chain of ~10 dependent
// memory round trips: the log-space kernel ran three waves per SIMD and spent 20 us per key waiting).
// -> false when the key is not eligible (tables absent or in log space, gap 0, N beyond the tables): the caller hands it to the wave kernel.
__device__ __forceinline__ bool tc_eval_ratio(int N, double delta, long long gap, const TcParams &P, const double *__restrict__ lg,
                                              const TcTables *__restrict__ tab, double &eK, double &p0)
{
    // (every lgamma of the loop from the table -- N + 10 000 < LG_TABLE --: no call in this kernel, its register count is the loop's own)
    if (!(tab->ok && tab->linear && gap >= 1 && gap <= (long long)tab->gap_max && (unsigned)N <= tab->n_max && delta > 0 && N + 10001 < LG_TABLE)) return false;
    const double *__restrict__ rowF = tab->lnS + (size_t)gap * tab->ldm;
    const double *__restrict__ lkt = lg + LG_TABLE;
    const double *__restrict__ invk = lg + LG_TABLE + LK_TABLE;
    const int lane = threadIdx.x & 63;
    const double n1 = (double)(N + 1);
    const double lg_n1 = lg[N + 1];
    const double pois = tab->pois[(size_t)gap * tab->ldn + N];
    const double lnS = tab->lnS_n[(size_t)gap * tab->ldn + N];
    {
        double l0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1);
        l0 = l0 - lg_n1 - lg[1] - delta * P.beta;
        l0 -= pois;
        p0 = l0 + (lnS - n1 * P.ln_lb);
    }
    const double upper = exp(P.ln_beta + delta * P.lamb + log(n1) - (P.ln_lamb + pois));
    const double c2 = n1 * P.ln_lamb - lg_n1 - delta * P.beta - pois + delta * (P.lamb + P.beta);
    auto t2_at = [&](int k) {
        const long long M = (long long)N + k;
        return c2 + (double)k * P.ln_beta + lg[M + 1] - lg[k + 1] + lkt[k] - (double)(M + 1) * P.ln_lb;
    };
    constexpr int FT = TRACS_TC_FT;
    const int k_peak = (int)fmin(9999.0, fmax(1.0, floor(P.beta * n1 / P.lamb)));
    const bool bounded = upper - P.thr > 0.0;
    int k_start = 1;
    const double t_peak = t2_at(k_peak);
    if (bounded && k_peak > 1 + 64 * FT) {
        const int kj = 1 + (int)(((long long)(k_peak - 1) * lane) >> 6);
        const double tj = t2_at(kj);
        const unsigned long long low = __ballot(tj <= t_peak - 60.0);
        if (low) {
            const int j = 63 - __clzll((long long)low);
            const int k_lo = __shfl(kj, j, 64);
            const double t_lo = __shfl(tj, j, 64);
            if (log((double)k_lo) + t_lo < log(upper - P.thr)) k_start = k_lo;
        }
    }
    auto wave_prefix = [&](double e) {                                // inclusive prefix sums across the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(e, off, 64);
            if (lane >= off) e += o;
        }
        return e;
    };
    const double ref = bounded ? t_peak : t2_at(1);                   // (a loop that stops at its first term: that term is 1)
    const double flim = (upper - P.thr) * exp(-ref);
    const double qb = P.beta / (P.lamb + P.beta);
    double S1 = 0.0, S2 = 0.0;                                        // sums of the steps before this one (wave-uniform)
    for (int k0 = k_start; k0 < 10000; k0 += 64 * FT) {
        const int kb = k0 + lane * FT;
        double F[FT], rk[FT];
#pragma unroll
        for (int q = 0; q < FT; q++) {                                // (the step's loads first: they do not wait for the exponential)
            const int k = kb + q;
            const bool in = k < 10000;
            F[q] = in ? rowF[(long long)N + k] : 0.0;
            rk[q] = invk[in ? k : 1];
        }
        const double e0 = kb < 10000 ? exp(t2_at(kb) - ref) : 0.0;
        double e = e0, r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int q = 0; q < FT; q++) {
            const int k = kb + q;
            const double ee = k < 10000 ? e : 0.0;
            r2 += ee; r1 += ee * F[q];
            e *= qb * (double)(N + k + 1) * rk[q];
        }
        const double i1 = wave_prefix(r1), i2 = wave_prefix(r2);
        const double b1 = S1 + (i1 - r1), b2 = S2 + (i2 - r2);
        // the lane whose run crosses the bound walks its terms once more for the exact one (once per key)
        const unsigned long long m = __ballot(kb < 10000 && !(b2 + r2 < flim));
        if (m) {
            const int f = __ffsll((long long)m) - 1;
            double val = 0.0;
            if (lane == f) {
                double a1 = b1, a2 = b2;
                e = e0;
                bool found = false;
#pragma unroll
                for (int q = 0; q < FT; q++) {
                    const int k = kb + q;
                    if (k < 10000 && !found) {
                        a2 += e; a1 += e * F[q];
                        val = a1;
                        found = !(a2 < flim);
                    }
                    e *= qb * (double)(N + k + 1) * rk[q];
                }
            }
            eK = __shfl(val, f, 64) * exp(ref);
            return true;
        }
        S1 = __shfl(b1 + r1, 63, 64); S2 = __shfl(b2 + r2, 63, 64);
    }
    eK = S1 * exp(ref);                                               // ran to k = 9999
    return true;
}

// The same for sampling dates a day apart or less (delta = 0: the reference's closed-form branch, :163-167).  t1(k) = (N + 1) ln lamb + k ln
// beta + lgG(N + k + 1) - lgG(N + 1) - lgG(k + 1) - (N + k + 1) ln(lamb + beta) + ln k and t2(k) = t1(k) - (N + k + 1) ln(lamb + beta):
// both ratios are rational (r1 = beta (N + k + 1) / ((lamb + beta) k), r2 = r1 / (lamb + beta)), no table is read, and nothing of the
// head may be skipped (t2 peaks early: its head IS its sum).  Two exponentials per lane and step.  With the default rates the t2 sum is
// e^-4000 of the bound and the loop runs its 9 999 terms, as the reference's does.
__device__ __forceinline__ bool tc_eval_ratio_zero(int N, const TcParams &P, const double *__restrict__ lg, double &eK, double &p0)
{
    if (N + 10001 >= LG_TABLE) return false;
    const double *__restrict__ lkt = lg + LG_TABLE;
    const double *__restrict__ invk = lg + LG_TABLE + LK_TABLE;
    const int lane = threadIdx.x & 63;
    const double n1 = (double)(N + 1);
    const double lg_n1 = lg[N + 1];
    p0 = (n1 * P.ln_lamb + 0.0 * P.ln_beta + lg_n1 - lg_n1 - lg[1] - n1 * P.ln_lb);
    const double upper = exp(P.ln_beta + log(n1) - P.ln_lamb);
    auto t1_at = [&](int k) {
        const long long M = (long long)N + k;
        return n1 * P.ln_lamb + (double)k * P.ln_beta + lg[M + 1] - lg_n1 - lg[k + 1] - (double)(M + 1) * P.ln_lb + lkt[k];
    };
    constexpr int FT = TRACS_TC_FT;
    const double lb = P.lamb + P.beta;
    const int k1 = (int)fmin(9999.0, fmax(1.0, floor(P.beta * n1 / P.lamb)));
    const double den2 = lb * lb - P.beta;
    const int k2 = den2 > 0.0 ? (int)fmin(9999.0, fmax(1.0, floor(P.beta * n1 / den2))) : 9999;
    const double ref1 = t1_at(k1);
    const double ref2 = t1_at(k2) - (double)(N + k2 + 1) * P.ln_lb;
    const double flim = (upper - P.thr) * exp(-ref2);
    const double qb = P.beta / lb, ilb = 1.0 / lb;
    auto wave_prefix = [&](double e) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(e, off, 64);
            if (lane >= off) e += o;
        }
        return e;
    };
    double S1 = 0.0, S2 = 0.0;
    for (int k0 = 1; k0 < 10000; k0 += 64 * FT) {
        const int kb = k0 + lane * FT;
        double rk[FT];
#pragma unroll
        for (int q = 0; q < FT; q++) rk[q] = invk[kb + q < 10000 ? kb + q : 1];
        const double t1b = kb < 10000 ? t1_at(kb) : 0.0;
        const double e10 = kb < 10000 ? exp(t1b - ref1) : 0.0;
        const double e20 = kb < 10000 ? exp(t1b - (double)(N + kb + 1) * P.ln_lb - ref2) : 0.0;
        double e1 = e10, e2 = e20, r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int q = 0; q < FT; q++) {
            const int k = kb + q;
            const bool in = k < 10000;
            r1 += in ? e1 : 0.0; r2 += in ? e2 : 0.0;
            const double r = qb * (double)(N + k + 1) * rk[q];
            e1 *= r; e2 *= r * ilb;
        }
        const double i1 = wave_prefix(r1), i2 = wave_prefix(r2);
        const double b1 = S1 + (i1 - r1), b2 = S2 + (i2 - r2);
        const unsigned long long m = __ballot(kb < 10000 && !(b2 + r2 < flim));
        if (m) {
            const int f = __ffsll((long long)m) - 1;
            double val = 0.0;
            if (lane == f) {
                double a1 = b1, a2 = b2;
                e1 = e10; e2 = e20;
                bool found = false;
#pragma unroll
                for (int q = 0; q < FT; q++) {
                    const int k = kb + q;
                    if (k < 10000 && !found) { a1 += e1; a2 += e2; val = a1; found = !(a2 < flim); }
                    const double r = qb * (double)(N + k + 1) * rk[q];
                    e1 *= r; e2 *= r * ilb;
                }
            }
            eK = __shfl(val, f, 64) * exp(ref1);
            return true;
        }
        S1 = __shfl(b1 + r1, 63, 64); S2 = __shfl(b2 + r2, 63, 64);
    }
    eK = S1 * exp(ref1);                                              // ran to k = 9999
    return true;
}

__global__ void lgamma_table_kernel(double *__restrict__ lg, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lg[i] = lgamma((double)i);   // lg[0] = +inf like std::lgamma(0.0) (:255-257)
    else if (i < n + LK_TABLE) lg[i] = log((double)(i - n));
    else if (i < n + 2 * LK_TABLE) lg[i] = 1.0 / (double)(i - n - LK_TABLE);
}

// ---- key sources ------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

struct ArraySource {            // trans_dist(snpdiff[], datediff[])
    const int *N;
    const double *delta;
    size_t n;
    __device__ size_t size() const { return n; }
    __device__ bool get(size_t e, int &Nv, double &dv) const { Nv = N[e]; dv = delta[e]; return true; }
    __device__ size_t out_index(size_t e) const { return e; }
    __device__ long long day_gap(size_t) const { return -1; }
    static constexpr bool HAS_GAPS = false;
};

struct DenseSource {            // cells of a dense distance block, delta from sampling days
    const unsigned *dist;
    const int *days;
    size_t ld, n, row_begin, row_end, col_begin;
    int thr;
    size_t row_begin2 = 0, row_end2 = 0;     // optional second row panel (the multi-GPU partition gives each rank two)
    __device__ __host__ size_t rows1() const { return row_end - row_begin; }
    __device__ size_t size() const { return (rows1() + (row_end2 - row_begin2)) * n; }
    __device__ size_t row_of(size_t r) const { return r < rows1() ? row_begin + r : row_begin2 + (r - rows1()); }
    __device__ bool get(size_t e, int &Nv, double &dv) const
    {
        const size_t i = row_of(e / n), j = e % n;
        if (j <= i || j < col_begin) return false;
        const unsigned d = dist[i * ld + j];
        if ((long long)d > (long long)thr) return false;
        Nv = (int)d;
        // tracs/transcluster.py:26-33: |t_i - t_j| / 31556952.0 with t = whole days in seconds (exact in f64)
        const long long dd = (long long)days[i] - (long long)days[j];
        dv = (double)((dd < 0 ? -dd : dd) * 86400ll) / 31556952.0;
        return true;
    }
    __device__ size_t out_index(size_t e) const { return row_of(e / n) * ld + e % n; }
    __device__ long long day_gap(size_t e) const
    {
        const long long dd = (long long)days[row_of(e / n)] - (long long)days[e % n];
        return dd < 0 ? -dd : dd;
    }
    static constexpr bool HAS_GAPS = true;
};

// The keys themselves as a source: element e of the (N, day gap) grid of stride `stride` = gap_max + 1.  What the key kernels read
// when the distinct keys of a dense block were MARKED in that grid (tc_mark_kernel) instead of found by hashing: key_elem holds
// grid indices, N and delta come from the index -- delta by the same expression as DenseSource::get, bit for bit.
struct GridSource {
    unsigned stride;
    size_t cells;
    __device__ size_t size() const { return cells; }
    __device__ bool get(size_t e, int &Nv, double &dv) const
    {
        Nv = (int)(e / stride);
        dv = (double)((long long)(e % stride) * 86400ll) / 31556952.0;
        return true;
    }
    __device__ size_t out_index(size_t e) const { return e; }
    __device__ long long day_gap(size_t e) const { return (long long)(e % stride); }
    static constexpr bool HAS_GAPS = true;
};

// Key table of the multi-GPU path: results of the distinct (N, day gap) keys in a dense [n_max + 1][d_max + 1] layout that
// every rank indexes the same way.  A rank evaluates the keys of its hash class (part of parts) and leaves the others 0, so an
// all-reduce (sum) of the tables completes them: 16 bytes per key travel instead of 16 bytes per pair, and a rank pays for
// 1 / parts of the key evaluations instead of all of them.
struct KeyTable {
    int part = 0, parts = 1;
    unsigned n_max = 0, d_max = 0;
    double *p0 = nullptr, *eK = nullptr;
    unsigned step = 1;                     // doubles between consecutive slots: 1 two arrays, 2 one array of (p0, eK) pairs (eK = p0 + 1)
    unsigned *overflow = nullptr;          // set when a key falls outside the table
    __device__ bool mine(int N, long long gap) const
    {
        return parts <= 1 || (int)(mix64(((unsigned long long)(unsigned)N << 32) ^ (unsigned long long)gap) % (unsigned long long)parts) == part;
    }
    __device__ long long index(int N, long long gap) const
    {
        if (N < 0 || (unsigned)N > n_max || gap < 0 || gap > (long long)d_max) { if (overflow) *overflow = 1u; return -1; }
        return (long long)N * ((long long)d_max + 1) + gap;
    }
};

constexpr unsigned EMPTY = 0xFFFFFFFFu;

// slots[h] = element index of the key's first claimant.  eslot[e] = slot of e's key.
template <class Src>
__global__ void dedup_insert_kernel(Src src, unsigned *__restrict__ slots, unsigned cap_mask,
                                    unsigned *__restrict__ eslot, unsigned *__restrict__ overflow)
{
    // The table starts small (L2-resident: real inputs have few distinct (N, delta) keys); a probe sequence longer than
    // MAX_PROBE means it is too full -- the host then retries with a 16x larger table.
    constexpr int MAX_PROBE = 128;
    const size_t total = src.size();
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int N; double d;
        if (!src.get(e, N, d)) { eslot[e] = EMPTY; continue; }
        unsigned h = (unsigned)mix64((unsigned long long)__double_as_longlong(d) ^ ((unsigned long long)(unsigned)N * 0x9E3779B97F4A7C15ull)) & cap_mask;
        for (int probe = 0;; probe++) {
            if (probe >= MAX_PROBE) { *overflow = 1u; eslot[e] = EMPTY; break; }
            unsigned cur = slots[h];
            if (cur == EMPTY) {
                cur = atomicCAS(&slots[h], EMPTY, (unsigned)e);
                if (cur == EMPTY) { eslot[e] = h; break; }
            }
            int N2; double d2;
            src.get((size_t)cur, N2, d2);
            if (N2 == N && __double_as_longlong(d2) == __double_as_longlong(d)) { eslot[e] = h; break; }
            h = (h + 1u) & cap_mask;
        }
    }
}

__global__ void dedup_count_kernel(const unsigned *__restrict__ slots, unsigned cap, unsigned *__restrict__ n_keys)
{
    unsigned c = 0;
    for (unsigned h = blockIdx.x * blockDim.x + threadIdx.x; h < cap; h += gridDim.x * blockDim.x) c += slots[h] != EMPTY;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_keys, c);
}

// compact the claimed slots into a dense list of keys
__global__ void dedup_collect_kernel(const unsigned *__restrict__ slots, unsigned cap, unsigned *__restrict__ slot_id,
                                     unsigned *__restrict__ key_elem, unsigned *__restrict__ n_keys)
{
    for (unsigned h = blockIdx.x * blockDim.x + threadIdx.x; h < cap; h += gridDim.x * blockDim.x) {
        const unsigned e = slots[h];
        if (e != EMPTY) {
            const unsigned id = atomicAdd(n_keys, 1u);
            slot_id[h] = id;
            key_elem[id] = e;
        }
    }
}

constexpr int TC_SERIAL_CAP = 192;      // terms evaluated by the one-thread-per-key kernel before a key is handed over
constexpr int TC_WAVE_PREFIX_MIN = 128; // delta > 0 and N at least this: the whole key goes to the wave kernel (O(N) prefix included)

template <class Src>
__global__ void tc_keys_kernel(Src src, const unsigned *__restrict__ key_elem, unsigned nk, TcParams P,
                               const double *__restrict__ lg, double *__restrict__ key_p0, double *__restrict__ key_eK,
                               unsigned *__restrict__ long_ids, unsigned *__restrict__ n_long, unsigned *__restrict__ n_zero,
                               double *__restrict__ key_state, KeyTable kt)
{
    P.ln_lamb = log(P.lamb); P.ln_beta = log(P.beta); P.ln_lb = log(P.lamb + P.beta);
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < nk; id += gridDim.x * blockDim.x) {
        int N; double d;
        const size_t elem = (size_t)key_elem[id];
        src.get(elem, N, d);
        long long slot = -1;
        if (kt.p0) {
            const long long gap = src.day_gap(elem);
            if (!kt.mine(N, gap)) { key_p0[id] = 0.0; key_eK[id] = 0.0; continue; }      // another rank's key
            slot = kt.index(N, gap);
        }
        // nothing summed here: the wave kernel does the prefix too.  (Same-day pairs as well: their series is as long, and one thread
        // folding its first TC_SERIAL_CAP terms kept this kernel's last waves running 0.6 ms after the others.)
        // The same-day keys are listed from the back of long_ids: theirs are the longest loops of the wave kernel (no head to skip),
        // which takes them first.
        if (N >= TC_WAVE_PREFIX_MIN) {
            key_state[4 * (size_t)id] = __builtin_nan("");
            if (d == 0.0) long_ids[nk - 1u - atomicAdd(n_zero, 1u)] = id;
            else long_ids[atomicAdd(n_long, 1u)] = id;
            continue;
        }
        double p0, eK = 0.0; int ks;
        const bool done = tc_eval(N, d, P, lg, p0, eK, ks, TC_SERIAL_CAP, key_state + 4 * (size_t)id);
        key_p0[id] = p0;
        if (slot >= 0) kt.p0[slot * kt.step] = p0;
        if (done) { key_eK[id] = eK; if (slot >= 0) kt.eK[slot * kt.step] = eK; }
        else long_ids[atomicAdd(n_long, 1u)] = id;
    }
}

// largest N and day gap among the keys of the call (bounds[0], bounds[1]); sources without day gaps leave bounds[1] = 0
template <class Src>
__global__ void tc_key_bounds_kernel(Src src, const unsigned *__restrict__ key_elem, const unsigned *__restrict__ n_keys, unsigned *__restrict__ bounds)
{
    const unsigned nk = *n_keys;
    unsigned mn = 0, mg = 0;
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < nk; id += gridDim.x * blockDim.x) {
        int N; double d;
        const size_t elem = (size_t)key_elem[id];
        src.get(elem, N, d);
        const long long gap = src.day_gap(elem);
        mn = max(mn, (unsigned)N);
        if (gap > 0) mg = max(mg, (unsigned)min(gap, 0x7FFFFFFFll));
    }
    for (int off = 32; off > 0; off >>= 1) { mn = max(mn, (unsigned)__shfl_xor((int)mn, off, 64)); mg = max(mg, (unsigned)__shfl_xor((int)mg, off, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&bounds[0], mn); atomicMax(&bounds[1], mg); }
}

// one wave per day gap: the two prefix-sum rows of the gap, 64 indices per step (scaled linear sums, like tc_eval_wave)
__global__ __launch_bounds__(64) void tc_tables_kernel(TcTables *__restrict__ tab, const unsigned *__restrict__ bounds, const unsigned *__restrict__ n_keys,
                                                       double *lnS, double *pois, double *lnS_n, unsigned long long pois_cap, TcParams P,
                                                       const double *__restrict__ lg)
{
    const unsigned n_max = bounds[0], gap_max = bounds[1];
    const unsigned long long ldm = (unsigned long long)n_max + 10001ull, ldn = (unsigned long long)n_max + 1ull;
    // the tables must fit, and must be less work than they save: a few thousand terms per key against one table entry per (gap, M)
    const bool ok = gap_max >= 1 && ((unsigned long long)gap_max + 1ull) * ldm <= TC_TABLE_ELEMS && ((unsigned long long)gap_max + 1ull) * ldn <= pois_cap &&
                    ((unsigned long long)gap_max + 1ull) * ldm <= (unsigned long long)*n_keys * 4096ull;
    // linear form (what the term-ratio loop of tc_eval_wave reads) while the largest x of the call keeps exp(-x) a normal double
    const bool linear = lnS_n != nullptr && (double)((long long)gap_max * 86400ll) / 31556952.0 * (P.lamb + P.beta) <= TC_LINEAR_X_MAX;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        tab->lnS = lnS; tab->pois = pois; tab->lnS_n = lnS_n; tab->n_max = n_max; tab->gap_max = gap_max; tab->ldm = (unsigned)ldm; tab->ldn = (unsigned)ldn;
        tab->ok = ok ? 1u : 0u; tab->linear = linear ? 1u : 0u;
    }
    if (!ok) return;
    P.ln_lamb = log(P.lamb); P.ln_beta = log(P.beta); P.ln_lb = log(P.lamb + P.beta);
    const int lane = threadIdx.x & 63;
    auto wave_max = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
        return v;
    };
    auto wave_prefix = [&](double e) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(e, off, 64);
            if (lane >= off) e += o;
        }
        return e;
    };
    if (linear) {
        // Both rows of a gap by the term-ratio recurrence: a_j = arg^j / j! (arg = x for S, lamb delta for the Poisson sum of :144-148),
        // a_(j+1) = a_j arg / (j + 1), in units of the largest term (at j = floor(arg) <= 600: the first term, e^-ref of it, is a normal
        // double) -- a lane takes one exponential per step and FT terms, the steps' run sums are prefix-summed over the wave: a quarter of
        // the steps of the log-space form below, no running maximum, no logarithm but for the short log-space rows (pois, ln S_N).
        constexpr int FT = 4;
        for (unsigned gap = 1 + blockIdx.x; gap <= gap_max; gap += gridDim.x) {
            const double delta = (double)((long long)gap * 86400ll) / 31556952.0;        // the sources' expression (DenseSource::get)
            const double x = delta * (P.lamb + P.beta);
            for (int which = 0; which < 2; which++) {
                const unsigned long long len = which ? ldn : ldm;
                double *__restrict__ row = (which ? pois + (size_t)gap * ldn : lnS + (size_t)gap * ldm);
                double *__restrict__ row_n = lnS_n + (size_t)gap * ldn;
                const double arg = which ? P.lamb * delta : x, la = log(arg);
                const long long jp = (long long)floor(arg);
                const double ref = imul(jp, la) - lg_at(lg, jp + 1);
                const double unit = exp(ref - x);                              // F = S exp(-x) = pre * unit
                double S = 0.0;                                                // (wave-uniform)
                for (unsigned long long i0 = 0; i0 < len; i0 += 64 * FT) {
                    const unsigned long long j0 = i0 + (unsigned long long)lane * FT;
                    double a = j0 < len ? exp(imul((long long)j0, la) - lg_at(lg, (long long)j0 + 1) - ref) : 0.0;
                    double p[FT], r = 0.0;
#pragma unroll
                    for (int q = 0; q < FT; q++) {
                        r += j0 + q < len ? a : 0.0;
                        p[q] = r;
                        a *= arg / (double)(j0 + q + 1);
                    }
                    const double incl = wave_prefix(r), base = S + (incl - r);
#pragma unroll
                    for (int q = 0; q < FT; q++) {
                        const unsigned long long j = j0 + q;
                        if (j >= len) continue;
                        const double pre = base + p[q];
                        if (which) row[j] = ref + log(pre);
                        else { row[j] = pre * unit; if (j < ldn) row_n[j] = ref + log(pre); }
                    }
                    S = __shfl(base + r, 63, 64);
                }
            }
        }
        return;
    }
    for (unsigned gap = 1 + blockIdx.x; gap <= gap_max; gap += gridDim.x) {
        const double delta = (double)((long long)gap * 86400ll) / 31556952.0;        // the sources' expression (DenseSource::get)
        const double ld = log(delta), lx = log(P.lamb * delta);
        for (int which = 0; which < 2; which++) {
            const unsigned long long len = which ? ldn : ldm;
            double *__restrict__ row = (which ? pois + (size_t)gap * ldn : lnS + (size_t)gap * ldm);
            double mx = -INFINITY, scaled = 0.0;
            for (unsigned long long i0 = 0; i0 < len; i0 += 64) {
                const long long i = (long long)(i0 + lane);
                const bool in = (unsigned long long)i < len;
                const double t = !in ? -INFINITY : which ? imul(i, lx) - lg_at(lg, i + 1) : imul(i, ld) + (double)i * P.ln_lb - lg_at(lg, i + 1);
                const double m = wave_max(t);
                if (m > mx) { scaled = mx == -INFINITY ? 0.0 : scaled * exp(mx - m); mx = m; }
                const double pre = scaled + wave_prefix(t == -INFINITY ? 0.0 : exp(t - mx));
                if (in) row[i] = mx + log(pre);
                scaled = __shfl(pre, 63, 64);
            }
        }
    }
}

template <class Src>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TRACS_TC_WAVES, TRACS_TC_WAVES))) void tc_long_keys_kernel(Src src, const unsigned *__restrict__ key_elem,
                                                          const unsigned *__restrict__ long_ids,
                                                          const unsigned *__restrict__ n_long, TcParams P,
                                                          const double *__restrict__ lg, double *__restrict__ key_p0,
                                                          double *__restrict__ key_eK, const double *__restrict__ key_state, KeyTable kt,
                                                          const TcTables *__restrict__ tab)
{
    P.ln_lamb = log(P.lamb); P.ln_beta = log(P.beta); P.ln_lb = log(P.lamb + P.beta);
    const unsigned nl = *n_long;
    for (unsigned w = blockIdx.x; w < nl; w += gridDim.x) {
        const unsigned id = long_ids[w];
        int N; double d;
        const size_t elem = (size_t)key_elem[id];
        src.get(elem, N, d);
        double eK, p0 = 0.0;
        const double *st = key_state + 4 * (size_t)id;
        const bool fresh = st[0] != st[0];
        tc_eval_wave(N, d, P, lg, eK, st, TC_SERIAL_CAP, &p0, tab, src.day_gap(elem));
        if ((threadIdx.x & 63) == 0) {
            key_eK[id] = eK;
            if (fresh) key_p0[id] = p0;
            if (kt.p0) {
                const long long slot = kt.index(N, src.day_gap(elem));
                if (slot >= 0) { kt.eK[slot * kt.step] = eK; if (fresh) kt.p0[slot * kt.step] = p0; }
            }
        }
    }
}

// the fresh long keys (nothing summed yet: N >= TC_WAVE_PREFIX_MIN) through the term-ratio loop; the others, and the keys it does not
// take, go on to tc_long_keys_kernel through rest_ids
template <class Src>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TRACS_TC_RW, TRACS_TC_RW))) void tc_ratio_keys_kernel(Src src, const unsigned *__restrict__ key_elem, const unsigned *__restrict__ long_ids,
                                                           const unsigned *__restrict__ n_long, const unsigned *__restrict__ n_zero, unsigned nk,
                                                           TcParams P, const double *__restrict__ lg,
                                                           double *__restrict__ key_p0, double *__restrict__ key_eK,
                                                           const double *__restrict__ key_state, KeyTable kt, const TcTables *__restrict__ tab,
                                                           unsigned *__restrict__ rest_ids, unsigned *__restrict__ n_rest)
{
    P.ln_lamb = log(P.lamb); P.ln_beta = log(P.beta); P.ln_lb = log(P.lamb + P.beta);
    const unsigned nz = *n_zero, nl = nz + *n_long;
    for (unsigned w = blockIdx.x; w < nl; w += gridDim.x) {
        const unsigned id = w < nz ? long_ids[nk - 1u - w] : long_ids[w - nz];      // (same-day keys first: the longest loops)
        int N; double d;
        const size_t elem = (size_t)key_elem[id];
        src.get(elem, N, d);
        const long long gap = src.day_gap(elem);
        const double *st = key_state + 4 * (size_t)id;
        const bool fresh = st[0] != st[0];
        double eK = 0.0, p0 = 0.0;
        const bool took = fresh && (d == 0.0 ? tc_eval_ratio_zero(N, P, lg, eK, p0) : tc_eval_ratio(N, d, gap, P, lg, tab, eK, p0));     // (wave-uniform)
        if ((threadIdx.x & 63) != 0) continue;
        if (!took) { rest_ids[atomicAdd(n_rest, 1u)] = id; continue; }
        key_eK[id] = eK; key_p0[id] = p0;
        if (kt.p0) {
            const long long slot = kt.index(N, gap);
            if (slot >= 0) { kt.eK[slot * kt.step] = eK; kt.p0[slot * kt.step] = p0; }
        }
    }
}

template <class Src>
__global__ void tc_gather_kernel(Src src, const unsigned *__restrict__ eslot, const unsigned *__restrict__ slot_id,
                                 const double *__restrict__ key_p0, const double *__restrict__ key_eK, int exp_p0,
                                 double *__restrict__ p0, double *__restrict__ eK)
{
    const size_t total = src.size();
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = eslot[e];
        if (h == EMPTY) continue;
        const unsigned id = slot_id[h];
        const size_t o = src.out_index(e);
        const double v = key_p0[id];
        p0[o] = exp_p0 ? exp(v) : v;
        eK[o] = key_eK[id];
    }
}

// ---- dense blocks whose keys fit a grid: no hash table, no per-cell slot array ---------------------------------------------------
// trans_dist memoises per (N, delta) key (src/transcluster.hpp:245-246,265-282).  The keys of a dense block are (SNP distance, day
// gap): when (largest distance + 1) x (largest gap + 1) <= TC_GRID_BITS they are marked in a bitmap indexed by the key itself (2 MB,
// L2-resident; a cell whose bit is set already costs one cached read), the set bits ARE the distinct keys, their results go into
// the dense (N, gap) tables the multi-GPU path already uses, and every cell reads its values from there -- round 3 wrote a 4-byte
// hash slot per cell (400 MB at 10 000 samples), claimed slots with atomicCAS and read the slots back to gather.
constexpr unsigned long long TC_GRID_BITS = 1ull << 24;

// Row-structured passes over a dense block (bounds, marking, gather): workgroup (x, y) is row x of the block and slice y of its
// columns; a thread takes four columns at a time, starting at the row's first cell: no division per cell, nothing issued below the
// diagonal.  The four are TWO PAIRS, 128 columns apart: of the 256 columns a wave covers per step, lane l takes 2 l, 2 l + 1 and
// 128 + 2 l, 128 + 2 l + 1 -- 8-byte loads of the distances and the days, and a pair's two doubles of P (of E(K)) are ONE 16-byte
// store whose 64 lanes write 1 KiB without a gap.  (Four consecutive columns per lane until round 6: each 16-byte store filled half
// of a 32-byte sector, the other half came with the next instruction -- the memory side saw twice the bytes: 1.55 GB written for
// 0.8 GB of upper triangle, profiles/r05/pmc_bench_c3.txt.)
struct RowQuad {
    size_t i, j[2];              // row (sample index), first column of either pair (even)
    unsigned d[4];               // cells j[0], j[0] + 1, j[1], j[1] + 1
    int day[4];
    bool ok[4];                  // the column is a cell of the block within the threshold
};
constexpr unsigned TC_ROW_THREADS = 256;
__device__ __forceinline__ bool tc_wide(const DenseSource &src)
{
    return src.ld % 4 == 0 && (reinterpret_cast<size_t>(src.dist) | reinterpret_cast<size_t>(src.days)) % 16 == 0;
}
template <class F>
__device__ __forceinline__ void tc_for_row_quads(const DenseSource &src, F f)
{
    const size_t i = src.row_of(blockIdx.x);
    const size_t jlo = max(i + 1, src.col_begin);
    const bool wide = tc_wide(src);
    const unsigned *row = src.dist + i * src.ld;
    const unsigned lane = threadIdx.x & 63u;
    // q0: the first quad of this thread's WAVE in this step (64 quads = 256 columns per wave and step)
    for (size_t q0 = jlo / 4 + (size_t)blockIdx.y * TC_ROW_THREADS + (threadIdx.x - lane); q0 * 4 < src.n; q0 += (size_t)gridDim.y * TC_ROW_THREADS) {
        RowQuad c;
        c.i = i; c.j[0] = q0 * 4 + 2 * lane; c.j[1] = c.j[0] + 128;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (wide && c.j[h] + 1 < src.n) {
                // (the distances stream by once: non-temporal, so that they do not push the key tables / the key bitmap -- read at random
                // by every cell -- out of L2)
                typedef unsigned tc_u32x2 __attribute__((ext_vector_type(2)));
                const tc_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const tc_u32x2 *>(row + c.j[h]));
                const int2 t = *reinterpret_cast<const int2 *>(src.days + c.j[h]);
                c.d[2 * h] = v.x; c.d[2 * h + 1] = v.y;
                c.day[2 * h] = t.x; c.day[2 * h + 1] = t.y;
            } else {
#pragma unroll
                for (int k = 0; k < 2; k++) { const bool in = c.j[h] + k < src.n; c.d[2 * h + k] = in ? row[c.j[h] + k] : 0u; c.day[2 * h + k] = in ? src.days[c.j[h] + k] : 0; }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) c.ok[2 * h + k] = c.j[h] + k >= jlo && c.j[h] + k < src.n && (long long)c.d[2 * h + k] <= (long long)src.thr;
        }
        f(c);
    }
}
static dim3 tc_row_grid(const DenseSource &src)
{
    const size_t rows = src.rows1() + (src.row_end2 - src.row_begin2);
    const size_t quads = (src.n + 3) / 4;
    return dim3((unsigned)rows, (unsigned)std::min<size_t>(4, std::max<size_t>(1, (quads + TC_ROW_THREADS - 1) / TC_ROW_THREADS)));
}

// cb[1] = smallest day + 2^31, cb[2] = largest day + 2^31 (over all samples): one workgroup
__global__ __launch_bounds__(1024) void tc_day_bounds_kernel(const int *__restrict__ days, size_t n, unsigned *__restrict__ cb)
{
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t s = threadIdx.x; s < n; s += blockDim.x) { const unsigned v = (unsigned)days[s] + 0x80000000u; lo = min(lo, v); hi = max(hi, v); }
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64)); hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&cb[1], lo); atomicMax(&cb[2], hi); }
}

// One pass over the cells: cb[0] = largest distance of a valid cell, and bit N * stride + gap of every valid cell's key (stride = the
// span of the days + 1, known before the pass: tc_day_bounds_kernel); cb[3] = 1 when a key falls outside the grid's TC_GRID_BITS bits --
// then the marks are incomplete and the caller takes the hash route.  (Round 4 swept the cells twice: the bounds first.)
// A row's keys fall into a narrow band of the grid (its distances differ by tens, the gaps span the days): the workgroup keeps a WINDOW
// of the bitmap in LDS -- TC_MARK_WIN words from the word of (the row's first distance - half the window's span of distances) x stride
// on --, sets its bits there (LDS atomics instead of a scattered global read and, for a new key, a global atomic per cell: 50 M of them at
// 10 000 samples, and 500 000 contending for 640 words at 1 000), and ORs the window's non-zero words into the grid at the end.  Keys
// outside the window take the global way.
constexpr unsigned TC_MARK_WIN = 3072;              // words: 98 304 keys = 134 distances x a two-year span of days
__global__ __launch_bounds__(TC_ROW_THREADS) void tc_mark_kernel(DenseSource src, unsigned *__restrict__ cb, unsigned *__restrict__ bits)
{
    __shared__ unsigned win[TC_MARK_WIN];
    __shared__ unsigned anchor_d;
    if (cb[2] < cb[1]) { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) cb[3] = 1u; return; }
    const unsigned long long stride = (unsigned long long)(cb[2] - cb[1]) + 1ull;
    const size_t i_row = src.row_of(blockIdx.x);
    const int di = src.days[i_row];
    for (unsigned w = threadIdx.x; w < TC_MARK_WIN; w += TC_ROW_THREADS) win[w] = 0u;
    if (threadIdx.x == 0) {
        const size_t j0 = max(i_row + 1, src.col_begin);
        anchor_d = j0 < src.n ? src.dist[i_row * src.ld + j0] : 0u;
    }
    __syncthreads();
    const unsigned long long span_d = ((unsigned long long)TC_MARK_WIN * 32ull) / stride;       // distances the window spans
    const unsigned long long d0 = anchor_d > span_d / 2 ? anchor_d - span_d / 2 : 0ull;
    const unsigned long long w_base = (d0 * stride) >> 5;
    unsigned mn = 0;
    bool beyond = false;
    tc_for_row_quads(src, [&](const RowQuad &c) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!c.ok[k]) continue;
            mn = max(mn, c.d[k]);
            const long long g = (long long)di - (long long)c.day[k];
            const unsigned long long key = (unsigned long long)c.d[k] * stride + (unsigned long long)(g < 0 ? -g : g);
            if (key >= TC_GRID_BITS) { beyond = true; continue; }
            const unsigned idx = (unsigned)key;
            const unsigned bit = 1u << (idx & 31u);
            const unsigned long long w = (unsigned long long)(idx >> 5);
            if (w >= w_base && w < w_base + TC_MARK_WIN) atomicOr(&win[(unsigned)(w - w_base)], bit);
            else if (!(bits[idx >> 5] & bit)) atomicOr(&bits[idx >> 5], bit);
        }
    });
    __syncthreads();
    for (unsigned w = threadIdx.x; w < TC_MARK_WIN; w += TC_ROW_THREADS) {
        const unsigned v = win[w];
        if (v && w_base + w < TC_GRID_BITS / 32) {
            unsigned *dst = &bits[(size_t)(w_base + w)];
            if ((*dst & v) != v) atomicOr(dst, v);
        }
    }
    for (int off = 32; off > 0; off >>= 1) mn = max(mn, (unsigned)__shfl_xor((int)mn, off, 64));
    // (one address for every workgroup of the grid: an atomic only from a wave that would raise what is there)
    if ((threadIdx.x & 63) == 0 && mn > __hip_atomic_load(&cb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cb[0], mn);
    if (__ballot(beyond) && (threadIdx.x & 63) == 0) cb[3] = 1u;
}

// the keys' log probabilities in a (p0, eK) pair table -> probabilities, once per KEY (the gather then copies: an exponential per cell
// was 50 M of them at 10 000 samples)
__global__ void tc_exp_keys_kernel(const unsigned *__restrict__ key_elem, const unsigned *__restrict__ n_keys, double *__restrict__ tables)
{
    const unsigned nk = *n_keys;
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < nk; id += gridDim.x * blockDim.x) {
        double *p = tables + 2 * (size_t)key_elem[id];
        *p = exp(*p);
    }
}

__global__ void tc_bits_count_kernel(const unsigned *__restrict__ bits, unsigned words, unsigned *__restrict__ n_keys)
{
    unsigned c = 0;
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x) c += __popc(bits[w]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_keys, c);
}

__global__ void tc_bits_collect_kernel(const unsigned *__restrict__ bits, unsigned words, unsigned *__restrict__ key_elem, unsigned *__restrict__ n_keys)
{
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x) {
        unsigned x = bits[w];
        if (!x) continue;
        unsigned id = atomicAdd(n_keys, (unsigned)__popc(x));
        while (x) { key_elem[id++] = w * 32u + (unsigned)(__ffs(x) - 1); x &= x - 1; }
    }
}

// P / E(K) of every cell from the completed key tables, four cells of a row per thread: 16-byte loads of d and the days, 16-byte
// stores of P and E(K) where both cells of a pair are cells of the block (a pair straddling the diagonal, the column bound or the
// threshold stores singly)
__global__ __launch_bounds__(TC_ROW_THREADS) void tc_table_gather2_kernel(DenseSource src, KeyTable kt, int exp_p0, double *__restrict__ p0,
                                                                          double *__restrict__ eK)
{
    const bool wide_out = tc_wide(src) && (reinterpret_cast<size_t>(p0) | reinterpret_cast<size_t>(eK)) % 16 == 0;
    const int di = src.days[src.row_of(blockIdx.x)];
    tc_for_row_quads(src, [&](const RowQuad &c) {
        double vp[4], ve[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ok[k] = c.ok[k];
            vp[k] = ve[k] = 0.0;
            if (ok[k]) {
                const long long g = (long long)di - (long long)c.day[k];
                const long long slot = kt.index((int)c.d[k], g < 0 ? -g : g);
                ok[k] = slot >= 0;
                if (ok[k]) {
                    double v;
                    if (kt.step == 2) { const double2 pe = *reinterpret_cast<const double2 *>(kt.p0 + 2 * slot); v = pe.x; ve[k] = pe.y; }      // one 16-byte read per cell
                    else { v = kt.p0[slot]; ve[k] = kt.eK[slot]; }
                    vp[k] = exp_p0 ? exp(v) : v;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 4; h += 2) {
            const size_t o = c.i * src.ld + c.j[h >> 1];
            if (wide_out && ok[h] && ok[h + 1]) {
                typedef double tc_f64x2 __attribute__((ext_vector_type(2)));
                tc_f64x2 a, b;
                a.x = vp[h]; a.y = vp[h + 1]; b.x = ve[h]; b.y = ve[h + 1];
                __builtin_nontemporal_store(a, reinterpret_cast<tc_f64x2 *>(p0 + o));
                __builtin_nontemporal_store(b, reinterpret_cast<tc_f64x2 *>(eK + o));
            } else {
                if (ok[h]) { p0[o] = vp[h]; eK[o] = ve[h]; }
                if (ok[h + 1]) { p0[o + 1] = vp[h + 1]; eK[o + 1] = ve[h + 1]; }
            }
        }
    });
}

// lprob_k_given_N (older formulation, exported for tests/test_llk.py): transcluster.hpp:90-129
__global__ void lprob_k_given_N_kernel(const unsigned long long *__restrict__ Ns, const unsigned long long *__restrict__ ks,
                                       const double *__restrict__ deltas, size_t n, double lamb, double beta,
                                       const double *__restrict__ lg, double *__restrict__ out_lprob,
                                       double *__restrict__ out_lhs)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const long long N = (long long)Ns[e], k = (long long)ks[e];
    const double delta = deltas[e];
    double lprob, lhs;
    if (delta > 0) {
        lprob = ((double)(N + 1) * log(lamb) - delta * (lamb + beta) + (double)k * log(beta) - lg[k + 1]);
        double pois = -INFINITY;
        const double lx = log(lamb * delta);
        for (long long i = 0; i <= N; i++) pois = lae(imul(i, lx) - lg[i + 1], pois);
        pois -= lamb * delta;
        lprob -= pois;
        double integral = -INFINITY;
        const double ld = log(delta), llb = log(lamb + beta);
        for (long long i = 0; i <= N + k; i++)
            integral = lae(lg[N + k + 1] - lg[i + 1] - lg[N + k - i + 1] + imul(N + k - i, ld) + lg[i + 1] -
                               (double)(i + 1) * llb,
                           integral);
        integral -= lg[N + 1];
        lhs = lprob;
        lprob += integral;
    } else {
        lprob = ((double)(N + 1) * log(lamb) + (double)k * log(beta) + lg[N + k + 1] - lg[N + 1] - lg[k + 1] -
                 (double)(N + k + 1) * log(lamb + beta));
        lhs = lprob;
    }
    out_lprob[e] = lprob;
    out_lhs[e] = lhs;
}

// process-wide lgamma table (built once per device)
static double *g_lg[64] = {nullptr};

static int get_lgamma_table(hipStream_t stream, const double **out)
{
    int dev = 0;
    TRACS_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index out of range"); return TRACS_E_HIP; }
    if (!g_lg[dev]) {
        double *p = nullptr;
        TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), (LG_TABLE + 2 * LK_TABLE) * sizeof(double)));
        hipLaunchKernelGGL(lgamma_table_kernel, dim3((LG_TABLE + 2 * LK_TABLE + 255) / 256), dim3(256), 0, stream, p, LG_TABLE);
        TRACS_HIP_CHECK(hipGetLastError());
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        g_lg[dev] = p;
    }
    *out = g_lg[dev];
    return TRACS_OK;
}

int get_lgamma_table_for_filter(hipStream_t stream, const double **out) { return get_lgamma_table(stream, out); }

struct TcWorkspaceIds { enum { SLOTS = 0, ESLOT, SLOT_ID, NKEYS, KEY_ELEM, KEY_P0, KEY_EK, LONG_IDS, KEY_STATE, TAB, TAB_LNS, TAB_POIS, GRID_BITS, GRID_TABLES, TAB_LNS_N, REST_IDS }; };
constexpr unsigned long long TC_POIS_ELEMS = 16ull << 20;        // doubles in the pois table (128 MB)

static unsigned long long g_last_keys = 0;        // distinct (N, delta) keys of the last entry-point call (bench.py reports it)
constexpr size_t TD_MAX_ELEMS = 1ull << 31;       // elements per pass: element indices and slots are 32-bit

// The distinct keys are known (key_elem[0 .. nk): elements of `src` that carry them; n_keys[0] = nk on the device, n_keys[1] = 0):
// evaluate each once -- short series by one thread, long ones by a wave, dense blocks with the (gap, M) tables of their prefix sums.
// Results in key_p0 / key_eK (by key id) and, when kt has tables, at kt.index(N, gap).
template <class Src>
static int tc_evaluate_keys(const Src &src, const unsigned *key_elem, unsigned nk, unsigned *n_keys, double lamb, double beta, double thr,
                            const double *lg, double **key_p0_out, double **key_eK_out, const KeyTable &kt, hipStream_t stream)
{
    int rc;
    unsigned *long_ids;
    double *key_p0, *key_eK;
    if ((rc = workspace_get(TcWorkspaceIds::KEY_P0, (size_t)nk * 8, reinterpret_cast<void **>(&key_p0)))) return rc;
    if ((rc = workspace_get(TcWorkspaceIds::KEY_EK, (size_t)nk * 8, reinterpret_cast<void **>(&key_eK)))) return rc;
    if ((rc = workspace_get(TcWorkspaceIds::LONG_IDS, (size_t)nk * 4, reinterpret_cast<void **>(&long_ids)))) return rc;
    double *key_state = nullptr;
    if ((rc = workspace_get(TcWorkspaceIds::KEY_STATE, (size_t)nk * 32, reinterpret_cast<void **>(&key_state)))) return rc;
    TcParams P;
    P.lamb = lamb; P.beta = beta; P.thr = thr;
    P.ln_lamb = P.ln_beta = P.ln_lb = 0.0;

    // one wave per block: keys differ widely in trip count, small blocks keep the SIMDs busy
    TRACS_HIP_CHECK(hipMemsetAsync(n_keys + 8, 0, 8, stream));            // [8] same-day long keys, [9] keys left to the log-space loop
    hipLaunchKernelGGL((tc_keys_kernel<Src>), dim3((nk + 63) / 64), dim3(64), 0, stream, src, key_elem, nk, P, lg, key_p0, key_eK,
                       long_ids, n_keys + 1, n_keys + 8, key_state, kt);
    // (gap, M) tables of the prefix sums, when the source has day gaps and the keys' bounds fit (decided on the device)
    TcTables *tab = nullptr;
    double *tab_lnS = nullptr, *tab_pois = nullptr, *tab_lnS_n = nullptr;
    if ((rc = workspace_get(TcWorkspaceIds::TAB, 256, reinterpret_cast<void **>(&tab)))) return rc;
    TRACS_HIP_CHECK(hipMemsetAsync(tab, 0, 256, stream));                 // ok = 0
    // The tables are an optional speed-up (ok = 0 is a working path): only worth their 512 MB of workspace when there are enough
    // keys to share the per-gap sums, and not getting the memory -- e.g. beside a rank's alignment, panels and lists -- is not an error.
    if (Src::HAS_GAPS && nk >= 2048) {
        const bool have = workspace_get(TcWorkspaceIds::TAB_LNS, TC_TABLE_ELEMS * 8, reinterpret_cast<void **>(&tab_lnS)) == TRACS_OK &&
                          workspace_get(TcWorkspaceIds::TAB_POIS, TC_POIS_ELEMS * 8, reinterpret_cast<void **>(&tab_pois)) == TRACS_OK;
        if (have) {
            // (the linear form needs ln S_N beside it; without that workspace the tables stay in log space)
            if (workspace_get(TcWorkspaceIds::TAB_LNS_N, TC_POIS_ELEMS * 8, reinterpret_cast<void **>(&tab_lnS_n)) != TRACS_OK) {
                (void)hipGetLastError(); set_error(""); tab_lnS_n = nullptr;
            }
            unsigned *bounds = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(tab) + 128);
            hipLaunchKernelGGL((tc_key_bounds_kernel<Src>), dim3(256), dim3(256), 0, stream, src, key_elem, n_keys, bounds);
            hipLaunchKernelGGL(tc_tables_kernel, dim3(1024), dim3(64), 0, stream, tab, bounds, n_keys, tab_lnS, tab_pois, tab_lnS_n, TC_POIS_ELEMS, P, lg);
        } else {
            (void)hipGetLastError();
            set_error("");
        }
    }
    // long series (E(K) loop beyond TC_SERIAL_CAP terms): one wave per key -- the term-ratio loop where the linear tables hold the key,
    // the log-space loop for the rest (n_keys[9] of them, listed by the first kernel)
    unsigned *rest_ids = nullptr;
    if ((rc = workspace_get(TcWorkspaceIds::REST_IDS, (size_t)nk * 4, reinterpret_cast<void **>(&rest_ids)))) return rc;
    hipLaunchKernelGGL((tc_ratio_keys_kernel<Src>), dim3(std::min<unsigned>(nk, 256u * 32u)), dim3(64), 0, stream, src, key_elem,
                       long_ids, n_keys + 1, n_keys + 8, nk, P, lg, key_p0, key_eK, key_state, kt, tab, rest_ids, n_keys + 9);
    hipLaunchKernelGGL((tc_long_keys_kernel<Src>), dim3(std::min<unsigned>(nk, 256u * 32u)), dim3(64), 0, stream, src, key_elem,
                       rest_ids, n_keys + 9, P, lg, key_p0, key_eK, key_state, kt, tab);
    *key_p0_out = key_p0; *key_eK_out = key_eK;
    return TRACS_OK;
}

template <class Src>
static int run_trans_dist(const Src &src, size_t total, double lamb, double beta, double thr, int exp_p0, double *p0,
                          double *eK, hipStream_t stream, const KeyTable &kt = KeyTable())
{
    if (total == 0) return TRACS_OK;
    DeviceCall guard(stream);
    if (total > TD_MAX_ELEMS) { set_error("trans_dist: more than 2^31 elements per pass (internal error)"); return TRACS_E_ARG; }
    const double *lg = nullptr;
    int rc = get_lgamma_table(stream, &lg);
    if (rc) return rc;
    unsigned cap_max = 1024;
    while ((size_t)cap_max < 2 * total && cap_max < (1u << 31)) cap_max <<= 1;
    unsigned cap = std::min(cap_max, 1u << 20);
    unsigned *slots, *eslot, *slot_id, *n_keys, *key_elem;
    if ((rc = workspace_get(TcWorkspaceIds::ESLOT, total * 4, reinterpret_cast<void **>(&eslot)))) return rc;
    if ((rc = workspace_get(TcWorkspaceIds::NKEYS, 64, reinterpret_cast<void **>(&n_keys)))) return rc;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    unsigned nk = 0;
    for (;;) {
        if ((rc = workspace_get(TcWorkspaceIds::SLOTS, (size_t)cap * 4, reinterpret_cast<void **>(&slots)))) return rc;
        if ((rc = workspace_get(TcWorkspaceIds::SLOT_ID, (size_t)cap * 4, reinterpret_cast<void **>(&slot_id)))) return rc;
        TRACS_HIP_CHECK(hipMemsetAsync(slots, 0xFF, (size_t)cap * 4, stream));
        TRACS_HIP_CHECK(hipMemsetAsync(n_keys, 0, 16, stream));     // [0] keys, [1] long keys, [2] overflow flag
        hipLaunchKernelGGL((dedup_insert_kernel<Src>), dim3(blocks), dim3(256), 0, stream, src, slots, cap - 1, eslot, n_keys + 2);
        TRACS_HIP_CHECK(hipGetLastError());
        // first pass over the table only counts the claimed slots so the key arrays can be sized
        // exactly (one small readback per call), the second assigns ids.
        hipLaunchKernelGGL(dedup_count_kernel, dim3((unsigned)std::min<size_t>((cap + 255) / 256, 256 * 32)), dim3(256), 0, stream,
                           slots, cap, n_keys);
        unsigned h3[3] = {0, 0, 0};
        TRACS_HIP_CHECK(hipMemcpyAsync(h3, n_keys, 12, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        nk = h3[0];
        if (!h3[2] && (size_t)nk * 2 <= cap) break;                 // fits with load factor <= 0.5
        if (cap >= cap_max) {
            if (h3[2]) { set_error("trans_dist: key table overflow"); return TRACS_E_HIP; }
            break;
        }
        cap = (unsigned)std::min<unsigned long long>((unsigned long long)cap * 16ull, cap_max);
    }
    g_last_keys += nk;
    if (nk == 0) return TRACS_OK;
    if ((rc = workspace_get(TcWorkspaceIds::KEY_ELEM, (size_t)nk * 4, reinterpret_cast<void **>(&key_elem)))) return rc;
    TRACS_HIP_CHECK(hipMemsetAsync(n_keys, 0, 8, stream));          // [0] key counter, [1] long-key counter
    hipLaunchKernelGGL(dedup_collect_kernel, dim3((unsigned)std::min<size_t>((cap + 255) / 256, 256 * 32)), dim3(256), 0, stream,
                       slots, cap, slot_id, key_elem, n_keys);
    double *key_p0 = nullptr, *key_eK = nullptr;
    if ((rc = tc_evaluate_keys(src, key_elem, nk, n_keys, lamb, beta, thr, lg, &key_p0, &key_eK, kt, stream))) return rc;
    if (p0 && eK)                        // (the key-table form fills its table only)
        hipLaunchKernelGGL((tc_gather_kernel<Src>), dim3(blocks), dim3(256), 0, stream, src, eslot, slot_id, key_p0, key_eK,
                           exp_p0, p0, eK);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// A dense block whose keys fit the (N, day gap) grid: marked, evaluated into dense tables, gathered (see tc_mark_kernel).  *done = 0
// when the grid does not fit (or its tables cannot be had): the caller takes the hash route.
static int run_trans_dist_grid(const DenseSource &src, size_t total, double lamb, double beta, double thr, int exp_p0, double *p0,
                               double *eK, hipStream_t stream, int *done)
{
    *done = 0;
    if (total == 0) { *done = 1; return TRACS_OK; }
    static const bool off = [] { const char *e = std::getenv("TRACS_TC_GRID"); return e && std::atoi(e) == 0; }();
    if (off) return TRACS_OK;
    DeviceCall guard(stream);
    const double *lg = nullptr;
    int rc = get_lgamma_table(stream, &lg);
    if (rc) return rc;
    unsigned *n_keys = nullptr, *bits = nullptr, *key_elem = nullptr;
    constexpr unsigned words = (unsigned)(TC_GRID_BITS / 32);
    if ((rc = workspace_get(TcWorkspaceIds::NKEYS, 64, reinterpret_cast<void **>(&n_keys)))) return rc;
    if ((rc = workspace_get(TcWorkspaceIds::GRID_BITS, (size_t)words * 4, reinterpret_cast<void **>(&bits)))) return rc;
    unsigned *cb = n_keys + 4;                                       // [4] max N, [5] min day, [6] max day, [7] does not fit
    const unsigned init[8] = {0u, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u};
    TRACS_HIP_CHECK(hipMemcpyAsync(n_keys, init, 32, hipMemcpyHostToDevice, stream));
    TRACS_HIP_CHECK(hipMemsetAsync(bits, 0, (size_t)words * 4, stream));
    const dim3 row_grid = tc_row_grid(src);
    hipLaunchKernelGGL(tc_day_bounds_kernel, dim3(1), dim3(1024), 0, stream, src.days, src.n, cb);
    // (marking: one workgroup per row -- its LDS window is cleared and flushed once per workgroup)
    hipLaunchKernelGGL(tc_mark_kernel, dim3(row_grid.x, src.n <= 32768 ? 1u : row_grid.y), dim3(TC_ROW_THREADS), 0, stream, src, cb, bits);
    hipLaunchKernelGGL(tc_bits_count_kernel, dim3(1024), dim3(256), 0, stream, bits, words, n_keys);
    unsigned h[8] = {0};
    TRACS_HIP_CHECK(hipMemcpyAsync(h, n_keys, 32, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    if (h[7]) return TRACS_OK;                                       // the grid does not fit: hash route
    const unsigned nk = h[0];
    g_last_keys += nk;
    *done = 1;
    if (nk == 0) return TRACS_OK;
    const unsigned n_max = h[4], d_max = h[6] - h[5];
    const size_t cells = ((size_t)n_max + 1) * ((size_t)d_max + 1);
    double *tables = nullptr;
    if (workspace_get(TcWorkspaceIds::GRID_TABLES, cells * 16, reinterpret_cast<void **>(&tables)) != TRACS_OK) {
        (void)hipGetLastError(); set_error(""); g_last_keys -= nk; *done = 0; return TRACS_OK;
    }
    if ((rc = workspace_get(TcWorkspaceIds::KEY_ELEM, (size_t)nk * 4, reinterpret_cast<void **>(&key_elem)))) return rc;
    TRACS_HIP_CHECK(hipMemsetAsync(n_keys, 0, 8, stream));          // [0] key counter, [1] long-key counter
    hipLaunchKernelGGL(tc_bits_collect_kernel, dim3(1024), dim3(256), 0, stream, bits, words, key_elem, n_keys);
    KeyTable kt;
    kt.n_max = n_max; kt.d_max = d_max; kt.p0 = tables; kt.eK = tables + 1; kt.step = 2;      // (p0, eK) pairs: what a cell reads is 16 contiguous bytes
    kt.overflow = n_keys + 7;                                        // (cannot happen: every cell's key was marked inside the grid)
    GridSource grid{d_max + 1u, cells};
    double *key_p0 = nullptr, *key_eK = nullptr;
    if ((rc = tc_evaluate_keys(grid, key_elem, nk, n_keys, lamb, beta, thr, lg, &key_p0, &key_eK, kt, stream))) return rc;
    // (GridSource: key_elem holds the keys' grid indices = their slots in `tables`; n_keys[0] = nk again after the collect)
    if (exp_p0) hipLaunchKernelGGL(tc_exp_keys_kernel, dim3(256), dim3(256), 0, stream, key_elem, n_keys, tables);
    hipLaunchKernelGGL(tc_table_gather2_kernel, row_grid, dim3(TC_ROW_THREADS), 0, stream, src, kt, 0, p0, eK);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// ---- the distinct keys split over the ranks (site shards, every rank holds ROWS of the summed matrix) -----------------
// A rank's rows see most of the matrix's distinct (N, day gap) keys, so evaluating "its" keys is nearly all of them on every rank.
// Instead: every rank marks the keys of its rows in the grid bitmap (tc_mark_kernel; KS_WORDS words + a trailer of four: largest
// distance, smallest / largest day + 2^31, "a key fell outside the grid"), the bitmaps are all-gathered and ORed -- every rank now
// holds the SAME union --, the keys are numbered by their position in the union (ordinal = set bits before it), rank r evaluates the
// ordinals o with o % P == r into a compact array (slot o / P: log p0, E(K)), the arrays are all-gathered (16 bytes per distinct key
// in total), and every rank fills its dense (N, gap) table from them and gathers its rows.  No atomics in the numbering: every rank
// derives the same ordinals from the same bitmap.
constexpr unsigned KS_WORDS = (unsigned)(TC_GRID_BITS / 32);
constexpr unsigned KS_TRAILER = 4;
constexpr unsigned KS_CHUNK = 1024;                         // words per workgroup of the numbering
constexpr unsigned KS_CHUNKS = KS_WORDS / KS_CHUNK;
static_assert(KS_CHUNKS == 512, "ks_chunk_scan_kernel: one thread per chunk");
struct KsWs { enum { RANK = 80, CHUNK, INFO }; };

__global__ void ks_merge_kernel(unsigned *__restrict__ own, const unsigned *__restrict__ all, int parts)
{
    const size_t stride = (size_t)KS_WORDS + KS_TRAILER;
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < KS_WORDS + KS_TRAILER; w += gridDim.x * blockDim.x) {
        unsigned v = all[w];
        for (int p = 1; p < parts; p++) {
            const unsigned o = all[(size_t)p * stride + w];
            if (w < KS_WORDS || w == KS_WORDS + 3u) v |= o;          // marks; "does not fit"
            else if (w == KS_WORDS + 1u) v = min(v, o);              // smallest day
            else v = max(v, o);                                      // largest distance, largest day
        }
        own[w] = v;
    }
}

// word_rank[w] = set bits of the chunk before word w; chunk_sum[c] = set bits of chunk c
__global__ __launch_bounds__(256) void ks_rank_kernel(const unsigned *__restrict__ bits, unsigned *__restrict__ word_rank, unsigned *__restrict__ chunk_sum)
{
    __shared__ unsigned wtot[4];
    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const size_t w0 = (size_t)blockIdx.x * KS_CHUNK + 4u * t;
    const uint4 v = *reinterpret_cast<const uint4 *>(bits + w0);
    const unsigned c0 = __popc(v.x), c1 = __popc(v.y), c2 = __popc(v.z), c3 = __popc(v.w), c = c0 + c1 + c2 + c3;
    unsigned x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(x, off, 64); if ((int)lane >= off) x += o; }
    if (lane == 63u) wtot[wave] = x;
    __syncthreads();
    unsigned before = x - c;
    for (unsigned k = 0; k < wave; k++) before += wtot[k];
    *reinterpret_cast<uint4 *>(word_rank + w0) = make_uint4(before, before + c0, before + c0 + c1, before + c0 + c1 + c2);
    if (t == 255u) chunk_sum[blockIdx.x] = before + c;
}

// chunk sums -> exclusive bases (in place); total[0] = distinct keys
__global__ __launch_bounds__(512) void ks_chunk_scan_kernel(unsigned *__restrict__ chunk, unsigned *__restrict__ total)
{
    __shared__ unsigned wtot[8];
    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const unsigned c = chunk[t];
    unsigned x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(x, off, 64); if ((int)lane >= off) x += o; }
    if (lane == 63u) wtot[wave] = x;
    __syncthreads();
    unsigned before = x - c;
    for (unsigned k = 0; k < wave; k++) before += wtot[k];
    chunk[t] = before;
    if (t == 511u) total[0] = before + c;
}

// the grid indices of this rank's keys (ordinal % parts == part) at key_elem[ordinal / parts]
__global__ void ks_collect_kernel(const unsigned *__restrict__ bits, const unsigned *__restrict__ word_rank, const unsigned *__restrict__ chunk_base,
                                  unsigned part, unsigned parts, unsigned *__restrict__ key_elem)
{
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < KS_WORDS; w += gridDim.x * blockDim.x) {
        unsigned x = bits[w];
        if (!x) continue;
        unsigned o = chunk_base[w / KS_CHUNK] + word_rank[w];
        while (x) {
            if (o % parts == part) key_elem[o / parts] = w * 32u + (unsigned)(__ffs(x) - 1);
            x &= x - 1; o++;
        }
    }
}

__global__ void ks_pack_kernel(const unsigned *__restrict__ key_elem, unsigned n_own, const double *__restrict__ tables, double *__restrict__ vals)
{
    for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n_own; j += gridDim.x * blockDim.x)
        *reinterpret_cast<double2 *>(vals + 2 * (size_t)j) = *reinterpret_cast<const double2 *>(tables + 2 * (size_t)key_elem[j]);
}

// every key's (log p0 -> p0 if asked, E(K)) from the gathered arrays into the dense table
__global__ void ks_unpack_kernel(const unsigned *__restrict__ bits, const unsigned *__restrict__ word_rank, const unsigned *__restrict__ chunk_base,
                                 unsigned parts, size_t per, const double *__restrict__ vals_all, int exp_p0, double *__restrict__ tables)
{
    for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < KS_WORDS; w += gridDim.x * blockDim.x) {
        unsigned x = bits[w];
        if (!x) continue;
        unsigned o = chunk_base[w / KS_CHUNK] + word_rank[w];
        while (x) {
            const size_t idx = (size_t)w * 32u + (unsigned)(__ffs(x) - 1);
            double2 v = *reinterpret_cast<const double2 *>(vals_all + 2 * ((size_t)(o % parts) * per + o / parts));
            if (exp_p0) v.x = exp(v.x);
            *reinterpret_cast<double2 *>(tables + 2 * idx) = v;
            x &= x - 1; o++;
        }
    }
}

static int ks_number(const unsigned *keys, unsigned **word_rank, unsigned **chunk_base, unsigned *total_dev, hipStream_t stream)
{
    int rc;
    if ((rc = workspace_get(KsWs::RANK, (size_t)KS_WORDS * 4, reinterpret_cast<void **>(word_rank)))) return rc;
    if ((rc = workspace_get(KsWs::CHUNK, (size_t)KS_CHUNKS * 4, reinterpret_cast<void **>(chunk_base)))) return rc;
    hipLaunchKernelGGL(ks_rank_kernel, dim3(KS_CHUNKS), dim3(256), 0, stream, keys, *word_rank, *chunk_base);
    hipLaunchKernelGGL(ks_chunk_scan_kernel, dim3(1), dim3(512), 0, stream, *chunk_base, total_dev);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

struct KsInfo { unsigned long long nk, n_max, d_max, fits; };
static int ks_info_of(const uint64_t *info, KsInfo &k, const char *who)
{
    if (!info) { set_error(std::string(who) + ": NULL info"); return TRACS_E_ARG; }
    k.nk = info[0]; k.n_max = info[1]; k.d_max = info[2]; k.fits = info[3];
    if (!k.fits || (k.n_max + 1ull) * (k.d_max + 1ull) > TC_GRID_BITS || k.nk > TC_GRID_BITS) {
        set_error(std::string(who) + ": the keys do not fit the grid (tracs_trans_keys_info said so: take tracs_trans_dist_dense2)");
        return TRACS_E_ARG;
    }
    return TRACS_OK;
}

}  // namespace tracs

using namespace tracs;

extern "C" {

unsigned long long tracs_debug_last_trans_dist_keys(void) { return g_last_keys; }

size_t tracs_trans_keys_words(void) { return (size_t)KS_WORDS + KS_TRAILER; }

int tracs_trans_keys_mark(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges, size_t col_begin,
                          int32_t dist_threshold, const int32_t *days, uint32_t *keys, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!days || !keys || (n_ranges > 0 && (!dist || !row_ranges))) { set_error("tracs_trans_keys_mark: NULL argument"); return TRACS_E_ARG; }
    if (n_ranges < 0 || n_ranges > 2) { set_error("tracs_trans_keys_mark: 0, 1 or 2 row ranges"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    DenseSource src{dist, days, ld, n, 0, 0, col_begin, dist_threshold};
    if (n_ranges >= 1) { src.row_begin = std::min(row_ranges[0], n); src.row_end = std::min(row_ranges[1], n); }
    if (n_ranges == 2) { src.row_begin2 = std::min(row_ranges[2], n); src.row_end2 = std::min(row_ranges[3], n); }
    if (src.row_end < src.row_begin || src.row_end2 < src.row_begin2) { set_error("tracs_trans_keys_mark: bad range"); return TRACS_E_ARG; }
    unsigned *bits = keys, *cb = keys + KS_WORDS;
    const unsigned init[KS_TRAILER] = {0u, 0xFFFFFFFFu, 0u, 0u};
    TRACS_HIP_CHECK(hipMemsetAsync(bits, 0, (size_t)KS_WORDS * 4, stream));
    TRACS_HIP_CHECK(hipMemcpyAsync(cb, init, sizeof(init), hipMemcpyHostToDevice, stream));
    if (n == 0) return TRACS_OK;
    hipLaunchKernelGGL(tc_day_bounds_kernel, dim3(1), dim3(1024), 0, stream, days, n, cb);
    const dim3 row_grid = tc_row_grid(src);
    if (row_grid.x > 0)
        hipLaunchKernelGGL(tc_mark_kernel, dim3(row_grid.x, n <= 32768 ? 1u : row_grid.y), dim3(TC_ROW_THREADS), 0, stream, src, cb, bits);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_trans_keys_merge(uint32_t *keys, const uint32_t *all, int parts, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!keys || !all || parts < 1) { set_error("tracs_trans_keys_merge: bad argument"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    hipLaunchKernelGGL(ks_merge_kernel, dim3(1024), dim3(256), 0, stream, keys, all, parts);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_trans_keys_info(const uint32_t *keys, uint64_t *info, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!keys || !info) { set_error("tracs_trans_keys_info: NULL argument"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    unsigned *word_rank = nullptr, *chunk_base = nullptr, *total = nullptr;
    int rc;
    if ((rc = workspace_get(KsWs::INFO, 64, reinterpret_cast<void **>(&total)))) return rc;
    if ((rc = ks_number(keys, &word_rank, &chunk_base, total, stream))) return rc;
    unsigned h[1 + KS_TRAILER] = {0};
    TRACS_HIP_CHECK(hipMemcpyAsync(h, total, 4, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipMemcpyAsync(h + 1, keys + KS_WORDS, KS_TRAILER * 4, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    const bool days_ok = h[3] >= h[2];
    info[0] = h[0]; info[1] = h[1]; info[2] = days_ok ? h[3] - h[2] : 0u;
    info[3] = (days_ok && !h[4] && ((unsigned long long)h[1] + 1ull) * ((unsigned long long)(h[3] - h[2]) + 1ull) <= TC_GRID_BITS) ? 1u : 0u;
    return TRACS_OK;
}

int tracs_trans_keys_evaluate(const uint32_t *keys, const uint64_t *info, int part, int parts, double lamb, double beta,
                              double threshold_Ek, double *vals, size_t per, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KsInfo k;
    int rc;
    if (!keys) { set_error("tracs_trans_keys_evaluate: NULL argument"); return TRACS_E_ARG; }
    if ((rc = ks_info_of(info, k, "tracs_trans_keys_evaluate"))) return rc;
    if (parts < 1 || part < 0 || part >= parts) { set_error("tracs_trans_keys_evaluate: bad key partition"); return TRACS_E_ARG; }
    const unsigned n_own = k.nk > (unsigned long long)part ? (unsigned)((k.nk - part + parts - 1) / parts) : 0u;
    g_last_keys = n_own;
    if (n_own == 0) return TRACS_OK;
    if (!vals || per < n_own) { set_error("tracs_trans_keys_evaluate: vals holds fewer than ceil(keys / parts) slots"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    const double *lg = nullptr;
    if ((rc = get_lgamma_table(stream, &lg))) return rc;
    unsigned *word_rank = nullptr, *chunk_base = nullptr, *n_keys = nullptr, *key_elem = nullptr;
    double *tables = nullptr;
    if ((rc = workspace_get(TcWorkspaceIds::NKEYS, 64, reinterpret_cast<void **>(&n_keys)))) return rc;
    if ((rc = ks_number(keys, &word_rank, &chunk_base, n_keys + 12, stream))) return rc;
    const size_t cells = (size_t)(k.n_max + 1) * (size_t)(k.d_max + 1);
    if ((rc = workspace_get(TcWorkspaceIds::GRID_TABLES, cells * 16, reinterpret_cast<void **>(&tables)))) return rc;
    if ((rc = workspace_get(TcWorkspaceIds::KEY_ELEM, (size_t)n_own * 4, reinterpret_cast<void **>(&key_elem)))) return rc;
    hipLaunchKernelGGL(ks_collect_kernel, dim3(1024), dim3(256), 0, stream, keys, word_rank, chunk_base, (unsigned)part, (unsigned)parts, key_elem);
    const unsigned init[8] = {n_own, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // [0] keys, [1] long keys; [7] a key outside the table (cannot happen)
    TRACS_HIP_CHECK(hipMemcpyAsync(n_keys, init, sizeof(init), hipMemcpyHostToDevice, stream));
    KeyTable kt;
    kt.n_max = (unsigned)k.n_max; kt.d_max = (unsigned)k.d_max; kt.p0 = tables; kt.eK = tables + 1; kt.step = 2; kt.overflow = n_keys + 7;
    GridSource grid{(unsigned)k.d_max + 1u, cells};
    double *key_p0 = nullptr, *key_eK = nullptr;
    if ((rc = tc_evaluate_keys(grid, key_elem, n_own, n_keys, lamb, beta, threshold_Ek, lg, &key_p0, &key_eK, kt, stream))) return rc;
    hipLaunchKernelGGL(ks_pack_kernel, dim3(std::min(1024u, (n_own + 255u) / 256u)), dim3(256), 0, stream, key_elem, n_own, tables, vals);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_trans_keys_gather(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges, size_t col_begin,
                            int32_t dist_threshold, const int32_t *days, const uint32_t *keys, const uint64_t *info,
                            const double *vals_all, int parts, size_t per, int exp_p0, double *p0, double *eK, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KsInfo k;
    int rc;
    if (!keys || !days || (n_ranges > 0 && (!dist || !row_ranges || !p0 || !eK))) { set_error("tracs_trans_keys_gather: NULL argument"); return TRACS_E_ARG; }
    if ((rc = ks_info_of(info, k, "tracs_trans_keys_gather"))) return rc;
    if (n_ranges < 0 || n_ranges > 2 || parts < 1) { set_error("tracs_trans_keys_gather: bad argument"); return TRACS_E_ARG; }
    DenseSource src{dist, days, ld, n, 0, 0, col_begin, dist_threshold};
    if (n_ranges >= 1) { src.row_begin = std::min(row_ranges[0], n); src.row_end = std::min(row_ranges[1], n); }
    if (n_ranges == 2) { src.row_begin2 = std::min(row_ranges[2], n); src.row_end2 = std::min(row_ranges[3], n); }
    if (src.row_end < src.row_begin || src.row_end2 < src.row_begin2) { set_error("tracs_trans_keys_gather: bad range"); return TRACS_E_ARG; }
    const dim3 row_grid = tc_row_grid(src);
    if (row_grid.x == 0 || n == 0 || k.nk == 0) return TRACS_OK;
    if (!vals_all || per * (unsigned long long)parts < k.nk) { set_error("tracs_trans_keys_gather: vals_all holds fewer slots than keys"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    unsigned *word_rank = nullptr, *chunk_base = nullptr, *n_keys = nullptr;
    double *tables = nullptr;
    if ((rc = workspace_get(TcWorkspaceIds::NKEYS, 64, reinterpret_cast<void **>(&n_keys)))) return rc;
    if ((rc = ks_number(keys, &word_rank, &chunk_base, n_keys + 12, stream))) return rc;
    const size_t cells = (size_t)(k.n_max + 1) * (size_t)(k.d_max + 1);
    if ((rc = workspace_get(TcWorkspaceIds::GRID_TABLES, cells * 16, reinterpret_cast<void **>(&tables)))) return rc;
    hipLaunchKernelGGL(ks_unpack_kernel, dim3(1024), dim3(256), 0, stream, keys, word_rank, chunk_base, (unsigned)parts, per, vals_all, exp_p0, tables);
    KeyTable kt;
    kt.n_max = (unsigned)k.n_max; kt.d_max = (unsigned)k.d_max; kt.p0 = tables; kt.eK = tables + 1; kt.step = 2; kt.overflow = n_keys + 7;
    hipLaunchKernelGGL(tc_table_gather2_kernel, row_grid, dim3(TC_ROW_THREADS), 0, stream, src, kt, 0, p0, eK);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// More than 2^31 elements go in passes (each with its own key table): an unthresholded `tracs distance --meta` run with
// ~92 700 samples or more hands over > 4.29 x 10^9 pairs in one call (src/transcluster.hpp:263 simply loops).
int tracs_trans_dist_device(const int32_t *snpdiff, const double *datediff, size_t n, double lamb, double beta,
                            double threshold_Ek, int exp_p0, double *p0, double *eK, void *stream)
{
    if (n && (!snpdiff || !datediff || !p0 || !eK)) { set_error("tracs_trans_dist_device: NULL argument"); return TRACS_E_ARG; }
    g_last_keys = 0;
    for (size_t o = 0; o < n; o += TD_MAX_ELEMS) {
        const size_t cnt = std::min(TD_MAX_ELEMS, n - o);
        ArraySource src{snpdiff + o, datediff + o, cnt};
        const int rc = run_trans_dist(src, cnt, lamb, beta, threshold_Ek, exp_p0, p0 + o, eK + o, static_cast<hipStream_t>(stream));
        if (rc) return rc;
    }
    return TRACS_OK;
}

// row panels of a dense block in passes of at most TD_MAX_ELEMS cells
static int dense_passes(DenseSource src, double lamb, double beta, double thr, int exp_p0, double *p0, double *eK, hipStream_t stream)
{
    if (src.n == 0) return TRACS_OK;
    if (src.n >= TD_MAX_ELEMS) { set_error("trans_dist: more than 2^31 samples"); return TRACS_E_ARG; }
    const size_t rows_max = std::max<size_t>(1, TD_MAX_ELEMS / src.n);
    const size_t ranges[2][2] = {{src.row_begin, src.row_end}, {src.row_begin2, src.row_end2}};
    g_last_keys = 0;
    // both panels in one pass (one key table) when they fit; else panel by panel, chunk by chunk
    if ((ranges[0][1] - ranges[0][0]) + (ranges[1][1] - ranges[1][0]) <= rows_max) {
        const size_t total = ((ranges[0][1] - ranges[0][0]) + (ranges[1][1] - ranges[1][0])) * src.n;
        int done = 0;
        const int rc = run_trans_dist_grid(src, total, lamb, beta, thr, exp_p0, p0, eK, stream, &done);
        if (rc || done) return rc;
        return run_trans_dist(src, total, lamb, beta, thr, exp_p0, p0, eK, stream);
    }
    for (int k = 0; k < 2; k++)
        for (size_t r0 = ranges[k][0]; r0 < ranges[k][1]; r0 += rows_max) {
            DenseSource part = src;
            part.row_begin = r0; part.row_end = std::min(ranges[k][1], r0 + rows_max);
            part.row_begin2 = part.row_end2 = 0;
            const int rc = run_trans_dist(part, (part.row_end - part.row_begin) * src.n, lamb, beta, thr, exp_p0, p0, eK, stream);
            if (rc) return rc;
        }
    return TRACS_OK;
}

int tracs_trans_dist_dense(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                           int32_t dist_threshold, const int32_t *days, double lamb, double beta, double threshold_Ek,
                           int exp_p0, double *p0, double *eK, void *stream)
{
    if (!dist || !days || !p0 || !eK) { set_error("tracs_trans_dist_dense: NULL argument"); return TRACS_E_ARG; }
    if (row_end > n) row_end = n;
    if (row_begin >= row_end) return TRACS_OK;
    DenseSource src{dist, days, ld, n, row_begin, row_end, col_begin, dist_threshold};
    return dense_passes(src, lamb, beta, threshold_Ek, exp_p0, p0, eK, static_cast<hipStream_t>(stream));
}

int tracs_trans_dist_dense2(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges, size_t col_begin,
                            int32_t dist_threshold, const int32_t *days, double lamb, double beta, double threshold_Ek,
                            int exp_p0, double *p0, double *eK, void *stream)
{
    if (!dist || !days || !p0 || !eK || !row_ranges) { set_error("tracs_trans_dist_dense2: NULL argument"); return TRACS_E_ARG; }
    if (n_ranges < 1 || n_ranges > 2) { set_error("tracs_trans_dist_dense2: 1 or 2 row ranges"); return TRACS_E_ARG; }
    DenseSource src{dist, days, ld, n, std::min(row_ranges[0], n), std::min(row_ranges[1], n), col_begin, dist_threshold};
    if (n_ranges == 2) { src.row_begin2 = std::min(row_ranges[2], n); src.row_end2 = std::min(row_ranges[3], n); }
    if (src.row_end < src.row_begin || src.row_end2 < src.row_begin2) { set_error("tracs_trans_dist_dense2: bad range"); return TRACS_E_ARG; }
    return dense_passes(src, lamb, beta, threshold_Ek, exp_p0, p0, eK, static_cast<hipStream_t>(stream));
}

int tracs_trans_table_dense(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                            int32_t dist_threshold, const int32_t *days, double lamb, double beta, double threshold_Ek, int part,
                            int parts, uint32_t n_max, uint32_t d_max, double *table_p0, double *table_eK, uint32_t *overflow,
                            void *stream)
{
    if (!dist || !days || !table_p0 || !table_eK || !overflow) { set_error("tracs_trans_table_dense: NULL argument"); return TRACS_E_ARG; }
    if (parts < 1 || part < 0 || part >= parts) { set_error("tracs_trans_table_dense: bad key partition"); return TRACS_E_ARG; }
    if (row_end > n) row_end = n;
    if (row_begin >= row_end) return TRACS_OK;
    if ((row_end - row_begin) * n > TD_MAX_ELEMS) { set_error("tracs_trans_table_dense: more than 2^31 cells per call"); return TRACS_E_ARG; }
    DenseSource src{dist, days, ld, n, row_begin, row_end, col_begin, dist_threshold};
    KeyTable kt;
    kt.part = part; kt.parts = parts; kt.n_max = n_max; kt.d_max = d_max; kt.p0 = table_p0; kt.eK = table_eK; kt.overflow = overflow;
    g_last_keys = 0;
    return run_trans_dist(src, (row_end - row_begin) * n, lamb, beta, threshold_Ek, 0, nullptr, nullptr, static_cast<hipStream_t>(stream), kt);
}

int tracs_trans_table_gather(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                             int32_t dist_threshold, const int32_t *days, uint32_t n_max, uint32_t d_max, const double *table_p0,
                             const double *table_eK, int exp_p0, double *p0, double *eK, uint32_t *overflow, void *stream)
{
    if (!dist || !days || !table_p0 || !table_eK || !p0 || !eK || !overflow) { set_error("tracs_trans_table_gather: NULL argument"); return TRACS_E_ARG; }
    if (row_end > n) row_end = n;
    if (row_begin >= row_end) return TRACS_OK;
    DenseSource src{dist, days, ld, n, row_begin, row_end, col_begin, dist_threshold};
    KeyTable kt;
    kt.n_max = n_max; kt.d_max = d_max; kt.p0 = const_cast<double *>(table_p0); kt.eK = const_cast<double *>(table_eK); kt.overflow = overflow;
    hipLaunchKernelGGL(tc_table_gather2_kernel, tc_row_grid(src), dim3(TC_ROW_THREADS), 0, static_cast<hipStream_t>(stream), src, kt, exp_p0, p0, eK);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_trans_dist(const int32_t *snpdiff, const double *datediff, size_t n, double lamb, double beta,
                     double threshold_Ek, double *p0_log, double *eK)
{
    if (n == 0) return TRACS_OK;
    if (!snpdiff || !datediff || !p0_log || !eK) { set_error("tracs_trans_dist: NULL argument"); return TRACS_E_ARG; }
    int *dN = nullptr; double *dD = nullptr, *dP = nullptr, *dE = nullptr;
    auto cleanup = [&]() { if (dN) (void)hipFree(dN); if (dD) (void)hipFree(dD); if (dP) (void)hipFree(dP); if (dE) (void)hipFree(dE); };
#define TD_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
    TD_CHECK(hipMalloc(reinterpret_cast<void **>(&dN), n * 4));
    TD_CHECK(hipMalloc(reinterpret_cast<void **>(&dD), n * 8));
    TD_CHECK(hipMalloc(reinterpret_cast<void **>(&dP), n * 8));
    TD_CHECK(hipMalloc(reinterpret_cast<void **>(&dE), n * 8));
    TD_CHECK(hipMemcpy(dN, snpdiff, n * 4, hipMemcpyHostToDevice));
    TD_CHECK(hipMemcpy(dD, datediff, n * 8, hipMemcpyHostToDevice));
    int rc = tracs_trans_dist_device(dN, dD, n, lamb, beta, threshold_Ek, 0, dP, dE, nullptr);
    if (rc) { cleanup(); return rc; }
    TD_CHECK(hipMemcpy(p0_log, dP, n * 8, hipMemcpyDeviceToHost));
    TD_CHECK(hipMemcpy(eK, dE, n * 8, hipMemcpyDeviceToHost));
#undef TD_CHECK
    cleanup();
    return TRACS_OK;
}

int tracs_lprob_k_given_N(const uint64_t *N, const uint64_t *k, const double *delta, size_t n, double lamb, double beta,
                          const double *lgamma_tab, size_t lgamma_len, double *lprob, double *lhs)
{
    if (n == 0) return TRACS_OK;
    if (!N || !k || !delta || !lgamma_tab || !lprob || !lhs) { set_error("tracs_lprob_k_given_N: NULL argument"); return TRACS_E_ARG; }
    for (size_t i = 0; i < n; i++)
        if (N[i] + k[i] + 1 >= lgamma_len) { set_error("tracs_lprob_k_given_N: lgamma table shorter than N+k+2"); return TRACS_E_ARG; }
    unsigned long long *dN = nullptr, *dK = nullptr; double *dD = nullptr, *dL = nullptr, *dO = nullptr, *dH = nullptr;
    auto cleanup = [&]() { void *p[] = {dN, dK, dD, dL, dO, dH}; for (void *q : p) if (q) (void)hipFree(q); };
#define LP_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dN), n * 8));
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dK), n * 8));
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dD), n * 8));
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dL), lgamma_len * 8));
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dO), n * 8));
    LP_CHECK(hipMalloc(reinterpret_cast<void **>(&dH), n * 8));
    LP_CHECK(hipMemcpy(dN, N, n * 8, hipMemcpyHostToDevice));
    LP_CHECK(hipMemcpy(dK, k, n * 8, hipMemcpyHostToDevice));
    LP_CHECK(hipMemcpy(dD, delta, n * 8, hipMemcpyHostToDevice));
    LP_CHECK(hipMemcpy(dL, lgamma_tab, lgamma_len * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lprob_k_given_N_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, dN, dK, dD, n, lamb, beta, dL, dO, dH);
    LP_CHECK(hipGetLastError());
    LP_CHECK(hipMemcpy(lprob, dO, n * 8, hipMemcpyDeviceToHost));
    LP_CHECK(hipMemcpy(lhs, dH, n * 8, hipMemcpyDeviceToHost));
#undef LP_CHECK
    cleanup();
    return TRACS_OK;
}

}  // extern "C"
