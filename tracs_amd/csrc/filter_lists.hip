// filter_lists.hip -- recombination filter from the samples' departure lists (gfx950).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   filter_recomb :251-318, range_count :223-248, cached_binomial_cdf :41-58; called per emitted pair :405-413.
// The reference flips a pair's match set into its SNP set -- a scan of all L sites per pair.  Here every site gets a
// reference base once per alignment (the commonest one-base code among the first samples that have one) and every sample
// the sorted list of the sites at which it DEPARTS from it: neither N (all four alleles) nor exactly that base.  A site at
// which neither sample of a pair is listed cannot be a SNP (both carry the reference base, or one is N), so
//     SNP set of (i, j)  =  { s in D_i u D_j :  M_i(s) & M_j(s) = 0 }
// with M the allele mask; a partner that is not listed at s carries the reference base or is N -- one bit of an N bitmap:
//   NT  sample-major  (N plane transposed: the bits of sample i along the genome)        -> "is i N at a site of D_j"
//   NS  site-major    (per site one bit per sample: the N plane bit-transposed)          -> "is j N at a site of D_i"
// both chosen so that the lookups of the pairs of one row i stay inside ~1 MB (row i of NT; the |D_i| rows of NS).
// One wave per pair: both lists into LDS, every entry finds its rank in the other list by binary search (merge ranks: the
// union comes out sorted without a sort), survivors are closed up by a prefix count, and every SNP looks for the ends of
// its window [i - w, i + w + 1) a few entries to either side.  The binomial tail depends on (d, span, count) only and is
// monotone in the span: per distinct d one row of thresholds "smallest span that survives with `count` SNPs in the
// window" is built on first use (the reference memoises (n, p, k) -> cdf in a std::map, :41-58), so the test per SNP is
// one compare.  Pairs whose lists do not fit a wave's LDS (partial IUPAC codes: tens of thousands of entries per sample) take the
// same merge over global memory (flt_pairs_long_kernel); alignments whose lists cannot be built (more than a tenth of all cells
// listed, no memory) the scan of the planes (filter.hip: the reference's own way, which also returns the positions).
// PARITY UNPINNED: Boost's ibetac is replaced by the exact finite sum.
#include "common.h"
#include "filter_math.h"
#include "pairsnp_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

namespace tracs {

int get_lgamma_table_for_filter(hipStream_t stream, const double **out);   // transcluster.hip
int filter_scan_route(const tracs_alignment *a, const unsigned *rows, const unsigned *cols, const unsigned *d, size_t n_pairs, unsigned max_d,
                      unsigned *filt, const unsigned *tbl, const unsigned char *tbl_state, const double *lg, unsigned *bad,
                      hipStream_t stream);                                   // filter.hip

constexpr unsigned FLT_GCHUNK = 256;          // groups per (sample, chunk) thread of the list builders
constexpr unsigned FLT_CAP_MAX = 4096;        // list entries per sample a wave can hold in LDS
constexpr unsigned FLT_POS_BITS = 27;         // entry = pos << 5 | w << 4 | mask

typedef unsigned flt_u32x4 __attribute__((ext_vector_type(4)));
// a 16-byte load that does not stay in the caches (the planes stream by once)
__device__ __forceinline__ uint4 flt_nt_load(const uint4 *p)
{
    const flt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const flt_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct FilterIndex {
    unsigned *dep = nullptr;                  // departure entries, sample-major, sorted by site
    size_t dep_cap = 0;
    unsigned long long *dep_off = nullptr;    // n + 1
    uint4 *ref = nullptr;                     // per group: one-hot reference planes A, C, G, T
    uint4 *nt = nullptr;                      // [sample][nt_groups] N words
    unsigned *ns = nullptr;                   // [site][ns_words] sample bits
    size_t nt_groups = 0, ns_words = 0;
    unsigned *tbl = nullptr;                  // [FLT_DCAP + 1][FLT_KT] smallest surviving span
    unsigned char *tbl_state = nullptr;       // per d: bit 0 built, bit 1 wanted by the current call
    unsigned tbl_L = 0;
    unsigned *counters = nullptr;             // [0] pairs whose SNP count differs from d, [1] pairs left to the scan, [2] largest d among those
    unsigned long long total = 0;
    unsigned max_len = 0;
    size_t n = 0, L = 0;
    bool usable = false;                      // the lists exist (else: every pair takes the scan)
    double build_ms[6] = {0, 0, 0, 0, 0, 0};  // ref, count, offsets, fill, NT, NS
    double alloc_ms = 0.0;
    ~FilterIndex()
    {
        void *p[] = {dep, dep_off, ref, nt, ns, tbl, tbl_state, counters};
        for (void *q : p) if (q) (void)hipFree(q);
    }
};

// ---- reference base per site ------------------------------------------------------------------------------------------
// One wave per 128-site group; the lanes hold 64 samples at a time.  A site takes the commonest one-base code among the
// first block of 64 samples in which any sample has one (ties: A < C < G < T) -- with the first sample alone every private
// mutation of sample 0 would be listed by everybody else.
__global__ __launch_bounds__(64) void flt_ref_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, unsigned groups,
                                                     uint4 *__restrict__ ref)
{
    const unsigned g = blockIdx.x, lane = threadIdx.x;
    unsigned r[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};     // [base][word]
    unsigned open[4] = {~0u, ~0u, ~0u, ~0u};
    const uint4 *base = P + (size_t)g * NPLANES * n_pad;
    for (unsigned s0 = 0; s0 < n; s0 += 64) {
        const unsigned s = s0 + lane;
        uint4 z = make_uint4(0, 0, 0, 0), a = z, c = z, gq = z, t = z;
        if (s < n) { a = base[s]; c = base[n_pad + s]; gq = base[2 * n_pad + s]; t = base[3 * n_pad + s]; }
        const unsigned av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w}, tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned oa = av[w] & ~(cv[w] | gv[w] | tv[w]), oc = cv[w] & ~(av[w] | gv[w] | tv[w]);
            const unsigned og = gv[w] & ~(av[w] | cv[w] | tv[w]), ot = tv[w] & ~(av[w] | cv[w] | gv[w]);
            unsigned todo = 0;                                            // open sites of this word some lane resolves
            {
                unsigned any = (oa | oc | og | ot) & open[w];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off, 64);
                todo = __builtin_amdgcn_readfirstlane(any);
            }
            while (todo) {
                const int b = __ffs(todo) - 1;
                todo &= todo - 1;
                const unsigned ca = __popcll(__ballot((oa >> b) & 1u)), cc = __popcll(__ballot((oc >> b) & 1u));
                const unsigned cg = __popcll(__ballot((og >> b) & 1u)), ct = __popcll(__ballot((ot >> b) & 1u));
                int best = 0;
                unsigned top = ca;
                if (cc > top) { top = cc; best = 1; }
                if (cg > top) { top = cg; best = 2; }
                if (ct > top) { top = ct; best = 3; }
                const unsigned bm = 1u << b;
                r[0][w] |= best == 0 ? bm : 0u; r[1][w] |= best == 1 ? bm : 0u; r[2][w] |= best == 2 ? bm : 0u; r[3][w] |= best == 3 ? bm : 0u;
                open[w] &= ~bm;
            }
        }
        if (!(open[0] | open[1] | open[2] | open[3])) break;
    }
    if (lane < 4) {
        uint4 o = make_uint4(r[0][0], r[0][1], r[0][2], r[0][3]);
        if (lane == 1) o = make_uint4(r[1][0], r[1][1], r[1][2], r[1][3]);
        if (lane == 2) o = make_uint4(r[2][0], r[2][1], r[2][2], r[2][3]);
        if (lane == 3) o = make_uint4(r[3][0], r[3][1], r[3][2], r[3][3]);
        ref[(size_t)g * 4 + lane] = o;
    }
}

// ---- departure lists ----------------------------------------------------------------------------------------------------
// thread = (sample, chunk of FLT_GCHUNK groups): lanes over consecutive samples (1 KiB runs of every plane).  FILL = false
// counts, FILL = true writes the entries of its chunk behind those of the chunks before it.
__device__ __forceinline__ unsigned flt_dep_bits(unsigned a, unsigned c, unsigned g, unsigned t, unsigned ra, unsigned rc,
                                                 unsigned rg, unsigned rt)
{
    return ~(a & c & g & t) & ((a ^ ra) | (c ^ rc) | (g ^ rg) | (t ^ rt));
}

template <bool FILL>
__global__ __launch_bounds__(256) void flt_dep_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, unsigned groups,
                                                      const uint4 *__restrict__ ref, unsigned *__restrict__ cnt,
                                                      const unsigned *__restrict__ rel,
                                                      const unsigned long long *__restrict__ dep_off, unsigned *__restrict__ dep)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (s >= n) return;
    const unsigned g0 = ch * FLT_GCHUNK, g1 = min(groups, g0 + FLT_GCHUNK);
    unsigned total = 0;
    unsigned *dst = nullptr;
    if (FILL) dst = dep + dep_off[s] + rel[(size_t)ch * n + s];
    for (unsigned g = g0; g < g1; g++) {
        const uint4 *b = P + (size_t)g * NPLANES * n_pad + s;
        const uint4 a = flt_nt_load(b), c = flt_nt_load(b + n_pad);
        const uint4 gq = flt_nt_load(b + 2 * n_pad), t = flt_nt_load(b + 3 * n_pad);
        const uint4 ra = ref[(size_t)g * 4], rc = ref[(size_t)g * 4 + 1], rg = ref[(size_t)g * 4 + 2], rt = ref[(size_t)g * 4 + 3];
        const unsigned av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w}, tv[4] = {t.x, t.y, t.z, t.w};
        const unsigned rav[4] = {ra.x, ra.y, ra.z, ra.w}, rcv[4] = {rc.x, rc.y, rc.z, rc.w}, rgv[4] = {rg.x, rg.y, rg.z, rg.w}, rtv[4] = {rt.x, rt.y, rt.z, rt.w};
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned d = flt_dep_bits(av[w], cv[w], gv[w], tv[w], rav[w], rcv[w], rgv[w], rtv[w]);
            if (!FILL) { total += __popc(d); continue; }
            const unsigned has = (av[w] & rav[w]) | (cv[w] & rcv[w]) | (gv[w] & rgv[w]) | (tv[w] & rtv[w]);
            while (d) {
                const int bit = __ffs(d) - 1;
                d &= d - 1;
                const unsigned m = ((av[w] >> bit) & 1u) | (((cv[w] >> bit) & 1u) << 1) | (((gv[w] >> bit) & 1u) << 2) | (((tv[w] >> bit) & 1u) << 3);
                const unsigned wbit = ((has >> bit) & 1u) ^ 1u;            // the reference base is not among the sample's alleles
                *dst++ = ((g * 128u + (unsigned)w * 32u + (unsigned)bit) << 5) | (wbit << 4) | m;
            }
        }
    }
    if (!FILL) cnt[(size_t)ch * n + s] = total;
}

// per sample: offsets of its chunks' entries inside its list, and the list's length
__global__ __launch_bounds__(256) void flt_rel_kernel(const unsigned *__restrict__ cnt, unsigned n, unsigned chunks,
                                                      unsigned *__restrict__ rel, unsigned *__restrict__ tot)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    unsigned long long run = 0;
    for (unsigned c = 0; c < chunks; c++) {
        rel[(size_t)c * n + s] = (unsigned)run;
        run += cnt[(size_t)c * n + s];
    }
    tot[s] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)run;
}

// exclusive scan of the list lengths (one workgroup); out[n] = all entries, stats[0] = longest list
__global__ __launch_bounds__(1024) void flt_scan_kernel(const unsigned *__restrict__ tot, unsigned n,
                                                        unsigned long long *__restrict__ off, unsigned *__restrict__ stats)
{
    __shared__ unsigned long long part[1024];
    __shared__ unsigned mx[1024];
    const unsigned t = threadIdx.x, per = (n + 1023u) / 1024u;
    const unsigned b = min(n, t * per), e = min(n, b + per);
    unsigned long long sum = 0;
    unsigned m = 0;
    for (unsigned s = b; s < e; s++) { sum += tot[s]; m = max(m, tot[s]); }
    part[t] = sum; mx[t] = m;
    __syncthreads();
    for (unsigned st = 1; st < 1024; st <<= 1) {
        const unsigned long long v = t >= st ? part[t - st] : 0ull;
        const unsigned w = t >= st ? mx[t - st] : 0u;
        __syncthreads();
        part[t] += v; mx[t] = max(mx[t], w);
        __syncthreads();
    }
    unsigned long long run = part[t] - sum;
    for (unsigned s = b; s < e; s++) { off[s] = run; run += tot[s]; }
    if (t == 1023) { off[n] = part[1023]; stats[0] = mx[1023]; }
}

// ---- N bitmaps ----------------------------------------------------------------------------------------------------------------
// NT[sample][group] = the sample's N word of the group: a transposition of 16-byte elements through LDS
__global__ __launch_bounds__(256) void flt_nt_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, unsigned groups,
                                                     uint4 *__restrict__ NT, size_t nt_groups)
{
    __shared__ uint4 tile[32][65];
    const unsigned t = threadIdx.x, g0 = blockIdx.x * 32, s0 = blockIdx.y * 64;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const unsigned gr = (unsigned)p * 4 + (t >> 6), g = g0 + gr, s = s0 + (t & 63);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < groups && s < n) v = flt_nt_load(P + ((size_t)g * NPLANES + 4) * n_pad + s);
        tile[gr][t & 63] = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const unsigned sl = (unsigned)p * 8 + (t >> 5), s = s0 + sl;
        if (s < n) NT[(size_t)s * nt_groups + g0 + (t & 31)] = tile[t & 31][sl];
    }
}

// NS[site][sample word]: the N plane bit-transposed, one workgroup per (group, 1 024 samples): every wave transposes 64 samples x
// 128 sites in registers (four 32 x 32 blocks per half wave), the workgroup's 128 x 128-byte tile leaves as whole lines
__global__ __launch_bounds__(256) void flt_ns_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned groups,
                                                     unsigned *__restrict__ NS, size_t ns_words)
{
    __shared__ unsigned tile[128][33];
    const unsigned t = threadIdx.x, lane = t & 63, wv = t >> 6, half = lane >> 5, k = lane & 31;
    const unsigned g = blockIdx.x, sb = blockIdx.y;
    const Transpose32 tr(lane);
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const size_t s = (size_t)sb * 1024 + ((unsigned)it * 4 + wv) * 64 + lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (s < n_pad) v = flt_nt_load(P + ((size_t)g * NPLANES + 4) * n_pad + s);
        const unsigned wi = ((unsigned)it * 4 + wv) * 2 + half;
        tile[k][wi] = tr(v.x);
        tile[32 + k][wi] = tr(v.y);
        tile[64 + k][wi] = tr(v.z);
        tile[96 + k][wi] = tr(v.w);
    }
    __syncthreads();
    const unsigned site = t >> 1, hw = t & 1;
    unsigned *dst = NS + ((size_t)g * 128 + site) * ns_words + (size_t)sb * 32 + hw * 16;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const unsigned c0 = hw * 16 + (unsigned)q * 4;
        if ((size_t)sb * 32 + c0 < ns_words)
            *reinterpret_cast<uint4 *>(dst + q * 4) = make_uint4(tile[site][c0], tile[site][c0 + 1], tile[site][c0 + 2], tile[site][c0 + 3]);
    }
}

void launch_ns_build(const uint4 *planes, size_t n_pad, unsigned groups, unsigned *ns, size_t ns_words, hipStream_t stream)
{
    hipLaunchKernelGGL(flt_ns_kernel, dim3(groups, (unsigned)(ns_words / 32)), dim3(256), 0, stream, planes, n_pad, groups, ns, ns_words);
}

// ---- thresholds of the binomial test ------------------------------------------------------------------------------------------
// wanted rows: the d of this call's pairs
__global__ __launch_bounds__(256) void flt_mark_kernel(const unsigned *__restrict__ d, size_t n_pairs, unsigned char *__restrict__ state)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n_pairs; t += (size_t)gridDim.x * 256) {
        const unsigned v = d[t];
        if (v >= 2 && v <= FLT_DCAP && state[v] == 0) state[v] = 2;
    }
}
// tbl[d][k] = the smallest span n in (k, 2 w + 1] with 1 - BinomCDF(k; n, p) >= 0.05 / d -- the tail grows with n --, or
// 0xFFFFFFFF when no window of the pair can reach it.  One thread per (d, k) of the rows marked above.
__global__ __launch_bounds__(64) void flt_table_kernel(unsigned L, const double *__restrict__ lg, unsigned char *__restrict__ state,
                                                       unsigned *__restrict__ tbl)
{
    const unsigned d = blockIdx.x + 2, k = threadIdx.x;
    if (state[d] != 2) return;
    unsigned out = 0xFFFFFFFFu;
    if (k >= 2) {
        const FilterWindow fw = filter_window((long long)d, L);
        long long hi = 2ll * fw.wh + 1;
        if (hi > (long long)L) hi = (long long)L;
        long long lo = (long long)k + 1;
        if (lo <= hi && filter_keep(hi, k, fw.p, fw.thr, lg)) {
            while (lo < hi) {                                            // smallest n with keep(n)
                const long long mid = (lo + hi) >> 1;
                if (filter_keep(mid, k, fw.p, fw.thr, lg)) hi = mid; else lo = mid + 1;
            }
            out = (unsigned)lo;
        }
    }
    tbl[(size_t)d * FLT_KT + k] = out;
}
__global__ __launch_bounds__(256) void flt_table_done_kernel(unsigned char *__restrict__ state)
{
    const unsigned d = blockIdx.x * 256 + threadIdx.x;
    if (d <= FLT_DCAP && state[d] == 2) state[d] = 1;
}

// ---- the pairs ----------------------------------------------------------------------------------------------------------------
struct FltPairArgs {
    const unsigned *rows, *cols, *d;
    size_t n_pairs;
    const unsigned *dep;
    const unsigned long long *dep_off;
    const unsigned *nt, *ns;
    size_t nt_words, ns_words;
    unsigned L, cap;
    const unsigned *tbl;                  // nullptr: sum the tail per SNP
    const unsigned char *tbl_state;
    const double *lg;
    unsigned *filt, *counters;
};

// first index in [lo, hi) whose entry is >= key
template <class Ptr>
__device__ __forceinline__ int flt_lower_bound(Ptr a, int lo, int hi, unsigned key)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void flt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the window test over a pair's sorted SNP sites S[0 .. dn): how many survive (:273-314).  One wave; returns the wave's total.
template <class Ptr>
__device__ __forceinline__ unsigned flt_window_test(Ptr S, int dn, unsigned L, const unsigned *__restrict__ tbl_row,
                                                    const double *__restrict__ lg, int lane)
{
    const FilterWindow fw = filter_window(dn, L);
    const int wh = fw.wh, aln = (int)L;
    unsigned kept = 0;
    for (int u = lane; u < dn; u += 64) {
        const int x = (int)S[u];
        const int left = max(0, x - wh);                                   // :284
        const int right = min(aln, x + wh + 1);                            // :285
        int f = u, l = u, steps = 0;
        while (f > 0 && (int)S[f - 1] >= left) {
            --f;
            if (++steps == 6) { f = flt_lower_bound(S, 0, f, (unsigned)left); break; }
        }
        steps = 0;
        while (l + 1 < dn && (int)S[l + 1] < right) {
            ++l;
            if (++steps == 6) { l = flt_lower_bound(S, l + 1, dn, (unsigned)right) - 1; break; }
        }
        const int count = l - f + 1;
        if (count > 1) {                                                   // :294
            const int length = (int)S[l] - (int)S[f] + 1;                  // :242
            bool keep;
            if (tbl_row && count < (int)FLT_KT) keep = (unsigned)length >= tbl_row[count];
            else keep = filter_keep(length, count, fw.p, fw.thr, lg);
            kept += keep ? 1u : 0u;
        } else {
            kept++;                                                        // :311
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    return kept;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void flt_pairs_kernel(FltPairArgs A)
{
    extern __shared__ unsigned flt_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t t = (size_t)blockIdx.x * WPB + wv;
    if (t >= A.n_pairs) return;
    unsigned *LA = flt_lds + (size_t)wv * 4 * A.cap, *LB = LA + A.cap, *M = LB + A.cap;
    const unsigned i = A.rows[t], j = A.cols[t];
    const unsigned long long oi = A.dep_off[i], oj = A.dep_off[j];
    const unsigned la = (unsigned)(A.dep_off[i + 1] - oi), lb = (unsigned)(A.dep_off[j + 1] - oj);
    if (la > A.cap || lb > A.cap) {                                        // left to the scan of the planes
        if (lane == 0) { A.filt[t] = 0xFFFFFFFFu; atomicAdd(A.counters + 1, 1u); atomicMax(A.counters + 2, A.d[t]); }
        return;
    }
    for (unsigned k = lane; k < la; k += 64) LA[k] = A.dep[oi + k];
    for (unsigned k = lane; k < lb; k += 64) LB[k] = A.dep[oj + k];
    flt_wave_sync();
    // merge ranks + verdicts.  A site both list: the masks decide, the entry of i carries the verdict.  A site one lists: the other
    // carries the reference base (a SNP iff the listed sample's alleles lack it) or is N (never a SNP).
    const unsigned *nsj = A.ns + (j >> 5);
    const unsigned jbit = j & 31u;
    const unsigned *nti = A.nt + (size_t)i * A.nt_words;
    for (unsigned k = lane; k < la; k += 64) {
        const unsigned e = LA[k], pos = e >> 5;
        const int lo = flt_lower_bound(LB, 0, (int)lb, pos << 5);
        unsigned snp;
        if ((unsigned)lo < lb && (LB[lo] >> 5) == pos) snp = ((e & LB[lo] & 15u) == 0u) ? 1u : 0u;
        else if (e & 16u) snp = ((nsj[(size_t)pos * A.ns_words] >> jbit) & 1u) ^ 1u;
        else snp = 0u;
        M[k + (unsigned)lo] = pos | (snp << 31);
    }
    for (unsigned k = lane; k < lb; k += 64) {
        const unsigned e = LB[k], pos = e >> 5;
        const int lo = flt_lower_bound(LA, 0, (int)la, pos << 5);
        if ((unsigned)lo < la && (LA[lo] >> 5) == pos) { M[k + (unsigned)lo + 1u] = pos; continue; }
        unsigned snp = 0u;
        if (e & 16u) snp = ((nti[pos >> 5] >> (pos & 31u)) & 1u) ^ 1u;
        M[k + (unsigned)lo] = pos | (snp << 31);
    }
    flt_wave_sync();
    // close the survivors up (S takes the place of the two lists)
    unsigned *S = LA;
    const unsigned tot = la + lb;
    unsigned dn = 0;
    for (unsigned c0 = 0; c0 < tot; c0 += 64) {
        const unsigned c = c0 + lane;
        const unsigned v = c < tot ? M[c] : 0u;
        const unsigned long long b = __ballot(v >> 31);
        if (v >> 31) S[dn + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = v & 0x7FFFFFFFu;
        dn += (unsigned)__popcll(b);
    }
    flt_wave_sync();
    if (dn != A.d[t] && lane == 0) atomicAdd(A.counters, 1u);
    if (dn <= 1) { if (lane == 0) A.filt[t] = dn; return; }               // :259-261
    const unsigned *row = (A.tbl && dn <= FLT_DCAP && A.tbl_state[dn] == 1) ? A.tbl + (size_t)dn * FLT_KT : nullptr;
    const unsigned kept = flt_window_test(S, (int)dn, A.L, row, A.lg, lane);
    if (lane == 0) A.filt[t] = kept;
}

// The window test (:273-314) over a pair's sorted SNP sites S[0 .. dn), dn >= 2, with S[-2] = S[-1] = -1 and S[dn] = S[dn + 1] = INT_MAX
// around them: a SNP reads two neighbours either way up front (windows rarely hold more); the thresholds of the pair's d sit in
// registers, one count per lane (rowv; have_row: the row exists).  One wave; returns the wave's total.
template <class Ptr>
__device__ __forceinline__ unsigned flt_window_test2(Ptr S, const unsigned dn, const FltPairArgs &A, const unsigned rowv, const bool have_row,
                                                     const unsigned lane)
{
    const FilterWindow fw = filter_window((long long)dn, A.L);
    const int wh = fw.wh, aln = (int)A.L, n_s = (int)dn;
    unsigned kept = 0;
    for (int u0 = 0; u0 < n_s; u0 += 64) {
        const int u = u0 + (int)lane;
        const bool act = u < n_s;
        int count = 0, length = 0;
        if (act) {
            const int x = (int)S[u];
            const int p1 = (int)S[u - 1], p2 = (int)S[u - 2], n1 = (int)S[u + 1], n2 = (int)S[u + 2];
            const int left = max(0, x - wh);                               // :284
            const int right = min(aln, x + wh + 1);                        // :285
            int f = u, l = u, xf = x, xl = x;
            if (p1 >= left) {
                f = u - 1; xf = p1;
                if (p2 >= left) {
                    f = u - 2;
                    int steps = 0;
                    while ((int)S[f - 1] >= left) {
                        --f;
                        if (++steps == 6) { f = flt_lower_bound(S, 0, f, (unsigned)left); break; }
                    }
                    xf = (int)S[f];
                }
            }
            if (n1 < right) {
                l = u + 1; xl = n1;
                if (n2 < right) {
                    l = u + 2;
                    int steps = 0;
                    while ((int)S[l + 1] < right) {
                        ++l;
                        if (++steps == 6) { l = flt_lower_bound(S, l + 1, n_s, (unsigned)right) - 1; break; }
                    }
                    xl = (int)S[l];
                }
            }
            count = l - f + 1;
            length = xl - xf + 1;                                          // :242
        }
        const unsigned thr = __shfl(rowv, count < (int)FLT_KT ? count : 0, 64);
        if (act) {
            bool keep = true;                                              // alone in its window: :311
            if (count > 1) {                                               // :294
                if (have_row && count < (int)FLT_KT) keep = (unsigned)length >= thr;
                else keep = filter_keep(length, count, fw.p, fw.thr, A.lg);
            }
            kept += keep ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
    return kept;
}

// ---- the pairs, second form (lists of up to 1 024 entries) ----------------------------------------------------------------------
// One wave per pair, R = rounds of 64 entries per list.  What the first form spends its time on is latency: ten dependent LDS reads
// per entry for the merge ranks, and one N lookup per entry in global memory whose answer the next instruction waits for.  Here
//   * both lists are loaded into registers and ALL their N lookups are issued before any answer is used (2 R loads in flight per
//     lane); the answer folds into the entry's w bit (a listed sample whose partner is N at the site can never make a SNP there),
//     so nothing later touches global memory;
//   * the merge is a merge path: lane l owns the merged entries [l C, (l + 1) C), finds where its share starts in the two lists
//     with ONE binary search along its diagonal and then takes C entries one after the other (one LDS read per entry);
//   * the lane's survivors are counted on the way, a prefix sum over the lanes places them, each lane copies its own;
//   * the window test reads a SNP's two neighbours on either side up front (windows rarely hold more), and the thresholds of the
//     pair's d sit in registers, one count per lane (ds_bpermute instead of a table read).
//   * a wave takes FLT_PB consecutive pairs: in a row-major emission they share the row i and, 31 times out of 32, the 32-sample word
//     of j -- the list of i and its NS words stay in registers from one pair to the next (half of a pair's scattered lookups).
#ifndef TRACS_FLT_PB
#define TRACS_FLT_PB 8
#endif
#ifndef TRACS_FLT_REGS
#define TRACS_FLT_REGS 1
#endif
constexpr unsigned FLT_PB = TRACS_FLT_PB;
template <int R>
struct FltRowCache {
    unsigned i = 0xFFFFFFFFu, jw = 0xFFFFFFFFu, la = 0;
    unsigned ea[R], na[R];
};

template <int R>
__device__ __forceinline__ void flt_pair2(const FltPairArgs &A, const size_t t, const unsigned lane, unsigned *LA, unsigned *LB, unsigned *M,
                                          FltRowCache<R> &rc)
{
    constexpr unsigned CAP = R * 64, INF = 0xFFFFFFFFu;
    const unsigned i = A.rows[t], j = A.cols[t];
    const unsigned long long oi = A.dep_off[i], oj = A.dep_off[j];
    const unsigned la = (unsigned)(A.dep_off[i + 1] - oi), lb = (unsigned)(A.dep_off[j + 1] - oj);
    if (la > CAP || lb > CAP) {                                            // left to the scan of the planes
        if (lane == 0) { A.filt[t] = 0xFFFFFFFFu; atomicAdd(A.counters + 1, 1u); atomicMax(A.counters + 2, A.d[t]); }
        return;
    }
    const unsigned dt = A.d[t];
    const bool have_row = A.tbl && dt >= 2 && dt <= FLT_DCAP && A.tbl_state[dt] == 1;
    const unsigned rowv = have_row ? A.tbl[(size_t)dt * FLT_KT + lane] : INF;      // smallest surviving span for count = lane
    const bool same_i = rc.i == i, same_w = same_i && rc.jw == (j >> 5);      // (wave-uniform)
    const unsigned lmax = max(la, lb);                                     // (rounds beyond both lists are skipped: wave-uniform)
    unsigned eb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned k = (unsigned)r * 64 + lane;
        eb[r] = INF;
        if ((unsigned)r * 64 >= lmax) continue;
        if (!same_i) rc.ea[r] = k < la ? A.dep[oi + k] : INF;
        eb[r] = k < lb ? A.dep[oj + k] : INF;
    }
    const unsigned *nsj = A.ns + (j >> 5);
    const unsigned jbit = j & 31u;
    const unsigned *nti = A.nt + (size_t)i * A.nt_words;
    unsigned nb[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned k = (unsigned)r * 64 + lane;
        nb[r] = 0u;
        if ((unsigned)r * 64 >= lmax) continue;
        if (!same_w) rc.na[r] = (k < la && (rc.ea[r] & 16u)) ? nsj[(size_t)(rc.ea[r] >> 5) * A.ns_words] : 0u;
        nb[r] = (k < lb && (eb[r] & 16u)) ? nti[eb[r] >> 10] : 0u;
    }
    rc.i = i; rc.jw = j >> 5; rc.la = la;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned k = (unsigned)r * 64 + lane;
        if ((unsigned)r * 64 >= lmax) continue;
        if (k < la) LA[k] = rc.ea[r] & ~(((rc.na[r] >> jbit) & 1u) << 4);
        if (k < lb) LB[k] = eb[r] & ~(((nb[r] >> ((eb[r] >> 5) & 31u)) & 1u) << 4);
    }
    flt_wave_sync();
#if defined(TRACS_FLT_CUT) && TRACS_FLT_CUT == 1
    if (lane == 0) A.filt[t] = LA[0] + LB[0];
    return;
#endif
    // merge path
    const unsigned tot = la + lb, C = (tot + 63u) / 64u, stride = C | 1u;
    const unsigned D = min(tot, lane * C);
    unsigned a, b;
    {
        unsigned lo = D > lb ? D - lb : 0u, hi = min(D, la);
        while (lo < hi) {
            const unsigned mid = (lo + hi) >> 1;
            if ((LA[mid] >> 5) <= (LB[D - 1u - mid] >> 5)) lo = mid + 1u; else hi = mid;
        }
        a = lo; b = D - lo;
    }
    unsigned va = a < la ? LA[a] : INF, vb = b < lb ? LB[b] : INF;
    unsigned lastA = a > 0u ? (LA[a - 1u] >> 5) : INF;
    unsigned cnt = 0;
#if TRACS_FLT_REGS
    // a lane's merged entries stay in its registers (at most 2 R of them: the step loop is unrolled, rounds beyond C skipped wave-
    // uniformly): no M array -- half the LDS per wave, and with it more waves per CU to hide the pair's memory round trips behind
    unsigned sv[2 * R];
    (void)M; (void)stride;
#pragma unroll
    for (int step = 0; step < 2 * R; step++) {
        sv[step] = 0u;
        if ((unsigned)step >= C) continue;
#else
    unsigned *Mrow = M + lane * stride;
    for (unsigned step = 0; step < C; step++) {
#endif
        const unsigned pa = va >> 5, pb = vb >> 5;
        const bool takeA = pa <= pb;
        // an entry of i: both listed -> the masks decide; else its w bit (partner carries the reference base; N folded in above).
        // an entry of j: dead when i lists the site too (i's entry carried the verdict)
        const unsigned snpA = pa == pb ? (((va & vb & 15u) == 0u) ? 1u : 0u) : ((va >> 4) & 1u);
        const unsigned snpB = lastA == pb ? 0u : ((vb >> 4) & 1u);
        const unsigned snp = (D + (unsigned)step < tot) ? (takeA ? snpA : snpB) : 0u;
#if TRACS_FLT_REGS
        sv[step] = (takeA ? pa : pb) | (snp << 31);
#else
        Mrow[step] = (takeA ? pa : pb) | (snp << 31);
#endif
        cnt += snp;
        const unsigned nidx = takeA ? a + 1u : CAP + b + 1u, nlim = takeA ? la : CAP + lb;
        const unsigned nxt = nidx < nlim ? LA[nidx] : INF;
        if (takeA) { lastA = pa; a++; va = nxt; } else { b++; vb = nxt; }
    }
    // place the survivors
    unsigned incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off, 64);
        if (lane >= (unsigned)off) incl += v;
    }
    const unsigned dn = __shfl(incl, 63, 64);
#if defined(TRACS_FLT_CUT) && TRACS_FLT_CUT == 2
    if (lane == 0) A.filt[t] = dn;
    return;
#endif
    flt_wave_sync();                                                       // every lane is done with the two lists
    // S[-2], S[-1] = -1 and S[dn], S[dn + 1] = INT_MAX: the window test reads two neighbours either way without asking where the list ends
    unsigned *S = LA + 2;
    {
        unsigned o = incl - cnt;
#if TRACS_FLT_REGS
#pragma unroll
        for (int step = 0; step < 2 * R; step++)
            if (sv[step] >> 31) S[o++] = sv[step] & 0x7FFFFFFFu;
#else
        for (unsigned step = 0; step < C; step++) {
            const unsigned v = Mrow[step];
            if (v >> 31) S[o++] = v & 0x7FFFFFFFu;
        }
#endif
        if (lane < 2) { LA[lane] = 0xFFFFFFFFu; S[dn + lane] = 0x7FFFFFFFu; }
    }
    flt_wave_sync();
#if defined(TRACS_FLT_CUT) && TRACS_FLT_CUT == 3
    if (lane == 0) A.filt[t] = dn + S[0];
    return;
#endif
    if (dn != dt && lane == 0) atomicAdd(A.counters, 1u);
    if (dn <= 1u) { if (lane == 0) A.filt[t] = dn; return; }              // :259-261
    const unsigned kept = flt_window_test2(S, dn, A, rowv, have_row, lane);
    if (lane == 0) A.filt[t] = kept;
}

// (four waves per SIMD: 238.9 ms for the 49 995 000 pairs of the bench alignment where the compiler's own choice -- 149 VGPRs, three
// waves -- takes 292.5; five: 249.9 (spills), six: 314.6; with the merged entries in LDS instead of registers and four waves: 269.1 --
// the kernel waits on memory, waves are what hides it: profiles/r06/filter_phase_cuts.txt)
#ifndef TRACS_FLT_WAVES
#define TRACS_FLT_WAVES 4
#endif
#if TRACS_FLT_WAVES > 0
#define TRACS_FLT_ATTR __attribute__((amdgpu_waves_per_eu(TRACS_FLT_WAVES, TRACS_FLT_WAVES)))
#else
#define TRACS_FLT_ATTR
#endif
template <int R>
__global__ __launch_bounds__(64) TRACS_FLT_ATTR void flt_pairs2_kernel(FltPairArgs A)
{
    extern __shared__ unsigned flt_lds[];
    constexpr unsigned CAP = R * 64;
    const unsigned lane = threadIdx.x;
    unsigned *LA = flt_lds, *LB = LA + CAP, *M = LB + CAP;                 // M: 2 CAP + 128 words
    FltRowCache<R> rc;
    const size_t t0 = (size_t)blockIdx.x * FLT_PB, t1 = min(A.n_pairs, t0 + FLT_PB);
    for (size_t t = t0; t < t1; t++) {
        flt_pair2<R>(A, t, lane, LA, LB, M, rc);
        flt_wave_sync();                                                   // (the next pair's lists take the place of this one's SNP sites)
    }
}

template <int R>
static void flt_launch2(const FltPairArgs &A, hipStream_t stream)
{
    const size_t lds_words = TRACS_FLT_REGS ? (size_t)2 * R * 64 + 8 : (size_t)4 * R * 64 + 128;
    hipLaunchKernelGGL((flt_pairs2_kernel<R>), dim3((unsigned)((A.n_pairs + FLT_PB - 1) / FLT_PB)), dim3(64), lds_words * 4, stream, A);
}

// ---- the pairs whose lists do not fit a wave's LDS (partial IUPAC codes: tens of thousands of entries per sample) ------------------
// The same merge path with the lists where they lie and the merged sequence in the wave's slot of a scratch buffer in global memory:
// lane l takes the merged entries [l C, (l + 1) C) one after the other -- its two streams advance four bytes a step, a cache line
// serves sixteen steps -- and writes  site | candidate << 31 | "ask NS for j" << 30 | "ask NT for i" << 29;  a second pass over
// the sequence, 64 entries per instruction, asks the N bitmaps (independent loads: nothing waits for an answer inside the lanes'
// sequential merge) and closes the survivors up IN PLACE (a survivor never lands ahead of what is still to be read); then the window
// test reads the SNP sites from there.  One wave per pair of idx[0 .. n_idx), as many waves as the scratch buffer has slots.
__global__ __launch_bounds__(64) void flt_pairs_long_kernel(FltPairArgs A, const unsigned *__restrict__ idx, size_t n_idx,
                                                            unsigned *__restrict__ scratch, size_t slot)
{
    constexpr unsigned INF = 0xFFFFFFFFu;
    const unsigned lane = threadIdx.x;
    unsigned *M = scratch + (size_t)blockIdx.x * slot + 2;                 // (two sentinels in front, two behind)
    for (size_t q = blockIdx.x; q < n_idx; q += gridDim.x) {
        const size_t t = idx[q];
        const unsigned i = A.rows[t], j = A.cols[t];
        const unsigned long long oi = A.dep_off[i], oj = A.dep_off[j];
        const unsigned la = (unsigned)(A.dep_off[i + 1] - oi), lb = (unsigned)(A.dep_off[j + 1] - oj);
        const unsigned *DA = A.dep + oi, *DB = A.dep + oj;
        const unsigned dt = A.d[t];
        const bool have_row = A.tbl && dt >= 2 && dt <= FLT_DCAP && A.tbl_state[dt] == 1;
        const unsigned rowv = have_row ? A.tbl[(size_t)dt * FLT_KT + lane] : INF;
        const unsigned tot = la + lb, C = (tot + 63u) / 64u;
        const unsigned D = min(tot, lane * C);
        unsigned a, b;
        {
            unsigned lo = D > lb ? D - lb : 0u, hi = min(D, la);
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if ((DA[mid] >> 5) <= (DB[D - 1u - mid] >> 5)) lo = mid + 1u; else hi = mid;
            }
            a = lo; b = D - lo;
        }
        unsigned va = a < la ? DA[a] : INF, vb = b < lb ? DB[b] : INF;
        unsigned lastA = a > 0u ? (DA[a - 1u] >> 5) : INF;
        for (unsigned step = 0; step < C; step++) {
            const unsigned pa = va >> 5, pb = vb >> 5;
            const bool takeA = pa <= pb;
            // i's entry: both listed -> the masks decide (final); else a candidate iff its alleles lack the reference base, unless j is N
            // there (NS).  j's entry: dead when i lists the site too; else the same with i's N bit (NT).
            const unsigned wa = (va >> 4) & 1u, wb = (vb >> 4) & 1u;
            const unsigned fa = pa == pb ? ((((va & vb & 15u) == 0u) ? 1u : 0u) << 31) : ((wa << 31) | (wa << 30));
            const unsigned fb = lastA == pb ? 0u : ((wb << 31) | (wb << 29));
            if (D + step < tot) M[D + step] = (takeA ? pa : pb) | (takeA ? fa : fb);
            if (takeA) { lastA = pa; a++; va = a < la ? DA[a] : INF; } else { b++; vb = b < lb ? DB[b] : INF; }
        }
        __threadfence_block();
        flt_wave_sync();
        const unsigned *nsj = A.ns + (j >> 5);
        const unsigned jbit = j & 31u;
        const unsigned *nti = A.nt + (size_t)i * A.nt_words;
        unsigned dn = 0;
        for (unsigned c0 = 0; c0 < tot; c0 += 256) {
            unsigned v[4], nw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const unsigned c = c0 + (unsigned)u * 64 + lane; v[u] = c < tot ? M[c] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned pos = v[u] & 0x07FFFFFFu;
                nw[u] = 0u;
                if (v[u] & (1u << 30)) nw[u] = (nsj[(size_t)pos * A.ns_words] >> jbit) & 1u;
                else if (v[u] & (1u << 29)) nw[u] = (nti[pos >> 5] >> (pos & 31u)) & 1u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool alive = (v[u] >> 31) && !nw[u];
                const unsigned long long bal = __ballot(alive);
                if (alive) M[dn + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = v[u] & 0x07FFFFFFu;
                dn += (unsigned)__popcll(bal);
            }
        }
        if (lane < 2) { M[(int)lane - 2] = 0xFFFFFFFFu; M[dn + lane] = 0x7FFFFFFFu; }
        __threadfence_block();
        flt_wave_sync();
        if (dn != dt && lane == 0) atomicAdd(A.counters, 1u);
        if (dn <= 1u) { if (lane == 0) A.filt[t] = dn; }                  // :259-261
        else {
            const unsigned kept = flt_window_test2(M, dn, A, rowv, have_row, lane);
            if (lane == 0) A.filt[t] = kept;
        }
        flt_wave_sync();
    }
}

// ---- long lists, tiled -------------------------------------------------------------------------------------------------------------
// The sequential merge above waits for a cache line every few steps.  Here the merged sequence is cut into TILES of FLT_TILE entries
// by merge-path diagonals (lane k finds where tile k starts in the two lists: one binary search over global memory per tile, all
// tiles of a round at once), and every tile is the short-list kernel's work: its share of either list loaded into LDS with ALL its
// N lookups in flight at once, a merge path in LDS, the lanes' survivors appended to the pair's SNP sites in the wave's scratch slot.
// (A site both samples list may straddle a tile boundary -- i's entry last in one tile, j's first in the next: i's entry looks at the
// entry of j behind the tile's share, j's entry at the site of i's entry before it.)
// (512 entries a tile: 24.4 s for the 49 995 000 pairs of the partial-code alignment, 27.3 with 256, 39.5 with 1 024, 45.2 with 2 048 --
// 9.5 KB of LDS and half the registers per wave: sixteen waves per CU hide the tile's memory round trips, profiles/r06/filter_tile_sweep.txt)
#ifndef TRACS_FLT_TILE
#define TRACS_FLT_TILE 512
#endif
constexpr unsigned FLT_TILE = TRACS_FLT_TILE, FLT_TILE_R = FLT_TILE / 64;
// (a tile's merged entries in registers -- 4.9 KB of LDS per wave --, six waves per SIMD and 24 per CU in the grid: 22.5 s where the
// M array in LDS and sixteen waves per CU take 24.0: profiles/r06/filter_tile_sweep.txt)
#ifndef TRACS_FLT_TWAVES
#define TRACS_FLT_TWAVES 6
#endif
#ifndef TRACS_FLT_TREGS
#define TRACS_FLT_TREGS 1
#endif
#ifndef TRACS_FLT_TGRID
#define TRACS_FLT_TGRID 24
#endif
#if TRACS_FLT_TWAVES > 0
#define TRACS_FLT_TATTR __attribute__((amdgpu_waves_per_eu(TRACS_FLT_TWAVES, TRACS_FLT_TWAVES)))
#else
#define TRACS_FLT_TATTR
#endif
__global__ __launch_bounds__(64) TRACS_FLT_TATTR void flt_pairs_tiled_kernel(FltPairArgs A, const unsigned *__restrict__ idx, size_t n_idx,
                                                             unsigned *__restrict__ scratch, size_t slot)
{
    extern __shared__ unsigned flt_lds[];
    constexpr unsigned CAP = FLT_TILE + 64, INF = 0xFFFFFFFFu, R = FLT_TILE_R;
    const unsigned lane = threadIdx.x;
#if TRACS_FLT_TREGS
    unsigned *LA = flt_lds, *LB = LA + CAP, *M = nullptr, *split = LB + CAP;                     // split: 65 words
#else
    unsigned *LA = flt_lds, *LB = LA + CAP, *M = LB + CAP, *split = M + 2 * FLT_TILE + 128;      // split: 65 words
#endif
    unsigned *S = scratch + (size_t)blockIdx.x * slot + 2;                 // (two sentinels in front, two behind)
    for (size_t q = blockIdx.x; q < n_idx; q += gridDim.x) {
        const size_t t = idx[q];
        const unsigned i = A.rows[t], j = A.cols[t];
        const unsigned long long oi = A.dep_off[i], oj = A.dep_off[j];
        const unsigned la = (unsigned)(A.dep_off[i + 1] - oi), lb = (unsigned)(A.dep_off[j + 1] - oj);
        const unsigned *DA = A.dep + oi, *DB = A.dep + oj;
        const unsigned dt = A.d[t];
        const bool have_row = A.tbl && dt >= 2 && dt <= FLT_DCAP && A.tbl_state[dt] == 1;
        const unsigned rowv = have_row ? A.tbl[(size_t)dt * FLT_KT + lane] : INF;
        const unsigned *nsj = A.ns + (j >> 5);
        const unsigned jbit = j & 31u;
        const unsigned *nti = A.nt + (size_t)i * A.nt_words;
        const unsigned tot = la + lb, tiles = (tot + FLT_TILE - 1) / FLT_TILE;
        unsigned dn = 0;
        for (unsigned k0 = 0; k0 < tiles; k0 += 64) {
            // where the tiles k0 .. k0 + 63 (and the one behind them) start in i's list: split[k]; in j's: k FLT_TILE - split[k]
            for (unsigned kk = lane; kk < 65u; kk += 64) {
                const unsigned D = min(tot, (k0 + kk) * FLT_TILE);
                unsigned lo = D > lb ? D - lb : 0u, hi = min(D, la);
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if ((DA[mid] >> 5) <= (DB[D - 1u - mid] >> 5)) lo = mid + 1u; else hi = mid;
                }
                split[kk] = lo;
            }
            flt_wave_sync();
            for (unsigned kt = 0; kt < 64u && k0 + kt < tiles; kt++) {
                const unsigned D0 = (k0 + kt) * FLT_TILE, D1 = min(tot, D0 + FLT_TILE);
                const unsigned a0 = split[kt], a1 = split[kt + 1], b0 = D0 - a0, b1 = D1 - a1;
                const unsigned la_t = a1 - a0, lb_t = b1 - b0, tot_t = la_t + lb_t;
                // the tile's entries (+ the entry of j behind its share: what a straddling twin of i's last entry would be), N lookups folded in
                unsigned ea[R], eb[R], na[R], nb[R];
#pragma unroll
                for (unsigned r = 0; r < R; r++) {
                    const unsigned k = r * 64 + lane;
                    ea[r] = k < la_t ? DA[a0 + k] : INF;
                    eb[r] = k < lb_t ? DB[b0 + k] : INF;
                }
                const unsigned peek = b1 < lb ? DB[b1] : INF;
                const unsigned prevA = a0 > 0u ? (DA[a0 - 1u] >> 5) : INF;
#pragma unroll
                for (unsigned r = 0; r < R; r++) {
                    const unsigned k = r * 64 + lane;
                    na[r] = (k < la_t && (ea[r] & 16u)) ? nsj[(size_t)(ea[r] >> 5) * A.ns_words] : 0u;
                    nb[r] = (k < lb_t && (eb[r] & 16u)) ? nti[eb[r] >> 10] : 0u;
                }
#pragma unroll
                for (unsigned r = 0; r < R; r++) {
                    const unsigned k = r * 64 + lane;
                    if (k < la_t) LA[k] = ea[r] & ~(((na[r] >> jbit) & 1u) << 4);
                    if (k < lb_t) LB[k] = eb[r] & ~(((nb[r] >> ((eb[r] >> 5) & 31u)) & 1u) << 4);
                }
                if (lane == 0) LB[lb_t] = peek;
                flt_wave_sync();
                // merge path inside the tile (flt_pair2's)
                const unsigned C = (tot_t + 63u) / 64u, stride = C | 1u;
                const unsigned D = min(tot_t, lane * C);
                unsigned a, b;
                {
                    unsigned lo = D > lb_t ? D - lb_t : 0u, hi = min(D, la_t);
                    while (lo < hi) {
                        const unsigned mid = (lo + hi) >> 1;
                        if ((LA[mid] >> 5) <= (LB[D - 1u - mid] >> 5)) lo = mid + 1u; else hi = mid;
                    }
                    a = lo; b = D - lo;
                }
                unsigned va = a < la_t ? LA[a] : INF, vb = b <= lb_t ? LB[b] : INF;
                unsigned lastA = a > 0u ? (LA[a - 1u] >> 5) : prevA;
                unsigned cnt = 0;
#if TRACS_FLT_TREGS
                unsigned sv[R];                                            // (C <= R: a tile's merged entries, FLT_TILE / 64 per lane)
                (void)stride;
#pragma unroll
                for (unsigned step = 0; step < R; step++) {
                    sv[step] = 0u;
                    if (step >= C) continue;
#else
                unsigned *Mrow = M + lane * stride;
                for (unsigned step = 0; step < C; step++) {
#endif
                    const unsigned pa = va >> 5, pb = vb >> 5;
                    const bool takeA = pa <= pb;
                    const unsigned snpA = pa == pb ? (((va & vb & 15u) == 0u) ? 1u : 0u) : ((va >> 4) & 1u);
                    const unsigned snpB = lastA == pb ? 0u : ((vb >> 4) & 1u);
                    const unsigned snp = (D + step < tot_t) ? (takeA ? snpA : snpB) : 0u;
#if TRACS_FLT_TREGS
                    sv[step] = (takeA ? pa : pb) | (snp << 31);
#else
                    Mrow[step] = (takeA ? pa : pb) | (snp << 31);
#endif
                    cnt += snp;
                    const unsigned nidx = takeA ? a + 1u : CAP + b + 1u, nlim = takeA ? la_t : CAP + lb_t + 1u;
                    const unsigned nxt = nidx < nlim ? LA[nidx] : INF;
                    if (takeA) { lastA = pa; a++; va = nxt; } else { b++; vb = nxt; }
                }
                unsigned incl = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned v = __shfl_up(incl, off, 64);
                    if (lane >= (unsigned)off) incl += v;
                }
                {
                    unsigned o = dn + incl - cnt;
#if TRACS_FLT_TREGS
#pragma unroll
                    for (unsigned step = 0; step < R; step++)
                        if (sv[step] >> 31) S[o++] = sv[step] & 0x7FFFFFFFu;
#else
                    for (unsigned step = 0; step < C; step++) {
                        const unsigned v = Mrow[step];
                        if (v >> 31) S[o++] = v & 0x7FFFFFFFu;
                    }
#endif
                }
                dn += __shfl(incl, 63, 64);
                flt_wave_sync();                                           // (the next tile's entries take the place of this one's)
            }
        }
        if (lane < 2) { S[(int)lane - 2] = 0xFFFFFFFFu; S[dn + lane] = 0x7FFFFFFFu; }
        __threadfence_block();
        flt_wave_sync();
        if (dn != dt && lane == 0) atomicAdd(A.counters, 1u);
        if (dn <= 1u) { if (lane == 0) A.filt[t] = dn; }                  // :259-261
        else {
            const unsigned kept = flt_window_test2(S, dn, A, rowv, have_row, lane);
            if (lane == 0) A.filt[t] = kept;
        }
        flt_wave_sync();
    }
}

// pairs whose lists did not fit the LDS kernel: their indices, closed up
__global__ __launch_bounds__(256) void flt_collect_kernel(const unsigned *__restrict__ filt, size_t n_pairs, unsigned *__restrict__ idx,
                                                          unsigned *__restrict__ cursor)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n_pairs; t += (size_t)gridDim.x * 256)
        if (filt[t] == 0xFFFFFFFFu) idx[atomicAdd(cursor, 1u)] = (unsigned)t;
}
__global__ __launch_bounds__(256) void flt_max_kernel(const unsigned *__restrict__ d, size_t n, unsigned *__restrict__ out)
{
    unsigned m = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) m = max(m, d[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
struct FltWs { enum { CNT = 70, REL, TOT, STATS, IDX, SCRATCH }; };

static bool flt_env_off(const char *name)
{
    const char *e = std::getenv(name);
    return e && e[0] == '0';
}

template <class T>
static bool flt_alloc(T **p, size_t bytes)
{
    if (*p) return true;
    if (hipMalloc(reinterpret_cast<void **>(p), bytes) == hipSuccess) return true;
    *p = nullptr;
    (void)hipGetLastError();
    return false;
}

struct FltEvents {
    hipEvent_t ev[8];
    int made = 0;
    hipError_t make() { for (; made < 8; made++) { const hipError_t e = hipEventCreate(&ev[made]); if (e != hipSuccess) return e; } return hipSuccess; }
    ~FltEvents() { for (int k = 0; k < made; k++) (void)hipEventDestroy(ev[k]); }
};

void filter_index_free(tracs_alignment *a)
{
    delete a->flt;
    a->flt = nullptr;
    a->flt_stale = true;
}

// Builds (or returns) the alignment's filter index.  On TRACS_OK, a->flt exists; a->flt->usable says whether the lists do.
static int filter_index_get(tracs_alignment *a, hipStream_t stream)
{
    if (a->flt && !a->flt_stale) return TRACS_OK;
    if (!a->flt) a->flt = new FilterIndex();
    FilterIndex *f = a->flt;
    f->usable = false;
    f->n = a->n; f->L = a->L;
    a->flt_stale = false;
    if (flt_env_off("TRACS_FILTER_LISTS") || a->L >= (1ull << FLT_POS_BITS) || a->n < 2 || a->L == 0) return TRACS_OK;
    const unsigned n = (unsigned)a->n, groups = (unsigned)a->groups;
    const unsigned chunks = (groups + FLT_GCHUNK - 1) / FLT_GCHUNK;
    FltEvents E;
    TRACS_HIP_CHECK(E.make());
    hipEvent_t *ev = E.ev;
    auto done = [&](int rc) { return rc; };
    auto soft_fail = [&]() { (void)hipGetLastError(); f->usable = false; return TRACS_OK; };
    const auto t0 = std::chrono::steady_clock::now();
    // fixed-size parts (kept from one build of the handle to the next)
    const size_t nt_groups = (groups + 31) / 32 * 32, ns_words = ns_words_for(a->n_pad);
    f->nt_groups = nt_groups; f->ns_words = ns_words;
    if (!flt_alloc(&f->ref, (size_t)groups * 4 * sizeof(uint4)) || !flt_alloc(&f->dep_off, ((size_t)n + 1) * 8) ||
        !flt_alloc(&f->counters, 64) || !flt_alloc(&f->nt, (size_t)n * nt_groups * sizeof(uint4)) ||
        !flt_alloc(&f->ns, (size_t)groups * 128 * ns_words * 4)) return soft_fail();
    f->alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    unsigned *cnt, *rel, *tot, *stats;
    int rc;
    if ((rc = workspace_get(FltWs::CNT, (size_t)chunks * n * 4, reinterpret_cast<void **>(&cnt))) ||
        (rc = workspace_get(FltWs::REL, (size_t)chunks * n * 4, reinterpret_cast<void **>(&rel))) ||
        (rc = workspace_get(FltWs::TOT, (size_t)n * 4, reinterpret_cast<void **>(&tot))) ||
        (rc = workspace_get(FltWs::STATS, 64, reinterpret_cast<void **>(&stats)))) return done(rc);

    TRACS_HIP_CHECK(hipEventRecord(ev[0], stream));
    hipLaunchKernelGGL(flt_ref_kernel, dim3(groups), dim3(64), 0, stream, a->planes, a->n_pad, n, groups, f->ref);
    TRACS_HIP_CHECK(hipEventRecord(ev[1], stream));
    hipLaunchKernelGGL((flt_dep_kernel<false>), dim3((n + 255) / 256, chunks), dim3(256), 0, stream, a->planes, a->n_pad, n, groups, f->ref,
                       cnt, nullptr, nullptr, nullptr);
    TRACS_HIP_CHECK(hipEventRecord(ev[2], stream));
    hipLaunchKernelGGL(flt_rel_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cnt, n, chunks, rel, tot);
    hipLaunchKernelGGL(flt_scan_kernel, dim3(1), dim3(1024), 0, stream, tot, n, f->dep_off, stats);
    TRACS_HIP_CHECK(hipGetLastError());
    unsigned long long total = 0;
    unsigned max_len = 0;
    TRACS_HIP_CHECK(hipMemcpyAsync(&total, f->dep_off + n, 8, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipMemcpyAsync(&max_len, stats, 4, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipEventRecord(ev[3], stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    f->total = total; f->max_len = max_len;
    // lists that outweigh a scan of the planes are not built: a pair's merge reads both lists (8 bytes per site and sample at
    // most), its scan 8 planes' bits (1 byte per site)
    if (max_len == 0xFFFFFFFFu || total >= (1ull << 32) || (double)total > 0.10 * (double)a->L * (double)n) return soft_fail();
    if (total + 64 > f->dep_cap) {
        if (f->dep) { (void)hipFree(f->dep); f->dep = nullptr; f->dep_cap = 0; }
        const size_t want = (size_t)total + (size_t)total / 8 + 4096;
        if (hipMalloc(reinterpret_cast<void **>(&f->dep), want * 4) != hipSuccess) { f->dep = nullptr; return soft_fail(); }
        f->dep_cap = want;
    }
    hipLaunchKernelGGL((flt_dep_kernel<true>), dim3((n + 255) / 256, chunks), dim3(256), 0, stream, a->planes, a->n_pad, n, groups, f->ref,
                       nullptr, rel, f->dep_off, f->dep);
    TRACS_HIP_CHECK(hipEventRecord(ev[4], stream));
    hipLaunchKernelGGL(flt_nt_kernel, dim3((unsigned)(nt_groups / 32), (n + 63) / 64), dim3(256), 0, stream, a->planes, a->n_pad, n, groups,
                       f->nt, nt_groups);
    TRACS_HIP_CHECK(hipEventRecord(ev[5], stream));
    launch_ns_build(a->planes, a->n_pad, groups, f->ns, ns_words, stream);
    TRACS_HIP_CHECK(hipEventRecord(ev[6], stream));
    TRACS_HIP_CHECK(hipGetLastError());
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    for (int k = 0; k < 6; k++) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
        f->build_ms[k] = ms;
    }
    f->usable = true;
    if (std::getenv("TRACS_CLASSES_TRACE"))
        std::fprintf(stderr, "[filter index] %llu entries (longest list %u), alloc %.2f ms, ref %.3f count %.3f offsets %.3f fill %.3f NT %.3f NS %.3f ms\n",
                     total, max_len, f->alloc_ms, f->build_ms[0], f->build_ms[1], f->build_ms[2], f->build_ms[3], f->build_ms[4], f->build_ms[5]);
    return done(TRACS_OK);
}

}  // namespace tracs

using namespace tracs;

extern "C" {

// The recombination filter on emitted pairs (device arrays): filt[t] = filter_recomb of pair (rows[t], cols[t]) whose SNP
// distance is d[t].  Builds the alignment's departure lists and N bitmaps on first use after a pack (kept on the handle).
int tracs_filter_recomb_pairs(tracs_alignment *a, const uint32_t *rows, const uint32_t *cols, const uint32_t *d, size_t n_pairs,
                              uint32_t *filt, void *stream_)
{
    if (n_pairs == 0) return TRACS_OK;
    if (!a || !rows || !cols || !d || !filt) { set_error("tracs_filter_recomb_pairs: NULL argument"); return TRACS_E_ARG; }
    if (a->L >= (1ull << 31)) { set_error("filter: alignment longer than 2^31 sites"); return TRACS_E_ARG; }
    if (n_pairs >= (1ull << 31)) { set_error("filter: more than 2^31 pairs per call"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    const double *lg = nullptr;
    int rc = get_lgamma_table_for_filter(stream, &lg);
    if (rc) return rc;
    if ((rc = filter_index_get(a, stream))) return rc;
    FilterIndex *f = a->flt;
    if (!f->counters && hipMalloc(reinterpret_cast<void **>(&f->counters), 64) != hipSuccess) { f->counters = nullptr; set_error("filter: out of device memory"); return TRACS_E_NOMEM; }
    TRACS_HIP_CHECK(hipMemsetAsync(f->counters, 0, 64, stream));
    // thresholds of the binomial test for the d of this call
    const bool use_tbl = !flt_env_off("TRACS_FILTER_TABLE");
    if (use_tbl) {
        if (!f->tbl || f->tbl_L != (unsigned)a->L) {
            if (!f->tbl) {
                if (hipMalloc(reinterpret_cast<void **>(&f->tbl), ((size_t)FLT_DCAP + 1) * FLT_KT * 4) != hipSuccess) { f->tbl = nullptr; set_error("filter: out of device memory"); return TRACS_E_NOMEM; }
                if (hipMalloc(reinterpret_cast<void **>(&f->tbl_state), FLT_DCAP + 256) != hipSuccess) { f->tbl_state = nullptr; set_error("filter: out of device memory"); return TRACS_E_NOMEM; }
            }
            TRACS_HIP_CHECK(hipMemsetAsync(f->tbl_state, 0, FLT_DCAP + 256, stream));
            f->tbl_L = (unsigned)a->L;
        }
        const unsigned blocks = (unsigned)std::min<size_t>((n_pairs + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(flt_mark_kernel, dim3(blocks), dim3(256), 0, stream, d, n_pairs, f->tbl_state);
        hipLaunchKernelGGL(flt_table_kernel, dim3(FLT_DCAP - 1), dim3(FLT_KT), 0, stream, (unsigned)a->L, lg, f->tbl_state, f->tbl);
        hipLaunchKernelGGL(flt_table_done_kernel, dim3((FLT_DCAP + 256) / 256), dim3(256), 0, stream, f->tbl_state);
    }
    const unsigned *tbl = use_tbl ? f->tbl : nullptr;
    size_t n_left = n_pairs;
    unsigned max_d_left = 0;
    bool all_left = true;
    if (f->usable) {
        unsigned cap = std::min(FLT_CAP_MAX, (std::max(f->max_len, 1u) + 63u) / 64u * 64u);
        if (const char *e = std::getenv("TRACS_FILTER_CAP")) cap = std::max(64u, std::min(FLT_CAP_MAX, (unsigned)std::atoi(e) / 64u * 64u));
        FltPairArgs A;
        A.rows = rows; A.cols = cols; A.d = d; A.n_pairs = n_pairs;
        A.dep = f->dep; A.dep_off = f->dep_off;
        A.nt = reinterpret_cast<const unsigned *>(f->nt); A.ns = f->ns;
        A.nt_words = f->nt_groups * 4; A.ns_words = f->ns_words;
        A.L = (unsigned)a->L; A.cap = cap;
        A.tbl = tbl; A.tbl_state = f->tbl_state; A.lg = lg;
        A.filt = filt; A.counters = f->counters;
        const char *form_env = std::getenv("TRACS_FILTER_KERNEL");
        const int form = form_env ? std::atoi(form_env) : 2;
        if (form == 2 && cap <= 1024) {
            const unsigned r = cap / 64;
            if (r <= 1) flt_launch2<1>(A, stream);
            else if (r <= 2) flt_launch2<2>(A, stream);
            else if (r <= 4) flt_launch2<4>(A, stream);
            else if (r <= 6) flt_launch2<6>(A, stream);
            else if (r <= 8) flt_launch2<8>(A, stream);
            else if (r <= 10) flt_launch2<10>(A, stream);
            else if (r <= 12) flt_launch2<12>(A, stream);
            else flt_launch2<16>(A, stream);
        } else if (cap <= 1024)
            hipLaunchKernelGGL((flt_pairs_kernel<4>), dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), (size_t)cap * 64, stream, A);
        else
            hipLaunchKernelGGL((flt_pairs_kernel<1>), dim3((unsigned)n_pairs), dim3(64), (size_t)cap * 16, stream, A);
        TRACS_HIP_CHECK(hipGetLastError());
        unsigned h[4] = {0, 0, 0, 0};
        TRACS_HIP_CHECK(hipMemcpyAsync(h, f->counters, 16, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        if (h[0]) { set_error("filter: SNP site list does not match the distance (internal error)"); return TRACS_E_HIP; }
        n_left = h[1];
        max_d_left = h[2];
        all_left = false;
    }
    if (n_left == 0) return TRACS_OK;
    // the scan of the planes for what is left (filter.hip)
    TRACS_HIP_CHECK(hipMemsetAsync(f->counters, 0, 4, stream));
    const unsigned blocks = (unsigned)std::min<size_t>((n_pairs + 255) / 256, 256 * 16);
    if (all_left) {
        hipLaunchKernelGGL(flt_max_kernel, dim3(blocks), dim3(256), 0, stream, d, n_pairs, f->counters + 2);
        TRACS_HIP_CHECK(hipMemcpyAsync(&max_d_left, f->counters + 2, 4, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        if ((rc = filter_scan_route(a, rows, cols, d, n_pairs, max_d_left, filt, tbl, f->tbl_state, lg, f->counters, stream))) return rc;
    } else {
        // lists too long for a wave's LDS: the same merge over global memory (flt_pairs_long_kernel)
        unsigned *idx, *scratch;
        const size_t slot = ((size_t)2 * f->max_len + 8 + 63) / 64 * 64;
        size_t waves = std::min<size_t>(n_left, 256 * TRACS_FLT_TGRID);
        while (waves > 256 && waves * slot * 4 > (3ull << 30)) waves /= 2;
        if ((rc = workspace_get(FltWs::IDX, n_left * 4, reinterpret_cast<void **>(&idx))) ||
            (rc = workspace_get(FltWs::SCRATCH, waves * slot * 4, reinterpret_cast<void **>(&scratch)))) return rc;
        TRACS_HIP_CHECK(hipMemsetAsync(f->counters + 4, 0, 4, stream));
        hipLaunchKernelGGL(flt_collect_kernel, dim3(blocks), dim3(256), 0, stream, filt, n_pairs, idx, f->counters + 4);
        FltPairArgs A;
        A.rows = rows; A.cols = cols; A.d = d; A.n_pairs = n_pairs;
        A.dep = f->dep; A.dep_off = f->dep_off;
        A.nt = reinterpret_cast<const unsigned *>(f->nt); A.ns = f->ns;
        A.nt_words = f->nt_groups * 4; A.ns_words = f->ns_words;
        A.L = (unsigned)a->L; A.cap = 0;
        A.tbl = tbl; A.tbl_state = f->tbl_state; A.lg = lg;
        A.filt = filt; A.counters = f->counters;
        // (TRACS_FILTER_LONG=seq: the per-lane sequential merge over global memory instead of the tiled one: tests, A/B)
        const char *lf = std::getenv("TRACS_FILTER_LONG");
        if (lf && lf[0] == 's')
            hipLaunchKernelGGL(flt_pairs_long_kernel, dim3((unsigned)waves), dim3(64), 0, stream, A, idx, n_left, scratch, slot);
        else
            hipLaunchKernelGGL(flt_pairs_tiled_kernel, dim3((unsigned)waves), dim3(64),
                               (size_t)(2 * (FLT_TILE + 64) + (TRACS_FLT_TREGS ? 0 : 2 * FLT_TILE + 128) + 80) * 4, stream, A, idx, n_left, scratch, slot);
    }
    TRACS_HIP_CHECK(hipGetLastError());
    unsigned bad = 0;
    TRACS_HIP_CHECK(hipMemcpyAsync(&bad, f->counters, 4, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    if (bad) { set_error("filter: SNP site list does not match the distance (internal error)"); return TRACS_E_HIP; }
    return TRACS_OK;
}

// out[0] lists usable (0 / 1), [1] list entries, [2] longest list, [3] allocation ms, [4..9] build ms: reference bases, count,
// offsets, fill, NT, NS (of the last build); returns 0 when the handle has no index yet
int tracs_debug_filter_index(const tracs_alignment *a, double *out)
{
    if (!a || !a->flt) return 0;
    const FilterIndex *f = a->flt;
    if (out) {
        out[0] = f->usable ? 1.0 : 0.0; out[1] = (double)f->total; out[2] = (double)f->max_len; out[3] = f->alloc_ms;
        for (int k = 0; k < 6; k++) out[4 + k] = f->build_ms[k];
    }
    return 1;
}

}  // extern "C"
