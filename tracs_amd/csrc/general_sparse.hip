// general_sparse.hip -- the partial-code terms of the general matrix-core path (gfx950).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp:398-403,417-420.
// For an alignment with partial IUPAC codes (M, R, W, S, Y, K, V, H, D, B; load_seqs :127-189) pairsnp_mfma_kernel<GENERAL>
// leaves, per pair,  dist = L - G + 3 NN  and  ncomp = NN  with the one-hot Gram G = sum_s |S_i n S_j| and NN = #(both N).
// |S n S'| over-counts a match exactly when both codes hold more than one allele:
//     (N, N): 4 instead of 1            -> the 3 NN above
//     (partial M, N): |M| instead of 1  -> T1 = sum over such sites of (|M| - 1)
//     (partial M, partial M'): |M n M'| instead of [M n M' != {}]  -> T2 = sum of (|M n M'| - 1)^+
// so  d = L - G + 3 NN + T1 + T2  and  nn = L - c_i - c_j + NN  (c_i = number of N sites of sample i; tests/test_host_logic.py
// checks the identity on random code matrices).  T1 and T2 only involve sites where a sample carries a partial code -- a
// fraction of a percent of a real alignment -- so they are computed from sparse lists:
//     per sample  : its sites that are N or partial, in site order            (s_off / s_ent: site << 5 | w << 4 | code, 15 = N)
//     per site    : the samples that are partial there, with their code       (p_off / p_ent: sample << 5 | w << 4 | mask)
//                   (w: only used by the minority lists of site_classes.hip, see general_fixup_kernel<MINOR>)
//                   the samples that are N there                              (n_off / n_ent: sample)
// general_fixup_kernel gives row i of the pair matrix to one workgroup: the row's correction is accumulated in LDS with
// ds_add (for every special site of sample i, walk the site's lists), then added to dist, together with nn's c_i, c_j terms.
// The lists are built once per pack (cached on the alignment handle, like the consensus planes).
#include "pairsnp_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tracs {

struct GeneralSparse {
    unsigned long long *s_off = nullptr, *p_off = nullptr, *n_off = nullptr;
    unsigned *s_ent = nullptr, *p_ent = nullptr, *n_ent = nullptr;
    unsigned *c_n = nullptr, *c_p = nullptr;      // per sample: its N sites, the sum of w over its listed (partial) sites
    double est_updates = 0.0;
    unsigned long long tot_s = 0;                 // entries of the per-sample lists
    unsigned long long max_row = 0;               // the longest per-sample list
    // site-class lists: what nn_rows_kernel walks -- per sample, for every NNL site at which it is N, where the site's N list
    // starts (in units of 8 entries; the list ends at its sentinel): no lookup of list bounds during the walk
    unsigned *s_nn = nullptr;
    unsigned long long *snn_off = nullptr;
    unsigned long long tot_nn = 0, max_row_nn = 0;
    // ... and per sample, for every minority site with one or two listed samples at which it is N, those samples themselves
    // (16 bits each, 0xFFFF: none): general_fixup_kernel<MINOR> applies them without looking at the site's list
    unsigned *s_inl = nullptr;
    unsigned long long *inl_off = nullptr;
    unsigned long long tot_inl = 0;
    unsigned split_at = 0;                 // N lists of 64 samples and more are in two parts (pairsnp_kernels.h: nn_list_is_split)
    bool in_arena = false;                        // the arrays live in the alignment's pack arena (released with it, not one by one)
    bool n16 = false;                             // n_ent holds 16-bit sample numbers (site-class lists of alignments below 65 535 samples)
    bool padded = false;                          // N lists start on 16-byte boundaries, padded with all-ones sentinels to 8 entries
};
constexpr int ENT_SHIFT = 5;                      // entries: index << 5 | w << 4 | 4-bit code

static constexpr int GS_CHUNKS = 64;          // per-sample list building: group chunks per sample

// special-site masks of one 32-site word: N, and "partial" = two or more alleles but not all four
__device__ __forceinline__ void special_masks(unsigned A, unsigned C, unsigned G, unsigned T, unsigned N, unsigned &nm, unsigned &pm)
{
    const unsigned two = (A & C) | (A & G) | (A & T) | (C & G) | (C & T) | (G & T);
    nm = N;
    pm = two & ~N;
}

__device__ __forceinline__ unsigned word_of(const uint4 &v, int w) { return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w; }

// What the list builders read.  A source hands out, per (group, sample): the N mask and the partial mask of each 32-site
// word, the 4-bit code of a listed bit, and the list index of a site (its rank among the listed sites).
//
// GeneralSrc: the five planes of a general alignment; every site is listed under its own index.
struct GeneralSrc {
    const uint4 *P;
    size_t n_pad;
    struct Group { uint4 A, C, G, T, N; };
    __device__ __forceinline__ Group load(size_t g, size_t s) const
    {
        const uint4 *base = P + (g * NPLANES) * n_pad + s;
        return Group{base[0], base[n_pad], base[2 * n_pad], base[3 * n_pad], base[4 * n_pad]};
    }
    __device__ __forceinline__ void masks(const Group &q, int w, unsigned &nm, unsigned &pm) const
    {
        special_masks(word_of(q.A, w), word_of(q.C, w), word_of(q.G, w), word_of(q.T, w), word_of(q.N, w), nm, pm);
    }
    __device__ __forceinline__ unsigned code(const Group &q, int w, int b) const
    {
        return ((word_of(q.A, w) >> b) & 1u) | (((word_of(q.C, w) >> b) & 1u) << 1) | (((word_of(q.G, w) >> b) & 1u) << 2) |
               (((word_of(q.T, w) >> b) & 1u) << 3);
    }
    __device__ __forceinline__ unsigned wmask(const Group &, int) const { return 0u; }
    __device__ __forceinline__ bool any_listed(size_t) const { return true; }
    __device__ __forceinline__ unsigned listed(size_t, int) const { return 0xFFFFFFFFu; }
    __device__ __forceinline__ size_t index(size_t g, int w, int b) const { return g * SITES_PER_GROUP + w * 32 + b; }
};

// pass A / B over the planes, lanes over samples (coalesced), one thread = (sample, chunk of groups), walked in site order.
// FILL = false: cnt[s * GS_CHUNKS + chunk] = special sites of the chunk, cn[s * GS_CHUNKS + chunk] = N sites of the chunk.
// FILL = true : entries written from off[s * GS_CHUNKS + chunk] on.
template <bool FILL, class SRC>
__global__ __launch_bounds__(256) void gs_sample_kernel(const SRC src, size_t n, size_t groups,
                                                        size_t gpc, unsigned *__restrict__ cnt, unsigned *__restrict__ cn, unsigned *__restrict__ cw,
                                                        const unsigned long long *__restrict__ off, unsigned *__restrict__ ent)
{
    const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const size_t chunk = (size_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (s >= n || chunk >= GS_CHUNKS) return;
    const size_t g0 = chunk * gpc, g1 = min(groups, g0 + gpc);
    unsigned c_all = 0, c_n = 0, c_w = 0;
    unsigned long long o = FILL ? off[s * GS_CHUNKS + chunk] : 0ull;
    for (size_t g = g0; g < g1; g++) {
        if (!src.any_listed(g)) continue;                     // wave-uniform
        const typename SRC::Group q = src.load(g, s);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned nm, pm;
            src.masks(q, w, nm, pm);
            if (!FILL) { c_all += __popc(nm | pm); c_n += __popc(nm); c_w += __popc(pm & src.wmask(q, w)); continue; }
            unsigned m = nm | pm;
            while (m) {
                const int b = __ffs(m) - 1;
                m &= m - 1;
                ent[o++] = (unsigned)(src.index(g, w, b) << ENT_SHIFT) | src.code(q, w, b);
            }
        }
    }
    if (!FILL) { cnt[s * GS_CHUNKS + chunk] = c_all; cn[s * GS_CHUNKS + chunk] = c_n; cw[s * GS_CHUNKS + chunk] = c_w; }
}

// per-sample totals -> c_n[s], and the exclusive scan of cnt over (sample, chunk) in row-major order -> off (u64).
// One workgroup; n * GS_CHUNKS elements.
__global__ __launch_bounds__(1024) void gs_scan_kernel(const unsigned *__restrict__ v, size_t count, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    // each thread takes a run of consecutive elements so the block-level scan runs once per 1024 * RUN elements
    constexpr int RUN = 16;
    for (size_t base = 0; base <= count; base += 1024 * RUN) {
        const size_t b0 = base + (size_t)threadIdx.x * RUN;
        unsigned long long local[RUN], sum = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) { local[k] = sum; sum += (b0 + k < count) ? v[b0 + k] : 0u; }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned long long excl = part[threadIdx.x] - sum + carry;
#pragma unroll
        for (int k = 0; k < RUN; k++) if (b0 + k <= count) out[b0 + k] = excl + local[k];
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

// Up to three such scans at once, one workgroup each (the per-sample streams' counts: three arrays of n x 32 elements, 0.7 ms
// each through gs_scan_kernel's LDS ladder, one after the other); wave scans through shuffles, one barrier pair per 16 384 elements.
struct ScanJob { const unsigned *v; size_t count; unsigned long long *out; };
__global__ __launch_bounds__(1024) void gs_scan_jobs_kernel(ScanJob j0, ScanJob j1, ScanJob j2)
{
    const ScanJob job = blockIdx.x == 0 ? j0 : blockIdx.x == 1 ? j1 : j2;
    if (!job.v) return;
    __shared__ unsigned long long wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RUN = 16;
    unsigned long long carry = 0;                            // (the same on every thread)
    for (size_t base = 0; base <= job.count; base += 1024 * RUN) {
        const size_t b0 = base + (size_t)threadIdx.x * RUN;
        unsigned long long local[RUN], sum = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) { local[k] = sum; sum += (b0 + k < job.count) ? job.v[b0 + k] : 0u; }
        unsigned long long incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned long long before = carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const unsigned long long t = wave_tot[w]; if (w < wave) before += t; all += t; }
        const unsigned long long excl = before + incl - sum;
#pragma unroll
        for (int k = 0; k < RUN; k++) if (b0 + k <= job.count) job.out[b0 + k] = excl + local[k];
        carry += all;
        __syncthreads();
    }
}

__global__ void gs_sample_totals_kernel(const unsigned *__restrict__ cw, const unsigned *__restrict__ cn, size_t n,
                                        unsigned *__restrict__ c_n, unsigned *__restrict__ c_p,
                                        const unsigned long long *__restrict__ off, unsigned long long *__restrict__ s_off)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n) return;
    s_off[s] = off[s * GS_CHUNKS];                       // off has n * GS_CHUNKS + 1 entries
    if (s < n) {
        unsigned t = 0, wsum = 0;
        for (int k = 0; k < GS_CHUNKS; k++) { t += cn[s * GS_CHUNKS + k]; wsum += cw[s * GS_CHUNKS + k]; }
        c_n[s] = t;
        c_p[s] = wsum;
    }
}

// pass C / D: one workgroup per 128-site group, threads over samples (coalesced).  FILL = false: per-site counts of partial
// and N samples (+ the work estimate sum_s cP (cN + cP / 2)).  FILL = true: entries placed through LDS cursors.
// Per-site arrays (cntP, cntN, p_off, n_off) are indexed by the source's list index of the site.
template <bool FILL, class SRC>
__global__ __launch_bounds__(256) void gs_site_kernel(const SRC src, size_t n, size_t L,
                                                      unsigned *__restrict__ cntP, unsigned *__restrict__ cntN,
                                                      const unsigned long long *__restrict__ p_off, const unsigned long long *__restrict__ n_off,
                                                      unsigned *__restrict__ p_ent, unsigned *__restrict__ n_ent, double *__restrict__ est)
{
    __shared__ unsigned cP[SITES_PER_GROUP], cN[SITES_PER_GROUP];
    __shared__ unsigned long long bP[SITES_PER_GROUP], bN[SITES_PER_GROUP];     // FILL: first entry of every site's lists
    const size_t g = blockIdx.x;
    if (!src.any_listed(g)) return;
    const int tw = (threadIdx.x & 127) >> 5, tb = threadIdx.x & 31;
    // this thread's site (threads 0..127): listed? under which index?
    const bool mine = threadIdx.x < SITES_PER_GROUP && ((src.listed(g, tw) >> tb) & 1u);
    const size_t my_index = mine ? src.index(g, tw, tb) : 0;
    const bool valid = mine && my_index < L;
    if (threadIdx.x < SITES_PER_GROUP) {
        cP[threadIdx.x] = 0; cN[threadIdx.x] = 0;
        if (FILL) { bP[threadIdx.x] = valid ? p_off[my_index] : 0ull; bN[threadIdx.x] = valid ? n_off[my_index] : 0ull; }
    }
    __syncthreads();
    for (size_t s = threadIdx.x; s < n; s += blockDim.x) {
        const typename SRC::Group q = src.load(g, s);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned nm, pm;
            src.masks(q, w, nm, pm);
            while (nm) {
                const int b = __ffs(nm) - 1;
                nm &= nm - 1;
                const unsigned slot = atomicAdd(&cN[w * 32 + b], 1u);
                if (FILL) n_ent[bN[w * 32 + b] + slot] = (unsigned)s;
            }
            while (pm) {
                const int b = __ffs(pm) - 1;
                pm &= pm - 1;
                const unsigned slot = atomicAdd(&cP[w * 32 + b], 1u);
                if (FILL) p_ent[bP[w * 32 + b] + slot] = ((unsigned)s << ENT_SHIFT) | src.code(q, w, b);
            }
        }
    }
    if (FILL) return;
    __syncthreads();
    if (threadIdx.x < SITES_PER_GROUP) {
        double e = 0.0;
        if (valid) {
            cntP[my_index] = cP[threadIdx.x];
            cntN[my_index] = cN[threadIdx.x];
            e = (double)cP[threadIdx.x] * ((double)cN[threadIdx.x] + 0.5 * (double)cP[threadIdx.x]);
        }
        for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
        if ((threadIdx.x & 63) == 0 && e > 0.0) atomicAdd(est, e);
    }
}

// Row i of the pair matrix: T1 + T2 accumulated in LDS, then added to dist; ncomp gets its c_i, c_j terms.
//
// MINOR: the same walk over the lists of the MINORITY sites of an alignment cut into site classes (site_classes.hip) -- sites
// at which all but a few samples are N or carry the site's reference base.  The few are listed with their allele mask M and
// w = [reference base not in M].  Such a site adds to d(i, j): w_i when i is listed and j carries the reference base,
// [M_i n M_j = {}] when both are listed, 0 when either is N -- i.e. over the sites S_i, S_j at which i / j is listed
//     d += sum_{S_i} w_i + sum_{S_j} w_j - sum_{s in S_i: j is N} w_i - sum_{s in S_j: i is N} w_j
//          + sum over S_i n S_j of ([M_i n M_j = {}] - w_i - w_j)
// (consensus alignments: M = {own base}, w = 1).  The first two sums are per-sample constants (c_p); negative terms wrap in the
// unsigned row and cancel in the final sum.  ncomp is not touched (the counting pass covers these sites).
template <bool MINOR, class NT>
__global__ __launch_bounds__(1024) void general_fixup_kernel(const unsigned long long *__restrict__ s_off, const unsigned *__restrict__ s_ent,
                                                             const unsigned long long *__restrict__ p_off, const unsigned *__restrict__ p_ent,
                                                             const unsigned long long *__restrict__ n_off, const NT *__restrict__ n_ent,
                                                             const unsigned *__restrict__ c_n, const unsigned *__restrict__ c_p, unsigned L, unsigned n, unsigned row_begin,
                                                             unsigned col_begin, unsigned chunk, unsigned *__restrict__ dist,
                                                             unsigned *__restrict__ ncomp, size_t ld,
                                                             const unsigned long long *__restrict__ inl_off, const unsigned *__restrict__ s_inl)
{
    extern __shared__ unsigned row[];
    const unsigned i = row_begin + blockIdx.x;
    const unsigned c0 = blockIdx.y * chunk, c1 = min(n, c0 + chunk);
    if (c1 <= i + 1 || c1 <= col_begin) return;            // no cell (i, j > i) in this column chunk
    for (unsigned j = threadIdx.x; j < c1 - c0; j += blockDim.x) row[j] = 0;
    __syncthreads();
    if (MINOR && s_inl) {
        // the row's N entries at sites with one or two listed samples carry those samples themselves (w = 1 each: a consensus
        // alignment): -1 for every listed j > i, straight from a coalesced stream -- no list bounds, no list
        const unsigned long long q1 = inl_off[i + 1];
        for (unsigned long long q = inl_off[i] + threadIdx.x; q < q1; q += blockDim.x) {
            const unsigned v = s_inl[q];
            const unsigned j1 = v & 0xFFFFu, j2 = v >> 16;       // (0xFFFF: none / padding)
            if (j1 > i && j1 >= c0 && j1 < c1) atomicAdd(&row[j1 - c0], 0xFFFFFFFFu);
            if (j2 > i && j2 >= c0 && j2 < c1) atomicAdd(&row[j2 - c0], 0xFFFFFFFFu);
        }
    }
    // A quarter wave takes 16 special sites of sample i at a time: lane l fetches entry l and its site's list bounds (one memory
    // round trip for the 16 of them), then the 16 lanes walk the 16 sites' lists together.
    const unsigned sub = threadIdx.x >> 4, nsub = blockDim.x >> 4, l16 = threadIdx.x & 15;
    const unsigned long long e0 = s_off[i], e1 = s_off[i + 1];
    for (unsigned long long base = e0 + (unsigned long long)sub * 16; base < e1; base += (unsigned long long)nsub * 16) {
        const unsigned long long e = base + l16;
        unsigned my_code = 0;
        unsigned long long my_pa = 0, my_pz = 0, my_na = 0, my_nz = 0;
        bool live = e < e1;
        if (live && MINOR && s_ent[e] == 0xFFFFFFFFu) live = false;    // padding of the stream (written 16 bytes at a time)
        if (live) {
            const unsigned ent = s_ent[e];
            const unsigned site = ent >> ENT_SHIFT;
            my_code = ent & 31u;                            // w << 4 | code (an N entry, code 15: bit 4 = NNL site, not a w)
            my_pa = p_off[site]; my_pz = p_off[site + 1];
            // the N list of the site is only walked when i is listed there -- and, on the minority lists, adds something (w_i = 1)
            if ((my_code & 15u) != 15u && (!MINOR || (my_code & 16u))) { my_na = n_off[site]; my_nz = n_off[site + 1]; }
        }
        // Short lists stay in their lane: where i is N and at most four samples are partial there (the usual case on the minority
        // lists: one or two), the lane applies its site's entries itself -- one round trip for the 16 sites together.
        bool coop = live;
        if (coop && (my_code & 15u) == 15u && my_pz - my_pa <= 4) {
            unsigned v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = my_pa + m < my_pz ? p_ent[my_pa + m] : 0xFFFFFFFFu;
#pragma unroll
            for (int m = 0; m < 4; m++)
                if (v[m] != 0xFFFFFFFFu) {
                    const unsigned j = v[m] >> ENT_SHIFT;
                    const int add = MINOR ? -(int)((v[m] >> 4) & 1u) : __popc(v[m] & 15u) - 1;      // MINOR: -w_j
                    if (add != 0 && (MINOR || add > 0) && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], (unsigned)add);
                }
            coop = false;
        }
        unsigned todo = (unsigned)(__ballot(coop) >> (threadIdx.x & 48)) & 0xFFFFu;     // this quarter wave's sites still to walk
        // The other sites' lists are walked one site after the other, 64 list entries per memory round trip (each lane requests its
        // four entries before it touches the first) -- and the first round trip of site k + 1 (the head of its partial list, and
        // of its N list when i is partial there) is already in flight while site k's entries are applied.
        struct Head { unsigned p[4], n[4]; };
        auto fetch_head = [&](int k, Head &h) {
            const unsigned code = __shfl(my_code, k, 16);
            const unsigned long long pa = __shfl(my_pa, k, 16) + l16, pz = __shfl(my_pz, k, 16);
            const unsigned long long na = __shfl(my_na, k, 16) + l16, nz = __shfl(my_nz, k, 16);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                h.p[m] = pa + 16 * m < pz ? p_ent[pa + 16 * m] : 0xFFFFFFFFu;
                h.n[m] = ((code & 15u) != 15u && na + 16 * m < nz) ? (unsigned)n_ent[na + 16 * m] : 0xFFFFFFFFu;
            }
        };
        Head cur, nxt;
        int k = todo ? __ffs(todo) - 1 : -1;
        if (k >= 0) fetch_head(k, cur);
        while (k >= 0) {
            todo &= todo - 1;
            const int kn = todo ? __ffs(todo) - 1 : -1;
            if (kn >= 0) fetch_head(kn, nxt);
            const unsigned code5 = __shfl(my_code, k, 16);
            const unsigned code = code5 & 15u, wi = code5 >> 4;
            const bool i_is_n = code == 15u;
            const unsigned kk = MINOR ? 0u - wi : (unsigned)__popc(code) - 1u;       // an N j: |M_i| - 1 (MINOR: -w_i)
            auto apply_p = [&](unsigned v) {                          // a partial j: i N -> |M_j| - 1; both partial -> (|M_i n M_j| - 1)^+
                const unsigned j = v >> ENT_SHIFT, mj = v & 15u;
                const int wj = (int)((v >> 4) & 1u);
                const int add = MINOR ? (i_is_n ? -wj : ((mj & code) == 0u ? 1 : 0) - (int)wi - wj)
                                      : (i_is_n ? __popc(mj) - 1 : __popc(mj & code) - 1);
                if (add != 0 && (MINOR || add > 0) && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], (unsigned)add);
            };
            auto apply_n = [&](unsigned j) {                          // an N j, i partial here: |M_i| - 1  (MINOR: -w_i)
                if (kk != 0u && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], kk);      // (0xFFFFFFFF: no entry)
            };
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (cur.p[m] != 0xFFFFFFFFu) apply_p(cur.p[m]);
                if (cur.n[m] != 0xFFFFFFFFu) apply_n(cur.n[m]);
            }
            // the tails beyond the first 64 entries (N lists of ~100 samples at 1 % N; rarely the partial list)
            const unsigned long long pz = __shfl(my_pz, k, 16), nz = __shfl(my_nz, k, 16);
            for (unsigned long long t = __shfl(my_pa, k, 16) + 64 + l16; t < pz; t += 64) {
                const bool h1 = t + 16 < pz, h2 = t + 32 < pz, h3 = t + 48 < pz;
                const unsigned v0 = p_ent[t], v1 = h1 ? p_ent[t + 16] : 0u, v2 = h2 ? p_ent[t + 32] : 0u, v3 = h3 ? p_ent[t + 48] : 0u;
                apply_p(v0); if (h1) apply_p(v1); if (h2) apply_p(v2); if (h3) apply_p(v3);
            }
            if (!i_is_n)
                for (unsigned long long t = __shfl(my_na, k, 16) + 64 + l16; t < nz; t += 64) {
                    const bool h1 = t + 16 < nz, h2 = t + 32 < nz, h3 = t + 48 < nz;
                    const unsigned v0 = n_ent[t], v1 = h1 ? n_ent[t + 16] : 0u, v2 = h2 ? n_ent[t + 32] : 0u, v3 = h3 ? n_ent[t + 48] : 0u;
                    apply_n(v0); if (h1) apply_n(v1); if (h2) apply_n(v2); if (h3) apply_n(v3);
                }
            cur = nxt;
            k = kn;
        }
    }
    __syncthreads();
    const unsigned ci = MINOR ? c_p[i] : c_n[i];
    for (unsigned j = c0 + threadIdx.x; j < c1; j += blockDim.x)
        if (j > i && j >= col_begin) {
            const size_t o = (size_t)i * ld + j;
            const unsigned t = MINOR ? row[j - c0] + ci + c_p[j] : row[j - c0];
            if (t) dist[o] += t;
            if (!MINOR && ncomp) ncomp[o] += L - ci - c_n[j];
        }
}

static void gs_free(GeneralSparse *g)
{
    if (!g) return;
    void *p[] = {g->s_off, g->p_off, g->n_off, g->s_ent, g->p_ent, g->n_ent, g->c_n, g->c_p, g->s_nn, g->snn_off, g->s_inl, g->inl_off};
    if (!g->in_arena) for (void *q : p) if (q) (void)hipFree(q);
    delete g;
}

void general_sparse_free(tracs_alignment *a)
{
    if (!a) return;
    gs_free(a->sparse);
    a->sparse = nullptr;
    a->sparse_state = 0;
}

void minority_lists_free(tracs_alignment *a)
{
    if (!a) return;
    gs_free(a->minor);
    a->minor = nullptr;
}

// The lists of a 5-plane alignment (planes, n samples, L sites).  *out = nullptr when the alignment is outside what the path
// supports (too long, too many entries, no memory) -- not an error.
// max_entries: the lists must stay a small fraction of the alignment's planes (one entry per 8 sites of the WHOLE alignment):
// beyond that the VALU kernel / the dense pair kernel is the better tool anyway
template <class SRC>
static int gs_build(const SRC src, size_t n, size_t L, size_t groups, double max_entries, hipStream_t stream, GeneralSparse **out)
{
    *out = nullptr;
    if (L >= (1ull << 27) || n >= (1ull << 27) || L == 0) return TRACS_OK;        // entries hold site << 5 / sample << 5
    auto *g = new GeneralSparse();
    unsigned *cnt = nullptr, *cn = nullptr, *cw = nullptr, *cntP = nullptr, *cntN = nullptr;
    unsigned long long *off = nullptr;
    double *d_est = nullptr;
    auto tmp_free = [&]() { void *p[] = {cnt, cn, cw, cntP, cntN, off, d_est}; for (void *q : p) if (q) (void)hipFree(q); };
    auto fail_soft = [&]() { tmp_free(); (void)hipGetLastError(); gs_free(g); return TRACS_OK; };
#define GS_TRY(x) do { if ((x) != hipSuccess) return fail_soft(); } while (0)
    const size_t nsc = n * GS_CHUNKS;
    const size_t gpc = (groups + GS_CHUNKS - 1) / GS_CHUNKS;
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&cnt), nsc * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&cn), nsc * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&cw), nsc * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&off), (nsc + 1) * 8));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&cntP), (L + 1) * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&cntN), (L + 1) * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&d_est), 8));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->s_off), (n + 1) * 8));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->p_off), (L + 1) * 8));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->n_off), (L + 1) * 8));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->c_n), std::max<size_t>(n, 1) * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->c_p), std::max<size_t>(n, 1) * 4));
    GS_TRY(hipMemsetAsync(d_est, 0, 8, stream));

    const dim3 sgrid((unsigned)((n + 63) / 64), GS_CHUNKS / 4);
    hipLaunchKernelGGL((gs_sample_kernel<false, SRC>), sgrid, dim3(256), 0, stream, src, n, groups, gpc, cnt, cn, cw, nullptr, nullptr);
    hipLaunchKernelGGL(gs_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, nsc, off);
    hipLaunchKernelGGL(gs_sample_totals_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream, cw, cn, n, g->c_n, g->c_p, off, g->s_off);
    GS_TRY(hipMemsetAsync(cntP, 0, (L + 1) * 4, stream));
    GS_TRY(hipMemsetAsync(cntN, 0, (L + 1) * 4, stream));
    hipLaunchKernelGGL((gs_site_kernel<false, SRC>), dim3((unsigned)groups), dim3(256), 0, stream, src, n, L, cntP, cntN,
                       nullptr, nullptr, nullptr, nullptr, d_est);
    hipLaunchKernelGGL(gs_scan_kernel, dim3(1), dim3(1024), 0, stream, cntP, L, g->p_off);
    hipLaunchKernelGGL(gs_scan_kernel, dim3(1), dim3(1024), 0, stream, cntN, L, g->n_off);
    unsigned long long tot_s = 0, tot_p = 0, tot_n = 0;
    double est = 0.0;
    GS_TRY(hipMemcpyAsync(&tot_s, off + nsc, 8, hipMemcpyDeviceToHost, stream));
    GS_TRY(hipMemcpyAsync(&tot_p, g->p_off + L, 8, hipMemcpyDeviceToHost, stream));
    GS_TRY(hipMemcpyAsync(&tot_n, g->n_off + L, 8, hipMemcpyDeviceToHost, stream));
    GS_TRY(hipMemcpyAsync(&est, d_est, 8, hipMemcpyDeviceToHost, stream));
    GS_TRY(hipStreamSynchronize(stream));
    if (tot_s != tot_p + tot_n) { tmp_free(); gs_free(g); set_error("general_sparse: list totals disagree (internal error)"); return TRACS_E_HIP; }
    // TRACS_LIST_CAP=<entries>: a smaller cap (diagnostics: exercises the paths taken when the lists are refused)
    static const double env_cap = [] { const char *e = std::getenv("TRACS_LIST_CAP"); return e ? std::atof(e) : -1.0; }();
    if ((double)tot_s > (env_cap >= 0.0 ? std::min(env_cap, max_entries) : max_entries)) return fail_soft();
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->s_ent), std::max<size_t>(tot_s, 1) * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->p_ent), std::max<size_t>(tot_p, 1) * 4));
    GS_TRY(hipMalloc(reinterpret_cast<void **>(&g->n_ent), std::max<size_t>(tot_n, 1) * 4));
    hipLaunchKernelGGL((gs_sample_kernel<true, SRC>), sgrid, dim3(256), 0, stream, src, n, groups, gpc, nullptr, nullptr, nullptr, off, g->s_ent);
    hipLaunchKernelGGL((gs_site_kernel<true, SRC>), dim3((unsigned)groups), dim3(256), 0, stream, src, n, L, nullptr, nullptr,
                       g->p_off, g->n_off, g->p_ent, g->n_ent, nullptr);
    GS_TRY(hipGetLastError());
    GS_TRY(hipStreamSynchronize(stream));
#undef GS_TRY
    tmp_free();
    g->est_updates = est;
    *out = g;
    return TRACS_OK;
}

int general_sparse_get(tracs_alignment *a, hipStream_t stream, int *ok, double *est_updates)
{
    *ok = 0;
    if (a->sparse_state == -1) return TRACS_OK;
    if (a->sparse_state == 1) { *ok = 1; *est_updates = a->sparse->est_updates; return TRACS_OK; }
    a->sparse_state = -1;
    // the variable sites only when site classes are in use
    const int rc = gs_build(GeneralSrc{pair_planes(a, false), a->n_pad}, a->n, pair_L(a), pair_groups(a), (double)a->n * (double)a->L / 8.0, stream, &a->sparse);
    if (rc || !a->sparse) return rc;
    a->sparse_state = 1;
    *ok = 1;
    *est_updates = a->sparse->est_updates;
    return TRACS_OK;
}

// ---- the lists of an alignment cut into site classes (site_classes.hip) ------------------------------------------------
// Two kinds of site carry lists, under one rank space (off_lst[g] = sites with lists before group g):
//   MINORITY sites  their listed samples (p lists) and their N samples (n lists): general_fixup_kernel<MINOR> adds their distances;
//   NNL sites       (2 <= cN <= a bound) only their N samples: nn_rows_kernel adds their N co-occurrences NN = sum n_i n_j to the
//                   compared-sites counts -- cN^2 list entries per site instead of n^2 / 2 pairs on the matrix cores.
// A sample's list holds its N entries at every site with lists (code 15; bit 4 = the site is an NNL site) and its listed entries.
// At a minority site every sample is N, or carries exactly the site's reference base (not listed), or is LISTED with its allele
// mask M and w = [reference base not in M] -- what the sample adds to its distance to every sample that carries the reference
// base.  classify_sites_kernel has already counted the listed and the N samples of every site, summed them per group
// (prefix sums: baseP / baseN) and flagged, per group, the samples that are listed somewhere in it; so the lists are built
// from ONE plane: per-site lists = the N plane masked with the minority sites (+ the five planes of the flagged samples
// only: ~1 % of them on a real alignment), per-sample lists = the N plane again (count, fill) + the listed entries the
// per-site pass recorded.
__device__ __forceinline__ unsigned minor_rank(const uint4 &m, unsigned off_g, int w, int b)
{
    unsigned r = off_g;
    if (w > 0) r += __popc(m.x);
    if (w > 1) r += __popc(m.y);
    if (w > 2) r += __popc(m.z);
    return r + __popc(word_of(m, w) & ((1u << b) - 1u));
}

// one workgroup per 128-site group, threads over samples: p_off / n_off of the group's minority sites, their N samples, and
// the listed samples (with code = w << 4 | allele mask); E[k] = (sample, rank << 5 | code) for the per-sample lists
template <class NT>
__global__ __launch_bounds__(256) void minor_site_lists_kernel(const MinorBuild mb, size_t n_pad, unsigned n,
                                                               unsigned long long *__restrict__ p_off, unsigned long long *__restrict__ n_off,
                                                               unsigned *__restrict__ p_ent, NT *__restrict__ n_ent, uint2 *__restrict__ E,
                                                               unsigned *__restrict__ site_inl, unsigned *__restrict__ site_start,
                                                               unsigned stage_entries)
{
    // The group's N lists are one contiguous run of n_ent (whole pads, sentinels included): they are built in LDS and leave with
    // 16-byte stores when the run fits `stage_entries` -- 5 x 10^8 scattered 2-byte stores were what the kernel was bound by --;
    // a longer run is initialised with sentinels in place and filled entry by entry.
    extern __shared__ uint4 stage_raw[];
    NT *stage = reinterpret_cast<NT *>(stage_raw);
    __shared__ unsigned long long lN[SITES_PER_GROUP];                       // a list's first entry within the group's run
    __shared__ unsigned long long run_entries;
    __shared__ unsigned kp[SITES_PER_GROUP], kn[SITES_PER_GROUP], curP[SITES_PER_GROUP], curN[SITES_PER_GROUP], rk[SITES_PER_GROUP];
    __shared__ unsigned curB[SITES_PER_GROUP], partB[SITES_PER_GROUP];      // a split list's second part: cursor, first entry
    __shared__ unsigned long long bP[SITES_PER_GROUP], bN[SITES_PER_GROUP];
    const size_t g = blockIdx.x;
    const int tid = threadIdx.x;
    if (g == 0 && tid == 0) { p_off[mb.sites] = mb.tot_p; n_off[mb.sites] = mb.tot_n; }
    const uint4 m4 = mb.lst_mask[g], q4 = mb.minor_mask[g];
    if ((m4.x | m4.y | m4.z | m4.w) == 0u) return;
    const unsigned m[4] = {m4.x, m4.y, m4.z, m4.w};          // sites with lists (N entries)
    const unsigned mp[4] = {q4.x, q4.y, q4.z, q4.w};         // minority sites among them (listed entries)
    const bool any_minor = (q4.x | q4.y | q4.z | q4.w) != 0u;
    const int tw = (tid & 127) >> 5, tb = tid & 31;
    const bool mine = tid < SITES_PER_GROUP && ((m[tw] >> tb) & 1u);
    if (tid < SITES_PER_GROUP) {
        kp[tid] = (mine && ((mp[tw] >> tb) & 1u)) ? mb.cntP[g * SITES_PER_GROUP + tid] : 0u;
        const unsigned cn = mine ? mb.cntN[g * SITES_PER_GROUP + tid] : 0u, ca = mine ? mb.cntA[g * SITES_PER_GROUP + tid] : 0u;
        kn[tid] = mine ? nn_list_padded(cn, ca, mb.split_at) : 0u;                      // (sentinel + padding: see the builder)
        // (an unsplit list has no second part: no sample passes `s >= n`)
        partB[tid] = nn_list_is_split(cn, mb.split_at) ? nn_list_first_part(ca) : 0u;
        curP[tid] = 0; curN[tid] = 0; curB[tid] = 0;
    }
    __syncthreads();
    if (tid == SITES_PER_GROUP - 1) {
        unsigned long long tot = 0;
        for (int t = 0; t < SITES_PER_GROUP; t++) tot += kn[t];
        run_entries = tot;
    }
    if (mine) {
        unsigned long long pp = 0, pn = 0;
        for (int t = 0; t < tid; t++) { pp += kp[t]; pn += kn[t]; }
        const unsigned rank = minor_rank(m4, mb.off_lst[g], tw, tb);
        bP[tid] = mb.baseP[g] + pp; bN[tid] = mb.baseN[g] + pn; rk[tid] = rank; lN[tid] = pn;
        p_off[rank] = bP[tid]; n_off[rank] = bN[tid];
    }
    __syncthreads();
    const unsigned long long run = run_entries;
    const bool staged = run <= stage_entries;                 // (block-uniform)
    constexpr unsigned EP16 = 16 / sizeof(NT);                // entries per 16 bytes (a run is whole pads: a multiple of it)
    {
        const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        uint4 *dst = staged ? stage_raw : reinterpret_cast<uint4 *>(n_ent + mb.baseN[g]);
        for (unsigned long long q = tid; q < run / EP16; q += 256) dst[q] = ones;
        if (!staged) __threadfence_block();
    }
    __syncthreads();
    NT *const out = staged ? stage : n_ent + mb.baseN[g];     // entry e of the group's run
    if (mine) {
        // the first part's last pad ends without a sentinel: the walk goes on into the second part
        const unsigned ca = mb.cntA[g * SITES_PER_GROUP + tid];
        if (partB[tid] && ca % NN_LIST_PAD) out[lN[tid] + partB[tid] - 1u] = (NT)(sizeof(NT) == 2 ? NN_LIST_FILL16 : NN_LIST_FILL32);
        // where a row's walk of this list starts, in units of 8 entries: [0] rows below split_at, [1] the others (the per-sample
        // pass looks these up instead of summing the group's padded sizes again)
        site_start[g * SITES_PER_GROUP + tid] = (unsigned)(bN[tid] / 8ull);
        site_start[(gridDim.x + g) * SITES_PER_GROUP + tid] = (unsigned)((bN[tid] + partB[tid]) / 8ull);
    }
    __syncthreads();
    const unsigned split_at = mb.split_at;
    const uint4 RX = mb.ref_x[g], RY = mb.ref_y[g];
    const uint4 *base = mb.planes + (g * NPLANES) * n_pad;
    for (unsigned s = tid; s < n; s += 256) {
        const uint4 N = base[4 * n_pad + s];
        const bool flagged = any_minor && ((mb.flags[g * mb.flag_words + (s >> 6)] >> (s & 63u)) & 1ull);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned nm = word_of(N, w) & m[w];
            while (nm) {
                const int b = __ffs(nm) - 1;
                nm &= nm - 1;
                const int t = w * 32 + b;
                const bool second = partB[t] && s >= split_at;
                const unsigned slot = second ? partB[t] + atomicAdd(&curB[t], 1u) : atomicAdd(&curN[t], 1u);
                out[lN[t] + slot] = (NT)s;
            }
        }
        if (!flagged) continue;
        const uint4 A = base[s], C = base[n_pad + s], G = base[2 * n_pad + s], T = base[3 * n_pad + s];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned a = word_of(A, w), c = word_of(C, w), gg = word_of(G, w), t = word_of(T, w), isn = word_of(N, w);
            const unsigned rx = word_of(RX, w), ry = word_of(RY, w);
            const unsigned ra = ~rx & ~ry, rc = rx & ~ry, rg = ~rx & ry, rt = rx & ry;
            const unsigned has_ref = (a & ra) | (c & rc) | (gg & rg) | (t & rt);
            const unsigned only_ref = ~((a ^ ra) | (c ^ rc) | (gg ^ rg) | (t ^ rt));
            unsigned pm = ~isn & ~only_ref & mp[w];
            while (pm) {
                const int b = __ffs(pm) - 1;
                pm &= pm - 1;
                const unsigned mask = ((a >> b) & 1u) | (((c >> b) & 1u) << 1) | (((gg >> b) & 1u) << 2) | (((t >> b) & 1u) << 3);
                const unsigned code = (((has_ref >> b) & 1u) ? 0u : 16u) | mask;
                const unsigned slot = atomicAdd(&curP[w * 32 + b], 1u);
                const unsigned long long pos = bP[w * 32 + b] + slot;
                p_ent[pos] = (s << ENT_SHIFT) | code;
                E[pos] = make_uint2(s, (rk[w * 32 + b] << ENT_SHIFT) | code);
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (staged) {
        uint4 *dst = reinterpret_cast<uint4 *>(n_ent + mb.baseN[g]);
        for (unsigned long long q = tid; q < run / EP16; q += 256) dst[q] = stage_raw[q];
    }
    // the one or two listed samples of a minority site, for the inline entries of the per-sample streams (0xFFFFFFFF: the site's N
    // entries point at its list instead)
    if (site_inl) {
        if (tid < SITES_PER_GROUP) {
            unsigned v = 0xFFFFFFFFu;
            if (mine && ((mp[tw] >> tb) & 1u) && kp[tid] >= 1u && kp[tid] <= 2u) {
                const unsigned j1 = p_ent[bP[tid]] >> ENT_SHIFT, j2 = kp[tid] == 2u ? p_ent[bP[tid] + 1] >> ENT_SHIFT : 0xFFFFu;
                v = j1 | (j2 << 16);
            }
            site_inl[g * SITES_PER_GROUP + tid] = v;
        }
    }
}

__device__ __forceinline__ bool minor_row_wanted(const MinorBuild &mb, size_t s)
{
    if (mb.n_rows == 0) return true;
    return (s >= mb.rows[0] && s < mb.rows[1]) || (mb.n_rows > 1 && s >= mb.rows[2] && s < mb.rows[3]);
}

// per-sample lists: thread = (sample, chunk of groups), lanes over samples.  Two streams per sample:
//   s_ent  what general_fixup_kernel<MINOR> walks: rank << 5 | 15 for every minority site at which the sample is N (FILL = false:
//          cnt[s * NCH + chunk] of them), then -- minor_listed_kernel -- its listed entries;
//   s_nn   what nn_rows_kernel walks: for every NNL site at which the sample is N, the start of the site's N list in units of 8
//          entries (cntq[s * GS_CHUNKS + chunk] of them);
//   s_inl  (consensus alignments below 65 535 samples) the N entries at minority sites with one or two listed samples, as those
//          samples themselves (site_inl, written by the per-site pass) instead of a pointer to the site: they leave s_ent.  The list starts of a group's 128 sites are summed once
//          per wave and group from the sites' (padded) N counts -- a wave prefix sum parked in LDS -- instead of being looked up.
static constexpr int MS_NCH = GS_CHUNKS + 1;      // the last "chunk" of a sample's s_ent list holds its listed entries
constexpr unsigned NULL_ENTRY = 0xFFFFFFFFu;      // padding of the per-sample streams: no site (both walks skip it)
template <bool FILL>
__global__ __launch_bounds__(256) void minor_sample_kernel(const MinorBuild mb, size_t n_pad, size_t n, size_t groups, size_t gpc,
                                                           unsigned *__restrict__ cnt, unsigned *__restrict__ cntq, unsigned *__restrict__ cnti,
                                                           const unsigned long long *__restrict__ off, const unsigned long long *__restrict__ offq,
                                                           const unsigned long long *__restrict__ offi, const unsigned *__restrict__ site_inl,
                                                           const unsigned *__restrict__ site_start,
                                                           unsigned *__restrict__ ent, unsigned *__restrict__ entq, unsigned *__restrict__ enti)
{
    __shared__ unsigned start8[4][SITES_PER_GROUP];
    __shared__ unsigned inl8[FILL ? 4 : 1][SITES_PER_GROUP];   // the group's site_inl words (FILL)
    const size_t s = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t chunk = (size_t)blockIdx.y * 4 + wave;
    const bool upper_half = mb.split_at && (size_t)blockIdx.x * 64 >= mb.split_at;      // (split_at is a multiple of 256: the wave's 64 samples agree)
    if (chunk >= GS_CHUNKS) return;                          // (wave-uniform)
    const bool mine = s < n && minor_row_wanted(mb, s);      // (cnt was zeroed: an unwanted sample's list is empty)
    const size_t g0 = chunk * gpc, g1 = min(groups, g0 + gpc);
    const uint4 *nplane = mb.planes + 4 * n_pad + min(s, n_pad - 1);
    unsigned c = 0, cq = 0, ci = 0;
    // a thread's entries leave four at a time (one 16-byte store instead of four scattered 4-byte ones: the fill is bound by the
    // number of store transactions, ~17 ps each, not by bytes)
    struct Stream4 {
        uint4 *out;
        unsigned b0, b1, b2, b3, nb;
        __device__ __forceinline__ void push(unsigned v)
        {
            b0 = b1; b1 = b2; b2 = b3; b3 = v;
            if (++nb == 4u) { *out++ = make_uint4(b0, b1, b2, b3); nb = 0; }
        }
        __device__ __forceinline__ void flush()              // the last 1..3 entries, then null entries
        {
            if (nb == 1u) *out = make_uint4(b3, NULL_ENTRY, NULL_ENTRY, NULL_ENTRY);
            else if (nb == 2u) *out = make_uint4(b2, b3, NULL_ENTRY, NULL_ENTRY);
            else if (nb == 3u) *out = make_uint4(b1, b2, b3, NULL_ENTRY);
        }
    };
    Stream4 es{(FILL && mine) ? reinterpret_cast<uint4 *>(ent + off[s * MS_NCH + chunk]) : nullptr, 0u, 0u, 0u, 0u, 0u};
    Stream4 qs{(FILL && mine) ? reinterpret_cast<uint4 *>(entq + offq[s * GS_CHUNKS + chunk]) : nullptr, 0u, 0u, 0u, 0u, 0u};
    Stream4 is{(FILL && mine && site_inl) ? reinterpret_cast<uint4 *>(enti + offi[s * GS_CHUNKS + chunk]) : nullptr, 0u, 0u, 0u, 0u, 0u};
    for (size_t g = g0; g < g1; g++) {
        const uint4 m4 = mb.lst_mask[g];                      // wave-uniform
        if ((m4.x | m4.y | m4.z | m4.w) == 0u) continue;
        const uint4 l4 = mb.nnl_mask[g], q4 = mb.minor_mask[g];
        const uint4 N = mine ? nplane[g * NPLANES * n_pad] : make_uint4(0u, 0u, 0u, 0u);
        if (!FILL) {
            cq += __popc(N.x & l4.x) + __popc(N.y & l4.y) + __popc(N.z & l4.z) + __popc(N.w & l4.w);
            // (which of the sample's N entries at minority sites travel inline: the sites of M_INL)
            const uint4 i4 = site_inl ? mb.inl_mask[g] : make_uint4(0u, 0u, 0u, 0u);
            ci += __popc(N.x & q4.x & i4.x) + __popc(N.y & q4.y & i4.y) + __popc(N.z & q4.z & i4.z) + __popc(N.w & q4.w & i4.w);
            c += __popc(N.x & q4.x & ~i4.x) + __popc(N.y & q4.y & ~i4.y) + __popc(N.z & q4.z & ~i4.z) + __popc(N.w & q4.w & ~i4.w);
            continue;
        }
        if ((l4.x | l4.y | l4.z | l4.w) != 0u) {
            // list starts of the group's sites (written by the per-site pass; the second table for rows from split_at on)
            const unsigned *st = site_start + ((upper_half ? groups : 0) + g) * SITES_PER_GROUP;
            const unsigned a0 = st[lane], a1 = st[64 + lane];
            __builtin_amdgcn_wave_barrier();
            start8[wave][lane] = a0;
            start8[wave][64 + lane] = a1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const unsigned og = mb.off_lst[g];
        const uint4 i4 = site_inl ? mb.inl_mask[g] : make_uint4(0u, 0u, 0u, 0u);
        if ((i4.x | i4.y | i4.z | i4.w) != 0u) {              // (wave-uniform) two coalesced loads instead of one dependent load per entry
            const unsigned a0 = site_inl[g * SITES_PER_GROUP + lane], a1 = site_inl[g * SITES_PER_GROUP + 64 + lane];
            __builtin_amdgcn_wave_barrier();
            inl8[wave][lane] = a0; inl8[wave][64 + lane] = a1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned nm = word_of(N, w) & word_of(m4, w);
            while (nm) {
                const int b = __ffs(nm) - 1;
                nm &= nm - 1;
                if ((word_of(q4, w) >> b) & 1u) {
                    if ((word_of(i4, w) >> b) & 1u) is.push(inl8[wave][w * 32 + b]);
                    else es.push((minor_rank(m4, og, w, b) << ENT_SHIFT) | 15u);
                }
                if ((word_of(l4, w) >> b) & 1u) qs.push(start8[wave][w * 32 + b]);
            }
        }
    }
    if (FILL && mine) { es.flush(); qs.flush(); if (site_inl) is.flush(); }
    // (a chunk's entries are padded to a multiple of four with null entries: 16-byte stores on 16-byte boundaries)
    if (!FILL && mine) {
        cnt[s * MS_NCH + chunk] = (c + 3u) & ~3u; cntq[s * GS_CHUNKS + chunk] = (cq + 3u) & ~3u;
        if (site_inl) cnti[s * GS_CHUNKS + chunk] = (ci + 3u) & ~3u;
    }
}

// per-sample lists, listed entries (from E).  FILL = false: cnt[s * NCH + GS_CHUNKS]++ and c_p[s] += w (c_p of EVERY sample: a
// row's cells need their column samples' sums too, whichever rows the lists are built for).
template <bool FILL>
__global__ __launch_bounds__(256) void minor_listed_kernel(const MinorBuild mb, const uint2 *__restrict__ E, unsigned long long count, unsigned *__restrict__ cnt,
                                                           unsigned *__restrict__ c_p, const unsigned long long *__restrict__ off,
                                                           unsigned *__restrict__ cur, unsigned *__restrict__ ent)
{
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const uint2 e = E[k];
    const bool wanted = minor_row_wanted(mb, e.x);
    if (!FILL) {
        if (wanted) atomicAdd(&cnt[(size_t)e.x * MS_NCH + GS_CHUNKS], 1u);
        if (e.y & 16u) atomicAdd(&c_p[e.x], 1u);
    } else if (wanted) {
        ent[off[(size_t)e.x * MS_NCH + GS_CHUNKS] + atomicAdd(&cur[e.x], 1u)] = e.y;
    }
}

// the listed entries of a sample (its last "chunk", counted with atomics) padded to a multiple of four like the other chunks
__global__ void minor_pad_listed_kernel(unsigned *__restrict__ cnt, size_t n)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) cnt[s * MS_NCH + GS_CHUNKS] = (cnt[s * MS_NCH + GS_CHUNKS] + 3u) & ~3u;
}

// per-sample offsets of a stream from the scan over its (sample, chunk) counts; the longest sample's length
__global__ void minor_sample_offsets_kernel(const unsigned long long *__restrict__ off, size_t n, int nch, unsigned long long *__restrict__ s_off,
                                            unsigned long long *__restrict__ max_row)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n) s_off[s] = off[s * nch];                   // off has n * nch + 1 entries
    if (s < n) atomicMax(max_row, off[(s + 1) * nch] - off[s * nch]);
}

int minority_lists_build(tracs_alignment *a, const MinorBuild &mb, hipStream_t stream, int *ok)
{
    *ok = 0;
    minority_lists_free(a);
    const size_t n = a->n, L = mb.sites, groups = a->groups;
    if (L == 0 || L >= (1ull << 27) || n >= (1ull << 27)) return TRACS_OK;           // entries hold rank << 5 / sample << 5
    auto *g = new GeneralSparse();
    g->in_arena = true;                                    // (pack_alloc: the alignment's arena, or hipMalloc tracked by it)
    auto fail_soft = [&]() { (void)hipGetLastError(); gs_free(g); return TRACS_OK; };
#define GS_TRY(x) do { if ((x) != hipSuccess) return fail_soft(); } while (0)
    const unsigned long long tot_s = mb.tot_p + mb.tot_minor_n - mb.tot_inl, tot_nn = mb.tot_nnl, tot_inl = mb.tot_inl;
    const bool inl = mb.inline_ok && tot_inl > 0;
    static const bool trace = std::getenv("TRACS_CLASSES_TRACE") != nullptr;
    const auto t_host0 = std::chrono::steady_clock::now();
    const size_t before = a->pack_extra.size();
    GS_TRY(pack_alloc(a, (n + 1) * 8, reinterpret_cast<void **>(&g->s_off)));
    GS_TRY(pack_alloc(a, (n + 1) * 8, reinterpret_cast<void **>(&g->snn_off)));
    GS_TRY(pack_alloc(a, (L + 1) * 8, reinterpret_cast<void **>(&g->p_off)));
    GS_TRY(pack_alloc(a, (L + 1) * 8, reinterpret_cast<void **>(&g->n_off)));
    GS_TRY(pack_alloc(a, std::max<size_t>(n, 1) * 4, reinterpret_cast<void **>(&g->c_p)));
    // (+ up to three null entries per sample and chunk: the streams are written 16 bytes at a time)
    GS_TRY(pack_alloc(a, (tot_s + 3 * n * MS_NCH + 4) * 4, reinterpret_cast<void **>(&g->s_ent)));
    GS_TRY(hipMemsetAsync(g->s_ent, 0xFF, (tot_s + 3 * n * MS_NCH + 4) * 4, stream));      // (the listed entries' padding is never written)
    GS_TRY(pack_alloc(a, (tot_nn + 3 * n * GS_CHUNKS + 4) * 4, reinterpret_cast<void **>(&g->s_nn)));
    if (inl) {
        GS_TRY(pack_alloc(a, (n + 1) * 8, reinterpret_cast<void **>(&g->inl_off)));
        GS_TRY(pack_alloc(a, (tot_inl + 3 * n * GS_CHUNKS + 4) * 4, reinterpret_cast<void **>(&g->s_inl)));
    }
    GS_TRY(pack_alloc(a, std::max<size_t>(mb.tot_p, 1) * 4, reinterpret_cast<void **>(&g->p_ent)));
    // sample numbers (and the all-ones sentinel) fit 16 bits: half the bytes of every list walk.  Every N list starts on a
    // cache-line boundary, ends with a sentinel and is padded with more to a multiple of NN_LIST_PAD entries (mb.tot_n counts the
    // padded sizes): a 16-lane group takes a whole list of up to 127 (63) samples with one 16-byte load per lane, a list of ~100
    // samples lies in 2 lines instead of 2.6, and the walk needs neither a length nor a look-up of list bounds
    g->n16 = a->n < 65535;
    g->padded = true;
    const size_t n_ent_bytes = (std::max<size_t>(mb.tot_n, 8) + 64) * (g->n16 ? 2 : 4);
    GS_TRY(pack_alloc(a, n_ent_bytes, reinterpret_cast<void **>(&g->n_ent)));
    // (minor_site_lists_kernel writes every entry of every list's pads, sentinels included; the 64 entries behind the last list are
    // only ever loaded, never taken)
    if (trace) std::fprintf(stderr, "[once per pack] host: list storage %.2f GB (%zu of 9 arrays outside the arena) %.2f ms\n",
                            ((double)(tot_s + mb.tot_p + tot_nn) * 4 + (double)n_ent_bytes) * 1e-9, a->pack_extra.size() - before,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count());
    unsigned *cnt = nullptr, *cur = nullptr, *cntq = nullptr, *cnti = nullptr, *site_inl = nullptr, *site_start = nullptr;
    unsigned long long *off = nullptr, *offq = nullptr, *offi = nullptr;
    uint2 *E = nullptr;
    const size_t nsc = n * MS_NCH, nsq = n * GS_CHUNKS;
    int rc;
    if ((rc = workspace_get(60, (nsc + 2 * nsq) * 4, reinterpret_cast<void **>(&cnt))) ||
        (rc = workspace_get(61, (nsc + 2 * nsq + 3) * 8, reinterpret_cast<void **>(&off))) ||
        (inl && (rc = workspace_get(51, groups * SITES_PER_GROUP * 4, reinterpret_cast<void **>(&site_inl)))) ||
        (rc = workspace_get(47, 2 * groups * SITES_PER_GROUP * 4, reinterpret_cast<void **>(&site_start))) ||
        (rc = workspace_get(62, std::max<size_t>(mb.tot_p, 1) * sizeof(uint2), reinterpret_cast<void **>(&E))) ||
        (rc = workspace_get(63, (std::max<size_t>(n, 1) + 8) * 4, reinterpret_cast<void **>(&cur)))) { gs_free(g); return rc; }
    cntq = cnt + nsc; offq = off + nsc + 1;
    cnti = cntq + nsq; offi = offq + nsq + 1;
    GS_TRY(hipMemsetAsync(cnt, 0, (nsc + 2 * nsq) * 4, stream));
    GS_TRY(hipMemsetAsync(cur, 0, (std::max<size_t>(n, 1) + 8) * 4, stream));
    GS_TRY(hipMemsetAsync(g->c_p, 0, std::max<size_t>(n, 1) * 4, stream));
    // LDS for a group's N lists: 34 KiB beside the kernel's ~6 KiB of cursors: four workgroups per CU (TRACS_LIST_STAGE=0: off)
    constexpr unsigned SITE_STAGE_BYTES = 34816;
    static const bool stage_on = [] { const char *e = std::getenv("TRACS_LIST_STAGE"); return !(e && std::atoi(e) == 0); }();
    if (g->n16)
        hipLaunchKernelGGL(minor_site_lists_kernel<unsigned short>, dim3((unsigned)groups), dim3(256), SITE_STAGE_BYTES, stream, mb, a->n_pad, (unsigned)n,
                           g->p_off, g->n_off, g->p_ent, reinterpret_cast<unsigned short *>(g->n_ent), E, site_inl, site_start, stage_on ? SITE_STAGE_BYTES / 2 : 0u);
    else
        hipLaunchKernelGGL(minor_site_lists_kernel<unsigned>, dim3((unsigned)groups), dim3(256), SITE_STAGE_BYTES, stream, mb, a->n_pad, (unsigned)n, g->p_off,
                           g->n_off, g->p_ent, g->n_ent, E, (unsigned *)nullptr, site_start, stage_on ? SITE_STAGE_BYTES / 4 : 0u);
    const double plane_b = (double)groups * (double)a->n_pad * sizeof(uint4);      // the N plane
    pack_stage_mark("lists: per site", stream, plane_b + (double)groups * SITES_PER_GROUP * 8.0,
                    (double)n_ent_bytes + (double)mb.tot_p * 12.0 + (double)groups * SITES_PER_GROUP * (inl ? 12.0 : 8.0) + (double)L * 16.0);
    const size_t gpc = (groups + GS_CHUNKS - 1) / GS_CHUNKS;
    const dim3 sgrid((unsigned)((n + 63) / 64), GS_CHUNKS / 4);
    const unsigned egrid = (unsigned)((mb.tot_p + 255) / 256);
    hipLaunchKernelGGL((minor_sample_kernel<false>), sgrid, dim3(256), 0, stream, mb, a->n_pad, n, groups, gpc, cnt, cntq, cnti, nullptr, nullptr, nullptr,
                       site_inl, site_start, nullptr, nullptr, nullptr);
    if (egrid) hipLaunchKernelGGL((minor_listed_kernel<false>), dim3(egrid), dim3(256), 0, stream, mb, E, mb.tot_p, cnt, g->c_p, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(minor_pad_listed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cnt, n);
    hipLaunchKernelGGL(gs_scan_jobs_kernel, dim3(inl ? 3 : 2), dim3(1024), 0, stream, ScanJob{cnt, nsc, off}, ScanJob{cntq, nsq, offq},
                       inl ? ScanJob{cnti, nsq, offi} : ScanJob{nullptr, 0, nullptr});
    unsigned long long *d_max = reinterpret_cast<unsigned long long *>(cur + ((std::max<size_t>(n, 1) + 1) & ~(size_t)1));     // behind `cur` (zeroed with it)
    hipLaunchKernelGGL(minor_sample_offsets_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream, off, n, MS_NCH, g->s_off, d_max);
    hipLaunchKernelGGL(minor_sample_offsets_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream, offq, n, GS_CHUNKS, g->snn_off, d_max + 1);
    if (inl) hipLaunchKernelGGL(minor_sample_offsets_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream, offi, n, GS_CHUNKS, g->inl_off, d_max + 2);
    hipLaunchKernelGGL((minor_sample_kernel<true>), sgrid, dim3(256), 0, stream, mb, a->n_pad, n, groups, gpc, nullptr, nullptr, nullptr, off, offq, offi,
                       site_inl, site_start, g->s_ent, g->s_nn, g->s_inl);
    if (egrid) hipLaunchKernelGGL((minor_listed_kernel<true>), dim3(egrid), dim3(256), 0, stream, mb, E, mb.tot_p, nullptr, nullptr, off, cur, g->s_ent);
    GS_TRY(hipMemcpyAsync(&g->max_row, d_max, 8, hipMemcpyDeviceToHost, stream));       // (read after the caller's synchronisation)
    GS_TRY(hipMemcpyAsync(&g->max_row_nn, d_max + 1, 8, hipMemcpyDeviceToHost, stream));
    GS_TRY(hipGetLastError());
    pack_stage_mark("lists: per sample", stream, 2.0 * plane_b + (double)groups * SITES_PER_GROUP * (inl ? 12.0 : 8.0) + (double)mb.tot_p * 16.0,
                    ((double)tot_s + (double)tot_nn + (double)(inl ? tot_inl : 0)) * 4.0 + (double)(nsc + 2 * nsq) * 12.0);
#undef GS_TRY
    g->tot_s = tot_s; g->tot_nn = tot_nn; g->tot_inl = inl ? tot_inl : 0; g->split_at = mb.split_at;
    a->minor = g;
    *ok = 1;
    return TRACS_OK;
}

// ---- N co-occurrences from lists (site classes: the NNL sites) -----------------------------------------------------------
// Row i of the pair matrix: NN(i, j) = number of NNL sites at which both i and j are N.  The row lives in LDS.  Every wave
// takes 64 entries of sample i's stream at a time -- each says where the N list of a site at which i is N starts and how long it
// is -- and parks them in its LDS scratch; then its four 16-LANE GROUPS each take one list per round -- 16 bytes per lane, i.e.
// a whole list of up to 128 samples (64 with 32-bit sample numbers) in ONE load instruction for four lists, NN_FLIGHT rounds
// in flight -- and ds_add every sample j > i they read.  (What bound the first form of this kernel was neither bytes nor LDS
// but the NUMBER of 64-lane load instructions of 2 bytes per lane: profiles/r03/nn_rows_sorted_lists_rejected.txt.)
// Work = sum over the NNL sites of cN^2 list entries, whatever the number of samples -- against n^2 / 2 pairs per site on the
// matrix cores.  The row is then added to ncomp -- with lu - c_i - c_j when no counting pass adds those terms.
// A sample with many N entries (N concentrated in few samples) would leave most of the chip idle behind a few rows: a row's
// entries are cut over up to NN_MAX_SPLITS workgroups of `target` entries (grid.z; the others exit at once), which then add their
// rows with atomics.
constexpr unsigned NN_MAX_SPLITS = 32;
#ifndef TRACS_NN_FLIGHT
#define TRACS_NN_FLIGHT 2
#endif
#ifndef TRACS_NN_THREADS
#define TRACS_NN_THREADS 1024
#endif
constexpr int NN_FLIGHT = TRACS_NN_FLIGHT;          // rounds (of four lists) per memory round trip and wave
template <class NT>
__global__ __launch_bounds__(TRACS_NN_THREADS) void nn_rows_kernel(const unsigned long long *__restrict__ s_off, const unsigned *__restrict__ s_nn,
                                                                   const NT *__restrict__ n_ent, const unsigned *__restrict__ c_u, unsigned n, unsigned row_begin, unsigned col_begin,
                                                                   unsigned chunk, unsigned long long target, unsigned *__restrict__ ncomp, size_t ld,
                                                                   int add_terms, unsigned lu, unsigned split_at)
{
    // `chunk` counters -- row[0] is column `lo`, the first cell of the row in this chunk --, 64 slots nobody reads, 64 list starts per wave
    extern __shared__ unsigned row[];
    constexpr unsigned EPL = 16 / sizeof(NT);                // entries per lane and load
    const unsigned i = row_begin + blockIdx.x;
    const unsigned c0 = blockIdx.y * chunk, c1 = min(n, c0 + chunk);
    if (c1 <= i + 1 || c1 <= col_begin) return;            // no cell (i, j > i) in this column chunk
    const unsigned long long e_first = s_off[i], e_last = s_off[i + 1];
    const unsigned long long len = e_last - e_first;
    const unsigned nz = (unsigned)min((unsigned long long)NN_MAX_SPLITS, max(1ull, (len + target - 1) / target));
    if (blockIdx.z >= nz) return;
    const unsigned long long per = ((len + nz - 1) / nz + 63) / 64 * 64;
    const unsigned long long e0 = e_first + blockIdx.z * per, e1 = min(e_last, e0 + per);
    const unsigned lo = max(max(i + 1, col_begin), c0);     // columns [lo, c1) of this chunk are cells of row i
    const unsigned span = c1 - lo;
    if (sizeof(NT) == 2 && (unsigned)(size_t)row != 0u) __builtin_trap();       // (the walk's LDS adds address row[] from 0)
    for (unsigned j = threadIdx.x; j < span + 64u; j += blockDim.x) row[j] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    constexpr unsigned LP = NN_LIST_PAD / EPL;               // lanes per pad: 8 (16-bit) or 16 (32-bit)
    // Lanes per list and load.  A row at or beyond split_at walks second parts and unsplit lists only -- one pad each, but for the
    // rare list of 64 and more samples in a part --, so eight lanes (one pad = one cache line) take a list and a wave load covers
    // eight lists; the rows below split_at take two pads at once with sixteen lanes.  What the walk is bound by is the lines it
    // pulls through the fabric (PMC: ~7.3 TB/s of requests with sixteen lanes for every row): the second pad of a
    // sixteen-lane load is fetched whether its entries are taken or not, so the upper rows must not ask for it.
    const unsigned lgs = (LP < 16u && split_at && i >= split_at) ? 3u : 4u, LG = 1u << lgs, lists_per_load = 64u >> lgs;
    const unsigned grp = lane >> lgs, l16 = lane & (LG - 1u);
    unsigned *scratch = row + chunk + 64 + wave * 64;
    constexpr unsigned SENT = (unsigned)(NT)~(NT)0;
    // The entries of one group load.  A list occupies whole pads of NN_LIST_PAD entries (LP lanes) and the pad behind its last
    // sample-holding pad belongs to the next list: a group load of two pads takes its second pad only if the first one ends
    // without a sentinel (one compare per lane and one ballot -- not a compare per entry).
    // Every entry of a pad taken adds to LDS, without a branch: column j goes to row[j - lo], and whatever is no cell of the row
    // (j <= i, another column chunk, the sentinels) wraps beyond `span` in j - lo and is clamped to the lane's own slot behind the
    // row, row[span + lane] -- two packed 16-bit instructions per pair of entries and a shift each.
    // Returns true when the load's last entry is no sentinel: the list goes on.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 lo2 = {(unsigned short)lo, (unsigned short)lo}, out2 = {(unsigned short)(span + lane), (unsigned short)(span + lane)};
    auto bump_all = [&](const uint4 &d) -> bool {
        const unsigned w[4] = {d.x, d.y, d.z, d.w};
        const unsigned tail = sizeof(NT) == 2 ? (w[3] >> 16) : w[3];
        const unsigned long long goes_on = __ballot(tail != SENT) >> (grp << lgs);      // this group's lanes from bit 0
        const bool first_full = (goes_on >> (LP - 1u)) & 1ull;                           // first pad without a sentinel
        const bool take = l16 < LP || first_full;
        const bool on = ((goes_on >> (LG - 1u)) & 1ull) && first_full;
        if (take) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (sizeof(NT) == 2) {
                    us2 v = __builtin_bit_cast(us2, w[k]);
                    v = __builtin_elementwise_min((us2)(v - lo2), out2);
                    // (byte offsets of the two counters: one SDWA shift per half -- the compiler takes a mask / bit-field extract
                    // and a shift each)
                    const unsigned pair = __builtin_bit_cast(unsigned, v);
                    unsigned b0, b1;
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(b0) : "v"(2u), "v"(pair));
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(b1) : "v"(2u), "v"(pair));
                    // (row[] is the kernel's only LDS object: byte offset = LDS address, checked at the top)
                    asm volatile("ds_add_u32 %0, %1" : : "v"(b0), "v"(1u) : "memory");
                    asm volatile("ds_add_u32 %0, %1" : : "v"(b1), "v"(1u) : "memory");
                } else atomicAdd(&row[min(w[k] - lo, span + lane)], 1u);
            }
        }
        return on;
    };
    const uint4 *__restrict__ lists = reinterpret_cast<const uint4 *>(n_ent);      // (every list starts on a 16-byte boundary)
    // The walk is a chain of dependent round trips (stream entry -> list -> LDS adds); the loop is software-pipelined twice over:
    // the NEXT batch's stream entries are requested when a batch starts, and the NEXT round's lists (NN_FLIGHT x 4 or 8 of them)
    // before the current round's entries are added.  (Measured: no faster than without -- at 32 waves per CU the fabric is
    // already kept full -- but no slower, and it does not depend on the occupancy.)
    const uint4 none = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    // (every load is issued unconditionally -- a lane without a list reads the head of the array and drops it when the round is
    // applied --: straight-line code, so that the compiler waits for the current round with the next one still in flight)
    struct Round { unsigned sl[NN_FLIGHT]; uint4 d[NN_FLIGHT]; };
    auto fetch = [&](unsigned t0, unsigned cnt, Round &r) {
#pragma unroll
        for (int u = 0; u < NN_FLIGHT; u++) {
            const unsigned idx = t0 + lists_per_load * u + grp;
            r.sl[u] = scratch[min(idx, 63u)];
            if (idx >= cnt) r.sl[u] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < NN_FLIGHT; u++) r.d[u] = lists[(size_t)(r.sl[u] != 0xFFFFFFFFu ? r.sl[u] : 0u) + l16];
    };
    auto apply = [&](const Round &r) {
#pragma unroll
        for (int u = 0; u < NN_FLIGHT; u++) {
            // a list goes on behind its first 16 x 16 bytes while the group has seen no sentinel (rare: lists of 128 (64)
            // samples and more); which of the wave's four groups go on is a ballot away
            const bool has = r.sl[u] != 0xFFFFFFFFu;
            bool on = bump_all(make_uint4(has ? r.d[u].x : 0xFFFFFFFFu, has ? r.d[u].y : 0xFFFFFFFFu, has ? r.d[u].z : 0xFFFFFFFFu, has ? r.d[u].w : 0xFFFFFFFFu));
            for (unsigned step = 1;; step++) {
                if (!__ballot(on)) break;                // wave-uniform
                const uint4 nx = on ? lists[(size_t)r.sl[u] + LG * step + l16] : none;
                const bool more = bump_all(nx);
                on = on && more;
            }
        }
    };
    const unsigned long long first = e0 + (unsigned long long)wave * 64, stride = (unsigned long long)nwaves * 64;
    unsigned st_next = first + lane < e1 ? s_nn[first + lane] : 0xFFFFFFFFu;          // (0xFFFFFFFF: padding of the stream)
    for (unsigned long long base = first; base < e1; base += stride) {
        // this lane's entry of the batch -- the start of a list, in units of 16 bytes -- parked in the wave's LDS scratch
        const unsigned st = st_next;
        const unsigned long long en = base + stride + lane;
        unsigned st_load = s_nn[min(en, e_last - 1)];        // (unconditional, like the list loads; dropped below when beyond the end)
        scratch[lane] = st == 0xFFFFFFFFu ? 0xFFFFFFFFu : (unsigned)((unsigned long long)st * 8ull / EPL);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned cnt = (unsigned)min(64ull, e1 - base);
        // (two rounds per trip, each in its own registers: handing a round over by copy would wait for its loads.  A fetch beyond
        // the batch finds no list and loads the head of the array, which stays in cache)
        Round ra, rb;
        fetch(0, cnt, ra);
        const unsigned per_round = lists_per_load * NN_FLIGHT;
        for (unsigned t0 = 0; t0 < cnt; t0 += 2 * per_round) {
            fetch(t0 + per_round, cnt, rb);
            apply(ra);
            fetch(t0 + 2 * per_round, cnt, ra);
            apply(rb);
        }
        st_next = en < e1 ? st_load : 0xFFFFFFFFu;
        __builtin_amdgcn_wave_barrier();                   // (the scratch is rewritten by the next batch)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");   // (the adds issued from inline assembly are not in the compiler's count)
    __syncthreads();
    const bool terms = add_terms && blockIdx.z == 0;
    const unsigned ci = terms ? c_u[i] : 0u;
    for (unsigned j = lo + threadIdx.x; j < c1; j += blockDim.x) {
        const unsigned v = row[j - lo] + (terms ? lu - ci - c_u[j] : 0u);
        if (v) {
            if (nz > 1) atomicAdd(&ncomp[(size_t)i * ld + j], v);
            else ncomp[(size_t)i * ld + j] += v;
        }
    }
}

int nn_rows_add(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *ncomp, size_t ld, int add_terms,
                unsigned lu, hipStream_t stream)
{
    const GeneralSparse *g = a->minor;
    if (!g || !g->s_nn) { set_error("nn_rows_add: lists not built"); return TRACS_E_ARG; }
    const size_t n = a->n;
    const unsigned chunk = (unsigned)std::min<size_t>((n + 63) / 64 * 64, 32768);
    static bool attr_set = false;
    if (!attr_set) {
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_rows_kernel<unsigned>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4 + 256 + TRACS_NN_THREADS * 4));
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_rows_kernel<unsigned short>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4 + 256 + TRACS_NN_THREADS * 4));
        attr_set = true;
    }
    // ~2048 workgroups' worth of entries each, never less than 8192 entries (a workgroup's fixed cost: its row in LDS)
    static const unsigned long long target_env = [] { const char *e = std::getenv("TRACS_NN_TARGET"); return e ? std::strtoull(e, nullptr, 10) : 0ull; }();
    const unsigned long long target = target_env ? target_env : std::max<unsigned long long>(8192ull, g->tot_nn / 2048ull);
    const unsigned splits = (unsigned)std::min<unsigned long long>(NN_MAX_SPLITS, std::max<unsigned long long>(1, (g->max_row_nn + target - 1) / target));
    const dim3 grid((unsigned)(row_end - row_begin), (unsigned)((n + chunk - 1) / chunk), splits);
    if (g->n16)
        hipLaunchKernelGGL(nn_rows_kernel<unsigned short>, grid, dim3(TRACS_NN_THREADS), chunk * 4 + 256 + TRACS_NN_THREADS * 4, stream, g->snn_off, g->s_nn,
                           reinterpret_cast<const unsigned short *>(g->n_ent), a->c_counted, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin,
                           chunk, target, ncomp, ld, add_terms, lu, g->split_at);
    else
        hipLaunchKernelGGL(nn_rows_kernel<unsigned>, grid, dim3(TRACS_NN_THREADS), chunk * 4 + 256 + TRACS_NN_THREADS * 4, stream, g->snn_off, g->s_nn, g->n_ent, a->c_counted,
                           (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, target, ncomp, ld, add_terms, lu, g->split_at);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

static int fixup_launch(const GeneralSparse *g, bool minor, unsigned L, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                        unsigned *dist, unsigned *ncomp, size_t ld, hipStream_t stream)
{
    const unsigned chunk = (unsigned)std::min<size_t>((n + 63) / 64 * 64, 32768);
    static bool attr_set = false;
    if (!attr_set) {
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(general_fixup_kernel<false, unsigned>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(general_fixup_kernel<true, unsigned>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(general_fixup_kernel<true, unsigned short>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
        attr_set = true;
    }
    const dim3 grid((unsigned)(row_end - row_begin), (unsigned)((n + chunk - 1) / chunk));
    if (minor && g->n16)
        hipLaunchKernelGGL((general_fixup_kernel<true, unsigned short>), grid, dim3(1024), chunk * 4, stream, g->s_off, g->s_ent, g->p_off, g->p_ent,
                           g->n_off, reinterpret_cast<const unsigned short *>(g->n_ent), g->c_n, g->c_p, L, (unsigned)n, (unsigned)row_begin,
                           (unsigned)col_begin, chunk, dist, ncomp, ld, g->inl_off, g->s_inl);
    else if (minor)
        hipLaunchKernelGGL((general_fixup_kernel<true, unsigned>), grid, dim3(1024), chunk * 4, stream, g->s_off, g->s_ent, g->p_off, g->p_ent, g->n_off,
                           g->n_ent, g->c_n, g->c_p, L, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, dist, ncomp, ld, g->inl_off, g->s_inl);
    else
        hipLaunchKernelGGL((general_fixup_kernel<false, unsigned>), grid, dim3(1024), chunk * 4, stream, g->s_off, g->s_ent, g->p_off, g->p_ent, g->n_off,
                           g->n_ent, g->c_n, g->c_p, L, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, dist, ncomp, ld, g->inl_off, g->s_inl);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// dist[i][j] += the minority sites' contribution (consensus alignments cut into site classes)
int minority_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist, size_t ld, hipStream_t stream)
{
    if (!a->minor) return TRACS_OK;
    return fixup_launch(a->minor, true, 0u, a->n, row_begin, row_end, col_begin, dist, nullptr, ld, stream);
}

int general_sparse_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist, unsigned *ncomp,
                         size_t ld, hipStream_t stream)
{
    if (a->sparse_state != 1 || !a->sparse) { set_error("general_sparse_fixup: lists not built"); return TRACS_E_ARG; }
    return fixup_launch(a->sparse, false, (unsigned)pair_L(a), a->n, row_begin, row_end, col_begin, dist, ncomp, ld, stream);
}

}  // namespace tracs
