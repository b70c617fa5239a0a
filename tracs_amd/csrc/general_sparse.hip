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
//                   (w: unused here -- the minority lists of an alignment cut into site classes carry it: site_lists.hip)
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
// (The minority sites of an alignment cut into site classes have their own lists and kernels: site_lists.hip.)
__global__ __launch_bounds__(1024) void general_fixup_kernel(const unsigned long long *__restrict__ s_off, const unsigned *__restrict__ s_ent,
                                                             const unsigned long long *__restrict__ p_off, const unsigned *__restrict__ p_ent,
                                                             const unsigned long long *__restrict__ n_off, const unsigned *__restrict__ n_ent,
                                                             const unsigned *__restrict__ c_n, unsigned L, unsigned n, unsigned row_begin,
                                                             unsigned col_begin, unsigned chunk, unsigned *__restrict__ dist,
                                                             unsigned *__restrict__ ncomp, size_t ld)
{
    extern __shared__ unsigned row[];
    const unsigned i = row_begin + blockIdx.x;
    const unsigned c0 = blockIdx.y * chunk, c1 = min(n, c0 + chunk);
    if (c1 <= i + 1 || c1 <= col_begin) return;            // no cell (i, j > i) in this column chunk
    for (unsigned j = threadIdx.x; j < c1 - c0; j += blockDim.x) row[j] = 0;
    __syncthreads();
    // A quarter wave takes 16 special sites of sample i at a time: lane l fetches entry l and its site's list bounds (one memory
    // round trip for the 16 of them), then the 16 lanes walk the 16 sites' lists together.
    const unsigned sub = threadIdx.x >> 4, nsub = blockDim.x >> 4, l16 = threadIdx.x & 15;
    const unsigned long long e0 = s_off[i], e1 = s_off[i + 1];
    for (unsigned long long base = e0 + (unsigned long long)sub * 16; base < e1; base += (unsigned long long)nsub * 16) {
        const unsigned long long e = base + l16;
        unsigned my_code = 0;
        unsigned long long my_pa = 0, my_pz = 0, my_na = 0, my_nz = 0;
        const bool live = e < e1;
        if (live) {
            const unsigned ent = s_ent[e];
            const unsigned site = ent >> ENT_SHIFT;
            my_code = ent & 31u;                            // code (15 = N)
            my_pa = p_off[site]; my_pz = p_off[site + 1];
            // the N list of the site is only walked when i is partial there
            if ((my_code & 15u) != 15u) { my_na = n_off[site]; my_nz = n_off[site + 1]; }
        }
        // Short lists stay in their lane: where i is N and at most four samples are partial there, the lane applies its site's
        // entries itself -- one round trip for the 16 sites together.
        bool coop = live;
        if (coop && (my_code & 15u) == 15u && my_pz - my_pa <= 4) {
            unsigned v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = my_pa + m < my_pz ? p_ent[my_pa + m] : 0xFFFFFFFFu;
#pragma unroll
            for (int m = 0; m < 4; m++)
                if (v[m] != 0xFFFFFFFFu) {
                    const unsigned j = v[m] >> ENT_SHIFT;
                    const int add = __popc(v[m] & 15u) - 1;
                    if (add > 0 && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], (unsigned)add);
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
                h.n[m] = ((code & 15u) != 15u && na + 16 * m < nz) ? n_ent[na + 16 * m] : 0xFFFFFFFFu;
            }
        };
        Head cur, nxt;
        int k = todo ? __ffs(todo) - 1 : -1;
        if (k >= 0) fetch_head(k, cur);
        while (k >= 0) {
            todo &= todo - 1;
            const int kn = todo ? __ffs(todo) - 1 : -1;
            if (kn >= 0) fetch_head(kn, nxt);
            const unsigned code = __shfl(my_code, k, 16) & 15u;
            const bool i_is_n = code == 15u;
            const unsigned kk = (unsigned)__popc(code) - 1u;              // an N j: |M_i| - 1
            auto apply_p = [&](unsigned v) {                          // a partial j: i N -> |M_j| - 1; both partial -> (|M_i n M_j| - 1)^+
                const unsigned j = v >> ENT_SHIFT, mj = v & 15u;
                const int add = i_is_n ? __popc(mj) - 1 : __popc(mj & code) - 1;
                if (add > 0 && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], (unsigned)add);
            };
            auto apply_n = [&](unsigned j) {                          // an N j, i partial here: |M_i| - 1
                if (kk != 0u && j > i && j >= c0 && j < c1) atomicAdd(&row[j - c0], kk);      // (0xFFFFFFFF: no entry)
            };
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (cur.p[m] != 0xFFFFFFFFu) apply_p(cur.p[m]);
                if (cur.n[m] != 0xFFFFFFFFu) apply_n(cur.n[m]);
            }
            // the tails beyond the first 64 entries
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
    const unsigned ci = c_n[i];
    for (unsigned j = c0 + threadIdx.x; j < c1; j += blockDim.x)
        if (j > i && j >= col_begin) {
            const size_t o = (size_t)i * ld + j;
            const unsigned t = row[j - c0];
            if (t) dist[o] += t;
            if (ncomp) ncomp[o] += L - ci - c_n[j];
        }
}

static void gs_free(GeneralSparse *g)
{
    if (!g) return;
    void *p[] = {g->s_off, g->p_off, g->n_off, g->s_ent, g->p_ent, g->n_ent, g->c_n, g->c_p};
    for (void *q : p) if (q) (void)hipFree(q);
    delete g;
}

void general_sparse_free(tracs_alignment *a)
{
    if (!a) return;
    gs_free(a->sparse);
    a->sparse = nullptr;
    a->sparse_state = 0;
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

static int fixup_launch(const GeneralSparse *g, unsigned L, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                        unsigned *dist, unsigned *ncomp, size_t ld, hipStream_t stream)
{
    const unsigned chunk = (unsigned)std::min<size_t>((n + 63) / 64 * 64, 32768);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[64] = {false};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(general_fixup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
        attr_set[dev] = true;
    }
    const dim3 grid((unsigned)(row_end - row_begin), (unsigned)((n + chunk - 1) / chunk));
    hipLaunchKernelGGL(general_fixup_kernel, grid, dim3(1024), chunk * 4, stream, g->s_off, g->s_ent, g->p_off, g->p_ent, g->n_off,
                       g->n_ent, g->c_n, L, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, dist, ncomp, ld);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int general_sparse_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist, unsigned *ncomp,
                         size_t ld, hipStream_t stream)
{
    if (a->sparse_state != 1 || !a->sparse) { set_error("general_sparse_fixup: lists not built"); return TRACS_E_ARG; }
    return fixup_launch(a->sparse, (unsigned)pair_L(a), a->n, row_begin, row_end, col_begin, dist, ncomp, ld, stream);
}

}  // namespace tracs
