// site_classes.hip -- site classes of a packed alignment, decided once per pack.
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   pair loop :395-420   d = L - popcount(match),  nn = L - popcount(Ni | Nj): every site is visited for every pair.
//
// A site at which every sample that is not N carries the SAME base never separates two samples: it adds 0 to d(i, j) and
// [neither i nor j is N there] to nn(i, j), whatever the pair; and a site at which only a FEW samples differ from the others
// separates only the pairs that involve one of those few.  Real alignments are mostly such sites.  The sites are cut into
// classes once per pack (k = samples that are neither N nor exactly the site's reference base -- the base of one of its one-base
// samples --, i.e. another base or a partial IUPAC code; cN = samples that are N there):
//     empty      every sample is N (or the tail bits behind L): contributes to nothing;
//     dense      k (cN + k) above the budget below: the usual pair kernel, over `vplanes` -- these sites re-packed in site
//                order, same planes and layout as the kernels' usual source;
//     counted    every other site with cN >= 1: NN = sum n_i n_j over `iplanes` (ONE plane, n = "this sample is N here";
//                pairsnp_mfma_kernel<COUNT>, one operand plane instead of four or five) and nn += sites - c_i - c_j + NN;
//     full       every other site with cN = 0: +1 to every nn, a constant;
//     minority   the counted / full sites with k >= 1: d gets their contribution from sparse lists -- the k samples with their
//                allele masks, the cN samples -- with the machinery general_sparse.hip uses for partial IUPAC codes
//                (general_fixup_kernel<MINOR>: [reference base not in the listed sample's mask] for a pair of a listed sample
//                and a reference-base sample, [masks disjoint] for two listed samples, 0 when either is N).
// The decomposition is exact site by site (tests/test_host_logic.py::test_site_class_identity); a pass costs
// (4 L_dense + L_counted) / 4 L of the dense one in matrix instructions plus ~sum k (cN + k) list entries.  Chosen when that
// is < 0.92; TRACS_SITE_CLASSES=0/1 forces, TRACS_MINORITY=0 keeps every site with k >= 1 dense.
#include "pairsnp_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace tracs {

__device__ __forceinline__ unsigned wave_or(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}

// One workgroup per 128-site group.  Pass 1: a reference base per site -- the base of the first sample, in thread order, that
// carries exactly one base there.  Pass 2: per site k = samples that are neither N nor exactly that base (another base, or a
// partial IUPAC code) and cN = samples that are N, counted with LDS atomics (both are sparse).  Then the class masks of the group.
// CONS: planes X, Y, V (3 per group; bases A = 0, C = 1, G = 2, T = 3 = X + 2 Y).  !CONS: planes A, C, G, T, N (5 per group).
template <bool CONS>
__global__ __launch_bounds__(256) void classify_sites_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, unsigned budget,
                                                             uint4 *__restrict__ dense_mask, uint4 *__restrict__ count_mask,
                                                             uint4 *__restrict__ minor_mask, uint4 *__restrict__ full_mask,
                                                             uint4 *__restrict__ ref_x, uint4 *__restrict__ ref_y)
{
    const size_t g = blockIdx.x;
    __shared__ unsigned red[4][4][4];
    __shared__ unsigned sref[4][4];                     // one-base sample seen, ref X, ref Y, somebody is not N
    __shared__ unsigned cM[SITES_PER_GROUP], cN[SITES_PER_GROUP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < SITES_PER_GROUP) { cM[threadIdx.x] = 0; cN[threadIdx.x] = 0; }
    // per sample and word: x, y = the base's two bits where the sample carries exactly one base (`one`), `some` = not N
    auto load = [&](unsigned s, unsigned (&x)[4], unsigned (&y)[4], unsigned (&one)[4], unsigned (&some)[4], unsigned (&isn)[4]) {
        if constexpr (CONS) {
            const uint4 X = P[(g * 3 + 0) * n_pad + s], Y = P[(g * 3 + 1) * n_pad + s], V = P[(g * 3 + 2) * n_pad + s];
            const unsigned xx[4] = {X.x, X.y, X.z, X.w}, yy[4] = {Y.x, Y.y, Y.z, Y.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
            for (int w = 0; w < 4; w++) { x[w] = xx[w]; y[w] = yy[w]; one[w] = vv[w]; some[w] = vv[w]; isn[w] = ~vv[w]; }
        } else {
            const uint4 A = P[(g * NPLANES + 0) * n_pad + s], C = P[(g * NPLANES + 1) * n_pad + s];
            const uint4 G = P[(g * NPLANES + 2) * n_pad + s], T = P[(g * NPLANES + 3) * n_pad + s];
            const uint4 N = P[(g * NPLANES + 4) * n_pad + s];
            const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w}, gg[4] = {G.x, G.y, G.z, G.w};
            const unsigned t[4] = {T.x, T.y, T.z, T.w}, nn[4] = {N.x, N.y, N.z, N.w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned two = (a[w] & c[w]) | (a[w] & gg[w]) | (a[w] & t[w]) | (c[w] & gg[w]) | (c[w] & t[w]) | (gg[w] & t[w]);
                const unsigned any = a[w] | c[w] | gg[w] | t[w];
                one[w] = any & ~two;                            // exactly one allele
                some[w] = any & ~nn[w];                         // a base or a partial code (tail bits: neither)
                isn[w] = nn[w];
                x[w] = (c[w] | t[w]) & one[w];
                y[w] = (gg[w] | t[w]) & one[w];
            }
        }
    };
    unsigned seen[4] = {0, 0, 0, 0}, rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0}, anyb[4] = {0, 0, 0, 0};
    for (unsigned s = threadIdx.x; s < n; s += 256) {
        unsigned x[4], y[4], one[4], some[4], isn[4];
        load(s, x, y, one, some, isn);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned fresh = one[w] & ~seen[w];
            rx[w] |= x[w] & fresh; ry[w] |= y[w] & fresh; seen[w] |= one[w]; anyb[w] |= some[w];
        }
    }
    // lower lanes first: the result is the same on every lane of the wave
#pragma unroll
    for (int w = 0; w < 4; w++) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned os = __shfl_xor(seen[w], off, 64), ox = __shfl_xor(rx[w], off, 64), oy = __shfl_xor(ry[w], off, 64);
            const bool me_first = (lane & off) == 0;
            const unsigned fs = me_first ? seen[w] : os, fx = me_first ? rx[w] : ox, fy = me_first ? ry[w] : oy;
            const unsigned ls = me_first ? os : seen[w], lx = me_first ? ox : rx[w], ly = me_first ? oy : ry[w];
            rx[w] = fx | (lx & ~fs); ry[w] = fy | (ly & ~fs); seen[w] = fs | ls;
            anyb[w] |= __shfl_xor(anyb[w], off, 64);
        }
        if (lane == 0) { red[wave][0][w] = seen[w]; red[wave][1][w] = rx[w]; red[wave][2][w] = ry[w]; red[wave][3][w] = anyb[w]; }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int w = threadIdx.x;
        unsigned fs = 0, fx = 0, fy = 0, fa = 0;
        for (int k = 0; k < 4; k++) {
            fx |= red[k][1][w] & ~fs; fy |= red[k][2][w] & ~fs; fs |= red[k][0][w]; fa |= red[k][3][w];
        }
        sref[0][w] = fs; sref[1][w] = fx; sref[2][w] = fy; sref[3][w] = fa;
    }
    __syncthreads();
    // (a site without any one-base sample keeps the reference A: every sample that is not N is listed there)
    const unsigned any[4] = {sref[3][0], sref[3][1], sref[3][2], sref[3][3]};
    const unsigned refx[4] = {sref[1][0], sref[1][1], sref[1][2], sref[1][3]}, refy[4] = {sref[2][0], sref[2][1], sref[2][2], sref[2][3]};
    for (unsigned s = threadIdx.x; s < n; s += 256) {
        unsigned x[4], y[4], one[4], some[4], isn[4];
        load(s, x, y, one, some, isn);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            // listed: not N and not exactly the reference base
            unsigned diff = some[w] & ~(one[w] & ~((x[w] ^ refx[w]) | (y[w] ^ refy[w])));
            while (diff) { const int b = __ffs(diff) - 1; diff &= diff - 1; atomicAdd(&cM[w * 32 + b], 1u); }
            unsigned nb = isn[w] & any[w];                      // N at a site where somebody is not
            while (nb) { const int b = __ffs(nb) - 1; nb &= nb - 1; atomicAdd(&cN[w * 32 + b], 1u); }
        }
    }
    __syncthreads();
    if (threadIdx.x < SITES_PER_GROUP) {
        const int t = threadIdx.x, w = t >> 5, b = t & 31;
        const bool some = (any[w] >> b) & 1u;
        const unsigned long long k = cM[t], c = cN[t];
        const bool minor = some && k >= 1 && budget > 0 && k * (c + k) <= (unsigned long long)budget;
        const bool dense = some && k >= 1 && !minor;
        const bool counted = some && !dense && c >= 1;
        const bool full = some && !dense && c == 0;
        const unsigned long long bd = __ballot(dense), bc = __ballot(counted), bm = __ballot(minor), bf = __ballot(full);
        if (lane == 0) {
            unsigned *pd = reinterpret_cast<unsigned *>(&dense_mask[g]), *pc = reinterpret_cast<unsigned *>(&count_mask[g]);
            unsigned *pm = reinterpret_cast<unsigned *>(&minor_mask[g]), *pf = reinterpret_cast<unsigned *>(&full_mask[g]);
            pd[2 * wave] = (unsigned)bd; pd[2 * wave + 1] = (unsigned)(bd >> 32);
            pc[2 * wave] = (unsigned)bc; pc[2 * wave + 1] = (unsigned)(bc >> 32);
            pm[2 * wave] = (unsigned)bm; pm[2 * wave + 1] = (unsigned)(bm >> 32);
            pf[2 * wave] = (unsigned)bf; pf[2 * wave + 1] = (unsigned)(bf >> 32);
        }
    }
    if (threadIdx.x < 4) {
        reinterpret_cast<unsigned *>(&ref_x[g])[threadIdx.x] = refx[threadIdx.x];
        reinterpret_cast<unsigned *>(&ref_y[g])[threadIdx.x] = refy[threadIdx.x];
    }
}

// exclusive prefix sums of one class's per-group sizes (one workgroup walks the groups 1024 at a time); *total
__global__ __launch_bounds__(1024) void class_offsets_kernel(const uint4 *__restrict__ mask, size_t groups, unsigned *__restrict__ off,
                                                             unsigned long long *__restrict__ total)
{
    __shared__ unsigned sv[1024];
    unsigned long long base = 0;
    const int t = threadIdx.x;
    for (size_t g0 = 0; g0 < groups; g0 += 1024) {
        const size_t g = g0 + t;
        unsigned c = 0;
        if (g < groups) { const uint4 a = mask[g]; c = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w); }
        sv[t] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const unsigned av = t >= o ? sv[t - o] : 0;
            __syncthreads();
            sv[t] += av;
            __syncthreads();
        }
        if (g < groups) off[g] = (unsigned)(base + sv[t] - c);
        base += sv[1023];
        __syncthreads();
    }
    if (t == 0) *total = base;
}

// the sites of a class in site order: list[off[g] ..] = the set bits of mask[g]
__global__ __launch_bounds__(256) void class_list_kernel(const uint4 *__restrict__ mask, const unsigned *__restrict__ off, size_t groups,
                                                         unsigned *__restrict__ list)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const uint4 m4 = mask[g];
    const unsigned m[4] = {m4.x, m4.y, m4.z, m4.w};
    unsigned o = off[g];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned x = m[w];
        while (x) {
            const unsigned b = __ffs(x) - 1;
            list[o++] = (unsigned)(g * SITES_PER_GROUP + w * 32 + b);
            x &= x - 1;
        }
    }
}

// Re-pack: one thread = one (sample, OUTPUT group of 128 listed sites), lanes over samples like pack_kernel, so the source
// site of every output bit is wave-uniform.  NPO output planes; out plane k = source plane `first_plane + k` of a source with
// `gp_src` planes per group; `invert`: store the complement of the (single) source plane at the listed sites ("is a base" from N).
template <int NPO>
__global__ __launch_bounds__(256) void compact_sites_kernel(const uint4 *__restrict__ src, int gp_src, int first_plane, bool invert,
                                                            const unsigned *__restrict__ list, unsigned count, uint4 *__restrict__ dst,
                                                            size_t n_pad, unsigned n, unsigned groups_dst)
{
    const unsigned s = blockIdx.y * 64 + (threadIdx.x & 63);
    const unsigned G = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (G >= groups_dst) return;
    const unsigned *__restrict__ srcw = reinterpret_cast<const unsigned *>(src);
    unsigned out[NPO][4];
    unsigned cur[NPO];
#pragma unroll
    for (int p = 0; p < NPO; p++) cur[p] = 0;
    unsigned cw = 0xFFFFFFFFu;
    const unsigned t0 = G * SITES_PER_GROUP;
#pragma unroll
    for (int ow = 0; ow < 4; ow++) {
        unsigned accw[NPO];
#pragma unroll
        for (int p = 0; p < NPO; p++) accw[p] = 0;
        const unsigned tb = t0 + ow * 32;
        const unsigned kn = tb >= count ? 0u : min(32u, count - tb);
        for (unsigned k = 0; k < kn; k++) {
            const unsigned site = __builtin_amdgcn_readfirstlane(list[tb + k]);
            const unsigned w = site >> 5;
            if (w != cw) {                                      // wave-uniform
                cw = w;
#pragma unroll
                for (int p = 0; p < NPO; p++) {
                    const unsigned x = srcw[(((size_t)(site >> 7) * gp_src + first_plane + p) * n_pad + s) * 4 + (w & 3u)];
                    cur[p] = invert ? ~x : x;
                }
            }
#pragma unroll
            for (int p = 0; p < NPO; p++) accw[p] |= ((cur[p] >> (site & 31u)) & 1u) << k;
        }
#pragma unroll
        for (int p = 0; p < NPO; p++) out[p][ow] = accw[p];
    }
    if (s < n)
#pragma unroll
        for (int p = 0; p < NPO; p++)
            dst[((size_t)G * NPO + p) * n_pad + s] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
}

// per sample: set bits of its one-plane row (lanes over samples: coalesced; grid.y cuts the groups, partial counts are added)
__global__ __launch_bounds__(256) void plane_popcount_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, size_t groups,
                                                             unsigned *__restrict__ out)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const size_t per = (groups + gridDim.y - 1) / gridDim.y;
    const size_t g0 = blockIdx.y * per, g1 = min(groups, g0 + per);
    unsigned c = 0;
    for (size_t g = g0; g < g1; g++) {
        const uint4 v = P[g * n_pad + s];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    if (c) atomicAdd(&out[s], c);
}

void site_classes_free(tracs_alignment *a)
{
    if (a->c_counted) (void)hipFree(a->c_counted);
    a->c_counted = nullptr;
    if (a->vplanes) (void)hipFree(a->vplanes);
    if (a->iplanes) (void)hipFree(a->iplanes);
    a->vplanes = a->iplanes = nullptr;
    a->L_var = a->L_inv = a->groups_var = a->groups_inv = 0;
    a->L_minor = a->L_full = 0;
    minority_lists_free(a);
    a->classes_state = 0;
}

static size_t class_plane_bytes(const tracs_alignment *a, size_t groups, int planes, int pad_groups)
{
    return ((groups + pad_groups) * (size_t)planes * a->n_pad + TAIL_PAD) * sizeof(uint4);
}

// Decides (once per pack) whether the pair kernels run on site classes and builds the re-packed alignments and lists if so.
// `consensus`: the source is a->cplanes (3 planes), else a->planes (5).  Soft-fails (classes_state = -1) when memory is short.
static int decide(tracs_alignment *a, bool consensus, bool allow_minor, hipStream_t stream)
{
    a->classes_state = -1;
    static const int force = [] { const char *e = std::getenv("TRACS_SITE_CLASSES"); return e ? std::atoi(e) : -1; }();
    static const bool no_minor = [] { const char *e = std::getenv("TRACS_MINORITY"); return e && std::atoi(e) == 0; }();
    if (force == 0 || a->L == 0 || a->L >= (1ull << 32) || a->n < 2) return TRACS_OK;
    const uint4 *src = consensus ? a->cplanes : a->planes;
    if (!src) return TRACS_OK;
    // TRACS_CLASSES_TRACE=1: wall time of every stage on stderr (synchronises after each: diagnostics only)
    static const bool trace = std::getenv("TRACS_CLASSES_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[site classes] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (trace) (void)hipStreamSynchronize(stream);
    t_last = std::chrono::steady_clock::now();
    const size_t groups = a->groups;
    // sample blocks in grid.y of the re-pack kernels (<= 65535 x 64 samples per launch; beyond that the classes are not used)
    const unsigned sblocks = (unsigned)(a->n_pad / 64);
    if (sblocks > 65535u) return TRACS_OK;
    uint4 *masks = nullptr;
    unsigned *offs = nullptr, *lists = nullptr;
    unsigned long long *totals = nullptr;
    auto cleanup = [&]() {
        void *p[] = {masks, offs, lists, totals};
        for (void *q : p) if (q) (void)hipFree(q);
    };
    auto soft_fail = [&]() { cleanup(); (void)hipGetLastError(); site_classes_free(a); a->classes_state = -1; return TRACS_OK; };
    if (hipMalloc(reinterpret_cast<void **>(&masks), 6 * groups * sizeof(uint4)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&offs), 4 * groups * sizeof(unsigned)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&totals), 32) != hipSuccess) return soft_fail();
    uint4 *dense_mask = masks, *count_mask = masks + groups, *minor_mask = masks + 2 * groups, *full_mask = masks + 3 * groups;
    uint4 *ref_x = masks + 4 * groups, *ref_y = masks + 5 * groups;
    unsigned *off_dense = offs, *off_count = offs + groups, *off_minor = offs + 2 * groups, *off_full = offs + 3 * groups;
    // a site goes to the lists while its k (cN + k) entries cost less than the extra operand planes over all pairs:
    // ~51 ns per site at 10 000 samples (pair kernel) against ~2-4 ps per list entry (general_fixup_kernel)
    const double bsites = (double)a->n * (double)a->n / 8000.0;
    const unsigned budget = (no_minor || !allow_minor) ? 0u : (unsigned)std::min(1.0e9, std::max(16.0, bsites));
    if (consensus)
        hipLaunchKernelGGL((classify_sites_kernel<true>), dim3((unsigned)groups), dim3(256), 0, stream, src, a->n_pad, (unsigned)a->n, budget,
                           dense_mask, count_mask, minor_mask, full_mask, ref_x, ref_y);
    else
        hipLaunchKernelGGL((classify_sites_kernel<false>), dim3((unsigned)groups), dim3(256), 0, stream, src, a->n_pad, (unsigned)a->n, budget,
                           dense_mask, count_mask, minor_mask, full_mask, ref_x, ref_y);
    stage("classify");
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(1024), 0, stream, dense_mask, groups, off_dense, totals + 0);
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(1024), 0, stream, count_mask, groups, off_count, totals + 1);
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(1024), 0, stream, minor_mask, groups, off_minor, totals + 2);
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(1024), 0, stream, full_mask, groups, off_full, totals + 3);    // total only
    unsigned long long tot[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(tot, totals, 32, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        cleanup();
        TRACS_HIP_CHECK(hipGetLastError());
        set_error("site_classes_decide: classification failed");
        return TRACS_E_HIP;
    }
    stage("offsets");
    const size_t L_dense = (size_t)tot[0], L_count = (size_t)tot[1], L_minor = (size_t)tot[2], L_full = (size_t)tot[3];
    // matrix instructions per pair: planes_full per site now; planes_full per dense site + one per counted site with classes
    const double planes_full = consensus ? 4.0 : 5.0;
    const double cost = (planes_full * (double)L_dense + (double)L_count) / (planes_full * (double)a->L);
    if (force != 1 && cost >= 0.92) { cleanup(); return TRACS_OK; }

    const int npv = consensus ? 3 : NPLANES;
    const size_t gv = groups_for(L_dense), gi = groups_for(L_count), gm = L_minor;
    const size_t vbytes = class_plane_bytes(a, gv, npv, PAD_GROUPS), ibytes = class_plane_bytes(a, gi, 1, COUNT_PAD_GROUPS);
    if (hipMalloc(reinterpret_cast<void **>(&lists), (L_dense + L_count + 1) * sizeof(unsigned)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&a->vplanes), vbytes) != hipSuccess) { a->vplanes = nullptr; return soft_fail(); }
    if (hipMalloc(reinterpret_cast<void **>(&a->iplanes), ibytes) != hipSuccess) { a->iplanes = nullptr; return soft_fail(); }
    unsigned *list_dense = lists, *list_count = lists + L_dense;
    bool ok = hipMemsetAsync(a->vplanes, 0, vbytes, stream) == hipSuccess && hipMemsetAsync(a->iplanes, 0, ibytes, stream) == hipSuccess;
    const dim3 lgrid((unsigned)((groups + 255) / 256));
    hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, dense_mask, off_dense, groups, list_dense);
    hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, count_mask, off_count, groups, list_count);
    stage("alloc + site lists");
    if (gv) {
        const dim3 grid((unsigned)((gv + 3) / 4), sblocks);
        if (consensus)
            hipLaunchKernelGGL((compact_sites_kernel<3>), grid, dim3(256), 0, stream, src, 3, 0, false, list_dense, (unsigned)L_dense, a->vplanes,
                               a->n_pad, (unsigned)a->n, (unsigned)gv);
        else
            hipLaunchKernelGGL((compact_sites_kernel<NPLANES>), grid, dim3(256), 0, stream, src, NPLANES, 0, false, list_dense, (unsigned)L_dense,
                               a->vplanes, a->n_pad, (unsigned)a->n, (unsigned)gv);
    }
    stage("re-pack dense");
    if (gi) {
        const dim3 grid((unsigned)((gi + 3) / 4), sblocks);
        // the N plane of the counted sites: consensus: the complement of plane 2 = V; general: plane 4 = N
        // (TRACS_COUNT_COMPLEMENT=1, diagnostics: the "is a base here" plane instead -- same counts, 99 % ones instead of 99 % zeros)
        static const bool complement = std::getenv("TRACS_COUNT_COMPLEMENT") != nullptr;
        a->count_complement = complement;
        hipLaunchKernelGGL((compact_sites_kernel<1>), grid, dim3(256), 0, stream, src, consensus ? 3 : NPLANES, consensus ? 2 : 4,
                           consensus != complement, list_count, (unsigned)L_count, a->iplanes, a->n_pad, (unsigned)a->n, (unsigned)gi);
        if (hipMalloc(reinterpret_cast<void **>(&a->c_counted), a->n_pad * sizeof(unsigned)) != hipSuccess) { a->c_counted = nullptr; return soft_fail(); }
        ok = ok && hipMemsetAsync(a->c_counted, 0, a->n_pad * sizeof(unsigned), stream) == hipSuccess;
        hipLaunchKernelGGL(plane_popcount_kernel, dim3((unsigned)((a->n + 255) / 256), 128), dim3(256), 0, stream, a->iplanes, a->n_pad, (unsigned)a->n,
                           gi, a->c_counted);
    }
    stage("re-pack counted");
    if (gm) {
        // the lists of the minority sites, read in place from the planes (general_sparse.hip, MinorSrc / GeneralMinorSrc)
        int built = 0;
        const int rc = minority_lists_build(a, consensus, src, minor_mask, ref_x, ref_y, off_minor, L_minor, stream, &built);
        stage("minority lists");
        if (rc) { cleanup(); site_classes_free(a); a->classes_state = -1; return rc; }
        if (!built) {                                          // lists too large / no memory: the same classes without them
            soft_fail();
            return decide(a, consensus, false, stream);
        }
    }
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    cleanup();
    if (!ok) { site_classes_free(a); a->classes_state = -1; set_error("site_classes_decide: re-pack failed"); return TRACS_E_HIP; }
    a->L_var = L_dense; a->L_inv = L_count; a->groups_var = gv; a->groups_inv = gi;
    a->L_minor = L_minor; a->L_full = L_full;
    a->classes_state = 1;
    return TRACS_OK;
}

int site_classes_decide(tracs_alignment *a, bool consensus, hipStream_t stream)
{
    if (a->classes_state != 0) return TRACS_OK;
    return decide(a, consensus, true, stream);
}

}  // namespace tracs
