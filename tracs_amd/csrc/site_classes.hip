// site_classes.hip -- variable / invariant site classes of a packed alignment, decided once per pack.
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   pair loop :395-420   d = L - popcount(match),  nn = L - popcount(Ni | Nj): every site is visited for every pair.
//
// A site at which every sample that is not N carries the SAME base never separates two samples: it adds 0 to d(i, j) and
// [neither i nor j is N there] to nn(i, j), whatever the pair.  Real alignments are mostly such sites.  So the sites are cut
// into three classes once per pack,
//     variable    two samples carry different bases (or, general encoding, some sample carries a partial IUPAC code),
//     invariant   not variable, and at least one sample is a base there,
//     empty       every sample is N (or the tail bits behind L): contributes to nothing,
// the alignment is re-packed per class -- `vplanes`: the variable sites in site order, same planes and layout as the pair
// kernels' usual source; `iplanes`: ONE plane, v = "this sample is a base here", over the invariant sites -- and a pass becomes
//     d, nn_var   the usual pair kernel over vplanes               (4 or 5 operand planes per site)
//     nn += nn_inv = sum v_i v_j                                   (1 operand plane per site: pairsnp_mfma_kernel<COUNT>)
// which is exact (the identity holds site by site; tests/test_host_logic.py::test_site_class_identity) and costs
// (4 L_var + L_inv) / 4 L of the dense pass in the consensus form.  Chosen when that is < 0.92; TRACS_SITE_CLASSES=0/1 forces.
#include "pairsnp_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace tracs {

__device__ __forceinline__ unsigned wave_or(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}

// One workgroup per 128-site group: OR-reductions over the samples, then the two class masks of the group.
// CONS: planes X, Y, V (3 per group).  !CONS: planes A, C, G, T, N (5 per group).
template <bool CONS>
__global__ __launch_bounds__(256) void classify_sites_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n,
                                                             uint4 *__restrict__ var_mask, uint4 *__restrict__ inv_mask)
{
    const size_t g = blockIdx.x;
    constexpr int NACC = 5;
    // CONS: acc = {V&X, V&~X, V&Y, V&~Y, V};  general: {A, C, G, T (each & ~N), partial | -- see below}
    unsigned acc[NACC][4], anyb[4];
#pragma unroll
    for (int k = 0; k < NACC; k++)
#pragma unroll
        for (int w = 0; w < 4; w++) acc[k][w] = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) anyb[w] = 0;
    for (unsigned s = threadIdx.x; s < n; s += 256) {
        if constexpr (CONS) {
            const uint4 X = P[(g * 3 + 0) * n_pad + s], Y = P[(g * 3 + 1) * n_pad + s], V = P[(g * 3 + 2) * n_pad + s];
            const unsigned x[4] = {X.x, X.y, X.z, X.w}, y[4] = {Y.x, Y.y, Y.z, Y.w}, v[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                acc[0][w] |= v[w] & x[w]; acc[1][w] |= v[w] & ~x[w];
                acc[2][w] |= v[w] & y[w]; acc[3][w] |= v[w] & ~y[w];
                anyb[w] |= v[w];
            }
        } else {
            const uint4 A = P[(g * NPLANES + 0) * n_pad + s], C = P[(g * NPLANES + 1) * n_pad + s];
            const uint4 G = P[(g * NPLANES + 2) * n_pad + s], T = P[(g * NPLANES + 3) * n_pad + s];
            const uint4 N = P[(g * NPLANES + 4) * n_pad + s];
            const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w}, gg[4] = {G.x, G.y, G.z, G.w};
            const unsigned t[4] = {T.x, T.y, T.z, T.w}, nn[4] = {N.x, N.y, N.z, N.w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned k = ~nn[w];
                const unsigned two = (a[w] & c[w]) | (a[w] & gg[w]) | (a[w] & t[w]) | (c[w] & gg[w]) | (c[w] & t[w]) | (gg[w] & t[w]);
                acc[0][w] |= a[w] & k; acc[1][w] |= c[w] & k; acc[2][w] |= gg[w] & k; acc[3][w] |= t[w] & k;
                acc[4][w] |= two & k;                           // a partial IUPAC code: the site is variable
                anyb[w] |= (a[w] | c[w] | gg[w] | t[w]) & k;
            }
        }
    }
    __shared__ unsigned red[4][NACC + 1][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) {
#pragma unroll
        for (int k = 0; k < NACC; k++) {
            const unsigned r = wave_or(acc[k][w]);
            if (lane == 0) red[wave][k][w] = r;
        }
        const unsigned r = wave_or(anyb[w]);
        if (lane == 0) red[wave][NACC][w] = r;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int w = threadIdx.x;
        unsigned t[NACC + 1];
#pragma unroll
        for (int k = 0; k <= NACC; k++) t[k] = red[0][k][w] | red[1][k][w] | red[2][k][w] | red[3][k][w];
        unsigned var;
        if constexpr (CONS) var = (t[0] & t[1]) | (t[2] & t[3]);
        else var = t[4] | (t[0] & t[1]) | (t[0] & t[2]) | (t[0] & t[3]) | (t[1] & t[2]) | (t[1] & t[3]) | (t[2] & t[3]);
        reinterpret_cast<unsigned *>(&var_mask[g])[w] = var;
        reinterpret_cast<unsigned *>(&inv_mask[g])[w] = t[NACC] & ~var;
    }
}

// exclusive prefix sums of the per-group class sizes (one workgroup walks the groups 1024 at a time); totals[0..1]
__global__ __launch_bounds__(1024) void class_offsets_kernel(const uint4 *__restrict__ var_mask, const uint4 *__restrict__ inv_mask,
                                                             size_t groups, unsigned *__restrict__ off_var, unsigned *__restrict__ off_inv,
                                                             unsigned long long *__restrict__ totals)
{
    __shared__ unsigned sv[1024], si[1024];
    unsigned long long base_v = 0, base_i = 0;
    const int t = threadIdx.x;
    for (size_t g0 = 0; g0 < groups; g0 += 1024) {
        const size_t g = g0 + t;
        unsigned cv = 0, ci = 0;
        if (g < groups) {
            const uint4 a = var_mask[g], b = inv_mask[g];
            cv = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
            ci = __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
        }
        sv[t] = cv; si[t] = ci;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned av = t >= off ? sv[t - off] : 0, ai = t >= off ? si[t - off] : 0;
            __syncthreads();
            sv[t] += av; si[t] += ai;
            __syncthreads();
        }
        if (g < groups) {
            off_var[g] = (unsigned)(base_v + sv[t] - cv);
            off_inv[g] = (unsigned)(base_i + si[t] - ci);
        }
        base_v += sv[1023]; base_i += si[1023];
        __syncthreads();
    }
    if (t == 0) { totals[0] = base_v; totals[1] = base_i; }
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

void site_classes_free(tracs_alignment *a)
{
    if (a->vplanes) (void)hipFree(a->vplanes);
    if (a->iplanes) (void)hipFree(a->iplanes);
    a->vplanes = a->iplanes = nullptr;
    a->L_var = a->L_inv = a->groups_var = a->groups_inv = 0;
    a->classes_state = 0;
}

static size_t class_plane_bytes(const tracs_alignment *a, size_t groups, int planes, int pad_groups)
{
    return ((groups + pad_groups) * (size_t)planes * a->n_pad + TAIL_PAD) * sizeof(uint4);
}

// Decides (once per pack) whether the pair kernels run on site classes and builds the two re-packed alignments if so.
// `consensus`: the source is a->cplanes (3 planes), else a->planes (5).  Soft-fails (classes_state = -1) when memory is short.
int site_classes_decide(tracs_alignment *a, bool consensus, hipStream_t stream)
{
    if (a->classes_state != 0) return TRACS_OK;
    a->classes_state = -1;
    static const int force = [] { const char *e = std::getenv("TRACS_SITE_CLASSES"); return e ? std::atoi(e) : -1; }();
    if (force == 0 || a->L == 0 || a->L >= (1ull << 32) || a->n < 2) return TRACS_OK;
    const uint4 *src = consensus ? a->cplanes : a->planes;
    if (!src) return TRACS_OK;
    const size_t groups = a->groups;
    uint4 *masks = nullptr;
    unsigned *offs = nullptr, *lists = nullptr;
    unsigned long long *totals = nullptr;
    auto cleanup = [&]() {
        if (masks) (void)hipFree(masks);
        if (offs) (void)hipFree(offs);
        if (lists) (void)hipFree(lists);
        if (totals) (void)hipFree(totals);
    };
    auto soft_fail = [&]() { cleanup(); (void)hipGetLastError(); site_classes_free(a); a->classes_state = -1; return TRACS_OK; };
    if (hipMalloc(reinterpret_cast<void **>(&masks), 2 * groups * sizeof(uint4)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&offs), 2 * groups * sizeof(unsigned)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&totals), 16) != hipSuccess) return soft_fail();
    uint4 *var_mask = masks, *inv_mask = masks + groups;
    unsigned *off_var = offs, *off_inv = offs + groups;
    if (consensus)
        hipLaunchKernelGGL((classify_sites_kernel<true>), dim3((unsigned)groups), dim3(256), 0, stream, src, a->n_pad, (unsigned)a->n, var_mask, inv_mask);
    else
        hipLaunchKernelGGL((classify_sites_kernel<false>), dim3((unsigned)groups), dim3(256), 0, stream, src, a->n_pad, (unsigned)a->n, var_mask, inv_mask);
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(1024), 0, stream, var_mask, inv_mask, groups, off_var, off_inv, totals);
    unsigned long long tot[2] = {0, 0};
    if (hipMemcpyAsync(tot, totals, 16, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
        cleanup();
        TRACS_HIP_CHECK(hipGetLastError());
        set_error("site_classes_decide: classification failed");
        return TRACS_E_HIP;
    }
    const size_t L_var = (size_t)tot[0], L_inv = (size_t)tot[1];
    // matrix instructions per pair: planes_full per site now; planes_full per variable site + one per invariant site with classes
    const double planes_full = consensus ? 4.0 : 5.0;
    const double cost = (planes_full * (double)L_var + (double)L_inv) / (planes_full * (double)a->L);
    if ((force != 1 && cost >= 0.92) || (!consensus && L_var == 0)) { cleanup(); return TRACS_OK; }

    const int npv = consensus ? 3 : NPLANES;
    const size_t gv = groups_for(L_var), gi = groups_for(L_inv);
    const size_t vbytes = class_plane_bytes(a, gv, npv, PAD_GROUPS), ibytes = class_plane_bytes(a, gi, 1, COUNT_PAD_GROUPS);
    if (hipMalloc(reinterpret_cast<void **>(&lists), (L_var + L_inv + 1) * sizeof(unsigned)) != hipSuccess) return soft_fail();
    if (hipMalloc(reinterpret_cast<void **>(&a->vplanes), vbytes) != hipSuccess) { a->vplanes = nullptr; return soft_fail(); }
    if (hipMalloc(reinterpret_cast<void **>(&a->iplanes), ibytes) != hipSuccess) { a->iplanes = nullptr; return soft_fail(); }
    unsigned *list_var = lists, *list_inv = lists + L_var;
    bool ok = hipMemsetAsync(a->vplanes, 0, vbytes, stream) == hipSuccess && hipMemsetAsync(a->iplanes, 0, ibytes, stream) == hipSuccess;
    const dim3 lgrid((unsigned)((groups + 255) / 256));
    hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, var_mask, off_var, groups, list_var);
    hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, inv_mask, off_inv, groups, list_inv);
    // sample blocks in grid.y (<= 65535 x 64 samples per launch: n_pad < 2^22; beyond that the classes are not used)
    const unsigned sblocks = (unsigned)(a->n_pad / 64);
    if (sblocks > 65535u) return soft_fail();
    if (gv) {
        const dim3 grid((unsigned)((gv + 3) / 4), sblocks);
        if (consensus)
            hipLaunchKernelGGL((compact_sites_kernel<3>), grid, dim3(256), 0, stream, src, 3, 0, false, list_var, (unsigned)L_var, a->vplanes,
                               a->n_pad, (unsigned)a->n, (unsigned)gv);
        else
            hipLaunchKernelGGL((compact_sites_kernel<NPLANES>), grid, dim3(256), 0, stream, src, NPLANES, 0, false, list_var, (unsigned)L_var,
                               a->vplanes, a->n_pad, (unsigned)a->n, (unsigned)gv);
    }
    if (gi) {
        const dim3 grid((unsigned)((gi + 3) / 4), sblocks);
        // consensus: plane 2 = V.  general: the complement of plane 4 = N (an invariant site holds bases and N only)
        hipLaunchKernelGGL((compact_sites_kernel<1>), grid, dim3(256), 0, stream, src, consensus ? 3 : NPLANES, consensus ? 2 : 4, !consensus,
                           list_inv, (unsigned)L_inv, a->iplanes, a->n_pad, (unsigned)a->n, (unsigned)gi);
    }
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    cleanup();
    if (!ok) { site_classes_free(a); a->classes_state = -1; set_error("site_classes_decide: re-pack failed"); return TRACS_E_HIP; }
    a->L_var = L_var; a->L_inv = L_inv; a->groups_var = gv; a->groups_inv = gi;
    a->classes_state = 1;
    return TRACS_OK;
}

}  // namespace tracs
