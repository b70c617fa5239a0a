// site_classes.hip -- site classes of a packed alignment and the encoding decision, made together once per pack.
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   pair loop :395-420   d = L - popcount(match),  nn = L - popcount(Ni | Nj): every site is visited for every pair.
//
// A site at which every sample that is not N carries the SAME base never separates two samples: it adds 0 to d(i, j) and
// [neither i nor j is N there] to nn(i, j), whatever the pair; and a site at which only a FEW samples differ from the others
// separates only the pairs that involve one of those few.  Real alignments are mostly such sites.  The sites are cut into
// classes once per pack (k = samples that are neither N nor exactly the site's reference base -- the base of its lowest-index
// one-base sample --, i.e. another base or a partial IUPAC code; cN = samples that are N there):
//     empty      every sample is N (or the tail bits behind L): contributes to nothing;
//     dense      k (cN + k) above the budget below: the usual pair kernel, over `vplanes` -- these sites re-packed in site
//                order, in the encoding the pair kernel reads (consensus X, Y, V when no sample carries a partial IUPAC code
//                anywhere, else the five general planes);
//     counted    every other site with cN >= 1: NN = sum n_i n_j (ONE plane, n = "this sample is N here";
//                pairsnp_mfma_kernel<COUNT>, one operand plane instead of four or five) and nn = sites - c_i - c_j + NN;
//     full       every other site with cN = 0: +1 to every nn, a constant;
//     minority   the counted / full sites with k >= 1: d gets their contribution from sparse lists -- the k samples with their
//                allele masks, the cN samples -- with the machinery general_sparse.hip uses for partial IUPAC codes
//                (general_fixup_kernel<MINOR>: [reference base not in the listed sample's mask] for a pair of a listed sample
//                and a reference-base sample, [masks disjoint] for two listed samples, 0 when either is N).
// The decomposition is exact site by site (tests/test_host_logic.py::test_site_class_identity); a pass costs
// (4 L_dense + L_counted) / 4 L of the dense one in matrix instructions plus ~sum k (cN + k) list entries.  Chosen when that
// is < 0.92; TRACS_SITE_CLASSES=0/1 forces, TRACS_MINORITY=0 keeps every site with k >= 1 dense.
//
// What a pack costs (round 3: one source, no second copy of the alignment).  Everything is read from the five general planes
// load_seqs builds -- the consensus copy of round 2 (a 3/5-size second alignment, derived before classifying and dropped
// afterwards) only exists when the classes are NOT used:
//     classify_sites_kernel   one read of the planes: reference bases from the first samples that resolve every site, then
//                             k and cN of every site with BIT-SLICED counters in registers (a carry-save adder per 32-site
//                             word: data-independent, no LDS atomics -- a uniformly random alignment costs what an invariant
//                             one does), the class masks, "some sample carries a partial code" (= the encoding decision),
//                             per sample and group "listed somewhere here" (64 samples per word: what the list builder may
//                             skip), per group the list sizes of its minority sites;
//     counting pass source    the stored N plane IN PLACE (stride 5 planes) when nearly every site needs counting -- then
//                             nn = L - c_i - c_j + NN over ALL sites and the pair kernel writes d only --, else the counted
//                             sites' N plane re-packed into `iplanes` (runs of consecutive sites are funnel-shifted, not gathered
//                             bit by bit);
//     minority lists          general_sparse.hip: per-site lists from the N plane + the flagged samples only, per-sample
//                             lists from the N plane: 3 reads of ONE plane instead of 4 reads of the alignment.
#include "pairsnp_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace tracs {

struct GroupWords { unsigned x[4], y[4], one[4], some[4], isn[4], bad[4]; };

// what pass 2 needs of a sample's 32-site word: diff = listed (not N and not exactly the reference base, given one-hot in ra..rt),
// isn = N, some = a base or a partial code, bad = a partial IUPAC code -- 3-input boolean instructions throughout
__device__ __forceinline__ void classify_word(unsigned a, unsigned c, unsigned g, unsigned t, unsigned ra, unsigned rc, unsigned rg, unsigned rt,
                                              unsigned &diff, unsigned &isn, unsigned &some, unsigned &bad)
{
    isn = __builtin_amdgcn_bitop3_b32(a, c, g, 0x80) & t;                                  // a & c & g & t
    const unsigned any = __builtin_amdgcn_bitop3_b32(a, c, g, 0xFE) | t;
    some = any & ~isn;
    unsigned d = a ^ ra;
    d = __builtin_amdgcn_bitop3_b32(c, rc, d, 0xBE);                                       // (c ^ rc) | d
    d = __builtin_amdgcn_bitop3_b32(g, rg, d, 0xBE);
    d = __builtin_amdgcn_bitop3_b32(t, rt, d, 0xBE);
    diff = some & d;
    // two or more alleles: (a & c) | (g & t) | ((a | c) & (g | t))
    unsigned two = __builtin_amdgcn_bitop3_b32(g, t, a & c, 0xEA);                         // (g & t) | (a & c)
    two = __builtin_amdgcn_bitop3_b32(a | c, g | t, two, 0xEA);
    bad = two & ~isn;
}

// per sample and 32-site word of group g: x, y = the base's two bits where the sample carries exactly one base (`one`),
// `some` = a base or a partial code (not N; tail bits: neither), isn = N, bad = a partial IUPAC code (2 or 3 alleles)
__device__ __forceinline__ void load_group_words(const uint4 *__restrict__ P, size_t n_pad, size_t g, unsigned s, GroupWords &q)
{
    const uint4 *base = P + (g * NPLANES) * n_pad + s;
    // (the stored N plane is A & C & G & T by construction -- pack_kernel, pack_codes_kernel --: four planes are read, not five)
    const uint4 A = base[0], C = base[n_pad], G = base[2 * n_pad], T = base[3 * n_pad];
    const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w}, gg[4] = {G.x, G.y, G.z, G.w};
    const unsigned t[4] = {T.x, T.y, T.z, T.w};
    const unsigned nn[4] = {a[0] & c[0] & gg[0] & t[0], a[1] & c[1] & gg[1] & t[1], a[2] & c[2] & gg[2] & t[2], a[3] & c[3] & gg[3] & t[3]};
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned two = (a[w] & c[w]) | (a[w] & gg[w]) | (a[w] & t[w]) | (c[w] & gg[w]) | (c[w] & t[w]) | (gg[w] & t[w]);
        const unsigned any = a[w] | c[w] | gg[w] | t[w];
        q.one[w] = any & ~two;
        q.some[w] = any & ~nn[w];
        q.isn[w] = nn[w];
        q.bad[w] = two & ~nn[w];
        q.x[w] = (c[w] | t[w]) & q.one[w];
        q.y[w] = (gg[w] | t[w]) & q.one[w];
    }
}

// One workgroup per 128-site group, threads over samples (coalesced 16-byte loads per plane).
//   pass 1  reference base per site = the base of the LOWEST-INDEX sample that carries exactly one base there; samples are
//           read 256 at a time until every site of the group is resolved (normally the first 256: 2.5 % of a pass at 10 000
//           samples; a group with an all-N site reads everybody);
//   pass 2  k (listed: not N and not exactly the reference base) and cN (N) per site over all samples, bit-sliced in
//           registers, flushed through LDS every 255 samples per thread; partial-code flag; per (64 samples, group) the
//           samples that are listed somewhere in the group.
// Outputs per group: the class masks, the reference base bits, k and cN of every site, the list sizes of its sites that
// carry lists (gP = sum of k over the minority sites, gN = sum of cN over the minority and NNL sites).
// Mask slots of a group (masks[slot * groups + g], one uint4 = 128 sites each):
enum { M_DENSE = 0, M_COUNT, M_MINOR, M_FULL, M_NNL, M_LST, M_UN, M_REFX, M_REFY, M_SLOTS };
//   M_COUNT  sites whose N co-occurrences go through the matrix cores (cN > nn_list_max, or lists not in use)
//   M_NNL    sites whose N co-occurrences come from their N lists (cN >= 2, cN x the lines of the list <= nn_list_max: nn_rows_kernel, site_lists.hip)
//   M_LST    sites that carry lists at all (minority or NNL): list index = rank among these
//   M_UN     every site outside the dense class with cN >= 1 (M_COUNT, M_NNL and the cN = 1 sites, which have no co-occurrence)
// gram != 0 (the N x listed terms of the minority sites on the matrix cores, decide() below): a site's cost no longer grows with its
// N samples -- minority while k^2 <= budget --, no site carries an N list, every N co-occurrence is counted on the matrix cores
// NT threads per workgroup: 128 (two waves), 64 for alignments of at most 2 048 samples; TRACS_CLASSIFY_THREADS=64|128|256 forces (256: the
// four waves of rounds 3 - 6a).  As workgroups of two waves a CU overlaps
// one group's flushes and barriers with another group's loads better than as workgroups of four: 4.4 - 4.55 -> 4.0 - 4.4 ms at 10 000 x 5 Mbp with eight samples per step (one wave
// per workgroup: 4.4 again); at 1 000 x 1 Mbp, where a group's fixed costs are what counts, 0.275 (256) -> 0.19 (128) -> 0.17 ms (64): config 2
// 1.22 -> 1.07 ms per call (profiles/r06/classify_threads.txt).
// Four samples per thread and step at three waves per SIMD (170 VGPRs as compiled, held to 168): 4.4 -> 4.2 ms at 10 000 x 5 Mbp against
// eight samples at two waves (213 VGPRs) -- the sliced adds cost twice the instructions per sample, which this kernel has to spare; four
// waves spill: 7.6 ms (profiles/r06/classify_threads.txt).  -DTRACS_CLASSIFY_SPS=8 -DTRACS_CLASSIFY_WAVES=0: the round-5 shape.
#ifndef TRACS_CLASSIFY_SPS
#define TRACS_CLASSIFY_SPS 4
#endif
#ifndef TRACS_CLASSIFY_WAVES
#define TRACS_CLASSIFY_WAVES (TRACS_CLASSIFY_SPS == 4 ? 3 : 0)
#endif
#if TRACS_CLASSIFY_WAVES > 0
#define TRACS_CLASSIFY_ATTR __attribute__((amdgpu_waves_per_eu(TRACS_CLASSIFY_WAVES, TRACS_CLASSIFY_WAVES)))
#else
#define TRACS_CLASSIFY_ATTR
#endif
template <int NT>
__global__ __launch_bounds__(NT) TRACS_CLASSIFY_ATTR void classify_sites_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, unsigned budget,
                                                             unsigned nn_list_max, unsigned gram, uint4 *__restrict__ masks, size_t groups,
                                                             unsigned *__restrict__ cntP, unsigned *__restrict__ cntN,
                                                             unsigned *__restrict__ gP, unsigned *__restrict__ gN, unsigned *__restrict__ gQ,
                                                             unsigned *__restrict__ gR, unsigned *__restrict__ gS, unsigned *__restrict__ gI,
                                                             unsigned *__restrict__ gJ,
                                                             unsigned long long *__restrict__ flags, size_t flag_words,
                                                             unsigned *__restrict__ partial_flag,
                                                             uint4 *__restrict__ masks2 = nullptr, unsigned *__restrict__ gcnt2 = nullptr)
{
    // masks2 / gcnt2 (may be NULL): the class masks and per-group sums of the SAME pass under the other criterion (gram = 1), so that
    // decide() can take the second form of the classes without classifying again (bit 2 of *partial_flag: some p list of that form is long)
    const size_t g = blockIdx.x;
    constexpr int WAVES = NT / 64;
    static_assert(NT == 64 || NT == 128 || NT == 256, "classify_sites_kernel: one, two or four waves");
    __shared__ unsigned red[WAVES][4][4];
    __shared__ unsigned sref[4][4];                     // one-base sample seen, ref X, ref Y, somebody is not N
    __shared__ unsigned planes_lds[4][8][NT];            // one counter's bit planes of every thread (32 KiB at 256 threads): [word][plane][thread]
    __shared__ unsigned tot[2][SITES_PER_GROUP];        // k, cN
    __shared__ unsigned wsum[2][2][7];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int st = tid; st < (int)SITES_PER_GROUP; st += NT) { tot[0][st] = 0; tot[1][st] = 0; }
    if (tid < 4) { sref[0][tid] = 0; sref[1][tid] = 0; sref[2][tid] = 0; sref[3][tid] = 0; }
    __syncthreads();

    // ---- pass 1: reference bases ----------------------------------------------------------------------------------------
    for (unsigned base = 0; base < n; base += NT) {
        unsigned seen[4] = {0, 0, 0, 0}, rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0};
        const unsigned s = base + tid;
        if (s < n) {
            GroupWords q;
            load_group_words(P, n_pad, g, s, q);
#pragma unroll
            for (int w = 0; w < 4; w++) { seen[w] = q.one[w]; rx[w] = q.x[w]; ry[w] = q.y[w]; }
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
            }
            if (lane == 0) { red[wave][0][w] = seen[w]; red[wave][1][w] = rx[w]; red[wave][2][w] = ry[w]; }
        }
        __syncthreads();
        if (tid < 4) {
            const int w = tid;
            unsigned fs = sref[0][w], fx = sref[1][w], fy = sref[2][w];           // earlier chunks first
            for (int k = 0; k < WAVES; k++) { fx |= red[k][1][w] & ~fs; fy |= red[k][2][w] & ~fs; fs |= red[k][0][w]; }
            sref[0][w] = fs; sref[1][w] = fx; sref[2][w] = fy;
        }
        __syncthreads();
        if ((sref[0][0] & sref[0][1] & sref[0][2] & sref[0][3]) == 0xFFFFFFFFu) break;     // block-uniform
    }
    // (a site without any one-base sample keeps the reference A: every sample that is not N is listed there)
    const unsigned refx[4] = {sref[1][0], sref[1][1], sref[1][2], sref[1][3]}, refy[4] = {sref[2][0], sref[2][1], sref[2][2], sref[2][3]};

    // ---- pass 2: k and cN of every site -----------------------------------------------------------------------------------
    unsigned anyb[4] = {0, 0, 0, 0}, bad = 0;
    // flush: the bit planes of all threads through LDS, one counter at a time.  Wave w takes word w (32 sites): for every plane and
    // every 64 threads, a 32 x 32 bit transpose across the half waves turns "bit b of thread t" into "bit t of lane b", whose
    // popcount is the site's count over those 32 threads -- 18 instructions per 64 words where summing bit by bit took 4 per bit
    // (at 10 000 samples the counters are flushed once per group: the flush was two thirds of the kernel's instructions)
    const Transpose32 tr((unsigned)lane);
    auto flush = [&](unsigned (&pl)[4][8], int which) {
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int j = 0; j < 8; j++) { planes_lds[w][j][tid] = pl[w][j]; pl[w][j] = 0; }
        __syncthreads();
        // (wave w takes words w, w + WAVES, ..)
#pragma unroll
        for (int w = wave; w < 4; w += WAVES) {
            unsigned acc = 0;
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int blk = 0; blk < WAVES; blk++) acc += (unsigned)__popc(tr(planes_lds[w][j][blk * 64 + lane])) << j;
            acc += __shfl_xor(acc, 32, 64);
            if (lane < 32) tot[which][w * 32 + lane] += acc;
        }
    };
    unsigned kp[4][8], np[4][8];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int j = 0; j < 8; j++) { kp[w][j] = 0; np[w][j] = 0; }
    // the reference bases one-hot per 32-site word
    unsigned ra[4], rc[4], rg[4], rt[4];
#pragma unroll
    for (int w = 0; w < 4; w++) { ra[w] = ~refx[w] & ~refy[w]; rc[w] = refx[w] & ~refy[w]; rg[w] = ~refx[w] & refy[w]; rt[w] = refx[w] & refy[w]; }
    // SPS samples per thread and step: their words are classified as they arrive (4 SPS 16-byte loads in flight per thread), then
    // added to the bit-sliced counters SPS at a time (sliced_add8 / sliced_add4).  The counters hold 8 bits: flushed every 31 / 63 steps.
    constexpr int SPS = TRACS_CLASSIFY_SPS;                 // samples per thread and step
    static_assert(SPS == 8 || SPS == 4, "classify_sites_kernel: four or eight samples per step");
    unsigned since = 0;
    for (unsigned base = 0; base < n; base += NT * SPS) {
        unsigned db[SPS][4], nb[SPS][4];
#pragma unroll
        for (int k = 0; k < SPS; k++) {
            const unsigned s = base + k * NT + tid;
            bool listed_here = false;
#pragma unroll
            for (int w = 0; w < 4; w++) { db[k][w] = 0; nb[k][w] = 0; }
            if (s < n) {
                const uint4 *bp = P + (g * NPLANES) * n_pad + s;
                // (the planes stream by once: non-temporal loads -- 4.68 -> 4.42 ms at 10 000 x 5 Mbp, 5.7 TB/s)
                typedef unsigned cls_u32x4 __attribute__((ext_vector_type(4)));
                const cls_u32x4 A = __builtin_nontemporal_load(reinterpret_cast<const cls_u32x4 *>(bp)), C = __builtin_nontemporal_load(reinterpret_cast<const cls_u32x4 *>(bp + n_pad)),
                                G = __builtin_nontemporal_load(reinterpret_cast<const cls_u32x4 *>(bp + 2 * n_pad)), T = __builtin_nontemporal_load(reinterpret_cast<const cls_u32x4 *>(bp + 3 * n_pad));
                const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w}, gg[4] = {G.x, G.y, G.z, G.w}, t[4] = {T.x, T.y, T.z, T.w};
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    unsigned some, bd;
                    classify_word(a[w], c[w], gg[w], t[w], ra[w], rc[w], rg[w], rt[w], db[k][w], nb[k][w], some, bd);
                    anyb[w] |= some;
                    bad |= bd;
                    listed_here = listed_here || db[k][w] != 0u;
                }
            }
            if (base + k * NT < n) {                            // (block-uniform)
                const unsigned long long fl = __ballot(listed_here);
                if (lane == 0) flags[g * flag_words + ((base + k * NT) >> 6) + wave] = fl;
            }
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if constexpr (SPS == 8) {
                const unsigned xd[8] = {db[0][w], db[1][w], db[2][w], db[3][w], db[SPS - 4][w], db[SPS - 3][w], db[SPS - 2][w], db[SPS - 1][w]};
                const unsigned xn[8] = {nb[0][w], nb[1][w], nb[2][w], nb[3][w], nb[SPS - 4][w], nb[SPS - 3][w], nb[SPS - 2][w], nb[SPS - 1][w]};
                sliced_add8(kp[w], xd);
                sliced_add8(np[w], xn);
            } else {
                const unsigned xd[4] = {db[0][w], db[1][w], db[2][w], db[3][w]}, xn[4] = {nb[0][w], nb[1][w], nb[2][w], nb[3][w]};
                sliced_add4(kp[w], xd);
                sliced_add4(np[w], xn);
            }
        }
        if (++since == (SPS == 8 ? 31u : 63u)) { flush(kp, 0); flush(np, 1); since = 0; }     // block-uniform
    }
    if (since) { flush(kp, 0); flush(np, 1); }
    // somebody is not N, per site
#pragma unroll
    for (int w = 0; w < 4; w++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) anyb[w] |= __shfl_xor(anyb[w], off, 64);
        if (lane == 0) red[wave][3][w] = anyb[w];
    }
    const int any_bad = __syncthreads_or(bad != 0u);
    if (tid == 0 && any_bad) atomicOr(partial_flag, 1u);
    // (a site per thread: with one wave per workgroup in two rounds; `hw`: the site's 64-site half)
    for (int st = tid; st < (int)SITES_PER_GROUP; st += NT) {
        const int w = st >> 5, b = st & 31, hw = st >> 6;
        unsigned anyw = 0;
#pragma unroll
        for (int k = 0; k < WAVES; k++) anyw |= red[k][3][w];
        const bool some = (anyw >> b) & 1u;
        const unsigned long long k = tot[0][st], c = some ? tot[1][st] : 0ull;      // (an empty site: every sample is N)
        cntP[g * SITES_PER_GROUP + st] = (unsigned)k;
        cntN[g * SITES_PER_GROUP + st] = (unsigned)c;
#pragma unroll
        for (int form = 0; form < 2; form++) {
            if (form == 1 && masks2 == nullptr) break;                  // (kernel-uniform)
            const bool gr = form == 1 || gram != 0u;
            uint4 *const mk = form == 0 ? masks : masks2;
            const bool minor = some && k >= 1 && budget > 0 && k * ((gr ? 0ull : c) + k) <= (unsigned long long)budget;
            const bool dense = some && k >= 1 && !minor;
            const bool un = some && !dense && c >= 1;
            // (a walk costs the lines of the list: pairsnp_kernels.h, n8 lines)
            const bool nnl = !gr && un && c >= 2 && (float)c * n8_lines_expected((unsigned)c, n) <= (float)nn_list_max;
            const bool counted = un && c >= 2 && !nnl;          // (a site with one N sample has no pair of N samples)
            const bool full = some && !dense && c == 0;
            const bool lst = minor || nnl;
            const bool cls[7] = {dense, counted, minor, full, nnl, lst, un};
            constexpr int slot[7] = {M_DENSE, M_COUNT, M_MINOR, M_FULL, M_NNL, M_LST, M_UN};
#pragma unroll
            for (int m = 0; m < 7; m++) {
                const unsigned long long bal = __ballot(cls[m]);
                if (lane == 0) {
                    unsigned *pm = reinterpret_cast<unsigned *>(&mk[(size_t)slot[m] * groups + g]);
                    pm[2 * hw] = (unsigned)bal; pm[2 * hw + 1] = (unsigned)(bal >> 32);
                }
            }
            // some p list is a long one (q lines: site_lists.hip): bit 1 for the first set of classes, bit 2 for the second
            if (__ballot(minor && k > P_SHORT_MAX) && lane == 0) atomicOr(partial_flag, form == 0 ? 2u : 4u);
            // per group: p-list entries; overflow lines the N lists of its listed sites can need at most (each has its primary line);
            // list entries one pass of the N co-occurrence walk decodes (cN per walk, cN walks) and its walks; N-list walks of the
            // minority fix-up (one per listed sample of a site with an N sample)
            // (gJ: lines a minority site's p list needs beyond its own: 31 dwords a line, the first one of the list its header -- q lines, site_lists.hip)
            unsigned sv[7] = {minor ? (unsigned)k : 0u, (lst && !gr) ? n8_lines_max((unsigned)c, n) - 1u : 0u, nnl ? (unsigned)min(c * c, 33554431ull) : 0u, nnl ? (unsigned)c : 0u,
                              (minor && c && !gr) ? (unsigned)k : 0u, (minor && !gr) ? n8_lines_max((unsigned)c, n) - 1u : 0u, minor ? (unsigned)(k / 31ull) : 0u};
#pragma unroll
            for (int m = 0; m < 7; m++) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sv[m] += __shfl_xor(sv[m], off, 64);
                if (lane == 0) wsum[form][hw][m] = sv[m];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        gP[g] = wsum[0][0][0] + wsum[0][1][0]; gN[g] = wsum[0][0][1] + wsum[0][1][1]; gQ[g] = wsum[0][0][2] + wsum[0][1][2];
        gR[g] = wsum[0][0][3] + wsum[0][1][3]; gS[g] = wsum[0][0][4] + wsum[0][1][4]; gI[g] = wsum[0][0][5] + wsum[0][1][5]; gJ[g] = wsum[0][0][6] + wsum[0][1][6];
    }
    if (tid < 7 && gcnt2 != nullptr) gcnt2[(size_t)tid * groups + g] = wsum[1][0][tid] + wsum[1][1][tid];
    if (tid < 4) {
        reinterpret_cast<unsigned *>(&masks[(size_t)M_REFX * groups + g])[tid] = refx[tid];
        reinterpret_cast<unsigned *>(&masks[(size_t)M_REFY * groups + g])[tid] = refy[tid];
    }
}

// Exclusive prefix sums over the groups, one workgroup per array (1024 groups at a time):
//   blocks 0..6  sizes of the mask slots M_DENSE .. M_UN (popcount of the mask) -> off32[b][g], totals[b]
//   blocks 7..13 per-group sums: gP (p-list entries), gN (overflow lines of the N lists, upper bound), gQ (entries the N co-occurrence
//                walk decodes), gR (its walks), gS (N-list walks of the minority fix-up), gI (overflow lines of the minority sites' N lists
//                alone), gJ (overflow lines of the p lists) -> off64[b - 7][g], totals[b]
__global__ __launch_bounds__(1024) void group_offsets_kernel(const uint4 *__restrict__ masks, const unsigned *__restrict__ gcounts, size_t groups,
                                                             unsigned *__restrict__ off32, unsigned long long *__restrict__ off64,
                                                             unsigned long long *__restrict__ totals)
{
    // 1 024 consecutive groups per step (coalesced loads, the next step's already under way): a wave scan through shuffles, the 16
    // wave totals through LDS -- two barriers per step where a Hillis-Steele scan in LDS took twenty
    __shared__ unsigned long long wave_tot[2][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto count_of = [&](size_t g) -> unsigned long long {
        if (g >= groups) return 0ull;
        if (b < 7) { const uint4 a = masks[(size_t)b * groups + g]; return (unsigned long long)(__popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w)); }
        return gcounts[(size_t)(b - 7) * groups + g];
    };
    unsigned long long base = 0, c_next = count_of((size_t)t), c_max = 0;
    int par = 0;
    for (size_t g0 = 0; g0 < groups; g0 += 1024, par ^= 1) {
        const size_t g = g0 + t;
        const unsigned long long c = c_next;
        c_max = max(c_max, c);
        c_next = count_of(g + 1024);
        unsigned long long incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wave_tot[par][wave] = incl;
        __syncthreads();                                  // (wave_tot[par ^ 1] was last read before the previous step's barrier)
        unsigned long long before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const unsigned long long v = wave_tot[par][w]; if (w < wave) before += v; all += v; }
        if (g < groups) {
            if (b < 7) off32[(size_t)b * groups + g] = (unsigned)(base + before + incl - c);
            else off64[(size_t)(b - 7) * groups + g] = base + before + incl - c;
        }
        base += all;
    }
    if (t == 0) totals[b] = base;
    if (b == 7) {                                            // the most p-list entries any group has: totals[14] (p_lists_kernel is launched for such groups only)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c_max = max(c_max, (unsigned long long)__shfl_xor(c_max, off, 64));
        if (lane == 0 && c_max) atomicMax(&totals[14], c_max);
    }
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
// site of every output bit is wave-uniform.  The source is the five general planes.
//   MODE 0  out = the five planes of the listed sites (general encoding's vplanes)
//   MODE 1  out = consensus planes X, Y, V of the listed sites (A = 00, C = 01, G = 10, T = 11; V = exactly one base)
//   MODE 2  out = the N plane of the listed sites (the counting pass's iplanes)
// A 32-site output word whose listed sites are consecutive takes its bits with one funnel shift per plane; only words that
// straddle a gap gather bit by bit.
template <int MODE>
__global__ __launch_bounds__(256) void compact_sites_kernel(const uint4 *__restrict__ src, const unsigned *__restrict__ list, unsigned count,
                                                            uint4 *__restrict__ dst, size_t n_pad, unsigned n, unsigned groups_dst)
{
    constexpr int NPO = MODE == 0 ? NPLANES : MODE == 1 ? 3 : 1;
    constexpr int NPI = MODE == 2 ? 1 : NPLANES, FIRST = MODE == 2 ? 4 : 0;
    const unsigned s = blockIdx.y * 64 + (threadIdx.x & 63);
    const unsigned G = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (G >= groups_dst) return;
    const unsigned *__restrict__ srcw = reinterpret_cast<const unsigned *>(src);
    auto word_at = [&](unsigned w, int p) -> unsigned {      // 32-site word w (absolute) of input plane p
        return srcw[(((size_t)(w >> 2) * NPLANES + FIRST + p) * n_pad + s) * 4 + (w & 3u)];
    };
    auto convert = [&](const unsigned (&in)[NPI], unsigned (&o)[NPO]) {
        if constexpr (MODE == 1) {
            const unsigned v = (in[0] | in[1] | in[2] | in[3]) & ~in[4];
            o[0] = (in[1] | in[3]) & v; o[1] = (in[2] | in[3]) & v; o[2] = v;
        } else {
#pragma unroll
            for (int p = 0; p < NPO; p++) o[p] = in[p];
        }
    };
    unsigned out[NPO][4];
    const unsigned t0 = G * SITES_PER_GROUP;
#pragma unroll
    for (int ow = 0; ow < 4; ow++) {
        unsigned accw[NPO];
#pragma unroll
        for (int p = 0; p < NPO; p++) accw[p] = 0;
        const unsigned tb = t0 + ow * 32;
        const unsigned kn = tb >= count ? 0u : min(32u, count - tb);
        if (kn) {
            const unsigned first = __builtin_amdgcn_readfirstlane(list[tb]), last = __builtin_amdgcn_readfirstlane(list[tb + kn - 1]);
            if (last - first == kn - 1) {                       // a run of consecutive sites (wave-uniform): funnel shift
                const unsigned w0 = first >> 5, sh = first & 31u, w1 = last >> 5;
                unsigned lo[NPI], hi[NPI], cv_lo[NPO], cv_hi[NPO];
#pragma unroll
                for (int p = 0; p < NPI; p++) { lo[p] = word_at(w0, p); hi[p] = w1 != w0 ? word_at(w1, p) : 0u; }
                convert(lo, cv_lo);
                convert(hi, cv_hi);
                const unsigned keep = kn == 32u ? 0xFFFFFFFFu : ((1u << kn) - 1u);
#pragma unroll
                for (int p = 0; p < NPO; p++)
                    accw[p] = (unsigned)((((unsigned long long)cv_hi[p] << 32) | cv_lo[p]) >> sh) & keep;
            } else {
                // scattered sites: their words are loaded eight sites at a time (independent loads in flight: one site at a time
                // paid a memory latency per site -- 100 us for the 476 dense sites of the bench workload)
                for (unsigned k0 = 0; k0 < kn; k0 += 8) {
                    unsigned site[8], in[8][NPI];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        site[q] = __builtin_amdgcn_readfirstlane(list[tb + min(k0 + q, kn - 1u)]);
#pragma unroll
                        for (int p = 0; p < NPI; p++) in[q][p] = word_at(site[q] >> 5, p);
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        unsigned cur[NPO];
                        convert(in[q], cur);
                        if (k0 + q < kn) {                          // (wave-uniform)
#pragma unroll
                            for (int p = 0; p < NPO; p++) accw[p] |= ((cur[p] >> (site[q] & 31u)) & 1u) << (k0 + q);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NPO; p++) out[p][ow] = accw[p];
    }
    if (s < n)
#pragma unroll
        for (int p = 0; p < NPO; p++)
            dst[((size_t)G * NPO + p) * n_pad + s] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
}

// per sample: set bits of one plane (stride `gp` planes per group; lanes over samples: coalesced; grid.y cuts the groups,
// partial counts are added)
__global__ __launch_bounds__(256) void plane_popcount_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, size_t groups, int gp,
                                                             unsigned *__restrict__ out)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const size_t per = (groups + gridDim.y - 1) / gridDim.y;
    const size_t g0 = blockIdx.y * per, g1 = min(groups, g0 + per);
    unsigned c = 0;
    for (size_t g = g0; g < g1; g++) {
        const uint4 v = P[g * gp * n_pad + s];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    if (c) atomicAdd(&out[s], c);
}

// per sample: its N sites among the sites of `mask` (the stored N plane, in place; lanes over samples, grid.y cuts the groups)
__global__ __launch_bounds__(256) void plane_popcount_masked_kernel(const uint4 *__restrict__ Nplane, const uint4 *__restrict__ mask, size_t n_pad,
                                                                    unsigned n, size_t groups, unsigned *__restrict__ out)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const size_t per = (groups + gridDim.y - 1) / gridDim.y;
    const size_t g0 = blockIdx.y * per, g1 = min(groups, g0 + per);
    unsigned c = 0;
    for (size_t g = g0; g < g1; g++) {
        const uint4 m = mask[g];                               // wave-uniform
        if ((m.x | m.y | m.z | m.w) == 0u) continue;
        const uint4 v = Nplane[g * NPLANES * n_pad + s];
        c += __popc(v.x & m.x) + __popc(v.y & m.y) + __popc(v.z & m.z) + __popc(v.w & m.w);
    }
    if (c) atomicAdd(&out[s], c);
}

// The U plane of an alignment whose minority sites take their N x listed terms from the matrix cores: per (site, sample) "is N, or --
// at a minority site -- is listed with an allele mask that lacks the site's reference base" (w = 1).  With n = the N plane and w = that
// second set (disjoint: a listed sample is not N),  U U^T - n n^T = w n^T + n w^T + w w^T  over the minority sites: what phase B of
// minor_fixup_kernel walks lists for (site_lists.hip), as two one-plane passes of pairsnp_mfma_kernel<COUNT>.  Layout: one plane per
// group, like `iplanes`.  thread = (group, sample), lanes over samples.
__global__ __launch_bounds__(256) void u_plane_kernel(const uint4 *__restrict__ P, size_t n_pad, unsigned n, size_t groups,
                                                      const uint4 *__restrict__ minor_mask, const uint4 *__restrict__ ref_x,
                                                      const uint4 *__restrict__ ref_y, uint4 *__restrict__ U)
{
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned s = blockIdx.y * 64 + (threadIdx.x & 63);
    if (g >= groups || s >= n) return;
    typedef unsigned upl_u32x4 __attribute__((ext_vector_type(4)));
    const uint4 *bp = P + (g * NPLANES) * n_pad + s;
    const upl_u32x4 A = __builtin_nontemporal_load(reinterpret_cast<const upl_u32x4 *>(bp)), C = __builtin_nontemporal_load(reinterpret_cast<const upl_u32x4 *>(bp + n_pad)),
                    G = __builtin_nontemporal_load(reinterpret_cast<const upl_u32x4 *>(bp + 2 * n_pad)), T = __builtin_nontemporal_load(reinterpret_cast<const upl_u32x4 *>(bp + 3 * n_pad));
    const uint4 M = minor_mask[g], RX = ref_x[g], RY = ref_y[g];
    const unsigned a[4] = {A.x, A.y, A.z, A.w}, c[4] = {C.x, C.y, C.z, C.w}, gg[4] = {G.x, G.y, G.z, G.w}, t[4] = {T.x, T.y, T.z, T.w};
    const unsigned m[4] = {M.x, M.y, M.z, M.w}, rx[4] = {RX.x, RX.y, RX.z, RX.w}, ry[4] = {RY.x, RY.y, RY.z, RY.w};
    unsigned u[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned isn = a[w] & c[w] & gg[w] & t[w];
        const unsigned ra = ~rx[w] & ~ry[w], rc = rx[w] & ~ry[w], rg = ~rx[w] & ry[w], rt = rx[w] & ry[w];
        const unsigned has_ref = (a[w] & ra) | (c[w] & rc) | (gg[w] & rg) | (t[w] & rt);
        u[w] = isn | ((a[w] | c[w] | gg[w] | t[w]) & ~has_ref & m[w]);
    }
    U[g * n_pad + s] = make_uint4(u[0], u[1], u[2], u[3]);
}

void site_classes_free(tracs_alignment *a)
{
    minority_lists_free(a);
    a->c_counted = nullptr;
    a->vplanes = a->iplanes = a->uplane = nullptr;
    a->nw_gram = a->nw_rows = false;
    pack_release(a);                                       // vplanes, iplanes, N counts, minority lists: one arena
    a->L_var = a->L_inv = a->groups_var = a->groups_inv = 0;
    a->L_minor = a->L_full = a->L_un = a->L_nnl = 0;
    a->count_in_place = false;
    a->classes_cons = false;
    a->classes_state = 0;
}

static size_t class_plane_bytes(const tracs_alignment *a, size_t groups, int planes, int pad_groups)
{
    return ((groups + pad_groups) * (size_t)planes * a->n_pad + TAIL_PAD) * sizeof(uint4);
}

static int g_force_classes = -2;          // tracs_debug_force_site_classes: -2 follow TRACS_SITE_CLASSES, -1 cost model, 0 never, 1 always

// ---- stage clock of the once-per-pack work (diagnostics: bench.py's single_pass.stages, TRACS_CLASSES_TRACE) --------------
// HIP events on the launch stream, read back afterwards: no synchronisation inside the build.
static constexpr int kMaxStages = 12;
static hipEvent_t g_stage_ev[kMaxStages + 1];
static const char *g_stage_name[kMaxStages];
static double g_stage_rd[kMaxStages], g_stage_wr[kMaxStages];     // bytes the stage reads / writes (its arrays, once per pass over them)
static int g_stage_n = 0;
static bool g_stage_on = false, g_stage_valid = false;

static void stage_begin(hipStream_t stream)
{
    static const bool trace = std::getenv("TRACS_CLASSES_TRACE") != nullptr;
    if (trace) g_stage_on = true;
    g_stage_n = 0; g_stage_valid = false;
    if (!g_stage_on) return;
    if (!g_stage_ev[0]) for (auto &e : g_stage_ev) (void)hipEventCreate(&e);
    (void)hipEventRecord(g_stage_ev[0], stream);
}
static void stage_mark(const char *name, hipStream_t stream, double rd = 0.0, double wr = 0.0)
{
    if (!g_stage_on || g_stage_n >= kMaxStages) return;
    g_stage_rd[g_stage_n] = rd; g_stage_wr[g_stage_n] = wr;
    g_stage_name[g_stage_n++] = name;
    (void)hipEventRecord(g_stage_ev[g_stage_n], stream);
    g_stage_valid = true;
}
void pack_stage_mark(const char *name, hipStream_t stream, double rd, double wr) { stage_mark(name, stream, rd, wr); }
void pack_stage_begin(hipStream_t stream) { stage_begin(stream); }
void pack_stage_end()
{
    static const bool trace = std::getenv("TRACS_CLASSES_TRACE") != nullptr;
    if (!trace || !g_stage_valid) return;
    (void)hipEventSynchronize(g_stage_ev[g_stage_n]);
    for (int k = 0; k < g_stage_n; k++) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, g_stage_ev[k], g_stage_ev[k + 1]);
        std::fprintf(stderr, "[once per pack] %-28s %8.2f ms\n", g_stage_name[k], ms);
    }
}

// Decides (once per pack) the encoding and whether the pair kernels run on site classes, and builds the re-packed alignments
// and lists if so.  *partial: some sample carries a partial IUPAC code (the alignment has no consensus form).  Soft-fails
// (classes_state = -1) when memory is short.
// gram: the second form of the classes, tried when the first is refused (or TRACS_NW_GRAM=1) -- the minority sites' N x listed terms
// on the matrix cores.  With many N samples per site (tracs align masks every position below its coverage thresholds: tens of per
// cent of every sample, tracs/align.py:599-613) a listed sample's walk of the site's N list costs cN entries, k (cN + k) leaves
// the list budget and every site would go through the pair kernel; the terms the N lists are walked for are
//     sum_s w_a(s) n_b(s) + n_a(s) w_b(s)  =  (U U^T - n n^T - w w^T)(a, b),   U = n | w  (u_plane_kernel above),
// two one-plane matrix passes over all sites whatever cN is (n n^T is the compared-sites count's own pass, in place), and w w^T
// joins the both-listed term of the p-list walk.  A minority site then costs k^2 list entries, and no site carries an N list.
// Tried when the list form costs more matrix work than this one can (two planes of every site), taken when it then costs less
// (`beat`: the list form's cost); refused -> the list form again (its masks were overwritten: classified once more, `no_gram`).
// TRACS_NW_GRAM=1 always, 0 never.
static int decide(tracs_alignment *a, bool allow_minor, bool allow_nnl, hipStream_t stream, int *partial, bool gram = false, double beat = 1.0,
                  bool no_gram = false)
{
    a->classes_state = -1;
    const int env_gram = [] { const char *e = std::getenv("TRACS_NW_GRAM"); return e ? std::atoi(e) : -1; }();    // (read per pack: tests switch it)
    if (env_gram == 1 && allow_minor) gram = true;
    const bool gram_next = !gram && !no_gram && env_gram != 0 && allow_minor && allow_nnl;
    static const int env_force = [] { const char *e = std::getenv("TRACS_SITE_CLASSES"); return e ? std::atoi(e) : -1; }();
    const int force = g_force_classes >= -1 ? g_force_classes : env_force;
    static const bool no_minor = [] { const char *e = std::getenv("TRACS_MINORITY"); return e && std::atoi(e) == 0; }();
    static const bool force_general = std::getenv("TRACS_FORCE_GENERAL") != nullptr;
    const size_t groups = a->groups;
    // sample blocks in grid.y of the re-pack kernels (<= 65535 x 64 samples per launch; beyond that the classes are not used)
    const unsigned sblocks = (unsigned)(a->n_pad / 64);
    if (sblocks > 65535u || groups >= (1ull << 31)) { *partial = -1; return TRACS_OK; }
    const size_t flag_words = (a->n_pad + 255) / 256 * 4;       // 64-sample words per group (whole 256-sample rounds)
    // scratch (grow-only per-device buffers: a second pack in the same process allocates nothing here)
    uint4 *masks = nullptr;
    unsigned *offs = nullptr, *cnts = nullptr, *gcnt = nullptr, *d_flag = nullptr;
    unsigned long long *off64 = nullptr, *totals = nullptr, *flags = nullptr;
    int rc;
    // (two sets of class masks / sums / offsets / totals: the second holds the classes of the same pass under the other form's criterion)
    if ((rc = workspace_get(52, 2 * M_SLOTS * groups * sizeof(uint4), reinterpret_cast<void **>(&masks)))) return rc;
    if ((rc = workspace_get(53, 2 * 7 * groups * sizeof(unsigned), reinterpret_cast<void **>(&offs)))) return rc;
    if ((rc = workspace_get(54, 2 * groups * SITES_PER_GROUP * sizeof(unsigned), reinterpret_cast<void **>(&cnts)))) return rc;
    if ((rc = workspace_get(55, 2 * 7 * groups * sizeof(unsigned), reinterpret_cast<void **>(&gcnt)))) return rc;
    if ((rc = workspace_get(56, 2 * 7 * (groups + 1) * sizeof(unsigned long long), reinterpret_cast<void **>(&off64)))) return rc;
    if ((rc = workspace_get(57, 256, reinterpret_cast<void **>(&totals)))) return rc;
    if ((rc = workspace_get(58, groups * flag_words * sizeof(unsigned long long), reinterpret_cast<void **>(&flags)))) return rc;
    d_flag = reinterpret_cast<unsigned *>(totals + 15);
    uint4 *const masks_ref = masks;                              // (the reference base bits are written with the first set only)
    uint4 *const masks2 = masks + (size_t)M_SLOTS * groups;
    unsigned *const offs2 = offs + 7 * groups, *const gcnt2 = gcnt + 7 * groups;
    unsigned long long *const off64_2 = off64 + 7 * (groups + 1), *const totals2 = totals + 16;
    bool switched = false;                                       // the second form entered in place, from the second set of classes
    auto mask_of = [&](int slot) { return (slot == M_REFX || slot == M_REFY ? masks_ref : masks) + (size_t)slot * groups; };
    auto off_of = [&](int slot) { return offs + (size_t)slot * groups; };
    unsigned *cntP = cnts, *cntN = cnts + groups * SITES_PER_GROUP;
    // a site goes to the minority lists while its k (cN + k) entries cost less than the extra operand planes over all pairs:
    // 51 ns per site at 10 000 samples in the pair kernel (consensus planes; 120 ns in the general encoding, and as much again to
    // re-pack the site) against 0.2-1.3 ps per list entry (minor_fixup_kernel: 0.65 ms for 5e8 entries on the bench alignment, 8.2 ms
    // for 3.8e10 with 0.5 % partial codes) -- k (cN + k) <= n^2 / 2000: 50 000 entries at 10 000 samples.  (n^2 / 8000 until round 5
    // sent the 16 896 sites of the partial-code alignment with k >= 73 through the pair kernel: 4.5 ms of a 51.5 ms call;
    // TRACS_MINOR_BUDGET_DIV overrides the divisor for that measurement)
    static const double budget_div = [] { const char *e = std::getenv("TRACS_MINOR_BUDGET_DIV"); return e && std::atof(e) > 0 ? std::atof(e) : 2000.0; }();
    const double bsites = (double)a->n * (double)a->n / budget_div;
    const unsigned budget = (no_minor || !allow_minor) ? 0u : (unsigned)std::min(1.0e9, std::max(16.0, bsites));
    // The N co-occurrences of a site with cN N samples cost cN list walks of one cache line per ~115 samples of the list (n8 lines,
    // pairsnp_kernels.h; nn_rows_kernel is bound by the lines it pulls through the fabric) against n^2 / 2 pairs on the matrix
    // cores, whatever cN: lists while cN x lines <= TRACS_NN_LIST_K x n^2 (default 6e-6: DESIGN.md 3.1); TRACS_NN_LISTS=0: never
    static const bool no_nnl = [] { const char *e = std::getenv("TRACS_NN_LISTS"); return e && std::atoi(e) == 0; }();
    static const double nnl_k = [] { const char *e = std::getenv("TRACS_NN_LIST_K"); return e ? std::atof(e) : 6e-6; }();
    const unsigned nn_list_max = (no_nnl || !allow_nnl) ? 0u : (unsigned)std::min(4.0e9, nnl_k * (double)a->n * (double)a->n);
    TRACS_HIP_CHECK(hipMemsetAsync(totals, 0, 256, stream));
#define TRACS_CLASSIFY_ARGS a->planes, a->n_pad, (unsigned)a->n, budget, nn_list_max, gram ? 1u : 0u, masks, groups, cntP, cntN, gcnt, gcnt + groups, \
                            gcnt + 2 * groups, gcnt + 3 * groups, gcnt + 4 * groups, gcnt + 5 * groups, gcnt + 6 * groups, flags, flag_words, d_flag,          \
                            gram_next ? masks2 : (uint4 *)nullptr, gram_next ? gcnt2 : (unsigned *)nullptr
    const int env_threads = [] { const char *e = std::getenv("TRACS_CLASSIFY_THREADS"); return e ? std::atoi(e) : 0; }();      // (read per pack: tests switch it)
    if (env_threads == 64 || (env_threads == 0 && a->n <= 2048))
        hipLaunchKernelGGL(classify_sites_kernel<64>, dim3((unsigned)groups), dim3(64), 0, stream, TRACS_CLASSIFY_ARGS);
    else if (env_threads != 256)
        hipLaunchKernelGGL(classify_sites_kernel<128>, dim3((unsigned)groups), dim3(128), 0, stream, TRACS_CLASSIFY_ARGS);
    else
        hipLaunchKernelGGL(classify_sites_kernel<256>, dim3((unsigned)groups), dim3(256), 0, stream, TRACS_CLASSIFY_ARGS);
#undef TRACS_CLASSIFY_ARGS
    const double plane_b = (double)groups * (double)a->n_pad * sizeof(uint4);      // one bit plane of the alignment
    stage_mark("classify", stream, 4.0 * plane_b, (double)groups * (M_SLOTS * 16.0 + 3.0 * SITES_PER_GROUP * 4.0 + flag_words * 8.0));
    hipLaunchKernelGGL(group_offsets_kernel, dim3(14), dim3(1024), 0, stream, masks, gcnt, groups, offs, off64, totals);
    unsigned long long tot[16] = {0};
    TRACS_HIP_CHECK(hipMemcpyAsync(tot, totals, 128, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    stage_mark("class sizes", stream);
    const unsigned class_flags = reinterpret_cast<const unsigned *>(&tot[15])[0];
    *partial = (int)(class_flags & 1u);
    bool long_p = (class_flags & 2u) != 0u;
    const bool consensus = !*partial && !force_general;
    size_t L_dense = (size_t)tot[M_DENSE], L_minor = (size_t)tot[M_MINOR], L_full = (size_t)tot[M_FULL];
    size_t L_count = (size_t)tot[M_COUNT], L_nnl = (size_t)tot[M_NNL], L_lst = (size_t)tot[M_LST], L_un = (size_t)tot[M_UN];
    unsigned long long tot_p = tot[7], tot_q = long_p ? tot[13] : 0ull;
    // (256-byte q lines when the minority sites list more than Q_WIDE_MEAN samples on average: site_lists.hip; TRACS_QLINE_DWORDS=32|64 forces)
    static const int force_qw = [] { const char *e = std::getenv("TRACS_QLINE_DWORDS"); return e ? std::atoi(e) : 0; }();
    unsigned qw = force_qw == 32 || force_qw == 64 ? (unsigned)force_qw : ((long_p && L_minor && (double)tot_p > Q_WIDE_MEAN * (double)L_minor) ? 64u : 32u);
    unsigned long long tot_o = tot[8], tot_nnl = tot[10];
    int lst_slot = M_LST, ovf_slot = 1;                      // which mask / per-group overflow bound the lists are built from
    if (force == 0 || a->L == 0 || a->n < 2) return TRACS_OK;
    // The counting pass reads the stored N plane IN PLACE -- every site, nothing re-packed, nothing from N co-occurrence lists: nn =
    // L - c_i - c_j + NN over all sites and the pair kernels write d only -- when that is cheaper than re-packing the counted sites'
    // N plane: counting the other sites too costs ~1.5e-13 ms per site and pair of samples on the matrix cores (75 ms for 5 Mbp x 10 000
    // samples), the re-pack ~4e-10 ms per counted site and sample (compact_sites_kernel<2>: a bit gather).  Always the case when nearly
    // every site is counted (10 % N); and for SMALL alignments (config 2: 1 000 samples -- 87 % of the sites counted, the re-pack 0.75 of
    // the call's 2.4 ms), whose NNL sites then go to the matrix cores as well: the lists are built for the minority sites alone
    // (TRACS_COUNT_IN_PLACE=0|1 forces the choice: diagnostics).
    static const int force_in_place = [] { const char *e = std::getenv("TRACS_COUNT_IN_PLACE"); return e ? std::atoi(e) : -1; }();
    const double t_extra = 1.5e-13 * (double)(a->L - L_count) * (double)a->n * (double)a->n;
    const double t_repack = 4.0e-10 * (double)L_count * (double)a->n_pad;
    // (gram: n n^T over every site is part of the distances: always the stored N plane in place)
    bool in_place = gram || (L_count > 0 && (force_in_place >= 0 ? force_in_place == 1 : t_extra < t_repack));
    if (in_place && L_nnl) {
        // the same classes with the NNL sites counted: lists = the minority sites (their masks, ranks and overflow bounds exist)
        L_count += L_nnl; L_lst = L_minor; L_nnl = 0;
        tot_o = tot[12]; tot_nnl = 0;
        lst_slot = M_MINOR; ovf_slot = 5;
    }
    // matrix instructions per pair: planes_full per site now; planes_full per dense site + one per counted site with classes
    const double planes_full = consensus ? 4.0 : 5.0;
    double cost = (planes_full * (double)L_dense + (double)(in_place ? a->L : L_count) + (gram ? (double)a->L : 0.0)) / (planes_full * (double)a->L);
    if (gram && env_gram != 1 && force != 1 && cost >= std::min(0.92, beat - 0.02)) return TRACS_OK;      // (the caller goes on with the list form)
    if (gram_next && force != 0 && cost > 2.0 / planes_full + 0.05) {
        // the list form costs more matrix work than the second form can: that form's classes came out of the same pass (masks2) --
        // their sizes, their cost; taken when it is the cheaper one
        hipLaunchKernelGGL(group_offsets_kernel, dim3(14), dim3(1024), 0, stream, masks2, gcnt2, groups, offs2, off64_2, totals2);
        unsigned long long t2[16] = {0};
        TRACS_HIP_CHECK(hipMemcpyAsync(t2, totals2, 128, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        stage_mark("class sizes (second form)", stream);
        const double cost2 = (planes_full * (double)t2[M_DENSE] + 2.0 * (double)a->L) / (planes_full * (double)a->L);
        if (cost2 < std::min(0.92, cost - 0.02)) {
            gram = true; switched = true;
            masks = masks2; offs = offs2; gcnt = gcnt2; off64 = off64_2;
            for (int k = 0; k < 15; k++) tot[k] = t2[k];
            long_p = (class_flags & 4u) != 0u;
            L_dense = (size_t)tot[M_DENSE]; L_minor = (size_t)tot[M_MINOR]; L_full = (size_t)tot[M_FULL];
            L_count = (size_t)tot[M_COUNT]; L_nnl = (size_t)tot[M_NNL]; L_lst = (size_t)tot[M_LST]; L_un = (size_t)tot[M_UN];
            tot_p = tot[7]; tot_q = long_p ? tot[13] : 0ull;
            qw = force_qw == 32 || force_qw == 64 ? (unsigned)force_qw : ((long_p && L_minor && (double)tot_p > Q_WIDE_MEAN * (double)L_minor) ? 64u : 32u);
            tot_o = tot[8]; tot_nnl = tot[10];
            lst_slot = M_LST; ovf_slot = 1;
            in_place = true;
            cost = cost2;
        }
    }
    // (a refusal further down, of a form entered in place: the list form again -- classified once more, its masks were kept but its
    // choices above were made for the other set)
    auto refuse_gram = [&]() -> int { return switched ? decide(a, allow_minor, allow_nnl, stream, partial, false, 1.0, true) : TRACS_OK; };
    if (force != 1 && cost >= 0.92) return TRACS_OK;
    // the lists must stay small beside the planes (<= one entry per 8 sites of the whole alignment; TRACS_LIST_CAP: diagnostics):
    // otherwise first without the N co-occurrence lists (those sites are counted on the matrix cores), then without any list
    // (the minority sites stay dense)
    if (L_lst) {
        // (TRACS_LIST_CAP=<entries of four bytes>: diagnostics)
        static const double env_cap = [] { const char *e = std::getenv("TRACS_LIST_CAP"); return e ? 4.0 * std::atof(e) : -1.0; }();
        const double cap = 0.8 * (double)NPLANES * (double)groups * (double)a->n_pad * sizeof(uint4);
        // N-list lines (primary + the overflow lines they can need at most), p lists and their per-sample form, the rows' N bitmaps
        const double bytes = ((double)L_lst + (double)tot_o) * 128.0 + 16.0 * (double)tot_p + (long_p ? ((double)L_lst + (double)tot_q) * 4.0 * (double)qw : 0.0) +
                             (L_nnl ? (double)a->n * (double)((groups + 7) / 8 * 8) * sizeof(uint4) : 0.0);
        if (L_lst >= (1ull << 26) || a->n >= (1ull << 27) || bytes > (env_cap >= 0.0 ? std::min(env_cap, cap) : cap) ||
            L_lst + tot_o >= (1ull << 32) || L_lst + tot_q >= (1ull << 32)) {                // (line indices are 32 bits)
            if (gram) return refuse_gram();                                                  // (refused: the list form instead)
            return L_nnl ? decide(a, allow_minor, false, stream, partial) : decide(a, false, false, stream, partial);
        }
    }

    auto soft_fail = [&]() { (void)hipGetLastError(); site_classes_free(a); a->classes_state = -1; return TRACS_OK; };
    const int npv = consensus ? 3 : NPLANES;
    const size_t gv = groups_for(L_dense), gi = in_place ? 0 : groups_for(L_count);
    const size_t vbytes = class_plane_bytes(a, gv, npv, PAD_GROUPS), ibytes = class_plane_bytes(a, gi, 1, PAD_GROUPS);
    unsigned *lists = nullptr;
    if ((rc = workspace_get(59, (L_dense + L_count + 1) * sizeof(unsigned), reinterpret_cast<void **>(&lists)))) return rc;
    if (pack_alloc(a, vbytes, reinterpret_cast<void **>(&a->vplanes)) != hipSuccess) return soft_fail();
    if (gi && pack_alloc(a, ibytes, reinterpret_cast<void **>(&a->iplanes)) != hipSuccess) return soft_fail();
    if (pack_alloc(a, a->n_pad * sizeof(unsigned), reinterpret_cast<void **>(&a->c_counted)) != hipSuccess) return soft_fail();
    unsigned *list_dense = lists, *list_count = lists + L_dense;
    bool ok = hipMemsetAsync(a->vplanes, 0, vbytes, stream) == hipSuccess &&
              (!gi || hipMemsetAsync(a->iplanes, 0, ibytes, stream) == hipSuccess) &&
              hipMemsetAsync(a->c_counted, 0, a->n_pad * sizeof(unsigned), stream) == hipSuccess;
    const dim3 lgrid((unsigned)((groups + 255) / 256));
    if (gv) {
        hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, mask_of(M_DENSE), off_of(M_DENSE), groups, list_dense);
        const dim3 grid((unsigned)((gv + 3) / 4), sblocks);
        if (consensus)
            hipLaunchKernelGGL((compact_sites_kernel<1>), grid, dim3(256), 0, stream, a->planes, list_dense, (unsigned)L_dense, a->vplanes,
                               a->n_pad, (unsigned)a->n, (unsigned)gv);
        else
            hipLaunchKernelGGL((compact_sites_kernel<0>), grid, dim3(256), 0, stream, a->planes, list_dense, (unsigned)L_dense, a->vplanes,
                               a->n_pad, (unsigned)a->n, (unsigned)gv);
    }
    stage_mark("re-pack dense sites", stream, (double)gv * NPLANES * a->n_pad * 16.0, (double)vbytes);
    if (gi) {
        hipLaunchKernelGGL(class_list_kernel, lgrid, dim3(256), 0, stream, mask_of(M_COUNT), off_of(M_COUNT), groups, list_count);
        const dim3 grid((unsigned)((gi + 3) / 4), sblocks);
        hipLaunchKernelGGL((compact_sites_kernel<2>), grid, dim3(256), 0, stream, a->planes, list_count, (unsigned)L_count, a->iplanes,
                           a->n_pad, (unsigned)a->n, (unsigned)gi);
    }
    // per sample: its N sites among the sites the compared-sites formula stands for: every site (in place), or every site
    // outside the dense class (re-packed counting pass and / or lists: nn = |U| - c_i - c_j + NN over U).  With N co-occurrence
    // lists the kernel that writes the rows' N bitmaps counts on its way (site_lists.hip); a build without them counts here.
    const bool counts_with_bitmaps = !in_place && L_nnl > 0;
    if (in_place)
        hipLaunchKernelGGL(plane_popcount_kernel, dim3((unsigned)((a->n + 255) / 256), 128), dim3(256), 0, stream, a->planes + 4 * a->n_pad,
                           a->n_pad, (unsigned)a->n, groups, NPLANES, a->c_counted);
    else if (L_un && !counts_with_bitmaps)
        hipLaunchKernelGGL(plane_popcount_masked_kernel, dim3((unsigned)((a->n + 255) / 256), 128), dim3(256), 0, stream,
                           a->planes + 4 * a->n_pad, mask_of(M_UN), a->n_pad, (unsigned)a->n, groups, a->c_counted);
    if (gi) stage_mark("re-pack counted sites", stream, plane_b, (double)ibytes);
    else if (in_place || (L_un && !counts_with_bitmaps)) stage_mark("N counts per sample", stream, plane_b, 0.0);
    bool gram_rows = false;
    if (L_lst) {
        // the lists (site_lists.hip): per-site lists from the N plane and the flagged samples, the rows' N bitmaps from the N plane
        int built = 0;
        MinorBuild mb;
        mb.planes = a->planes; mb.minor_mask = mask_of(M_MINOR); mb.nnl_mask = mask_of(M_NNL); mb.lst_mask = mask_of(lst_slot);
        mb.ref_x = mask_of(M_REFX); mb.ref_y = mask_of(M_REFY); mb.un_mask = mask_of(M_UN); mb.off_lst = off_of(lst_slot);
        mb.cntP = cntP; mb.cntN = cntN; mb.gP = gcnt; mb.baseP = off64; mb.baseO = off64 + (size_t)ovf_slot * groups; mb.flags = flags; mb.flag_words = flag_words;
        mb.sites = L_lst; mb.tot_p = tot_p; mb.tot_o = tot_o; mb.tot_nnl = tot_nnl; mb.baseQ = off64 + 6 * groups; mb.tot_q = tot_q; mb.long_p = long_p ? 1 : 0;
        mb.max_gp = tot[14];
        mb.qw = qw;
        // second form: the N x listed terms from the rows of the site-major N matrix, summed per listed sample (|W| x ns_words x 4 bytes from
        // HBM at ~4.5 TB/s, + the matrix itself: one read and one write of a plane), when that is less than a one-plane pass over every site
        // (1.5e-13 ms per site and pair of samples) + the U plane (four plane reads); tot_p bounds |W| from above.  TRACS_NW_ROWS=0|1 forces.
        const int env_rows = [] { const char *e = std::getenv("TRACS_NW_ROWS"); return e ? std::atoi(e) : -1; }();
        const double t_rows = (double)tot_p * (double)ns_words_for(a->n_pad) * 4.0 / 4.5e9 + 2.0 * plane_b / 5.0e9;
        const double t_upass = 1.5e-13 * (double)a->L * (double)a->n * (double)a->n + 4.0 * plane_b / 5.0e9;
        gram_rows = gram && (env_rows >= 0 ? env_rows == 1 : t_rows < t_upass);
        mb.gram = gram ? (gram_rows ? 2 : 1) : 0;
        mb.n_rows = a->n_row_hint;
        for (int k = 0; k < 4; k++) mb.rows[k] = (unsigned)std::min<size_t>(a->row_hint[k], a->n);
        rc = minority_lists_build(a, mb, stream, &built);
        if (rc) { site_classes_free(a); a->classes_state = -1; return rc; }
        if (!built) {                                          // no memory: the same classes with fewer lists
            soft_fail();
            if (gram) return refuse_gram();
            return L_nnl ? decide(a, allow_minor, false, stream, partial) : decide(a, false, false, stream, partial);
        }
    }
    const bool gram_on = gram && L_minor > 0 && a->lists != nullptr;
    if (gram_on && !gram_rows) {
        const size_t ubytes = class_plane_bytes(a, groups, 1, PAD_GROUPS);
        if (pack_alloc(a, ubytes, reinterpret_cast<void **>(&a->uplane)) != hipSuccess) { soft_fail(); return refuse_gram(); }
        // (the pad groups and the slack behind them: zero)
        ok = ok && hipMemsetAsync(a->uplane + groups * a->n_pad, 0, ubytes - groups * a->n_pad * sizeof(uint4), stream) == hipSuccess;
        if (a->n_pad > a->n) ok = ok && hipMemsetAsync(a->uplane, 0, groups * a->n_pad * sizeof(uint4), stream) == hipSuccess;
        hipLaunchKernelGGL(u_plane_kernel, dim3((unsigned)((groups + 3) / 4), sblocks), dim3(256), 0, stream, a->planes, a->n_pad, (unsigned)a->n,
                           groups, mask_of(M_MINOR), mask_of(M_REFX), mask_of(M_REFY), a->uplane);
        stage_mark("U plane (N | listed, w = 1)", stream, 4.0 * plane_b, plane_b);
    }
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok) { site_classes_free(a); a->classes_state = -1; set_error("site_classes_decide: re-pack failed"); return TRACS_E_HIP; }
    a->L_var = L_dense; a->L_inv = L_count; a->groups_var = gv; a->groups_inv = gi;
    a->L_minor = L_minor; a->L_full = L_full; a->L_un = L_un; a->L_nnl = L_nnl;
    a->nn_visits = tot[9]; a->list_entries_n = (L_lst ? (unsigned long long)L_lst + tot_o : 0ull); a->list_entries_p = tot_p; a->nn_walks = tot_nnl; a->fix_walks = tot[11];
    a->count_in_place = in_place;
    a->nw_gram = gram_on;
    a->nw_rows = gram_on && gram_rows;
    a->classes_cons = consensus;
    a->classes_state = 1;
    return TRACS_OK;
}

int site_classes_decide(tracs_alignment *a, hipStream_t stream, int *partial)
{
    *partial = -1;
    if (a->classes_state != 0) return TRACS_OK;
    return decide(a, true, true, stream, partial);
}

}  // namespace tracs

extern "C" {

// Diagnostics (bench.py's sensitivity legs: the same alignment with and without site classes in one process): overrides
// TRACS_SITE_CLASSES for alignments decided from now on.  -2 environment, -1 cost model, 0 never, 1 always.
void tracs_debug_force_site_classes(int mode) { tracs::g_force_classes = mode; }

// Diagnostics (bench.py's single_pass.stages): record the stages of the once-per-pack work of later dense calls with HIP events
void tracs_debug_pack_timing(int on) { tracs::g_stage_on = on != 0; }

// The stages of the LAST once-per-pack build: names joined by '\n' into `names` (cap bytes), milliseconds into ms[0 .. max).
// Returns the number of stages (0: nothing recorded).
int tracs_debug_pack_stages(char *names, size_t cap, float *ms, int max_stages)
{
    using namespace tracs;
    if (!g_stage_valid || g_stage_n == 0) return 0;
    if (hipEventSynchronize(g_stage_ev[g_stage_n]) != hipSuccess) return 0;
    size_t used = 0;
    int k = 0;
    for (; k < g_stage_n && k < max_stages; k++) {
        float t = 0.0f;
        (void)hipEventElapsedTime(&t, g_stage_ev[k], g_stage_ev[k + 1]);
        if (ms) ms[k] = t;
        const size_t len = std::strlen(g_stage_name[k]);
        if (names && used + len + 2 <= cap) {
            if (k) names[used++] = '\n';
            std::memcpy(names + used, g_stage_name[k], len);
            used += len;
            names[used] = 0;
        }
    }
    return k;
}

// bytes read / written by the stages of the last once-per-pack build (the library's own accounting: each array once per pass
// over it) -- bench.py's roofline_per_pack.  Returns the number of stages.
int tracs_debug_pack_stage_bytes(double *rd, double *wr, int max_stages)
{
    using namespace tracs;
    if (!g_stage_valid) return 0;
    int k = 0;
    for (; k < g_stage_n && k < max_stages; k++) { if (rd) rd[k] = g_stage_rd[k]; if (wr) wr[k] = g_stage_wr[k]; }
    return k;
}

}  // extern "C"
