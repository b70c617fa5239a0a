// pairsnp.hip -- pack kernels, the VALU pair-tile kernel, COO extraction and the host driver of the pair loop (gfx950).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   load_seqs :107-199  (IUPAC -> four allele bit sets)        -> pack_kernel
//   pair loop :395-420  (match/popcount, compared sites)        -> pairsnp_mfma_kernel (pairsnp_mfma.hip), pairsnp_tile_kernel
//   emit d <= dist :405, row-major order :451-455               -> coo_count/coo_fill
//
// What a dense call launches (pairsnp_dense_impl):
//   once per pack                                               the encoding (consensus: no partial IUPAC code anywhere) and the
//                                                               site classes (site_classes.hip): dense / counted / full / minority
//   sites read by the pair kernel (the dense class, or every site when the classes are not used)
//     consensus alignments                                      matrix-core kernel, operands x, y, z, v (3 planes)
//     general alignments                                        matrix-core kernel, one-hot operands (5 planes) + the sparse
//                                                               partial-code correction of general_sparse.hip
//     lists unavailable / TRACS_MFMA=0                          pairsnp_tile_kernel: integer VALU, one workgroup per TI x TJ tile,
//                                                               row and column samples staged HBM -> LDS directly, per 32 sites
//                                                               and pair v_and, 3 x v_bitop3 (and-or), v_bcnt(+acc), v_or, v_bcnt
//   minority sites                                              general_fixup_kernel<MINOR>: distances from sparse lists
//   counted / full sites                                        pairsnp_mfma_kernel<COUNT>: compared-sites counts, one operand plane
#include "pairsnp_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

namespace tracs {

// ---------------------------------------------------------------------------------------
// IUPAC letter -> allele-set nibble (bit0=A,1=C,2=G,3=T).  load_seqs upper-cases, maps the
// 14 unambiguous/partial codes, and sends EVERYTHING else to all four bits
// (pairsnp.hpp:110,112-198).  26 nibbles packed in two 64-bit constants, index = letter-'A'.
__host__ __device__ __forceinline__ unsigned iupac_mask(unsigned ch)
{
    const unsigned up = ch & 0xDFu;            // 'a'..'z' -> 'A'..'Z'; nothing else lands in A..Z
    const unsigned idx = up - 'A';
    if (ch > 0x7Fu || idx > 25u) return 15u;
    //                       PONMLKJIHGFEDCBA                ......ZYXWVUTSRQ
    const unsigned long long lo = 0xFFF3FCFFB4FFD2E1ull, hi = 0xFFFFFFFAF97F865Full;
    return (unsigned)(((idx < 16u ? lo : hi) >> ((idx & 15u) * 4u)) & 15ull);
}

// ---------------------------------------------------------------------------------------
// pack: one thread = one (sample, 128-site group): 128 ASCII bytes -> 5 planes x uint4.
// Lanes run over samples so the 5 stores per thread are 1 KiB-coalesced per wave.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii, size_t L, size_t count,
                                                   size_t first, uint4 *__restrict__ planes, size_t n_pad,
                                                   size_t groups)
{
    const size_t s = (size_t)blockIdx.y * 64 + (threadIdx.x & 63);      // grid.y: sample blocks (<= 65535 x 64 per launch)
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // grid.x: group quads (any alignment length)
    if (s >= count || g >= groups) return;
    const size_t site0 = g * SITES_PER_GROUP;
    const uint8_t *src = ascii + s * L + site0;
    const unsigned nvalid = (unsigned)((L - site0) < (size_t)SITES_PER_GROUP ? (L - site0) : SITES_PER_GROUP);
    unsigned pl[NPLANES][4];
#pragma unroll
    for (int p = 0; p < NPLANES; p++)
#pragma unroll
        for (int w = 0; w < 4; w++) pl[p][w] = 0;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && nvalid == SITES_PER_GROUP;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned char chunk[32];
        if (aligned) {
            const uint4 *v = reinterpret_cast<const uint4 *>(src + w * 32);
            uint4 a = v[0], b = v[1];
            *reinterpret_cast<uint4 *>(&chunk[0]) = a;
            *reinterpret_cast<uint4 *>(&chunk[16]) = b;
        } else {
#pragma unroll
            for (int b = 0; b < 32; b++) {
                const unsigned idx = w * 32 + b;
                chunk[b] = idx < nvalid ? src[idx] : 0;
            }
        }
        unsigned A = 0, C = 0, G = 0, T = 0, N = 0;
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const unsigned idx = w * 32 + b;
            unsigned m = iupac_mask(chunk[b]);
            if (idx >= nvalid) m = 0;      // tail bits: no allele, not N => never match, never masked
            A |= (m & 1u) << b;
            C |= ((m >> 1) & 1u) << b;
            G |= ((m >> 2) & 1u) << b;
            T |= ((m >> 3) & 1u) << b;
            N |= (m == 15u ? 1u : 0u) << b;
        }
        pl[0][w] = A; pl[1][w] = C; pl[2][w] = G; pl[3][w] = T; pl[4][w] = N;
    }
#pragma unroll
    for (int p = 0; p < NPLANES; p++)
        planes[(g * NPLANES + p) * n_pad + first + s] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
}

// pack samples from 4-bit allele masks (two sites per byte, low nibble first; posterior_codes_kernel's output):
// one thread = one (sample, 128-site group) = 64 code bytes, lanes over samples like pack_kernel.
__global__ __launch_bounds__(256) void pack_codes_kernel(const uint8_t *__restrict__ codes, size_t stride, size_t L, size_t count,
                                                         size_t first, uint4 *__restrict__ planes, size_t n_pad, size_t groups)
{
    const size_t s = (size_t)blockIdx.y * 64 + (threadIdx.x & 63);
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= count || g >= groups) return;
    const uint8_t *src = codes + s * stride + g * (SITES_PER_GROUP / 2);
    const size_t site0 = g * SITES_PER_GROUP;
    const unsigned nvalid = (unsigned)((L - site0) < (size_t)SITES_PER_GROUP ? (L - site0) : SITES_PER_GROUP);
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && nvalid == SITES_PER_GROUP;
    unsigned pl[NPLANES][4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned cw[4];                                  // 32 sites = 16 bytes = 4 dwords of nibbles
        if (aligned) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + w * 16);
            cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned x = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const unsigned site = w * 32 + k * 8 + b * 2;          // first site of this byte
                    if (site < nvalid) x |= (unsigned)src[w * 16 + k * 4 + b] << (8 * b);
                }
                cw[k] = x;
            }
        }
        unsigned A = 0, C = 0, G = 0, T = 0, N = 0;
#pragma unroll
        for (int b = 0; b < 32; b++) {
            unsigned m = (cw[b >> 3] >> (4 * (b & 7))) & 15u;
            if (m == 0) m = 15u;                         // 'X' -> everything else -> all four alleles (pairsnp.hpp:192-197)
            if ((unsigned)(w * 32 + b) >= nvalid) m = 0; // tail bits: no allele, not N
            A |= (m & 1u) << b; C |= ((m >> 1) & 1u) << b; G |= ((m >> 2) & 1u) << b; T |= ((m >> 3) & 1u) << b;
            N |= (m == 15u ? 1u : 0u) << b;
        }
        pl[0][w] = A; pl[1][w] = C; pl[2][w] = G; pl[3][w] = T; pl[4][w] = N;
    }
#pragma unroll
    for (int p = 0; p < NPLANES; p++)
        planes[(g * NPLANES + p) * n_pad + first + s] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
}

// ---------------------------------------------------------------------------------------
// (x & v) | m in ONE VALU op.  Left to itself hipcc re-associates the four-plane OR into and, and_or, and, and, or3 (6 ops
// with the popcount); the asm pins and + 3 fused ops (5 ops).  Which fused op matters (scripts/micro/valu_ops.hip,
// profiles/r01/valu_ops_microbench.txt): gfx950's v_bitop3_b32 (any 3-input boolean function, truth table 0xEA = (a & b) | c)
// issues at the rate of a plain v_and_b32 when all three sources are VGPRs, v_and_or_b32 takes ~1.6x as long.
__device__ __forceinline__ unsigned and_or(unsigned x, unsigned v, unsigned m)
{
    unsigned r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(r) : "v"(x), "v"(v), "v"(m));
    return r;
}

__device__ __forceinline__ void pair_words(unsigned ai_a, unsigned ai_c, unsigned ai_g, unsigned ai_t,
                                           unsigned bj_a, unsigned bj_c, unsigned bj_g, unsigned bj_t, unsigned &acc)
{
    unsigned m = ai_a & bj_a;
    m = and_or(ai_c, bj_c, m);
    m = and_or(ai_g, bj_g, m);
    m = and_or(ai_t, bj_t, m);
    acc += __popc(m);                            // v_bcnt_u32_b32 (popcount + accumulate)
}

// The VALU tile kernel.  NW waves per workgroup, R rows per wave, C columns per lane, GC groups per LDS stage; the tile's
// TJ = 64 C column samples and TI = NW R row samples of GC groups are staged HBM/L2 -> LDS with global_load_lds_dwordx4
// (the group-major layout makes every wave-instruction's 64 x 16 B land in stage order), double-buffered, one barrier per
// stage; column words: lane-consecutive ds_read_b128 (conflict-free), row words: wave-uniform (broadcast) ds_read_b128.
// ENC = 0: general IUPAC encoding, 5 planes (A, C, G, T, N).
// ENC = 1: consensus encoding, 3 planes (X, Y, V): d = popc(((Xi^Xj)|(Yi^Yj)) & Vi & Vj), nn = popc(Vi & Vj).
template <int NW, int R, int C, int GC, bool WITH_NN, int MINW, int ENC>
__global__ __launch_bounds__(NW * 64, MINW) void pairsnp_tile_kernel(
    const uint4 *__restrict__ P, size_t n_pad, int groups, const int2 *__restrict__ tiles, int n_tiles,
    int groups_per_split, int ksplit, unsigned L, unsigned n, unsigned row_end, unsigned col_begin,
    unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld, unsigned thr, TilePhase ph)
{
    constexpr int NP = ENC ? 3 : NPLANES;                   // planes of this encoding
    constexpr int NT = NW * 64;
    constexpr int TI = NW * R;
    constexpr int TJ = 64 * C;
    constexpr int TS = TJ + TI;                             // samples staged per (group, plane)
    constexpr int STAGE = GC * NP * TS;                     // uint4 per LDS stage
    constexpr int LPT = (STAGE + NT - 1) / NT;              // staging loads per thread
    constexpr int NPL = ENC ? 3 : (WITH_NN ? NP : 4);       // planes actually read
    static_assert(TS % 64 == 0, "a staging wave-instruction must stay inside one (group, plane) run");
    __shared__ uint4 lds[2][STAGE];

    const unsigned q = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = (int)(q / (unsigned)n_tiles);
    const unsigned tile_no = q - (unsigned)ks * (unsigned)n_tiles;
    const int2 tile = tiles[tile_no];
    const int i0 = tile.x, j0 = tile.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g_begin = ph.g_base + ks * groups_per_split;
    const int g_end = min(groups, g_begin + groups_per_split);
    if (g_begin >= g_end) return;

    unsigned accM[R][C], accN[R][C];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < C; c++) { accM[r][c] = 0; accN[r][c] = 0; }

    // direct-to-LDS stage: LDS address = wave-uniform base + lane * 16 -- exactly the stage's linear order
    auto stage_glds = [&](int gs, int b) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e0 = (wave + k * NW) * 64;        // wave-uniform
            if (STAGE % NT == 0 || e0 < STAGE) {
                const int gp = e0 / TS;
                const int sidx = e0 - gp * TS + lane;
                if (gs + gp / NP < g_end) {
                    const size_t smp = sidx < TJ ? (size_t)j0 + sidx : (size_t)i0 + (sidx - TJ);
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(P + ((size_t)gs * NP + gp) * n_pad + smp),
                                                     (lds_void_t *)&lds[b][e0], 16, 0, 0);
                }
            }
        }
    };
    // one row (all planes) against the lane's C columns
    auto consume = [&](const uint4 (&ai)[NP], int r, const uint4 (&bj)[C][NP]) {
        if (ENC) {
#pragma unroll
            for (int c = 0; c < C; c++) {
#define TRACS_CONS_WORD(W)                                                                          \
                {                                                                                   \
                    const unsigned v = ai[2].W & bj[c][2].W;                                        \
                    if (WITH_NN) accN[r][c] += __popc(v);                                           \
                    accM[r][c] += __popc(((ai[0].W ^ bj[c][0].W) | (ai[1].W ^ bj[c][1].W)) & v);    \
                }
                TRACS_CONS_WORD(x) TRACS_CONS_WORD(y) TRACS_CONS_WORD(z) TRACS_CONS_WORD(w)
#undef TRACS_CONS_WORD
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            pair_words(ai[0].x, ai[1].x, ai[2].x, ai[3 % NP].x, bj[c][0].x, bj[c][1].x, bj[c][2].x, bj[c][3 % NP].x, accM[r][c]);
            pair_words(ai[0].y, ai[1].y, ai[2].y, ai[3 % NP].y, bj[c][0].y, bj[c][1].y, bj[c][2].y, bj[c][3 % NP].y, accM[r][c]);
            pair_words(ai[0].z, ai[1].z, ai[2].z, ai[3 % NP].z, bj[c][0].z, bj[c][1].z, bj[c][2].z, bj[c][3 % NP].z, accM[r][c]);
            pair_words(ai[0].w, ai[1].w, ai[2].w, ai[3 % NP].w, bj[c][0].w, bj[c][1].w, bj[c][2].w, bj[c][3 % NP].w, accM[r][c]);
        }
        if (WITH_NN) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                accN[r][c] += __popc(ai[4 % NP].x | bj[c][4 % NP].x);
                accN[r][c] += __popc(ai[4 % NP].y | bj[c][4 % NP].y);
                accN[r][c] += __popc(ai[4 % NP].z | bj[c][4 % NP].z);
                accN[r][c] += __popc(ai[4 % NP].w | bj[c][4 % NP].w);
            }
        }
    };

    stage_glds(g_begin, 0);
    __syncthreads();

    // Early out for thresholded runs (thr != ~0u): every 4th stage the workgroup takes the minimum, over its valid cells,
    // of the partial distance accumulated in ITS group range; partial distances only grow, so once that minimum exceeds the
    // threshold no pair of the tile can be emitted (src/pairsnp.hpp:405) and the rest of the range is skipped.  Cells then
    // hold 0xFFFFFFFF (whole alignment in one workgroup) or have bit 31 set (split range).  The per-wave minima are
    // double-buffered by check parity: a fast wave may start the next check while a slow one still reads this verdict.
    __shared__ unsigned wmin[2][NW];
    const bool can_exit = thr != 0xFFFFFFFFu && j0 >= i0 + TI;      // tiles touching the diagonal hold d(i,i) = 0 cells
    bool early = false;
    int stage_no = 0, n_checks = 0;

    int buf = 0;
    for (int gs = g_begin; gs < g_end; gs += GC) {
        const bool more = gs + GC < g_end;
        if (more) stage_glds(gs + GC, buf ^ 1);
#pragma unroll
        for (int gl = 0; gl < GC; gl++) {
            if (gs + gl < g_end) {                // wave-uniform
                uint4 bj[C][NP];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int p = 0; p < NPL; p++) bj[c][p] = lds[buf][(gl * NP + p) * TS + lane + 64 * c];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    uint4 ai[NP];
#pragma unroll
                    for (int p = 0; p < NPL; p++) ai[p] = lds[buf][(gl * NP + p) * TS + TJ + wave * R + r];
                    consume(ai, r, bj);
                }
            }
        }
        // checked every 4th stage, and at the end of a prefix pass (that verdict decides whether phase 2 visits the tile)
        const bool check = can_exit && (more ? ((++stage_no) & 3) == 0 : ph.phase == 1);
        const int par = n_checks & 1;
        if (check) {
            const unsigned done_sites = min(L, (unsigned)min(gs + GC, g_end) * SITES_PER_GROUP) - (unsigned)g_begin * SITES_PER_GROUP;
            unsigned mn = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool row_ok = (unsigned)(i0 + wave * R + r) < row_end;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const bool ok = row_ok && (unsigned)(j0 + lane + 64 * c) < n;
                    const unsigned dsofar = ENC ? accM[r][c] : done_sites - accM[r][c];
                    mn = min(mn, ok ? dsofar : 0xFFFFFFFFu);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, off, 64));
            if (lane == 0) wmin[par][wave] = mn;
            n_checks++;
        }
        __syncthreads();
        buf ^= 1;
        if (check) {
            unsigned m = wmin[par][0];
#pragma unroll
            for (int w = 1; w < NW; w++) m = min(m, wmin[par][w]);
            if (m > thr) { early = true; break; }
        }
    }

    // epilogue: d = sites - matches, nn = sites - masked over this workgroup's group range; only cells of the requested set
    // are written.  `single`: this workgroup is the cell's only writer and stores; otherwise partial counts are added.
    const bool single = ksplit == 1 && ph.phase != 2;
    // sites of this range (the last group is clipped to L); phase 0 keeps the whole-alignment form d = L - sum(matches)
    const unsigned Lc = ph.phase ? min(L, (unsigned)g_end * SITES_PER_GROUP) - (unsigned)g_begin * SITES_PER_GROUP : L;
    if (ph.phase == 1 && tid == 0) ph.live[tile_no] = early ? 0 : 1;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned i = (unsigned)(i0 + wave * R + r);
        if (i >= row_end) continue;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const unsigned j = (unsigned)(j0 + lane + 64 * c);
            if (j < n && j > i && j >= col_begin) {
                const size_t o = (size_t)i * ld + j;
                if (early) {                              // every pair of the tile is beyond the threshold
                    // shared cell: this range alone already exceeds the threshold; the other ranges keep adding their
                    // partial counts (< 2^31), so the marker bit survives whatever order the atomics land in
                    if (single) dist[o] = 0xFFFFFFFFu; else atomicOr(&dist[o], 0x80000000u);
                    if (WITH_NN && single) ncomp[o] = 0u;
                } else if (ENC) {                         // accumulators hold d and nn themselves
                    if (single) {
                        dist[o] = accM[r][c];
                        if (WITH_NN) ncomp[o] = accN[r][c];
                    } else {                              // cells were initialised to 0 (phase 0) / hold the prefix counts
                        atomicAdd(&dist[o], accM[r][c]);
                        if (WITH_NN) atomicAdd(&ncomp[o], accN[r][c]);
                    }
                } else if (single) {
                    dist[o] = Lc - accM[r][c];
                    if (WITH_NN) ncomp[o] = Lc - accN[r][c];
                } else if (ph.phase == 2) {               // cells hold the prefix counts
                    atomicAdd(&dist[o], Lc - accM[r][c]);
                    if (WITH_NN) atomicAdd(&ncomp[o], Lc - accN[r][c]);
                } else {                                  // cells were initialised to L
                    atomicSub(&dist[o], accM[r][c]);
                    if (WITH_NN) atomicSub(&ncomp[o], accN[r][c]);
                }
            }
        }
    }
}

// general planes -> consensus planes (+ flag: does any site carry a partial ambiguity code?)
__global__ __launch_bounds__(256) void derive_consensus_kernel(const uint4 *__restrict__ P, uint4 *__restrict__ Q, size_t n_pad,
                                                               size_t groups, unsigned *__restrict__ partial_flag)
{
    const size_t total = groups * n_pad;
    unsigned bad = 0;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t g = e / n_pad, s = e - g * n_pad;
        const uint4 A = P[(g * NPLANES + 0) * n_pad + s], Cc = P[(g * NPLANES + 1) * n_pad + s];
        const uint4 G = P[(g * NPLANES + 2) * n_pad + s], T = P[(g * NPLANES + 3) * n_pad + s];
        const uint4 N = P[(g * NPLANES + 4) * n_pad + s];
        uint4 X, Y, V;
#define TRACS_DERIVE(W)                                                                         \
        {                                                                                       \
            const unsigned two = (A.W & Cc.W) | (A.W & G.W) | (A.W & T.W) | (Cc.W & G.W) | (Cc.W & T.W) | (G.W & T.W); \
            bad |= two & ~N.W;                          /* 2 or 3 alleles set: IUPAC partial code */ \
            V.W = (A.W | Cc.W | G.W | T.W) & ~N.W;      /* exactly one allele (tail bits: none) */ \
            X.W = (Cc.W | T.W) & V.W;                   /* A=00 C=01 G=10 T=11 */              \
            Y.W = (G.W | T.W) & V.W;                                                            \
        }
        TRACS_DERIVE(x) TRACS_DERIVE(y) TRACS_DERIVE(z) TRACS_DERIVE(w)
#undef TRACS_DERIVE
        Q[(g * 3 + 0) * n_pad + s] = X;
        Q[(g * 3 + 1) * n_pad + s] = Y;
        Q[(g * 3 + 2) * n_pad + s] = V;
    }
    if (bad) atomicOr(partial_flag, 1u);
}

// live tiles of a prefix pass -> compact list (order does not matter: every tile owns its cells)
__global__ void compact_live_kernel(const int2 *__restrict__ tiles, const unsigned char *__restrict__ live, unsigned n_tiles,
                                    int2 *__restrict__ out, unsigned *__restrict__ n_out)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles && live[t]) out[atomicAdd(n_out, 1u)] = tiles[t];
}

// ncomp cells of the block += lu - c_i - c_j (site classes: the sites outside the pair kernel's, when neither the counting pass nor
// the list pass adds those terms; c == nullptr: no sample is N at any of them).  Rows go over grid.y with a stride, so a region
// of any height fits one launch (grid.y <= 65535).
__global__ void add_terms_kernel(unsigned *__restrict__ ncomp, size_t ld, unsigned n, unsigned row_begin, unsigned row_end,
                                 unsigned col_begin, unsigned lu, const unsigned *__restrict__ c)
{
    for (unsigned i = row_begin + blockIdx.y; i < row_end; i += gridDim.y) {
        const unsigned ci = c ? c[i] : 0u;
        for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
            if (j > i && j >= col_begin) ncomp[(size_t)i * ld + j] += lu - ci - (c ? c[j] : 0u);
    }
}

// cells of the block <- L (dist and, if given, ncomp; dist may be NULL: ncomp only)
__global__ void init_cells_kernel(unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld, unsigned n,
                                  unsigned row_begin, unsigned row_end, unsigned col_begin, unsigned L)
{
    for (unsigned i = row_begin + blockIdx.y; i < row_end; i += gridDim.y)
        for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
            if (j > i && j >= col_begin) {
                if (dist) dist[(size_t)i * ld + j] = L;
                if (ncomp) ncomp[(size_t)i * ld + j] = L;
            }
}

// ---------------------------------------------------------------------------------------
// COO extraction, row-major like combine_vectors (pairsnp.hpp:451-455).  One wave per row.
__global__ __launch_bounds__(64) void coo_count_kernel(const unsigned *__restrict__ dist, size_t ld, unsigned n,
                                                       unsigned row_begin, unsigned row_end, unsigned col_begin,
                                                       int thr, long long *__restrict__ counts)
{
    const unsigned i = row_begin + blockIdx.x;
    if (i >= row_end) return;
    const unsigned jb = max(col_begin, i + 1);
    long long c = 0;
    for (unsigned j = jb + threadIdx.x; j < n; j += 64) {
        const long long d = (long long)dist[(size_t)i * ld + j];
        if (d <= (long long)thr) c++;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}

// exclusive scan of per-row counts (rows <= a few 100k): single workgroup, serial over chunks
__global__ __launch_bounds__(1024) void scan_rows_kernel(long long *__restrict__ counts, size_t nrows)
{
    __shared__ long long part[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < nrows + 1; base += 1024) {
        const size_t idx = base + threadIdx.x;
        const long long v = idx < nrows ? counts[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            long long t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const long long incl = part[threadIdx.x];
        const long long c0 = carry;
        __syncthreads();
        if (idx <= nrows) counts[idx] = c0 + incl - v;      // exclusive
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void coo_fill_kernel(const unsigned *__restrict__ dist,
                                                      const unsigned *__restrict__ ncomp, size_t ld, unsigned n,
                                                      unsigned row_begin, unsigned row_end, unsigned col_begin,
                                                      int thr, const long long *__restrict__ offsets,
                                                      unsigned *__restrict__ rows, unsigned *__restrict__ cols,
                                                      unsigned *__restrict__ dd, unsigned *__restrict__ nn)
{
    const unsigned i = row_begin + blockIdx.x;
    if (i >= row_end) return;
    const unsigned jb = max(col_begin, i + 1);
    long long o = offsets[blockIdx.x];
    for (unsigned j0 = jb; j0 < n; j0 += 64) {
        const unsigned j = j0 + threadIdx.x;
        unsigned d = 0;
        bool keep = false;
        if (j < n) {
            d = dist[(size_t)i * ld + j];
            keep = (long long)d <= (long long)thr;
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const long long pos = o + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
            rows[pos] = i; cols[pos] = j; dd[pos] = d;
            if (nn) nn[pos] = ncomp ? ncomp[(size_t)i * ld + j] : 0u;
        }
        o += __popcll(mask);
    }
}

// two f64 panels (P(direct), E(K)) of the same cells, into the same COO positions as coo_fill_kernel's
__global__ __launch_bounds__(64) void coo_fill_f64_kernel(const unsigned *__restrict__ dist, size_t ld, unsigned n, unsigned row_begin, unsigned row_end,
                                                          unsigned col_begin, int thr, const long long *__restrict__ offsets,
                                                          const double *__restrict__ a, const double *__restrict__ b,
                                                          double *__restrict__ oa, double *__restrict__ ob)
{
    const unsigned i = row_begin + blockIdx.x;
    if (i >= row_end) return;
    const unsigned jb = max(col_begin, i + 1);
    long long o = offsets[blockIdx.x];
    for (unsigned j0 = jb; j0 < n; j0 += 64) {
        const unsigned j = j0 + threadIdx.x;
        const bool keep = j < n && (long long)dist[(size_t)i * ld + j] <= (long long)thr;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const long long pos = o + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
            oa[pos] = a[(size_t)i * ld + j]; ob[pos] = b[(size_t)i * ld + j];
        }
        o += __popcll(mask);
    }
}

// Threshold edges of a float64 matrix (E(K) or P(direct) panels; `tracs cluster -D expectedK|direct`, tracs/cluster.py:110-112):
// cells (i, j > i) whose SNP distance was emitted (dist <= dist_thr) and whose value is <= thr, row-major, one wave per row.
__global__ __launch_bounds__(64) void edge_count_f64_kernel(const double *__restrict__ val, const unsigned *__restrict__ dist, size_t ld,
                                                            unsigned n, unsigned row_begin, unsigned row_end, unsigned col_begin,
                                                            int dist_thr, double thr, long long *__restrict__ counts)
{
    const unsigned i = row_begin + blockIdx.x;
    if (i >= row_end) return;
    long long c = 0;
    for (unsigned j = max(col_begin, i + 1) + threadIdx.x; j < n; j += 64) {
        const size_t o = (size_t)i * ld + j;
        if ((long long)dist[o] <= (long long)dist_thr && val[o] <= thr) c++;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}

__global__ __launch_bounds__(64) void edge_fill_f64_kernel(const double *__restrict__ val, const unsigned *__restrict__ dist, size_t ld,
                                                           unsigned n, unsigned row_begin, unsigned row_end, unsigned col_begin,
                                                           int dist_thr, double thr, const long long *__restrict__ offsets,
                                                           unsigned *__restrict__ rows, unsigned *__restrict__ cols, double *__restrict__ vals)
{
    const unsigned i = row_begin + blockIdx.x;
    if (i >= row_end) return;
    long long o = offsets[blockIdx.x];
    for (unsigned j0 = max(col_begin, i + 1); j0 < n; j0 += 64) {
        const unsigned j = j0 + threadIdx.x;
        bool keep = false;
        double v = 0.0;
        if (j < n) {
            const size_t c = (size_t)i * ld + j;
            v = val[c];
            keep = (long long)dist[c] <= (long long)dist_thr && v <= thr;
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const long long pos = o + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
            rows[pos] = i; cols[pos] = j;
            if (vals) vals[pos] = v;
        }
        o += __popcll(mask);
    }
}



// tile schedule: upper-trapezoid tiles of the block, supertile-major so that consecutive
// entries (= tiles resident together on one XCD after xcd_remap) share row/column panels.
static void build_tiles(size_t n, size_t row_begin, size_t row_end, size_t col_begin, int ti, int tj,
                        std::vector<int2> &out)
{
    out.clear();
    if (row_end <= row_begin) return;
    const size_t nbi = (row_end - row_begin + ti - 1) / ti;
    const size_t nbj = (n + tj - 1) / tj;
    // tiles per supertile ~ one XCD's resident set; TRACS_SUPERTILE=<rows>x<cols> overrides (diagnostics)
    static const std::pair<size_t, size_t> shape = [] {
        std::pair<size_t, size_t> v{4, 8};
        if (const char *e = std::getenv("TRACS_SUPERTILE")) {
            unsigned a = 0, b = 0;
            if (std::sscanf(e, "%ux%u", &a, &b) == 2 && a >= 1 && b >= 1 && a <= 64 && b <= 64) v = {a, b};
        }
        return v;
    }();
    const size_t SI = shape.first, SJ = shape.second;
    for (size_t sbi = 0; sbi < nbi; sbi += SI)
        for (size_t sbj = 0; sbj < nbj; sbj += SJ)
            for (size_t bi = sbi; bi < std::min(nbi, sbi + SI); bi++)
                for (size_t bj = sbj; bj < std::min(nbj, sbj + SJ); bj++) {
                    const size_t i0 = row_begin + bi * ti, j0 = bj * tj;
                    const size_t jmax = std::min(n, j0 + tj) - 1;          // last column of the tile
                    const size_t jneed = std::max(col_begin, i0 + 1);      // first cell of the tile's first row
                    if (jmax < jneed) continue;                            // wholly below the diagonal / left of col_begin
                    out.push_back(make_int2((int)i0, (int)j0));
                }
}

}  // namespace tracs

using namespace tracs;

// VALU tile-kernel shapes (TRACS_TILE_VARIANT=<index> selects one for both encodings; diagnostics).  The defaults are the
// fastest measured on MI355X (profiles/r01/tile_variant_sweeps.txt): one per encoding.
typedef void (*TileLaunch)(bool with_nn, unsigned nwg, hipStream_t stream, const uint4 *P, size_t n_pad, int groups,
                           const int2 *tiles, int n_tiles, int gps, int ksplit, unsigned L, unsigned n, unsigned row_end,
                           unsigned col_begin, unsigned *dist, unsigned *ncomp, size_t ld, unsigned thr, TilePhase ph);
struct TileVariant {
    const char *name;
    int ti, tj, gc;
    int wg_per_cu[2];                  // resident workgroups per CU (VGPR/LDS bound), general / consensus
    TileLaunch launch[2];              // general (5 planes) / consensus (3 planes)
};

template <int NW, int R, int C, int GC, int MINW, int ENC>
static void launch_variant(bool with_nn, unsigned nwg, hipStream_t stream, const uint4 *P, size_t n_pad, int groups,
                           const int2 *tiles, int n_tiles, int gps, int ksplit, unsigned L, unsigned n, unsigned row_end,
                           unsigned col_begin, unsigned *dist, unsigned *ncomp, size_t ld, unsigned thr, TilePhase ph)
{
    if (with_nn)
        hipLaunchKernelGGL((pairsnp_tile_kernel<NW, R, C, GC, true, MINW, ENC>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups,
                           tiles, n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld, thr, ph);
    else
        hipLaunchKernelGGL((pairsnp_tile_kernel<NW, R, C, GC, false, MINW, ENC>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups,
                           tiles, n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld, thr, ph);
}
#define TRACS_VARIANT(NW, R, C, GC, MINW, WG, WC) {#NW "x" #R "x" #C "x" #GC "/w" #MINW, (NW) * (R), 64 * (C), GC, {WG, WC}, \
                                                   {launch_variant<NW, R, C, GC, MINW, 0>, launch_variant<NW, R, C, GC, MINW, 1>}}
static const TileVariant kVariants[] = {
    TRACS_VARIANT(8, 8, 2, 2, 4, 2, 2),      // 0: 64 x 128 tile, 8 waves            -- default, general encoding
    TRACS_VARIANT(4, 16, 1, 2, 4, 2, 5),     // 1: 64 x 64 tile, 4 waves, 94 VGPRs   -- default, consensus encoding
    TRACS_VARIANT(4, 16, 2, 2, 3, 2, 3),     // 2: 64 x 128 tile, 4 waves
    TRACS_VARIANT(8, 16, 2, 2, 2, 2, 2),     // 3: 128 x 128 tile
};
#undef TRACS_VARIANT
static constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

static const TileVariant &current_variant(bool consensus)
{
    static const int forced = [] {
        if (const char *e = std::getenv("TRACS_TILE_VARIANT")) { const int v = std::atoi(e); if (v >= 0 && v < kNumVariants) return v; }
        return -1;
    }();
    return kVariants[forced >= 0 ? forced : (consensus ? 1 : 0)];
}

static size_t plane_bytes(const tracs_alignment *a, int planes)
{
    return ((a->groups + PAD_GROUPS) * (size_t)planes * a->n_pad + TAIL_PAD) * sizeof(uint4);
}

namespace tracs {
hipError_t pack_alloc(tracs_alignment *a, size_t bytes, void **out)
{
    const size_t need = (bytes + 255) / 256 * 256;
    if (a->arena && a->arena_used + need <= a->arena_bytes) {
        *out = a->arena + a->arena_used;
        a->arena_used += need;
        return hipSuccess;
    }
    // a block the previous pack released, if one is large enough (the smallest such; not one more than twice the size)
    size_t best = a->pack_spare.size();
    for (size_t k = 0; k < a->pack_spare.size(); k++)
        if (a->pack_spare[k].bytes >= need && a->pack_spare[k].bytes <= 2 * need + (1u << 20) &&
            (best == a->pack_spare.size() || a->pack_spare[k].bytes < a->pack_spare[best].bytes)) best = k;
    if (best != a->pack_spare.size()) {
        *out = a->pack_spare[best].p;
        a->pack_extra.push_back(a->pack_spare[best]);
        a->pack_spare.erase(a->pack_spare.begin() + (long)best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, need);
    if (e != hipSuccess && !a->pack_spare.empty()) {
        // out of memory with blocks parked from the previous pack: give those back and try once more
        (void)hipGetLastError();
        for (auto &b : a->pack_spare) (void)hipFree(b.p);
        a->pack_spare.clear();
        e = hipMalloc(out, need);
    }
    if (e == hipSuccess) a->pack_extra.push_back({*out, need});
    else { *out = nullptr; a->pack_oom = true; }
    return e;
}
void pack_release(tracs_alignment *a)
{
    // (the spare blocks nobody took this time are freed: a handle keeps what its last pack needed beside the arena, not more --
    // and nothing at all after an allocation failed: what follows an out-of-memory soft fail needs the memory itself)
    for (auto &b : a->pack_spare) (void)hipFree(b.p);
    a->pack_spare.clear();
    if (a->pack_oom) { for (auto &b : a->pack_extra) (void)hipFree(b.p); a->pack_oom = false; }
    else a->pack_spare = std::move(a->pack_extra);
    a->pack_extra.clear();
    a->arena_used = 0;
}
}  // namespace tracs

extern "C" {

int tracs_debug_iupac_mask(int ch) { return (int)iupac_mask((unsigned)ch & 0xFFu); }

int tracs_alignment_create(size_t n, size_t L, tracs_alignment **out)
{
    if (!out) { set_error("tracs_alignment_create: out is NULL"); return TRACS_E_ARG; }
    *out = nullptr;
    if (n >= (1ull << 31) || L >= (1ull << 32)) { set_error("alignment too large (n < 2^31, L < 2^32)"); return TRACS_E_ARG; }
    auto *a = new tracs_alignment();
    a->n = n; a->L = L; a->n_pad = pad_samples(n); a->groups = groups_for(L);
    const size_t bytes = tracs_alignment_bytes(a);
    if (bytes) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&a->planes), bytes);
        if (e != hipSuccess) { set_error(std::string("hipMalloc(planes): ") + hipGetErrorString(e)); delete a; return TRACS_E_NOMEM; }
        e = hipMemset(a->planes, 0, bytes);
        if (e != hipSuccess) { set_error(std::string("hipMemset(planes): ") + hipGetErrorString(e)); (void)hipFree(a->planes); delete a; return TRACS_E_HIP; }
        // the arena of the once-per-pack structures: 30 % of the planes + up to 128 MiB covers the lists of alignments with N lists of
        // up to ~190 samples at every site of 10 000 samples -- as far as the cost model takes lists (TRACS_PACK_ARENA=<fraction>
        // overrides; 0: none).  Not getting it is not an error; what does not fit is allocated on its own.
        static const double frac = [] { const char *v = std::getenv("TRACS_PACK_ARENA"); return v ? std::atof(v) : 0.30; }();
        if (frac > 0.0 && n >= 2) {
            const size_t want = (size_t)((double)bytes * frac) + std::min<size_t>(128u << 20, 2 * bytes + (1u << 20));
            if (hipMalloc(reinterpret_cast<void **>(&a->arena), want) == hipSuccess) a->arena_bytes = want;
            else { a->arena = nullptr; (void)hipGetLastError(); }
        }
    }
    *out = a;
    return TRACS_OK;
}

void tracs_alignment_free(tracs_alignment *a)
{
    if (!a) return;
    general_sparse_free(a);
    site_classes_free(a);
    filter_index_free(a);
    for (auto &b : a->pack_spare) (void)hipFree(b.p);
    a->pack_spare.clear();
    if (a->arena) (void)hipFree(a->arena);
    if (a->planes) (void)hipFree(a->planes);
    if (a->cplanes) (void)hipFree(a->cplanes);
    if (a->d_flag) (void)hipFree(a->d_flag);
    for (auto &c : a->tile_cache)
        if (c.d) (void)hipFree(c.d);
    delete a;
}

size_t tracs_alignment_n(const tracs_alignment *a) { return a ? a->n : 0; }
size_t tracs_alignment_len(const tracs_alignment *a) { return a ? a->L : 0; }
size_t tracs_alignment_bytes(const tracs_alignment *a)
{
    if (!a || !a->n || !a->L) return 0;
    // + PAD_GROUPS all-zero groups (two-group stages may overhang) and TAIL_PAD entries of slack: row tiles start at
    // row_begin + k*TI and may read (never use) up to one tile edge past n_pad in the last (group, plane) run
    return plane_bytes(a, NPLANES);
}
void *tracs_alignment_planes(const tracs_alignment *a) { return a ? a->planes : nullptr; }

int tracs_alignment_touch(tracs_alignment *a)
{
    if (!a) { set_error("tracs_alignment_touch: NULL argument"); return TRACS_E_ARG; }
    DeviceCall guard(nullptr);
    a->dirty = true;                   // the consensus form / sparse lists (if any) must be re-derived
    a->flt_stale = true;
    return TRACS_OK;
}

int tracs_alignment_hint_rows(tracs_alignment *a, const size_t *ranges, int n_ranges)
{
    if (!a || n_ranges < 0 || n_ranges > 2 || (n_ranges && !ranges)) { set_error("tracs_alignment_hint_rows: 0 to 2 row ranges"); return TRACS_E_ARG; }
    for (int k = 0; k < n_ranges; k++)
        if (ranges[2 * k] > ranges[2 * k + 1]) { set_error("tracs_alignment_hint_rows: bad range"); return TRACS_E_ARG; }
    DeviceCall guard(nullptr);
    bool same = n_ranges == a->n_row_hint;
    for (int k = 0; same && k < 2 * n_ranges; k++) same = ranges[k] == a->row_hint[k];
    if (same) return TRACS_OK;
    a->n_row_hint = n_ranges;
    for (int k = 0; k < 4; k++) a->row_hint[k] = k < 2 * n_ranges ? ranges[k] : 0;
    a->dirty = true;                   // lists built for other rows (or for all of them) are re-decided
    return TRACS_OK;
}

int tracs_alignment_pack(tracs_alignment *a, const uint8_t *ascii, size_t first, size_t count, int ascii_on_device,
                         void *stream_)
{
    if (!a || (!ascii && count)) { set_error("tracs_alignment_pack: NULL argument"); return TRACS_E_ARG; }
    if (first + count > a->n) { set_error("tracs_alignment_pack: sample range outside the alignment"); return TRACS_E_ARG; }
    if (!count || !a->L) return TRACS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    a->dirty = true;                   // the consensus form / sparse lists (if any) must be re-derived
    a->flt_stale = true;
    const uint8_t *d_ascii = ascii;
    uint8_t *tmp = nullptr;
    if (!ascii_on_device) {
        TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&tmp), count * a->L));
        hipError_t e = hipMemcpyAsync(tmp, ascii, count * a->L, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { (void)hipFree(tmp); set_error(std::string("H2D ascii: ") + hipGetErrorString(e)); return TRACS_E_HIP; }
        d_ascii = tmp;
    }
    // samples go over grid.y in slices of 65535 x 64; `ascii`/`first` are advanced per slice
    const size_t slice = 65535ull * 64ull;
    for (size_t c0 = 0; c0 < count; c0 += slice) {
        const size_t cnt = std::min(slice, count - c0);
        dim3 grid((unsigned)((a->groups + 3) / 4), (unsigned)((cnt + 63) / 64));
        hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, stream, d_ascii + c0 * a->L, a->L, cnt, first + c0, a->planes, a->n_pad,
                           a->groups);
    }
    TRACS_HIP_CHECK(hipGetLastError());
    if (tmp) {
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        TRACS_HIP_CHECK(hipFree(tmp));
    }
    return TRACS_OK;
}

int tracs_alignment_pack_codes_batch(tracs_alignment *a, const uint8_t *codes, size_t stride_bytes, size_t first, size_t count,
                                     void *stream_)
{
    if (!a || (!codes && count)) { set_error("tracs_alignment_pack_codes: NULL argument"); return TRACS_E_ARG; }
    if (first + count > a->n) { set_error("tracs_alignment_pack_codes: sample index outside the alignment"); return TRACS_E_ARG; }
    if (count > 1 && stride_bytes < (a->L + 1) / 2) { set_error("tracs_alignment_pack_codes: stride shorter than one sample"); return TRACS_E_ARG; }
    if (!a->L || !count) return TRACS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    a->dirty = true;
    a->flt_stale = true;
    const size_t slice = 65535ull * 64ull;
    for (size_t c0 = 0; c0 < count; c0 += slice) {
        const size_t cnt = std::min(slice, count - c0);
        dim3 grid((unsigned)((a->groups + 3) / 4), (unsigned)((cnt + 63) / 64));
        hipLaunchKernelGGL(pack_codes_kernel, grid, dim3(256), 0, stream, codes + c0 * stride_bytes, stride_bytes, a->L, cnt, first + c0,
                           a->planes, a->n_pad, a->groups);
    }
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_alignment_pack_codes(tracs_alignment *a, const uint8_t *codes, size_t sample, void *stream_)
{
    if (!a || !codes) { set_error("tracs_alignment_pack_codes: NULL argument"); return TRACS_E_ARG; }
    return tracs_alignment_pack_codes_batch(a, codes, (a->L + 1) / 2, sample, 1, stream_);
}

const char *tracs_debug_tile_variant(void) { return current_variant(false).name; }
const char *tracs_debug_mfma_shape(void) { return mfma_shape(mfma_shape_current(false)).name; }
// 1 if the last dense call on this alignment used the consensus (3-plane) encoding, 0 general, -1 not decided yet
int tracs_debug_alignment_encoding(const tracs_alignment *a) { return !a ? -1 : (a->dirty ? -1 : a->enc); }
// kernel of the last dense call on this alignment: 0 VALU tile kernel, 1 matrix-core kernel (consensus operands),
// 2 matrix-core kernel (one-hot operands) + sparse partial-code correction, -1 none yet
int tracs_debug_alignment_kernel(const tracs_alignment *a) { return !a ? -1 : a->last_kernel; }

int tracs_debug_alignment_site_classes(const tracs_alignment *a, uint64_t *out)
{
    if (!a) return 0;
    if (out) {
        const bool on = a->classes_state == 1;
        out[0] = on ? a->L_var : 0; out[1] = on ? a->L_un : 0; out[2] = on ? a->L_minor : 0; out[3] = on ? a->L_full : 0;
    }
    return a->classes_state;
}

// 1 when the minority sites' N x listed terms of the last decided classes come from the matrix cores (nw_gram, site_classes.hip)
int tracs_debug_alignment_nw_gram(const tracs_alignment *a) { return (a && a->classes_state == 1 && a->nw_gram) ? (a->nw_rows ? 2 : 1) : 0; }

// what completes the compared-sites counts of the last decided classes: out[0] = sites the counting pass reads on the matrix
// cores, out[1] = 1 when that is the stored N plane in place, out[2] = sites whose N co-occurrences come from lists, out[3] = list
// entries one pass of that walk visits, out[4], out[5] = entries of the N lists / the listed-sample lists, out[6] = bytes per N entry,
// out[7] = list walks of one pass (one per N sample and site)
int tracs_debug_alignment_count_source(const tracs_alignment *a, uint64_t *out)
{
    if (!a || !out || a->classes_state != 1) return 0;
    out[0] = a->count_in_place ? a->L : a->L_inv;
    out[1] = a->count_in_place ? 1 : 0;
    out[2] = a->L_nnl;
    out[3] = a->nn_visits;
    out[4] = a->list_entries_n;
    out[5] = a->list_entries_p;
    out[6] = 1;                                             // bytes per N list entry (n8 lines: byte deltas)
    out[7] = a->nn_walks;
    return 1;
}

// Diagnostics for bench.py: HIP events on the launch stream around the four parts of a dense call (pair kernel incl. its cell
// initialisation / sparse partial-code correction + minority lists / counting pass on the matrix cores / N co-occurrence lists).
// Off by default.
static bool g_pair_timing = false;
constexpr int PAIR_EV_RING = 64;                       // the last 64 calls keep their events (bench.py averages over its timed steps)
static hipEvent_t g_pair_ev[PAIR_EV_RING][5] = {};
static unsigned long long g_pair_calls = 0;            // dense calls whose five events have all been recorded
static int g_pair_slot = 0;
void tracs_debug_pair_timing(int on)
{
    g_pair_timing = on != 0;
    if (g_pair_timing && !g_pair_ev[0][0])
        for (auto &set : g_pair_ev)
            for (auto &e : set) (void)hipEventCreate(&e);
}
// mean over the last `n_last` dense calls (at most PAIR_EV_RING, at most the calls made); n_last = 1: the last call
int tracs_debug_pair_ms_mean(int n_last, float *out)
{
    if (!out || n_last < 1 || g_pair_calls == 0) return TRACS_E_ARG;
    const int cnt = (int)std::min<unsigned long long>({(unsigned long long)n_last, g_pair_calls, (unsigned long long)PAIR_EV_RING});
    double acc[4] = {0, 0, 0, 0};
    for (int c = 0; c < cnt; c++) {
        const int slot = (int)((g_pair_calls - 1 - c) % PAIR_EV_RING);
        if (hipEventSynchronize(g_pair_ev[slot][4]) != hipSuccess) return TRACS_E_HIP;
        for (int k = 0; k < 4; k++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, g_pair_ev[slot][k], g_pair_ev[slot][k + 1]) != hipSuccess) return TRACS_E_HIP;
            acc[k] += ms;
        }
    }
    for (int k = 0; k < 4; k++) out[k] = (float)(acc[k] / cnt);
    return TRACS_OK;
}
int tracs_debug_last_pair_ms(float *out) { return tracs_debug_pair_ms_mean(1, out); }
// tracs_pairsnp_notify_distances: recorded by the next dense call once its distances are final
static thread_local hipEvent_t g_dist_event = nullptr;
void tracs_pairsnp_notify_distances(void *event) { g_dist_event = static_cast<hipEvent_t>(event); }
static inline void dist_final(hipStream_t stream)
{
    if (g_dist_event) { (void)hipEventRecord(g_dist_event, stream); g_dist_event = nullptr; }
}

static inline void pair_mark(int k, hipStream_t stream)
{
    if (!g_pair_timing || !g_pair_ev[0][0]) return;
    if (k == 0) g_pair_slot = (int)(g_pair_calls % PAIR_EV_RING);
    (void)hipEventRecord(g_pair_ev[g_pair_slot][k], stream);
    if (k == 4) g_pair_calls++;
}

static int pairsnp_dense_impl(const tracs_alignment *a_, size_t row_begin, size_t row_end, size_t col_begin, uint32_t *dist,
                              uint32_t *ncomp, size_t ld, void *stream_, unsigned thr);

int tracs_pairsnp_dense(const tracs_alignment *a_, size_t row_begin, size_t row_end, size_t col_begin, uint32_t *dist,
                        uint32_t *ncomp, size_t ld, void *stream_)
{
    return pairsnp_dense_impl(a_, row_begin, row_end, col_begin, dist, ncomp, ld, stream_, 0xFFFFFFFFu);
}

int tracs_pairsnp_dense_thr(const tracs_alignment *a_, size_t row_begin, size_t row_end, size_t col_begin, uint32_t *dist,
                            uint32_t *ncomp, size_t ld, int32_t dist_threshold, void *stream_)
{
    // a negative threshold emits nothing; any d > threshold may come back as 0xFFFFFFFF
    const unsigned thr = dist_threshold < 0 ? 0u : (unsigned)dist_threshold;
    return pairsnp_dense_impl(a_, row_begin, row_end, col_begin, dist, ncomp, ld, stream_, dist_threshold == 2147483647 ? 0xFFFFFFFFu : thr);
}

static int env_flag(const char *name)
{
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : -1;
}

static int pairsnp_dense_impl(const tracs_alignment *a_, size_t row_begin, size_t row_end, size_t col_begin, uint32_t *dist,
                              uint32_t *ncomp, size_t ld, void *stream_, unsigned thr)
{
    tracs_alignment *a = const_cast<tracs_alignment *>(a_);
    if (!a || !dist) { set_error("tracs_pairsnp_dense: NULL argument"); return TRACS_E_ARG; }
    if (row_end > a->n) row_end = a->n;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // an event armed by tracs_pairsnp_notify_distances belongs to THIS call: recorded when the distances are final, or -- on every
    // other way out, errors included -- recorded at the exit, never left armed for an unrelated later call
    struct DistEventGuard { hipStream_t s; ~DistEventGuard() { dist_final(s); } } dist_guard{stream};
    if (row_begin >= row_end || a->n < 2) return TRACS_OK;
    if (ld < a->n) { set_error("tracs_pairsnp_dense: ld < n"); return TRACS_E_ARG; }
    DeviceCall guard(stream);
    if (a->n_row_hint) {
        bool inside = false;
        for (int k = 0; k < a->n_row_hint; k++) inside = inside || (row_begin >= a->row_hint[2 * k] && row_end <= std::min(a->n, a->row_hint[2 * k + 1]));
        if (!inside) { set_error("tracs_pairsnp_dense: rows outside the ranges given to tracs_alignment_hint_rows"); return TRACS_E_ARG; }
    }

    if (a->L == 0) {   // every pair: d = 0, nn = 0
        dim3 grid(64, (unsigned)std::min<size_t>(row_end - row_begin, 65535));
        hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, dist, ncomp, ld, (unsigned)a->n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, 0u);
        dist_final(stream);
        TRACS_HIP_CHECK(hipGetLastError());
        return TRACS_OK;
    }

    // ---- kernel family: matrix cores unless switched off (TRACS_MFMA=0, or an explicit TRACS_TILE_VARIANT) ----------------------
    // (the matrix-core kernels keep 32-bit element offsets inside a stage: n_pad < 2^28 samples)
    static const bool mfma_env_off = env_flag("TRACS_MFMA") == 0 || std::getenv("TRACS_TILE_VARIANT") != nullptr;
    const bool mfma_off = mfma_env_off || a->n_pad >= (1ull << 28);

    // ---- once per pack: encoding and site classes, decided together from the general planes (site_classes.hip) -----------------
    // Classes in use: the pair kernels read `vplanes` (consensus form when no sample carries a partial IUPAC code), nothing else
    // is derived.  Classes not in use: the consensus planes (3 of 5: a second, smaller copy) are derived when the alignment has
    // a consensus form and there is room for them; otherwise the general planes are read as they are.
    if (a->dirty) {
        a->enc = 0;
        general_sparse_free(a);
        site_classes_free(a);
        pack_stage_begin(stream);
        int partial = -1;
        if (!mfma_off) {
            const int rc = site_classes_decide(a, stream, &partial);
            if (rc) { pack_stage_end(); return rc; }
        }
        static const bool force_general = std::getenv("TRACS_FORCE_GENERAL") != nullptr;
        if (a->classes_state == 1) {
            a->enc = a->classes_cons ? 1 : 0;
            if (a->cplanes) { TRACS_HIP_CHECK(hipFree(a->cplanes)); a->cplanes = nullptr; }
        } else if (!force_general && partial != 1) {
            const size_t cbytes = plane_bytes(a, 3);
            bool have = a->cplanes != nullptr;
            if (!have) {
                // no room for the second copy (very large alignments): stay on the general encoding
                if (hipMalloc(reinterpret_cast<void **>(&a->cplanes), cbytes) == hipSuccess) {
                    have = true;
                    TRACS_HIP_CHECK(hipMemsetAsync(a->cplanes, 0, cbytes, stream));
                } else {
                    a->cplanes = nullptr;
                    (void)hipGetLastError();
                }
            }
            if (have) {
                if (!a->d_flag) TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&a->d_flag), 64));
                TRACS_HIP_CHECK(hipMemsetAsync(a->d_flag, 0, 4, stream));
                hipLaunchKernelGGL(derive_consensus_kernel, dim3(256 * 16), dim3(256), 0, stream, a->planes, a->cplanes, a->n_pad,
                                   a->groups, a->d_flag);
                unsigned flag = 1;
                TRACS_HIP_CHECK(hipMemcpyAsync(&flag, a->d_flag, 4, hipMemcpyDeviceToHost, stream));
                TRACS_HIP_CHECK(hipStreamSynchronize(stream));
                if (flag == 0) a->enc = 1;
                else { TRACS_HIP_CHECK(hipFree(a->cplanes)); a->cplanes = nullptr; }
            }
            pack_stage_mark("consensus planes", stream);
        } else if (a->cplanes) {
            TRACS_HIP_CHECK(hipFree(a->cplanes));
            a->cplanes = nullptr;
        }
        pack_stage_end();
        a->dirty = false;
    }
    const bool cons = a->enc == 1;
    const TileVariant &V = current_variant(cons);
    const double cells = (double)(row_end - row_begin) * (double)a->n;      // upper bound of the cells of this call

    bool mfma = cons && !mfma_off;
    bool mfma_general = false;
    static const int gen_force = env_flag("TRACS_GENERAL_MFMA");          // 1: always, 0: never the matrix cores for general alignments
    for (int attempt = 0; attempt < 2 && !cons && !mfma_off && pair_L(a) < (1ull << 27) && pair_L(a) > 0; attempt++) {
        // general alignment: one-hot Gram on the matrix cores + the sparse partial-code terms, when the side lists exist (or
        // can be built) and the sparse work is small beside what the VALU kernel would cost.  Thresholded passes too: the
        // kernel's value bounds the distance from below, so tiles it declares dead are dead (pairsnp_mfma.hip)
        int ok = 0;
        double updates = 0.0;
        const int rc = general_sparse_get(a, stream, &ok, &updates);
        if (rc) return rc;
        if (ok) {
            const double all_cells = 0.5 * (double)a->n * (double)a->n;
            const double frac = std::min(1.0, cells / std::max(1.0, all_cells));
            const double t_valu = cells * (double)a->groups * 4.0 * 7.0 / 38e12;
            const double count_sites = a->classes_state != 1 ? 0.0 : (double)(a->count_in_place ? a->L : a->L_inv) + (a->nw_gram ? (double)a->L : 0.0);
            const double t_mfma = cells * ((double)pair_L(a) * 10.0 + count_sites * 2.0) / 5.4e15
                                  + updates * frac / 3.0e11;                  // measured: 5.4 PFLOP/s, 4 x 10^11 list entries/s
            mfma_general = gen_force == 1 || (gen_force != 0 && t_mfma < t_valu);
        }
        if (mfma_general || a->classes_state != 1) break;
        // the VALU kernel reads the whole alignment: drop the classes (and the lists built on them) and look again
        general_sparse_free(a);
        site_classes_free(a);
        a->classes_state = -1;
    }
    // (a general alignment whose variable sites all went to the minority lists has no dense site left: nothing for the pair
    // kernel to read, the lists and the counting pass do everything)
    const bool no_dense_site = !cons && !mfma_off && gen_force != 0 && a->classes_state == 1 && pair_L(a) == 0;
    if (!cons && !mfma_general && !no_dense_site && a->classes_state == 1) { general_sparse_free(a); site_classes_free(a); a->classes_state = -1; }
    mfma = mfma || mfma_general || no_dense_site;
    const bool classes = a->classes_state == 1;
    const int groups = (int)pair_groups(a);
    const int shape_id = mfma_shape_current(mfma_general);
    const MfmaShape &S = mfma_shape(shape_id);
    const int kGC = mfma ? (mfma_general ? S.gc_gen : S.gc_cons) : V.gc;
    const int kTI = mfma ? S.ti : V.ti, kTJ = mfma ? S.tj : V.tj;
    a->last_kernel = (mfma_general || no_dense_site) ? 2 : mfma ? 1 : 0;

    // ---- (re)build the cached tile schedule ------------------------------------------------------------------------------
    auto ensure_tiles = [&](int ti, int tj, tracs_alignment::TileCache **out) -> int {
        for (auto &c : a->tile_cache)
            if (c.rb == row_begin && c.re == row_end && c.cb == col_begin && c.ti == ti && c.tj == tj) { *out = &c; return TRACS_OK; }
        tracs_alignment::TileCache &c = a->tile_cache[a->tile_next];
        a->tile_next = (a->tile_next + 1) % 4;
        c.rb = (size_t)-1;                                   // not valid until filled
        std::vector<int2> tiles;
        build_tiles(a->n, row_begin, row_end, col_begin, ti, tj, tiles);
        if (tiles.size() > c.cap) {
            if (c.d) TRACS_HIP_CHECK(hipFree(c.d));
            c.d = nullptr;
            c.cap = tiles.size() * 2 + 64;
            TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&c.d), c.cap * sizeof(int2)));
        }
        if (!tiles.empty()) {
            TRACS_HIP_CHECK(hipMemcpyAsync(c.d, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
            TRACS_HIP_CHECK(hipStreamSynchronize(stream));   // `tiles` is a stack vector
        }
        c.n = tiles.size();
        c.rb = row_begin; c.re = row_end; c.cb = col_begin; c.ti = ti; c.tj = tj;
        *out = &c;
        return TRACS_OK;
    };
    tracs_alignment::TileCache *main_tiles = nullptr;
    { const int rc = ensure_tiles(kTI, kTJ, &main_tiles); if (rc) return rc; }
    const tracs_alignment::TileCache &T = *main_tiles;
    if (T.n == 0) { dist_final(stream); return TRACS_OK; }

    // Split the group range over workgroups (integer atomics, still exact) when that fills the chip better:
    // too few tiles (config 2), or a ragged last round of resident workgroups (tail effect).
    double slots = 512.0;
    {
        static int cus = 0;
        if (!cus) { hipDeviceProp_t pr; int dv = 0; (void)hipGetDevice(&dv); cus = (hipGetDeviceProperties(&pr, dv) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
        slots = (double)(mfma ? S.wg_per_cu : V.wg_per_cu[cons ? 1 : 0]) * cus;
    }
    auto pick_split = [&](size_t n_tiles, int range, int gc) {
        int pick = 1;
        const int max_split = std::max(1, range / (8 * gc));
        double best = -1.0;
        for (int k = 1; k <= std::min(max_split, 64); k++) {
            const double wgs = (double)n_tiles * k;
            const double rounds = std::ceil(wgs / slots);
            const double eff = wgs / (rounds * slots) - 0.002 * (k - 1);     // small bias towards fewer splits
            if (eff > best + 1e-9) { best = eff; pick = k; }
            if (rounds >= 40) break;                                         // tail < 2.5 % from here on
        }
        if (const char *e = std::getenv("TRACS_KSPLIT")) { const int v = std::atoi(e); if (v >= 1 && v <= max_split) pick = v; }
        return pick;
    };
    int ksplit = pick_split(T.n, groups, kGC);
    auto stage_split = [&](int range, int k, int &gps_out) {       // k workgroups over `range` groups, stage aligned
        int g = (range + k - 1) / k;
        g = (g + kGC - 1) / kGC * kGC;
        gps_out = g;
        return (range + g - 1) / g;
    };
    // fp32 accumulators of the matrix-core kernels are exact while every partial sum stays below 2^24:
    // consensus |S| <= 3 sites -> 2^22 sites per range; general G <= 4 sites -> 2^21
    const int max_gps = mfma ? (1 << (mfma_general ? 21 : 22)) / SITES_PER_GROUP : groups;
    // in-place counting source: the counting pass alone makes nn, the pair kernels (and the partial-code correction) write d only
    unsigned *const ncomp_pair = (classes && a->count_in_place) ? nullptr : ncomp;
    bool nn_zeroed = false;                                // this call has set the region's nn cells to 0
    auto launch = [&](const int2 *tl, unsigned nwg, int ntl, int g_end, int gps, int k, unsigned t, TilePhase ph) -> int {
        if (mfma) {
            MfmaArgs A;
            A.P = pair_planes(a, !mfma_general);
            A.n_pad = a->n_pad; A.groups = g_end; A.tiles = tl; A.n_tiles = ntl; A.gps = gps; A.ksplit = k;
            A.L = (unsigned)pair_L(a); A.n = (unsigned)a->n; A.row_end = (unsigned)row_end; A.col_begin = (unsigned)col_begin;
            A.dist = dist; A.ncomp = ncomp_pair; A.ld = ld; A.thr = t; A.ph = ph;
            A.keep_bound = (classes && a->lists) ? 1 : 0;      // terms are added to the cells afterwards: no flag values
            return launch_pairsnp_mfma(shape_id, mfma_general, nwg, stream, A);
        }
        V.launch[cons ? 1 : 0](ncomp != nullptr, nwg, stream, cons ? a->cplanes : a->planes, a->n_pad, g_end, tl, ntl, gps, k,
                               (unsigned)a->L, (unsigned)a->n, (unsigned)row_end, (unsigned)col_begin, dist, ncomp, ld, t, ph);
        return TRACS_OK;
    };

    // Site classes: the compared-sites counts of every site the pair kernel does not read.  Over those sites U (in place: over
    // every site, and the pair kernels wrote d only)   nn = |U| - c_i - c_j + NN,   NN = sum n_i n_j  (n = "is N here"):
    //   matrix cores  pairsnp_mfma_kernel<COUNT> over the N plane of the sites with many N samples (or of every site, in place),
    //                 for the tiles whose cells are complete (all, or the live ones of a thresholded run); ranges of at most 2^23
    //                 sites keep the fp32 partial sums exact, ranges add with integer atomics; range 0 adds |U| - c_i - c_j;
    //   lists         nn_rows_kernel over the sites with few N samples (cN^2 list entries per site);
    //   neither       sites with one N sample or none only add their part of |U| - c_i - c_j.
    // nw_gram (site_classes.hip): the same pass over the stored N plane also ADDS n n^T to the distances, and a second one over the
    // U plane subtracts U U^T -- the minority sites' N x listed terms; both run whether or not the caller wants nn.
    // (nw_rows: the same terms from the rows of the site-major N matrix inside the fix-up -- the passes below then count compared sites only)
    const bool gram = classes && a->nw_gram && !a->nw_rows;
    auto count_pass = [&](const int2 *tl, size_t ntl) -> int {
        if (!classes || (!ncomp && !gram) || (a->L_un == 0 && a->L_full == 0 && !gram)) { pair_mark(3, stream); return TRACS_OK; }
        const bool in_place = a->count_in_place;
        const unsigned lu = (unsigned)(in_place ? a->L : a->L_full + a->L_un);
        bool terms_added = false;
        if (a->L_inv > 0 || gram) {
            // tl == nullptr: every tile of the region, in the counting pass's own workgroup tile (its own cached schedule);
            // otherwise the given tiles (the live ones of a thresholded run) in the pair kernel's geometry
            CountShape C = count_shape_like(kTI, kTJ);
            if (tl && !C.fn) tl = nullptr;                     // no counting kernel with the pair kernel's tile: count every tile
            if (!tl) {
                C = count_shape_current();
                tracs_alignment::TileCache *ct = nullptr;
                const int rc = ensure_tiles(C.ti, C.tj, &ct);
                if (rc) return rc;
                tl = ct->d; ntl = ct->n;
            }
            const int gi = (int)(in_place ? a->groups : a->groups_inv), gcc = C.gc;
            const double keep_slots = slots;
            slots = (double)C.wg_per_cu * (slots / (double)(mfma ? S.wg_per_cu : V.wg_per_cu[cons ? 1 : 0]));
            int k = std::max(pick_split(std::max<size_t>(ntl, 1), gi, gcc), (gi + (1 << 16) - 1) >> 16);
            slots = keep_slots;
            int g = (gi + k - 1) / k;
            g = (g + gcc - 1) / gcc * gcc;
            k = (gi + g - 1) / g;
            if (in_place && k > 1 && !nn_zeroed && ncomp) {
                // several ranges add onto the cells: zero them first (nothing else has written nn) -- unless the call already has
                dim3 grid(64, (unsigned)std::min<size_t>(row_end - row_begin, 65535));
                hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, (unsigned *)nullptr, ncomp, ld, (unsigned)a->n,
                                   (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, 0u);
            }
            if (ntl) {
                MfmaArgs A;
                A.P = in_place ? a->planes + 4 * a->n_pad : a->iplanes;
                A.count_gp = in_place ? NPLANES : 1;
                A.count_store = (in_place && k == 1) ? 1 : 0;
                A.n_pad = a->n_pad; A.groups = gi; A.tiles = tl; A.n_tiles = (int)ntl; A.gps = g; A.ksplit = k;
                A.L = lu;                                      // range 0 adds |U| - c_i - c_j once per cell
                A.c_n = a->c_counted;
                A.n = (unsigned)a->n; A.row_end = (unsigned)row_end; A.col_begin = (unsigned)col_begin;
                A.dist = dist; A.ncomp = ncomp; A.ld = ld; A.thr = 0xFFFFFFFFu; A.ph = TilePhase{0, 0, nullptr};
                A.count_mode = gram ? 1 : 0;
                C.fn((unsigned)(ntl * (size_t)k), stream, A);
                if (gram) {
                    A.P = a->uplane; A.count_gp = 1; A.count_store = 0; A.c_n = nullptr; A.count_mode = 2;
                    C.fn((unsigned)(ntl * (size_t)k), stream, A);
                }
            }
            // (a thresholded run counts its live tiles only: the cells of the others keep whatever the pair kernel left -- ncomp of a
            // pair beyond the threshold is unspecified -- and the list pass below adds to every cell of the region all the same)
            terms_added = true;
        }
        pair_mark(3, stream);
        if (!ncomp) return TRACS_OK;                           // (gram without nn: the distances' passes were all there was to do)
        if (a->L_nnl > 0) {
            const int rc = nn_rows_add(a, row_begin, row_end, col_begin, ncomp, ld, terms_added ? 0 : 1, lu, stream);
            if (rc) return rc;
            terms_added = true;
        }
        if (!terms_added) {                                    // no pair of N samples anywhere outside the dense sites
            dim3 grid(64, (unsigned)std::min<size_t>(row_end - row_begin, 65535));
            hipLaunchKernelGGL(add_terms_kernel, grid, dim3(256), 0, stream, ncomp, ld, (unsigned)a->n, (unsigned)row_begin, (unsigned)row_end,
                               (unsigned)col_begin, lu, a->L_un ? a->c_counted : (const unsigned *)nullptr);
        }
        return TRACS_OK;
    };
    // Site classes, consensus form: the minority sites' distances from their lists (general_sparse.hip, general_fixup_kernel<MINOR>)
    auto minor_pass = [&]() -> int {
        return (classes && a->lists) ? minority_fixup(a, row_begin, row_end, col_begin, dist, ld, stream) : TRACS_OK;
    };
    if (classes && groups == 0) {
        // no dense site at all: the distances come from the lists, the compared-sites counts from the counting pass
        dim3 grid(64, (unsigned)std::min<size_t>(row_end - row_begin, 65535));
        pair_mark(0, stream);
        hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, dist, ncomp, ld, (unsigned)a->n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, 0u);
        nn_zeroed = true;
        pair_mark(1, stream);
        int rc = minor_pass();
        if (rc) return rc;
        pair_mark(2, stream);
        if (!gram) dist_final(stream);                         // (gram: the counting passes still add to the distances)
        if ((rc = count_pass(nullptr, 0))) return rc;
        pair_mark(4, stream);
        TRACS_HIP_CHECK(hipGetLastError());
        return TRACS_OK;
    }

    // Thresholded run on a long alignment: two passes.  The prefix pass (1/8 of the groups, one workgroup per tile) leaves
    // exact partial counts and a live flag per tile; the remainder pass visits the live tiles only, its range split so that the
    // few surviving tiles still fill the chip.  Dead tiles cost 1/8 of a full pass or less (they also stop inside the prefix).
    static const bool no_two_pass = std::getenv("TRACS_THR_ONE_PASS") != nullptr;
    if (thr != 0xFFFFFFFFu && !no_two_pass && groups >= 64 * kGC) {
        int prefix = std::min(max_gps / kGC * kGC, std::max(8 * kGC, groups / 8 / kGC * kGC));
        if (const char *e = std::getenv("TRACS_THR_PREFIX")) { const int v = std::atoi(e) / kGC * kGC; if (v >= kGC && v < groups) prefix = v; }
        unsigned char *live = nullptr;
        int2 *live_tiles = nullptr;
        unsigned *n_live_d = nullptr;
        int rc;
        if ((rc = tracs::workspace_get(48, T.n, reinterpret_cast<void **>(&live)))) return rc;
        if ((rc = tracs::workspace_get(49, T.n * sizeof(int2), reinterpret_cast<void **>(&live_tiles)))) return rc;
        if ((rc = tracs::workspace_get(50, 64, reinterpret_cast<void **>(&n_live_d)))) return rc;
        TRACS_HIP_CHECK(hipMemsetAsync(n_live_d, 0, 4, stream));
        pair_mark(0, stream);
        if ((rc = launch(T.d, (unsigned)T.n, (int)T.n, prefix, prefix, 1, thr, TilePhase{1, 0, live}))) return rc;
        hipLaunchKernelGGL(compact_live_kernel, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, stream, T.d, live,
                           (unsigned)T.n, live_tiles, n_live_d);
        unsigned n_live = 0;
        TRACS_HIP_CHECK(hipMemcpyAsync(&n_live, n_live_d, 4, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        if (n_live) {
            // enough workgroups for ~4 rounds of the chip, never more than 32 ranges (each range costs one atomic per cell)
            int gps2 = 0;
            const int want = (int)std::ceil(4.0 * slots / (double)n_live);
            const int k2 = stage_split(groups - prefix, std::max({1, std::min({32, want, (groups - prefix) / (16 * kGC)}), (groups - prefix + max_gps - 1) / max_gps}), gps2);
            if ((rc = launch(live_tiles, (unsigned)(n_live * (size_t)k2), (int)n_live, groups, gps2, k2, thr, TilePhase{2, prefix, nullptr}))) return rc;
        }
        pair_mark(1, stream);
        if (mfma_general && (rc = general_sparse_fixup(a, row_begin, row_end, col_begin, dist, ncomp_pair, ld, stream))) return rc;
        if ((rc = minor_pass())) return rc;
        pair_mark(2, stream);
        if (!gram) dist_final(stream);                         // (gram: the counting passes still add to the distances)
        if ((rc = count_pass(live_tiles, n_live))) return rc;
        pair_mark(4, stream);
        TRACS_HIP_CHECK(hipGetLastError());
        return TRACS_OK;
    }

    if (mfma) ksplit = std::max(ksplit, (groups + max_gps - 1) / max_gps);
    int gps = 0;
    ksplit = stage_split(groups, ksplit, gps);
    pair_mark(0, stream);
    if (ksplit > 1) {
        dim3 grid(64, (unsigned)std::min<size_t>(row_end - row_begin, 65535));
        hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, dist, ncomp, ld, (unsigned)a->n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, (cons || mfma) ? 0u : (unsigned)a->L);
        nn_zeroed = cons || mfma;
    }
    int rc = launch(T.d, (unsigned)(T.n * (size_t)ksplit), (int)T.n, groups, gps, ksplit, thr, TilePhase{0, 0, nullptr});
    if (rc) return rc;
    pair_mark(1, stream);
    if (mfma_general && (rc = general_sparse_fixup(a, row_begin, row_end, col_begin, dist, ncomp_pair, ld, stream))) return rc;
    if ((rc = minor_pass())) return rc;
    pair_mark(2, stream);
    if (!gram) dist_final(stream);
    if ((rc = count_pass(nullptr, 0))) return rc;
    pair_mark(4, stream);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_coo_count(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                    int32_t thr, int64_t *offsets, void *stream_)
{
    if (!dist || !offsets) { set_error("tracs_coo_count: NULL argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (row_end > n) row_end = n;
    const size_t nrows = row_end > row_begin ? row_end - row_begin : 0;
    if (nrows)
        hipLaunchKernelGGL(coo_count_kernel, dim3((unsigned)nrows), dim3(64), 0, stream, dist, ld, (unsigned)n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, (int)thr,
                           reinterpret_cast<long long *>(offsets));
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, stream, reinterpret_cast<long long *>(offsets), nrows);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_coo_fill(const uint32_t *dist, const uint32_t *ncomp, size_t ld, size_t n, size_t row_begin, size_t row_end,
                   size_t col_begin, int32_t thr, const int64_t *offsets, uint32_t *rows, uint32_t *cols, uint32_t *d,
                   uint32_t *nn, void *stream_)
{
    if (!dist || !offsets || !rows || !cols || !d) { set_error("tracs_coo_fill: NULL argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (row_end > n) row_end = n;
    const size_t nrows = row_end > row_begin ? row_end - row_begin : 0;
    if (!nrows) return TRACS_OK;
    hipLaunchKernelGGL(coo_fill_kernel, dim3((unsigned)nrows), dim3(64), 0, stream, dist, ncomp, ld, (unsigned)n,
                       (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, (int)thr,
                       reinterpret_cast<const long long *>(offsets), rows, cols, d, nn);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_coo_fill_f64(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin, int32_t thr,
                       const int64_t *offsets, const double *a, const double *b, double *out_a, double *out_b, void *stream_)
{
    if (!dist || !offsets || !a || !b || !out_a || !out_b) { set_error("tracs_coo_fill_f64: NULL argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (row_end > n) row_end = n;
    const size_t nrows = row_end > row_begin ? row_end - row_begin : 0;
    if (!nrows) return TRACS_OK;
    hipLaunchKernelGGL(coo_fill_f64_kernel, dim3((unsigned)nrows), dim3(64), 0, stream, dist, ld, (unsigned)n, (unsigned)row_begin, (unsigned)row_end,
                       (unsigned)col_begin, (int)thr, reinterpret_cast<const long long *>(offsets), a, b, out_a, out_b);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_edges_count_f64(const double *val, const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                          size_t col_begin, int32_t dist_threshold, double threshold, int64_t *offsets, void *stream_)
{
    if (!val || !dist || !offsets) { set_error("tracs_edges_count_f64: NULL argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (row_end > n) row_end = n;
    const size_t nrows = row_end > row_begin ? row_end - row_begin : 0;
    if (nrows)
        hipLaunchKernelGGL(edge_count_f64_kernel, dim3((unsigned)nrows), dim3(64), 0, stream, val, dist, ld, (unsigned)n, (unsigned)row_begin,
                           (unsigned)row_end, (unsigned)col_begin, (int)dist_threshold, threshold, reinterpret_cast<long long *>(offsets));
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, stream, reinterpret_cast<long long *>(offsets), nrows);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_edges_fill_f64(const double *val, const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                         size_t col_begin, int32_t dist_threshold, double threshold, const int64_t *offsets, uint32_t *rows,
                         uint32_t *cols, double *vals, void *stream_)
{
    if (!val || !dist || !offsets || !rows || !cols) { set_error("tracs_edges_fill_f64: NULL argument"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (row_end > n) row_end = n;
    const size_t nrows = row_end > row_begin ? row_end - row_begin : 0;
    if (!nrows) return TRACS_OK;
    hipLaunchKernelGGL(edge_fill_f64_kernel, dim3((unsigned)nrows), dim3(64), 0, stream, val, dist, ld, (unsigned)n, (unsigned)row_begin,
                       (unsigned)row_end, (unsigned)col_begin, (int)dist_threshold, threshold, reinterpret_cast<const long long *>(offsets),
                       rows, cols, vals);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

}  // extern "C"
