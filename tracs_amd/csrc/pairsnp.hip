// pairsnp.hip -- pack + all-pairs SNP/compared-sites kernels for gfx950 (CDNA4).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   load_seqs :107-199  (IUPAC -> four allele bit sets)        -> pack_kernel
//   pair loop :395-420  (match/popcount, compared sites)        -> pairsnp_tile_kernel
//   emit d <= dist :405, row-major order :451-455               -> coo_count/coo_fill
//
// Design (DESIGN.md "pairsnp kernel"): integer VALU-bound after tiling.  One workgroup owns a
// TI x TJ tile of the pair matrix and walks the alignment in 128-site groups:
//   * the TJ column samples of a group are staged through LDS (double buffered, 16 B/lane
//     coalesced loads, conflict-free ds_read_b128);
//   * each wave owns R rows whose words are WAVE-UNIFORM, so they are fetched with scalar
//     loads (s_load_dwordx4..x16 through the scalar cache) and enter the VALU as SGPR
//     operands -- no LDS traffic and no VGPRs for the row side;
//   * per 32 sites and pair: v_and, 3 x v_and_or, v_bcnt(+acc) for d; v_or, v_bcnt(+acc) for nn.
#include "common.h"

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdlib>
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

// pack one sample from 4-bit allele masks (two sites per byte, low nibble first): one thread per 128-site group
__global__ __launch_bounds__(256) void pack_codes_kernel(const uint8_t *__restrict__ codes, size_t L, size_t sample,
                                                         uint4 *__restrict__ planes, size_t n_pad, size_t groups)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    unsigned pl[NPLANES][4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned A = 0, C = 0, G = 0, T = 0, N = 0;
        for (int b = 0; b < 32; b++) {
            const size_t site = g * SITES_PER_GROUP + w * 32 + b;
            if (site < L) {
                unsigned m = (codes[site >> 1] >> (4 * (site & 1))) & 15u;
                if (m == 0) m = 15u;             // 'X' -> everything else -> all four alleles (pairsnp.hpp:192-197)
                A |= (m & 1u) << b; C |= ((m >> 1) & 1u) << b; G |= ((m >> 2) & 1u) << b; T |= ((m >> 3) & 1u) << b;
                N |= (m == 15u ? 1u : 0u) << b;
            }
        }
        pl[0][w] = A; pl[1][w] = C; pl[2][w] = G; pl[3][w] = T; pl[4][w] = N;
    }
#pragma unroll
    for (int p = 0; p < NPLANES; p++)
        planes[(g * NPLANES + p) * n_pad + sample] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
}

// ---------------------------------------------------------------------------------------
// XCD-aware, bijective remap of the hardware block id (guide T1): blocks b, b+8, b+16.. share
// an XCD (and its L2); give each XCD a contiguous run of the logical schedule so that the
// tiles resident on one XCD at a time are neighbours and share row/column panels in L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, k = b >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

// (x & v) | m in ONE VALU op.  Left to itself hipcc re-associates the four-plane OR into
// and, and_or, and, and, or3 (6 ops with the popcount); the asm pins and + 3 fused ops (5 ops).
// Which fused op matters (scripts/micro/valu_ops.hip, profiles/r01/valu_ops_microbench.txt): gfx950's
// v_bitop3_b32 (any 3-input boolean function, here truth table 0xEA = (a & b) | c) issues at the rate of a plain
// v_and_b32 when all three sources are VGPRs, while v_and_or_b32 / v_or3_b32 / v_bcnt_u32_b32 -- and ANY op with an
// SGPR source -- take ~1.6x as long.  SCALAR (row word in an SGPR) keeps v_and_or_b32: it is in the slow class anyway.
template <bool SCALAR>
__device__ __forceinline__ unsigned and_or(unsigned x, unsigned v, unsigned m)
{
    unsigned r;
    if (SCALAR) asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "s"(x), "v"(v), "v"(m));
    else asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(r) : "v"(x), "v"(v), "v"(m));
    return r;
}

template <bool SCALAR>
__device__ __forceinline__ void pair_words(unsigned ai_a, unsigned ai_c, unsigned ai_g, unsigned ai_t,
                                           unsigned bj_a, unsigned bj_c, unsigned bj_g, unsigned bj_t,
                                           unsigned &acc)
{
    unsigned m = ai_a & bj_a;                    // v_and_b32
    m = and_or<SCALAR>(ai_c, bj_c, m);           // v_and_or_b32
    m = and_or<SCALAR>(ai_g, bj_g, m);           // v_and_or_b32
    m = and_or<SCALAR>(ai_t, bj_t, m);           // v_and_or_b32
    acc += __popc(m);                            // v_bcnt_u32_b32 (popcount + accumulate)
}

// Two-pass thresholded runs (tracs_pairsnp_dense_thr on long alignments):
//   phase 1  "prefix":    one workgroup per tile walks groups [0, groups) of a SHORT prefix; a tile whose every pair already
//                         exceeds the threshold there is dead (cells 0xFFFFFFFF, live[tile] = 0), the others keep their exact
//                         partial counts in dist/ncomp (live[tile] = 1);
//   phase 2  "remainder": groups [g_base, groups) of the live tiles only (compacted tile list), split over ksplit workgroups
//                         that ADD their partial counts onto the prefix's.
//   phase 0  everything in one launch (the unthresholded path and short alignments).
struct TilePhase {
    int phase;
    int g_base;
    unsigned char *live;
};

// Where a wave's row words come from.
//   ROW_SMEM    scalar loads (s_load_dwordx8/16) straight from HBM/L2 through the scalar cache -> SGPR operands
//   ROW_SMEM_PF the same, software-pipelined: the next RB rows are requested before the current RB are consumed
//   ROW_LDS     rows staged in LDS next to the columns and read back with a wave-uniform (broadcast) ds_read -> VGPRs
enum { ROW_SMEM = 0, ROW_SMEM_PF = 1, ROW_LDS = 2, ROW_SMEM_PF1 = 6,
       // timing-only ablations (WRONG RESULTS; never the default): operands frozen outside the group loop
       ABL_ROWS_FIXED = 3, ABL_COLS_FIXED = 4, ABL_BOTH_FIXED = 5 };

// NW waves per workgroup, R rows per wave, C columns per lane, GC groups per LDS stage.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// GLDS: stage with global_load_lds_dwordx4 (HBM/L2 -> LDS directly, no staging VGPRs, no ds_write)
// ENC = 0: general IUPAC encoding, 5 planes (A, C, G, T, N), 24 issue cycles per 32 sites and pair.
// ENC = 1: consensus encoding, 3 planes (X = base bit 0, Y = base bit 1, V = site is an unambiguous base), valid when
//          every site of every sample is A/C/G/T or fully ambiguous: d = popc(((Xi^Xj)|(Yi^Yj)) & Vi & Vj),
//          nn = popc(Vi & Vj): VOP2 logic only, 18 issue cycles, 3/5 of the bytes.
template <int NW, int R, int C, int GC, bool WITH_NN, int ROWSRC, int MINW = 1, bool GLDS = false, int ENC = 0>
__global__ __launch_bounds__(NW * 64, MINW) void pairsnp_tile_kernel(
    const uint4 *__restrict__ P, size_t n_pad, int groups, const int2 *__restrict__ tiles, int n_tiles,
    int groups_per_split, int ksplit, unsigned L, unsigned n, unsigned row_end, unsigned col_begin,
    unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld, unsigned thr, TilePhase ph)
{
    constexpr int NP = ENC ? 3 : NPLANES;                   // planes of this encoding
    constexpr int NT = NW * 64;
    constexpr int TI = NW * R;
    constexpr int TJ = 64 * C;
    constexpr int TS = TJ + (ROWSRC == ROW_LDS ? TI : 0);   // samples staged per (group, plane)
    constexpr int STAGE = GC * NP * TS;                // uint4 per LDS stage
    constexpr int LPT = (STAGE + NT - 1) / NT;              // staging loads per thread
    constexpr bool SC = ROWSRC != ROW_LDS;   // row operands are scalar (SGPR)
    constexpr int NPL = ENC ? 3 : (WITH_NN ? NP : 4);    // planes actually read
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

    static_assert(!GLDS || TS % 64 == 0, "glds needs wave-uniform (group, plane) per wave instruction");
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
    uint4 stage_regs[GLDS ? 1 : LPT];
    auto stage_load = [&](int gs) {
#pragma unroll
        for (int k = 0; k < (GLDS ? 0 : LPT); k++) {
            const int e = tid + k * NT;
            const int gp = e / TS;               // local group*5 + plane
            const int sidx = e - gp * TS;
            const int g = gs + gp / NP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((STAGE % NT == 0 || e < STAGE) && g < g_end) {
                const size_t smp = sidx < TJ ? (size_t)j0 + sidx : (size_t)i0 + (sidx - TJ);
                v = P[((size_t)gs * NP + gp) * n_pad + smp];
            }
            stage_regs[k] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < (GLDS ? 0 : LPT); k++) {
            const int e = tid + k * NT;
            if (STAGE % NT == 0 || e < STAGE) lds[buf][e] = stage_regs[k];
        }
    };
    // one row batch (RB rows, all planes) against the lane's C columns
    auto consume = [&](const uint4 (*ai)[NP], int r0, int nrows, const uint4 (&bj)[C][NP]) {
#pragma unroll
        for (int rr = 0; rr < nrows; rr++) {
            const int r = r0 + rr;
            if (ENC) {
#pragma unroll
                for (int c = 0; c < C; c++) {
#define TRACS_CONS_WORD(W)                                                                              \
                    {                                                                                   \
                        const unsigned v = ai[rr][2].W & bj[c][2].W;                                    \
                        if (WITH_NN) accN[r][c] += __popc(v);                                           \
                        accM[r][c] += __popc(((ai[rr][0].W ^ bj[c][0].W) | (ai[rr][1].W ^ bj[c][1].W)) & v); \
                    }
                    TRACS_CONS_WORD(x) TRACS_CONS_WORD(y) TRACS_CONS_WORD(z) TRACS_CONS_WORD(w)
#undef TRACS_CONS_WORD
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                pair_words<SC>(ai[rr][0].x, ai[rr][1].x, ai[rr][2].x, ai[rr][3 % NP].x, bj[c][0].x, bj[c][1].x, bj[c][2].x, bj[c][3 % NP].x, accM[r][c]);
                pair_words<SC>(ai[rr][0].y, ai[rr][1].y, ai[rr][2].y, ai[rr][3 % NP].y, bj[c][0].y, bj[c][1].y, bj[c][2].y, bj[c][3 % NP].y, accM[r][c]);
                pair_words<SC>(ai[rr][0].z, ai[rr][1].z, ai[rr][2].z, ai[rr][3 % NP].z, bj[c][0].z, bj[c][1].z, bj[c][2].z, bj[c][3 % NP].z, accM[r][c]);
                pair_words<SC>(ai[rr][0].w, ai[rr][1].w, ai[rr][2].w, ai[rr][3 % NP].w, bj[c][0].w, bj[c][1].w, bj[c][2].w, bj[c][3 % NP].w, accM[r][c]);
            }
            if (WITH_NN) {
#pragma unroll
                for (int c = 0; c < C; c++) {
                    accN[r][c] += __popc(ai[rr][4 % NP].x | bj[c][4 % NP].x);
                    accN[r][c] += __popc(ai[rr][4 % NP].y | bj[c][4 % NP].y);
                    accN[r][c] += __popc(ai[rr][4 % NP].z | bj[c][4 % NP].z);
                    accN[r][c] += __popc(ai[rr][4 % NP].w | bj[c][4 % NP].w);
                }
            }
        }
    };
    constexpr bool PF = ROWSRC == ROW_SMEM_PF || ROWSRC == ROW_SMEM_PF1;
    constexpr int RB = (ROWSRC == ROW_SMEM_PF) ? 2 : ((ROWSRC == ROW_LDS || ROWSRC == ROW_SMEM_PF1) ? 1 : (R >= 4 ? 4 : R));   // rows per batch
    constexpr int NB = R / RB;
    static_assert(R % RB == 0, "row batch must divide R");
    const size_t row0 = (size_t)(i0 + wave * R);
    auto load_rows_global = [&](int g, int b, uint4 (*dst)[NP]) {   // wave-uniform addresses: scalar loads
        const uint4 *rowp = P + (size_t)g * NP * n_pad + row0 + (size_t)(b * RB);
#pragma unroll
        for (int rr = 0; rr < RB; rr++)
#pragma unroll
            for (int p = 0; p < NPL; p++) dst[rr][p] = rowp[(size_t)p * n_pad + rr];
    };

    if (GLDS) stage_glds(g_begin, 0);
    else { stage_load(g_begin); stage_store(0); }
    __syncthreads();

    uint4 cur[RB][NP];
    if (PF && g_begin < g_end) load_rows_global(g_begin, 0, cur);
    constexpr bool ROWS_FIXED = ROWSRC == ABL_ROWS_FIXED || ROWSRC == ABL_BOTH_FIXED;
    constexpr bool COLS_FIXED = ROWSRC == ABL_COLS_FIXED || ROWSRC == ABL_BOTH_FIXED;
    uint4 fixed_rows[NB][RB][NP];
    uint4 fixed_bj[C][NP];
    if (ROWS_FIXED) {
#pragma unroll
        for (int b = 0; b < NB; b++) load_rows_global(g_begin, b, fixed_rows[b]);
    }
    if (COLS_FIXED) {
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
            for (int p = 0; p < NPL; p++) fixed_bj[c][p] = lds[0][p * TS + lane + 64 * c];
    }

    // Early out for thresholded runs (thr != ~0u): every 4th stage the workgroup takes the minimum, over its valid cells,
    // of the partial distance accumulated in ITS group range; partial distances only grow, so once that minimum exceeds the
    // threshold no pair of the tile can be emitted (src/pairsnp.hpp:405) and the rest of the range is skipped.  Cells then
    // hold 0xFFFFFFFF (whole alignment in one workgroup) or have bit 31 set (split range).
    __shared__ unsigned wmin[NW];
    const bool can_exit = thr != 0xFFFFFFFFu && j0 >= i0 + TI;      // tiles touching the diagonal hold d(i,i) = 0 cells
    bool early = false;
    int stage_no = 0;

    int buf = 0;
    for (int gs = g_begin; gs < g_end; gs += GC) {
        const bool more = !COLS_FIXED && gs + GC < g_end;
        if (more) { if (GLDS) stage_glds(gs + GC, buf ^ 1); else stage_load(gs + GC); }
#pragma unroll
        for (int gl = 0; gl < GC; gl++) {
            const int g = gs + gl;
            if (g < g_end) {                      // wave-uniform
                uint4 bj[C][NP];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        if (COLS_FIXED) { bj[c][p] = fixed_bj[c][p]; bj[c][p].x ^= (unsigned)g; }   // keep it group-dependent
                        else bj[c][p] = lds[buf][(gl * NP + p) * TS + lane + 64 * c];
                    }
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    if (ROWS_FIXED) {
                        consume(fixed_rows[b], b * RB, RB, bj);
                    } else if (ROWSRC == ROW_SMEM || ROWSRC == ABL_COLS_FIXED) {
                        load_rows_global(g, b, cur);
                        consume(cur, b * RB, RB, bj);
                    } else if (PF) {
                        uint4 nxt[RB][NP];
#pragma unroll
                        for (int rr = 0; rr < RB; rr++)
#pragma unroll
                            for (int p = 0; p < NPL; p++) nxt[rr][p] = cur[rr][p];
                        if (b + 1 < NB) load_rows_global(g, b + 1, nxt);
                        else if (g + 1 < g_end) load_rows_global(g + 1, 0, nxt);
                        __builtin_amdgcn_sched_barrier(0);      // keep the next batch's loads ahead of this batch's VALU
                        consume(cur, b * RB, RB, bj);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rr = 0; rr < RB; rr++)
#pragma unroll
                            for (int p = 0; p < NPL; p++) cur[rr][p] = nxt[rr][p];
                    } else {
#pragma unroll
                        for (int p = 0; p < NPL; p++) cur[0][p] = lds[buf][(gl * NP + p) * TS + TJ + wave * R + b];
                        consume(cur, b, 1, bj);
                    }
                }
            }
        }
        if (more) stage_store(buf ^ 1);
        // checked every 4th stage, and at the end of a prefix pass (that verdict decides whether phase 2 visits the tile)
        const bool check = can_exit && (more ? ((++stage_no) & 3) == 0 : ph.phase == 1);
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
            if (lane == 0) wmin[wave] = mn;
        }
        if (!COLS_FIXED) { __syncthreads(); buf ^= 1; }
        if (check) {
            unsigned m = wmin[0];
#pragma unroll
            for (int w = 1; w < NW; w++) m = min(m, wmin[w]);
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

// ---------------------------------------------------------------------------------------
// Matrix-core form of the CONSENSUS pair loop (default for plain passes; TRACS_MFMA=0 disables): the pair loop is integer-VALU-bound
// (DESIGN.md 3.1), and it is a Gram matrix.  With every base as three signs  x = (-1)^X, y = (-1)^Y, z = x*y  (all three 0
// where the site is not a base) two samples contribute  x x' + y y' + z z' = +3  at a site where they agree and  -1  where
// they differ, so over a site range   S = 4*matches - nn,   nn = sum v v',   d = nn - matches = (3 nn - S) / 4.
// S and nn are accumulated by v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 (E2M1) operands: +1.0 = 0x2, -1.0 = 0xA, 0 = 0x0,
// scale 2^0.  Products and partial sums are integers below 2^24 (the host limits a workgroup's range to 2^22 sites), so
// the fp32 accumulators are exact and the result is bit-identical to the VALU kernel's.
// Operands are expanded in registers from the 3 bit planes, never stored: the K index of the MFMA is ours to choose as
// long as both operands agree, so dword q of a 32-site chunk takes the sites whose bit index is = q (mod 4) and the
// expansion is "mask, shift, or" -- 32 VALU ops per (sample, 32 sites) for all four operand planes (x, y, z, v).
// One workgroup = 4 waves = a 128 x 128 tile (each wave 64 x 64 = 2 x 2 MFMA blocks, two accumulator sets); the bit planes
// of the tile's 256 samples are staged HBM -> LDS directly like in the VALU kernel.
typedef int mfma_v8i __attribute__((ext_vector_type(8)));
typedef float mfma_v16f __attribute__((ext_vector_type(16)));

struct Fp4Planes { unsigned x[4], y[4], z[4], v[4]; };

// dword q of a chunk holds the sites with bit index = q (mod 4): nibble = (sign & valid) << 3 | valid << 1.
// (A v_bitop3_b32 form with the masks in VGPRs -- every op in the fast issue class -- measured 3 % SLOWER: the kernel is
// paced by the matrix pipe, not by VALU issue.)
__device__ __forceinline__ void expand_fp4(unsigned X, unsigned Y, unsigned V, Fp4Planes &o)
{
    const unsigned tx = X, ty = Y, tz = X ^ Y;          // the consensus planes are stored masked: X = Y = 0 where V = 0
    o.v[0] = (V & 0x11111111u) << 1;
    o.v[1] = V & 0x22222222u;
    o.v[2] = (V & 0x44444444u) >> 1;
    o.v[3] = (V & 0x88888888u) >> 2;
#define TRACS_SIGN_PLANE(T, O)                              \
    O[0] = ((T & 0x11111111u) << 3) | o.v[0];               \
    O[1] = ((T & 0x22222222u) << 2) | o.v[1];               \
    O[2] = ((T & 0x44444444u) << 1) | o.v[2];               \
    O[3] = (T & 0x88888888u) | o.v[3];
    TRACS_SIGN_PLANE(tx, o.x) TRACS_SIGN_PLANE(ty, o.y) TRACS_SIGN_PLANE(tz, o.z)
#undef TRACS_SIGN_PLANE
}

__device__ __forceinline__ mfma_v8i fp4_operand(const unsigned (&w)[4])
{
    mfma_v8i r = {(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
    return r;
}

typedef float mfma_v4f __attribute__((ext_vector_type(4)));

// MS = 32: v_mfma_scale_f32_32x32x64 (a lane's operand = sample l & 31, 32-site word 2 * step + (l >> 5); two steps per group)
// MS = 16: v_mfma_scale_f32_16x16x128 (sample l & 15, word l >> 4; one step per 128-site group).  Same flops per clock on
//          paper; the bare 16 x 16 instruction holds a higher clock on this kernel's data (profiles/r01/mfma_fp4_rate.txt: 8.4 vs
//          7.5-7.9 PFLOP/s) but the whole kernel is 3.5 % slower with it (31.9 vs 30.7 ms per 400 kbp): TRACS_MFMA_SHAPE=16.
template <int GC, bool WITH_NN, int ABL = 0, int MS = 32>
__global__ __launch_bounds__(256, 2) void pairsnp_mfma_kernel(
    const uint4 *__restrict__ P, size_t n_pad, int groups, const int2 *__restrict__ tiles, int n_tiles,
    int groups_per_split, int ksplit, unsigned n, unsigned row_end, unsigned col_begin,
    unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld, int scale, unsigned thr, TilePhase ph)
{
    constexpr int NP = 3, NW = 4, TI = 128, TJ = 128, TS = TI + TJ, NT = NW * 64;
    constexpr int STAGE = GC * NP * TS;
    constexpr int LPT = STAGE / NT;
    constexpr int NB = 64 / MS;                  // MFMA blocks per side of a wave's 64 x 64 tile
    constexpr int STEPS = MS == 32 ? 2 : 1;      // matrix steps per 128-site group
    constexpr int AR = MS == 32 ? 16 : 4;        // accumulator registers per block
    using AccT = typename std::conditional<MS == 32, mfma_v16f, mfma_v4f>::type;
    static_assert(STAGE % NT == 0, "stage must divide over the workgroup");
    __shared__ uint4 lds[2][STAGE];

    const unsigned q = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = (int)(q / (unsigned)n_tiles);
    const unsigned tile_no = q - (unsigned)ks * (unsigned)n_tiles;
    const int2 tile = tiles[tile_no];
    const int i0 = tile.x, j0 = tile.y;
    const int tid = threadIdx.x, lane = tid & 63, lb = lane & (MS - 1), hk = lane / MS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g_begin = ph.g_base + ks * groups_per_split;
    const int g_end = min(groups, g_begin + groups_per_split);
    if (g_begin >= g_end) return;

    auto stage_glds = [&](int gs, int b) {
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int e0 = (wave + k * NW) * 64;
            const int gp = e0 / TS;
            const int sidx = e0 - gp * TS + lane;
            // groups past this workgroup's range are read from the zeroed tail behind the last plane (never packed into):
            // zero words expand to zero operands, so the compute loop needs no range branch
            const size_t smp = sidx < TJ ? (size_t)j0 + sidx : (size_t)i0 + (sidx - TJ);
            const uint4 *src = gs + gp / NP < g_end ? P + ((size_t)gs * NP + gp) * n_pad + smp : P + (size_t)groups * NP * n_pad + sidx;
            __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)&lds[b][e0], 16, 0, 0);
        }
    };
    // this lane's sample inside a staged (group, plane) row, per row block / column block of the wave's tile
    const int row_slot = TJ + wr * 64 + lb, col_slot = wc * 64 + lb;

    AccT accS[NB][NB], accV[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int r = 0; r < AR; r++) { accS[a][b][r] = 0.0f; accV[a][b][r] = 0.0f; }

    auto load_expand = [&](int buf, int gl, int slot, int word, Fp4Planes &o) {
        const unsigned *wx = reinterpret_cast<const unsigned *>(&lds[buf][(gl * NP + 0) * TS + slot]);
        const unsigned *wy = reinterpret_cast<const unsigned *>(&lds[buf][(gl * NP + 1) * TS + slot]);
        const unsigned *wv = reinterpret_cast<const unsigned *>(&lds[buf][(gl * NP + 2) * TS + slot]);
        if (ABL == 1) {   // timing only: no expansion
            const unsigned a0 = wx[word], a1 = wy[word], a2 = wv[word];
            for (int k = 0; k < 4; k++) { o.x[k] = a0; o.y[k] = a1; o.z[k] = a2; o.v[k] = a0 ^ a1; }
        } else {
            expand_fp4(wx[word], wy[word], wv[word], o);
        }
    };

    stage_glds(g_begin, 0);
    __syncthreads();
    int buf = 0;
    for (int gs = g_begin; gs < g_end; gs += GC) {
        const bool more = gs + GC < g_end;
        if (more) stage_glds(gs + GC, buf ^ 1);
#pragma unroll
        for (int gl = 0; gl < GC; gl++) {
#pragma unroll
            for (int st = 0; st < STEPS; st++) {
                const int word = MS == 32 ? 2 * st + hk : hk;
                Fp4Planes rows[NB];
#pragma unroll
                for (int rb = 0; rb < NB; rb++) load_expand(buf, gl, row_slot + rb * MS, word, rows[rb]);
#pragma unroll
                for (int cb = 0; cb < NB; cb++) {
                    Fp4Planes col;
                    load_expand(buf, gl, col_slot + cb * MS, word, col);
#pragma unroll
                    for (int rb = 0; rb < NB; rb++) {
                        if (ABL == 2) {   // timing only: no matrix instructions
                            accS[rb][cb][0] += __int_as_float((rows[rb].x[0] ^ col.x[1]) + (rows[rb].y[2] ^ col.y[3]) + (rows[rb].z[0] ^ col.z[1]) + (rows[rb].v[2] ^ col.v[3])
                                                              + (rows[rb].x[2] ^ col.x[3]) + (rows[rb].y[0] ^ col.y[1]) + (rows[rb].z[2] ^ col.z[3]) + (rows[rb].v[0] ^ col.v[1]));
                            continue;
                        }
#define TRACS_MFMA(ACC, PL)                                                                                                                  \
    if constexpr (MS == 32)                                                                                                                  \
        ACC[rb][cb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_operand(rows[rb].PL), fp4_operand(col.PL), ACC[rb][cb], 4, 4, 0, scale, 0, scale); \
    else                                                                                                                                     \
        ACC[rb][cb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fp4_operand(rows[rb].PL), fp4_operand(col.PL), ACC[rb][cb], 4, 4, 0, scale, 0, scale);
                        TRACS_MFMA(accS, x) TRACS_MFMA(accS, y) TRACS_MFMA(accS, z) TRACS_MFMA(accV, v)
#undef TRACS_MFMA
                    }
                }
            }
        }
        // one matrix instruction, then the VALU ops (and the LDS read) of a later expansion in its shadow
#pragma unroll
        for (int k = 0; k < GC * STEPS * NB * NB * 4; k++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MS == 32 ? 8 : 4, 0);
            if (MS == 32 || (k & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __syncthreads();
        buf ^= 1;
    }

    // C/D layout: 32 x 32: register r of lane l = column l & 31, row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    //             16 x 16: column l & 15, row 4 * (l >> 4) + r
    auto cell_row = [&](int rb, int r) { return (unsigned)(i0 + wr * 64 + rb * MS + (MS == 32 ? (r & 3) + 8 * (r >> 2) + 4 * hk : 4 * hk + r)); };
    auto cell_col = [&](int cb) { return (unsigned)(j0 + wc * 64 + cb * MS + lb); };

    // Thresholded two-pass runs (TilePhase, see the tile kernel): at the end of the prefix pass a tile whose every pair is
    // already past the threshold is dead -- cells 0xFFFFFFFF, live flag 0 -- and the remainder pass never visits it.
    bool dead = false;
    if (ph.phase == 1) {
        unsigned mn = 0xFFFFFFFFu;
        if (j0 >= i0 + TI) {                                   // tiles touching the diagonal hold d(i,i) = 0 cells: always live
#pragma unroll
            for (int rb = 0; rb < NB; rb++)
#pragma unroll
                for (int cb = 0; cb < NB; cb++)
#pragma unroll
                    for (int r = 0; r < AR; r++) {
                        const unsigned d = (unsigned)((3 * (int)accV[rb][cb][r] - (int)accS[rb][cb][r]) >> 2);
                        mn = min(mn, (cell_row(rb, r) < row_end && cell_col(cb) < n) ? d : 0xFFFFFFFFu);
                    }
        } else {
            mn = 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, off, 64));
        unsigned *wmin = reinterpret_cast<unsigned *>(&lds[0][0]);      // the staging buffers are idle now (last barrier passed)
        if (lane == 0) wmin[wave] = mn;
        __syncthreads();
        unsigned m = wmin[0];
#pragma unroll
        for (int w = 1; w < NW; w++) m = min(m, wmin[w]);
        dead = m > thr;
        if (tid == 0) ph.live[tile_no] = dead ? 0 : 1;
    }
    const bool single = ksplit == 1 && ph.phase != 2;
#pragma unroll
    for (int rb = 0; rb < NB; rb++)
#pragma unroll
        for (int cb = 0; cb < NB; cb++)
#pragma unroll
            for (int r = 0; r < AR; r++) {
                const unsigned i = cell_row(rb, r), j = cell_col(cb);
                if (i < row_end && j < n && j > i && j >= col_begin) {
                    const int nn = (int)accV[rb][cb][r];
                    const int S = (int)accS[rb][cb][r];
                    const unsigned d = (unsigned)((3 * nn - S) >> 2);
                    const size_t o = (size_t)i * ld + j;
                    if (dead) {
                        dist[o] = 0xFFFFFFFFu;
                        if (WITH_NN) ncomp[o] = 0u;
                    } else if (single) {
                        dist[o] = d;
                        if (WITH_NN) ncomp[o] = (unsigned)nn;
                    } else {                                   // cells were zeroed (split range) / hold the prefix counts
                        atomicAdd(&dist[o], d);
                        if (WITH_NN) atomicAdd(&ncomp[o], (unsigned)nn);
                    }
                }
            }
}

// ---------------------------------------------------------------------------------------
// "rowcast" kernel: no LDS, no barriers.
//   * all NW waves of a workgroup work on the SAME R rows: the row words are wave-uniform AND shared by
//     the whole workgroup, so the scalar loads of waves 1..NW-1 hit the scalar cache (the 64 x 128 tile
//     kernel above gives every wave its own rows and is limited by scalar-cache MISS throughput:
//     DESIGN.md "what limits the tile kernel");
//   * every wave owns its own 64*C columns, fetched straight into VGPRs with 16 B/lane coalesced loads,
//     one group ahead (double buffered in registers);
//   * accumulators: R x C x 2 VGPRs per lane (R = 32, C = 2 -> 128), 2 waves per SIMD.
template <int NW, int R, int C, bool WITH_NN>
__global__ __launch_bounds__(NW * 64) void pairsnp_rowcast_kernel(
    const uint4 *__restrict__ P, size_t n_pad, int groups, const int2 *__restrict__ tiles, int n_tiles,
    int groups_per_split, int ksplit, unsigned L, unsigned n, unsigned row_end, unsigned col_begin,
    unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld)
{
    constexpr int NPL = WITH_NN ? NPLANES : 4;
    constexpr int RB = 4;                         // rows per scalar batch: 5 x s_load_dwordx16 = 80 SGPRs
    static_assert(R % RB == 0, "R must be a multiple of 4");
    const unsigned q = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = (int)(q / (unsigned)n_tiles);
    const int2 tile = tiles[q - (unsigned)ks * (unsigned)n_tiles];
    const int i0 = tile.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jw = tile.y + wave * (64 * C);      // first column of this wave
    // nothing to do if every column of the wave is <= the tile's first row, or outside the matrix
    if (jw + 64 * C - 1 <= i0 || (unsigned)jw >= n || (unsigned)(jw + 64 * C - 1) < col_begin) return;
    const int g_begin = ks * groups_per_split;
    const int g_end = min(groups, g_begin + groups_per_split);

    unsigned accM[R][C], accN[R][C];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < C; c++) { accM[r][c] = 0; accN[r][c] = 0; }

    const uint4 *colp = P + (size_t)jw + lane;
    auto load_cols = [&](int g, uint4 (&bj)[C][NPLANES]) {
#pragma unroll
        for (int c = 0; c < C; c++)
#pragma unroll
            for (int p = 0; p < NPL; p++) bj[c][p] = colp[((size_t)g * NPLANES + p) * n_pad + 64 * c];
    };
    auto do_group = [&](int g, const uint4 (&bj)[C][NPLANES]) {
        const uint4 *rowp = P + (size_t)g * NPLANES * n_pad + (size_t)i0;       // wave-uniform, workgroup-uniform
#pragma unroll
        for (int b = 0; b < R / RB; b++) {
            uint4 ai[RB][NPLANES];
#pragma unroll
            for (int rr = 0; rr < RB; rr++)
#pragma unroll
                for (int p = 0; p < NPL; p++) ai[rr][p] = rowp[(size_t)p * n_pad + b * RB + rr];
#pragma unroll
            for (int rr = 0; rr < RB; rr++) {
                const int r = b * RB + rr;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    pair_words<true>(ai[rr][0].x, ai[rr][1].x, ai[rr][2].x, ai[rr][3].x, bj[c][0].x, bj[c][1].x, bj[c][2].x, bj[c][3].x, accM[r][c]);
                    pair_words<true>(ai[rr][0].y, ai[rr][1].y, ai[rr][2].y, ai[rr][3].y, bj[c][0].y, bj[c][1].y, bj[c][2].y, bj[c][3].y, accM[r][c]);
                    pair_words<true>(ai[rr][0].z, ai[rr][1].z, ai[rr][2].z, ai[rr][3].z, bj[c][0].z, bj[c][1].z, bj[c][2].z, bj[c][3].z, accM[r][c]);
                    pair_words<true>(ai[rr][0].w, ai[rr][1].w, ai[rr][2].w, ai[rr][3].w, bj[c][0].w, bj[c][1].w, bj[c][2].w, bj[c][3].w, accM[r][c]);
                }
                if (WITH_NN) {
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        accN[r][c] += __popc(ai[rr][4].x | bj[c][4].x);
                        accN[r][c] += __popc(ai[rr][4].y | bj[c][4].y);
                        accN[r][c] += __popc(ai[rr][4].z | bj[c][4].z);
                        accN[r][c] += __popc(ai[rr][4].w | bj[c][4].w);
                    }
                }
            }
        }
    };

    // two groups per iteration so the register double buffer needs no copies
    uint4 bjA[C][NPLANES], bjB[C][NPLANES];
    if (g_begin < g_end) load_cols(g_begin, bjA);
    for (int g = g_begin; g < g_end; g += 2) {
        if (g + 1 < g_end) load_cols(g + 1, bjB);
        do_group(g, bjA);
        if (g + 1 < g_end) {
            if (g + 2 < g_end) load_cols(g + 2, bjA);
            do_group(g + 1, bjB);
        }
    }

#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned i = (unsigned)(i0 + r);
        if (i >= row_end) continue;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const unsigned j = (unsigned)(jw + lane + 64 * c);
            if (j < n && j > i && j >= col_begin) {
                const size_t o = (size_t)i * ld + j;
                if (ksplit == 1) {
                    dist[o] = L - accM[r][c];
                    if (WITH_NN) ncomp[o] = L - accN[r][c];
                } else {
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

// cells of the block <- L (only needed when the group range is split over workgroups)
// live tiles of a prefix pass -> compact list (order does not matter: every tile owns its cells)
__global__ void compact_live_kernel(const int2 *__restrict__ tiles, const unsigned char *__restrict__ live, unsigned n_tiles,
                                    int2 *__restrict__ out, unsigned *__restrict__ n_out)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles && live[t]) out[atomicAdd(n_out, 1u)] = tiles[t];
}

__global__ void init_cells_kernel(unsigned *__restrict__ dist, unsigned *__restrict__ ncomp, size_t ld, unsigned n,
                                  unsigned row_begin, unsigned row_end, unsigned col_begin, unsigned L)
{
    const unsigned i = row_begin + blockIdx.y;
    if (i >= row_end) return;
    for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        if (j > i && j >= col_begin) {
            dist[(size_t)i * ld + j] = L;
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

// ---------------------------------------------------------------------------------------
// host side
struct TileCfg { int ti, tj; };

// tile schedule: upper-trapezoid tiles of the block, supertile-major so that consecutive
// entries (= tiles resident together on one XCD after xcd_remap) share row/column panels.
static void build_tiles(size_t n, size_t row_begin, size_t row_end, size_t col_begin, int ti, int tj,
                        std::vector<int2> &out)
{
    out.clear();
    if (row_end <= row_begin) return;
    const size_t nbi = (row_end - row_begin + ti - 1) / ti;
    const size_t nbj = (n + tj - 1) / tj;
    const size_t SI = 4, SJ = 8;     // 32 tiles per supertile = one XCD's resident set
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

// Kernel variants (tile shape x row-operand source).  TRACS_TILE_VARIANT selects one at run time
// (default: the fastest measured on MI355X, see DESIGN.md "pairsnp kernel: variants measured").
typedef void (*TileLaunch)(bool with_nn, unsigned nwg, hipStream_t stream, const uint4 *P, size_t n_pad, int groups,
                           const int2 *tiles, int n_tiles, int gps, int ksplit, unsigned L, unsigned n, unsigned row_end,
                           unsigned col_begin, unsigned *dist, unsigned *ncomp, size_t ld, unsigned thr, TilePhase ph);
struct TileVariant {
    const char *name;
    int ti, tj, gc, nthreads;
    TileLaunch launch;                 // general IUPAC encoding (5 planes)
    TileLaunch launch_cons = nullptr;  // consensus encoding (3 planes), when the variant has one
};

template <int NW, int R, int C, int GC, int ROWSRC, int MINW = 1, bool GLDS = false, int ENC = 0>
static void launch_variant(bool with_nn, unsigned nwg, hipStream_t stream, const uint4 *P, size_t n_pad, int groups,
                           const int2 *tiles, int n_tiles, int gps, int ksplit, unsigned L, unsigned n, unsigned row_end,
                           unsigned col_begin, unsigned *dist, unsigned *ncomp, size_t ld, unsigned thr, TilePhase ph)
{
    if (with_nn)
        hipLaunchKernelGGL((pairsnp_tile_kernel<NW, R, C, GC, true, ROWSRC, MINW, GLDS, ENC>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups,
                           tiles, n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld, thr, ph);
    else
        hipLaunchKernelGGL((pairsnp_tile_kernel<NW, R, C, GC, false, ROWSRC, MINW, GLDS, ENC>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups,
                           tiles, n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld, thr, ph);
}

template <int NW, int R, int C>
static void launch_rowcast(bool with_nn, unsigned nwg, hipStream_t stream, const uint4 *P, size_t n_pad, int groups,
                           const int2 *tiles, int n_tiles, int gps, int ksplit, unsigned L, unsigned n, unsigned row_end,
                           unsigned col_begin, unsigned *dist, unsigned *ncomp, size_t ld, unsigned /*thr: no early out here*/, TilePhase)
{
    if (with_nn)
        hipLaunchKernelGGL((pairsnp_rowcast_kernel<NW, R, C, true>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups, tiles,
                           n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld);
    else
        hipLaunchKernelGGL((pairsnp_rowcast_kernel<NW, R, C, false>), dim3(nwg), dim3(NW * 64), 0, stream, P, n_pad, groups, tiles,
                           n_tiles, gps, ksplit, L, n, row_end, col_begin, dist, ncomp, ld);
}
#define TRACS_ROWCAST(NW, R, C) {"rowcast" #NW "x" #R "x" #C, R, (NW) * 64 * (C), 1, (NW) * 64, launch_rowcast<NW, R, C>}

#define TRACS_VARIANT_W(NW, R, C, GC, SRC, MINW) {#NW "x" #R "x" #C "x" #GC ":" #SRC "/w" #MINW, (NW) * (R), 64 * (C), GC, (NW) * 64, launch_variant<NW, R, C, GC, SRC, MINW>}
#define TRACS_VARIANT_G(NW, R, C, GC, SRC, MINW) {#NW "x" #R "x" #C "x" #GC ":" #SRC "/w" #MINW "/glds", (NW) * (R), 64 * (C), GC, (NW) * 64, launch_variant<NW, R, C, GC, SRC, MINW, true>}
#define TRACS_VARIANT_GC(NW, R, C, GC, SRC, MINW) {#NW "x" #R "x" #C "x" #GC ":" #SRC "/w" #MINW "/glds+cons", (NW) * (R), 64 * (C), GC, (NW) * 64, launch_variant<NW, R, C, GC, SRC, MINW, true, 0>, launch_variant<NW, R, C, GC, SRC, MINW, true, 1>}
#define TRACS_VARIANT(NW, R, C, GC, SRC) {#NW "x" #R "x" #C "x" #GC ":" #SRC, (NW) * (R), 64 * (C), GC, (NW) * 64, launch_variant<NW, R, C, GC, SRC>}
static const TileVariant kVariants[] = {
    TRACS_VARIANT(8, 8, 2, 4, ROW_SMEM),      // 0: 64 x 128 tile, rows by scalar loads
    TRACS_VARIANT(8, 8, 2, 2, ROW_SMEM),      // 1
    TRACS_VARIANT(8, 8, 2, 4, ROW_SMEM_PF),   // 2: + software-pipelined row loads
    TRACS_VARIANT(8, 8, 2, 2, ROW_SMEM_PF),   // 3
    TRACS_VARIANT(8, 8, 2, 2, ROW_LDS),       // 4: rows through LDS
    TRACS_VARIANT(8, 4, 4, 2, ROW_SMEM_PF),   // 5: 32 x 256 tile
    TRACS_VARIANT(16, 4, 4, 2, ROW_SMEM_PF),  // 6: 64 x 256 tile, 16 waves
    TRACS_VARIANT(16, 8, 2, 2, ROW_SMEM_PF),  // 7: 128 x 128 tile, 16 waves
    TRACS_VARIANT(16, 8, 2, 2, ROW_LDS),      // 8
    TRACS_VARIANT(8, 4, 4, 2, ROW_LDS),       // 9
    TRACS_VARIANT(8, 8, 2, 4, ROW_SMEM_PF1),     // 10: one-row batches
    TRACS_VARIANT(8, 8, 2, 2, ROW_SMEM_PF1),     // 11
    TRACS_VARIANT(8, 4, 4, 2, ROW_SMEM_PF1),     // 12
    TRACS_VARIANT(8, 8, 2, 4, ABL_ROWS_FIXED),   // 13 timing-only
    TRACS_VARIANT(8, 8, 2, 4, ABL_COLS_FIXED),   // 14 timing-only
    TRACS_VARIANT(8, 8, 2, 4, ABL_BOTH_FIXED),   // 15 timing-only
    TRACS_ROWCAST(8, 32, 2),                     // 16: 32 rows x 1024 cols per workgroup
    TRACS_ROWCAST(4, 32, 2),                     // 17: 32 x 512
    TRACS_ROWCAST(8, 16, 4),                     // 18: 16 x 2048
    TRACS_ROWCAST(4, 16, 4),                     // 19: 16 x 1024
    TRACS_ROWCAST(8, 16, 2),                     // 20: 16 x 1024, 64 accumulators
    TRACS_ROWCAST(16, 32, 2),                    // 21: 32 x 2048 (16 waves -> VGPR cap 128: expect spills)
    TRACS_VARIANT(8, 8, 4, 2, ROW_LDS),          // 22: 64 x 256 tile, rows through LDS
    TRACS_VARIANT(8, 8, 4, 1, ROW_LDS),          // 23
    TRACS_VARIANT(4, 8, 4, 2, ROW_LDS),          // 24: 32 x 256, 4 waves
    TRACS_VARIANT(8, 4, 8, 1, ROW_LDS),          // 25: 32 x 512
    TRACS_VARIANT(8, 16, 2, 2, ROW_LDS),         // 26: 128 x 128
    TRACS_VARIANT(4, 16, 4, 1, ROW_LDS),         // 27: 64 x 256, 4 waves, 128 accumulators
    TRACS_VARIANT_W(8, 8, 2, 2, ROW_LDS, 4),     // 28: 64 x 128, <=128 VGPRs so two workgroups share a CU
    TRACS_VARIANT_W(8, 8, 2, 1, ROW_LDS, 4),     // 29
    TRACS_VARIANT_W(4, 8, 2, 2, ROW_LDS, 4),     // 30: 32 x 128, four workgroups per CU
    TRACS_VARIANT_W(4, 16, 2, 2, ROW_LDS, 3),    // 31: 64 x 128 with 4 waves
    TRACS_VARIANT_W(8, 8, 2, 4, ROW_SMEM, 4),    // 32
    TRACS_VARIANT_G(4, 16, 2, 2, ROW_LDS, 2),    // 33: variant 31 with direct-to-LDS staging
    TRACS_VARIANT_G(4, 16, 2, 4, ROW_LDS, 2),    // 34
    TRACS_VARIANT_GC(8, 8, 2, 2, ROW_LDS, 4),    // 35: default
    TRACS_VARIANT_G(8, 8, 2, 4, ROW_LDS, 4),     // 36
    TRACS_VARIANT_GC(4, 16, 2, 1, ROW_LDS, 2),   // 37
    TRACS_VARIANT_G(8, 16, 2, 2, ROW_LDS, 2),    // 38
    TRACS_VARIANT_G(4, 16, 2, 2, ROW_LDS, 3),    // 39
    TRACS_VARIANT_G(16, 8, 2, 2, ROW_LDS, 4),    // 40
    TRACS_VARIANT_G(8, 8, 2, 1, ROW_LDS, 4),     // 41
    TRACS_VARIANT_G(8, 8, 2, 3, ROW_LDS, 4),     // 42
    TRACS_VARIANT_G(4, 16, 2, 1, ROW_LDS, 3),    // 43
    TRACS_VARIANT_G(8, 8, 2, 2, ROW_LDS, 3),     // 44
    TRACS_VARIANT_G(8, 8, 2, 2, ROW_LDS, 5),     // 45
    TRACS_VARIANT_GC(8, 8, 4, 1, ROW_LDS, 2),    // 46: 64 x 256 (consensus planes are lighter)
    TRACS_VARIANT_GC(8, 8, 2, 4, ROW_LDS, 4),    // 47
    TRACS_VARIANT_GC(8, 16, 2, 2, ROW_LDS, 2),   // 48
    TRACS_VARIANT_GC(16, 8, 2, 2, ROW_LDS, 4),   // 49
    TRACS_VARIANT_GC(8, 8, 4, 2, ROW_LDS, 3),    // 50
    TRACS_VARIANT_GC(8, 16, 2, 3, ROW_LDS, 3),   // 51
    TRACS_VARIANT_GC(4, 16, 2, 4, ROW_LDS, 3),   // 52
    TRACS_VARIANT_GC(4, 16, 4, 2, ROW_LDS, 2),   // 53
    TRACS_VARIANT_GC(8, 8, 2, 5, ROW_LDS, 4),    // 54
    TRACS_VARIANT_GC(8, 8, 2, 3, ROW_LDS, 4),    // 55
    TRACS_VARIANT_GC(4, 16, 2, 2, ROW_LDS, 3),   // 56
    TRACS_VARIANT_GC(4, 16, 3, 2, ROW_LDS, 3),   // 57: 64 x 192 tile
    TRACS_VARIANT_GC(8, 8, 3, 2, ROW_LDS, 3),    // 58
    TRACS_VARIANT_GC(4, 16, 3, 1, ROW_LDS, 3),   // 59
    TRACS_VARIANT_GC(4, 16, 3, 2, ROW_LDS, 2),   // 60
    TRACS_VARIANT_GC(4, 16, 2, 2, ROW_LDS, 4),   // 61: 56 squeezed to 128 VGPRs (4 workgroups/CU)
    TRACS_VARIANT_GC(4, 16, 2, 1, ROW_LDS, 4),   // 62
    TRACS_VARIANT_GC(4, 16, 2, 3, ROW_LDS, 4),   // 63
    TRACS_VARIANT_GC(4, 16, 1, 2, ROW_LDS, 4),   // 64: 64 x 64 tile
    TRACS_VARIANT_GC(4, 16, 1, 1, ROW_LDS, 4),   // 65
    TRACS_VARIANT_GC(4, 16, 1, 4, ROW_LDS, 4),   // 66
    TRACS_VARIANT_GC(8, 8, 1, 2, ROW_LDS, 4),    // 67
    TRACS_VARIANT_GC(8, 16, 1, 2, ROW_LDS, 2),   // 68: 128 x 64
    TRACS_VARIANT_GC(4, 32, 1, 2, ROW_LDS, 2),   // 69: 128 x 64, 4 waves
    TRACS_VARIANT_GC(4, 16, 1, 2, ROW_LDS, 6),   // 70
    TRACS_VARIANT_GC(4, 16, 1, 2, ROW_LDS, 8),   // 71
    TRACS_VARIANT_GC(8, 8, 1, 2, ROW_LDS, 8),    // 72
    TRACS_VARIANT_GC(16, 4, 1, 2, ROW_LDS, 8),   // 73
    TRACS_VARIANT_GC(2, 32, 1, 2, ROW_LDS, 4),   // 74: 64 x 64, 2 waves
    TRACS_VARIANT_GC(4, 16, 1, 3, ROW_LDS, 4),   // 75
    TRACS_VARIANT_GC(4, 16, 1, 2, ROW_LDS, 5),   // 76
};
static constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
// fastest measured on MI355X (profiles/r01/tile_variant_sweeps.txt): one default per encoding
#ifndef TRACS_DEFAULT_VARIANT
#define TRACS_DEFAULT_VARIANT 35            // general IUPAC encoding: 8 waves x 8 rows x 2 cols, GC = 2
#endif
#ifndef TRACS_DEFAULT_VARIANT_CONS
#define TRACS_DEFAULT_VARIANT_CONS 64       // consensus encoding: 4 waves x 16 rows x 1 col (64 x 64 tile), GC = 2, 94 VGPRs
#endif

static const TileVariant &current_variant(bool consensus = false)
{
    static int chosen[2] = {-1, -1};
    if (chosen[0] < 0) {
        chosen[0] = TRACS_DEFAULT_VARIANT;
        chosen[1] = TRACS_DEFAULT_VARIANT_CONS;
        if (const char *e = std::getenv("TRACS_TILE_VARIANT")) {
            const int v = std::atoi(e);
            // ids 13-15 are timing-only ablations that produce WRONG RESULTS: reachable only with TRACS_ALLOW_ABLATION=1
            const bool ablation = v >= 13 && v <= 15;
            if (v >= 0 && v < kNumVariants && (!ablation || std::getenv("TRACS_ALLOW_ABLATION"))) chosen[0] = chosen[1] = v;
        }
        if (!kVariants[chosen[1]].launch_cons) chosen[1] = chosen[0];
    }
    return kVariants[chosen[consensus ? 1 : 0]];
}

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
    }
    *out = a;
    return TRACS_OK;
}

void tracs_alignment_free(tracs_alignment *a)
{
    if (!a) return;
    if (a->planes) (void)hipFree(a->planes);
    if (a->cplanes) (void)hipFree(a->cplanes);
    if (a->d_flag) (void)hipFree(a->d_flag);
    if (a->d_tiles) (void)hipFree(a->d_tiles);
    delete a;
}

size_t tracs_alignment_n(const tracs_alignment *a) { return a ? a->n : 0; }
size_t tracs_alignment_len(const tracs_alignment *a) { return a ? a->L : 0; }
size_t tracs_alignment_bytes(const tracs_alignment *a)
{
    if (!a || !a->n || !a->L) return 0;
    // + one tile edge of slack: row tiles start at row_begin + k*TI and may read (never use) up to
    // TI-1 samples past n_pad in the last (group, plane) run
    return (a->groups * NPLANES * a->n_pad + SAMPLE_PAD) * sizeof(uint4);
}
void *tracs_alignment_planes(const tracs_alignment *a) { return a ? a->planes : nullptr; }

int tracs_alignment_pack(tracs_alignment *a, const uint8_t *ascii, size_t first, size_t count, int ascii_on_device,
                         void *stream_)
{
    if (!a || (!ascii && count)) { set_error("tracs_alignment_pack: NULL argument"); return TRACS_E_ARG; }
    if (first + count > a->n) { set_error("tracs_alignment_pack: sample range outside the alignment"); return TRACS_E_ARG; }
    if (!count || !a->L) return TRACS_OK;
    a->dirty = true;                   // the consensus form (if any) must be re-derived
    hipStream_t stream = static_cast<hipStream_t>(stream_);
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

const char *tracs_debug_tile_variant(void) { return current_variant().name; }
// 1 if the last dense call on this alignment used the consensus (3-plane) encoding, 0 general, -1 not decided yet
int tracs_debug_alignment_encoding(const tracs_alignment *a) { return !a ? -1 : (a->dirty ? -1 : a->enc); }
// kernel of the last dense call on this alignment: 0 VALU tile kernel, 1 matrix-core kernel, -1 none yet
int tracs_debug_alignment_kernel(const tracs_alignment *a) { return !a ? -1 : a->last_kernel; }

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

static int pairsnp_dense_impl(const tracs_alignment *a_, size_t row_begin, size_t row_end, size_t col_begin, uint32_t *dist,
                              uint32_t *ncomp, size_t ld, void *stream_, unsigned thr)
{
    tracs_alignment *a = const_cast<tracs_alignment *>(a_);
    if (!a || !dist) { set_error("tracs_pairsnp_dense: NULL argument"); return TRACS_E_ARG; }
    if (row_end > a->n) row_end = a->n;
    if (row_begin >= row_end || a->n < 2) return TRACS_OK;
    if (ld < a->n) { set_error("tracs_pairsnp_dense: ld < n"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    if (a->L == 0) {   // every pair: d = 0, nn = 0
        dim3 grid(64, (unsigned)(row_end - row_begin));
        hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, dist, ncomp, ld, (unsigned)a->n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, 0u);
        TRACS_HIP_CHECK(hipGetLastError());
        return TRACS_OK;
    }

    // (re)build the cached tile schedule
    // choose the encoding: consensus (3 planes, VOP2/bitop3 logic only) when no sample carries a partial IUPAC code
    if (a->dirty) {
        a->enc = 0;
        static const bool force_general = std::getenv("TRACS_FORCE_GENERAL") != nullptr;
        if (current_variant(true).launch_cons && !force_general) {
            const size_t cbytes = (a->groups * 3 * a->n_pad + SAMPLE_PAD) * sizeof(uint4);
            bool have = a->cplanes != nullptr;
            if (!have) {
                // no room for the second copy (very large alignments): stay on the general kernel
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
        }
        a->dirty = false;
    }
    const TileVariant &V = current_variant(a->enc == 1);
    const bool cons = a->enc == 1 && V.launch_cons;
    // matrix-core kernel for the consensus encoding; TRACS_MFMA=0 (or an explicit TRACS_TILE_VARIANT) keeps the VALU tile kernel
    static const bool mfma_off = [] { const char *e = std::getenv("TRACS_MFMA"); return e && e[0] == '0'; }();
    const bool mfma = cons && !mfma_off && !std::getenv("TRACS_TILE_VARIANT");
    // groups per LDS stage of the matrix-core kernel: 1 measured best (29.3 ms per 400 kbp; 2: 30.2, 3: 29.8)
    static const int mfma_gc = [] { const char *e = std::getenv("TRACS_MFMA_GC"); const int v = e ? std::atoi(e) : 1; return v == 2 || v == 3 ? v : 1; }();
    const int kTI = mfma ? 128 : V.ti, kTJ = mfma ? 128 : V.tj, kGC = mfma ? mfma_gc : V.gc;
    a->last_kernel = mfma ? 1 : 0;
    if (a->key_rb != row_begin || a->key_re != row_end || a->key_cb != col_begin || a->key_ti != kTI || a->key_tj != kTJ) {
        std::vector<int2> tiles;
        build_tiles(a->n, row_begin, row_end, col_begin, kTI, kTJ, tiles);
        if (tiles.size() > a->tiles_cap) {
            if (a->d_tiles) TRACS_HIP_CHECK(hipFree(a->d_tiles));
            a->d_tiles = nullptr;
            a->tiles_cap = tiles.size() * 2 + 64;
            TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&a->d_tiles), a->tiles_cap * sizeof(int2)));
        }
        if (!tiles.empty()) {
            TRACS_HIP_CHECK(hipMemcpyAsync(a->d_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice, stream));
            TRACS_HIP_CHECK(hipStreamSynchronize(stream));   // `tiles` is a stack vector
        }
        a->n_tiles = tiles.size();
        a->key_rb = row_begin; a->key_re = row_end; a->key_cb = col_begin; a->key_ti = kTI; a->key_tj = kTJ;
    }
    if (a->n_tiles == 0) return TRACS_OK;

    // Split the group range over workgroups (integer atomics, still exact) when that fills the chip better:
    // too few tiles (config 2), or a ragged last round of resident workgroups (tail effect).  Workgroup slots =
    // CUs x workgroups per CU for this variant (VGPR/LDS bound: 2 for every default shape).
    const int groups = (int)a->groups;
    int ksplit = 1;
    double slots = 512.0;
    {
        static int cus = 0;
        if (!cus) { hipDeviceProp_t pr; int dv = 0; (void)hipGetDevice(&dv); cus = (hipGetDeviceProperties(&pr, dv) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
        // resident workgroups per CU (VGPR/LDS bound) of the shapes that are defaults; 2 is right for every other general shape
        const int vid = (int)(&V - kVariants);
        slots = (mfma ? 2.0 : cons ? (vid == 64 ? 5.0 : vid == 56 ? 3.0 : 2.0) : 2.0) * cus;
        const int max_split = std::max(1, groups / (8 * kGC));
        double best = -1.0;
        for (int k = 1; k <= std::min(max_split, 64); k++) {
            const double wgs = (double)a->n_tiles * k;
            const double rounds = std::ceil(wgs / slots);
            const double eff = wgs / (rounds * slots) - 0.002 * (k - 1);     // small bias towards fewer splits
            if (eff > best + 1e-9) { best = eff; ksplit = k; }
            if (rounds >= 40) break;                                         // tail < 2.5 % from here on
        }
        if (const char *e = std::getenv("TRACS_KSPLIT")) { const int v = std::atoi(e); if (v >= 1 && v <= max_split) ksplit = v; }
    }
    auto stage_split = [&](int range, int k, int &gps_out) {       // k workgroups over `range` groups, stage aligned
        int g = (range + k - 1) / k;
        g = (g + kGC - 1) / kGC * kGC;
        gps_out = g;
        return (range + g - 1) / g;
    };
    // fp32 accumulators of the matrix-core kernel are exact while a workgroup's range stays below 2^22 sites (|S| <= 3 * 2^22 < 2^24)
    const int max_gps = mfma ? (1 << 22) / SITES_PER_GROUP : groups;
    auto launch = [&](const int2 *tl, unsigned nwg, int ntl, int g_end, int gps, int k, unsigned t, TilePhase ph) {
        if (mfma) {
            int abl = 0;
            if (const char *ab = std::getenv("TRACS_MFMA_ABL")) abl = std::getenv("TRACS_ALLOW_ABLATION") ? std::atoi(ab) : 0;   // timing only, WRONG RESULTS
            static const int shape = [] { const char *e = std::getenv("TRACS_MFMA_SHAPE"); return e && std::atoi(e) == 16 ? 16 : 32; }();   // 32 measured 3.5 % faster
#define TRACS_MFMA_LAUNCH(G, NN, A, MS) hipLaunchKernelGGL((pairsnp_mfma_kernel<G, NN, A, MS>), dim3(nwg), dim3(256), 0, stream, a->cplanes, a->n_pad, g_end, tl, ntl, gps, k, \
                                                           (unsigned)a->n, (unsigned)row_end, (unsigned)col_begin, dist, ncomp, ld, 127, t, ph)
#define TRACS_MFMA_BY_NN(G, MS) do { if (ncomp) TRACS_MFMA_LAUNCH(G, true, 0, MS); else TRACS_MFMA_LAUNCH(G, false, 0, MS); } while (0)
            if (abl == 1) TRACS_MFMA_LAUNCH(1, true, 1, 32);
            else if (abl == 2) TRACS_MFMA_LAUNCH(1, true, 2, 32);
            else if (shape == 16) TRACS_MFMA_BY_NN(1, 16);
            else if (kGC == 2) TRACS_MFMA_BY_NN(2, 32);
            else if (kGC == 3) TRACS_MFMA_BY_NN(3, 32);
            else TRACS_MFMA_BY_NN(1, 32);
#undef TRACS_MFMA_BY_NN
#undef TRACS_MFMA_LAUNCH
            return;
        }
        (cons ? V.launch_cons : V.launch)(ncomp != nullptr, nwg, stream, cons ? a->cplanes : a->planes, a->n_pad, g_end, tl,
                                          ntl, gps, k, (unsigned)a->L, (unsigned)a->n, (unsigned)row_end,
                                          (unsigned)col_begin, dist, ncomp, ld, t, ph);
    };

    // Thresholded run on a long alignment: two passes.  The prefix pass (1/8 of the groups, one workgroup per tile) leaves
    // exact partial counts and a live flag per tile; the remainder pass visits the live tiles only, its range split so that the
    // few surviving tiles still fill the chip.  Dead tiles cost 1/8 of a full pass or less (they also stop inside the prefix).
    static const bool no_two_pass = std::getenv("TRACS_THR_ONE_PASS") != nullptr;
    if (thr != 0xFFFFFFFFu && V.launch != nullptr && V.gc > 1 && !no_two_pass && groups >= 64 * kGC) {
        int prefix = std::min(max_gps / kGC * kGC, std::max(8 * kGC, groups / 8 / kGC * kGC));
        if (const char *e = std::getenv("TRACS_THR_PREFIX")) { const int v = std::atoi(e) / kGC * kGC; if (v >= kGC && v < groups) prefix = v; }
        unsigned char *live = nullptr;
        int2 *live_tiles = nullptr;
        unsigned *n_live_d = nullptr;
        int rc;
        if ((rc = tracs::workspace_get(48, a->n_tiles, reinterpret_cast<void **>(&live)))) return rc;
        if ((rc = tracs::workspace_get(49, a->n_tiles * sizeof(int2), reinterpret_cast<void **>(&live_tiles)))) return rc;
        if ((rc = tracs::workspace_get(50, 64, reinterpret_cast<void **>(&n_live_d)))) return rc;
        TRACS_HIP_CHECK(hipMemsetAsync(n_live_d, 0, 4, stream));
        launch(a->d_tiles, (unsigned)a->n_tiles, (int)a->n_tiles, prefix, prefix, 1, thr, TilePhase{1, 0, live});
        hipLaunchKernelGGL(compact_live_kernel, dim3((unsigned)((a->n_tiles + 255) / 256)), dim3(256), 0, stream, a->d_tiles, live,
                           (unsigned)a->n_tiles, live_tiles, n_live_d);
        unsigned n_live = 0;
        TRACS_HIP_CHECK(hipMemcpyAsync(&n_live, n_live_d, 4, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        if (n_live) {
            // enough workgroups for ~4 rounds of the chip, never more than 32 ranges (each range costs one atomic per cell)
            int gps2 = 0;
            const int want = (int)std::ceil(4.0 * slots / (double)n_live);
            const int k2 = stage_split(groups - prefix, std::max({1, std::min({32, want, (groups - prefix) / (16 * kGC)}), (groups - prefix + max_gps - 1) / max_gps}), gps2);
            launch(live_tiles, (unsigned)(n_live * (size_t)k2), (int)n_live, groups, gps2, k2, thr, TilePhase{2, prefix, nullptr});
        }
        TRACS_HIP_CHECK(hipGetLastError());
        return TRACS_OK;
    }

    if (mfma) ksplit = std::max(ksplit, (groups + max_gps - 1) / max_gps);
    int gps = 0;
    ksplit = stage_split(groups, ksplit, gps);
    if (ksplit > 1) {
        dim3 grid(64, (unsigned)(row_end - row_begin));
        hipLaunchKernelGGL(init_cells_kernel, grid, dim3(256), 0, stream, dist, ncomp, ld, (unsigned)a->n,
                           (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, cons ? 0u : (unsigned)a->L);
    }
    launch(a->d_tiles, (unsigned)(a->n_tiles * (size_t)ksplit), (int)a->n_tiles, groups, gps, ksplit, thr, TilePhase{0, 0, nullptr});
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_alignment_pack_codes(tracs_alignment *a, const uint8_t *codes, size_t sample, void *stream_)
{
    if (!a || !codes) { set_error("tracs_alignment_pack_codes: NULL argument"); return TRACS_E_ARG; }
    if (sample >= a->n) { set_error("tracs_alignment_pack_codes: sample index outside the alignment"); return TRACS_E_ARG; }
    if (!a->L) return TRACS_OK;
    a->dirty = true;
    hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)((a->groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       codes, a->L, sample, a->planes, a->n_pad, a->groups);
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

}  // extern "C"
