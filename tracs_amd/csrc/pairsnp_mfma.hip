// pairsnp_mfma.hip -- the pair loop on the matrix cores of gfx950 (CDNA4).
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp
//   pair loop :395-420   match = (Ai&Aj)|(Ci&Cj)|(Gi&Gj)|(Ti&Tj), d = L - popcount(match), nn = L - popcount(Ni|Nj)
//
// The pair loop is integer-VALU-bound once it is tiled, and it is a Gram matrix.  Two exact forms, both on
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands whose products are small integers, so that the fp32
// accumulators are exact (the host limits a workgroup's range so that every partial sum stays below 2^24):
//
//   CONSENSUS encoding (every site of every sample is A/C/G/T or fully ambiguous; planes X, Y, V):
//     every base as three signs  x = (-1)^X, y = (-1)^Y, z = x*y  (all three 0 where the site is not a base): two samples
//     contribute  x x' + y y' + z z' = +3  where they agree and  -1  where they differ, so over a site range
//         S = 4 * matches - nn,   nn = sum v v',   d = nn - matches = (3 nn - S) / 4.
//
//   GENERAL encoding (any IUPAC code; planes A, C, G, T, N as load_seqs builds them):
//     one-hot Gram  G = sum_s |S_i n S_j|  (four planes, values 0/1)  and  NN = sum_s n_i n_j  (n = A & C & G & T: only the
//     four allele planes are staged, n is formed in registers).  For every pair of codes of which at most one is a partial code,  [S n S' != {}] = |S n S'| - 3 [both N] - (|M| - 1) [one partial, one N],
//     so   d = L - G + 3 NN + T1 + T2,   nn = L - c_i - c_j + NN     (c_i = number of N sites of sample i)
//     where T1 (partial x N) and T2 (partial x partial sharing >= 2 alleles) are sums over the few sites at which a sample
//     carries a partial code: general_sparse.hip adds them (and the c_i, c_j terms) afterwards.
//
// Operands are expanded in registers from the bit planes, never stored (an fp4 image in HBM would be 16-20 bits per site
// instead of 3-5).  The K index of the MFMA is ours to choose as long as both operands agree:
//   consensus: dword q of a 32-site word takes the sites whose bit index is = q (mod 4): "mask, shift, or", ~29 VALU ops
//              per (sample, 32 sites) for the four operand planes;
//   general:   one instruction takes ONE residue class q of FOUR 32-site words (two groups per stage), so that the bit
//              already sits at a magnitude position of its nibble: 0001 = 0.5, 0010 = 1.0, 0100 = 2.0 (q = 3 is shifted
//              onto 0100); the per-instruction block scale (2^+2, 2^0, 2^-2) makes every product exactly 1:
//              5 VALU ops per (sample, plane, 32 sites).
//
//   COUNT form (site classes, site_classes.hip): one stored plane n = "this sample is N here" over the sites that only need
//     their compared-sites count: NN = sum n n' with the general form's residue-class operands, nn += sites - c_i - c_j + NN
//     (the complement plane "is a base here" would do without the c terms, but the matrix pipe holds a higher clock on
//     mostly-zero operands); one accumulator set, so four waves fit a SIMD; its own workgroup tiles (CountShape).
//
// Structure: a workgroup = NWR x NWC waves (2 x 2 for the consensus form, 4 x 2 for the general one, which is the
// memory-hungrier); each wave owns NBR x NBC blocks of 32 x 32 pairs (two fp32 accumulator sets).  The wave tile sets the
// VALU : MFMA ratio -- (NBR + NBC) expansions feed 4 NBR NBC (5 NBR NBC) instructions -- but larger wave tiles mean one wave
// per SIMD and lose (measured).  The bit planes of the tile's samples are staged HBM -> LDS directly
// (global_load_lds_dwordx4), double-buffered, the staging instructions spread over the units of the software pipeline.
#include "pairsnp_kernels.h"

#include <cstdlib>
#include <cstring>

namespace tracs {

typedef int mfma_v8i __attribute__((ext_vector_type(8)));
typedef float mfma_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ mfma_v8i fp4_operand(const unsigned (&w)[4])
{
    mfma_v8i r = {(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
    return r;
}

// fp4 magnitudes used by the general form: 0001 = 0.5, 0010 = 1.0, 0100 = 2.0
#define TRACS_MFMA_FP4(ACC, A_, B_, SC) \
    ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A_, B_, ACC, 4, 4, 0, SC, 0, SC)

// COUNT: the one-operand pass over the counted sites of an alignment cut into site classes (site_classes.hip): a single
// stored plane n ("this sample is N here"), NN = sum n n' with the general form's residue-class operands, nothing else.
template <bool GENERAL, int NBR, int NBC, int GC, int NWR = 2, int NWC = 2, bool COUNT = false>
__global__ __launch_bounds__(NWR * NWC * 64, (NBR * NBC > 4 ? 1 : COUNT ? 4 : 2)) void pairsnp_mfma_kernel(const MfmaArgs A)
{
    // NP planes are staged per group, GP is the group's stride in the stored planes: the general form stages A, C, G, T only and
    // forms N = A & C & G & T in registers (3 VALU ops per word against a fifth of the staging traffic)
    constexpr int NP = COUNT ? 1 : GENERAL ? 4 : 3, GP = COUNT ? 1 : GENERAL ? NPLANES : 3, NW = NWR * NWC;
    static_assert(!(COUNT && GENERAL), "COUNT is its own form");
    constexpr int WI = NBR * 32, WJ = NBC * 32;             // wave tile
    constexpr int TI = NWR * WI, TJ = NWC * WJ, TS = TI + TJ;   // workgroup tile, samples staged per (group, plane)
    constexpr int STAGE = GC * NP * TS;                     // uint4 per LDS stage
    static_assert(TS % 64 == 0, "a staging wave-instruction must stay inside one (group, plane) run");
    static_assert(STAGE % 64 == 0, "stage must be whole wave-instructions");
    static_assert(!GENERAL || GC == 2, "the general form takes one residue class of four words: two groups per stage");
    static_assert(!COUNT || GC % 2 == 0, "the counting form takes one residue class of four words: pairs of groups");
    static_assert(GC - 1 <= PAD_GROUPS, "stages may overhang the alignment by GC - 1 zero groups");
    __shared__ uint4 lds[2][STAGE];

    const unsigned q = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = (int)(q / (unsigned)A.n_tiles);
    const unsigned tile_no = q - (unsigned)ks * (unsigned)A.n_tiles;
    const int2 tile = A.tiles[tile_no];
    const int i0 = tile.x, j0 = tile.y;
    const int tid = threadIdx.x, lane = tid & 63, lb = lane & 31, hk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;
    const int g_begin = A.ph.g_base + ks * A.gps;
    const int g_end = min(A.groups, g_begin + A.gps);
    if (g_begin >= g_end) return;
    const uint4 *__restrict__ P = A.P;
    const size_t n_pad = A.n_pad;
    // the counting form's plane is either its own array (one plane per group) or the stored N plane of the alignment, in place
    // (five planes per group): a run-time stride, the runs themselves are the same
    const int GPr = COUNT ? A.count_gp : GP;

    // One staging wave-instruction = 64 consecutive samples of one (group, plane) run: a wave-uniform base (kept in SGPRs)
    // plus lane * 16 bytes.  Ranges are whole stages except the last one of the alignment, which may run GC - 1 groups past
    // `groups`: the planes carry that many all-zero groups behind the last real one (PAD_GROUPS), zero words expand to zero
    // operands, and the loop needs neither a range branch nor a second source pointer.
    // wave w stages the 64-sample chunks w, w + 4, .. of every (group, plane) run of the stage
    constexpr int CH = TS / 64;                             // chunks per run
    constexpr int CPW = (CH + NW - 1) / NW;                 // chunks per wave and run
    size_t chunk_smp[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; cc++) {
        const int s0 = (wave + cc * NW) * 64;
        chunk_smp[cc] = s0 < TJ ? (size_t)j0 + s0 : (size_t)i0 + (s0 - TJ);
    }
    auto stage_glds = [&](int gs, int b) {
#pragma unroll
        for (int r = 0; r < GC * NP; r++) {
            const uint4 *run = P + ((size_t)(gs + r / NP) * GPr + (r % NP)) * n_pad + lane;
#pragma unroll
            for (int cc = 0; cc < CPW; cc++) {
                const int c = wave + cc * NW;
                if (CH % NW == 0 || c < CH)                 // wave-uniform
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(run + chunk_smp[cc]), (lds_void_t *)&lds[b][r * TS + c * 64], 16, 0, 0);
            }
        }
    };
    // The same loads one at a time, dealt round-robin to the waves (piece k of this wave = wave-instruction wave + 4 k of
    // the stage): inside the loop one piece goes out per unit, so that its issue cost (tens of cycles for an LDS-DMA
    // instruction) lands in the shadow of matrix instructions already queued instead of in a burst at the top of the stage.
    constexpr int NLOAD = GC * NP * CH, LPW = (NLOAD + NW - 1) / NW;
    auto stage_piece = [&](int gs, int b, int k) {
        // a wave without a piece of its own repeats the stage's last one (same bytes to the same place): no branch in the unit
        const int t = min(wave + k * NW, NLOAD - 1);
        const int r = t / CH, c = t - r * CH;
        const int s0 = c * 64;
        // 32-bit element offset inside the stage (n_pad < 2^28 samples): one SGPR per piece instead of an address pair
        const unsigned off = (unsigned)((r / NP) * GPr + (r % NP)) * (unsigned)n_pad + (unsigned)(s0 < TJ ? j0 + s0 : i0 + (s0 - TJ));
        __builtin_amdgcn_global_load_lds((glb_void_t *)(P + (size_t)gs * GPr * n_pad + off + lane),
                                         (lds_void_t *)&lds[b][r * TS + c * 64], 16, 0, 0);
    };
    // this lane's sample inside a staged (group, plane) run, per row block / column block of the wave's tile; a lane takes
    // the two 32-site words 2 hk, 2 hk + 1 of the group (one ds_read_b64, conflict-free at the 16-byte sample stride)
    const int row_slot = TJ + wr * WI + lb, col_slot = wc * WJ + lb;
    auto rd2 = [&](int buf, int gl, int p, int slot) -> uint2 {
        return reinterpret_cast<const uint2 *>(&lds[buf][(gl * NP + p) * TS + slot])[hk];
    };

    // nibble masks held in VGPRs: an op whose sources are all VGPRs (v_and_b32, v_bitop3_b32) issues in the fast class, one with
    // a literal/SGPR source or v_and_or_b32 does not (profiles/r01/valu_ops_microbench.txt)
    unsigned M1 = 0x11111111u, M2 = 0x22222222u, M4 = 0x44444444u, M8 = 0x88888888u;
    asm volatile("" : "+v"(M1), "+v"(M2), "+v"(M4), "+v"(M8));

    mfma_v16f accS[NBR][NBC], accV[NBR][NBC];
#pragma unroll
    for (int a = 0; a < NBR; a++)
#pragma unroll
        for (int b = 0; b < NBC; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) { accS[a][b][r] = 0.0f; accV[a][b][r] = 0.0f; }

    // The main loop is a software pipeline of UNITS: one unit = the NBR x NBC matrix instructions of ONE operand plane (and,
    // in the general form, one residue class), issued while the VALU builds the NEXT unit's operands of all NB = NBR + NBC
    // blocks; a scheduling fence after every unit keeps the compiler from hoisting later expansions (and their registers)
    // across it.  Live operand registers: two sets of NB x 4 dwords + the raw words, whatever the tile.
    constexpr int NB = NBR + NBC;
    auto slot_of = [&](int b) { return b < NBR ? row_slot + b * 32 : col_slot + (b - NBR) * 32; };
#define TRACS_UNIT_MFMAS(ACC, OPS, SC)                                                                      \
    _Pragma("unroll") for (int cb = 0; cb < NBC; cb++)                                                      \
        _Pragma("unroll") for (int rb = 0; rb < NBR; rb++)                                                  \
            TRACS_MFMA_FP4(ACC[rb][cb], fp4_operand(OPS[rb]), fp4_operand(OPS[NBR + cb]), SC);
#define TRACS_UNIT_STAGE(UNITS)                                                                             \
    _Pragma("unroll") for (int k = 0; k < (LPW + (UNITS) - 1) / (UNITS); k++)                                \
        if (piece < LPW) stage_piece(gn, buf ^ 1, piece++);
#define TRACS_UNIT_SCHED(NVALU, NDS)                                                                        \
    _Pragma("unroll") for (int k = 0; k < NBR * NBC; k++) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
        if (k == 0) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, ((NVALU) + NBR * NBC - 1) / (NBR * NBC), 0);            \
        if (k < (NDS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);

    stage_glds(g_begin, 0);
    __syncthreads();
    int buf = 0;
    for (int gs = g_begin; gs < g_end; gs += GC) {
        // next stage to fetch; on the last stage the current one is fetched again (valid memory, never read): no branch in the body
        const int gn = gs + GC < g_end ? gs + GC : gs;
        int piece = 0;
        if constexpr (COUNT) {
            // units: residue class q of the four words of a pair of groups, all into accV
            unsigned raw[2][NB][4], op[2][NB][4];
            auto load_raw = [&](int gp) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const uint2 a = rd2(buf, 2 * gp, 0, slot_of(b)), c = rd2(buf, 2 * gp + 1, 0, slot_of(b));
                    raw[gp & 1][b][0] = a.x; raw[gp & 1][b][1] = a.y; raw[gp & 1][b][2] = c.x; raw[gp & 1][b][3] = c.y;
                }
            };
#define TRACS_MAKE_RES(SRC_, Q_, O_)                                                                        \
            _Pragma("unroll") for (int b = 0; b < NB; b++)                                                  \
                _Pragma("unroll") for (int w = 0; w < 4; w++) {                                                 \
                    const unsigned rw_ = SRC_[b][w];                                                        \
                    O_[b][w] = (Q_) == 0 ? (rw_ & M1) : (Q_) == 1 ? (rw_ & M2) : (Q_) == 2 ? (rw_ & M4) : ((rw_ >> 1) & M4); \
                }
            load_raw(0);
            TRACS_MAKE_RES(raw[0], 0, op[0])
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gp = 0; gp < GC / 2; gp++) {
                TRACS_UNIT_MFMAS(accV, op[0], 128)
                TRACS_MAKE_RES(raw[gp & 1], 1, op[1])
                if (gp + 1 < GC / 2) load_raw(gp + 1);
                TRACS_UNIT_STAGE(GC * 2)
                TRACS_UNIT_SCHED(NB * 4, NB * 2)
                TRACS_UNIT_MFMAS(accV, op[1], 127)
                TRACS_MAKE_RES(raw[gp & 1], 2, op[0])
                TRACS_UNIT_STAGE(GC * 2)
                TRACS_UNIT_SCHED(NB * 4, 0)
                TRACS_UNIT_MFMAS(accV, op[0], 126)
                TRACS_MAKE_RES(raw[gp & 1], 3, op[1])
                TRACS_UNIT_STAGE(GC * 2)
                TRACS_UNIT_SCHED(NB * 8, 0)
                TRACS_UNIT_MFMAS(accV, op[1], 126)
                if (gp + 1 < GC / 2) { TRACS_MAKE_RES(raw[(gp + 1) & 1], 0, op[0]) }
                TRACS_UNIT_STAGE(GC * 2)
                TRACS_UNIT_SCHED(NB * 4, 0)
            }
#undef TRACS_MAKE_RES
        } else if constexpr (!GENERAL) {
            // units per 32-site step: v, x, y, z.  vq = the v operand, also the magnitude bits of the three sign operands.
            uint2 rawV[NB], rawXY[NB][2];               // this lane's two words (st = 0, 1) of the current group, per block
            unsigned vq[NB][4], op[2][NB][4];
            auto load_v = [&](int gl) {
#pragma unroll
                for (int b = 0; b < NB; b++) rawV[b] = rd2(buf, gl, 2, slot_of(b));
            };
            auto load_xy = [&](int gl) {
#pragma unroll
                for (int b = 0; b < NB; b++) { rawXY[b][0] = rd2(buf, gl, 0, slot_of(b)); rawXY[b][1] = rd2(buf, gl, 1, slot_of(b)); }
            };
            auto make_v = [&](int st) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const unsigned V = st ? rawV[b].y : rawV[b].x;
                    vq[b][0] = (V << 1) & M2; vq[b][1] = V & M2;
                    vq[b][2] = (V >> 1) & M2; vq[b][3] = (V >> 2) & M2;
                }
            };
            auto make_sign = [&](int st, int which, unsigned (&o)[NB][4]) {      // which: 0 x, 1 y, 2 z = x*y
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const unsigned X = st ? rawXY[b][0].y : rawXY[b][0].x, Y = st ? rawXY[b][1].y : rawXY[b][1].x;
                    const unsigned T = which == 0 ? X : which == 1 ? Y : (X ^ Y);
                    o[b][0] = __builtin_amdgcn_bitop3_b32(T << 3, M8, vq[b][0], 0xEA);     // (a & b) | c
                    o[b][1] = __builtin_amdgcn_bitop3_b32(T << 2, M8, vq[b][1], 0xEA);
                    o[b][2] = __builtin_amdgcn_bitop3_b32(T << 1, M8, vq[b][2], 0xEA);
                    o[b][3] = __builtin_amdgcn_bitop3_b32(T, M8, vq[b][3], 0xEA);
                }
            };
            load_v(0);
            load_xy(0);
            make_v(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gl = 0; gl < GC; gl++) {
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    const bool more = !(gl == GC - 1 && st == 1);
                    const bool next_group = st == 1 && gl + 1 < GC;
                    // unit v: nn += v v'   | build x
                    TRACS_UNIT_MFMAS(accV, vq, 127)
                    make_sign(st, 0, op[0]);
                    TRACS_UNIT_STAGE(GC * 8)
                    TRACS_UNIT_SCHED(NB * 7, 0)
                    // unit x | build y
                    TRACS_UNIT_MFMAS(accS, op[0], 127)
                    make_sign(st, 1, op[1]);
                    TRACS_UNIT_STAGE(GC * 8)
                    TRACS_UNIT_SCHED(NB * 7, 0)
                    // unit y | build z; the V words of the next group are requested here (this group's are done with)
                    TRACS_UNIT_MFMAS(accS, op[1], 127)
                    make_sign(st, 2, op[0]);
                    if (next_group) load_v(gl + 1);
                    TRACS_UNIT_STAGE(GC * 8)
                    TRACS_UNIT_SCHED(NB * 8, NB)
                    // unit z | build the next step's v; the X, Y words of the next group are requested
                    TRACS_UNIT_MFMAS(accS, op[0], 127)
                    if (more) make_v(st ^ 1);
                    if (next_group) load_xy(gl + 1);
                    TRACS_UNIT_STAGE(GC * 8)
                    TRACS_UNIT_SCHED(NB * 7, NB * 2)
                }
            }
        } else {
            // units: plane p (A, C, G, T -> G; N -> NN) x residue class q of the four words of the stage's two groups
            unsigned raw[2][NB][4], nraw[NB][4], op[2][NB][4];
            auto load_raw = [&](int p) {
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const uint2 a = rd2(buf, 0, p, slot_of(b)), c = rd2(buf, 1, p, slot_of(b));
                    raw[p & 1][b][0] = a.x; raw[p & 1][b][1] = a.y; raw[p & 1][b][2] = c.x; raw[p & 1][b][3] = c.y;
                }
            };
#define TRACS_MAKE_RES(SRC_, Q_, O_)                                                                        \
            _Pragma("unroll") for (int b = 0; b < NB; b++)                                                  \
                _Pragma("unroll") for (int w = 0; w < 4; w++) {                                                 \
                    const unsigned rw_ = SRC_[b][w];                                                        \
                    O_[b][w] = (Q_) == 0 ? (rw_ & M1) : (Q_) == 1 ? (rw_ & M2) : (Q_) == 2 ? (rw_ & M4) : ((rw_ >> 1) & M4); \
                }
            // N plane = A & C & G & T, folded in as each allele plane's words arrive
#define TRACS_FOLD_N(P_)                                                                                    \
            _Pragma("unroll") for (int b = 0; b < NB; b++)                                                  \
                _Pragma("unroll") for (int w = 0; w < 4; w++) nraw[b][w] = (P_) == 0 ? raw[0][b][w] : (nraw[b][w] & raw[(P_) & 1][b][w]);
            load_raw(0);
            TRACS_MAKE_RES(raw[0], 0, op[0])
            TRACS_FOLD_N(0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < NPLANES; p++) {
                // residue 0 (0.5 * 0.5 * 2^2) | build residue 1; request the next allele plane's words
                if (p < 4) { TRACS_UNIT_MFMAS(accS, op[0], 128) TRACS_MAKE_RES(raw[p & 1], 1, op[1]) }
                else { TRACS_UNIT_MFMAS(accV, op[0], 128) TRACS_MAKE_RES(nraw, 1, op[1]) }
                if (p + 1 < 4) load_raw(p + 1);
                TRACS_UNIT_STAGE(NPLANES * 4)
                TRACS_UNIT_SCHED(NB * 4, NB * 2)
                // residue 1 (1.0 * 1.0) | build residue 2
                if (p < 4) { TRACS_UNIT_MFMAS(accS, op[1], 127) TRACS_MAKE_RES(raw[p & 1], 2, op[0]) }
                else { TRACS_UNIT_MFMAS(accV, op[1], 127) TRACS_MAKE_RES(nraw, 2, op[0]) }
                TRACS_UNIT_STAGE(NPLANES * 4)
                TRACS_UNIT_SCHED(NB * 4, 0)
                // residue 2 (2.0 * 2.0 * 2^-2) | build residue 3
                if (p < 4) { TRACS_UNIT_MFMAS(accS, op[0], 126) TRACS_MAKE_RES(raw[p & 1], 3, op[1]) }
                else { TRACS_UNIT_MFMAS(accV, op[0], 126) TRACS_MAKE_RES(nraw, 3, op[1]) }
                TRACS_UNIT_STAGE(NPLANES * 4)
                TRACS_UNIT_SCHED(NB * 8, 0)
                // residue 3 | build the next plane's residue 0 (and fold the next allele plane into N)
                if (p < 4) { TRACS_UNIT_MFMAS(accS, op[1], 126) } else { TRACS_UNIT_MFMAS(accV, op[1], 126) }
                if (p + 1 < 4) { TRACS_MAKE_RES(raw[(p + 1) & 1], 0, op[0]) TRACS_FOLD_N(p + 1) }
                else if (p + 1 == 4) { TRACS_MAKE_RES(nraw, 0, op[0]) }
                TRACS_UNIT_STAGE(NPLANES * 4)
                TRACS_UNIT_SCHED(NB * 8, 0)
            }
#undef TRACS_FOLD_N
#undef TRACS_MAKE_RES
        }
        __syncthreads();
        buf ^= 1;
    }
#undef TRACS_UNIT_MFMAS
#undef TRACS_UNIT_SCHED
#undef TRACS_UNIT_STAGE

    // C/D layout of the 32 x 32 instruction: register r of lane l = column l & 31, row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
    auto cell_row = [&](int rb, int r) { return (unsigned)(i0 + wr * WI + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hk); };
    auto cell_col = [&](int cb) { return (unsigned)(j0 + wc * WJ + cb * 32 + lb); };

    // sites of this workgroup's range (the last group is clipped to L): the general form's  L_range - G + 3 NN
    const unsigned Lc = min(A.L, (unsigned)g_end * SITES_PER_GROUP) - (unsigned)g_begin * SITES_PER_GROUP;
    // Thresholded two-pass runs (TilePhase): at the end of the prefix pass a tile whose every pair is already past the
    // threshold is dead -- live flag 0, the remainder pass never visits it -- and its cells are flagged.  In the general form the
    // kernel's value is a LOWER bound of the range's distance (the sparse terms T1, T2 >= 0 are added later), which is the safe
    // side for "already past the threshold"; it can be negative before the correction, hence the signed compare.
    bool dead = false;
    if (A.ph.phase == 1) {
        int mn = 0x7FFFFFFF;
        if (j0 >= i0 + TI) {                                   // tiles touching the diagonal hold d(i,i) = 0 cells: always live
#pragma unroll
            for (int rb = 0; rb < NBR; rb++)
#pragma unroll
                for (int cb = 0; cb < NBC; cb++)
                {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int V = (int)accV[rb][cb][r], S = (int)accS[rb][cb][r];
                        const int d = GENERAL ? (int)Lc - S + 3 * V : ((3 * V - S) >> 2);
                        mn = min(mn, (cell_row(rb, r) < A.row_end && cell_col(cb) < A.n) ? d : 0x7FFFFFFF);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // one block at a time: the accumulators stay where they are
                }
        } else {
            mn = -0x7FFFFFFF;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, __shfl_xor(mn, off, 64));
        int *wmin = reinterpret_cast<int *>(&lds[0][0]);      // the staging buffers are idle now (last barrier passed)
        if (lane == 0) wmin[wave] = mn;
        __syncthreads();
        int m = wmin[0];
#pragma unroll
        for (int w = 1; w < NW; w++) m = min(m, wmin[w]);
        dead = (long long)m > (long long)A.thr;
        if (tid == 0) A.ph.live[tile_no] = dead ? 0 : 1;
    }
    const bool single = A.ksplit == 1 && A.ph.phase != 2;
#pragma unroll
    for (int rb = 0; rb < NBR; rb++)
#pragma unroll
        for (int cb = 0; cb < NBC; cb++) {
            __builtin_amdgcn_sched_barrier(0);                 // one block at a time (register pressure, see above)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned i = cell_row(rb, r), j = cell_col(cb);
                if (i < A.row_end && j < A.n && j > i && j >= A.col_begin) {
                    const int V = (int)accV[rb][cb][r];
                    if constexpr (COUNT) {
                        // operand plane n = "is N here" (mostly zero words): NN = sum n n', and nn = sites - c_i - c_j + NN
                        // (A.L: the sites this pass stands for, added once per cell by range 0).  The cells hold the dense sites'
                        // counts already (or zeros: in-place source, split range); a single range over the in-place source stores.
                        // count_mode (nw_gram): the same sums are terms of the distances -- + n n^T, - U U^T (site_classes.hip)
                        if (A.count_mode == 2) { atomicAdd(&A.dist[(size_t)i * A.ld + j], 0u - (unsigned)V); continue; }
                        if (A.count_mode == 1) atomicAdd(&A.dist[(size_t)i * A.ld + j], (unsigned)V);
                        if (!A.ncomp) continue;
                        const unsigned val = (unsigned)V + (ks == 0 ? A.L - (A.c_n ? A.c_n[i] + A.c_n[j] : 0u) : 0u);
                        if (A.count_store) A.ncomp[(size_t)i * A.ld + j] = val;
                        else atomicAdd(&A.ncomp[(size_t)i * A.ld + j], val);
                        continue;
                    }
                    const int S = (int)accS[rb][cb][r];
                    // consensus: d, nn of the range.  general: L_range - G + 3 NN and NN (general_sparse_fixup completes both)
                    const unsigned d = GENERAL ? Lc - (unsigned)S + 3u * (unsigned)V : (unsigned)((3 * V - S) >> 2);
                    const size_t o = (size_t)i * A.ld + j;
                    if (dead) {
                        // consensus: flagged.  general: the prefix's lower bound itself (> threshold, and the sparse terms added
                        // later only raise it) -- a flag value could collide with a legitimately negative intermediate
                        A.dist[o] = (GENERAL || A.keep_bound) ? d : 0xFFFFFFFFu;
                        if (A.ncomp) A.ncomp[o] = 0u;
                    } else if (single) {
                        A.dist[o] = d;
                        if (A.ncomp) A.ncomp[o] = (unsigned)V;
                    } else {                                   // cells were zeroed (split range) / hold the prefix counts
                        atomicAdd(&A.dist[o], d);
                        if (A.ncomp) atomicAdd(&A.ncomp[o], (unsigned)V);
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
typedef void (*MfmaLaunchFn)(unsigned nwg, hipStream_t stream, const MfmaArgs &a);
template <bool GENERAL, int NBR, int NBC, int GC, int NWR = 2, int NWC = 2, bool COUNT = false>
static void launch_one(unsigned nwg, hipStream_t stream, const MfmaArgs &a)
{
    hipLaunchKernelGGL((pairsnp_mfma_kernel<GENERAL, NBR, NBC, GC, NWR, NWC, COUNT>), dim3(nwg), dim3(NWR * NWC * 64), 0, stream, a);
}

struct ShapeEntry { MfmaShape s; MfmaLaunchFn cons, gen; };
#define TRACS_SHAPE_N(NAME, R, C, GCC, WPC) {{NAME, R, C, 64 * (R), 64 * (C), GCC, 2, WPC}, launch_one<false, R, C, GCC>, launch_one<true, R, C, 2>}
#define TRACS_SHAPE_W(NAME, R, C, GCC, WR, WC, WPC) {{NAME, R, C, 32 * (R) * (WR), 32 * (C) * (WC), GCC, 2, WPC}, launch_one<false, R, C, GCC, WR, WC>, launch_one<true, R, C, 2, WR, WC>}
#define TRACS_SHAPE(R, C, GCC, WPC) TRACS_SHAPE_N(#R "x" #C, R, C, GCC, WPC)
static const ShapeEntry kShapes[] = {
    TRACS_SHAPE(2, 2, 1, 2),                      // 0: 128 x 128 pairs per workgroup, four waves, two workgroups per CU -- consensus default
    TRACS_SHAPE_W("2x2w4x2", 2, 2, 1, 4, 2, 1),   // 1: eight waves (4 x 2), 256 x 128 pairs, one workgroup per CU: a quarter less staging per
                                                  //    matrix instruction -- general default (505 vs 528 ms; consensus 354 vs 348)
    TRACS_SHAPE(3, 2, 2, 1),                      // 2: 192 x 128, 192 accumulator AGPRs, one wave per SIMD: fewer expansions per matrix
                                                  //    instruction, but 18 % slower: one wave cannot hide its own stalls
#ifdef TRACS_MFMA_SWEEP
    TRACS_SHAPE_N("2x2g2", 2, 2, 2, 2),           // consensus with two groups per stage (half the barriers)
    TRACS_SHAPE(2, 3, 2, 1),
    TRACS_SHAPE_W("2x2w2x4", 2, 2, 1, 2, 4, 1),   // 128 x 256
#endif
};
#undef TRACS_SHAPE
#undef TRACS_SHAPE_N
#undef TRACS_SHAPE_W

int mfma_shape_count() { return (int)(sizeof(kShapes) / sizeof(kShapes[0])); }
const MfmaShape &mfma_shape(int idx) { return kShapes[idx].s; }
int mfma_shape_current(bool general)
{
    static const int forced = [] {
        if (const char *e = std::getenv("TRACS_MFMA_TILE"))
            for (int i = 0; i < mfma_shape_count(); i++)
                if (!std::strcmp(e, kShapes[i].s.name)) return i;
        return -1;
    }();
    return forced >= 0 ? forced : (general ? 1 : 0);
}

int launch_pairsnp_mfma(int shape, bool general, unsigned nwg, hipStream_t stream, const MfmaArgs &a)
{
    if (shape < 0 || shape >= mfma_shape_count()) { set_error("launch_pairsnp_mfma: bad shape"); return TRACS_E_ARG; }
    (general ? kShapes[shape].gen : kShapes[shape].cons)(nwg, stream, a);
    return TRACS_OK;
}

// ---- the counting form's workgroup tiles: wave tile 2 x 2 blocks (64 accumulator registers), NWR x NWC waves ------------
constexpr int GC_COUNT = 4;                       // groups per stage (two residue-class rounds per barrier)
#define TRACS_COUNT_SHAPE_G(NAME, R, C, WR, WC, WPC, G) {NAME, 32 * (R) * (WR), 32 * (C) * (WC), G, WPC, launch_one<false, R, C, G, WR, WC, true>}
#define TRACS_COUNT_SHAPE(NAME, R, C, WR, WC, WPC) TRACS_COUNT_SHAPE_G(NAME, R, C, WR, WC, WPC, GC_COUNT)
static const CountShape kCountShapes[] = {
    // measured at 10 000 x 5 Mbp (profiles/r02/count_tile_sweep.txt): 4x2 73.7 ms, 2x2 75.8, 4x4 77.4 (sixteen-wave barriers)
    TRACS_COUNT_SHAPE("4x2", 2, 2, 4, 2, 2),      // 256 x 128 pairs, eight waves, two workgroups per CU -- default
    TRACS_COUNT_SHAPE("2x2", 2, 2, 2, 2, 4),      // 128 x 128, four waves
    TRACS_COUNT_SHAPE("4x4", 2, 2, 4, 4, 1),      // 256 x 256, sixteen waves
    TRACS_COUNT_SHAPE("3x2b", 3, 2, 2, 2, 2),     // 192 x 128 (the pair kernel's alternative tile: live tiles of a thresholded run)
#ifdef TRACS_MFMA_SWEEP
    TRACS_COUNT_SHAPE_G("4x2g2", 2, 2, 4, 2, 2, 2),
    TRACS_COUNT_SHAPE_G("4x2g8", 2, 2, 4, 2, 1, 8),
    TRACS_COUNT_SHAPE_G("2x2g2", 2, 2, 2, 2, 4, 2),
    TRACS_COUNT_SHAPE_G("2x2g8", 2, 2, 2, 2, 2, 8),
    TRACS_COUNT_SHAPE("2x4", 2, 2, 2, 4, 2),      // 128 x 256
#endif
};
#undef TRACS_COUNT_SHAPE
#undef TRACS_COUNT_SHAPE_G

CountShape count_shape_current()
{
    static const int forced = [] {
        if (const char *e = std::getenv("TRACS_COUNT_TILE"))
            for (size_t i = 0; i < sizeof(kCountShapes) / sizeof(kCountShapes[0]); i++)
                if (!std::strcmp(e, kCountShapes[i].name)) return (int)i;
        return 0;
    }();
    return kCountShapes[forced];
}

CountShape count_shape_like(int ti, int tj)
{
    for (const CountShape &c : kCountShapes)
        if (c.ti == ti && c.tj == tj) return c;
    return CountShape{"none", ti, tj, GC_COUNT, 1, nullptr};
}

}  // namespace tracs
