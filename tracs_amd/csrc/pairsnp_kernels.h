// pairsnp_kernels.h -- declarations shared by the pair-loop translation units (pairsnp.hip: pack, VALU tile kernel,
// COO, host driver; pairsnp_mfma.hip: matrix-core kernels; general_sparse.hip: partial-code correction).
#pragma once
#include "common.h"

namespace tracs {

// Two-pass thresholded runs (tracs_pairsnp_dense_thr on long alignments):
//   phase 1  "prefix":    one workgroup per tile walks a SHORT prefix of the alignment; a tile whose every pair already
//                         exceeds the threshold there is dead (cells 0xFFFFFFFF, live[tile] = 0), the others keep their
//                         exact partial counts in dist/ncomp (live[tile] = 1);
//   phase 2  "remainder": groups [g_base, groups) of the live tiles only (compacted tile list), split over ksplit
//                         workgroups that ADD their partial counts onto the prefix's;
//   phase 0  everything in one launch (the unthresholded path and short alignments).
struct TilePhase {
    int phase;
    int g_base;
    unsigned char *live;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// XCD-aware, bijective remap of the hardware block id: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a
// contiguous run of the logical schedule so that the tiles resident on one XCD at a time are neighbours and share
// row/column panels in L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, k = b >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

// ---- matrix-core kernels (pairsnp_mfma.hip) ---------------------------------------------------------------------
struct MfmaArgs {
    const uint4 *P;            // consensus planes (3 per group) or general planes (5 per group)
    size_t n_pad;
    int groups;                // end of the group range of this launch
    const int2 *tiles;
    int n_tiles;
    int gps, ksplit;           // groups per workgroup range, ranges per tile
    unsigned L, n, row_end, col_begin;
    unsigned *dist, *ncomp;
    size_t ld;
    unsigned thr;
    TilePhase ph;
    const unsigned *c_n = nullptr;   // counting form: per sample, its N sites among the sites the pass reads
    int count_gp = 1;          // counting form: planes per group of the source (1: iplanes; NPLANES: the stored N plane in place)
    int count_store = 0;       // counting form: store nn instead of adding to the cells (in-place source, one range per tile)
    int keep_bound = 0;        // consensus form, thresholded runs: dead cells keep their lower bound instead of the 0xFFFFFFFF flag
                               // (terms are added to the cells afterwards: minority sites)
    int count_mode = 0;        // counting form: 0 compared-sites counts only; 1 and dist += sum v_i v_j; 2 dist -= sum v_i v_j, nothing else
                               // (nw_gram: the two passes of U U^T - n n^T)
};

struct MfmaShape {
    const char *name;
    int nbr, nbc;              // 32 x 32 blocks per wave: the workgroup (2 x 2 waves) owns a (64 nbr) x (64 nbc) tile
    int ti, tj;
    int gc_cons, gc_gen;       // groups per LDS stage, consensus / general encoding (0: shape not built for it)
    int wg_per_cu;
};
// The shapes compiled into the library; one default per encoding.  TRACS_MFMA_TILE=<name> selects another for both (diagnostics).
int mfma_shape_count();
const MfmaShape &mfma_shape(int idx);
int mfma_shape_current(bool general);
// general = false: consensus encoding (operands x, y, z, v);  true: general encoding (one-hot A, C, G, T + N).
int launch_pairsnp_mfma(int shape, bool general, unsigned nwg, hipStream_t stream, const MfmaArgs &a);
// The counting form (site classes): a.P = one plane per group ("is a base here"), ncomp[i][j] += sum v_i v_j over the groups of
// the launch (+ a.L once per cell).  Its own workgroup tiles: one accumulator set per wave leaves room for more waves per CU, and
// the pass is the memory-hungrier per matrix instruction, so larger workgroup tiles pay.
struct CountShape {
    const char *name;
    int ti, tj, gc, wg_per_cu;
    void (*fn)(unsigned nwg, hipStream_t stream, const MfmaArgs &a);
};
CountShape count_shape_current();               // the default (TRACS_COUNT_TILE=<name> selects another: diagnostics)
CountShape count_shape_like(int ti, int tj);    // the shape with this workgroup tile (fn == nullptr when there is none)

// ---- site classes and the encoding decision (site_classes.hip) -------------------------------------------------------
// Classifies the sites of the general planes once per pack.  *partial: 1 some sample carries a partial IUPAC code, 0 none
// does (the alignment has a consensus form), -1 not determined (classification skipped).  On return a->classes_state is 1
// (classes in use: vplanes in the encoding a->classes_cons says, counting pass source, minority lists) or -1.
int site_classes_decide(tracs_alignment *a, hipStream_t stream, int *partial);
void site_classes_free(tracs_alignment *a);
// stage clock of the once-per-pack work (HIP events on the launch stream; tracs_debug_pack_stages, TRACS_CLASSES_TRACE)
void pack_stage_begin(hipStream_t stream);
void pack_stage_mark(const char *name, hipStream_t stream, double bytes_read = 0.0, double bytes_written = 0.0);
void pack_stage_end();

// ---- sparse side structures of the general matrix-core path (general_sparse.hip) -------------------------------
struct GeneralSparse;
// Builds (or returns the cached) per-site / per-sample lists of N and partial-code entries.  *ok = 0 when the alignment is
// outside what the path supports (too long, too many entries, no memory): the caller then stays on the VALU kernel.
int general_sparse_get(tracs_alignment *a, hipStream_t stream, int *ok, double *est_updates);
void general_sparse_free(tracs_alignment *a);
// dist[i][j] += T1 + T2 (partial-code terms), ncomp[i][j] += L - c_i - c_j for the cells of the dense region.
int general_sparse_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist,
                         unsigned *ncomp, size_t ld, hipStream_t stream);
// ---- lists of an alignment cut into site classes (site_lists.hip) ---------------------------------------------------------
// Two kinds of site carry lists, under one rank space (the sites of M_LST in site order):
//   MINORITY sites  a sample that is neither N nor exactly the site's reference base is LISTED with its allele mask M and
//                   w = [reference base not in M] (p lists); minority_fixup adds the sites' contribution to dist;
//   NNL sites       few N samples: nn_rows_add takes their N co-occurrences NN = sum n_i n_j from the sites' N lists.
// N LISTS ("n8" lines).  The N samples of a site, in sample order, as byte deltas in 128-byte lines: line r of `lines` is the
// primary line of the site of rank r; 124 payload bytes, then the index of the list's next line (0xFFFFFFFF: none).  With p the
// position before a byte b (-1 at the head of a list):
//     b <= 252   p += b, sample p is N here           (b = 0 only straight behind a skip)
//     b == 253   p += 253, no sample                   (a gap g is g / 253 skips and the byte g % 253)
//     b == 255   padding behind the last sample (nothing follows it in the line)
// A site of ~100 N samples among 10 000 is ONE cache line (two 128-byte lines of 16-bit sample numbers before); a row's walk
// decodes a line with eight lanes, 16 bytes each: byte sums per lane, an 8-lane prefix, one SDWA add per byte.
constexpr unsigned N8_PAYLOAD = 124, N8_SKIP = 253, N8_NONE = 0xFFFFFFFFu;
// lines a list of c samples among n can need at most (its gaps sum to <= n: at most n / 253 skip bytes in all)
__host__ __device__ inline unsigned n8_lines_max(unsigned c, unsigned n) { const unsigned b = c + n / N8_SKIP; return b <= N8_PAYLOAD ? 1u : (b + N8_PAYLOAD - 1u) / N8_PAYLOAD; }
// ... and will need on average (cost model: uniformly placed samples -- a gap exceeds 253 with probability q = (1 - c / n)^253)
__host__ __device__ inline float n8_lines_expected(unsigned c, unsigned n)
{
    const float q = __builtin_expf(-253.0f * (float)c / (float)(n ? n : 1u));
    const float bytes = (float)c + (float)(c + 1u) * q / (1.0f - (q < 0.99f ? q : 0.99f));
    const float cap = (float)c + (float)(n / N8_SKIP);
    return __builtin_ceilf((bytes < cap ? bytes : cap) / (float)N8_PAYLOAD);
}
// p lists of at most this many samples stay in p_ent (walked in the lane that finds them); longer ones are q lines (site_lists.hip)
constexpr unsigned P_SHORT_MAX = 4;
// q lines are 256 bytes instead of 128 when the minority sites list more than this many samples on average
constexpr double Q_WIDE_MEAN = 24.0;

struct MinorBuild {
    const uint4 *planes;                     // the five general planes
    const uint4 *minor_mask, *nnl_mask, *lst_mask, *un_mask;   // per group: minority sites, N co-occurrence list sites, their union; sites outside the dense class with an N
    const uint4 *ref_x, *ref_y;              // per group: reference base bits
    const unsigned *off_lst;                 // per group: sites with lists before it (a site's list index = its rank among them)
    const unsigned *cntP, *cntN;             // per site: listed samples, N samples
    const unsigned *gP;                      // per group: p-list entries of its minority sites
    unsigned long long max_gp;               //   ... and the most any group has
    const unsigned long long *baseP, *baseO; // per group: p-list entries / overflow lines (upper bounds) of the groups before it
    const unsigned long long *baseQ;         // per group: overflow lines of the p lists (q lines) of the groups before it
    const unsigned long long *flags;         // per group and 64 samples: listed somewhere in the group
    size_t flag_words;
    unsigned rows[4];                        // the N bitmap rows (nn_rows_add) only for the samples of these [begin, end) ranges
    int n_rows;                              //  (n_rows of them; 0: every sample) -- tracs_alignment_hint_rows
    size_t sites;                            // sites with lists
    unsigned long long tot_p, tot_o;         // p-list entries, overflow lines (upper bound) in all
    unsigned long long tot_q;                // overflow lines of the p lists in all
    int long_p;                              // some minority site lists more than P_SHORT_MAX samples: its p list is a q line (else: no q lines at all)
    unsigned qw;                             // dwords per q line: 32, or 64 when the lists are long on average (site_lists.hip)
    unsigned long long tot_nnl;              // N samples at the NNL sites: list walks of one pass of nn_rows_add
    int gram = 0;                            // != 0: no N lists at all -- the minority sites' N x listed terms come from the matrix cores (1: U U^T - n n^T,
                                             // site_classes.hip) or from the rows of the site-major N matrix (2: the builder also makes NS and the ranks' sites)
};
int minority_lists_build(tracs_alignment *a, const MinorBuild &mb, hipStream_t stream, int *ok);
void minority_lists_free(tracs_alignment *a);
int minority_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist, size_t ld, hipStream_t stream);
// ncomp[i][j] += NN over the sites whose N co-occurrences come from lists (+ lu - c_i - c_j when add_terms: nobody else adds them)
int nn_rows_add(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *ncomp, size_t ld, int add_terms,
                unsigned lu, hipStream_t stream);

// NS[site][sample word]: the N plane bit-transposed to site-major, ns_words 32-sample words per site (a multiple of 32: whole 128-byte
// lines): what the recombination filter asks "is j N at a site of i's list" (filter_lists.hip) and what the second form of the site
// classes sums per listed sample (site_lists.hip, minor_fixup_kernel<NSROWS>)
static inline size_t ns_words_for(size_t n_pad) { return (n_pad / 32 + 31) / 32 * 32; }
void launch_ns_build(const uint4 *planes, size_t n_pad, unsigned groups, unsigned *ns, size_t ns_words, hipStream_t stream);   // filter_lists.hip

#ifdef __HIPCC__
// add the 32 one-bit values of `m` to 32 bit-sliced counters (plane j holds bit j of every counter); <= 255 adds between flushes
__device__ __forceinline__ void sliced_add(unsigned (&p)[8], unsigned m)
{
    unsigned c = m;
#pragma unroll
    for (int j = 0; j < 8; j++) { const unsigned t = p[j] & c; p[j] ^= c; c = t; }
}

// carry-save adder over 32 one-bit lanes: three addends of weight w -> sum (weight w) and carry (weight 2w); gfx950's v_bitop3_b32
// evaluates either in one instruction (truth tables 0x96 = a ^ b ^ c, 0xE8 = majority)
__device__ __forceinline__ void csa(unsigned &hi, unsigned &lo, unsigned a, unsigned b, unsigned c)
{
    lo = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    hi = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
}

// add eight 32-lane one-bit values to the bit-sliced counters: a Harley-Seal tree folds them into the planes of weight 1, 2, 4 and
// ONE carry of weight 8, which then ripples through planes 3..7 -- 24 instructions for eight addends (sliced_add: 16 each)
__device__ __forceinline__ void sliced_add8(unsigned (&p)[8], const unsigned (&x)[8])
{
    unsigned twos_a, twos_b, fours_a, fours_b, eights;
    csa(twos_a, p[0], p[0], x[0], x[1]);
    csa(twos_b, p[0], p[0], x[2], x[3]);
    csa(fours_a, p[1], p[1], twos_a, twos_b);
    csa(twos_a, p[0], p[0], x[4], x[5]);
    csa(twos_b, p[0], p[0], x[6], x[7]);
    csa(fours_b, p[1], p[1], twos_a, twos_b);
    csa(eights, p[2], p[2], fours_a, fours_b);
    unsigned c = eights;
#pragma unroll
    for (int j = 3; j < 8; j++) { const unsigned t = p[j] & c; p[j] ^= c; c = t; }
}

// four addends: planes of weight 1 and 2 and one carry of weight 4 that ripples through planes 2..7
__device__ __forceinline__ void sliced_add4(unsigned (&p)[8], const unsigned (&x)[4])
{
    unsigned twos_a, twos_b, fours;
    csa(twos_a, p[0], p[0], x[0], x[1]);
    csa(twos_b, p[0], p[0], x[2], x[3]);
    csa(fours, p[1], p[1], twos_a, twos_b);
    unsigned c = fours;
#pragma unroll
    for (int j = 2; j < 8; j++) { const unsigned t = p[j] & c; p[j] ^= c; c = t; }
}

// 32 x 32 bit transpose across the 32 lanes of a half wave: lane k holds row k; afterwards bit j of lane k is bit k of what lane j
// held.  Five butterfly steps (j = 16, 8, 4, 2, 1): lanes k and k ^ j exchange words (ds_swizzle, bit mode) and swap the high
// half-blocks of the lower lane's word with the low half-blocks of the higher lane's -- per lane: keep the bits of K, take the
// partner's word rotated by R everywhere else: one rotate (v_alignbit_b32) and one bitfield insert (v_bfi_b32).
struct Transpose32 {
    unsigned K[5], R[5];                       // per lane and step: bits kept, rotation of the partner's word
    __device__ __forceinline__ explicit Transpose32(unsigned lane)
    {
        constexpr unsigned masks[5] = {0x0000FFFFu, 0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
        for (int st = 0; st < 5; st++) {
            const unsigned j = 16u >> st;
            const bool hi = (lane & j) != 0u;
            K[st] = hi ? (masks[st] << j) : masks[st];
            R[st] = hi ? j : 32u - j;          // v_alignbit(v, v, R) = rotate right by R: the higher lane takes v >> j, the lower v << j
        }
    }
    __device__ __forceinline__ unsigned operator()(unsigned a) const
    {
        unsigned v;
        // (ds_swizzle bit mode: offset = xor_mask << 10 | or_mask << 5 | and_mask, inside groups of 32 lanes)
        v = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, (16 << 10) | 0x1F); a = (a & K[0]) | (__builtin_amdgcn_alignbit(v, v, R[0]) & ~K[0]);
        v = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, (8 << 10) | 0x1F);  a = (a & K[1]) | (__builtin_amdgcn_alignbit(v, v, R[1]) & ~K[1]);
        v = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, (4 << 10) | 0x1F);  a = (a & K[2]) | (__builtin_amdgcn_alignbit(v, v, R[2]) & ~K[2]);
        v = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, (2 << 10) | 0x1F);  a = (a & K[3]) | (__builtin_amdgcn_alignbit(v, v, R[3]) & ~K[3]);
        v = (unsigned)__builtin_amdgcn_ds_swizzle((int)a, (1 << 10) | 0x1F);  a = (a & K[4]) | (__builtin_amdgcn_alignbit(v, v, R[4]) & ~K[4]);
        return a;
    }
};

#endif

}  // namespace tracs
