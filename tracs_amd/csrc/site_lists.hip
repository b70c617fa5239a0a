// site_lists.hip -- the lists of an alignment cut into site classes (site_classes.hip), and the two kernels that walk them.
//
// Reference behaviour restated (never copied): /root/reference/src/pairsnp.hpp:395-420 -- every site is visited for every pair.
// Here a site at which only a few samples differ from the others (MINORITY) or are N (NNL) contributes through lists:
//
//   per site     N LIST ("n8" lines, pairsnp_kernels.h): its N samples in sample order, byte deltas, 124 per 128-byte line -- the site
//                of rank r (among the sites with lists) owns line r; a list that needs more goes on in overflow lines of its group;
//                P LIST ("q" lines of 32 dwords): its listed samples (sample << 5 | w << 4 | allele mask) behind a header, minority sites only --
//                the site of rank r owns q line r, longer lists go on in consecutive overflow lines;
//   per sample   its LISTED entries with w = 1 (rank << 5 | w << 4 | mask: its mask lacks the site's reference base -- the entries that
//                walk lists) -- few: a sample differs from the others at a few hundred sites;
//                its N BITMAP over the NNL sites, sample-major (T: 16 bytes per 128-site group, the transposed N plane): what
//                nn_rows_kernel reads instead of a stream of list addresses.
//
// Built from the N plane alone (+ the five planes of the flagged samples, ~1 %): site_lists_kernel reads it once, group by group
// (bits transposed through LDS, encoded by the site's own thread), n_bitmap_kernel once more, sample by sample (+ each sample's N count).
//
// The walks.  nn_rows_kernel: row i of the pair matrix in LDS; every N site of sample i (a set bit of its bitmap) is a work item --
// the site's line(s), scanned by four lanes and decoded piece by piece: NN(i, j) += 1 for every listed j > i.  minor_fixup_kernel:
// row x; every such entry of sample x walks its site's P list (both listed: [masks disjoint] - w_x - w_j) and the site's N LIST:
// -w_x for EVERY N sample y -- y > x lands in row x of dist, y < x in cell (y, x): a scratch row that transpose_add_kernel folds into
// the rows above.  So an N sample never looks at the p list of a site: the walks go from the few listed samples to the many N
// samples (k walks of a cN-entry list, not cN walks of a k-entry list), and the per-sample streams of N entries that rounds 2-3
// built (s_nn, s_inl: 3.3 GB written once per pack from two more reads of the N plane) do not exist.
#include "pairsnp_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace tracs {

struct SiteLists {
    uint4 *lines = nullptr;                    // n8 lines: [0, sites) primary, then each group's overflow block
    unsigned long long n_lines = 0;
    unsigned long long *p_off = nullptr;       // [sites + 1]
    unsigned *p_ent = nullptr;                 // p lists of at most P_SHORT_MAX samples (the usual case: one or two): p_ent[p_off[r] ..]
    uint4 *qlines = nullptr;                   // the longer p lists as "q lines" of 32 dwords: the site of rank r owns line r -- dword 0 its header
                                               //   (k | w1 << 16: listed samples, and how many of them, at the front, have w = 1: a listed sample
                                               //   whose mask contains the reference base only ever pairs up with those), dwords 1..30 its first
                                               //   entries (sample << 5 | w << 4 | mask), dword 31 the index of its first overflow line; overflow
                                               //   lines (31 entries each) are consecutive.  ONE line fetch per walk for lists of up to 30
    unsigned long long n_qlines = 0;
    unsigned qw = 32;                          // dwords per q line: 32 -- or 64 (a 256-byte line: header, 62 entries, first overflow line; 63 entries per
                                               //   overflow line) when the alignment's minority sites list more than Q_WIDE_MEAN samples on average: a
                                               //   51-entry list is one fetch then, not a line and, behind its header, a second one
    unsigned long long *s_off = nullptr;       // [n + 1]
    unsigned *s_ent = nullptr;
    unsigned *c_p = nullptr;                   // per sample: sum of w over its listed entries
    uint4 *T = nullptr;                        // N bitmaps of the rows over the NNL sites: T[s * tgroups + g]
    size_t tgroups = 0;
    uint4 *lst_mask = nullptr;                 // per group: sites with lists (copies: the classification's own live in shared scratch)
    unsigned *off_lst = nullptr;               // per group: rank of its first site with lists
    size_t sites = 0, groups = 0;
    unsigned long long tot_p = 0, tot_nnl = 0;
    unsigned max_row = 0;                      // the most N sites (outside the dense class) any sample has: row splits of nn_rows_kernel
    unsigned rows[4] = {0, 0, 0, 0};           // T holds the rows of these ranges only (n_rows of them; 0: all)
    int n_rows = 0;
    // nw_rows (the second form of the classes without its U pass): the N plane site-major and the site of every rank
    unsigned *ns = nullptr;                    // [site][ns_words] sample bits (pairsnp_kernels.h: launch_ns_build)
    size_t ns_words = 0;
    unsigned *site_of = nullptr;               // [sites]: site of the list rank
};
constexpr int ENT_SHIFT = 5;                   // entries: index << 5 | w << 4 | 4-bit allele mask
constexpr unsigned ENT_HOLE = 0xFFFFFFFFu;     // E: (sample, entry) by list position; sample = ENT_HOLE where the list's entry has w = 0
constexpr unsigned ENT_LONG = 0x80000000u;     // per-sample entries: the site's p list is a q line (ranks stay below 2^26)

__device__ __forceinline__ unsigned word_of(const uint4 &v, int w) { return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w; }

__device__ __forceinline__ unsigned lst_rank(const uint4 &m, unsigned off_g, int w, int b)
{
    unsigned r = off_g;
    if (w > 0) r += __popc(m.x);
    if (w > 1) r += __popc(m.y);
    if (w > 2) r += __popc(m.z);
    return r + __popc(word_of(m, w) & ((1u << b) - 1u));
}

__device__ __forceinline__ bool row_wanted(const MinorBuild &mb, size_t s)
{
    if (mb.n_rows == 0) return true;
    return (s >= mb.rows[0] && s < mb.rows[1]) || (mb.n_rows > 1 && s >= mb.rows[2] && s < mb.rows[3]);
}

// ---- per site: N lists (n8 lines) and p lists ---------------------------------------------------------------------------------
// One workgroup per 128-site group.  The group's N bits are TRANSPOSED -- samples x sites, as the plane holds them, into sites x
// samples -- through LDS, 512 samples (a PIECE) at a time: a wave takes 64 samples (coalesced 16-byte loads), transposes each
// 32 x 32 bit block in registers (five butterfly steps of lane exchanges: Transpose32) and writes the site-major words; then the
// site's own thread reads its 32 words of the piece in order and feeds every set bit to its encoder -- the samples come out sorted,
// no cursor, no atomic, no sort, and the work does not depend on how many samples are N (rounds 3-4a dropped sample numbers
// into per-site runs through LDS cursors and sorted the runs: quadratic in the drift of the waves, 114 ms per pack at 10 % N).
// The encoder state (last position, the line's fill, the pending 16 bytes) stays in the site thread's registers from piece to piece.
// (512 samples a piece: 3.3 ms per call at 10 000 x 5 Mbp against 4.0 with 1 024 -- eight workgroups of 14 KB per CU, the registers'
// limit, instead of seven of 22 KB --, 3.6 with 256, 4.6 with 128, 7.3 with 2 048: profiles/r05/site_lists_piece_sweep.txt)
// ... for consensus alignments; with partial IUPAC codes (47 % of the samples of a group flagged for the p lists at 0.5 % of them)
// the queue of flagged samples wants the longer piece: 10.9 ms with 1 024 against 13.3 with 512 -- the kernel is a template on it.
// a group's p lists are built in LDS by p_lists_kernel when they hold PL_MIN .. PL_CAP entries (fewer: not worth a workgroup of its
// own -- the bench alignment has 128 per group --, more: beyond the image; both: site_lists_kernel, entry by entry)
// (512 threads: 8.17 ms of per-site list building against 8.42 with 256 on the partial-code alignment -- profiles/r06/partial_floor.txt)
#ifndef TRACS_PL_THREADS
#define TRACS_PL_THREADS 512
#endif
constexpr unsigned PL_THREADS = TRACS_PL_THREADS, PL_MIN = 1024, PL_CAP = 8192, PL_CHUNK = 4096, PL_PER_THREAD = PL_CHUNK / PL_THREADS;
static_assert(PL_PER_THREAD * PL_THREADS == PL_CHUNK && PL_PER_THREAD <= 32 && 64 % PL_PER_THREAD == 0, "p_lists_kernel: a thread's samples sit in one flag word");
__device__ __forceinline__ bool p_lists_in_lds(unsigned group_entries) { return group_entries >= PL_MIN && group_entries <= PL_CAP; }
constexpr unsigned SITE_THREADS = 128;         // one thread per site of the group: both waves busy in both phases, eight workgroups per CU (17 KiB of LDS each)

struct N8Encoder {
    uint4 *lines;
    unsigned line, next_ovf, fill, prev;
    unsigned left;                             // samples of the site still to come
    unsigned a0, a1, a2, a3;                   // the pending bytes, shifted in from the top
    static constexpr unsigned DONE = 0xFFFFu;  // fill once the list has left with its last line
    // one byte; `more`: something of the list follows it (a skip byte, or a sample's byte with samples still to come).  One store
    // site for both kinds of piece -- a full 16 bytes, and the line's last 12 with the index of the line that goes on -- : the
    // wave runs it whenever one of its 64 sites stores, i.e. nearly every time, so it is run once per byte, not twice
    __device__ __forceinline__ void put(unsigned b, bool more)
    {
        a0 = __builtin_amdgcn_alignbit(a1, a0, 8); a1 = __builtin_amdgcn_alignbit(a2, a1, 8); a2 = __builtin_amdgcn_alignbit(a3, a2, 8);
        a3 = (a3 >> 8) | (b << 24);
        fill++;
        const bool full = fill == N8_PAYLOAD;
        if ((fill & 15u) == 0u || full) {
            const unsigned nx = more ? next_ovf : N8_NONE;
            lines[(size_t)line * 8 + (full ? 7u : (fill >> 4) - 1u)] = full ? make_uint4(a1, a2, a3, nx) : make_uint4(a0, a1, a2, a3);
            if (full) {
                if (more) { line = next_ovf++; fill = 0; }
                else fill = DONE;
            }
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (fill == DONE) return;              // (the list ended with the last byte of a line)
        const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        while ((fill & 15u) != 0u && fill != DONE) put(0xFFu, false);          // the piece under way (at 124 bytes: the line's last)
        if (fill == DONE) return;
        for (unsigned q = fill >> 4; q < 8u; q++) lines[(size_t)line * 8 + q] = ones;     // (the last: padding + N8_NONE)
    }
};

// (a register budget of eight waves per SIMD -- 64 VGPRs; LDS holds six: 3.32 -> 3.21 ms at 10 000 x 5 Mbp against the budget of six, 80
// VGPRs, of rounds 5 - 6a; five: 3.49, four: 4.06, the compiler's own choice: 3.24 -- profiles/r06/classify_threads.txt)
#ifndef TRACS_SITE_WAVES
#define TRACS_SITE_WAVES 8
#endif
#if TRACS_SITE_WAVES > 0
#define TRACS_SITE_ATTR __attribute__((amdgpu_waves_per_eu(TRACS_SITE_WAVES, TRACS_SITE_WAVES)))
#else
#define TRACS_SITE_ATTR
#endif
template <unsigned PIECE_SAMPLES>
__global__ __launch_bounds__(SITE_THREADS) TRACS_SITE_ATTR void site_lists_kernel(const MinorBuild mb, size_t n_pad, unsigned n,
                                                         unsigned long long *__restrict__ p_off, unsigned *__restrict__ p_ent,
                                                         unsigned *__restrict__ qd, uint2 *__restrict__ E, unsigned *__restrict__ e_cnt,
                                                         uint4 *__restrict__ lines,
                                                         unsigned *__restrict__ c_p)
{
    constexpr unsigned PIECE_WORDS = PIECE_SAMPLES / 32, BM_STRIDE = PIECE_WORDS + 1;     // (odd stride: conflict-free column writes)
    __shared__ unsigned bm[SITES_PER_GROUP * BM_STRIDE];     // the piece's N bits, site-major: bm[site * BM_STRIDE + 32-sample word]
    // (LDS decides how many of these workgroups a CU holds -- seven at 22.5 KB --: ovf is only read while the sites' bases are summed and
    // serves as the back cursor of the p lists afterwards; a site's own N count lives in its thread)
    __shared__ unsigned kp[SITES_PER_GROUP], ovf[SITES_PER_GROUP], curP[SITES_PER_GROUP], rk[SITES_PER_GROUP];
    __shared__ unsigned qb[SITES_PER_GROUP];                 // first overflow line of the site's p list
    unsigned *const curQ = ovf;
    __shared__ unsigned long long bP[SITES_PER_GROUP];
    __shared__ unsigned short queue[PIECE_SAMPLES];          // the piece's samples with listed sites in this group (offsets into the piece)
    __shared__ unsigned qn[2];                               // (by parity of the piece: the other one is reset while this one is read)
    const size_t g = blockIdx.x;
    const int tid = threadIdx.x;
    const unsigned lane = tid & 63u, wave = tid >> 6;
    if (g == 0 && tid == 0) p_off[mb.sites] = mb.tot_p;
    const uint4 m4 = mb.lst_mask[g], q4 = mb.minor_mask[g];
    if ((m4.x | m4.y | m4.z | m4.w) == 0u) return;
    const unsigned m[4] = {m4.x, m4.y, m4.z, m4.w};          // sites with lists
    const unsigned mp[4] = {q4.x, q4.y, q4.z, q4.w};         // minority sites among them (p lists)
    const bool any_minor = (q4.x | q4.y | q4.z | q4.w) != 0u;
    // the group's p lists: p_lists_kernel's, or -- few entries, or more than that kernel sorts in LDS -- built here, from the planes of
    // its flagged samples, piece by piece, entry by entry
    const bool from_planes = any_minor && !p_lists_in_lds(mb.gP[g]);
    if (from_planes && tid == 0) e_cnt[g] = 0xFFFFFFFFu;     // (its entries in E: by list position, with holes)
    static_assert(SITE_THREADS == SITES_PER_GROUP, "thread = site");
    const int tw = tid >> 5, tb = tid & 31;
    const bool mine = (m[tw] >> tb) & 1u;
    const bool want_n = mb.gram == 0;                        // (gram: the sites carry p lists only -- the N plane is not even read)
    const unsigned my_cn = (mine && want_n) ? mb.cntN[g * SITES_PER_GROUP + tid] : 0u;
    {
        const unsigned c = my_cn;
        kp[tid] = (mine && ((mp[tw] >> tb) & 1u)) ? mb.cntP[g * SITES_PER_GROUP + tid] : 0u;
        ovf[tid] = mine ? n8_lines_max(c, n) - 1u : 0u;
        curP[tid] = 0;
        if (tid < 2) qn[tid] = 0;
    }
    __syncthreads();
    const Transpose32 transpose(lane);
    N8Encoder enc{lines, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    if (mine) {
        unsigned long long pp = 0;
        unsigned po = 0, pq = 0;
        for (int t = 0; t < tid; t++) { pp += kp[t]; po += ovf[t]; pq += kp[t] / 31u; }
        const unsigned rank = lst_rank(m4, mb.off_lst[g], tw, tb);
        bP[tid] = mb.baseP[g] + pp; rk[tid] = rank;
        qb[tid] = (unsigned)(mb.sites + mb.baseQ[g] + pq);
        p_off[rank] = bP[tid];
        enc.line = rank;
        enc.left = my_cn;
        enc.next_ovf = (unsigned)(mb.sites + mb.baseO[g] + po);
    }
    __syncthreads();
    curQ[tid] = 0;                                           // (first used behind the piece loop's barrier)
    const uint4 RX = mb.ref_x[g], RY = mb.ref_y[g];
    const uint4 *base = mb.planes + (g * NPLANES) * n_pad;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // listed sample s at site `st` of the group (a minority site) with code = w << 4 | allele mask: into the site's p list -- its w = 1
    // entries from the front of its run, the others from the back -- and, when w = 1, into E
    auto place = [&](unsigned s, unsigned st, unsigned code) {
        const unsigned slot = (code & 16u) ? atomicAdd(&curP[st], 1u) : kp[st] - 1u - atomicAdd(&curQ[st], 1u);
        const unsigned long long pos = bP[st] + slot;
        const bool is_long = kp[st] > P_SHORT_MAX;
        if (is_long) {
            // entry `slot` of the site's list: dword slot + 1 of its q lines (31 dwords a line; dword 0 of the list is the header)
            const unsigned qs = slot + 1u, qt = qs / (mb.qw - 1u);
            qd[(size_t)(qt ? qb[st] + qt - 1u : rk[st]) * mb.qw + (qs - qt * (mb.qw - 1u))] = (s << ENT_SHIFT) | code;
        } else {
            p_ent[pos] = (s << ENT_SHIFT) | code;
        }
        // (only a sample with w = 1 walks lists afterwards: the others' entries exist in the site's list alone -- a hole in E)
        E[pos] = (code & 16u) ? make_uint2(s, (is_long ? ENT_LONG : 0u) | (rk[st] << ENT_SHIFT) | code) : make_uint2(ENT_HOLE, 0u);
    };
    for (unsigned piece = 0; piece < n; piece += PIECE_SAMPLES) {
        const unsigned par = (piece / PIECE_SAMPLES) & 1u;
        // ---- the piece's samples, 64 per wave and step (the next step's N words and flag word are requested before this step's
        // are transposed): N bits into bm, listed samples into the p lists (every sample is seen in exactly one piece)
        unsigned sl = wave;
        uint4 N_next = zero4;
        unsigned long long fl_next = 0ull;
        {
            const unsigned s = piece + sl * 64u + lane;
            if (s < n) { if (want_n) N_next = base[4 * n_pad + s]; if (from_planes) fl_next = mb.flags[g * mb.flag_words + (s >> 6)]; }
        }
        for (; sl < PIECE_SAMPLES / 64u; sl += SITE_THREADS / 64u) {
            const unsigned s = piece + sl * 64u + lane;
            const uint4 N = N_next;
            const unsigned long long fl = fl_next;
            {
                const unsigned sn = s + SITE_THREADS;
                N_next = zero4; fl_next = 0ull;
                if (sl + SITE_THREADS / 64u < PIECE_SAMPLES / 64u && sn < n) { if (want_n) N_next = base[4 * n_pad + sn]; if (from_planes) fl_next = mb.flags[g * mb.flag_words + (sn >> 6)]; }
            }
            if (!want_n) {
            } else if (piece + sl * 64u < n) {                 // (wave-uniform: a slab beyond the last sample leaves its words zero below)
                const unsigned col = sl * 2u + (lane >> 5), r = lane & 31u;
                bm[(0u + r) * BM_STRIDE + col] = transpose(N.x & m[0]);
                bm[(32u + r) * BM_STRIDE + col] = transpose(N.y & m[1]);
                bm[(64u + r) * BM_STRIDE + col] = transpose(N.z & m[2]);
                bm[(96u + r) * BM_STRIDE + col] = transpose(N.w & m[3]);
            } else {
                const unsigned col = sl * 2u + (lane >> 5), r = lane & 31u;
                bm[(0u + r) * BM_STRIDE + col] = 0u; bm[(32u + r) * BM_STRIDE + col] = 0u;
                bm[(64u + r) * BM_STRIDE + col] = 0u; bm[(96u + r) * BM_STRIDE + col] = 0u;
            }
            // a sample that is listed somewhere in this group (~1 % of them on the bench workload): queued, so that the p-list work
            // below runs with full waves instead of once per slab with a lane or two
            const bool flagged = s < n && ((fl >> (s & 63u)) & 1ull);
            const unsigned long long fm = __ballot(flagged);
            if (fm) {                                          // (wave-uniform)
                unsigned qb = 0;
                if (lane == 0) qb = atomicAdd(&qn[par], (unsigned)__popcll(fm));
                qb = __shfl(qb, 0, 64);
                if (flagged) queue[qb + __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u))] = (unsigned short)(s - piece);
            }
        }
        __syncthreads();
        // ---- the queued samples: their listed sites go into the p lists
        const unsigned qcount = qn[par];
        if (tid == 0) qn[par ^ 1u] = 0;
        for (unsigned k = tid; k < qcount; k += SITE_THREADS) {
            const unsigned s = piece + queue[k];
            // (the stored N plane is A & C & G & T by construction: four loads, not five)
            const uint4 A = base[s], C = base[n_pad + s], G = base[2 * n_pad + s], T = base[3 * n_pad + s];
            unsigned listed_w = 0;                           // this sample's listed entries with w = 1 in the group
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned a = word_of(A, w), c = word_of(C, w), gg = word_of(G, w), t = word_of(T, w), isn = a & c & gg & t;
                const unsigned rx = word_of(RX, w), ry = word_of(RY, w);
                const unsigned ra = ~rx & ~ry, rc = rx & ~ry, rg = ~rx & ry, rt = rx & ry;
                const unsigned has_ref = (a & ra) | (c & rc) | (gg & rg) | (t & rt);
                const unsigned only_ref = ~((a ^ ra) | (c ^ rc) | (gg ^ rg) | (t ^ rt));
                // (the classification's predicate exactly -- `diff` = carries some allele bit, is not N, is not exactly the reference base:
                // the p lists' sizes come from there; a sample whose four planes are all zero at a site is listed by neither)
                unsigned pm = (a | c | gg | t) & ~isn & ~only_ref & mp[w];
                while (pm) {
                    const int b = __ffs(pm) - 1;
                    pm &= pm - 1;
                    const unsigned mask = ((a >> b) & 1u) | (((c >> b) & 1u) << 1) | (((gg >> b) & 1u) << 2) | (((t >> b) & 1u) << 3);
                    const unsigned code = (((has_ref >> b) & 1u) ? 0u : 16u) | mask;
                    place(s, (unsigned)(w * 32 + b), code);
                    listed_w += code >> 4;
                }
            }
            if (listed_w) atomicAdd(&c_p[s], listed_w);
        }
        // ---- the site's thread: its 32 words of the piece in order (four independent reads at a time), every set bit a sample
        if (mine && my_cn != 0u) {
            // (one loop over the lane's own set bits: the wave runs as many rounds as its busiest lane has samples in the piece --
            // word by word it ran the busiest lane of every word, four times as many)
            static_assert(PIECE_WORDS <= 64, "one bit per word of the piece");
            const unsigned *rowp = bm + (unsigned)tid * BM_STRIDE;
            unsigned long long nz = 0;
#pragma unroll
            for (unsigned c = 0; c < PIECE_WORDS; c++) nz |= (unsigned long long)(rowp[c] != 0u ? 1u : 0u) << c;
            // one byte per round: a sample whose gap needs skip bytes stays for as many rounds (the wave runs the encoder once per
            // round whatever the lanes emit)
            unsigned w = 0, c = 0, gap = 0;
            bool have = false;
            for (;;) {
                if (!have) {
                    if (w == 0u) {
                        if (nz == 0ull) break;
                        c = __ffsll((long long)nz) - 1; nz &= nz - 1;
                        w = rowp[c];
                    }
                    const unsigned b = __ffs(w) - 1;
                    w &= w - 1;
                    const unsigned smp = piece + 32u * c + b;
                    gap = smp - enc.prev;      // (prev = 0xFFFFFFFF before the first: smp + 1)
                    enc.prev = smp; enc.left--;
                    have = true;
                }
                const bool skip = gap >= N8_SKIP;
                enc.put(skip ? N8_SKIP : gap, skip || enc.left != 0u);
                if (skip) gap -= N8_SKIP; else have = false;
            }
        }
        __syncthreads();
    }
    if (mine) {
        if (want_n) enc.finish();
        if (from_planes && kp[tid] > P_SHORT_MAX) { qd[(size_t)rk[tid] * mb.qw] = kp[tid] | (curP[tid] << 16); qd[(size_t)rk[tid] * mb.qw + mb.qw - 1u] = qb[tid]; }
    }
}

// ---- per site: p lists ---------------------------------------------------------------------------------------------------------------
// One workgroup per group: its FLAGGED samples (listed somewhere in the group: classify_sites_kernel's flag words) are queued 4 096
// samples at a time, their four allele words read, and every listed (sample, site) pair is dropped into an LDS image of the group's
// p lists -- a counting sort on the sites' known sizes; within a site the w = 1 entries from the front, the others from the back --
// which then leaves as whole lines: the short lists into p_ent, the long ones as q lines written 128 bytes at a time, the w = 1
// entries into E.  (Dropped one by one into global memory -- site_lists_kernel's way until round 5, kept for the groups whose
// lists outgrow the image -- the 253 M entries of an alignment with 0.5 % partial codes were 253 M four-byte stores into 1.3 GB of
// q lines that left the L2 half written and came back to be finished.)
__global__ __launch_bounds__(PL_THREADS) void p_lists_kernel(const MinorBuild mb, size_t n_pad, unsigned n, unsigned *__restrict__ p_ent,
                                                             unsigned *__restrict__ qd, uint2 *__restrict__ E, unsigned *__restrict__ e_cnt,
                                                             int holes, unsigned *__restrict__ c_p)
{
    __shared__ unsigned sorted[PL_CAP];
    __shared__ unsigned w1off[SITES_PER_GROUP];
    __shared__ unsigned char site_of[PL_CAP];                // the site each entry of the image belongs to
    __shared__ unsigned kp[SITES_PER_GROUP], loff[SITES_PER_GROUP], curP[SITES_PER_GROUP], curQ[SITES_PER_GROUP], rk[SITES_PER_GROUP], qb[SITES_PER_GROUP];
    __shared__ unsigned short queue[PL_CHUNK];
    __shared__ unsigned wtot[PL_THREADS / 64];
    const size_t g = blockIdx.x;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint4 q4 = mb.minor_mask[g];
    if ((q4.x | q4.y | q4.z | q4.w) == 0u) return;
    const unsigned long long baseP = mb.baseP[g];
    if (!p_lists_in_lds(mb.gP[g])) return;                   // (site_lists_kernel builds this group's)
    if (tid < SITES_PER_GROUP) {
        const bool mine = (word_of(q4, (int)(tid >> 5)) >> (tid & 31u)) & 1u;
        kp[tid] = mine ? mb.cntP[g * SITES_PER_GROUP + tid] : 0u;
        curP[tid] = 0; curQ[tid] = 0;
    }
    __syncthreads();
    if (tid < SITES_PER_GROUP) {
        unsigned pp = 0, pq = 0;
        for (unsigned t = 0; t < tid; t++) { pp += kp[t]; pq += kp[t] / 31u; }
        loff[tid] = pp;
        rk[tid] = lst_rank(mb.lst_mask[g], mb.off_lst[g], (int)(tid >> 5), (int)(tid & 31u));
        qb[tid] = (unsigned)(mb.sites + mb.baseQ[g] + pq);
    }
    const uint4 RX = mb.ref_x[g], RY = mb.ref_y[g];
    const uint4 *base = mb.planes + (g * NPLANES) * n_pad;
    const unsigned mp[4] = {q4.x, q4.y, q4.z, q4.w};
#if defined(TRACS_PL_CUT) && TRACS_PL_CUT == 3
    for (unsigned c0 = 0; c0 < 0; c0 += PL_CHUNK) {
#else
    for (unsigned c0 = 0; c0 < n; c0 += PL_CHUNK) {
#endif
        // the chunk's flagged samples: thread t looks at samples c0 + PL_PER_THREAD t .. (8 of them)
        const unsigned s0 = c0 + PL_PER_THREAD * tid;
        unsigned bits = s0 < n ? (unsigned)(mb.flags[g * mb.flag_words + (s0 >> 6)] >> (s0 & 63u)) & ((1u << PL_PER_THREAD) - 1u) : 0u;
        const unsigned cnt = (unsigned)__popc(bits);
        unsigned x = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(x, off, 64); if ((int)lane >= off) x += o; }
        if (lane == 63u) wtot[wave] = x;
        __syncthreads();                                     // (also: the sites' offsets, the previous chunk's queue worked off)
        unsigned pos = x - cnt, qcount = 0;
#pragma unroll
        for (unsigned w = 0; w < PL_THREADS / 64u; w++) { const unsigned t = wtot[w]; if (w < wave) pos += t; qcount += t; }
        while (bits) { const unsigned b = (unsigned)__ffs(bits) - 1u; bits &= bits - 1u; queue[pos++] = (unsigned short)(PL_PER_THREAD * tid + b); }
        __syncthreads();
        for (unsigned k0 = 0; k0 < qcount; k0 += 2u * PL_THREADS) {
            // two samples per thread and step: eight 16-byte loads under way
            const unsigned ka = k0 + tid, kb = k0 + PL_THREADS + tid;
            const bool ha = ka < qcount, hb = kb < qcount;
            const unsigned sa = c0 + (ha ? queue[ka] : 0u), sb = c0 + (hb ? queue[kb] : 0u);
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            // (the stored N plane is A & C & G & T by construction: four loads, not five)
            const uint4 Aa = ha ? base[sa] : z, Ca = ha ? base[n_pad + sa] : z, Ga = ha ? base[2 * n_pad + sa] : z, Ta = ha ? base[3 * n_pad + sa] : z;
            const uint4 Ab = hb ? base[sb] : z, Cb = hb ? base[n_pad + sb] : z, Gb = hb ? base[2 * n_pad + sb] : z, Tb = hb ? base[3 * n_pad + sb] : z;
            auto sample = [&](unsigned s, const uint4 &A, const uint4 &C, const uint4 &G, const uint4 &T) {
                unsigned listed_w = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const unsigned a = word_of(A, w), c = word_of(C, w), gg = word_of(G, w), t = word_of(T, w), isn = a & c & gg & t;
                    const unsigned rx = word_of(RX, w), ry = word_of(RY, w);
                    const unsigned ra = ~rx & ~ry, rc = rx & ~ry, rg = ~rx & ry, rt = rx & ry;
                    const unsigned has_ref = (a & ra) | (c & rc) | (gg & rg) | (t & rt);
                    const unsigned only_ref = ~((a ^ ra) | (c ^ rc) | (gg ^ rg) | (t ^ rt));
                    // (the classification's predicate exactly: carries some allele bit, is not N, is not exactly the reference base)
                    unsigned pm = (a | c | gg | t) & ~isn & ~only_ref & mp[w];
                    while (pm) {
                        const unsigned b = (unsigned)__ffs(pm) - 1u;
                        pm &= pm - 1u;
                        const unsigned mask = ((a >> b) & 1u) | (((c >> b) & 1u) << 1) | (((gg >> b) & 1u) << 2) | (((t >> b) & 1u) << 3);
                        const unsigned code = (((has_ref >> b) & 1u) ? 0u : 16u) | mask;
                        const unsigned st = (unsigned)w * 32u + b;
                        const unsigned slot = (code & 16u) ? atomicAdd(&curP[st], 1u) : kp[st] - 1u - atomicAdd(&curQ[st], 1u);
                        sorted[loff[st] + slot] = (s << ENT_SHIFT) | code;
                        site_of[loff[st] + slot] = (unsigned char)st;
                        listed_w += code >> 4;
                    }
                }
#ifndef TRACS_PL_NO_CP
                if (listed_w) atomicAdd(&c_p[s], listed_w);
#endif
            };
#if defined(TRACS_PL_CUT) && TRACS_PL_CUT == 2
            if ((Aa.x ^ Ca.y ^ Ga.z ^ Ta.w ^ Ab.x ^ Cb.y ^ Gb.z ^ Tb.w) == 0x12345679u) c_p[0] = 1u;      // (timing build: the loads stay)
#else
            if (ha) sample(sa, Aa, Ca, Ga, Ta);
            if (hb) sample(sb, Ab, Cb, Gb, Tb);
#endif
        }
    }
#if defined(TRACS_PL_CUT) && (TRACS_PL_CUT == 1 || TRACS_PL_CUT == 2)
    return;
#endif
    __syncthreads();
    // the image leaves: E -- the group's w = 1 entries closed up at the front of its run (e_cnt[g] of them; `holes`: by list position,
    // holes for the others, what the one-pass fill of small alignments reads) --, the short lists into p_ent
    if (tid < 64u) {
        const unsigned a = curP[tid], b = curP[64 + tid];
        unsigned xa = a, xb = b;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned oa = __shfl_up(xa, off, 64), ob = __shfl_up(xb, off, 64);
            if ((int)tid >= off) { xa += oa; xb += ob; }
        }
        const unsigned ta = __shfl(xa, 63, 64), tb = __shfl(xb, 63, 64);
        w1off[tid] = xa - a; w1off[64 + tid] = ta + xb - b;
        if (tid == 0) e_cnt[g] = holes ? 0xFFFFFFFFu : ta + tb;
    }
    __syncthreads();
    const unsigned total = mb.gP[g];
    for (unsigned i = tid; i < total; i += PL_THREADS) {
        const unsigned ent = sorted[i], st = site_of[i];
        const bool is_long = kp[st] > P_SHORT_MAX;
        const uint2 ev = make_uint2(ent >> ENT_SHIFT, (is_long ? ENT_LONG : 0u) | (rk[st] << ENT_SHIFT) | (ent & 31u));
        if (holes) E[baseP + i] = (ent & 16u) ? ev : make_uint2(ENT_HOLE, 0u);
        else if (ent & 16u) E[baseP + w1off[st] + (i - loff[st])] = ev;
        if (!is_long) p_ent[baseP + i] = ent;
    }
    // q lines: qw dwords = 128 or 256 bytes per store, half a wave or a wave per line
    const unsigned qw = mb.qw, per = qw - 1u;
    const unsigned hw = tid / qw, l = tid % qw;
    for (unsigned st = hw; st < SITES_PER_GROUP; st += PL_THREADS / qw) {
        const unsigned k = kp[st];
        if (k <= P_SHORT_MAX) continue;
        const unsigned lo = loff[st], lines_n = k / per + 1u;
        for (unsigned j = 0; j < lines_n; j++) {
            const unsigned qs = per * j + l;                 // dword of the list (0: the header; entry i at dword i + 1)
            unsigned v = 0u;
            if (l == per) v = j == 0u ? qb[st] : 0u;
            else if (qs == 0u) v = k | (curP[st] << 16);
            else if (qs <= k) v = sorted[lo + qs - 1u];
            qd[(size_t)(j ? qb[st] + j - 1u : rk[st]) * qw + l] = v;
        }
    }
}

// ---- per sample: listed entries (from E) ---------------------------------------------------------------------------------------
// (their number per sample, c_p[s], comes from the per-site pass.)  The entries of E are placed through per-sample cursors.
__global__ __launch_bounds__(256) void listed_entries_kernel(const uint2 *__restrict__ E, unsigned long long count,
                                                             const unsigned long long *__restrict__ off, unsigned *__restrict__ cur,
                                                             unsigned *__restrict__ ent)
{
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const uint2 e = E[k];
    if (e.x == ENT_HOLE) return;
    ent[off[e.x] + atomicAdd(&cur[e.x], 1u)] = e.y;
}

// ---- the same in two bucketing passes --------------------------------------------------------------------------------------------
// E is site-major, the per-sample lists are sample-major: placed one entry at a time (above) that is tot_p scattered 4-byte writes
// behind tot_p returning atomics -- 14.6 ms for the 250 M entries of an alignment with 0.5 % partial codes.  A sample's run in s_ent
// is known once the counts are scanned (s_off), so the entries are bucketed twice, a tile of 8 192 at a time with its counters in LDS:
//   pass A   by sample >> shift (at most 256 buckets: bucket b's run is [s_off[b << shift], s_off[(b + 1) << shift]), into tmp;
//   pass B   within a bucket by sample (at most 1 024 samples), into s_ent.
// A tile reserves its share of every run it touches with one global atomic per run, and its writes into a run are consecutive.
constexpr unsigned ENT_TILE_THREADS = 1024, ENT_PER_THREAD = 8, ENT_TILE = ENT_TILE_THREADS * ENT_PER_THREAD;
constexpr unsigned ENT_GROUPS = 64;            // groups per workgroup of the first pass
__global__ __launch_bounds__(ENT_TILE_THREADS) void entries_to_buckets_kernel(const uint2 *__restrict__ E, const unsigned *__restrict__ e_cnt,
                                                                              const unsigned *__restrict__ gP, const unsigned long long *__restrict__ baseP,
                                                                              size_t groups, unsigned shift, unsigned n,
                                                                              const unsigned long long *__restrict__ s_off,
                                                                              unsigned *__restrict__ bcur, uint2 *__restrict__ tmp)
{
    // A workgroup takes ENT_GROUPS consecutive groups: group g's entries are E[baseP[g] .. + min(e_cnt[g], gP[g])) -- closed up by
    // p_lists_kernel, or all of its list positions (holes among them) --; the concatenation is worked off a tile at a time.
    __shared__ unsigned hist[256], base[256];
    __shared__ unsigned pre[ENT_GROUPS + 1];
    __shared__ unsigned long long gbase[ENT_GROUPS];
    const unsigned tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * ENT_GROUPS;
    if (tid < 64u) {
        const size_t g = g0 + tid;
        const unsigned c = g < groups ? min(e_cnt[g], gP[g]) : 0u;      // (a group without p lists: gP = 0, e_cnt never written)
        unsigned x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(x, off, 64); if ((int)tid >= off) x += o; }
        pre[tid + 1] = x;
        if (tid == 0) pre[0] = 0;
        gbase[tid] = g < groups ? baseP[g] : 0ull;
    }
    __syncthreads();
    const unsigned total = pre[ENT_GROUPS];
    for (unsigned t0 = 0; t0 < total; t0 += ENT_TILE) {
        if (tid < 256u) hist[tid] = 0;
        __syncthreads();
        uint2 e[ENT_PER_THREAD];
        unsigned slot[ENT_PER_THREAD];
#pragma unroll
        for (int q = 0; q < (int)ENT_PER_THREAD; q++) {
            const unsigned k = t0 + (unsigned)q * ENT_TILE_THREADS + tid;
            e[q] = make_uint2(ENT_HOLE, 0u);
            if (k < total) {
                unsigned lo = 0, hi = ENT_GROUPS;            // the group of entry k: pre[lo] <= k < pre[lo + 1]
#pragma unroll
                for (int st = 0; st < 6; st++) { const unsigned mid = (lo + hi) >> 1; if (pre[mid] <= k) lo = mid; else hi = mid; }
                e[q] = E[gbase[lo] + (k - pre[lo])];
            }
            slot[q] = e[q].x != ENT_HOLE ? atomicAdd(&hist[e[q].x >> shift], 1u) : 0u;      // (holes: the p lists' entries with w = 0)
        }
        __syncthreads();
        if (tid < 256u && hist[tid]) base[tid] = atomicAdd(&bcur[tid], hist[tid]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (int)ENT_PER_THREAD; q++) {
            if (e[q].x == ENT_HOLE) continue;
            const unsigned b = e[q].x >> shift;
            tmp[s_off[min((unsigned long long)n, (unsigned long long)b << shift)] + base[b] + slot[q]] = e[q];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(ENT_TILE_THREADS) void buckets_to_samples_kernel(const uint2 *__restrict__ tmp, unsigned shift, unsigned n,
                                                                              const unsigned long long *__restrict__ s_off, unsigned *__restrict__ cur,
                                                                              unsigned *__restrict__ ent)
{
    __shared__ unsigned hist[1024], base[1024];
    const unsigned tid = threadIdx.x;
    const unsigned long long s_lo = (unsigned long long)blockIdx.y << shift;
    if (s_lo >= n) return;
    const unsigned bins = (unsigned)min((unsigned long long)n - s_lo, 1ull << shift);
    const unsigned long long r0 = s_off[s_lo], r1 = s_off[s_lo + bins];
    for (unsigned long long t0 = r0 + (unsigned long long)blockIdx.x * ENT_TILE; t0 < r1; t0 += (unsigned long long)gridDim.x * ENT_TILE) {
        if (tid < bins) hist[tid] = 0;
        __syncthreads();
        uint2 e[ENT_PER_THREAD];
        unsigned slot[ENT_PER_THREAD];
#pragma unroll
        for (int q = 0; q < (int)ENT_PER_THREAD; q++) {
            const unsigned long long k = t0 + (unsigned long long)q * ENT_TILE_THREADS + tid;
            e[q] = k < r1 ? tmp[k] : make_uint2((unsigned)s_lo, 0u);
            slot[q] = k < r1 ? atomicAdd(&hist[e[q].x - (unsigned)s_lo], 1u) : 0u;
        }
        __syncthreads();
        if (tid < bins && hist[tid]) base[tid] = atomicAdd(&cur[s_lo + tid], hist[tid]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (int)ENT_PER_THREAD; q++) {
            const unsigned long long k = t0 + (unsigned long long)q * ENT_TILE_THREADS + tid;
            if (k >= r1) continue;
            const unsigned key = e[q].x - (unsigned)s_lo;
            ent[s_off[e[q].x] + base[key] + slot[q]] = e[q].y;
        }
        __syncthreads();
    }
}

// exclusive scan of v[0 .. count) -> out[0 .. count] (one workgroup; wave scans through shuffles); the largest element -> *vmax
__global__ __launch_bounds__(1024) void scan_counts_kernel(const unsigned *__restrict__ v, size_t count, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RUN = 8;
    unsigned long long carry = 0;                            // (the same on every thread)
    for (size_t base = 0; base <= count; base += 1024 * RUN) {
        const size_t b0 = base + (size_t)threadIdx.x * RUN;
        unsigned long long local[RUN], sum = 0;
#pragma unroll
        for (int k = 0; k < RUN; k++) { local[k] = sum; sum += (b0 + k < count) ? v[b0 + k] : 0u; }
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
        for (int k = 0; k < RUN; k++) if (b0 + k <= count) out[b0 + k] = excl + local[k];
        carry += all;
        __syncthreads();
    }
}

// ---- per sample: its N bitmap over the NNL sites (the N plane transposed), and its N count ---------------------------------------
// A wave takes 8 samples x 8 consecutive groups at a time: lane = (sample s0 + lane / 8, group g0 + lane % 8).  Its load touches, per
// group, the 128 contiguous bytes of the 8 samples; its store, per sample, the 128 contiguous bytes of the 8 groups: whole cache
// lines on both sides of the transposition, no LDS.  c_counted[s] += the sample's N sites among the sites of un_mask (what the
// compared-sites formula needs), summed over the wave's octets in registers, then over the 8 lanes of a sample.
#ifndef TRACS_NB_GW
#define TRACS_NB_GW 8
#endif
constexpr unsigned NB_GW = TRACS_NB_GW, NB_SW = 64u / NB_GW;      // groups x samples of a wave's tile (8 x 8; 4 x 16 / 16 x 4: timing builds)
static_assert(NB_GW == 4 || NB_GW == 8 || NB_GW == 16, "n_bitmap_kernel: 4, 8 or 16 groups per wave");
__global__ __launch_bounds__(256) void n_bitmap_kernel(const MinorBuild mb, size_t n_pad, unsigned n, size_t groups, size_t tgroups,
                                                       size_t oct_per_chunk, uint4 *__restrict__ T, unsigned *__restrict__ c_counted)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t s = ((size_t)blockIdx.x * 4 + wave) * NB_SW + (lane / NB_GW);
    const size_t o0 = (size_t)blockIdx.y * oct_per_chunk, o1 = min(tgroups / NB_GW, o0 + oct_per_chunk);
    const bool live = s < n;
    const bool row = live && T != nullptr && row_wanted(mb, s);
    const uint4 *np = mb.planes + 4 * n_pad + min(s, n_pad - 1);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    unsigned cnt = 0;
#ifndef TRACS_NB_INF
#define TRACS_NB_INF 4
#endif
    constexpr int INF = TRACS_NB_INF;                        // octets in flight per thread (64 bytes: the loop is bound by what is in flight)
    for (size_t o = o0; o < o1; o += INF) {
        uint4 v[INF];
#pragma unroll
        for (int k = 0; k < INF; k++) {
            const size_t g = (o + k) * NB_GW + (lane % NB_GW);
            v[k] = make_uint4(0u, 0u, 0u, 0u);
            if (o + k < o1 && g < groups && live) v[k] = np[g * NPLANES * n_pad];
        }
#pragma unroll
        for (int k = 0; k < INF; k++) {
            const size_t g = (o + k) * NB_GW + (lane % NB_GW);
            const size_t gm = min(g, groups - 1);
            const uint4 um = mb.un_mask[gm], lm = mb.nnl_mask[gm];
            cnt += __popc(v[k].x & um.x) + __popc(v[k].y & um.y) + __popc(v[k].z & um.z) + __popc(v[k].w & um.w);      // (zero past o1)
            if (row && o + k < o1) T[s * tgroups + g] = g < groups ? make_uint4(v[k].x & lm.x, v[k].y & lm.y, v[k].z & lm.z, v[k].w & lm.w) : zero4;
        }
    }
#pragma unroll
    for (unsigned off = 1; off < NB_GW; off <<= 1) cnt += __shfl_xor(cnt, off, 64);
    if (live && (lane % NB_GW) == 0u && cnt) atomicAdd(&c_counted[s], cnt);
}

__global__ void max_count_kernel(const unsigned *__restrict__ c, size_t n, unsigned *__restrict__ out)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) atomicMax(out, c[s]);
}

// ---- the walk of n8 lines ---------------------------------------------------------------------------------------------------------
// A wave keeps a ring of LINES in LDS (index, position before the line's first byte): the sites of the row's bitmap, and the lines
// that go on behind them.  A round takes 16 lines, four lanes per line, two 16-byte pieces per lane:
//   scan    byte sums per piece (v_sad_u8), inclusive prefixes over the line's four lanes (DPP) -- now every lane knows the positions
//           its pieces span.  A piece that ends below the row's cut (sorted lists: row i needs j > i) or holds padding only is left
//           alone: on a row in the middle of the matrix half of them.  A line that goes on (its `next`) is queued as a line;
//   decode  the pieces that count, where they are (exec mask): per byte one SDWA add (position), one shift-add (LDS byte address),
//           one SDWA compare (skip and padding bytes sit the add out), one ds_add_u32.
// Forms that were slower (profiles/r04/nn_rows_n8.txt): every piece decoded with skips and padding added to dump slots (LDS pipe);
// the pieces that count queued and decoded 64 at a time with every lane busy -- fewer instructions, but a second load per piece,
// 64 different cache lines per instruction through the CU's L1; two scan rounds in flight per wave (registers: half the waves).
// a 16-byte load of data read once (a row's bitmap): non-temporal, so that it does not push the lines of the current segment --
// read a hundred times over -- out of the Infinity Cache
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream(const uint4 *p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
typedef __attribute__((address_space(3))) unsigned lds_u32;
constexpr unsigned LINE_RING = 128;
constexpr unsigned WALK_LDS_PER_WAVE = LINE_RING * 8;

// CLAMP = false: the row's counters cover every sample (one column chunk: n <= 36 800): a decoded position needs no range check.
template <bool CLAMP>
struct Walk {
    const uint4 *lines;
    uint2 *lring;                              // this wave's ring of lines (LDS): index, position before the line's first byte (-1 at a
                                               //   site's first line)
    unsigned lhead, lcount;                    // (wave-uniform)
    unsigned lane, grp, l4;                    // the scan's view of the wave: 16 lines x 4 lanes
    unsigned neg4lo, dump4, val;               // row[] starts at LDS byte 0: counter of column j at 4 (j - c0); the lane's dump slot
    unsigned cut;                              // a piece that ends below this position has nothing to add
    bool lt3, lt2;

    __device__ __forceinline__ void init(const uint4 *lines_, unsigned *lds, unsigned lane_)
    {
        lines = lines_;
        lring = reinterpret_cast<uint2 *>(lds);
        lhead = lcount = 0;
        lane = lane_; grp = lane_ >> 2; l4 = lane_ & 3u;
        lt3 = l4 < 3u; lt2 = l4 < 2u;
        // (a scan round reads 16 slots whatever the count: every slot holds a line that exists)
        lring[lane] = make_uint2(0u, 0u); lring[64 + lane] = make_uint2(0u, 0u);
    }
    __device__ __forceinline__ unsigned rank_of(bool has, unsigned long long &m) const
    {
        m = __ballot(has);
        return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    }
    __device__ __forceinline__ void push_line(bool has, unsigned line)
    {
        unsigned long long m;
        const unsigned r = rank_of(has, m);
        if (has) lring[(lhead + lcount + r) & (LINE_RING - 1u)] = make_uint2(line, 0xFFFFFFFFu);
        lcount += (unsigned)__popcll(m);
    }
    __device__ __forceinline__ void sync_wave() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // one 16-byte piece: per byte the position (SDWA add), the LDS byte address (shift-add), the add -- sat out by skip and padding bytes
    __device__ __forceinline__ void decode_piece(unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned p) const
    {
        const unsigned w[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned b = (w[q4] >> (8 * q)) & 0xFFu;
                p += b;
                unsigned a = (p << 2) + neg4lo;
                if (CLAMP) a = min(a, dump4);
                if (b < N8_SKIP) __hip_atomic_fetch_add(reinterpret_cast<lds_u32 *>((size_t)a), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __device__ __forceinline__ void scan_round()
    {
        // 16 lines, FOUR lanes per line, two pieces per lane (k4 and 4 + k4: each load instruction reads 64 contiguous bytes of a
        // line): one ring read, one address, one follow-on step per round where eight lanes per line took two of each
        const unsigned k = min(lcount, 16u);
        const uint2 ref = lring[(lhead + grp) & (LINE_RING - 1u)];
        const uint4 *lp = lines + (size_t)ref.x * 8 + l4;
        const uint4 d0 = lp[0], d1 = lp[4];
        __builtin_amdgcn_sched_barrier(0);                  // (both loads on their way before anything waits for the first)
        lhead = (lhead + k) & (LINE_RING - 1u); lcount -= k;
        const bool has = grp < k;
        // (w3 of the line's last piece is the line's `next`, not payload)
        const unsigned S0 = __builtin_amdgcn_sad_u8(d0.x, 0u, __builtin_amdgcn_sad_u8(d0.y, 0u, __builtin_amdgcn_sad_u8(d0.z, 0u, __builtin_amdgcn_sad_u8(d0.w, 0u, 0u))));
        const unsigned S1 = __builtin_amdgcn_sad_u8(d1.x, 0u, __builtin_amdgcn_sad_u8(d1.y, 0u, __builtin_amdgcn_sad_u8(d1.z, 0u, __builtin_amdgcn_sad_u8(lt3 ? d1.w : 0u, 0u, 0u))));
        // inclusive prefixes over the line's four lanes (what a step hands on is zeroed where it would cross into the next line);
        // the second four pieces start behind the first four: the total of those, from the line's last lane
        unsigned x0 = S0, x1 = S1;
        x0 += (unsigned)__builtin_amdgcn_update_dpp(0, (int)(lt3 ? x0 : 0u), 0x111, 0xF, 0xF, true);      // row_shr:1
        x1 += (unsigned)__builtin_amdgcn_update_dpp(0, (int)(lt3 ? x1 : 0u), 0x111, 0xF, 0xF, true);
        x0 += (unsigned)__builtin_amdgcn_update_dpp(0, (int)(lt2 ? x0 : 0u), 0x112, 0xF, 0xF, true);      // row_shr:2
        x1 += (unsigned)__builtin_amdgcn_update_dpp(0, (int)(lt2 ? x1 : 0u), 0x112, 0xF, 0xF, true);
        const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x0, 0xFF, 0xF, 0xF, false);      // quad_perm:[3,3,3,3]
        const unsigned e0 = ref.y + x0, e1 = ref.y + t0 + x1;          // the positions behind this lane's two pieces
        const int icut = (int)cut;
        const bool w0 = has && (int)e0 >= icut && (d0.x & 0xFFu) != 0xFFu;
        const bool w1 = has && (int)e1 >= icut && (d1.x & 0xFFu) != 0xFFu;
        // the pieces are decoded where they are, by the lanes whose piece counts (exec mask).  Queueing the pieces that count and
        // decoding 64 of them per round with every lane busy (the round's first half) cost a second load per piece -- 64 different
        // cache lines per instruction through the CU's L1: 12.4 ms against 10.6 (profiles/r04/nn_rows_n8.txt)
        if (w0) decode_piece(d0.x, d0.y, d0.z, d0.w, e0 - S0);
        if (w1) decode_piece(d1.x, d1.y, d1.z, lt3 ? d1.w : 0xFFFFFFFFu, e1 - S1);
        unsigned long long m;
        unsigned r;
        const bool goes_on = has && !lt3 && d1.w != N8_NONE;
        r = rank_of(goes_on, m);
        if (goes_on) lring[(lhead + lcount + r) & (LINE_RING - 1u)] = make_uint2(d1.w, e1);
        lcount += (unsigned)__popcll(m);
    }
    // rounds while more than `keep` lines wait
    __device__ __forceinline__ void drain_lines_to(unsigned keep)
    {
        while (lcount > keep) {
            sync_wave();
            scan_round();
        }
    }
    __device__ __forceinline__ void finish() { drain_lines_to(0); }
};

// ---- N co-occurrences from lists (the NNL sites) ----------------------------------------------------------------------------------
// Row i of the pair matrix: NN(i, j) = number of NNL sites at which both i and j are N.  The row's counters live in LDS; a wave
// takes 64 groups of sample i's bitmap at a time (16 bytes per lane), every set bit is a site whose line goes into the wave's ring
// (slots from a prefix scan of the lanes' bit counts), and the ring is worked off 16 lines at a time (Walk).  Work = sum over the NNL
// sites of cN walks of the site's line(s), whatever the number of samples -- against n^2 / 2 pairs per site on the matrix cores.
// A sample with many N sites would leave most of the chip idle behind a few rows: a row's groups are cut over up to NN_MAX_SPLITS
// workgroups (grid.z; a row with few N sites uses one and the others exit at once), which then add their rows with atomics.
constexpr unsigned NN_MAX_SPLITS = 32;
#ifndef TRACS_NN_THREADS
#define TRACS_NN_THREADS 1024
#endif
template <bool CLAMP>
__global__ __launch_bounds__(TRACS_NN_THREADS) void nn_rows_kernel(const uint4 *__restrict__ T, size_t tgroups, const uint4 *__restrict__ lst_mask,
                                                                   const unsigned *__restrict__ off_lst, size_t groups, const uint4 *__restrict__ lines,
                                                                   const unsigned *__restrict__ c_u, unsigned n, unsigned row_begin, unsigned col_begin,
                                                                   unsigned chunk, unsigned target, unsigned segments, unsigned *__restrict__ ncomp,
                                                                   size_t ld, int add_terms, unsigned lu)
{
    // `chunk` counters -- row[0] is column c0 --, 64 slots nobody reads, the waves' rings
    extern __shared__ unsigned row[];
    const unsigned i = row_begin + blockIdx.x;
    const unsigned c0 = blockIdx.y * chunk, c1 = min(n, c0 + chunk);
    if (c1 <= i + 1 || c1 <= col_begin) return;            // no cell (i, j > i) in this column chunk
    const unsigned ci = c_u[i];
    const unsigned nz = min(NN_MAX_SPLITS, max(segments, (ci + target - 1u) / target));
    if (blockIdx.z >= nz) return;
    const unsigned lo = max(max(i + 1, col_begin), c0);     // columns [lo, c1) of this chunk are cells of row i
    const unsigned span = c1 - c0;
    if ((unsigned)(size_t)row != 0u) __builtin_trap();      // (the walk's LDS adds address row[] from 0)
    for (unsigned j = threadIdx.x; j < span + 64u; j += blockDim.x) row[j] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    Walk<CLAMP> W;
    W.init(lines, row + chunk + 64 + wave * (WALK_LDS_PER_WAVE / 4), lane);
    W.neg4lo = 0u - 4u * c0; W.dump4 = 4u * (span + lane); W.val = 1u; W.cut = lo;
    const size_t batches = (tgroups + 63) / 64;
    const size_t per = (batches + nz - 1) / nz;
    const size_t b_first = (size_t)blockIdx.z * per, b_last = min(batches, b_first + per);
    const uint4 *Trow = T + (size_t)i * tgroups;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    size_t b = b_first + wave;
    uint4 tw_next = (b < b_last && b * 64 + lane < tgroups) ? load_stream(Trow + b * 64 + lane) : zero4;
    for (; b < b_last; b += nwaves) {
        const size_t g = b * 64 + lane;
        const uint4 tw = tw_next;
        const size_t bn = b + nwaves;
        tw_next = (bn < b_last && bn * 64 + lane < tgroups) ? load_stream(Trow + bn * 64 + lane) : zero4;     // the next batch's bitmap: in flight during this one
        unsigned r0 = tw.x, r1 = tw.y, r2 = tw.z, r3 = tw.w;
        // this lane's set bits, and where its items go in the ring: an inclusive wave scan of the counts (DPP)
        const unsigned cnt = __popc(r0) + __popc(r1) + __popc(r2) + __popc(r3);
        unsigned x = cnt;
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);      // row_shr:1 .. 8: scan inside the rows of 16
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
        x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
        if (total == 0u) continue;
        const size_t gm = min(g, groups - 1);
        const uint4 lm = lst_mask[gm];
        const unsigned og = off_lst[gm];
        const unsigned pre1 = og + __popc(lm.x), pre2 = pre1 + __popc(lm.y), pre3 = pre2 + __popc(lm.z);
        W.drain_lines_to(31);
        if (total <= LINE_RING - 32u) {
            // (the usual case: all of the batch's sites fit the ring at once -- every lane drops its own, word by word)
            unsigned pos = W.lhead + W.lcount + x - cnt;
            while (r0) { const unsigned bit = __ffs(r0) - 1; r0 &= r0 - 1; W.lring[pos++ & (LINE_RING - 1u)] = make_uint2(og + __popc(lm.x & ((1u << bit) - 1u)), 0xFFFFFFFFu); }
            while (r1) { const unsigned bit = __ffs(r1) - 1; r1 &= r1 - 1; W.lring[pos++ & (LINE_RING - 1u)] = make_uint2(pre1 + __popc(lm.y & ((1u << bit) - 1u)), 0xFFFFFFFFu); }
            while (r2) { const unsigned bit = __ffs(r2) - 1; r2 &= r2 - 1; W.lring[pos++ & (LINE_RING - 1u)] = make_uint2(pre2 + __popc(lm.z & ((1u << bit) - 1u)), 0xFFFFFFFFu); }
            while (r3) { const unsigned bit = __ffs(r3) - 1; r3 &= r3 - 1; W.lring[pos++ & (LINE_RING - 1u)] = make_uint2(pre3 + __popc(lm.w & ((1u << bit) - 1u)), 0xFFFFFFFFu); }
            W.lcount += total;
            continue;
        }
        for (;;) {                                            // a sample that is N nearly everywhere: one bit per lane and step
            W.drain_lines_to(31);                            // room for 64 more
            const bool has = (r0 | r1 | r2 | r3) != 0u;
            if (!__ballot(has)) break;
            unsigned rank = 0;
            if (r0) { const unsigned bit = __ffs(r0) - 1; r0 &= r0 - 1; rank = og + __popc(lm.x & ((1u << bit) - 1u)); }
            else if (r1) { const unsigned bit = __ffs(r1) - 1; r1 &= r1 - 1; rank = pre1 + __popc(lm.y & ((1u << bit) - 1u)); }
            else if (r2) { const unsigned bit = __ffs(r2) - 1; r2 &= r2 - 1; rank = pre2 + __popc(lm.z & ((1u << bit) - 1u)); }
            else if (r3) { const unsigned bit = __ffs(r3) - 1; r3 &= r3 - 1; rank = pre3 + __popc(lm.w & ((1u << bit) - 1u)); }
            W.push_line(has, rank);
        }
    }
    W.finish();
    __syncthreads();
    const bool terms = add_terms && blockIdx.z == 0;
    for (unsigned j = lo + threadIdx.x; j < c1; j += blockDim.x) {
        const unsigned v = row[j - c0] + (terms ? lu - ci - c_u[j] : 0u);
        if (v) {
            if (nz > 1) atomicAdd(&ncomp[(size_t)i * ld + j], v);
            else ncomp[(size_t)i * ld + j] += v;
        }
    }
}

// ---- the minority sites' distances --------------------------------------------------------------------------------------------------
// At a minority site every sample is N, or carries exactly the site's reference base (not listed), or is LISTED with its allele
// mask M and w = [reference base not in M].  Such a site adds to d(i, j): w_i when i is listed and j carries the reference base,
// [M_i n M_j = {}] when both are listed, 0 when either is N -- i.e. over the sites S_i, S_j at which i / j is listed
//     d += sum_{S_i} w_i + sum_{S_j} w_j - sum_{s in S_i: j is N} w_i - sum_{s in S_j: i is N} w_j
//          + sum over S_i n S_j of ([M_i n M_j = {}] - w_i - w_j)
// (consensus alignments: M = {own base}, w = 1).  The first two sums are per-sample constants (c_p).  Row x (one workgroup):
//     phase A  every listed entry of x with w = 1 walks its site's p list: the last sum -- for j > x (row x of dist) and, where w_j = 0,
//              for j < x too (cell (j, x)): the entries with w = 0 do not walk (two such never add anything: both masks hold the reference base);
//     phase B  every listed entry of x with w = 1 walks its site's N list: -1 for EVERY N sample y there -- the third sum for y > x
//              (row x of dist), the fourth for y < x (cell (y, x): a scratch row, folded in by transpose_add_kernel).
// Negative terms wrap in the unsigned row and cancel in the final sum.
// GRAM (site_classes.hip, nw_gram): the third and fourth sums come from the matrix cores as (U U^T - n n^T)(x, j) = w n^T + n w^T + w w^T
// over the minority sites -- phase B does not exist, and the both-listed term of phase A gives w_x w_j back: [masks disjoint] - w_x - w_j
// + w_x w_j = [masks disjoint] - 1 for the walker's w_x = 1.
// NSROWS (nw_rows): phase B without N lists -- sum_s w_x(s) n_y(s) over the sites x is listed at (w = 1) is the column sum of those
// sites' rows of the site-major N matrix NS: thread w of a group of threads owns the 32 samples of word w, adds the words of x's sites
// into bit-sliced counters in registers (the classification's carry-save tree: 3 instructions per word) and takes them off row[] every
// 248 sites.  The rows stream from HBM once per listed (sample, site): |W| x ns_words x 4 bytes -- where W is sparse (0.2 % of the cells
// on bench.py's coverage workload: 96 GB) a fraction of the U U^T pass it replaces.
template <bool CLAMP, unsigned QW, bool GRAM, bool NSROWS = false>
__global__ __launch_bounds__(1024) void minor_fixup_kernel(const unsigned long long *__restrict__ s_off, const unsigned *__restrict__ s_ent,
                                                           const unsigned long long *__restrict__ p_off, const unsigned *__restrict__ p_ent,
                                                           const unsigned *__restrict__ qd,
                                                           const uint4 *__restrict__ lines, const unsigned *__restrict__ c_p, unsigned n,
                                                           unsigned row_begin, unsigned row_end, unsigned col_begin, unsigned chunk,
                                                           unsigned *__restrict__ dist, size_t ld, unsigned *__restrict__ S, size_t s_pitch,
                                                           const unsigned *__restrict__ ns = nullptr, size_t ns_words = 0,
                                                           const unsigned *__restrict__ site_of = nullptr)
{
    extern __shared__ unsigned row[];
    const unsigned x = row_begin + blockIdx.x;
    const unsigned c0 = blockIdx.y * chunk, c1 = min(n, c0 + chunk);
    const bool in_rows = x < row_end;
    // cells this workgroup feeds: (x, y) for y in [max(x + 1, col_begin, c0), c1) when x is a row of the call; (y, x) for y in
    // [max(row_begin, c0), min(x, row_end, c1)) when column x is wanted
    const unsigned up0 = max(max(x + 1, col_begin), c0), up1 = in_rows ? c1 : 0u;
    const unsigned lw0 = max(row_begin, c0), lw1 = x >= col_begin ? min(min(x, row_end), c1) : 0u;
    const bool upper = up0 < up1, lower = lw0 < lw1;
    if (!upper && !lower) return;
    const unsigned long long e0 = s_off[x], e1 = s_off[x + 1];
    if ((unsigned)(size_t)row != 0u) __builtin_trap();
    const unsigned span = c1 - c0;
    for (unsigned j = threadIdx.x; j < span + 64u; j += blockDim.x) row[j] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if ((upper || lower) && e1 > e0) {
        // phase A.  Every entry of sample x (w_x = 1) is a walk of its site's p list: q line `rank` -- header (k, w1), 30 entries, the index
        // of the first overflow line -- and, for lists beyond 30, the consecutive overflow lines.  A wave keeps a ring of (line, what) in
        // LDS and works it off 16 lines at a time, FOUR lanes per line, 8 dwords per lane (the line is one 128-byte fetch, its header tells
        // how far to go).  The listed samples whose mask holds the reference base (w = 0: partial codes) do not walk: against each other
        // [masks disjoint] - w - w is 0, and their pairs with a w = 1 sample are that sample's -- in both triangles, like the N samples of
        // phase B (with 0.5 % partial codes 98 % of the entries are such: 253 M walks of a line each, 11 ms, are 5 M).
        const unsigned *__restrict__ q = qd;
        uint2 *ring = reinterpret_cast<uint2 *>(row + chunk + 64 + wave * (WALK_LDS_PER_WAVE / 4));
        unsigned rhead = 0, rcount = 0;
        const unsigned grp = lane >> 2, l4 = lane & 3u;
        ring[lane] = make_uint2(0u, 0u); ring[64 + lane] = make_uint2(0u, 0u);
        auto wave_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto push = [&](bool has, unsigned line, unsigned what) {
            const unsigned long long m = __ballot(has);
            const unsigned r = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (has) ring[(rhead + rcount + r) & (LINE_RING - 1u)] = make_uint2(line, what);
            rcount += (unsigned)__popcll(m);
        };
        // what: code of x (5 bits) | overflow line (bit 5) | entries still to read from here on (overflow lines)
        auto round = [&]() {
            const unsigned k16 = min(rcount, 16u);
            const uint2 ref = ring[(rhead + grp) & (LINE_RING - 1u)];
            // (a q line is QW dwords -- 128 or 256 bytes --, QW / 4 per lane)
            constexpr unsigned DPL = QW / 4u, PER = QW - 1u;
            const uint4 *lp = reinterpret_cast<const uint4 *>(q + (size_t)ref.x * QW) + l4 * (DPL / 4u);
            uint4 dq[DPL / 4u];
#pragma unroll
            for (unsigned t = 0; t < DPL / 4u; t++) dq[t] = lp[t];
            rhead = (rhead + k16) & (LINE_RING - 1u); rcount -= k16;
            const bool has = grp < k16;
            const unsigned code = ref.y & 31u;
            const bool ovf = (ref.y >> 5) & 1u;
            const unsigned hdr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)dq[0].x, 0x00, 0xF, 0xF, false);      // quad_perm:[0,0,0,0]
            const unsigned tail = (unsigned)__builtin_amdgcn_update_dpp(0, (int)dq[DPL / 4u - 1u].w, 0xFF, 0xF, 0xF, false);     // quad_perm:[3,3,3,3]
            const unsigned first = ovf ? 0u : 1u;
            const unsigned limit = ovf ? (ref.y >> 6) : (hdr & 0xFFFFu);
            unsigned v[DPL];
#pragma unroll
            for (unsigned t = 0; t < DPL / 4u; t++) { v[4 * t] = dq[t].x; v[4 * t + 1] = dq[t].y; v[4 * t + 2] = dq[t].z; v[4 * t + 3] = dq[t].w; }
            const unsigned mx = code & 15u;
#pragma unroll
            for (unsigned t = 0; t < DPL; t++) {
                const unsigned dw = l4 * DPL + t;
                const unsigned j = v[t] >> ENT_SHIFT;
                const unsigned wj = (v[t] >> 4) & 1u;
                // cell (x, j) for j > x; cell (j, x) for a j < x that does not walk itself (w_j = 0)
                const bool in = has && dw >= first && dw < PER && dw - first < limit &&
                                ((j >= up0 && j < up1) || (wj == 0u && j >= lw0 && j < lw1));
                const int add = (((v[t] & 15u) & mx) == 0u ? 1 : 0) - 1 - (GRAM ? 0 : (int)wj);      // both listed: [masks disjoint] - w_x - w_j (+ w_x w_j: GRAM)
                if (in && add != 0) atomicAdd(&row[j - c0], (unsigned)add);
            }
            const unsigned cap = PER - first;
            push(has && l4 == 0u && limit > cap, ovf ? ref.x + 1u : tail, code | 32u | ((limit - cap) << 6));
        };
        for (unsigned long long base = e0 + (unsigned long long)wave * 64; base < e1; base += (unsigned long long)nwaves * 64) {
            const unsigned long long e = base + lane;
            const unsigned ent = e < e1 ? s_ent[e] : 0u;
            const bool is_long = (ent & ENT_LONG) != 0u;
            const unsigned rank = (ent & ~ENT_LONG) >> ENT_SHIFT;
            if (e < e1 && !is_long) {
                // a list of at most P_SHORT_MAX samples (the usual case: one or two) stays in the lane that found it
                const unsigned long long pa = p_off[rank], pz = p_off[rank + 1];
                const unsigned mx = ent & 15u;
                unsigned v[P_SHORT_MAX];
#pragma unroll
                for (unsigned m = 0; m < P_SHORT_MAX; m++) v[m] = pa + m < pz ? p_ent[pa + m] : 0xFFFFFFFFu;
#pragma unroll
                for (unsigned m = 0; m < P_SHORT_MAX; m++) {
                    const unsigned j = v[m] >> ENT_SHIFT;
                    const unsigned wj = (v[m] >> 4) & 1u;
                    const int add = (((v[m] & 15u) & mx) == 0u ? 1 : 0) - 1 - (GRAM ? 0 : (int)wj);
                    if (v[m] != 0xFFFFFFFFu && add != 0 && ((j >= up0 && j < up1) || (wj == 0u && j >= lw0 && j < lw1))) atomicAdd(&row[j - c0], (unsigned)add);
                }
            }
            while (rcount > LINE_RING - 64u - 16u) { wave_sync(); round(); }
            push(e < e1 && is_long, rank, ent & 31u);
        }
        while (rcount) { wave_sync(); round(); }
        wave_sync();
    }
    // phase B: N-list walks, both triangles: column y goes to row[y - c0]
    if constexpr (NSROWS) {
        __syncthreads();                                       // (phase A's atomics on row[] are done: the flush below reads and writes it plainly per column)
        const unsigned w0 = c0 >> 5, w1 = (c1 + 31u) >> 5;     // the words of this chunk's columns
        // groups of threads share the sites of x; a chunk of more words than the workgroup has threads takes several passes
        const unsigned tpg = min((w1 - w0 + 63u) / 64u * 64u, blockDim.x), ngrp = max(1u, blockDim.x / tpg);
        const unsigned grp = threadIdx.x / tpg;
        for (unsigned wb = w0; wb < w1; wb += tpg) {
            const unsigned wi = wb + threadIdx.x % tpg;
            const bool active = grp < ngrp && wi < w1 && (upper || lower);
            unsigned pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            unsigned since = 0;
            auto flush = [&]() {
#pragma unroll 4
                for (unsigned b = 0; b < 32u; b++) {
                    unsigned c = 0;
#pragma unroll
                    for (int jx = 0; jx < 8; jx++) c |= ((pl[jx] >> b) & 1u) << jx;
                    const unsigned y = wi * 32u + b;
                    if (c && y >= c0 && y < c1) atomicAdd(&row[y - c0], 0u - c);
                }
#pragma unroll
                for (int jx = 0; jx < 8; jx++) pl[jx] = 0;
                since = 0;
            };
            if (active) {
                const unsigned *col = ns + wi;
                unsigned long long e = e0 + grp;
                for (; e + 7ull * ngrp < e1; e += 8ull * ngrp) {
                    unsigned xs[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const unsigned ent = s_ent[e + (unsigned long long)k * ngrp];
                        xs[k] = __builtin_nontemporal_load(col + (size_t)site_of[(ent & ~ENT_LONG) >> ENT_SHIFT] * ns_words);
                    }
                    sliced_add8(pl, xs);
                    if (++since == 31u) flush();
                }
                for (; e < e1; e += ngrp) {
                    const unsigned ent = s_ent[e];
                    sliced_add(pl, __builtin_nontemporal_load(col + (size_t)site_of[(ent & ~ENT_LONG) >> ENT_SHIFT] * ns_words));
                    if (++since >= 24u) flush();               // (single adds count one each: 8 x 31 + these stay below 255)
                }
                if (since) flush();
            }
        }
    } else if constexpr (!GRAM) {
        Walk<CLAMP> W;
        W.init(lines, row + chunk + 64 + wave * (WALK_LDS_PER_WAVE / 4), lane);
        W.neg4lo = 0u - 4u * c0; W.dump4 = 4u * (span + lane); W.val = 0xFFFFFFFFu; W.cut = 0u;
        for (unsigned long long base = e0 + (unsigned long long)wave * 64; base < e1; base += (unsigned long long)nwaves * 64) {
            const unsigned long long e = base + lane;
            const unsigned ent = e < e1 ? s_ent[e] : 0u;
            W.drain_lines_to(31);
            W.push_line(e < e1, (ent & ~ENT_LONG) >> ENT_SHIFT);
        }
        W.finish();
    }
    __syncthreads();
    if (upper) {
        const unsigned cx = c_p[x];
        for (unsigned j = up0 + threadIdx.x; j < up1; j += blockDim.x) {
            const unsigned t = row[j - c0] + cx + c_p[j];
            if (t) dist[(size_t)x * ld + j] += t;
        }
    }
    if (lower) {
        unsigned *dst = S + (size_t)(x - row_begin) * s_pitch;
        for (unsigned y = lw0 + threadIdx.x; y < lw1; y += blockDim.x) dst[y - row_begin] = row[y - c0];
    }
}

// dist[y][x] += S[x - row_begin][y - row_begin] for the cells (y, x > y) of the call: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_add_kernel(const unsigned *__restrict__ S, size_t s_pitch, unsigned n, unsigned row_begin,
                                                            unsigned row_end, unsigned col_begin, unsigned *__restrict__ dist, size_t ld)
{
    __shared__ unsigned tile[32][33];
    const unsigned xb = row_begin + blockIdx.x * 32, yb = row_begin + blockIdx.y * 32;
    if (xb + 31 <= yb) return;                              // every x of the tile <= every y: no cell above the diagonal
    const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned x = xb + ty + 8 * k, y = yb + tx;
        tile[ty + 8 * k][tx] = (x < n && y < row_end && y < x) ? S[(size_t)(x - row_begin) * s_pitch + (y - row_begin)] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned y = yb + ty + 8 * k, x = xb + tx;
        const unsigned v = tile[tx][ty + 8 * k];
        if (v && x < n && y < row_end && y < x && x >= col_begin) dist[(size_t)y * ld + x] += v;
    }
}

// site_of[rank] = the site of the list rank: the set bits of the groups' list masks in site order
__global__ __launch_bounds__(256) void site_of_rank_kernel(const uint4 *__restrict__ mask, const unsigned *__restrict__ off, size_t groups,
                                                           unsigned *__restrict__ site_of)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const uint4 m4 = mask[g];
    const unsigned m[4] = {m4.x, m4.y, m4.z, m4.w};
    unsigned o = off[g];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned xw = m[w];
        while (xw) { const unsigned b = __ffs(xw) - 1; site_of[o++] = (unsigned)(g * SITES_PER_GROUP + w * 32 + b); xw &= xw - 1; }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
void minority_lists_free(tracs_alignment *a)
{
    if (!a) return;
    delete a->lists;                               // (the arrays live in the alignment's pack arena: released with it)
    a->lists = nullptr;
}

int minority_lists_build(tracs_alignment *a, const MinorBuild &mb_, hipStream_t stream, int *ok)
{
    *ok = 0;
    minority_lists_free(a);
    MinorBuild mb = mb_;
    const size_t n = a->n, L = mb.sites, groups = a->groups;
    if (L == 0 || L >= (1ull << 26) || n >= (1ull << 27)) return TRACS_OK;           // entries hold rank << 5 (+ a flag bit) / sample << 5
    auto *g = new SiteLists();
    auto fail_soft = [&]() { (void)hipGetLastError(); delete g; return TRACS_OK; };
#define SL_TRY(x) do { if ((x) != hipSuccess) return fail_soft(); } while (0)
    g->sites = L; g->groups = groups; g->tot_p = mb.tot_p; g->tot_nnl = mb.tot_nnl;
    g->n_rows = mb.n_rows;
    for (int k = 0; k < 4; k++) g->rows[k] = mb.rows[k];
    g->n_lines = mb.gram ? 0ull : (unsigned long long)L + mb.tot_o;      // (gram: no site carries an N list)
    g->tgroups = (groups + 15) / 16 * 16;
    SL_TRY(pack_alloc(a, (g->n_lines + 1) * 128, reinterpret_cast<void **>(&g->lines)));
    SL_TRY(pack_alloc(a, (L + 1) * 8, reinterpret_cast<void **>(&g->p_off)));
    SL_TRY(pack_alloc(a, std::max<size_t>(mb.tot_p, 1) * 4, reinterpret_cast<void **>(&g->p_ent)));
    g->n_qlines = mb.long_p ? (unsigned long long)L + mb.tot_q : 0ull;
    g->qw = mb.qw;
    SL_TRY(pack_alloc(a, (g->n_qlines + 1) * (size_t)g->qw * 4, reinterpret_cast<void **>(&g->qlines)));
    SL_TRY(pack_alloc(a, (n + 1) * 8, reinterpret_cast<void **>(&g->s_off)));
    SL_TRY(pack_alloc(a, std::max<size_t>(mb.tot_p, 1) * 4, reinterpret_cast<void **>(&g->s_ent)));
    SL_TRY(pack_alloc(a, std::max<size_t>(n, 1) * 4, reinterpret_cast<void **>(&g->c_p)));
    SL_TRY(pack_alloc(a, groups * sizeof(uint4), reinterpret_cast<void **>(&g->lst_mask)));
    SL_TRY(pack_alloc(a, groups * sizeof(unsigned), reinterpret_cast<void **>(&g->off_lst)));
    const bool bitmaps = mb.tot_nnl > 0;
    if (bitmaps) SL_TRY(pack_alloc(a, n * g->tgroups * sizeof(uint4), reinterpret_cast<void **>(&g->T)));
    unsigned *cnt = nullptr, *e_cnt = nullptr;
    uint2 *E = nullptr, *tmp = nullptr;
    int rc;
    // E: (sample, entry) of every p-list entry with w = 1: what the per-sample lists are bucketed from.  A group's entries sit in its run of
    // list positions -- closed up at the front (e_cnt[g] of them: p_lists_kernel) or where their list has them, holes between
    // (e_cnt[g] = all ones: site_lists_kernel; everywhere when the fill is the one-pass one)
    if ((rc = workspace_get(60, (2 * std::max<size_t>(n, 1) + 8 + 256) * 4, reinterpret_cast<void **>(&cnt))) ||
        (rc = workspace_get(61, std::max<size_t>(groups, 1) * sizeof(unsigned), reinterpret_cast<void **>(&e_cnt))) ||
        (rc = workspace_get(62, std::max<size_t>(mb.tot_p, 1) * sizeof(uint2), reinterpret_cast<void **>(&E)))) { delete g; return rc; }
    unsigned *cur = cnt + std::max<size_t>(n, 1), *d_max = cur + std::max<size_t>(n, 1), *bcur = d_max + 8;
    unsigned shift = 0;
    while (n && ((n - 1) >> shift) >= 256u) shift++;                                  // at most 256 buckets of 2^shift samples
    const bool two_pass = mb.tot_p >= ENT_TILE && mb.tot_p < (1ull << 32) && shift <= 10 &&      // (32-bit run cursors; at most 1 024 samples per bucket)
                          workspace_get(63, (size_t)mb.tot_p * sizeof(uint2), reinterpret_cast<void **>(&tmp)) == TRACS_OK;
    if (!two_pass) { (void)hipGetLastError(); set_error(""); }
    SL_TRY(hipMemsetAsync(cnt, 0, (2 * std::max<size_t>(n, 1) + 8 + 256) * 4, stream));
    SL_TRY(hipMemsetAsync(g->c_p, 0, std::max<size_t>(n, 1) * 4, stream));
    SL_TRY(hipMemcpyAsync(g->lst_mask, mb.lst_mask, groups * sizeof(uint4), hipMemcpyDeviceToDevice, stream));
    SL_TRY(hipMemcpyAsync(g->off_lst, mb.off_lst, groups * sizeof(unsigned), hipMemcpyDeviceToDevice, stream));
    const double plane_b = (double)groups * (double)a->n_pad * sizeof(uint4);      // the N plane
#ifndef TRACS_SITE_PIECE
#define TRACS_SITE_PIECE 512
#endif
    hipLaunchKernelGGL((site_lists_kernel<TRACS_SITE_PIECE>), dim3((unsigned)groups), dim3(SITE_THREADS), 0, stream, mb, a->n_pad, (unsigned)n, g->p_off, g->p_ent, reinterpret_cast<unsigned *>(g->qlines), E, e_cnt, g->lines, g->c_p);
    // (only when some group's lists are that kernel's: at least PL_MIN entries)
    if (mb.max_gp >= PL_MIN)
        hipLaunchKernelGGL(p_lists_kernel, dim3((unsigned)groups), dim3(PL_THREADS), 0, stream, mb, a->n_pad, (unsigned)n, g->p_ent, reinterpret_cast<unsigned *>(g->qlines), E, e_cnt, two_pass ? 0 : 1, g->c_p);
    // bytes as the memory system moves them (PMC, profiles/r06/pmc_site_lists_kernel.txt: 9.1 GB fetched, 1.7 GB written at 10 000 x 5 Mbp
    // where the arrays alone are 6.3 + 0.8 GB): a listed (sample, group) pair's four allele words are four 16-byte gathers that pull a
    // whole 128-byte line each (3.9 M pairs: 2.0 GB for 0.25 GB of words), and an n8 line leaves as 16-byte pieces, half a 32-byte sector each
    const double flagged = (double)std::min<unsigned long long>(mb.tot_p, (unsigned long long)n * groups);
    pack_stage_mark("lists: per site", stream, (mb.gram ? 0.0 : plane_b) + (double)groups * SITES_PER_GROUP * 8.0 + std::min(flagged * 4.0 * 128.0, 4.0 * plane_b),
                    (mb.gram ? 0.0 : (double)L * 128.0 * 2.0) + (double)mb.tot_p * 12.0 + (double)L * 8.0 + (double)(mb.tot_q + std::min<unsigned long long>(L, mb.tot_p)) * 8.0);
    // the per-sample lists hold the w = 1 entries: c_p[s] of them
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, g->c_p, n, g->s_off);
    if (mb.tot_p) {
        if (two_pass) {
            const unsigned buckets = (unsigned)(((n - 1) >> shift) + 1);
            hipLaunchKernelGGL(entries_to_buckets_kernel, dim3((unsigned)((groups + ENT_GROUPS - 1) / ENT_GROUPS)), dim3(ENT_TILE_THREADS), 0, stream, E, e_cnt, mb.gP, mb.baseP,
                               groups, shift, (unsigned)n, g->s_off, bcur, tmp);
            const unsigned per_bucket = (unsigned)std::min<unsigned long long>(64, std::max<unsigned long long>(1, mb.tot_p / buckets / ENT_TILE + 1));
            hipLaunchKernelGGL(buckets_to_samples_kernel, dim3(per_bucket, buckets), dim3(ENT_TILE_THREADS), 0, stream, tmp, shift, (unsigned)n, g->s_off, cur, g->s_ent);
        } else {
            hipLaunchKernelGGL(listed_entries_kernel, dim3((unsigned)((mb.tot_p + 255) / 256)), dim3(256), 0, stream, E, mb.tot_p, g->s_off, cur, g->s_ent);
        }
    }
    pack_stage_mark("listed entries per sample", stream, (double)std::min<unsigned long long>(mb.tot_p, 2 * L) * (two_pass ? 16.0 : 8.0) + (double)n * 4.0,
                    (double)std::min<unsigned long long>(mb.tot_p, 2 * L) * (two_pass ? 12.0 : 4.0) + (double)n * 12.0);
    if (bitmaps) {
        // (a->c_counted was zeroed by the caller: this kernel is what fills it when the rows' bitmaps are built)
        const size_t octs = g->tgroups / NB_GW;
        const unsigned chunks = (unsigned)std::min<size_t>(64, std::max<size_t>(1, octs / 16));
        const size_t opc = (octs + chunks - 1) / chunks;
        const dim3 grid((unsigned)((n + 4 * NB_SW - 1) / (4 * NB_SW)), (unsigned)((octs + opc - 1) / opc));
        hipLaunchKernelGGL(n_bitmap_kernel, grid, dim3(256), 0, stream, mb, a->n_pad, (unsigned)n, groups, g->tgroups, opc, g->T, a->c_counted);
        hipLaunchKernelGGL(max_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a->c_counted, n, d_max);
        double rows = (double)n;
        if (mb.n_rows) { rows = 0.0; for (int k = 0; k < mb.n_rows; k++) rows += (double)(std::min<size_t>(mb.rows[2 * k + 1], n) - std::min<size_t>(mb.rows[2 * k], n)); }
        pack_stage_mark("N bitmaps of the rows", stream, plane_b, rows * (double)g->tgroups * sizeof(uint4));
        SL_TRY(hipMemcpyAsync(&g->max_row, d_max, 4, hipMemcpyDeviceToHost, stream));       // (read after the caller's synchronisation)
    }
    if (mb.gram == 2) {
        // nw_rows: the N plane site-major, and the site of every rank (class_list_kernel's job, here over the mask of the sites with lists)
        g->ns_words = ns_words_for(a->n_pad);
        SL_TRY(pack_alloc(a, groups * 128 * g->ns_words * 4, reinterpret_cast<void **>(&g->ns)));
        SL_TRY(pack_alloc(a, std::max<size_t>(L, 1) * 4, reinterpret_cast<void **>(&g->site_of)));
        launch_ns_build(a->planes, a->n_pad, (unsigned)groups, g->ns, g->ns_words, stream);
        hipLaunchKernelGGL(site_of_rank_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, g->lst_mask, g->off_lst, groups, g->site_of);
        pack_stage_mark("N plane site-major (NS)", stream, plane_b, (double)groups * 128.0 * (double)g->ns_words * 4.0);
    }
    SL_TRY(hipGetLastError());
#undef SL_TRY
    a->lists = g;
    *ok = 1;
    return TRACS_OK;
}

// columns of the pair matrix per LDS row: the row's counters + 64 dump slots + the waves' rings must fit the CU's 160 KiB
constexpr unsigned ROW_CHUNK_MAX = 36800;
static unsigned row_chunk(size_t n) { return (unsigned)std::min<size_t>((n + 63) / 64 * 64, ROW_CHUNK_MAX); }
static constexpr size_t kWalkLds = (size_t)(TRACS_NN_THREADS / 64) * WALK_LDS_PER_WAVE;

int nn_rows_add(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *ncomp, size_t ld, int add_terms,
                unsigned lu, hipStream_t stream)
{
    const SiteLists *g = a->lists;
    if (!g || !g->T) { set_error("nn_rows_add: lists not built"); return TRACS_E_ARG; }
    const size_t n = a->n;
    const unsigned chunk = row_chunk(n);
    const size_t lds = (size_t)chunk * 4 + 256 + kWalkLds;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[64] = {false};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_CHUNK_MAX * 4 + 256 + (int)kWalkLds));
        TRACS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_CHUNK_MAX * 4 + 256 + (int)kWalkLds));
        attr_set[dev] = true;
    }
    // ~2048 workgroups' worth of walks each, never less than 8192 (a workgroup's fixed cost: its row in LDS)
    const unsigned target = (unsigned)std::min<unsigned long long>(1u << 30, std::max<unsigned long long>(8192ull, g->tot_nnl / 2048ull));
    // Every row walks the sites in order, and the rows in flight are anywhere along the genome: a line is re-read (by the ~cN rows
    // that are N at its site) long after it left the caches -- the walk then runs at the rate of random 128-byte reads from HBM.
    // Cut into SEGMENTS of sites that all rows walk before any row starts on the next (grid.z is the slowest dimension of the
    // dispatch order), the lines of a segment stay in the 256 MiB Infinity Cache between their uses (the rows' bitmaps, read once,
    // are loaded non-temporally).  The price is one flush of the row per segment (atomic adds): 10 000 x 5 Mbp, 1.2 GB of lines:
    // 10.9 ms in one segment, 10.2 in 3 to 5 (512 / 256 MiB), 10.6 in 10, 12.9 in 32 (profiles/r04/nn_rows_n8.txt).
    // TRACS_NN_SEGMENT_MB overrides the segment size (0: one segment) for that measurement.
    const char *seg_env = getenv("TRACS_NN_SEGMENT_MB");
    const unsigned long long seg_bytes = (seg_env ? strtoull(seg_env, nullptr, 10) : 256ull) << 20;
    const unsigned segments = seg_bytes ? (unsigned)std::min<unsigned long long>(NN_MAX_SPLITS, std::max<unsigned long long>(1, (g->n_lines * 128ull + seg_bytes - 1) / seg_bytes)) : 1u;
    const unsigned splits = (unsigned)std::min<unsigned long long>(NN_MAX_SPLITS, std::max<unsigned long long>(segments, ((unsigned long long)g->max_row + target - 1) / target));
    const dim3 grid((unsigned)(row_end - row_begin), (unsigned)((n + chunk - 1) / chunk), splits);
    if (grid.y == 1)                                       // one column chunk: every decoded position is a counter of the row
        hipLaunchKernelGGL(nn_rows_kernel<false>, grid, dim3(TRACS_NN_THREADS), lds, stream, g->T, g->tgroups, g->lst_mask, g->off_lst, g->groups, g->lines,
                           a->c_counted, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, target, segments, ncomp, ld, add_terms, lu);
    else
        hipLaunchKernelGGL(nn_rows_kernel<true>, grid, dim3(TRACS_NN_THREADS), lds, stream, g->T, g->tgroups, g->lst_mask, g->off_lst, g->groups, g->lines,
                           a->c_counted, (unsigned)n, (unsigned)row_begin, (unsigned)col_begin, chunk, target, segments, ncomp, ld, add_terms, lu);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

// dist[i][j] += the minority sites' contribution (alignments cut into site classes)
int minority_fixup(tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin, unsigned *dist, size_t ld, hipStream_t stream)
{
    const SiteLists *g = a->lists;
    if (!g || g->tot_p == 0) return TRACS_OK;
    const size_t n = a->n;
    const unsigned chunk = row_chunk(n);
    const size_t lds = (size_t)chunk * 4 + 256 + (size_t)16 * WALK_LDS_PER_WAVE;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[64] = {false};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        const int fix_lds = (int)ROW_CHUNK_MAX * 4 + 256 + 16 * (int)WALK_LDS_PER_WAVE;
        const void *fns[12] = {reinterpret_cast<const void *>(minor_fixup_kernel<true, 32, false>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 32, false>),
                               reinterpret_cast<const void *>(minor_fixup_kernel<true, 64, false>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 64, false>),
                               reinterpret_cast<const void *>(minor_fixup_kernel<true, 32, true>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 32, true>),
                               reinterpret_cast<const void *>(minor_fixup_kernel<true, 64, true>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 64, true>),
                               reinterpret_cast<const void *>(minor_fixup_kernel<true, 32, false, true>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 32, false, true>),
                               reinterpret_cast<const void *>(minor_fixup_kernel<true, 64, false, true>), reinterpret_cast<const void *>(minor_fixup_kernel<false, 64, false, true>)};
        for (const void *fn : fns) TRACS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, fix_lds));
        attr_set[dev] = true;
    }
    // scratch rows for the cells (y, x) with y < x that row x's walks feed: (n - row_begin) rows of (row_end - row_begin) columns
    const size_t s_pitch = (row_end - row_begin + 63) / 64 * 64;
    unsigned *S = nullptr;
    const int rc = workspace_get(46, (n - row_begin) * s_pitch * sizeof(unsigned), reinterpret_cast<void **>(&S));
    if (rc) return rc;
    const dim3 grid((unsigned)(n - row_begin), (unsigned)((n + chunk - 1) / chunk));
#define TRACS_FIXUP_LAUNCH(CL, QWV) hipLaunchKernelGGL((minor_fixup_kernel<CL, QWV, GR, NR>), grid, dim3(1024), lds, stream, g->s_off, g->s_ent, g->p_off, g->p_ent, \
        reinterpret_cast<const unsigned *>(g->qlines), g->lines, g->c_p, (unsigned)n, (unsigned)row_begin, (unsigned)row_end, (unsigned)col_begin, chunk, dist, ld, S, s_pitch, \
        g->ns, g->ns_words, g->site_of)
    if (a->nw_rows) {
        constexpr bool GR = false, NR = true;
        if (grid.y == 1) { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(false, 64); else TRACS_FIXUP_LAUNCH(false, 32); }
        else { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(true, 64); else TRACS_FIXUP_LAUNCH(true, 32); }
    } else if (a->nw_gram) {
        constexpr bool GR = true, NR = false;
        if (grid.y == 1) { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(false, 64); else TRACS_FIXUP_LAUNCH(false, 32); }
        else { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(true, 64); else TRACS_FIXUP_LAUNCH(true, 32); }
    } else {
        constexpr bool GR = false, NR = false;
        if (grid.y == 1) { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(false, 64); else TRACS_FIXUP_LAUNCH(false, 32); }
        else { if (g->qw == 64u) TRACS_FIXUP_LAUNCH(true, 64); else TRACS_FIXUP_LAUNCH(true, 32); }
    }
#undef TRACS_FIXUP_LAUNCH
    const dim3 tgrid((unsigned)((n - row_begin + 31) / 32), (unsigned)((row_end - row_begin + 31) / 32));
    hipLaunchKernelGGL(transpose_add_kernel, tgrid, dim3(256), 0, stream, S, s_pitch, (unsigned)n, (unsigned)row_begin, (unsigned)row_end,
                       (unsigned)col_begin, dist, ld);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

}  // namespace tracs

extern "C" {

// Diagnostics (tests/test_gpu_lists.py): the lists of an alignment on site classes, copied to the host.
//   what 0  sizes: out64[0..7] = sites with lists, lines, p entries, tgroups, groups, bitmap present, n, the most N sites of a sample
//   what 1  lines (n_lines x 128 bytes)     what 2  lst_mask (groups x 16 bytes)     what 3  off_lst (groups x 4 bytes)
//   what 4  p_off ((sites + 1) x 8)          what 5  p_ent (tot_p x 4: short lists)   what 6  s_off ((n + 1) x 8)
//   what 7  s_ent (tot_p x 4)                what 8  T (n x tgroups x 16)             what 9  c_p (n x 4)
//   what 10 q lines (n_qlines x 4 qw bytes: the p lists)     what 11 out64[0] = n_qlines     what 12 out64[0] = qw (dwords per q line)
// Returns the bytes copied (what >= 1), 0 when the lists do not exist or `cap` is too small.
size_t tracs_debug_lists(const tracs_alignment *a, int what, void *out, size_t cap)
{
    using namespace tracs;
    if (!a || !a->lists || !out) return 0;
    const SiteLists *g = a->lists;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (what) {
    case 0: {
        if (cap < 64) return 0;
        uint64_t *o = static_cast<uint64_t *>(out);
        o[0] = g->sites; o[1] = g->n_lines; o[2] = g->tot_p; o[3] = g->tgroups; o[4] = g->groups; o[5] = g->T ? 1 : 0; o[6] = a->n; o[7] = g->max_row;
        return 64;
    }
    case 1: src = g->lines; bytes = g->n_lines * 128; break;
    case 2: src = g->lst_mask; bytes = g->groups * 16; break;
    case 3: src = g->off_lst; bytes = g->groups * 4; break;
    case 4: src = g->p_off; bytes = (g->sites + 1) * 8; break;
    case 5: src = g->p_ent; bytes = g->tot_p * 4; break;
    case 6: src = g->s_off; bytes = (a->n + 1) * 8; break;
    case 7: src = g->s_ent; bytes = g->tot_p * 4; break;
    case 8: src = g->T; bytes = g->T ? a->n * g->tgroups * 16 : 0; break;
    case 9: src = g->c_p; bytes = a->n * 4; break;
    case 10: src = g->qlines; bytes = g->n_qlines * g->qw * 4; break;
    case 11: if (cap < 8) return 0; *static_cast<uint64_t *>(out) = g->n_qlines; return 8;
    case 12: if (cap < 8) return 0; *static_cast<uint64_t *>(out) = g->qw; return 8;
    default: return 0;
    }
    if (!src || bytes == 0 || bytes > cap) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return bytes;
}

}  // extern "C"
