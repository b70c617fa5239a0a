// common.h -- shared host-side helpers of libtracs_hip.so (error slot, layout constants).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/tracs_hip.h"

namespace tracs {

void set_error(const std::string &msg);

#define TRACS_HIP_CHECK(expr)                                                                   \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            ::tracs::set_error(std::string(#expr) + ": " + hipGetErrorString(e__));             \
            return TRACS_E_HIP;                                                                 \
        }                                                                                       \
    } while (0)

// ---- packed alignment layout -----------------------------------------------------------
// Sites are cut into groups of 128 (one uint4 = 4 x 32 sites).  For group g, plane p
// (0..3 = A,C,G,T allele planes, 4 = N plane = A&C&G&T) and sample s the 16 bytes live at
//     planes[(g * NPLANES + p) * n_pad + s]
// i.e. sample-minor, so a tile's 64..256 consecutive samples of one (group, plane) are one
// contiguous 1..4 KiB run: coalesced for the LDS stage and for the scalar row loads.
constexpr int NPLANES = 5;
constexpr int SITES_PER_GROUP = 128;
constexpr int SAMPLE_PAD = 64;       // n_pad is a multiple of one staging wave-instruction (64 samples x 16 B)
constexpr int TAIL_PAD = 512;        // zeroed uint4 behind the last plane: tiles may read (never use) up to one tile edge past n_pad
constexpr int PAD_GROUPS = 7;        // all-zero groups behind the last real one: the matrix-core kernels' stages (two groups; the
                                     // counting pass, which also reads the stored N plane in place: four, sweeps up to eight) may
                                     // overhang the alignment (zero planes contribute nothing)

static inline size_t groups_for(size_t L) { return (L + SITES_PER_GROUP - 1) / SITES_PER_GROUP; }
static inline size_t pad_samples(size_t n) { return (n + SAMPLE_PAD - 1) / SAMPLE_PAD * SAMPLE_PAD; }

}  // namespace tracs

namespace tracs { struct GeneralSparse; struct SiteLists; struct FilterIndex; }

struct tracs_alignment {
    size_t n = 0, L = 0, n_pad = 0, groups = 0;
    uint4 *planes = nullptr;     // device: general encoding, 5 planes
    uint4 *cplanes = nullptr;    // device: consensus encoding, 3 planes (derived on demand, only if valid)
    unsigned *d_flag = nullptr;  // device: "some site has a partial IUPAC code"
    bool dirty = true;           // packed since the encoding was last decided
    int enc = 0;                 // 0 general, 1 consensus
    int last_kernel = -1;        // kernel of the last dense call: 0 VALU tile kernel, 1 / 2 matrix-core kernel (consensus / one-hot)
    // Site classes (site_classes.hip), decided once per pack: a site at which every sample that is not N carries the same
    // base adds 0 to every distance and [neither is N] to every compared-sites count.  When enough sites are like that the
    // pair kernels read `vplanes` -- the alignment restricted to the DENSE sites, same encoding and layout as their usual
    // source -- and a one-operand matrix-core pass over the N plane of the other sites (`iplanes`, or the stored N plane in
    // place) completes the compared-sites counts.
    uint4 *vplanes = nullptr, *iplanes = nullptr;
    uint4 *uplane = nullptr;                  // nw_gram: "is N, or listed with w = 1 at a minority site" (one plane per group, site_classes.hip)
    bool nw_gram = false;                     // the minority sites' N x listed terms come from two one-plane matrix passes (U U^T - n n^T),
                                              // their lists hold the listed samples only (no N lists)
    bool nw_rows = false;                     // ... or, where the listed samples are few, from the rows of the site-major N matrix summed per listed
                                              // sample (site_lists.hip, minor_fixup_kernel<NSROWS>): no U plane, no second matrix pass
    size_t L_var = 0, L_inv = 0, groups_var = 0, groups_inv = 0;
    size_t L_minor = 0, L_full = 0;           // minority sites (listed in `minor`; they are part of L_un or L_full as well);
                                              // sites without any N outside vplanes: +1 to every compared-sites count
    size_t L_un = 0, L_nnl = 0;               // sites outside vplanes with an N sample (L_inv of them on the matrix cores, L_nnl
                                              // through their N lists, the rest with a single N sample: no co-occurrence)
    unsigned long long nn_visits = 0;         // list entries one pass of the N co-occurrence walk visits (sum of cN^2 over the L_nnl sites)
    unsigned long long list_entries_n = 0, list_entries_p = 0;     // 128-byte lines of the per-site N lists (primary + overflow reserve) / listed-sample entries
    unsigned long long nn_walks = 0;          // list walks of one pass of the N co-occurrence walk (sum of cN over the L_nnl sites)
    unsigned long long fix_walks = 0;         // N-list walks of one pass of the minority fix-up (listed samples of minority sites with an N sample)
    unsigned *c_counted = nullptr;            // per sample: its N sites among the sites the counting pass reads
    bool count_in_place = false;              // the counting pass reads the stored N plane of `planes` (every site) instead of iplanes:
                                              // nn = L - c_i - c_j + NN comes from it alone and the pair kernels write d only
    bool classes_cons = false;                // vplanes hold consensus planes (X, Y, V) / the five general planes
    tracs::SiteLists *lists = nullptr;        // lists of the sites with lists: minority and NNL sites (site_lists.hip)
    size_t row_hint[4] = {0, 0, 0, 0};   // tracs_alignment_hint_rows: the only rows this handle will be asked for
    int n_row_hint = 0;
    int classes_state = 0;       // 0 not decided, 1 in use, -1 not in use for this alignment
    // Arena for what is built once per pack (vplanes, iplanes, N counts, minority lists): reserved together with the planes at
    // tracs_alignment_create, because a fresh device allocation costs ~24 ms per GB on this platform (the driver clears VRAM) and
    // would otherwise land inside the first pass.  pack_alloc falls back to hipMalloc when the arena is too small.
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    struct PackBlock { void *p; size_t bytes; };
    std::vector<PackBlock> pack_extra;       // what did not fit the arena, in use by the current pack ...
    std::vector<PackBlock> pack_spare;       // ... and released by the one before: handed out again before anything is allocated (a
                                             //   handle that is packed and called again and again -- bench.py -- would otherwise free and
                                             //   allocate gigabytes per call: ~24 ms per GB when the driver has to clear them, seconds at times)
    bool pack_oom = false;                   // a pack_alloc of the current pack failed: pack_release frees its blocks instead of parking them
    tracs::GeneralSparse *sparse = nullptr;   // general matrix-core path: per-site / per-sample lists of N and partial codes
    int sparse_state = 0;        // 0 not built, 1 built, -1 not available for this alignment (too dense / too large / no memory)
    tracs::FilterIndex *flt = nullptr;        // recombination filter: per-sample departure lists + N bitmaps (filter_lists.hip)
    bool flt_stale = true;       // packed since the filter index was built
    // cached tile schedules of the last dense regions (region x workgroup tile): the pair kernel's and the counting pass's, for
    // the two row ranges a multi-GPU rank alternates between; replaced round-robin
    struct TileCache {
        int2 *d = nullptr;
        size_t n = 0, cap = 0;
        size_t rb = (size_t)-1, re = 0, cb = 0;
        int ti = 0, tj = 0;
    } tile_cache[4];
    int tile_next = 0;
};

namespace tracs {
// what the pair kernels read: the whole alignment, or its variable sites (site classes in use)
static inline const uint4 *pair_planes(const tracs_alignment *a, bool consensus)
{
    return a->classes_state == 1 ? a->vplanes : (consensus ? a->cplanes : a->planes);
}
static inline size_t pair_L(const tracs_alignment *a) { return a->classes_state == 1 ? a->L_var : a->L; }
static inline size_t pair_groups(const tracs_alignment *a) { return a->classes_state == 1 ? a->groups_var : a->groups; }

void filter_index_free(tracs_alignment *a);        // filter_lists.hip

// device memory that lives until the alignment is packed again (site_classes_free releases all of it at once)
hipError_t pack_alloc(tracs_alignment *a, size_t bytes, void **out);
void pack_release(tracs_alignment *a);

// Grow-only per-device scratch buffers (slot ids are small integers owned by each .hip file), shared by every entry point.
// Entry points that use them, or the cached state of a tracs_alignment, hold a DeviceCall for their whole body:
//   * calls on one device are serialised (ctypes releases the GIL, so two Python threads can be inside the library);
//   * scratch is only stream-ordered, so when a call arrives on a different stream than the previous call on that device
//     the device is synchronised first.  Re-entrant on the owning thread (tracs_pairsnp calls the dense entry points).
int workspace_get(int slot, size_t bytes, void **out);
void workspace_release_all();
struct DeviceCall {
    explicit DeviceCall(hipStream_t stream);
    ~DeviceCall();
    DeviceCall(const DeviceCall &) = delete;
    DeviceCall &operator=(const DeviceCall &) = delete;
    int dev;
};
}  // namespace tracs
