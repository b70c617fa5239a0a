// common.h -- shared host-side helpers of libtracs_hip.so (error slot, layout constants).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

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
constexpr int SAMPLE_PAD = 2048;     // n_pad is a multiple of every tile edge (widest: rowcast 2048 columns)

static inline size_t groups_for(size_t L) { return (L + SITES_PER_GROUP - 1) / SITES_PER_GROUP; }
static inline size_t pad_samples(size_t n) { return (n + SAMPLE_PAD - 1) / SAMPLE_PAD * SAMPLE_PAD; }

}  // namespace tracs

struct tracs_alignment {
    size_t n = 0, L = 0, n_pad = 0, groups = 0;
    uint4 *planes = nullptr;     // device: general encoding, 5 planes
    uint4 *cplanes = nullptr;    // device: consensus encoding, 3 planes (derived on demand, only if valid)
    unsigned *d_flag = nullptr;  // device: "some site has a partial IUPAC code"
    bool dirty = true;           // packed since the encoding was last decided
    int enc = 0;                 // 0 general, 1 consensus
    int last_kernel = -1;        // kernel of the last dense call: 0 VALU tile kernel, 1 matrix-core kernel
    // cached tile schedule for the last dense region (device + host mirror)
    int2 *d_tiles = nullptr;
    size_t n_tiles = 0, tiles_cap = 0;
    size_t key_rb = (size_t)-1, key_re = 0, key_cb = 0;
    int key_ti = 0, key_tj = 0;
};

namespace tracs {
// Grow-only per-device scratch buffers (slot ids are small integers owned by each .hip file).
// Not thread-safe across concurrent calls on one device: one in-flight library call per device.
int workspace_get(int slot, size_t bytes, void **out);
void workspace_release_all();
}  // namespace tracs
