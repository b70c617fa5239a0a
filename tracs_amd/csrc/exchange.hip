// exchange.hip -- the compact form of the site-sharded exchange (gfx950): pack / sum of the upper-triangle cells of row panels.
//
// The reference has no counterpart (one process: src/pairsnp.hpp:380-382).  d(i, j) and the compared-sites count nn(i, j) are sums
// over sites (src/pairsnp.hpp:398-403,417-420): ranks that each hold a slice of the sites compute partial matrices for ALL pairs,
// and rank q needs the sums of the rows it owns.  Only the cells (i, j >= max(col_begin, i + 1)) exist, a slice's partial distances
// are small, and a slice's partial compared-sites counts sit just below the slice's length: so every rank packs, for every OTHER
// rank q, the cells of q's rows -- 16 bits per cell where the values fit: d as it is, nn as its deficit base - nn (base = the
// slice's sites) -- into block q of a send buffer, the blocks travel point to point (tracs_alltoall: every xGMI link at once),
// and the receiver adds the P - 1 blocks it got to its own partial rows in 32 bits:
//     d  = d_own  + sum_p d_p                      nn = nn_own + (L - L_own) - sum_p (L_p - nn_p)
// Both kernels are streams over the panel (HBM-bound: 4 B read + 2 B written per cell, and back), one workgroup row per matrix row.
#include "common.h"

#include <algorithm>

namespace tracs {

// first column of row i that exists in the output (pairsnp.hpp:383: j from max(j_start, i + 1))
__device__ __forceinline__ size_t first_col(size_t i, size_t col_begin) { return i + 1 > col_begin ? i + 1 : col_begin; }

template <typename T>
__global__ void __launch_bounds__(256) tri_pack_kernel(const uint32_t *__restrict__ mat, size_t ld, size_t n, size_t row_begin, size_t col_begin,
                                                       const unsigned long long *__restrict__ row_slot, uint32_t base, int negate,
                                                       T *__restrict__ packed, uint32_t *__restrict__ stats)
{
    const size_t i = row_begin + blockIdx.x;
    const unsigned long long slot = row_slot[blockIdx.x];
    if (slot == ~0ull) return;                                        // a row this rank owns: nothing travels
    const size_t jb = first_col(i, col_begin);
    const uint32_t *src = mat + i * ld;
    T *dst = packed + slot;
    uint32_t vmax = 0, over = 0;
    for (size_t j = jb + (size_t)blockIdx.y * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.y * blockDim.x) {
        const uint32_t v = negate ? base - src[j] : src[j];
        vmax = v > vmax ? v : vmax;
        if (sizeof(T) == 2 && v > 0xFFFFu) over++;
        if (packed) dst[j - jb] = (T)v;
    }
    // one atomic per wave that has something to say
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m = __shfl_xor(vmax, o);
        vmax = m > vmax ? m : vmax;
        over += __shfl_xor(over, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (vmax) atomicMax(&stats[0], vmax);
        if (over) atomicAdd(&stats[1], over);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) tri_sum_kernel(uint32_t *__restrict__ mat, size_t ld, size_t n, size_t row_begin, size_t col_begin,
                                                      const unsigned long long *__restrict__ row_slot, const T *__restrict__ recv,
                                                      size_t block_elems, int n_blocks, int skip_block, uint32_t add, int negate)
{
    const size_t i = row_begin + blockIdx.x;
    const unsigned long long slot = row_slot[blockIdx.x];
    if (slot == ~0ull) return;                                        // not a row of this rank
    const size_t jb = first_col(i, col_begin);
    uint32_t *row = mat + i * ld;
    for (size_t j = jb + (size_t)blockIdx.y * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.y * blockDim.x) {
        uint32_t acc = 0;
        const T *p = recv + slot + (j - jb);
        for (int b = 0; b < n_blocks; b++)
            if (b != skip_block) acc += (uint32_t)p[(size_t)b * block_elems];
        row[j] = row[j] + add + (negate ? 0u - acc : acc);
    }
}

}  // namespace tracs

extern "C" {

int tracs_tri_pack(const void *mat, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin, const uint64_t *row_slot,
                   int width, uint32_t base, int negate, void *packed, uint32_t *stats, void *stream_)
{
    if (!mat || !row_slot || !stats || (width != 2 && width != 4) || row_end < row_begin || row_end > n + (1u << 30)) {
        tracs::set_error("tracs_tri_pack: bad argument (width 2 or 4)");
        return TRACS_E_ARG;
    }
    if (row_end == row_begin || n == 0) return TRACS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned gx = (unsigned)std::min<size_t>(16, (n + 1023) / 1024);
    const dim3 grid((unsigned)(row_end - row_begin), gx ? gx : 1);
    const auto *slots = reinterpret_cast<const unsigned long long *>(row_slot);
    if (width == 2)
        tracs::tri_pack_kernel<uint16_t><<<grid, 256, 0, stream>>>(static_cast<const uint32_t *>(mat), ld, n, row_begin, col_begin, slots, base, negate,
                                                                   static_cast<uint16_t *>(packed), stats);
    else
        tracs::tri_pack_kernel<uint32_t><<<grid, 256, 0, stream>>>(static_cast<const uint32_t *>(mat), ld, n, row_begin, col_begin, slots, base, negate,
                                                                   static_cast<uint32_t *>(packed), stats);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

int tracs_tri_sum(void *mat, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin, const uint64_t *row_slot, int width,
                  const void *recv, size_t block_elems, int n_blocks, int skip_block, uint32_t add, int negate, void *stream_)
{
    if (!mat || !row_slot || !recv || (width != 2 && width != 4) || row_end < row_begin || n_blocks < 1) {
        tracs::set_error("tracs_tri_sum: bad argument (width 2 or 4)");
        return TRACS_E_ARG;
    }
    if (row_end == row_begin || n == 0) return TRACS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned gx = (unsigned)std::min<size_t>(16, (n + 1023) / 1024);
    const dim3 grid((unsigned)(row_end - row_begin), gx ? gx : 1);
    const auto *slots = reinterpret_cast<const unsigned long long *>(row_slot);
    if (width == 2)
        tracs::tri_sum_kernel<uint16_t><<<grid, 256, 0, stream>>>(static_cast<uint32_t *>(mat), ld, n, row_begin, col_begin, slots,
                                                                  static_cast<const uint16_t *>(recv), block_elems, n_blocks, skip_block, add, negate);
    else
        tracs::tri_sum_kernel<uint32_t><<<grid, 256, 0, stream>>>(static_cast<uint32_t *>(mat), ld, n, row_begin, col_begin, slots,
                                                                  static_cast<const uint32_t *>(recv), block_elems, n_blocks, skip_block, add, negate);
    TRACS_HIP_CHECK(hipGetLastError());
    return TRACS_OK;
}

}  // extern "C"
