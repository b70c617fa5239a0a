// dirichlet.hip -- fit of the Dirichlet-multinomial prior (alpha) from per-site allele counts, on the GPU.
//
// Reference behaviour restated (never copied): /root/reference/tracs/dirichlet_multinomial.py:9-73
//   optional error filter: cells whose within-site frequency is below the threshold are zeroed (:13-15);
//   keep sites with more than one non-zero allele; if at most 5 such sites return (0,..,0,1) (:20-35);
//   sort each kept row ascending (:36); alpha0 = column means + 0.5 (:40);
//   Minka fixed point  alpha_k <- alpha_k * sum_i[psi(x_ik + alpha_k) - psi(alpha_k)] / sum_i[psi(n_i + a0) - psi(a0)]
//   until sum|delta| < tol, clamping at 1e-16 (:55-68), or the leave-one-out update until max|delta| < tol (:42-54);
//   result sorted descending (:70).
// The per-iteration sums over the polymorphic sites are block reductions (deterministic: site-ordered compaction, fixed two-stage tree); the
// 4-element update runs in a one-wave kernel so the whole fit stays on the device; the host only polls the "converged" flag.
#include "common.h"

namespace tracs {

constexpr int DK = 8;           // max alleles
constexpr int DM_BLOCKS = 512;  // partial-sum blocks

// digamma for x > 0: recurrence up to x >= 6, then the asymptotic series (|err| < 1e-15 relative there)
__device__ __forceinline__ double digamma_pos(double x)
{
    double r = 0.0;
    while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x -
           f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0 - f * (691.0 / 32760.0 - f / 12.0))))));
}

struct DmState {
    double alpha[DK];
    double sums[DK + 1];
    int done, iters, n_rows;
};

// filter + select + sort: kept rows (each sorted ascending) go to `rows` IN SITE ORDER -- a deterministic compaction (every
// block owns a contiguous run of sites; PASS 0 counts its kept rows, dm_block_offsets_kernel scans the block counts, PASS 1
// writes at block offset + rank inside the block), so the f64 summation order of dm_sums_kernel, and with it every bit of the
// fitted alphas and the iteration count, is the same on every run.
template <int PASS>
__global__ __launch_bounds__(256) void dm_select_kernel(const double *__restrict__ counts, size_t L, int K, double filt, int use_filt,
                                                        size_t chunk, unsigned *__restrict__ block_cnt,
                                                        const unsigned *__restrict__ block_off, double *__restrict__ rows)
{
    __shared__ unsigned wave_tot[4];
    const size_t begin = (size_t)blockIdx.x * chunk, end = min(L, begin + chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned running = PASS ? block_off[blockIdx.x] : 0u;
    for (size_t base = begin; base < end; base += 256) {
        const size_t i = base + threadIdx.x;
        double v[DK];
        bool keep = false;
        if (i < end) {
            double tot = 0.0;
            for (int k = 0; k < K; k++) { v[k] = counts[i * K + k]; tot += v[k]; }
            if (use_filt)
                for (int k = 0; k < K; k++)
                    if (v[k] / tot < filt) v[k] = 0.0;       // NaN (0/0) compares false: row unchanged, as in numpy
            int nz = 0;
            for (int k = 0; k < K; k++) nz += v[k] != 0.0;
            keep = nz > 1;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_tot[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int w = 0; w < 4; w++) { if (w < wave) before += wave_tot[w]; total += wave_tot[w]; }
        if (PASS && keep) {
            for (int a = 1; a < K; a++) {                  // ascending insertion sort
                const double x = v[a];
                int b = a;
                while (b > 0 && v[b - 1] > x) { v[b] = v[b - 1]; b--; }
                v[b] = x;
            }
            const size_t o = (size_t)running + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int k = 0; k < K; k++) rows[o * K + k] = v[k];
        }
        running += total;
        __syncthreads();
    }
    if (!PASS && threadIdx.x == 0) block_cnt[blockIdx.x] = running;
}

// exclusive scan of the (<= 1024) block counts by one wave; the grand total is the number of kept rows
__global__ __launch_bounds__(64) void dm_block_offsets_kernel(const unsigned *__restrict__ block_cnt, int nblocks, unsigned *__restrict__ block_off,
                                                              unsigned *__restrict__ n_kept)
{
    unsigned carry = 0;
    for (int base = 0; base < nblocks; base += 64) {
        const int b = base + (int)threadIdx.x;
        const unsigned c = b < nblocks ? block_cnt[b] : 0u;
        unsigned incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += t;
        }
        if (b < nblocks) block_off[b] = carry + incl - c;
        carry += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) *n_kept = carry;
}

// mode 0: column sums (for the initial alpha); mode 1: FPI sums; mode 2: LOO sums
__global__ __launch_bounds__(256) void dm_sums_kernel(const double *__restrict__ rows, const unsigned *__restrict__ n_kept, int K,
                                                      int mode, const DmState *__restrict__ st, double *__restrict__ partial)
{
    __shared__ double sh[256][DK + 1];
    const unsigned M = *n_kept;
    double acc[DK + 1];
    for (int k = 0; k <= DK; k++) acc[k] = 0.0;
    double alpha[DK], a0 = 0.0, psi_a[DK], psi_a0 = 0.0;
    if (mode != 0) {
        if (st->done) return;
        for (int k = 0; k < K; k++) { alpha[k] = st->alpha[k]; a0 += alpha[k]; }
        if (mode == 1) { for (int k = 0; k < K; k++) psi_a[k] = digamma_pos(alpha[k]); psi_a0 = digamma_pos(a0); }
    }
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        double tot = 0.0;
        for (int k = 0; k < K; k++) {
            const double x = rows[(size_t)i * K + k];
            tot += x;
            if (mode == 0) acc[k] += x;
            else if (mode == 1) acc[k] += digamma_pos(x + alpha[k]) - psi_a[k];
            else acc[k] += x / (x - 1.0 + alpha[k]);
        }
        if (mode == 1) acc[DK] += digamma_pos(tot + a0) - psi_a0;
        else if (mode == 2) acc[DK] += tot / (tot - 1.0 + a0);
    }
    for (int k = 0; k <= DK; k++) sh[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k <= DK; k++) sh[threadIdx.x][k] += sh[threadIdx.x + off][k];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k <= DK; k++) partial[(size_t)blockIdx.x * (DK + 1) + k] = sh[0][k];
}

__global__ __launch_bounds__(64) void dm_update_kernel(const double *__restrict__ partial, int nblocks, const unsigned *__restrict__ n_kept, int K,
                                 int mode, double tol, DmState *__restrict__ st)
{
    // one wave: lane l adds the partials of blocks l, l + 64, ... then a fixed xor tree -- same order every run
    if (mode != 0 && st->done) return;
    double s[DK + 1];
    for (int k = 0; k <= DK; k++) s[k] = 0.0;
    for (int b = (int)threadIdx.x; b < nblocks; b += 64)
        for (int k = 0; k <= DK; k++) s[k] += partial[(size_t)b * (DK + 1) + k];
    for (int k = 0; k <= DK; k++)
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off, 64);
    if (threadIdx.x != 0) return;
    const unsigned M = *n_kept;
    if (mode == 0) {                                                  // alpha0 = mean + 0.5 (:40)
        for (int k = 0; k < K; k++) st->alpha[k] = s[k] / (double)M + 0.5;
        st->done = 0; st->iters = 0; st->n_rows = (int)M;
        return;
    }
    double na[DK], delta_sum = 0.0, delta_max = 0.0;
    for (int k = 0; k < K; k++) {
        na[k] = st->alpha[k] * s[k] / s[DK];                          // :46-51 / :58-62
        const double d = fabs(na[k] - st->alpha[k]);
        delta_sum += d;
        delta_max = fmax(delta_max, d);
    }
    const bool conv = mode == 1 ? (delta_sum < tol) : (delta_max < tol);   // :63 / :52
    for (int k = 0; k < K; k++) {
        double v = na[k];
        if (mode == 1 && !conv && v < 1e-16) v = 1e-16;               // :68 (only on the non-converged path)
        st->alpha[k] = v;
    }
    st->iters++;
    if (conv) st->done = 1;
}

}  // namespace tracs

using namespace tracs;

extern "C" {

// counts: device f64 [L][K] (A,C,G,T,..).  method: 0 = fixed point (any string but "LOO" in the reference), 1 = LOO.
// error_filt_threshold < 0 means None.  alphas_out: host, K doubles, sorted descending.  iters_out optional.
int tracs_find_dirichlet_priors_device(const double *counts, size_t L, size_t K, int max_iter, double tol, int method,
                                       double error_filt_threshold, double *alphas_out, int *iters_out, void *stream_)
{
    if (!counts || !alphas_out) { set_error("tracs_find_dirichlet_priors_device: NULL argument"); return TRACS_E_ARG; }
    if (K < 1 || K > DK) { set_error("find_dirichlet_priors: 1 <= K <= 8 alleles supported"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    double *rows = nullptr, *partial = nullptr;
    unsigned *n_kept = nullptr;
    DmState *st = nullptr;
    int rc;
    enum { WS_ROWS = 32, WS_PARTIAL, WS_NKEPT, WS_STATE, WS_BLOCKS };
    if ((rc = workspace_get(WS_ROWS, std::max<size_t>(L, 1) * K * 8, reinterpret_cast<void **>(&rows)))) return rc;
    if ((rc = workspace_get(WS_PARTIAL, (size_t)DM_BLOCKS * (DK + 1) * 8, reinterpret_cast<void **>(&partial)))) return rc;
    if ((rc = workspace_get(WS_NKEPT, 64, reinterpret_cast<void **>(&n_kept)))) return rc;
    if ((rc = workspace_get(WS_STATE, sizeof(DmState), reinterpret_cast<void **>(&st)))) return rc;
    TRACS_HIP_CHECK(hipMemsetAsync(n_kept, 0, 4, stream));
    TRACS_HIP_CHECK(hipMemsetAsync(st, 0, sizeof(DmState), stream));
    if (L) {
        const unsigned blocks = (unsigned)std::min<size_t>((L + 255) / 256, 1024);
        const size_t chunk = ((L + blocks - 1) / blocks + 255) / 256 * 256;           // contiguous sites per block, whole tiles
        unsigned *block_cnt = nullptr;
        if ((rc = workspace_get(WS_BLOCKS, 2 * 1024 * sizeof(unsigned), reinterpret_cast<void **>(&block_cnt)))) return rc;
        unsigned *block_off = block_cnt + 1024;
        const int use_filt = error_filt_threshold >= 0 ? 1 : 0;
        hipLaunchKernelGGL(dm_select_kernel<0>, dim3(blocks), dim3(256), 0, stream, counts, L, (int)K, error_filt_threshold, use_filt, chunk,
                           block_cnt, block_off, rows);
        hipLaunchKernelGGL(dm_block_offsets_kernel, dim3(1), dim3(64), 0, stream, block_cnt, (int)blocks, block_off, n_kept);
        hipLaunchKernelGGL(dm_select_kernel<1>, dim3(blocks), dim3(256), 0, stream, counts, L, (int)K, error_filt_threshold, use_filt, chunk,
                           block_cnt, block_off, rows);
    }
    unsigned M = 0;
    TRACS_HIP_CHECK(hipMemcpyAsync(&M, n_kept, 4, hipMemcpyDeviceToHost, stream));
    TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    if (iters_out) *iters_out = 0;
    if (M <= 5) {                                                      // "less than 5 polymorphic loci" (:26-29)
        for (size_t k = 0; k < K; k++) alphas_out[k] = 0.0;
        alphas_out[K - 1] = 1.0;
        return TRACS_OK;
    }
    const int nblocks = (int)std::min<unsigned>((M + 255) / 256, DM_BLOCKS);
    hipLaunchKernelGGL(dm_sums_kernel, dim3(nblocks), dim3(256), 0, stream, rows, n_kept, (int)K, 0, st, partial);
    hipLaunchKernelGGL(dm_update_kernel, dim3(1), dim3(64), 0, stream, partial, nblocks, n_kept, (int)K, 0, tol, st);
    const int mode = method == 1 ? 2 : 1;
    DmState h;
    h.done = 0;
    int it = 0;
    while (it < max_iter) {
        const int burst = std::min(32, max_iter - it);                 // launches after convergence are no-ops
        for (int b = 0; b < burst; b++) {
            hipLaunchKernelGGL(dm_sums_kernel, dim3(nblocks), dim3(256), 0, stream, rows, n_kept, (int)K, mode, st, partial);
            hipLaunchKernelGGL(dm_update_kernel, dim3(1), dim3(64), 0, stream, partial, nblocks, n_kept, (int)K, mode, tol, st);
        }
        it += burst;
        TRACS_HIP_CHECK(hipMemcpyAsync(&h, st, sizeof(DmState), hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
        if (h.done) break;
    }
    if (!h.done) {
        TRACS_HIP_CHECK(hipMemcpyAsync(&h, st, sizeof(DmState), hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    }
    TRACS_HIP_CHECK(hipGetLastError());
    for (size_t k = 0; k < K; k++) alphas_out[k] = h.alpha[k];
    for (size_t a = 1; a < K; a++) {                                   // descending (:70)
        const double v = alphas_out[a];
        size_t b = a;
        while (b > 0 && alphas_out[b - 1] < v) { alphas_out[b] = alphas_out[b - 1]; b--; }
        alphas_out[b] = v;
    }
    if (iters_out) *iters_out = h.iters;
    return TRACS_OK;
}

int tracs_find_dirichlet_priors(const double *counts, size_t L, size_t K, int max_iter, double tol, int method,
                                double error_filt_threshold, double *alphas_out, int *iters_out)
{
    if (!counts || !alphas_out) { set_error("tracs_find_dirichlet_priors: NULL argument"); return TRACS_E_ARG; }
    double *d = nullptr;
    const size_t bytes = std::max<size_t>(L * K, 1) * 8;
    TRACS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d), bytes));
    hipError_t e = L ? hipMemcpy(d, counts, L * K * 8, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) { (void)hipFree(d); set_error(std::string("H2D counts: ") + hipGetErrorString(e)); return TRACS_E_HIP; }
    const int rc = tracs_find_dirichlet_priors_device(d, L, K, max_iter, tol, method, error_filt_threshold, alphas_out, iters_out, nullptr);
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
