// capi.hip -- library-wide state (error slot, scratch pool) and the host-level pairsnp entry.
//
// tracs_pairsnp restates the DRIVER part of /root/reference/src/pairsnp.hpp:320-457:
// argument check (:340-343), one- vs two-file pair ranges (:352-360), the output tuple (:451-457).
// The arithmetic is in pairsnp.hip.
#include "common.h"
#include "fasta.h"
#include "rowwriter.h"

#include <algorithm>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace tracs {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

struct Scratch { void *ptr = nullptr; size_t bytes = 0; };
constexpr int kScratchSlots = 96;
static Scratch g_scratch[8][kScratchSlots];
static std::mutex g_scratch_mu;

int workspace_get(int slot, size_t bytes, void **out)
{
    int dev = 0;
    TRACS_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 8 || slot < 0 || slot >= kScratchSlots) { set_error("workspace_get: bad device/slot"); return TRACS_E_ARG; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch &s = g_scratch[dev][slot];
    if (s.bytes < bytes) {
        if (s.ptr) { TRACS_HIP_CHECK(hipDeviceSynchronize()); TRACS_HIP_CHECK(hipFree(s.ptr)); s.ptr = nullptr; s.bytes = 0; }
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&s.ptr, want);
        if (e != hipSuccess) { s.ptr = nullptr; set_error(std::string("hipMalloc(workspace): ") + hipGetErrorString(e)); return TRACS_E_NOMEM; }
        s.bytes = want;
    }
    *out = s.ptr;
    return TRACS_OK;
}

static std::recursive_mutex g_call_mu[8];
static hipStream_t g_last_stream[8];
static bool g_have_stream[8];
static bool g_caller_orders_streams = false;     // tracs_set_stream_policy

DeviceCall::DeviceCall(hipStream_t stream) : dev(0)
{
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 8) dev = 0;
    g_call_mu[dev].lock();
    if (g_have_stream[dev] && g_last_stream[dev] != stream && !g_caller_orders_streams) (void)hipDeviceSynchronize();
    g_last_stream[dev] = stream;
    g_have_stream[dev] = true;
}
DeviceCall::~DeviceCall() { g_call_mu[dev].unlock(); }

void workspace_release_all()
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &dev : g_scratch)
        for (auto &s : dev)
            if (s.ptr) { (void)hipFree(s.ptr); s.ptr = nullptr; s.bytes = 0; }
}

}  // namespace tracs

using namespace tracs;

// TRACS_STAGE_TRACE=1: wall time of the stages of the host-level entry points on stderr, one "[stage] name seconds" line each
// (scripts/bench_e2e.py collects them into the end-to-end table of DESIGN.md 5)
struct StageClock {
    bool on;
    std::chrono::steady_clock::time_point t;
    StageClock() : on(std::getenv("TRACS_STAGE_TRACE") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char *name, double bytes = 0.0)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(now - t).count();
        if (bytes > 0.0) std::fprintf(stderr, "[stage] %s %.4f s (%.2f GB/s)\n", name, s, bytes / s / 1e9);
        else std::fprintf(stderr, "[stage] %s %.4f s\n", name, s);
        t = std::chrono::steady_clock::now();
    }
};

// dst[base + t] = src[t] (uint32 -> uint64) on several host threads: at 5 x 10^7 pairs the five result columns are 2 GB
static void widen_append(std::vector<uint64_t> &dst, const unsigned *src, size_t count)
{
    const size_t base = dst.size();
    dst.resize(base + count);
    uint64_t *out = dst.data() + base;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t nthr = count < (1u << 20) ? 1 : hw;
    if (nthr == 1) { for (size_t t = 0; t < count; t++) out[t] = src[t]; return; }
    std::vector<std::thread> pool;
    const size_t per = (count + nthr - 1) / nthr;
    for (size_t k = 0; k < nthr; k++) {
        const size_t b = k * per, e = std::min(count, b + per);
        if (b < e) pool.emplace_back([=]() { for (size_t t = b; t < e; t++) out[t] = src[t]; });
    }
    for (auto &th : pool) th.join();
}

// Ctrl-C during tracs_pairsnp (src/pairsnp.hpp:21-25,326,385-388,434-441: the reference installs its own SIGINT handler, lets
// the loop drain, prints "Interrupted by user!" and exit(1)s).  Here the handler only lives for the duration of the call, the
// panel loop looks at the flag between row panels (a running kernel cannot be stopped: <= one panel, ~0.4 s at 10 000 x 5 Mbp),
// and the call returns TRACS_E_INTERRUPTED with the reference's message; the Python layer raises KeyboardInterrupt.
static volatile sig_atomic_t g_sigint = 0;
static void on_sigint(int) { g_sigint = 1; }
struct SigintScope {
    struct sigaction old;
    bool installed;
    SigintScope()
    {
        g_sigint = 0;
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_sigint;
        sigemptyset(&sa.sa_mask);
        installed = sigaction(SIGINT, &sa, &old) == 0;
    }
    ~SigintScope() { if (installed) (void)sigaction(SIGINT, &old, nullptr); }
};

struct tracs_pairsnp_result {
    size_t nseq = 0, L = 0;
    std::vector<uint64_t> rows, cols, dist, filt, ncomp;
    std::vector<std::string> names;
};

extern "C" {

void tracs_set_stream_policy(int caller_orders_streams) { g_caller_orders_streams = caller_orders_streams != 0; }

const char *tracs_last_error(void) { return g_error.c_str(); }
int tracs_abi_version(void) { return 1; }

int tracs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void tracs_free(void *p) { std::free(p); }

// host-only: parse a FASTA and return record count, length and an FNV-1a hash over names and sequences (tests)
int tracs_debug_read_fasta(const char *path, size_t *n, size_t *L, uint64_t *hash)
{
    FastaData fd;
    std::string err;
    const int rc = read_fasta(path, fd, err);
    if (rc) { set_error(err); return rc; }
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const uint8_t *b, size_t k) { for (size_t i = 0; i < k; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    if (hash) {
        for (auto &nm : fd.names) { mix(reinterpret_cast<const uint8_t *>(nm.data()), nm.size()); const uint8_t z = 0; mix(&z, 1); }
        mix(fd.seq.data(), fd.seq.size());
    }
    if (n) *n = fd.n;
    if (L) *L = fd.L;
    if (hash) *hash = h;
    return TRACS_OK;
}

// The start-up of a process's first call, off the caller's thread: the HIP runtime, the device context and this library's code
// object (loaded on its first kernel launch: tens of milliseconds for ~2 MB of kernels).  `tracs distance` calls it before it reads
// its metadata and the FASTA, so a small alignment does not wait for them in tracs_distance_open.  Only with ONE visible device:
// a fresh thread's current device is 0, which is not necessarily the caller's.
__global__ void warm_up_kernel(unsigned *p) { if (p) *p = 0u; }
static void warm_up_body()
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd != 1) { (void)hipGetLastError(); return; }
    (void)hipFree(nullptr);
    hipLaunchKernelGGL(warm_up_kernel, dim3(1), dim3(1), 0, nullptr, (unsigned *)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
}
extern "C" void tracs_warm_up(void)
{
    static std::once_flag once;
    std::call_once(once, [] { std::thread(warm_up_body).detach(); });
}

int tracs_alignment_from_fasta(const char *const *fasta, int n_fasta, tracs_alignment **out, char **names_out,
                               size_t *names_bytes, size_t *n_first_file)
{
    if (out) *out = nullptr;
    if (names_out) *names_out = nullptr;
    if (!fasta || !out || n_fasta < 1 || n_fasta > 2) { set_error("Invalid number of fasta files!"); return TRACS_E_ARG; }
    FastaData fd;
    size_t n0 = 0;
    StageClock clock;
    // the HIP runtime comes up (first call of the process: ~0.3-0.5 s) while the host threads read the text
    // (a context only where there is one device to have it on: with several visible, a fresh thread's current device is 0, not
    // necessarily the caller's -- a rank of a multi-GPU job would leave a primary context on GPU 0 --; the runtime itself still comes up)
    std::thread warm(warm_up_body);
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } join_warm{warm};
    for (int f = 0; f < n_fasta; f++) {
        std::string err;
        FastaData one;
        int rc = read_fasta(fasta[f], one, err);
        if (rc) { set_error(err); return rc; }
        if (f == 0) { fd = std::move(one); n0 = fd.n; }
        else {
            // load_seqs only checks lengths inside one file (pairsnp.hpp:94-98); two files of different
            // length would make the reference AND bitsets of different sizes (undefined) -- refuse.
            if (fd.n && one.n && one.L != fd.L) { set_error("Error reading FASTA, variable sequence lengths!"); return TRACS_E_RAGGED; }
            if (!fd.n) fd.L = one.L;
            fd.seq.insert(fd.seq.end(), one.seq.begin(), one.seq.end());
            fd.names.insert(fd.names.end(), one.names.begin(), one.names.end());
            fd.n += one.n;
        }
    }
    if (warm.joinable()) warm.join();
    clock.mark("read FASTA (host)", (double)fd.n * (double)fd.L);
    tracs_alignment *a = nullptr;
    int rc = tracs_alignment_create(fd.n, fd.L, &a);
    if (rc) return rc;
    clock.mark("allocate planes + arena");
    // pack in sample batches of <= 1 GiB of ASCII
    const size_t batch = fd.L ? std::max<size_t>(1, (1ull << 30) / fd.L) : fd.n;
    for (size_t s = 0; s < fd.n && fd.L; s += batch) {
        const size_t cnt = std::min(batch, fd.n - s);
        rc = tracs_alignment_pack(a, fd.seq.data() + s * fd.L, s, cnt, 0, nullptr);
        if (rc) { tracs_alignment_free(a); return rc; }
    }
    clock.mark("H2D + pack", (double)fd.n * (double)fd.L);
    if (names_out) {
        size_t bytes = 0;
        for (auto &nm : fd.names) bytes += nm.size() + 1;
        char *blk = static_cast<char *>(std::malloc(bytes ? bytes : 1));
        size_t o = 0;
        for (auto &nm : fd.names) { std::memcpy(blk + o, nm.c_str(), nm.size() + 1); o += nm.size() + 1; }
        *names_out = blk;
        if (names_bytes) *names_bytes = bytes;
    }
    if (n_first_file) *n_first_file = n0;
    *out = a;
    return TRACS_OK;
}

int tracs_pairsnp(const char *const *fasta, int n_fasta, int n_threads, int dist, int filter, tracs_pairsnp_result **out)
{
    (void)n_threads;
    if (!out) { set_error("tracs_pairsnp: out is NULL"); return TRACS_E_ARG; }
    *out = nullptr;
    if (n_fasta < 1 || n_fasta > 2 || !fasta) { set_error("Invalid number of fasta files!"); return TRACS_E_ARG; }   // :340-343
    SigintScope sigint;
    tracs_alignment *a = nullptr;
    char *names = nullptr;
    size_t names_bytes = 0, n0 = 0;
    int rc = tracs_alignment_from_fasta(fasta, n_fasta, &a, &names, &names_bytes, &n0);
    if (rc) return rc;
    auto *res = new tracs_pairsnp_result();
    res->nseq = a->n; res->L = a->L;
    { size_t o = 0; for (size_t i = 0; i < a->n; i++) { res->names.emplace_back(names + o); o += res->names.back().size() + 1; } }
    tracs_free(names);
    // pair ranges (:348-360)
    const size_t n = a->n;
    const size_t i_end = n_fasta == 1 ? n : n0;
    const size_t j_start = n_fasta == 1 ? 0 : n0;

    unsigned *d_dist = nullptr, *d_nn = nullptr, *d_rows = nullptr, *d_cols = nullptr, *d_d = nullptr, *d_n = nullptr;
    unsigned *d_filt = nullptr;
    long long *d_off = nullptr;
    size_t pair_cap = 0;
    auto cleanup = [&]() {
        void *p[] = {d_dist, d_nn, d_rows, d_cols, d_d, d_n, d_off, d_filt};
        for (void *q : p) if (q) (void)hipFree(q);
        tracs_alignment_free(a);
    };
#define PS_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); delete res; set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
#define PS_RC(x) do { int r__ = (x); if (r__) { cleanup(); delete res; return r__; } } while (0)
    StageClock clock;
    double t_dense = 0.0, t_coo = 0.0, t_pull = 0.0, t_filter = 0.0;
    auto lap = [&](double &acc) {
        if (!clock.on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double>(now - clock.t).count();
        clock.t = now;
    };
    if (n >= 2 && i_end > 0) {
        // row panels bounded to ~1 GiB per dense matrix
        const size_t panel = std::max<size_t>(64, std::min<size_t>(i_end, (1ull << 28) / std::max<size_t>(n, 1)));
        PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_dist), panel * n * 4));
        PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_nn), panel * n * 4));
        PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_off), (panel + 1) * 8));
        size_t cap = 0;
        std::vector<unsigned> h32;
        for (size_t r0 = 0; r0 < i_end; r0 += panel) {
            if (g_sigint) { cleanup(); delete res; set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
            const size_t r1 = std::min(i_end, r0 + panel);
            // the dense block is addressed as base[(i) * ld + j] with i absolute: shift the base
            unsigned *bd = d_dist - r0 * n, *bn = d_nn - r0 * n;
            lap(t_pull);
            PS_RC(tracs_pairsnp_dense_thr(a, r0, r1, j_start, bd, bn, n, dist, nullptr));   // early out beyond `dist`
            lap(t_dense);
            PS_RC(tracs_coo_count(bd, n, n, r0, r1, j_start, dist, reinterpret_cast<int64_t *>(d_off), nullptr));
            long long total = 0;
            PS_CHECK(hipMemcpy(&total, d_off + (r1 - r0), 8, hipMemcpyDeviceToHost));
            if (total <= 0) continue;
            if ((size_t)total > cap) {
                void *p[] = {d_rows, d_cols, d_d, d_n};
                for (void *q : p) if (q) PS_CHECK(hipFree(q));
                d_rows = d_cols = d_d = d_n = nullptr;
                cap = (size_t)total;
                PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_rows), cap * 4));
                PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_cols), cap * 4));
                PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_d), cap * 4));
                PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_n), cap * 4));
            }
            PS_RC(tracs_coo_fill(bd, bn, n, n, r0, r1, j_start, dist, reinterpret_cast<int64_t *>(d_off), d_rows, d_cols, d_d, d_n, nullptr));
            lap(t_coo);
            h32.resize((size_t)total);
            auto pull = [&](unsigned *src, std::vector<uint64_t> &dst) -> hipError_t {
                hipError_t e = hipMemcpy(h32.data(), src, (size_t)total * 4, hipMemcpyDeviceToHost);
                if (e != hipSuccess) return e;
                widen_append(dst, h32.data(), (size_t)total);
                return hipSuccess;
            };
            PS_CHECK(pull(d_rows, res->rows));
            PS_CHECK(pull(d_cols, res->cols));
            PS_CHECK(pull(d_d, res->dist));
            PS_CHECK(pull(d_n, res->ncomp));
            lap(t_pull);
            if (filter) {
                // recombination filter on the pairs just emitted (:405-413): SNP sites from the samples' departure lists
                if (g_sigint) { cleanup(); delete res; set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
                if ((size_t)total > pair_cap) {
                    if (d_filt) PS_CHECK(hipFree(d_filt));
                    d_filt = nullptr;
                    pair_cap = (size_t)total + (size_t)total / 4 + 16;
                    PS_CHECK(hipMalloc(reinterpret_cast<void **>(&d_filt), pair_cap * 4));
                }
                PS_RC(tracs_filter_recomb_pairs(a, d_rows, d_cols, d_d, (size_t)total, d_filt, nullptr));
                PS_CHECK(hipMemcpy(h32.data(), d_filt, (size_t)total * 4, hipMemcpyDeviceToHost));
                widen_append(res->filt, h32.data(), (size_t)total);
                lap(t_filter);
            }
        }
    }
    if (clock.on)
        std::fprintf(stderr, "[stage] dense panels (once-per-pack work + pair kernels) %.4f s\n[stage] COO extraction (device) %.4f s\n"
                             "[stage] COO D2H + widening to uint64 (%zu pairs) %.4f s\n[stage] recombination filter %.4f s\n",
                     t_dense, t_coo, res->rows.size(), t_pull, t_filter);
#undef PS_CHECK
#undef PS_RC
    // a Ctrl-C that arrived during the last panel / filter batch is not swallowed (the reference looks at its flag on every row)
    if (g_sigint) { cleanup(); delete res; set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
    if (!filter) res->filt.assign(res->rows.size(), 0);      // filter off: `len` zeros (:452 via combine_vectors :31)
    cleanup();
    *out = res;
    return TRACS_OK;
}

// ---- `tracs distance` for one alignment, results on the device until the CSV rows (tracs/distance.py:159-258) --------------------------
// open: read + pack the FASTA(s) (the names are what the caller needs to look the sampling dates up); run: row panel by row panel --
// dense call (d, compared sites; early out beyond the threshold), transcluster on the panel (P(direct), E(K): src/transcluster.hpp:
// 240-287 through the dense route, delta from the day numbers), the pairs within the threshold extracted in row-major order
// (src/pairsnp.hpp:451-455) with their P and E(K) -- then ONE device-to-host pass, a few million rows at a time, each batch formatted
// and written by the host threads while the next one arrives.  Nothing widens to uint64, nothing goes back to the device.
struct tracs_distance {
    tracs_alignment *a = nullptr;
    std::vector<std::string> names;
    std::vector<const char *> name_ptr;
    size_t n0 = 0;
    int n_fasta = 0;
};

int tracs_distance_open(const char *const *fasta, int n_fasta, tracs_distance **out)
{
    if (!out) { set_error("tracs_distance_open: out is NULL"); return TRACS_E_ARG; }
    *out = nullptr;
    if (n_fasta < 1 || n_fasta > 2 || !fasta) { set_error("Invalid number of fasta files!"); return TRACS_E_ARG; }   // :340-343
    auto *h = new tracs_distance();
    char *names = nullptr;
    size_t names_bytes = 0;
    const int rc = tracs_alignment_from_fasta(fasta, n_fasta, &h->a, &names, &names_bytes, &h->n0);
    if (rc) { delete h; return rc; }
    h->n_fasta = n_fasta;
    { size_t o = 0; for (size_t i = 0; i < h->a->n; i++) { h->names.emplace_back(names + o); o += h->names.back().size() + 1; } }
    tracs_free(names);
    for (auto &s : h->names) h->name_ptr.push_back(s.c_str());
    *out = h;
    return TRACS_OK;
}

size_t tracs_distance_nseq(const tracs_distance *h) { return h ? h->names.size() : 0; }
const char *tracs_distance_name(const tracs_distance *h, size_t i) { return (h && i < h->names.size()) ? h->names[i].c_str() : nullptr; }
void tracs_distance_free(tracs_distance *h) { if (h) { if (h->a) tracs_alignment_free(h->a); delete h; } }

// the date difference of every emitted pair, as tracs/transcluster.py:26-33 takes it: |t_i - t_j| / 31556952.0 with t = whole days in seconds
__global__ __launch_bounds__(256) void coo_delta_kernel(const unsigned *__restrict__ rows, const unsigned *__restrict__ cols, const int *__restrict__ days,
                                                        size_t n, double *__restrict__ out)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const long long dd = (long long)days[rows[t]] - (long long)days[cols[t]];
        out[t] = (double)((dd < 0 ? -dd : dd) * 86400ll) / 31556952.0;
    }
}

int tracs_distance_run(tracs_distance *h, int dist, const int32_t *days, double lamb, double beta, double precision, double k_max,
                       const char *path, const char *ref, int filter, uint64_t *rows_written, uint64_t *n_pairs)
{
    if (rows_written) *rows_written = 0;
    if (n_pairs) *n_pairs = 0;
    if (!h || !h->a || !path || !ref) { set_error("tracs_distance_run: NULL argument"); return TRACS_E_ARG; }
    SigintScope sigint;
    tracs_alignment *a = h->a;
    const size_t n = a->n;
    const size_t i_end = h->n_fasta == 1 ? n : h->n0;               // pair ranges (:348-360)
    const size_t j_start = h->n_fasta == 1 ? 0 : h->n0;
    const bool with_dates = days != nullptr;
    // rows per device-to-host batch (TRACS_DISTANCE_BATCH_ROWS: diagnostics -- small batches in tests)
    static const size_t CH_MAX = [] { const char *e = std::getenv("TRACS_DISTANCE_BATCH_ROWS"); const long long v = e ? std::atoll(e) : 0; return v >= 16 ? (size_t)v : (size_t)1 << 22; }();
    // (never more than the pairs there can be: ten isolates do not pin a quarter of a gigabyte of host memory)
    const size_t CH = std::max<size_t>(64, std::min<size_t>(CH_MAX, (h->n_fasta == 1 ? h->a->n : h->n0) * h->a->n));
    unsigned *d_dist = nullptr, *d_nn = nullptr, *d_coo = nullptr;
    double *d_p = nullptr, *d_e = nullptr, *d_cp = nullptr;
    const bool dense_tc = with_dates && !filter;                   // --filter: the transmission model is driven by the FILTERED distance
                                                                    // (tracs/distance.py:183-193): per emitted pair, after the filter
    int *d_days = nullptr;
    long long *d_off = nullptr;
    char *pin[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr}, ready = nullptr;
    tracs::DistanceRowWriter writer;
    auto cleanup = [&]() {
        void *p[] = {d_dist, d_nn, d_coo, d_p, d_e, d_cp, d_days, d_off};
        for (void *q : p) if (q) (void)hipFree(q);
        for (char *q : pin) if (q) (void)hipHostFree(q);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (ready) (void)hipEventDestroy(ready);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    };
#define DR_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
#define DR_RC(x) do { int r__ = (x); if (r__) { cleanup(); return r__; } } while (0)
    DR_RC(writer.open(path, h->name_ptr.data(), h->name_ptr.size(), ref));
    StageClock clock;
    double t_dense = 0.0, t_tc = 0.0, t_coo = 0.0, t_rows = 0.0;
    auto lap = [&](double &acc) {
        if (!clock.on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double>(now - clock.t).count();
        clock.t = now;
    };
    uint64_t pairs = 0;
    if (n >= 2 && i_end > 0) {
        // row panels bounded to ~1 GiB per uint32 matrix (2 GiB per f64 one)
        const size_t panel = std::max<size_t>(64, std::min<size_t>(i_end, (1ull << 28) / std::max<size_t>(n, 1)));
        DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_dist), panel * n * 4));
        DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_nn), panel * n * 4));
        DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_off), (panel + 1) * 8));
        if (dense_tc) {
            DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_p), panel * n * 8));
            DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_e), panel * n * 8));
        }
        if (with_dates) {
            DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_days), n * 4));
            DR_CHECK(hipMemcpy(d_days, days, n * 4, hipMemcpyHostToDevice));
        }
        // a batch on the host: four uint32 columns (five with --filter), then two f64 ones; two of them, so that one is formatted while the
        // next arrives
        const size_t n32 = filter ? 5 : 4;
        const size_t batch_bytes = CH * (n32 * 4 + (with_dates ? 16 : 0) + 4);
        for (auto &q : pin) DR_CHECK(hipHostMalloc(reinterpret_cast<void **>(&q), batch_bytes, hipHostMallocDefault));
        DR_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (auto &e : ev) DR_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DR_CHECK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        std::vector<uint32_t> zeros;                                  // the filtered column without metadata: `len` zeros (:240-258)
        if (!with_dates) zeros.assign(CH, 0u);
        size_t cap = 0;
        for (size_t r0 = 0; r0 < i_end; r0 += panel) {
            if (g_sigint) { cleanup(); set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
            const size_t r1 = std::min(i_end, r0 + panel);
            unsigned *bd = d_dist - r0 * n, *bn = d_nn - r0 * n;     // addressed as base[i * ld + j] with i absolute
            double *bp = dense_tc ? d_p - r0 * n : nullptr, *be = dense_tc ? d_e - r0 * n : nullptr;
            DR_RC(tracs_pairsnp_dense_thr(a, r0, r1, j_start, bd, bn, n, dist, nullptr));
            lap(t_dense);
            if (dense_tc)
                DR_RC(tracs_trans_dist_dense(bd, n, n, r0, r1, j_start, dist, d_days, lamb, beta, precision, 1, bp, be, nullptr));
            lap(t_tc);
            DR_RC(tracs_coo_count(bd, n, n, r0, r1, j_start, dist, reinterpret_cast<int64_t *>(d_off), nullptr));
            long long total = 0;
            DR_CHECK(hipMemcpy(&total, d_off + (r1 - r0), 8, hipMemcpyDeviceToHost));
            if (total <= 0) continue;
            if ((size_t)total > cap) {
                if (d_coo) DR_CHECK(hipFree(d_coo));
                if (d_cp) DR_CHECK(hipFree(d_cp));
                d_coo = nullptr; d_cp = nullptr;
                cap = (size_t)total;
                DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_coo), cap * 4 * n32));
                if (with_dates) DR_CHECK(hipMalloc(reinterpret_cast<void **>(&d_cp), cap * (filter ? 24 : 16)));
            }
            unsigned *c_rows = d_coo, *c_cols = d_coo + cap, *c_d = d_coo + 2 * cap, *c_n = d_coo + 3 * cap, *c_f = filter ? d_coo + 4 * cap : nullptr;
            double *c_p = d_cp, *c_e = with_dates ? d_cp + cap : nullptr, *c_delta = (with_dates && filter) ? d_cp + 2 * cap : nullptr;
            DR_RC(tracs_coo_fill(bd, bn, n, n, r0, r1, j_start, dist, reinterpret_cast<int64_t *>(d_off), c_rows, c_cols, c_d, c_n, nullptr));
            if (dense_tc)
                DR_RC(tracs_coo_fill_f64(bd, n, n, r0, r1, j_start, dist, reinterpret_cast<int64_t *>(d_off), bp, be, c_p, c_e, nullptr));
            if (filter) {
                // the recombination filter on the emitted pairs (src/pairsnp.hpp:405-413), then -- with dates -- P(direct) and E(K) of the
                // filtered distances, pair by pair (tracs/distance.py:183-193 -> tracs/transcluster.py:8-41 -> trans_dist)
                DR_RC(tracs_filter_recomb_pairs(a, c_rows, c_cols, c_d, (size_t)total, c_f, nullptr));
                if (with_dates) {
                    hipLaunchKernelGGL(coo_delta_kernel, dim3((unsigned)std::min<size_t>(((size_t)total + 255) / 256, 65535)), dim3(256), 0, nullptr, c_rows, c_cols,
                                       d_days, (size_t)total, c_delta);
                    DR_RC(tracs_trans_dist_device(reinterpret_cast<const int32_t *>(c_f), c_delta, (size_t)total, lamb, beta, precision, 1, c_p, c_e, nullptr));
                }
                lap(t_tc);
            }
            DR_CHECK(hipEventRecord(ready, nullptr));
            DR_CHECK(hipStreamWaitEvent(copy_stream, ready, 0));
            lap(t_coo);
            const size_t nb = ((size_t)total + CH - 1) / CH;
            auto post = [&](size_t k) -> hipError_t {                 // batch k -> pinned set k % 2
                const size_t o = k * CH, cnt = std::min(CH, (size_t)total - o);
                char *dst = pin[k & 1];
                const unsigned *src32[5] = {c_rows, c_cols, c_d, c_n, c_f};
                for (size_t q = 0; q < n32; q++) {
                    const hipError_t e = hipMemcpyAsync(dst + (size_t)q * CH * 4, src32[q] + o, cnt * 4, hipMemcpyDeviceToHost, copy_stream);
                    if (e != hipSuccess) return e;
                }
                if (with_dates) {
                    const size_t f64_at = (CH * n32 * 4 + 7) / 8 * 8;
                    hipError_t e = hipMemcpyAsync(dst + f64_at, c_p + o, cnt * 8, hipMemcpyDeviceToHost, copy_stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(dst + f64_at + CH * 8, c_e + o, cnt * 8, hipMemcpyDeviceToHost, copy_stream);
                    if (e != hipSuccess) return e;
                }
                return hipEventRecord(ev[k & 1], copy_stream);
            };
            DR_CHECK(post(0));
            for (size_t k = 0; k < nb; k++) {
                if (g_sigint) { (void)hipStreamSynchronize(copy_stream); cleanup(); set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
                DR_CHECK(hipEventSynchronize(ev[k & 1]));
                if (k + 1 < nb) DR_CHECK(post(k + 1));
                const size_t cnt = std::min(CH, (size_t)total - k * CH);
                const char *src = pin[k & 1];
                const uint32_t *hr = reinterpret_cast<const uint32_t *>(src), *hc = hr + CH, *hd = hr + 2 * CH, *hn = hr + 3 * CH;
                const size_t f64_at = (CH * n32 * 4 + 7) / 8 * 8;
                const double *hp = reinterpret_cast<const double *>(src + f64_at), *he = hp + CH;
                // the filtered column: --filter: the filtered distances; else metadata on: a column of "NA" (:204), metadata off: zeros (:240-258)
                const uint32_t *hf = filter ? hr + 4 * CH : (with_dates ? nullptr : zeros.data());
                DR_RC(writer.append_u32(hr, hc, hd, hf, hn, days, with_dates ? hp : nullptr, with_dates ? he : nullptr,
                                        cnt, with_dates ? 1 : 0, with_dates ? k_max : -1.0));
            }
            pairs += (uint64_t)total;
            lap(t_rows);
        }
    }
    if (clock.on)
        std::fprintf(stderr, "[stage] dense panels (once-per-pack work + pair kernels) %.4f s\n[stage] transcluster on the panels (device) %.4f s\n"
                             "[stage] COO extraction (device) %.4f s\n[stage] rows: device -> host, format, write (%llu pairs) %.4f s\n",
                     t_dense, t_tc, t_coo, (unsigned long long)pairs, t_rows);
#undef DR_CHECK
#undef DR_RC
    cleanup();
    const int rc = writer.close();
    if (rows_written) *rows_written = writer.written();
    if (n_pairs) *n_pairs = pairs;
    if (g_sigint) { set_error("Interrupted by user!"); return TRACS_E_INTERRUPTED; }
    return rc;
}

size_t tracs_pairsnp_len(const tracs_pairsnp_result *r) { return r ? r->rows.size() : 0; }
size_t tracs_pairsnp_nseq(const tracs_pairsnp_result *r) { return r ? r->nseq : 0; }
size_t tracs_pairsnp_seqlen(const tracs_pairsnp_result *r) { return r ? r->L : 0; }
const uint64_t *tracs_pairsnp_rows(const tracs_pairsnp_result *r) { return r ? r->rows.data() : nullptr; }
const uint64_t *tracs_pairsnp_cols(const tracs_pairsnp_result *r) { return r ? r->cols.data() : nullptr; }
const uint64_t *tracs_pairsnp_distances(const tracs_pairsnp_result *r) { return r ? r->dist.data() : nullptr; }
const uint64_t *tracs_pairsnp_filt_distances(const tracs_pairsnp_result *r) { return r ? r->filt.data() : nullptr; }
const uint64_t *tracs_pairsnp_ncompared(const tracs_pairsnp_result *r) { return r ? r->ncomp.data() : nullptr; }
const char *tracs_pairsnp_name(const tracs_pairsnp_result *r, size_t i) { return (r && i < r->names.size()) ? r->names[i].c_str() : nullptr; }
void tracs_pairsnp_free(tracs_pairsnp_result *r) { delete r; }

}  // extern "C"
