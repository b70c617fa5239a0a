// alignio.cpp -- the on-disk formats either side of the distance path, host side:
//   tracs_pileup_counts        `htsbox pileup -C -s 0` text(.gz) -> allele counts [L,4]      (tracs/align.py:444-475)
//   tracs_write_posterior_csv  posterior [L,4] -> <prefix>_posterior_counts_ref_<ref>.csv.gz   (tracs/align.py:580-596)
//   tracs_combine_fasta        per-sample posterior FASTA files -> <ref>_combined.fasta.gz    (tracs/combine.py:220-239)
// Own implementation; the behaviour (field positions, what is skipped, what overwrites what) follows the lines cited.
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "fasta.h"
#include "rowwriter.h"

namespace tracs {
void set_error(const std::string &msg);
}
using tracs::set_error;

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }   // str.split() separators (ASCII)

struct Field { const char *p; size_t n; };

// Python's int() on a short ASCII token: optional sign, digits, optional single underscores between digits are NOT
// accepted here (htsbox never writes them).  Returns false on anything else.
bool parse_int(const char *p, size_t n, long long &out)
{
    if (!n) return false;
    size_t i = 0;
    bool neg = false;
    if (p[0] == '+' || p[0] == '-') { neg = p[0] == '-'; i = 1; }
    if (i >= n || n - i > 18) return false;
    long long v = 0;
    for (; i < n; i++) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (p[i] - '0');
    }
    out = neg ? -v : v;
    return true;
}

inline int base_index(const char *p, size_t n)      // the reference's npos dict: exactly "A","C","G","T" (align.py:445)
{
    if (n != 1) return -1;
    switch (p[0]) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// split [p, p+n) at `sep`, like str.split(sep): always at least one (possibly empty) piece
void split_char(const char *p, size_t n, char sep, std::vector<Field> &out)
{
    out.clear();
    const char *s = p, *e = p + n;
    for (const char *q = p; q < e; q++)
        if (*q == sep) { out.push_back({s, (size_t)(q - s)}); s = q + 1; }
    out.push_back({s, (size_t)(e - s)});
}

// str(float) / str(numpy.float64) as CPython and numpy print them (float_repr_style "short"): the shortest digit string
// that round-trips, positional notation when 1e-4 <= |x| < 1e16, otherwise d[.ddd]e[+-]XX; "inf", "nan", "-0.0".
size_t format_py_float(double x, char *out)
{
    if (std::isnan(x)) { std::memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { const char *t = x < 0 ? "-inf" : "inf"; const size_t k = std::strlen(t); std::memcpy(out, t, k); return k; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);     // [-]d[.ddd]e[+-]XX, shortest
    const char *p = sci, *end = r.ptr;
    char *o = out;
    if (*p == '-') { *o++ = '-'; p++; }
    char digits[24];
    int nd = 0;
    const char *e = p;
    while (e < end && *e != 'e') { if (*e != '.') digits[nd++] = *e; e++; }
    int exp10 = 0;
    {
        const char *q = e + 1;
        const bool neg = *q == '-';
        if (*q == '+' || *q == '-') q++;
        while (q < end) exp10 = exp10 * 10 + (*q++ - '0');
        if (neg) exp10 = -exp10;
    }
    if (exp10 < -4 || exp10 >= 16) {                      // exponent form, at least two exponent digits
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; std::memcpy(o, digits + 1, (size_t)nd - 1); o += nd - 1; }
        *o++ = 'e';
        *o++ = exp10 < 0 ? '-' : '+';
        const int a = exp10 < 0 ? -exp10 : exp10;
        if (a >= 100) { *o++ = (char)('0' + a / 100); *o++ = (char)('0' + a / 10 % 10); *o++ = (char)('0' + a % 10); }
        else { *o++ = (char)('0' + a / 10); *o++ = (char)('0' + a % 10); }
    } else if (exp10 >= 0) {                              // ddd[.ddd] or ddd000.0
        const int ip = exp10 + 1;                         // digits before the point
        if (nd <= ip) {
            std::memcpy(o, digits, (size_t)nd); o += nd;
            for (int k = nd; k < ip; k++) *o++ = '0';
            *o++ = '.'; *o++ = '0';
        } else {
            std::memcpy(o, digits, (size_t)ip); o += ip;
            *o++ = '.';
            std::memcpy(o, digits + ip, (size_t)(nd - ip)); o += nd - ip;
        }
    } else {                                              // 0.000ddd
        *o++ = '0'; *o++ = '.';
        for (int k = 0; k < -exp10 - 1; k++) *o++ = '0';
        std::memcpy(o, digits, (size_t)nd); o += nd;
    }
    return (size_t)(o - out);
}

size_t format_u64(uint64_t v, char *out)
{
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < k; i++) out[i] = tmp[k - 1 - i];
    return (size_t)k;
}

}  // namespace

extern "C" {

int tracs_pileup_counts(const char *path, const char *const *contig_names, const uint64_t *contig_lengths, size_t n_contigs,
                        int require_both_strands, double *counts, uint64_t *n_lines_out)
{
    if (!path || (!contig_names && n_contigs) || (!contig_lengths && n_contigs) || !counts) {
        set_error("tracs_pileup_counts: NULL argument");
        return TRACS_E_ARG;
    }
    std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> where;      // name -> (first row, length)
    uint64_t total = 0;
    for (size_t c = 0; c < n_contigs; c++) {
        if (!where.emplace(contig_names[c], std::make_pair(total, contig_lengths[c])).second) {
            set_error(std::string("tracs_pileup_counts: duplicate contig name '") + contig_names[c] + "'");
            return TRACS_E_ARG;
        }
        total += contig_lengths[c];
    }
    std::memset(counts, 0, (size_t)total * 4 * sizeof(double));                 // np.zeros((len(seq), 4)), align.py:449-450
    gzFile f = gzopen(path, "rb");
    if (!f) { set_error(std::string("cannot open '") + path + "'"); return TRACS_E_OPEN; }
    gzbuffer(f, 1u << 20);

    std::vector<char> buf(8u << 20);
    size_t have = 0;
    bool eof = false;
    uint64_t line_no = 0;
    std::vector<Field> fields, nucs, parts, fwd, rev;
    std::string last_contig;
    std::pair<uint64_t, uint64_t> last_where{0, 0};
    int rc = TRACS_OK;
    auto fail = [&](const std::string &what) {
        set_error("pileup line " + std::to_string(line_no) + ": " + what);
        rc = TRACS_E_FASTA;
    };
    auto do_line = [&](const char *p, size_t n) {
        line_no++;
        // line.strip().split(): whitespace-separated fields (align.py:456)
        fields.clear();
        size_t i = 0;
        while (i < n) {
            while (i < n && is_space((unsigned char)p[i])) i++;
            if (i >= n) break;
            const size_t s = i;
            while (i < n && !is_space((unsigned char)p[i])) i++;
            fields.push_back({p + s, i - s});
        }
        if (fields.size() < 3) { fail("fewer than 3 fields"); return; }
        const Field &contig = fields[0];
        if (last_contig.size() != contig.n || std::memcmp(last_contig.data(), contig.p, contig.n) != 0) {
            last_contig.assign(contig.p, contig.n);
            auto it = where.find(last_contig);
            if (it == where.end()) { fail("contig '" + last_contig + "' is not in the reference"); return; }
            last_where = it->second;
        }
        long long pos1 = 0;
        if (!parse_int(fields[1].p, fields[1].n, pos1)) { fail("position is not an integer"); return; }
        if (pos1 < 1 || (uint64_t)pos1 > last_where.second) { fail("position outside the contig"); return; }
        const bool ref_ok = base_index(fields[2].p, fields[2].n) >= 0;          // line[2] not in npos -> every allele skipped (:466)
        const Field &fn = fields[fields.size() - 2], &fc = fields[fields.size() - 1];
        split_char(fn.p, fn.n, ',', nucs);                                       // line[-2].split(",")   (:459)
        split_char(fc.p, fc.n, ':', parts);                                      // line[-1].split(":")[1:] (:460)
        if (parts.size() < 3) { fail("allele-count column has fewer than two strand lists"); return; }
        split_char(parts[1].p, parts[1].n, ',', fwd);
        split_char(parts[2].p, parts[2].n, ',', rev);
        double row[4] = {0.0, 0.0, 0.0, 0.0};
        const size_t m = std::min(nucs.size(), std::min(fwd.size(), rev.size()));   // zip() stops at the shortest (:462-464)
        for (size_t k = 0; k < m; k++) {
            long long c1 = 0, c2 = 0;
            if (!parse_int(fwd[k].p, fwd[k].n, c1) || !parse_int(rev[k].p, rev[k].n, c2)) { fail("allele count is not an integer"); return; }
            const int b = base_index(nucs[k].p, nucs[k].n);
            if (b < 0 || !ref_ok) continue;
            if (require_both_strands && (c1 == 0 || c2 == 0)) c1 = c2 = 0;       // :468-470
            row[b] = (double)(c1 + c2);                                          // assignment: a repeated allele overwrites (:471)
        }
        double *dst = counts + (size_t)(last_where.first + (uint64_t)pos1 - 1) * 4;
        dst[0] = row[0]; dst[1] = row[1]; dst[2] = row[2]; dst[3] = row[3];      // a repeated position overwrites (:472)
    };
    while (rc == TRACS_OK) {
        if (!eof) {
            if (have == buf.size()) buf.resize(buf.size() * 2);                  // a line longer than the buffer
            const int r = gzread(f, buf.data() + have, (unsigned)std::min<size_t>(buf.size() - have, 1u << 30));
            if (r < 0) { set_error(std::string("error reading '") + path + "'"); rc = TRACS_E_FASTA; break; }
            if (r == 0) eof = true;
            have += (size_t)r;
        }
        size_t start = 0;
        for (;;) {
            const char *nl = static_cast<const char *>(std::memchr(buf.data() + start, '\n', have - start));
            if (!nl) break;
            do_line(buf.data() + start, (size_t)(nl - (buf.data() + start)));
            start = (size_t)(nl - buf.data()) + 1;
            if (rc != TRACS_OK) break;
        }
        if (rc != TRACS_OK) break;
        std::memmove(buf.data(), buf.data() + start, have - start);
        have -= start;
        if (eof) {
            if (have) do_line(buf.data(), have);                                 // last line without a newline
            break;
        }
    }
    gzclose(f);
    if (n_lines_out) *n_lines_out = line_no;
    return rc;
}

// np.savetxt(fmt="%0.5f", delimiter=",") through gzip, plus the extra "\n" the reference appends (align.py:580-596).
int tracs_write_posterior_csv(const char *path, const double *post, size_t L, size_t K, int gzip_level)
{
    if (!path || (!post && L) || K == 0 || K > 64) { set_error("tracs_write_posterior_csv: bad argument"); return TRACS_E_ARG; }
    char mode[8];
    std::snprintf(mode, sizeof mode, "wb%d", std::max(0, std::min(9, gzip_level)));
    gzFile f = gzopen(path, mode);
    if (!f) { set_error(std::string("cannot open '") + path + "' for writing"); return TRACS_E_OPEN; }
    gzbuffer(f, 1u << 20);
    // rows are formatted in parallel into per-chunk buffers, written in order
    const size_t chunk = 1u << 16;
    const unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    int rc = TRACS_OK;
    for (size_t base = 0; base < L && rc == TRACS_OK; base += chunk * T) {
        std::vector<std::string> out(T);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                const size_t r0 = std::min(L, base + (size_t)t * chunk), r1 = std::min(L, r0 + chunk);
                std::string &s = out[t];
                s.reserve((r1 - r0) * K * 10);
                char tmp[512];
                for (size_t r = r0; r < r1; r++)
                    for (size_t k = 0; k < K; k++) {
                        const int w = std::snprintf(tmp, sizeof tmp, "%0.5f", post[r * K + k]);
                        s.append(tmp, (size_t)w);
                        s.push_back(k + 1 < K ? ',' : '\n');
                    }
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T; t++)
            if (!out[t].empty() && gzwrite(f, out[t].data(), (unsigned)out[t].size()) != (int)out[t].size()) {
                set_error(std::string("error writing '") + path + "'");
                rc = TRACS_E_OPEN;
                break;
            }
    }
    if (rc == TRACS_OK && gzwrite(f, "\n", 1) != 1) { set_error(std::string("error writing '") + path + "'"); rc = TRACS_E_OPEN; }
    if (gzclose(f) != Z_OK && rc == TRACS_OK) { set_error(std::string("error closing '") + path + "'"); rc = TRACS_E_OPEN; }
    return rc;
}

// str(float) memoised over a call: delta, P and E(K) take a few hundred thousand distinct values over tens of millions of rows
// (one per distinct (SNP distance, day gap) key: src/transcluster.hpp:245-246), and Python's shortest round-trip repr costs ten
// times a table look-up.  Open addressing on the bit pattern, lock-free: a slot is claimed (0 -> 1), filled, published (2); a
// reader that finds a slot being filled, or a full neighbourhood, formats for itself.
namespace {
struct FloatMemo {
    struct Slot { std::atomic<uint32_t> state; uint32_t len; uint64_t bits; char str[32]; };
    std::vector<Slot> slots;
    size_t mask;
    explicit FloatMemo(size_t log2_slots) : slots((size_t)1 << log2_slots), mask(((size_t)1 << log2_slots) - 1)
    {
        for (auto &x : slots) x.state.store(0, std::memory_order_relaxed);
    }
    // appends str(x) at out, returns its length
    size_t put(double x, char *out)
    {
        uint64_t bits;
        std::memcpy(&bits, &x, 8);
        uint64_t h = bits * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        for (int probe = 0; probe < 8; probe++) {
            Slot &sl = slots[(h + (uint64_t)probe) & mask];
            uint32_t st = sl.state.load(std::memory_order_acquire);
            if (st == 2u) {
                if (sl.bits == bits) { std::memcpy(out, sl.str, 32); return sl.len; }
                continue;
            }
            if (st == 0u) {
                uint32_t expect = 0u;
                if (sl.state.compare_exchange_strong(expect, 1u, std::memory_order_acq_rel)) {
                    char tmp[64];
                    const size_t len = format_py_float(x, tmp);
                    if (len <= 32) {
                        sl.bits = bits; sl.len = (uint32_t)len;
                        std::memcpy(sl.str, tmp, len);
                        sl.state.store(2u, std::memory_order_release);
                    }                                            // (a longer string never gets published: the slot stays claimed)
                    std::memcpy(out, tmp, len);
                    return len;
                }
            }
            break;                                               // being filled by somebody else: format here
        }
        return format_py_float(x, out);
    }
};
}  // namespace

// Rows of `tracs distance`'s CSV (tracs/distance.py:206-258), appended to `path`:
//   nameA,nameB,str(delta),str(int(d)),str(P),str(E(K)),filtered,str(nn),ref
// with_dates = 0 writes "NA" for delta, P and E(K).  filt == NULL writes "NA" in the filtered column (metadata on, --filter
// off, :204), otherwise the integers.  k_max < 0 means no -K filter; else only rows with k_max >= E(K) are written (:222).
// Rows are formatted by all cores into raw buffers (names copied, floats through the memo) and the buffers of a batch are written
// in parallel at their offsets (pwrite): the 5 GB of a 10 000-sample run neither queue behind one formatter nor behind one writer.
// One writer serves a whole run: the file stays open and the memo of formatted floats lives across its batches
// (tracs_distance_run hands the rows over a few million at a time, as they arrive from the device).
}  // extern "C"

namespace tracs {

struct DistanceRowWriter::Impl {
    int fd = -1;
    off_t file_off = 0;
    std::string path, ref;
    const char *const *names = nullptr;
    std::vector<uint32_t> name_len;
    size_t longest = 0;
    std::unique_ptr<FloatMemo> memo;
    std::vector<std::vector<char>> buf;
    uint64_t written = 0;
};

DistanceRowWriter::DistanceRowWriter() : p(new Impl()) {}
DistanceRowWriter::~DistanceRowWriter() { if (p->fd >= 0) (void)::close(p->fd); delete p; }
uint64_t DistanceRowWriter::written() const { return p->written; }

int DistanceRowWriter::open(const char *path, const char *const *names, size_t n_names, const char *ref)
{
    p->fd = ::open(path, O_WRONLY | O_CREAT, 0644);
    if (p->fd < 0) { set_error(std::string("cannot open '") + path + "' for writing"); return TRACS_E_OPEN; }
    struct stat st;
    if (fstat(p->fd, &st) != 0) { set_error(std::string("cannot stat '") + path + "'"); return TRACS_E_OPEN; }
    p->file_off = st.st_size;                                 // append
    p->path = path; p->ref = ref; p->names = names;
    p->name_len.resize(n_names);
    for (size_t k = 0; k < n_names; k++) { p->name_len[k] = (uint32_t)std::strlen(names[k]); p->longest = std::max<size_t>(p->longest, p->name_len[k]); }
    return TRACS_OK;
}

int DistanceRowWriter::close()
{
    if (p->fd < 0) return TRACS_OK;
    const int rc = ::close(p->fd);
    p->fd = -1;
    if (rc != 0) { set_error(std::string("error closing '") + p->path + "'"); return TRACS_E_OPEN; }
    return TRACS_OK;
}

// T: the integer columns' type (uint64_t: the pybind-shaped arrays of TRACS.pairsnp; uint32_t: as they leave the device);
// delta_of(r): the row's date difference in years
template <class T, class DeltaOf>
int DistanceRowWriter::append(const T *rows, const T *cols, const T *snpd, const T *filt, const T *ncomp, DeltaOf delta_of,
                              const double *p_direct, const double *e_k, size_t n, int with_dates, double k_max)
{
    Impl &w = *p;
    for (size_t r = 0; r < n; r++)
        if ((size_t)rows[r] >= w.name_len.size() || (size_t)cols[r] >= w.name_len.size()) { set_error("distance rows: sample index beyond the names"); return TRACS_E_ARG; }
    const size_t ref_len = w.ref.size();
    const char *ref = w.ref.c_str();
    const size_t row_cap = 2 * w.longest + ref_len + 3 * 32 + 3 * 20 + 16;      // two names, three floats, three integers, separators
    const unsigned TH = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const size_t chunk = std::max<size_t>(1024, std::min<size_t>(1u << 17, (n + TH - 1) / TH));
    if (with_dates && !w.memo) w.memo.reset(new FloatMemo(n >= (1u << 20) ? 21 : 16));
    FloatMemo *memo = w.memo.get();
    if (w.buf.size() < TH) w.buf.resize(TH);
    std::vector<size_t> used(TH, 0);
    std::vector<uint64_t> cnt(TH, 0);
    int rc = TRACS_OK;
    for (size_t base = 0; base < n && rc == TRACS_OK; base += chunk * TH) {
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < TH; t++)
                th.emplace_back([&, t]() {
                    const size_t r0 = std::min(n, base + (size_t)t * chunk), r1 = std::min(n, r0 + chunk);
                    if (w.buf[t].size() < (r1 - r0) * row_cap + 64) w.buf[t].resize((r1 - r0) * row_cap + 64);
                    char *o = w.buf[t].data();
                    uint64_t c = 0;
                    for (size_t r = r0; r < r1; r++) {
                        if (with_dates && k_max >= 0.0 && !(k_max >= e_k[r])) continue;
                        std::memcpy(o, w.names[rows[r]], w.name_len[rows[r]]); o += w.name_len[rows[r]]; *o++ = ',';
                        std::memcpy(o, w.names[cols[r]], w.name_len[cols[r]]); o += w.name_len[cols[r]]; *o++ = ',';
                        if (with_dates) o += memo->put(delta_of(r), o); else { *o++ = 'N'; *o++ = 'A'; }
                        *o++ = ',';
                        o += format_u64((uint64_t)snpd[r], o); *o++ = ',';
                        if (with_dates) o += memo->put(p_direct[r], o); else { *o++ = 'N'; *o++ = 'A'; }
                        *o++ = ',';
                        if (with_dates) o += memo->put(e_k[r], o); else { *o++ = 'N'; *o++ = 'A'; }
                        *o++ = ',';
                        if (filt) o += format_u64((uint64_t)filt[r], o); else { *o++ = 'N'; *o++ = 'A'; }
                        *o++ = ',';
                        o += format_u64((uint64_t)ncomp[r], o); *o++ = ',';
                        std::memcpy(o, ref, ref_len); o += ref_len;
                        *o++ = '\n';
                        c++;
                    }
                    used[t] = (size_t)(o - w.buf[t].data());
                    cnt[t] = c;
                });
            for (auto &x : th) x.join();
        }
        std::vector<off_t> at(TH);
        for (unsigned t = 0; t < TH; t++) { at[t] = w.file_off; w.file_off += (off_t)used[t]; w.written += cnt[t]; }
        std::atomic<int> bad{0};
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < TH; t++)
                th.emplace_back([&, t]() {
                    size_t done = 0;
                    while (done < used[t]) {
                        const ssize_t wr = pwrite(w.fd, w.buf[t].data() + done, used[t] - done, at[t] + (off_t)done);
                        if (wr <= 0) { bad.store(1); return; }
                        done += (size_t)wr;
                    }
                });
            for (auto &x : th) x.join();
        }
        if (bad.load()) { set_error(std::string("error writing '") + w.path + "'"); rc = TRACS_E_OPEN; }
    }
    return rc;
}

int DistanceRowWriter::append_u32(const uint32_t *rows, const uint32_t *cols, const uint32_t *snpd, const uint32_t *filt, const uint32_t *ncomp,
                                  const int32_t *days, const double *p_direct, const double *e_k, size_t n, int with_dates, double k_max)
{
    // tracs/transcluster.py:26-33: |t_i - t_j| / 31556952.0 with t = whole days in seconds (exact in f64): DenseSource::get's expression
    auto delta_of = [=](size_t r) {
        const long long dd = (long long)days[rows[r]] - (long long)days[cols[r]];
        return (double)((dd < 0 ? -dd : dd) * 86400ll) / 31556952.0;
    };
    return append<uint32_t>(rows, cols, snpd, filt, ncomp, delta_of, p_direct, e_k, n, with_dates, k_max);
}

}  // namespace tracs

extern "C" {

int tracs_write_distance_rows(const char *path, const char *const *names, const uint64_t *rows, const uint64_t *cols,
                              const uint64_t *snpd, const uint64_t *filt, const uint64_t *ncomp, const double *delta,
                              const double *p_direct, const double *e_k, size_t n, int with_dates, double k_max,
                              const char *ref, uint64_t *rows_written)
{
    if (!path || !ref || (n && (!names || !rows || !cols || !snpd || !ncomp)) || (with_dates && n && (!delta || !p_direct || !e_k))) {
        set_error("tracs_write_distance_rows: NULL argument");
        return TRACS_E_ARG;
    }
    size_t max_row = 0;
    for (size_t r = 0; r < n; r++) max_row = std::max<size_t>(max_row, std::max(rows[r], cols[r]));
    tracs::DistanceRowWriter w;
    int rc = w.open(path, names, n ? max_row + 1 : 0, ref);
    if (rc == TRACS_OK) rc = w.append<uint64_t>(rows, cols, snpd, filt, ncomp, [=](size_t r) { return delta[r]; }, p_direct, e_k, n, with_dates, k_max);
    const int rc2 = w.close();
    if (rows_written) *rows_written = w.written();
    return rc ? rc : rc2;
}

// test hook: str(float) of n doubles, '\n'-separated, into buf (cap bytes); returns the bytes used or -1
long tracs_debug_format_floats(const double *x, size_t n, char *buf, size_t cap)
{
    size_t used = 0;
    for (size_t i = 0; i < n; i++) {
        if (cap - used < 40) return -1;
        used += format_py_float(x[i], buf + used);
        buf[used++] = '\n';
    }
    return (long)used;
}

// ---- `tracs cluster` input: the distance CSV -> node ids + thresholded edges (tracs/cluster.py:100-116) -------------------
struct tracs_edge_list {
    std::vector<std::string> names;       // node id -> name, ids in first-appearance order (sampleA before sampleB)
    std::vector<int32_t> I, J;            // edges with value <= threshold
    uint64_t n_rows = 0;                  // data lines read (the reference's `count`)
};

// Header line skipped; every other line: strip, split at ',', node ids for fields 0 and 1 (new names get the next id;
// ids continue after the n_seed names passed in, which is how the reference's function-level table behaves across calls,
// tracs/cluster.py:11-21), edge iff float(field[column]) <= threshold.  A field that is not a float is an error with
// Python's message; chunks are parsed in parallel and merged in file order, so ids equal the serial scan's.
int tracs_read_distance_edges(const char *path, int column, double threshold, const char *const *seed_names, size_t n_seed,
                              tracs_edge_list **out)
{
    if (out) *out = nullptr;
    if (!path || !out || column < 2 || (n_seed && !seed_names)) { set_error("tracs_read_distance_edges: bad argument"); return TRACS_E_ARG; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error(std::string("cannot open '") + path + "'"); return TRACS_E_OPEN; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); set_error(std::string("cannot stat '") + path + "'"); return TRACS_E_OPEN; }
    const size_t size = (size_t)st.st_size;
    auto res = std::make_unique<tracs_edge_list>();
    for (size_t i = 0; i < n_seed; i++) res->names.emplace_back(seed_names[i]);
    if (size == 0) { close(fd); set_error("StopIteration"); return TRACS_E_FASTA; }        // next(infile) on an empty file raises
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { set_error(std::string("cannot map '") + path + "'"); return TRACS_E_OPEN; }
    const char *p = static_cast<const char *>(m);
    const char *hdr = static_cast<const char *>(std::memchr(p, '\n', size));
    const size_t body0 = hdr ? (size_t)(hdr - p) + 1 : size;

    struct Chunk {
        std::vector<std::string_view> local;                       // distinct names, first-appearance order within the chunk
        std::unordered_map<std::string_view, int32_t> idx;
        std::vector<int32_t> a, b;                                 // per kept row: local ids
        uint64_t rows = 0;
        std::string error;
        uint64_t error_row = 0;
    };
    const unsigned T = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const size_t nchunks = std::max<size_t>(1, std::min<size_t>(T * 4, (size - body0) / (1u << 20) + 1));
    std::vector<size_t> cut(nchunks + 1, size);
    cut[0] = body0;
    for (size_t c = 1; c < nchunks; c++) {                        // chunk boundaries at line starts
        size_t pos = body0 + (size - body0) * c / nchunks;
        if (pos < cut[c - 1]) pos = cut[c - 1];
        const char *nl = pos < size ? static_cast<const char *>(std::memchr(p + pos, '\n', size - pos)) : nullptr;
        cut[c] = nl ? (size_t)(nl - p) + 1 : size;
    }
    std::vector<Chunk> chunks(nchunks);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= nchunks) return;
            Chunk &ch = chunks[c];
            auto local_id = [&](std::string_view nm) {
                auto it = ch.idx.find(nm);
                if (it != ch.idx.end()) return it->second;
                const int32_t id = (int32_t)ch.local.size();
                ch.local.push_back(nm);
                ch.idx.emplace(nm, id);
                return id;
            };
            size_t pos = cut[c];
            const size_t end = cut[c + 1];
            while (pos < end) {
                const char *nl = static_cast<const char *>(std::memchr(p + pos, '\n', end - pos));
                size_t e = nl ? (size_t)(nl - p) : end;
                const size_t nextpos = nl ? e + 1 : end;
                size_t b = pos;
                while (b < e && is_space((unsigned char)p[b])) b++;          // line.strip()
                while (e > b && is_space((unsigned char)p[e - 1])) e--;
                ch.rows++;
                // fields 0, 1 and `column`
                size_t f0 = b, fe = b;
                int field = 0;
                std::string_view n0, n1, val;
                bool have = false;
                for (size_t k = b; k <= e; k++) {
                    if (k == e || p[k] == ',') {
                        fe = k;
                        if (field == 0) n0 = std::string_view(p + f0, fe - f0);
                        else if (field == 1) n1 = std::string_view(p + f0, fe - f0);
                        if (field == column) { val = std::string_view(p + f0, fe - f0); have = true; }
                        field++;
                        f0 = k + 1;
                        if (have && field > 1) break;
                    }
                }
                if (!have || field < 2) {
                    if (ch.error.empty()) { ch.error = "list index out of range"; ch.error_row = ch.rows; }
                    pos = nextpos;
                    continue;
                }
                const int32_t ia = local_id(n0), ib = local_id(n1);
                // float(): optional surrounding whitespace, the whole token must convert
                size_t v0 = 0, v1 = val.size();
                while (v0 < v1 && is_space((unsigned char)val[v0])) v0++;
                while (v1 > v0 && is_space((unsigned char)val[v1 - 1])) v1--;
                char tmp[64];
                double x = 0.0;
                bool ok = v1 > v0 && v1 - v0 < sizeof tmp;
                if (ok) {
                    std::memcpy(tmp, val.data() + v0, v1 - v0);
                    tmp[v1 - v0] = 0;
                    char *endp = nullptr;
                    x = std::strtod(tmp, &endp);
                    ok = endp == tmp + (v1 - v0) && !(tmp[0] == '0' && (tmp[1] == 'x' || tmp[1] == 'X'));
                }
                if (!ok) {
                    if (ch.error.empty()) { ch.error = "could not convert string to float: '" + std::string(val) + "'"; ch.error_row = ch.rows; }
                    pos = nextpos;
                    continue;
                }
                if (x <= threshold) { ch.a.push_back(ia); ch.b.push_back(ib); }
                pos = nextpos;
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < std::min<size_t>(T, nchunks); t++) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    // merge in file order: global ids by first appearance
    std::unordered_map<std::string, int32_t> gid;
    for (size_t i = 0; i < res->names.size(); i++) gid.emplace(res->names[i], (int32_t)i);
    int rc = TRACS_OK;
    uint64_t rows_before = 0;
    for (size_t c = 0; c < nchunks && rc == TRACS_OK; c++) {
        Chunk &ch = chunks[c];
        if (!ch.error.empty()) {                                    // the serial scan would have stopped at this row
            set_error(ch.error);
            rc = TRACS_E_FASTA;
            break;
        }
        std::vector<int32_t> map(ch.local.size());
        for (size_t k = 0; k < ch.local.size(); k++) {
            std::string nm(ch.local[k]);
            auto it = gid.find(nm);
            if (it == gid.end()) {
                it = gid.emplace(nm, (int32_t)res->names.size()).first;
                res->names.push_back(std::move(nm));
            }
            map[k] = it->second;
        }
        for (size_t k = 0; k < ch.a.size(); k++) { res->I.push_back(map[ch.a[k]]); res->J.push_back(map[ch.b[k]]); }
        rows_before += ch.rows;
    }
    res->n_rows = rows_before;
    munmap(m, size);
    if (rc != TRACS_OK) return rc;
    *out = res.release();
    return TRACS_OK;
}
size_t tracs_edges_count(const tracs_edge_list *e) { return e ? e->I.size() : 0; }
uint64_t tracs_edges_rows(const tracs_edge_list *e) { return e ? e->n_rows : 0; }
size_t tracs_edges_n_names(const tracs_edge_list *e) { return e ? e->names.size() : 0; }
const char *tracs_edges_name(const tracs_edge_list *e, size_t i) { return e && i < e->names.size() ? e->names[i].c_str() : ""; }
const int32_t *tracs_edges_i(const tracs_edge_list *e) { return e ? e->I.data() : nullptr; }
const int32_t *tracs_edges_j(const tracs_edge_list *e) { return e ? e->J.data() : nullptr; }
void tracs_edges_free(tracs_edge_list *e) { delete e; }

// One gzip member per sample (a multi-member gzip file is what gzopen/zcat/kseq read as one stream), compressed in
// parallel and written in input order:  ">" sample "\n" sequence "\n"  (tracs/combine.py:227-231).
int tracs_combine_fasta(const char *out_path, const char *const *sample_names, const char *const *fasta_paths, size_t n,
                        int n_threads, int gzip_level, double *frac_n, uint64_t *lengths)
{
    if (!out_path || (n && (!sample_names || !fasta_paths))) { set_error("tracs_combine_fasta: NULL argument"); return TRACS_E_ARG; }
    FILE *fo = std::fopen(out_path, "wb");
    if (!fo) { set_error(std::string("cannot open '") + out_path + "' for writing"); return TRACS_E_OPEN; }
    const unsigned T = (unsigned)std::max(1, std::min(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), 64));
    const int level = gzip_level < 0 ? 6 : std::min(9, gzip_level);
    std::mutex mu;
    std::string first_error;
    int rc = TRACS_OK;
    for (size_t base = 0; base < n && rc == TRACS_OK; base += T) {
        const size_t cnt = std::min<size_t>(T, n - base);
        std::vector<std::vector<unsigned char>> member(cnt);
        std::vector<std::thread> th;
        for (size_t t = 0; t < cnt; t++)
            th.emplace_back([&, t]() {
                const size_t s = base + t;
                tracs::FastaData fd;
                std::string err;
                int r = tracs::read_fasta(fasta_paths[s], fd, err);
                if (r == TRACS_E_RAGGED || (r == TRACS_OK && fd.n > 1)) {       // combine.py:233-237
                    r = TRACS_E_FASTA;
                    err = std::string("ERROR: ") + fasta_paths[s] + " contains more than one sequence";
                }
                if (r != TRACS_OK) {
                    std::lock_guard<std::mutex> lock(mu);
                    if (rc == TRACS_OK) { rc = r; first_error = err; }
                    return;
                }
                if (fd.n == 0) {                                                 // no record: nothing written, no ncov entry
                    if (frac_n) frac_n[s] = -1.0;
                    if (lengths) lengths[s] = 0;
                    return;
                }
                std::string text;
                text.reserve(fd.L + std::strlen(sample_names[s]) + 4);
                text.push_back('>');
                text += sample_names[s];
                text.push_back('\n');
                text.append(reinterpret_cast<const char *>(fd.seq.data()), fd.L);
                text.push_back('\n');
                size_t nN = 0;
                for (size_t k = 0; k < fd.L; k++) nN += fd.seq[k] == 'N';
                if (frac_n) frac_n[s] = fd.L ? (double)nN / (double)fd.L : 0.0;  // seq.count("N") / len(seq)  (:238)
                if (lengths) lengths[s] = fd.L;
                z_stream zs;
                std::memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                    std::lock_guard<std::mutex> lock(mu);
                    if (rc == TRACS_OK) { rc = TRACS_E_NOMEM; first_error = "deflateInit2 failed"; }
                    return;
                }
                // gzip FEXTRA subfield "TR": the member's total size in bytes (patched in below), so a reader can hop from
                // member to member and inflate them in parallel (fasta.cpp); every other gzip reader ignores extra fields
                unsigned char extra[12] = {'T', 'R', 8, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                gz_header gh;
                std::memset(&gh, 0, sizeof gh);
                gh.os = 3;
                gh.extra = extra;
                gh.extra_len = sizeof extra;
                deflateSetHeader(&zs, &gh);
                std::vector<unsigned char> &o = member[t];
                o.resize(deflateBound(&zs, (uLong)text.size()) + 128);
                zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(text.data()));
                zs.avail_in = (uInt)text.size();
                zs.next_out = o.data();
                zs.avail_out = (uInt)o.size();
                const int zr = deflate(&zs, Z_FINISH);
                o.resize(zr == Z_STREAM_END ? zs.total_out : 0);
                deflateEnd(&zs);
                if (o.size() > 24 && o[3] == 4 && o[12] == 'T' && o[13] == 'R') {   // FLG = FEXTRA only: payload at byte 16
                    const uint64_t sz = o.size();
                    for (int b = 0; b < 8; b++) o[16 + b] = (unsigned char)(sz >> (8 * b));
                }
                if (zr != Z_STREAM_END) {
                    std::lock_guard<std::mutex> lock(mu);
                    if (rc == TRACS_OK) { rc = TRACS_E_NOMEM; first_error = "deflate failed"; }
                }
            });
        for (auto &x : th) x.join();
        if (rc != TRACS_OK) break;
        for (size_t t = 0; t < cnt; t++)
            if (!member[t].empty() && std::fwrite(member[t].data(), 1, member[t].size(), fo) != member[t].size()) {
                rc = TRACS_E_OPEN;
                first_error = std::string("error writing '") + out_path + "'";
                break;
            }
    }
    if (std::fclose(fo) != 0 && rc == TRACS_OK) { rc = TRACS_E_OPEN; first_error = std::string("error closing '") + out_path + "'"; }
    if (rc != TRACS_OK) set_error(first_error);
    return rc;
}

}  // extern "C"
