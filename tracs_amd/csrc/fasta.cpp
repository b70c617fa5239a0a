// fasta.cpp -- host FASTA/FASTQ(.gz) reader feeding the pack kernel.
//
// Behaviour follows the reference's reader, klib kseq as driven by load_seqs
// (/root/reference/src/kseq.h:170-208, /root/reference/src/pairsnp.hpp:75-101):
//   * the first record starts at the first '>' or '@' anywhere in the stream;
//   * name = header bytes up to the first whitespace; the rest of the line is ignored;
//   * sequence = every printable non-space byte up to the next '>', '+' or '@' (at any position,
//     not only at line starts);
//   * '+' opens a FASTQ quality block: the rest of that line is skipped, then as many printable
//     bytes as the sequence has are consumed; a shorter block is an error;
//   * all records must have one length ("Error reading FASTA, variable sequence lengths!").
// Own implementation: a streaming state machine over 4 MiB gzread blocks (zlib reads plain and gzip files alike, as
// gzopen does for the reference), plus a multi-threaded mmap fast path for large well-formed plain FASTA files.
#include "fasta.h"

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstring>
#include <thread>

namespace tracs {

namespace {

// Block reader over gzread with direct access to the current block (the hot loops below scan it in place).
class ByteStream {
public:
    explicit ByteStream(gzFile f) : f_(f), buf_(4u << 20) {}
    bool fill()                                       // true if at least one byte is available
    {
        if (pos_ < len_) return true;
        if (eof_) return false;
        const int r = gzread(f_, buf_.data(), (unsigned)buf_.size());
        if (r <= 0) { eof_ = true; return false; }
        len_ = (size_t)r;
        pos_ = 0;
        return true;
    }
    int get() { return fill() ? (unsigned char)buf_[pos_++] : -1; }
    const unsigned char *cur() const { return reinterpret_cast<const unsigned char *>(buf_.data()) + pos_; }
    size_t avail() const { return len_ - pos_; }
    void advance(size_t k) { pos_ += k; }

private:
    gzFile f_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
};

inline bool is_graph(unsigned c) { return c - 33u < 94u; }      // isgraph() in the C locale: 33..126

// The text of a large alignment is gigabytes that are touched once (by the reader threads), copied to the device and given back:
// with 4-KiB pages that is millions of faults on the way in and as many pages to unmap on the way out (0.25 s of a 3 s command at
// 5 GB).  Ask for transparent huge pages where the system offers them on request; a no-op elsewhere.
inline void huge_pages_hint(ByteVec &v)
{
#ifdef MADV_HUGEPAGE
    const uintptr_t two_mb = (uintptr_t)2 << 20, lo = (reinterpret_cast<uintptr_t>(v.data()) + two_mb - 1) & ~(two_mb - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(v.data()) + v.size()) & ~(two_mb - 1);
    if (v.size() >= ((size_t)64 << 20) && hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
#endif
}

}  // namespace

// ---- parallel fast path for plain (uncompressed) FASTA -------------------------------------------------------
// Valid exactly when the byte-level rules above reduce to "records start at '>' in column 0": the file begins with
// '>', every other '>' follows a newline, and no '>', '+' or '@' occurs inside a sequence line.  Anything else
// (FASTQ, junk before the first header, '>' mid-line, ragged lengths) returns false and the serial state machine
// below -- the exact restatement of kseq -- reads the file and produces the reference's behaviour and messages.
static bool read_fasta_parallel(const std::string &path, FastaData &out)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 64 * 1024 * 1024) { close(fd); return false; }   // small files: serial
    const size_t size = (size_t)st.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    const unsigned char *p = static_cast<const unsigned char *>(m);
    bool ok = p[0] == '>' && !(p[0] == 0x1f && p[1] == 0x8b);
    std::vector<size_t> starts;
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (ok) {
        // record starts: '>' at offset 0 or right after a newline
        std::vector<std::vector<size_t>> part(T);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                const size_t a = size * t / T, b = size * (t + 1) / T;
                const unsigned char *q = p + std::max<size_t>(a, 1);
                const unsigned char *e = p + b;
                while (q < e && (q = static_cast<const unsigned char *>(memchr(q, '>', (size_t)(e - q)))) != nullptr) {
                    if (q[-1] == '\n') part[t].push_back((size_t)(q - p));
                    q++;
                }
            });
        for (auto &x : th) x.join();
        starts.push_back(0);
        for (auto &v : part) starts.insert(starts.end(), v.begin(), v.end());
    }
    size_t L = 0;
    const size_t nrec = starts.size();
    std::vector<std::string> names(nrec);
    ByteVec seq;
    if (ok) {
        // length of the first record fixes L
        auto body_of = [&](size_t r, size_t &b0, size_t &b1) {
            const size_t s0 = starts[r], s1 = r + 1 < nrec ? starts[r + 1] : size;
            const unsigned char *nl = static_cast<const unsigned char *>(memchr(p + s0, '\n', s1 - s0));
            b0 = nl ? (size_t)(nl - p) + 1 : s1;
            b1 = s1;
        };
        size_t b0, b1;
        body_of(0, b0, b1);
        for (size_t k = b0; k < b1; k++) L += is_graph(p[k]);
        seq.resize(nrec * L);
        huge_pages_hint(seq);
        std::atomic<bool> bad{false};
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&]() {
                for (;;) {
                    const size_t r = next.fetch_add(1);
                    if (r >= nrec || bad.load(std::memory_order_relaxed)) return;
                    size_t c0, c1;
                    body_of(r, c0, c1);
                    // header: name = bytes after '>' up to the first whitespace
                    size_t h = starts[r] + 1;
                    const size_t hend = c0 ? c0 - 1 : 0;
                    size_t he = h;
                    while (he < hend && !std::isspace(p[he])) he++;
                    names[r].assign(reinterpret_cast<const char *>(p + h), he - h);
                    uint8_t *dst = seq.data() + r * L;
                    size_t w = 0;
                    bool stray = false;
                    for (size_t k = c0; k < c1; k++) {
                        const unsigned ch = p[k];
                        stray |= (ch == '>') | (ch == '+') | (ch == '@');
                        if (w < L) dst[w] = (uint8_t)ch;
                        w += is_graph(ch);
                    }
                    if (stray || w != L) bad.store(true);
                }
            });
        for (auto &x : th) x.join();
        ok = !bad.load();
    }
    munmap(m, size);
    if (!ok) return false;
    out.n = nrec;
    out.L = L;
    out.seq = std::move(seq);
    out.names = std::move(names);
    return true;
}

// ---- parallel fast path for gzip files made of indexed members ------------------------------------------------------
// `tracs combine` here (tracs_combine_fasta, alignio.cpp) writes one gzip member per sample and records each member's
// size in a gzip FEXTRA subfield "TR".  Such a file is walked member to member without inflating anything, the members
// are inflated in parallel, and each must hold exactly one plain record (">name ...\nSEQUENCE\n", no '>', '+', '@' in the
// sequence).  Any other gzip file -- no subfield, a chain that does not end at EOF, a member that is not one clean record,
// ragged lengths -- returns false and is read by the serial state machine below, which owns every error message.
static bool read_fasta_members(const std::string &path, FastaData &out)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 32) { close(fd); return false; }
    const size_t size = (size_t)st.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    const unsigned char *p = static_cast<const unsigned char *>(m);
    std::vector<std::pair<size_t, size_t>> members;       // (offset, size)
    bool ok = true;
    for (size_t off = 0; off < size;) {
        const unsigned char *h = p + off;
        // 10-byte header, FLG has FEXTRA, first subfield is ours
        if (size - off < 32 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'T' || h[13] != 'R' ||
            h[14] != 8 || h[15] != 0) { ok = false; break; }
        uint64_t sz = 0;
        for (int b = 0; b < 8; b++) sz |= (uint64_t)h[16 + b] << (8 * b);
        if (sz < 32 || sz > size - off) { ok = false; break; }
        members.emplace_back(off, (size_t)sz);
        off += (size_t)sz;
    }
    if (!ok || members.empty()) { munmap(m, size); return false; }
    const size_t nrec = members.size();
    auto isize_of = [&](size_t r) {                       // gzip trailer: uncompressed size mod 2^32
        const unsigned char *t = p + members[r].first + members[r].second - 4;
        return (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    };
    std::vector<std::string> names(nrec);
    ByteVec seq;
    std::atomic<bool> bad{false};
    std::atomic<size_t> next{0};
    size_t L = 0;
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    // one member -> its record; r == 0 runs first, alone, and fixes L
    auto do_member = [&](size_t r, std::vector<unsigned char> &text) -> bool {
        const size_t want = isize_of(r);
        if (members[r].second > 0xFFFFFFFFu) return false;
        text.resize(want + 1);
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(p + members[r].first);
        zs.avail_in = (uInt)members[r].second;
        zs.next_out = text.data();
        zs.avail_out = (uInt)text.size();
        const int zr = inflate(&zs, Z_FINISH);
        const size_t got = zs.total_out;
        const bool whole = zr == Z_STREAM_END && zs.avail_in == 0;
        inflateEnd(&zs);
        if (!whole || got != want || got < 2 || text[0] != '>') return false;
        const unsigned char *nl = static_cast<const unsigned char *>(std::memchr(text.data(), '\n', got));
        if (!nl) return false;
        size_t he = 1;
        const size_t hend = (size_t)(nl - text.data());
        while (he < hend && !std::isspace(text[he])) he++;
        names[r].assign(reinterpret_cast<const char *>(text.data() + 1), he - 1);
        const size_t b0 = hend + 1;
        if (r == 0) {
            size_t cnt = 0;
            for (size_t k = b0; k < got; k++) cnt += is_graph(text[k]);
            L = cnt;
            seq.resize(nrec * L);
            huge_pages_hint(seq);
        }
        uint8_t *dst = seq.data() + r * L;
        size_t w = 0;
        bool stray = false;
        for (size_t k = b0; k < got; k++) {
            const unsigned ch = text[k];
            stray |= (ch == '>') | (ch == '+') | (ch == '@');
            if (w < L) dst[w] = (uint8_t)ch;
            w += is_graph(ch);
        }
        return !stray && w == L;
    };
    {
        std::vector<unsigned char> text;
        if (!do_member(0, text)) { munmap(m, size); return false; }
    }
    next.store(1);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&]() {
            std::vector<unsigned char> text;
            for (;;) {
                const size_t r = next.fetch_add(1);
                if (r >= nrec || bad.load(std::memory_order_relaxed)) return;
                if (!do_member(r, text)) bad.store(true);
            }
        });
    for (auto &x : th) x.join();
    munmap(m, size);
    if (bad.load()) return false;
    out.n = nrec;
    out.L = L;
    out.seq = std::move(seq);
    out.names = std::move(names);
    return true;
}

int read_fasta(const std::string &path, FastaData &out, std::string &err)
{
    if (out.n == 0 && out.seq.empty() && !getenv("TRACS_SERIAL_FASTA") && read_fasta_members(path, out)) return TRACS_OK;
    if (out.n == 0 && out.seq.empty() && !getenv("TRACS_SERIAL_FASTA") && read_fasta_parallel(path, out)) return TRACS_OK;
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { err = "cannot open '" + path + "'"; return TRACS_E_OPEN; }
    gzbuffer(f, 1u << 20);
    ByteStream in(f);
    int pending = 0;          // header character already consumed ('>' or '@'), 0 if none
    std::string name;
    std::vector<uint8_t> rec;
    int rc = TRACS_OK;
    for (;;) {
        int c;
        if (!pending) {
            while ((c = in.get()) != -1 && c != '>' && c != '@') {}
            if (c == -1) break;
            pending = c;
        }
        // header
        name.clear();
        c = in.get();
        if (c == -1) break;                                   // header char was the last byte: no record
        while (c != -1 && !std::isspace(c)) { name.push_back((char)c); c = in.get(); }
        if (c != -1 && c != '\n') while ((c = in.get()) != -1 && c != '\n') {}
        // sequence: scan whole blocks in place -- every printable byte up to the next '>', '+' or '@'
        rec.clear();
        if (out.L) rec.reserve(out.L + 64);
        c = -1;
        while (in.fill()) {
            const unsigned char *p = in.cur();
            const size_t n = in.avail();
            const size_t base = rec.size();
            rec.resize(base + n);
            uint8_t *dst = rec.data() + base;
            size_t k = 0, w = 0;
            for (; k < n; k++) {
                const unsigned ch = p[k];
                if (ch == '>' || ch == '+' || ch == '@') break;
                dst[w] = (uint8_t)ch;                         // branch-free append of printable bytes
                w += is_graph(ch);
            }
            rec.resize(base + w);
            if (k < n) { c = p[k]; in.advance(k + 1); break; }
            in.advance(n);
        }
        pending = (c == '>' || c == '@') ? c : 0;
        if (c == '+') {
            while ((c = in.get()) != -1 && c != '\n') {}
            if (c == -1) { err = "Error reading FASTA!"; rc = TRACS_E_FASTA; break; }
            size_t q = 0;
            while ((c = in.get()) != -1 && q < rec.size())
                if (c >= 33 && c <= 127) q++;
            if (q != rec.size()) { err = "Error reading FASTA!"; rc = TRACS_E_FASTA; break; }
        }
        if (out.n > 0 && rec.size() != out.L) {
            err = "Error reading FASTA, variable sequence lengths!";
            rc = TRACS_E_RAGGED;
            break;
        }
        out.L = rec.size();
        out.seq.insert(out.seq.end(), rec.begin(), rec.end());
        out.names.push_back(name);
        out.n++;
        if (c == -1) break;
    }
    gzclose(f);
    return rc;
}

}  // namespace tracs
