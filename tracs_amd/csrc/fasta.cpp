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
// Own implementation: a streaming state machine over 1 MiB gzread blocks (zlib reads plain and
// gzip files alike, as gzopen does for the reference).
#include "fasta.h"

#include <zlib.h>

#include <cctype>
#include <cstring>

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

}  // namespace

int read_fasta(const std::string &path, FastaData &out, std::string &err)
{
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
