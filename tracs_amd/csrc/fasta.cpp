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

class ByteStream {
public:
    explicit ByteStream(gzFile f) : f_(f), buf_(1u << 20) {}
    // next byte or -1 at end of file
    int get()
    {
        if (pos_ >= len_) {
            if (eof_) return -1;
            const int r = gzread(f_, buf_.data(), (unsigned)buf_.size());
            if (r <= 0) { eof_ = true; return -1; }
            len_ = (size_t)r;
            pos_ = 0;
        }
        return (unsigned char)buf_[pos_++];
    }

private:
    gzFile f_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
};

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
        // sequence
        rec.clear();
        while ((c = in.get()) != -1 && c != '>' && c != '+' && c != '@')
            if (std::isgraph(c)) rec.push_back((uint8_t)c);
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
