// san_driver.cpp -- host-only code of libtracs_hip.so (fasta.cpp, alignio.cpp) under ASan + UBSan (CPU build; GPU
// sanitizers are not available on the pool).  usage: san_driver <scratch dir>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../tracs_amd/csrc/fasta.h"

namespace tracs {
static std::string g_err;
void set_error(const std::string &m) { g_err = m; }
}  // namespace tracs

extern "C" long tracs_debug_format_floats(const double *x, size_t n, char *buf, size_t cap);

static void put(const std::string &path, const std::string &data, bool gz)
{
    if (gz) { gzFile f = gzopen(path.c_str(), "wb"); gzwrite(f, data.data(), (unsigned)data.size()); gzclose(f); }
    else { FILE *f = fopen(path.c_str(), "wb"); fwrite(data.data(), 1, data.size(), f); fclose(f); }
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    int fails = 0;
    // ---- FASTA reader on adversarial inputs
    const char *cases[] = {"", ">", ">a", ">a\n", ">a\nACGT", ">a\nACGT\n>b\nAC\n", "junk>a\nAC>b\nGT", "@r\nACGT\n+\nII", "@r\nACGT\n+\nIIII\n",
                           ">a desc\r\nAC\r\nGT\r\n>b\nACGT", "\n\n>x\n\n\nA\n", ">a\nAC\n+\n", "+\n", ">a\n\x01\x02" "AC\n"};
    for (size_t c = 0; c < sizeof cases / sizeof *cases; c++)
        for (int gz = 0; gz < 2; gz++) {
            const std::string p = dir + "/san_case.fa";
            put(p, cases[c], gz);
            tracs::FastaData fd;
            std::string err;
            (void)tracs::read_fasta(p, fd, err);
            if (fd.seq.size() != fd.n * fd.L && err.empty()) { printf("case %zu: inconsistent sizes\n", c); fails++; }
        }
    {   // big plain file -> parallel reader, with a ragged record in the middle (falls back, reports the reference's message)
        std::string big;
        for (int r = 0; r < 40; r++) { big += ">r" + std::to_string(r) + "\n"; big.append(r == 17 ? 1999999 : 2000000, "ACGT"[r & 3]); big += "\n"; }
        const std::string p = dir + "/san_big.fa";
        put(p, big, false);
        tracs::FastaData fd;
        std::string err;
        const int rc = tracs::read_fasta(p, fd, err);
        if (rc != TRACS_E_RAGGED) { printf("big ragged: rc %d (%s)\n", rc, err.c_str()); fails++; }
    }
    // ---- pileup parser, CSV writers, combine, edge reader
    {
        const std::string p = dir + "/san_pile.txt";
        put(p, "c 1 A A,C 3:1,1:1,0\nc 2 N A 1:1:0\nc 3 A A,-1C,+2GG 9:1,2,3:4,5\nc 3 A A 9:1\n", false);
        const char *names[] = {"c"};
        const uint64_t lens[] = {3};
        double counts[12];
        uint64_t nl = 0;
        const int rc = tracs_pileup_counts(p.c_str(), names, lens, 1, 1, counts, &nl);
        if (rc == 0) { printf("pileup: malformed last line accepted\n"); fails++; }
        put(p, "c 1 A A,C 3:1,1:1,0\nc 3 A A,-1C,+2GG 9:1,2,3:4,5", true);
        if (tracs_pileup_counts(p.c_str(), names, lens, 1, 0, counts, &nl) != 0 || counts[0] != 2 || counts[8] != 5) { printf("pileup values\n"); fails++; }
    }
    {
        std::vector<double> post(4 * 70001);
        for (size_t i = 0; i < post.size(); i++) post[i] = (double)(i % 977) / 977.0;
        if (tracs_write_posterior_csv((dir + "/san_post.csv.gz").c_str(), post.data(), 70001, 4, 1) != 0) { printf("csv\n"); fails++; }
        const char *nm[] = {"a", "b", "c"};
        const uint64_t r[] = {0, 0, 1}, c[] = {1, 2, 2}, d[] = {3, 4, 5};
        const double x[] = {0.5, 1e-7, 3e22};
        uint64_t wrote = 0;
        const std::string p = dir + "/san_rows.csv";
        put(p, "h\n", false);
        if (tracs_write_distance_rows(p.c_str(), nm, r, c, d, nullptr, d, x, x, x, 3, 1, 1.0, "ref", &wrote) != 0 || wrote != 2) { printf("rows %llu\n", (unsigned long long)wrote); fails++; }
        tracs_edge_list *e = nullptr;
        if (tracs_read_distance_edges(p.c_str(), 3, 3.5, nullptr, 0, &e) != 0 || tracs_edges_count(e) != 1 || tracs_edges_n_names(e) != 3) { printf("edges\n"); fails++; }
        tracs_edges_free(e);
        e = nullptr;
        put(p, "h\na,b\n", false);
        if (tracs_read_distance_edges(p.c_str(), 3, 3.5, nullptr, 0, &e) == 0) { printf("edges: short row accepted\n"); fails++; tracs_edges_free(e); }
    }
    {
        std::vector<std::string> paths;
        for (int s = 0; s < 5; s++) {
            paths.push_back(dir + "/san_s" + std::to_string(s) + ".fa");
            put(paths.back(), ">x\n" + std::string(30000 + (s == 4 ? 0 : 0), "ACGTN"[s]) + "\n", s & 1);
        }
        const char *names[] = {"s0", "s1", "s2", "s3", "s4"};
        const char *pp[5];
        for (int s = 0; s < 5; s++) pp[s] = paths[s].c_str();
        double fr[5];
        uint64_t ln[5];
        const std::string out = dir + "/san_comb.fa.gz";
        if (tracs_combine_fasta(out.c_str(), names, pp, 5, 3, 6, fr, ln) != 0 || fr[4] != 1.0 || ln[0] != 30000) { printf("combine\n"); fails++; }
        tracs::FastaData fd;
        std::string err;
        if (tracs::read_fasta(out, fd, err) != 0 || fd.n != 5 || fd.L != 30000 || fd.names[3] != "s3") { printf("combine read back: %s\n", err.c_str()); fails++; }
        put(paths[2], ">x\nAC\n>y\nAC\n", false);
        if (tracs_combine_fasta(out.c_str(), names, pp, 5, 3, 6, fr, ln) == 0) { printf("combine: two records accepted\n"); fails++; }
    }
    {
        const double v[] = {0.0, -0.0, 1e300, 5e-324, 123456.789, 1e16, 1e-5};
        char buf[512];
        if (tracs_debug_format_floats(v, 7, buf, sizeof buf) <= 0) { printf("fmt\n"); fails++; }
    }
    printf(fails ? "FAILED %d\n" : "sanitizer driver: all host paths clean (%d failures)\n", fails);
    return fails != 0;
}
