/*
 * tracs_oracle.c -- CPU restatement of the TRACS all-pairs distance path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path, never the
 * thing shipped or measured: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (tracs_amd/) never links or imports it.
 *
 * Every function cites the reference lines (under /root/reference/) it restates.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_lprob_k_given_N, orc_trans_dist, orc_calculate_posteriors: PINNED against
 *     the reference's own known-answer tests (tests/test_llk.py:28-29,
 *     tests/test_trans_distance.py:29-42) and against oracle/_ref (the reference's
 *     src/transcluster.hpp + src/dmultinomial.hpp compiled from where they lie with
 *     setup.py's flags) through tests/golden/ fixtures.
 *   - orc_read_fasta: PINNED against oracle/_ref/kseq_dump (reference src/kseq.h).
 *   - orc_pairsnp / orc_pack: PARITY UNPINNED.  src/pairsnp.hpp needs Boost
 *     (dynamic_bitset, math/binomial) which this image lacks, so it is unbuildable
 *     here, and the reference's golden inputs (ambig.aln, long_filt.aln) are not in
 *     the tree.  The restatement is cross-checked against an independent per-site
 *     numpy brute force (oracle/oracle.py: brute_pairsnp) instead.
 */
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* IUPAC -> allele-set mask.  src/pairsnp.hpp:110-198: toupper, then A/C/G/T one
 * bit, M R W S Y K two, V H D B three, everything else (N, '-', '?', ...) all
 * four.  bit0=A bit1=C bit2=G bit3=T.                                        */
uint8_t orc_iupac_mask(int ch)
{
    switch (toupper(ch)) {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 4;
    case 'T': return 8;
    case 'M': return 1 | 2;
    case 'R': return 1 | 4;
    case 'W': return 1 | 8;
    case 'S': return 2 | 4;
    case 'Y': return 2 | 8;
    case 'K': return 4 | 8;
    case 'V': return 1 | 2 | 4;
    case 'H': return 1 | 2 | 8;
    case 'D': return 1 | 4 | 8;
    case 'B': return 2 | 4 | 8;
    default:  return 15;
    }
}

/* ------------------------------------------------------------------------- */
/* FASTA/FASTQ(.gz) reader restating src/kseq.h:170-208 (kseq_read) as used by
 * load_seqs (src/pairsnp.hpp:75-101):
 *   - skip to the first '>' or '@' anywhere in the stream (kseq.h:175-178);
 *   - name = bytes up to the first isspace() (kseq.h:181, ks_getuntil sep 0);
 *     rest of the header line is the comment (kseq.h:182);
 *   - sequence = every isgraph() byte until the next '>', '+' or '@' at ANY
 *     position (kseq.h:183-192);
 *   - '+' starts a quality block: skip the line, then read printable bytes
 *     until as many as the sequence (kseq.h:199-206); shorter => error -2.
 * A whole-file slurp replaces the 4 KiB stream; the byte-level rules are the same.
 * Returns 0, or -2 ("Error reading FASTA!", pairsnp.hpp:84-91), -4 ("variable
 * sequence lengths", pairsnp.hpp:94-98), -5 cannot open.                     */
typedef struct {
    size_t n, L;
    char *seq;     /* n*L bytes, raw (not upper-cased) */
    char *names;   /* NUL-separated */
    size_t names_bytes;
} orc_fasta;

static char *slurp_gz(const char *path, size_t *len)
{
    gzFile fp = gzopen(path, "r");
    if (!fp) return NULL;
    size_t cap = 1 << 20, n = 0;
    char *buf = (char *)malloc(cap);
    for (;;) {
        if (cap - n < (1 << 16)) { cap *= 2; buf = (char *)realloc(buf, cap); }
        int r = gzread(fp, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (r <= 0) break;
        n += (size_t)r;
    }
    gzclose(fp);
    *len = n;
    return buf;
}

int orc_read_fasta(const char *path, orc_fasta *out)
{
    memset(out, 0, sizeof(*out));
    size_t len = 0;
    char *buf = slurp_gz(path, &len);
    if (!buf) return -5;
    size_t p = 0, seqcap = 0, namecap = 0;
    int last_char = 0, rc = 0;
    char *rec = NULL; size_t reccap = 0;
    for (;;) {
        if (last_char == 0) {                       /* kseq.h:175-178 */
            while (p < len && buf[p] != '>' && buf[p] != '@') p++;
            if (p >= len) break;
            last_char = buf[p++];
        }
        if (p >= len) break;                        /* ks_getuntil < 0 => -1 */
        size_t ns = p;
        while (p < len && !isspace((unsigned char)buf[p])) p++;
        size_t nl = p - ns;
        int c = (p < len) ? buf[p++] : 0;
        if (c != '\n' && c != 0) { while (p < len && buf[p] != '\n') p++; if (p < len) p++; }
        size_t l = 0;
        c = -1;
        while (p < len) {                           /* kseq.h:183-192 */
            c = buf[p++];
            if (c == '>' || c == '+' || c == '@') break;
            if (isgraph((unsigned char)c)) {
                if (l + 1 > reccap) { reccap = reccap ? reccap * 2 : 1 << 16; rec = (char *)realloc(rec, reccap); }
                rec[l++] = (char)c;
            }
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;
        if (c == '+') {                             /* kseq.h:194-206 */
            while (p < len && buf[p] != '\n') p++;
            if (p >= len) { rc = -2; break; }
            p++;
            size_t ql = 0;
            while (p < len && ql < l) { int q = buf[p++]; if (q >= 33 && q <= 127) ql++; }
            if (p < len && ql >= l) p++;            /* the extra ks_getc of the loop test */
            last_char = 0;
            if (ql != l) { rc = -2; break; }
        }
        if (out->n > 0 && l != out->L) { rc = -4; break; }   /* pairsnp.hpp:94-98 */
        out->L = l;
        if ((out->n + 1) * l > seqcap) { seqcap = seqcap ? seqcap * 2 : (l * 16 + 64); if (seqcap < (out->n + 1) * l) seqcap = (out->n + 1) * l; out->seq = (char *)realloc(out->seq, seqcap); }
        memcpy(out->seq + out->n * l, rec, l);
        if (out->names_bytes + nl + 1 > namecap) { namecap = (namecap + nl + 1) * 2; out->names = (char *)realloc(out->names, namecap); }
        memcpy(out->names + out->names_bytes, buf + ns, nl);
        out->names[out->names_bytes + nl] = 0;
        out->names_bytes += nl + 1;
        out->n++;
        if (c != '>' && c != '@' && c != '+') break;   /* EOF inside the record */
    }
    free(rec);
    free(buf);
    if (rc) { free(out->seq); free(out->names); memset(out, 0, sizeof(*out)); }
    return rc;
}

void orc_free_fasta(orc_fasta *f) { free(f->seq); free(f->names); memset(f, 0, sizeof(*f)); }

/* ------------------------------------------------------------------------- */
/* Bit-plane packing.  load_seqs builds four L-bit sets per sample
 * (src/pairsnp.hpp:102-203).  Here: planes[(p*n + s)*W + w], 64-bit words,
 * W = ceil(L/64), plane order A,C,G,T; tail bits of the last word are zero.  */
size_t orc_words(size_t L) { return (L + 63) / 64; }

void orc_pack(const char *seq, size_t n, size_t L, uint64_t *planes)
{
    size_t W = orc_words(L);
    memset(planes, 0, 4 * n * W * sizeof(uint64_t));
    for (size_t s = 0; s < n; s++)
        for (size_t j = 0; j < L; j++) {
            uint8_t m = orc_iupac_mask((unsigned char)seq[s * L + j]);
            for (int p = 0; p < 4; p++)
                if (m & (1u << p)) planes[((size_t)p * n + s) * W + (j >> 6)] |= 1ull << (j & 63);
        }
}

/* ------------------------------------------------------------------------- */
/* Pair loop.  src/pairsnp.hpp:380-432.
 *   match = (Ai&Aj)|(Ci&Cj)|(Gi&Gj)|(Ti&Tj);  d = L - popcount(match)   (:398-403)
 *   emit iff d <= dist (signed int compare)                              (:405)
 *   nn = L - popcount((Ai&Ci&Gi&Ti)|(Aj&Cj&Gj&Tj))                       (:417-420)
 * rows i in [0,i_end), cols j in [max(j_start,i+1), n)                   (:382,395)
 * Output is row-major (i, then j), independent of thread count           (:372-376,451-455).
 * If rows==NULL only the count is returned.  d/nn for pad bits: pad bits are 0 in
 * every plane => never "match" => subtract nothing; we count matches over real
 * bits only, so d = L - matches is exact.                                    */
int64_t orc_pairsnp(const uint64_t *planes, size_t n, size_t L, size_t i_end, size_t j_start,
                    int dist, int n_threads,
                    uint64_t *rows, uint64_t *cols, uint64_t *dists, uint64_t *ncomp)
{
    size_t W = orc_words(L);
    const uint64_t *A = planes, *C = planes + n * W, *G = planes + 2 * n * W, *T = planes + 3 * n * W;
    int64_t *cnt = (int64_t *)calloc(i_end + 1, sizeof(int64_t));
    int *dtmp = NULL;
    /* pass 1: all d for the rows (kept, so pass 2 only fills) */
    size_t total_cols = n;
    dtmp = (int *)malloc(sizeof(int) * (i_end ? i_end : 1) * total_cols);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t ii = 0; ii < (int64_t)i_end; ii++) {
        size_t i = (size_t)ii;
        int64_t c = 0;
        size_t j0 = j_start > i + 1 ? j_start : i + 1;
        for (size_t j = j0; j < n; j++) {
            int64_t m = 0;
            for (size_t w = 0; w < W; w++) {
                uint64_t r = (A[i * W + w] & A[j * W + w]) | (C[i * W + w] & C[j * W + w]) |
                             (G[i * W + w] & G[j * W + w]) | (T[i * W + w] & T[j * W + w]);
                m += __builtin_popcountll(r);
            }
            int d = (int)((int64_t)L - m);
            dtmp[i * total_cols + j] = d;
            if (d <= dist) c++;
        }
        cnt[i + 1] = c;
    }
    for (size_t i = 0; i < i_end; i++) cnt[i + 1] += cnt[i];
    int64_t total = cnt[i_end];
    if (rows) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
        for (int64_t ii = 0; ii < (int64_t)i_end; ii++) {
            size_t i = (size_t)ii;
            int64_t o = cnt[i];
            size_t j0 = j_start > i + 1 ? j_start : i + 1;
            for (size_t j = j0; j < n; j++) {
                int d = dtmp[i * total_cols + j];
                if (d > dist) continue;
                int64_t nm = 0;
                for (size_t w = 0; w < W; w++) {
                    uint64_t ni = A[i * W + w] & C[i * W + w] & G[i * W + w] & T[i * W + w];
                    uint64_t nj = A[j * W + w] & C[j * W + w] & G[j * W + w] & T[j * W + w];
                    nm += __builtin_popcountll(ni | nj);
                }
                rows[o] = i; cols[o] = j; dists[o] = (uint64_t)(int64_t)d; ncomp[o] = (uint64_t)((int64_t)L - nm);
                o++;
            }
        }
    }
    free(cnt); free(dtmp);
    return total;
}

/* Dense timing kernel for the cpu_baseline leg: d and nn for every i<j of the first
 * n_rows rows, no thresholding, returns a checksum so the work cannot be elided. */
uint64_t orc_pairsnp_rows_checksum(const uint64_t *planes, size_t n, size_t L, size_t n_rows, int n_threads)
{
    size_t W = orc_words(L);
    const uint64_t *A = planes, *C = planes + n * W, *G = planes + 2 * n * W, *T = planes + 3 * n * W;
    uint64_t sum = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : sum)
    for (int64_t ii = 0; ii < (int64_t)n_rows; ii++) {
        size_t i = (size_t)ii;
        for (size_t j = i + 1; j < n; j++) {
            int64_t m = 0, nm = 0;
            for (size_t w = 0; w < W; w++) {
                uint64_t ai = A[i * W + w], ci = C[i * W + w], gi = G[i * W + w], ti = T[i * W + w];
                uint64_t aj = A[j * W + w], cj = C[j * W + w], gj = G[j * W + w], tj = T[j * W + w];
                m += __builtin_popcountll((ai & aj) | (ci & cj) | (gi & gj) | (ti & tj));
                nm += __builtin_popcountll((ai & ci & gi & ti) | (aj & cj & gj & tj));
            }
            sum += (uint64_t)((int64_t)L - m) * 1000003ull + (uint64_t)((int64_t)L - nm);
        }
    }
    return sum;
}

/* ------------------------------------------------------------------------- */
/* transcluster.  src/transcluster.hpp.                                       */

/* :62-75 */
static double logaddexpd(double x, double y)
{
    double const tmp = x - y;
    if (x == y) return x + M_LN2;
    if (tmp > 0) return x + log1p(exp(-tmp));
    else if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp;
}

/* i*log(lamb*delta) as the shipped (-ffast-math) build evaluates it: the i == 0
 * term is 0 even when log() is -inf (delta == 0).  SURVEY.md 8c; checked against
 * oracle/_ref in tests/test_oracle_vs_ref.py.                                */
static double i_times_log(size_t i, double lg)
{
    return i == 0 ? 0.0 : (double)i * lg;
}

/* src/transcluster.hpp:90-129.  lgam[n] = lgamma(n) supplied by the caller.  */
void orc_lprob_k_given_N(size_t N, size_t k, double delta, double lamb, double beta,
                         const double *lgam, double *out2)
{
    double lprob, lhs;
    if (delta > 0) {
        lprob = ((double)(N + 1) * log(lamb) - delta * (lamb + beta) + (double)k * log(beta) - lgam[k + 1]);
        double pois_cdf = -INFINITY;
        for (size_t i = 0; i <= N; i++)
            pois_cdf = logaddexpd(i_times_log(i, log(lamb * delta)) - lgam[i + 1], pois_cdf);
        pois_cdf -= lamb * delta;
        lprob -= pois_cdf;
        double integral = -INFINITY;
        for (size_t i = 0; i <= N + k; i++)
            integral = logaddexpd(lgam[N + k + 1] - lgam[i + 1] - lgam[N + k - i + 1] +
                                      i_times_log(N + k - i, log(delta)) + lgam[i + 1] -
                                      (double)(i + 1) * log(lamb + beta),
                                  integral);
        integral -= lgam[N + 1];
        lhs = lprob;
        lprob += integral;
    } else {
        lprob = ((double)(N + 1) * log(lamb) + (double)k * log(beta) + lgam[N + k + 1] - lgam[N + 1] -
                 lgam[k + 1] - (double)(N + k + 1) * log(lamb + beta));
        lhs = lprob;
    }
    out2[0] = lprob; out2[1] = lhs;
}

/* lgamma(n) with the reference's table for n < 10000 (:253-258) continued by the
 * true function beyond it (the reference reads out of bounds there: SURVEY.md 7,
 * hard part 3 -- defined behaviour is ours).                                  */
static double lgam_at(const double *lgam, size_t n_tab, size_t n)
{
    return n < n_tab ? lgam[n] : lgamma((double)n);
}

/* src/transcluster.hpp:131-170 */
static void lprob_k_given_N_2(size_t N, size_t k, double delta, double lamb, double beta,
                              const double *lgam, size_t n_tab, double *out2)
{
    double lprob, lhs;
    if (delta > 0) {
        lprob = ((double)(N + 1) * log(lamb) + (double)k * log(beta) + lgam_at(lgam, n_tab, N + k + 1));
        lprob = lprob - lgam_at(lgam, n_tab, N + 1) - lgam_at(lgam, n_tab, k + 1) - delta * beta;
        double pois_cdf = -INFINITY;
        for (size_t i = 0; i <= N; i++)
            pois_cdf = logaddexpd(i_times_log(i, log(lamb * delta)) - lgam_at(lgam, n_tab, i + 1), pois_cdf);
        lprob -= pois_cdf;
        double integral = -INFINITY;
        for (size_t i = 0; i <= N + k; i++)
            integral = logaddexpd(i_times_log(N + k - i, log(delta)) - lgam_at(lgam, n_tab, N + k - i + 1) -
                                      (double)(i + 1) * log(lamb + beta),
                                  integral);
        lhs = lprob;
        lprob += integral;
    } else {
        lprob = ((double)(N + 1) * log(lamb) + (double)k * log(beta) + lgam_at(lgam, n_tab, N + k + 1) -
                 lgam_at(lgam, n_tab, N + 1) - lgam_at(lgam, n_tab, k + 1) -
                 (double)(N + k + 1) * log(lamb + beta));
        lhs = lprob;
    }
    out2[0] = lprob; out2[1] = lhs;
}

void orc_lprob_k_given_N_2(size_t N, size_t k, double delta, double lamb, double beta, double *out2)
{
    size_t n_tab = N + k + 2;
    double *lg = (double *)malloc(n_tab * sizeof(double));
    for (size_t i = 0; i < n_tab; i++) lg[i] = lgamma((double)i);
    lprob_k_given_N_2(N, k, delta, lamb, beta, lg, n_tab, out2);
    free(lg);
}

/* src/transcluster.hpp:173-188 */
static double upper_bound_E(const double *lgam, size_t n_tab, double delta, double lamb, double beta, size_t N)
{
    double pois_cdf = -INFINITY;
    for (size_t i = 0; i <= N; i++)
        pois_cdf = logaddexpd(i_times_log(i, log(lamb * delta)) - lgam_at(lgam, n_tab, i + 1), pois_cdf);
    return exp(log(beta) + delta * lamb + log((double)(N + 1)) - (log(lamb) + pois_cdf));
}

/* src/transcluster.hpp:191-238.  The (N,k,delta) memo (:220-225) only saves time. */
static double expected_k(int N, double delta, double lamb, double beta, double threshold_Ek,
                         const double *lgam, size_t n_tab, int *k_stop)
{
    double lprob = -INFINITY, elprob = -INFINITY, upper_bound, diff_bound;
    int k = 1;
    upper_bound = upper_bound_E(lgam, n_tab, delta, lamb, beta, (size_t)N);
    diff_bound = threshold_Ek + 1;
    while ((diff_bound > threshold_Ek) && (k < 10000)) {
        double r[2];
        lprob_k_given_N_2((size_t)N, (size_t)k, delta, lamb, beta, lgam, n_tab, r);
        lprob = logaddexpd(lprob, r[0] + log((double)k));
        elprob = logaddexpd(elprob, r[1] + log((double)k) + delta * (lamb + beta) -
                                        (double)(N + k + 1) * log(lamb + beta));
        diff_bound = upper_bound - exp(elprob);
        k++;
    }
    if (k_stop) *k_stop = k;
    return exp(lprob);
}

double orc_expected_k(int N, double delta, double lamb, double beta, double threshold_Ek, int *k_stop)
{
    size_t n_tab = 10000;
    double *lg = (double *)malloc(n_tab * sizeof(double));
    for (size_t i = 0; i < n_tab; i++) lg[i] = lgamma((double)i);
    double r = expected_k(N, delta, lamb, beta, threshold_Ek, lg, n_tab, k_stop);
    free(lg);
    return r;
}

/* Trace of the same loop for the tests: partial[k] = exp(lprob) after the iteration with index k
 * (k = 1..), diffs[k] = diff_bound after it, for `extra` iterations PAST the reference's stopping
 * point.  Where the bound `upper` is huge (few SNPs over a long time gap) the reference's loop ends
 * when exp(elprob) meets `upper` to the last bit, i.e. the truncation point is decided by rounding
 * noise (it moves with libm / FMA contraction); tests use the trace to accept a truncation within a
 * few k of the oracle's there (DESIGN.md "E(K) truncation").  Returns k_stop (the value of k when the
 * reference's while-loop exits).                                                                    */
int orc_expected_k_trace(int N, double delta, double lamb, double beta, double threshold_Ek, int extra,
                         int cap, double *partial, double *diffs, double *upper_out)
{
    size_t n_tab = 10000;
    double *lg = (double *)malloc(n_tab * sizeof(double));
    for (size_t i = 0; i < n_tab; i++) lg[i] = lgamma((double)i);
    double lprob = -INFINITY, elprob = -INFINITY;
    double upper_bound = upper_bound_E(lg, n_tab, delta, lamb, beta, (size_t)N);
    double diff_bound = threshold_Ek + 1;
    int k = 1, k_stop = -1;
    while (k < cap) {
        if (k_stop < 0 && !((diff_bound > threshold_Ek) && (k < 10000))) k_stop = k;
        if (k_stop >= 0 && k >= k_stop + extra) break;
        double r[2];
        lprob_k_given_N_2((size_t)N, (size_t)k, delta, lamb, beta, lg, n_tab, r);
        lprob = logaddexpd(lprob, r[0] + log((double)k));
        elprob = logaddexpd(elprob, r[1] + log((double)k) + delta * (lamb + beta) -
                                        (double)(N + k + 1) * log(lamb + beta));
        diff_bound = upper_bound - exp(elprob);
        partial[k] = exp(lprob);
        diffs[k] = diff_bound;
        k++;
    }
    if (k_stop < 0) k_stop = k;
    if (upper_out) *upper_out = upper_bound;
    free(lg);
    return k_stop;
}

/* src/transcluster.hpp:240-287.  Returns p0 (log) and eK per pair.  The two hash
 * caches (:245-246) are replaced by a sort-free linear memo over distinct keys:
 * same values, since every cached entry is a pure function of its key.        */
void orc_trans_dist(const int *snpdiff, const double *datediff, size_t n, double lamb, double beta,
                    double threshold_Ek, double *p0, double *eK)
{
    size_t n_tab = 10000;                                  /* :253-258 */
    double *lg = (double *)malloc(n_tab * sizeof(double));
    for (size_t i = 0; i < n_tab; i++) lg[i] = lgamma((double)i);
    /* tiny open-addressing memo keyed on (N, bits(delta)) */
    size_t cap = 1; while (cap < 2 * n + 16) cap <<= 1;
    int64_t *slot = (int64_t *)malloc(cap * sizeof(int64_t));
    for (size_t i = 0; i < cap; i++) slot[i] = -1;
    for (size_t i = 0; i < n; i++) {
        uint64_t db; memcpy(&db, &datediff[i], 8);
        uint64_t h = (db * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)snpdiff[i] * 0xC2B2AE3D27D4EB4Full);
        h ^= h >> 29;
        size_t s = (size_t)h & (cap - 1);
        for (;;) {
            if (slot[s] < 0) {
                slot[s] = (int64_t)i;
                eK[i] = expected_k(snpdiff[i], datediff[i], lamb, beta, threshold_Ek, lg, n_tab, NULL);
                double r[2];
                lprob_k_given_N_2((size_t)snpdiff[i], 0, datediff[i], lamb, beta, lg, n_tab, r);
                p0[i] = r[0];
                break;
            }
            size_t r0 = (size_t)slot[s];
            if (snpdiff[r0] == snpdiff[i] && memcmp(&datediff[r0], &datediff[i], 8) == 0) {
                eK[i] = eK[r0]; p0[i] = p0[r0];
                break;
            }
            s = (s + 1) & (cap - 1);
        }
    }
    free(slot); free(lg);
}

/* ------------------------------------------------------------------------- */
/* src/dmultinomial.hpp:8-86.  counts [L][K] row-major f64, alphas length K.   */
void orc_calculate_posteriors(const double *counts, size_t L, size_t K, const double *alphas_in,
                              int keep, double expected, double *post)
{
    double alphas[16];
    size_t idx[16];
    for (size_t j = 0; j < K; j++) alphas[j] = alphas_in[j];
    for (size_t a = 1; a < K; a++) {                    /* sort desc (:13) */
        double v = alphas[a]; size_t b = a;
        while (b > 0 && alphas[b - 1] < v) { alphas[b] = alphas[b - 1]; b--; }
        alphas[b] = v;
    }
    double a0 = 0.0;
    for (size_t j = 0; j < K; j++) a0 += alphas[j];     /* :14 */
    double a_min = alphas[0] / a0;                      /* :15 */
    for (size_t i = 0; i < L; i++) {
        const double *row = counts + i * K;
        double *res = post + i * K;
        double denom = 0;
        for (size_t j = 0; j < K; j++) { denom += row[j]; idx[j] = j; }   /* :38-42 */
        for (size_t a = 1; a < K; a++) {                /* stable argsort desc (:45-47) */
            size_t v = idx[a]; size_t b = a;
            while (b > 0 && row[idx[b - 1]] < row[v]) { idx[b] = idx[b - 1]; b--; }
            idx[b] = v;
        }
        size_t alpha_index = 0;
        for (size_t j = 0; j < K; j++) {                /* :51-66 */
            if (denom <= 0) {
                res[j] = a_min;
            } else {
                res[idx[j]] = (row[idx[j]] + alphas[alpha_index]) / (denom + a0);
                if ((j < K - 1) && (row[idx[j]] != row[idx[j + 1]])) alpha_index += 1;
            }
        }
        for (size_t j = 0; j < K; j++) {                /* :69-82 */
            if (res[j] <= expected) {
                if (keep && (row[j] > 0)) res[j] = expected;
                else res[j] = 0.0;
            }
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Recombination filter.  src/pairsnp.hpp:223-318 (filter_recomb, range_count) with
 * cached_binomial_cdf (:41-58) = boost::math::cdf(binomial_distribution(n, p), k).
 * PARITY UNPINNED: Boost is absent and the reference's golden input (long_filt.aln,
 * tests/test_pairsnp.py:14-21) is not in the tree.  Boost documents
 *   cdf(binomial(n, p), k) = ibetac(k + 1, n - k, p)  for k < n,  1 for k = n;
 * here ibetac comes from the Lentz continued fraction for the regularised incomplete
 * beta function (checked against scipy.stats.binom.cdf in tests/test_filter_recomb.py). */
static double betacf(double a, double b, double x)
{
    const double FPMIN = 1e-300, EPS = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < FPMIN) d = FPMIN;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < EPS) break;
    }
    return h;
}

static double ibeta_reg(double a, double b, double x)      /* I_x(a, b) */
{
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
    return 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}

double orc_binomial_cdf(int n, double p, int k)
{
    if (k >= n) return 1.0;
    if (k < 0) return 0.0;
    return 1.0 - ibeta_reg((double)k + 1.0, (double)(n - k), p);      /* ibetac(k+1, n-k, p) */
}

/* positions[] = sorted SNP sites of one pair (the set bits of the flipped match set, :254). */
uint64_t orc_filter_recomb_positions(const int64_t *pos, int64_t d_count, int64_t aln_length)
{
    double d = (double)d_count;
    if (d <= 1) return (uint64_t)d_count;                                   /* :259-261 */
    double p = d / (double)aln_length;                                      /* :265 */
    double p_value_threshold = 0.05 / d;                                    /* :266 */
    int window_size_half = (int)(1.0 / p / 2.0 + 1);                        /* :269 */
    if (window_size_half > 5000) window_size_half = 5000;                   /* :270 */
    if (window_size_half < 50) window_size_half = 50;                       /* :271 */
    uint64_t filtered_d = 0;
    for (int64_t t = 0; t < d_count; t++) {                                 /* :281 */
        int i = (int)pos[t];
        int64_t left = i - window_size_half; if (left < 0) left = 0;        /* :284 */
        int64_t right = (int64_t)i + window_size_half + 1; if (right > aln_length) right = aln_length;   /* :285 */
        /* range_count :223-248: scan from the first set bit; count those in [left, right), span first..last */
        int64_t count = 0, first = 0, length = 0;
        for (int64_t u = 0; u < d_count && pos[u] < right; u++)
            if (pos[u] >= left) {
                if (count == 0) first = pos[u];
                count++;
                length = pos[u] - first + 1;
            }
        if (count > 1) {                                                    /* :294-309 */
            double p_value = 1.0 - orc_binomial_cdf((int)length, p, (int)count);
            if (p_value >= p_value_threshold) filtered_d++;
        } else {
            filtered_d++;
        }
    }
    return filtered_d;
}

/* filtered distance of every listed pair: planes as orc_pack; rows/cols index samples. */
void orc_filter_recomb_pairs(const uint64_t *planes, size_t n, size_t L, const uint64_t *rows, const uint64_t *cols,
                             size_t n_pairs, int n_threads, uint64_t *filt)
{
    size_t W = orc_words(L);
    const uint64_t *A = planes, *C = planes + n * W, *G = planes + 2 * n * W, *T = planes + 3 * n * W;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t t = 0; t < (int64_t)n_pairs; t++) {
        size_t i = rows[t], j = cols[t];
        int64_t cap = 1024, cnt = 0;
        int64_t *pos = (int64_t *)malloc(cap * sizeof(int64_t));
        for (size_t w = 0; w < W; w++) {
            uint64_t m = (A[i * W + w] & A[j * W + w]) | (C[i * W + w] & C[j * W + w]) | (G[i * W + w] & G[j * W + w]) |
                         (T[i * W + w] & T[j * W + w]);
            uint64_t snp = ~m;                                              /* res.flip() :254 (L bits only) */
            if (w == W - 1 && (L & 63)) snp &= (1ull << (L & 63)) - 1;
            while (snp) {
                int b = __builtin_ctzll(snp);
                snp &= snp - 1;
                if (cnt == cap) { cap *= 2; pos = (int64_t *)realloc(pos, cap * sizeof(int64_t)); }
                pos[cnt++] = (int64_t)(w * 64 + b);
            }
        }
        filt[t] = orc_filter_recomb_positions(pos, cnt, (int64_t)L);
        free(pos);
    }
}
