/*
 * tracs_hip.h -- C ABI of libtracs_hip.so, the MI355X (gfx950) implementation of the TRACS
 * all-pairs distance path.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Two layers:
 *   (1) HOST entry points -- exactly what the reference's pybind11 module binds
 *       (/root/reference/src/python_bindings.cpp:12-25).  Host pointers in, host pointers out.
 *   (2) DEVICE entry points -- the same kernels with device pointers and a hipStream_t
 *       (passed as void*), for callers that keep the alignment resident in HBM (bench.py,
 *       the multi-GPU driver, `tracs distance`).  Asynchronous on the given stream.
 *
 * Threading: entry points that run kernels are serialised per device inside the library (scratch buffers and the cached
 * state of a tracs_alignment are shared), and scratch is stream-ordered: a call that arrives on a different stream than
 * the previous call on that device synchronises the device first.  Use one stream per device for overlap-free pipelines.
 *
 * Every function returns 0 on success or a negative TRACS_E_* code; tracs_last_error()
 * returns the message for the calling thread (the Python layer raises RuntimeError with it,
 * reproducing the reference's messages: src/pairsnp.hpp:86,90,96-97,342).
 */
#ifndef TRACS_HIP_H
#define TRACS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRACS_OK            0
#define TRACS_E_ARG        -1   /* bad argument (e.g. "Invalid number of fasta files!") */
#define TRACS_E_FASTA      -2   /* "Error reading FASTA!"                        (pairsnp.hpp:86,90) */
#define TRACS_E_RAGGED     -4   /* "Error reading FASTA, variable sequence lengths!" (pairsnp.hpp:96-97) */
#define TRACS_E_OPEN       -5   /* file cannot be opened (the reference does not diagnose this) */
#define TRACS_E_HIP        -6   /* HIP runtime error / no device */
#define TRACS_E_NOMEM      -7
#define TRACS_E_INTERRUPTED -8  /* "Interrupted by user!" (SIGINT during tracs_pairsnp; pairsnp.hpp:434-441 exits instead) */

const char *tracs_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int tracs_abi_version(void);
/* Number of visible HIP devices (<=0: none).  Does not throw. */
int tracs_device_count(void);

/* ===================================================================================== */
/* (1) HOST ENTRY POINTS -- the pybind11 surface                                          */
/* ===================================================================================== */

/* pairsnp(fasta, n_threads, dist, filter)
 *   replaces: src/python_bindings.cpp:12-13 -> src/pairsnp.hpp:320-457.
 *   fasta: 1 path (all i<j) or 2 paths (file0 x file1 only, pairsnp.hpp:352-360); plain or gzip.
 *   n_threads: accepted for signature parity; the pair loop runs on the GPU.
 *   dist: emit pairs with d <= dist (signed int compare, pairsnp.hpp:405).
 *   filter: recombination filter (pairsnp.hpp:251-318; parity unpinned); 0 => filt_distances are `len` zeros
 *           (pairsnp.hpp:452 with combine_vectors :31).
 * The result is an opaque handle read through the accessors below; rows/cols/... are
 * row-major (i, then j) like the reference (pairsnp.hpp:451-455).                         */
typedef struct tracs_pairsnp_result tracs_pairsnp_result;
int tracs_pairsnp(const char *const *fasta, int n_fasta, int n_threads, int dist, int filter,
                  tracs_pairsnp_result **out);
size_t tracs_pairsnp_len(const tracs_pairsnp_result *r);          /* number of emitted pairs */
size_t tracs_pairsnp_nseq(const tracs_pairsnp_result *r);         /* records loaded, both files */
size_t tracs_pairsnp_seqlen(const tracs_pairsnp_result *r);       /* alignment length L */
const uint64_t *tracs_pairsnp_rows(const tracs_pairsnp_result *r);
const uint64_t *tracs_pairsnp_cols(const tracs_pairsnp_result *r);
const uint64_t *tracs_pairsnp_distances(const tracs_pairsnp_result *r);
const uint64_t *tracs_pairsnp_filt_distances(const tracs_pairsnp_result *r);
const uint64_t *tracs_pairsnp_ncompared(const tracs_pairsnp_result *r);
const char *tracs_pairsnp_name(const tracs_pairsnp_result *r, size_t i);
void tracs_pairsnp_free(tracs_pairsnp_result *r);

/* trans_dist(snpdiff, datediff, lamb, beta, threshold_Ek) -> (p0_log[n], eK[n])
 *   replaces: src/python_bindings.cpp:19-21 -> src/transcluster.hpp:240-287.               */
int tracs_trans_dist(const int32_t *snpdiff, const double *datediff, size_t n, double lamb, double beta,
                     double threshold_Ek, double *p0_log, double *eK);

/* lprob_k_given_N(N, k, delta, lamb, beta, lgamma) -> (lprob, lhs)
 *   replaces: src/python_bindings.cpp:15-17 -> src/transcluster.hpp:90-129.
 *   lgamma[i] = ln Gamma(i), length > N+k+1 (caller supplied, as in tests/test_llk.py:26).
 *   Batched: n independent (N,k,delta) triples share lamb, beta and the table.             */
int tracs_lprob_k_given_N(const uint64_t *N, const uint64_t *k, const double *delta, size_t n, double lamb,
                          double beta, const double *lgamma, size_t lgamma_len, double *lprob, double *lhs);

/* calculate_posteriors(counts[L,K], alphas[K], keep, threshold) -> posterior[L,K]
 *   replaces: src/python_bindings.cpp:23-25 -> src/dmultinomial.hpp:8-86.  K <= 8.          */
int tracs_calculate_posteriors(const double *counts, size_t L, size_t K, const double *alphas, int keep,
                               double threshold, double *posterior);

/* find_dirichlet_priors(counts[L,K], max_iter, tol, method, error_filt_threshold) -> alphas[K] (descending)
 *   replaces: tracs/dirichlet_multinomial.py:9-73 (the fit whose output calculate_posteriors consumes,
 *   tracs/align.py:536-538).  method 0 = Minka fixed point (the reference's default branch), 1 = leave-one-out;
 *   error_filt_threshold < 0 = None.  Fewer than 6 polymorphic sites -> (0,..,0,1) like the reference.        */
int tracs_find_dirichlet_priors(const double *counts, size_t L, size_t K, int max_iter, double tol, int method,
                                double error_filt_threshold, double *alphas_out, int *iters_out);

/* Threshold single-linkage clustering = connected components of the edge list, labelled as
 * scipy.sparse.csgraph.connected_components(directed=False) labels them
 *   replaces: tracs/cluster.py:126-129.  edges (I[e], J[e]) over nodes 0..n-1.             */
int tracs_connected_components(const int32_t *I, const int32_t *J, size_t n_edges, size_t n_nodes,
                               int32_t *labels, int32_t *n_components);

/* ===================================================================================== */
/* (2) DEVICE ENTRY POINTS                                                                */
/* ===================================================================================== */

/* A packed alignment resident in HBM: five bit planes (A,C,G,T and N = A&C&G&T) per sample,
 * 128-site groups, sample-minor: uint4[group][plane][n_pad] (DESIGN.md "Data layout").     */
typedef struct tracs_alignment tracs_alignment;
int tracs_alignment_create(size_t n, size_t L, tracs_alignment **out);
void tracs_alignment_free(tracs_alignment *a);
size_t tracs_alignment_n(const tracs_alignment *a);
size_t tracs_alignment_len(const tracs_alignment *a);
size_t tracs_alignment_bytes(const tracs_alignment *a);   /* HBM bytes of the packed planes */
void *tracs_alignment_planes(const tracs_alignment *a);   /* device pointer of the packed planes (tracs_alignment_bytes long) */
/* The planes were written through that pointer from outside the library (e.g. an RCCL broadcast from the rank that packed):
 * forget every cached derived form (consensus planes, sparse lists, tile schedule stay valid per geometry).          */
int tracs_alignment_touch(tracs_alignment *a);
/* Multi-GPU ranks (every rank holds the whole alignment and computes its row panels): this handle will only be
 * asked for rows inside the given [begin, end) ranges (up to 2; n_ranges = 0 lifts the promise).  What is built once per pack
 * PER ROW -- the per-sample lists of the site classes -- is then built for those rows only (1 / ranks of that work); a dense
 * call for other rows fails with TRACS_E_ARG rather than returning results from lists that were never built.                */
int tracs_alignment_hint_rows(tracs_alignment *a, const size_t *ranges, int n_ranges);
/* Pack `count` samples of ASCII (IUPAC, any case; load_seqs pairsnp.hpp:107-199) into samples
 * [first, first+count).  `ascii` is count*L bytes, row-major; host or device pointer
 * (ascii_on_device).  The pack itself is a HIP kernel either way.                           */
int tracs_alignment_pack(tracs_alignment *a, const uint8_t *ascii, size_t first, size_t count,
                         int ascii_on_device, void *stream);
/* Read FASTA/FASTQ(.gz) with kseq semantics (src/kseq.h:170-208) and pack it.  names_out
 * receives a malloc'ed block of NUL-separated names (free with tracs_free).                 */
int tracs_alignment_from_fasta(const char *const *fasta, int n_fasta, tracs_alignment **out,
                               char **names_out, size_t *names_bytes, size_t *n_first_file);
void tracs_free(void *p);

/* Dense pair block: for rows i in [row_begin,row_end) and columns j in [max(col_begin,i+1), n)
 *   dist[i*ld + j]  = d(i,j)  = L - popcount(match)           (pairsnp.hpp:398-403)
 *   ncomp[i*ld + j] = nn(i,j) = L - popcount(N_i | N_j)       (pairsnp.hpp:417-420)
 * Other cells are not written.  dist/ncomp are device uint32 matrices with leading dimension
 * ld >= n; ncomp may be NULL (then only d is written).                                           */
int tracs_pairsnp_dense(const tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin,
                        uint32_t *dist, uint32_t *ncomp, size_t ld, void *stream);

/* Two streams.  The distances of a dense call are final before its compared-sites counts are (site classes: the counting pass and
 * the N co-occurrence walk only touch ncomp), and transcluster reads distances only.
 *   tracs_pairsnp_notify_distances(event)  the NEXT dense call on this thread's device records `event` (a hipEvent_t) on its
 *                                          stream once `dist` is final; a second stream that waits on it can run
 *                                          tracs_trans_dist_dense* beside the rest of the call (bench.py: memory-bound list walk and
 *                                          f64-bound key evaluation overlap);
 *   tracs_set_stream_policy(1)             the caller orders its streams itself (events).  Default 0: a call that arrives on another
 *                                          stream than the previous call on the device synchronises the device first, because the
 *                                          library's scratch buffers are only stream-ordered.  With policy 1 calls that share
 *                                          scratch (two transcluster calls; two first-calls on freshly packed handles; two
 *                                          posterior-codes calls with different parameters: their per-total table; two dense calls
 *                                          on alignments with minority lists: the fix-up's scratch rows) must be ordered by the
 *                                          caller; a dense call on a decided handle and a transcluster call share none. */
void tracs_pairsnp_notify_distances(void *event);
void tracs_set_stream_policy(int caller_orders_streams);

/* Thresholded form: identical (d and nn) for every pair with d <= dist_threshold.  A pair beyond the threshold (never
 * emitted, src/pairsnp.hpp:405) may come back with bit 31 of its distance set (0xFFFFFFFF, or a partial count | 2^31), or
 * (general alignments, and alignments cut into site classes) with a value > threshold that is not its distance, and an unspecified ncomp,
 * because workgroups stop reading the alignment once every pair of their tile is past the threshold: a pair is within the
 * threshold iff its cell, read as an UNSIGNED 32-bit value, is <= dist_threshold (what tracs_coo_count/fill test).
 * Long alignments take two passes (a 1/8 prefix over all tiles, then the rest over the surviving tiles only); this call
 * synchronises the stream once between them.                                                                          */
int tracs_pairsnp_dense_thr(const tracs_alignment *a, size_t row_begin, size_t row_end, size_t col_begin,
                            uint32_t *dist, uint32_t *ncomp, size_t ld, int32_t dist_threshold, void *stream);

/* Thresholded COO extraction from a dense block, row-major, same cell set as above.
 * Phase 1 (counts==per-row counts, device int64[row_end-row_begin+1] exclusive offsets on return),
 * phase 2 fills rows/cols/d/nn (device uint32) at those offsets.                            */
int tracs_coo_count(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                    size_t col_begin, int32_t dist_threshold, int64_t *offsets, void *stream);
int tracs_coo_fill(const uint32_t *dist, const uint32_t *ncomp, size_t ld, size_t n, size_t row_begin,
                   size_t row_end, size_t col_begin, int32_t dist_threshold, const int64_t *offsets,
                   uint32_t *rows, uint32_t *cols, uint32_t *d, uint32_t *nn, void *stream);
/* ... and two float64 panels of the same geometry (P(direct), E(K) as tracs_trans_dist_dense writes them) into the same positions */
int tracs_coo_fill_f64(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                       int32_t dist_threshold, const int64_t *offsets, const double *a, const double *b, double *out_a,
                       double *out_b, void *stream);

/* Threshold edges of a float64 panel (E(K) or P(direct) as written by tracs_trans_dist_dense; what `tracs cluster -D
 * expectedK|direct -c T` keeps, tracs/cluster.py:110-112): cells (i, j) of the same cell set whose SNP distance was emitted
 * (dist <= dist_threshold) and whose value is <= threshold, row-major.  Same two phases as tracs_coo_count/fill; vals (the
 * kept values) may be NULL.  This is the per-rank edge list the multi-GPU clustering path gathers.       */
int tracs_edges_count_f64(const double *val, const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                          size_t col_begin, int32_t dist_threshold, double threshold, int64_t *offsets, void *stream);
int tracs_edges_fill_f64(const double *val, const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                         size_t col_begin, int32_t dist_threshold, double threshold, const int64_t *offsets, uint32_t *rows,
                         uint32_t *cols, double *vals, void *stream);

/* Recombination filter (src/pairsnp.hpp:251-318) on emitted pairs.  rows/cols: device uint32[n_pairs];
 * pos_off: device int64[n_pairs+1] = exclusive scan of the pairs' SNP distances; positions: device uint32
 * workspace of pos_off[n_pairs] entries (receives each pair's sorted SNP sites); found[t] = SNP bits seen
 * (equals the distance); filt[t] = filtered distance.                                                  */
int tracs_filter_recomb_device(const tracs_alignment *a, const uint32_t *rows, const uint32_t *cols, size_t n_pairs,
                               const int64_t *pos_off, uint32_t *positions, uint32_t *found, uint32_t *filt,
                               void *stream);

/* The same filter without the positions (what `tracs distance --filter` and tracs_pairsnp(filter = 1) use): filt[t] =
 * filter_recomb (src/pairsnp.hpp:251-318) of the emitted pair (rows[t], cols[t]) whose SNP distance is d[t] (:405-413); all
 * device uint32[n_pairs].  A pair's SNP sites come from the two samples' departure lists (the sites at which a sample is neither
 * N nor exactly the site's reference base), built with two N bitmaps once per packed alignment and kept on the handle; pairs
 * whose lists do not fit a wave's LDS, and alignments too divergent for lists, are scanned like the reference scans them.
 * TRACS_FILTER_LISTS=0: always the scan; TRACS_FILTER_TABLE=0: the binomial tail summed per SNP instead of per distinct
 * (d, count).  Synchronises the stream (an internal consistency check: every pair's SNP count must equal d[t]).            */
int tracs_filter_recomb_pairs(tracs_alignment *a, const uint32_t *rows, const uint32_t *cols, const uint32_t *d, size_t n_pairs,
                              uint32_t *filt, void *stream);
/* transcluster on device arrays (same math as tracs_trans_dist).  workspace is managed
 * internally (hipMallocAsync on the stream).  exp_p0 != 0 writes exp(p0) (what
 * tracs/transcluster.py:38-39 returns with log=False).                                     */
int tracs_trans_dist_device(const int32_t *snpdiff, const double *datediff, size_t n, double lamb,
                            double beta, double threshold_Ek, int exp_p0, double *p0, double *eK,
                            void *stream);
/* Dense variant used by the distance pipeline: for the same cell set as tracs_pairsnp_dense,
 * delta(i,j) = |day_i - day_j| * 86400 / 31556952.0  (tracs/transcluster.py:5,26-33), N = dist[i*ld+j];
 * writes p0[i*ld+j] (exp'ed if exp_p0) and eK[i*ld+j].  Cells with dist > dist_threshold are skipped. */
int tracs_trans_dist_dense(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end,
                           size_t col_begin, int32_t dist_threshold, const int32_t *days, double lamb,
                           double beta, double threshold_Ek, int exp_p0, double *p0, double *eK,
                           void *stream);

/* Same over one or two row panels [b0,e0) (, [b1,e1)) in ONE pass (one key table): row_ranges = {b0,e0[,b1,e1]}. */
int tracs_trans_dist_dense2(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges,
                            size_t col_begin, int32_t dist_threshold, const int32_t *days, double lamb, double beta,
                            double threshold_Ek, int exp_p0, double *p0, double *eK, void *stream);

/* Multi-GPU form of the dense variant: the distinct (N, day gap) keys of the cells are evaluated into a dense key
 * table [n_max + 1][d_max + 1] (device f64, zeroed by the caller; log p0 and E(K)); a rank evaluates only the keys of its hash
 * class `part` of `parts`, the caller all-reduces (sums) the tables over the ranks, and tracs_trans_table_gather then writes
 * p0 / eK of every cell from the completed table.  *overflow (device uint32, zeroed by the caller) is set if a key falls
 * outside the table.  Same arithmetic as tracs_trans_dist_dense.                                                       */
int tracs_trans_table_dense(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                            int32_t dist_threshold, const int32_t *days, double lamb, double beta, double threshold_Ek,
                            int part, int parts, uint32_t n_max, uint32_t d_max, double *table_p0, double *table_eK,
                            uint32_t *overflow, void *stream);
int tracs_trans_table_gather(const uint32_t *dist, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin,
                             int32_t dist_threshold, const int32_t *days, uint32_t n_max, uint32_t d_max,
                             const double *table_p0, const double *table_eK, int exp_p0, double *p0, double *eK,
                             uint32_t *overflow, void *stream);

/* The distinct keys SPLIT over ranks that each hold rows of the matrix (site shards): src/transcluster.hpp:245-246,
 * 265-282 memoises per (N, delta) key -- here every key of the WHOLE matrix is evaluated by exactly one rank.
 *   tracs_trans_keys_words     uint32 words of a key bitmap (the (N, day gap) grid + four words: largest distance, smallest and
 *                              largest day + 2^31, "a key fell outside the grid")
 *   tracs_trans_keys_mark      the keys of the cells (i in the row ranges -- 0, 1 or 2 --, j >= max(col_begin, i + 1), d <=
 *                              dist_threshold) marked into `keys` (device, tracs_trans_keys_words() words, overwritten)
 *   tracs_trans_keys_merge     keys <- the union of `parts` bitmaps laid end to end in `all` (what an all-gather leaves)
 *   tracs_trans_keys_info      host info[4] = {distinct keys, largest distance, span of the days, 1 if the keys fit the grid};
 *                              synchronises the stream.  info[3] == 0: take tracs_trans_dist_dense2 on the own rows instead
 *   tracs_trans_keys_evaluate  the keys whose ordinal (set bits before it in the union) is = part mod parts, evaluated into
 *                              vals[2 (ordinal / parts)] = log p0, [.. + 1] = E(K) (device f64, per >= ceil(keys / parts) slots)
 *   tracs_trans_keys_gather    vals_all = the `parts` arrays of `per` slots each, end to end (what an all-gather leaves): p0 / eK
 *                              of every cell of the row ranges, as tracs_trans_dist_dense2 writes them
 * Same arithmetic per key as tracs_trans_dist_dense.                                                                          */
size_t tracs_trans_keys_words(void);
int tracs_trans_keys_mark(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges, size_t col_begin,
                          int32_t dist_threshold, const int32_t *days, uint32_t *keys, void *stream);
int tracs_trans_keys_merge(uint32_t *keys, const uint32_t *all, int parts, void *stream);
int tracs_trans_keys_info(const uint32_t *keys, uint64_t *info, void *stream);
int tracs_trans_keys_evaluate(const uint32_t *keys, const uint64_t *info, int part, int parts, double lamb, double beta,
                              double threshold_Ek, double *vals, size_t per, void *stream);
int tracs_trans_keys_gather(const uint32_t *dist, size_t ld, size_t n, const size_t *row_ranges, int n_ranges, size_t col_begin,
                            int32_t dist_threshold, const int32_t *days, const uint32_t *keys, const uint64_t *info,
                            const double *vals_all, int parts, size_t per, int exp_p0, double *p0, double *eK, void *stream);

/* calculate_posteriors on device arrays; counts/posterior are device f64 [L][K].            */
int tracs_calculate_posteriors_device(const double *counts, size_t L, size_t K, const double *alphas_host,
                                      int keep, double threshold, double *posterior, void *stream);
/* Fused production form (config 4): uint16 counts [L][4] -> 4-bit allele mask per site
 * (bit0=A..bit3=T set where posterior > 0, i.e. what tracs/align.py:616-622 turns into an
 * IUPAC letter), two sites per output byte (low nibble = even site).                        */
int tracs_posterior_codes_device(const uint16_t *counts, size_t L, const double *alphas_host, int keep,
                                 double threshold, uint8_t *codes, void *stream);

int tracs_find_dirichlet_priors_device(const double *counts, size_t L, size_t K, int max_iter, double tol, int method,
                                       double error_filt_threshold, double *alphas_out_host, int *iters_out,
                                       void *stream);

/* The same with the align stage's coverage rules (tracs/align.py:599-613): sites with total count < min_cov, or with
 * cov_lo <= total <= cov_hi (outlier band; pass cov_lo > cov_hi to disable), become fully ambiguous (mask 15).        */
int tracs_posterior_codes_cov_device(const uint16_t *counts, size_t L, const double *alphas_host, int keep,
                                     double threshold, uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes,
                                     void *stream);
/* Align stage before the posterior filter (tracs/align.py:473-516): one pass over device f64 counts [L][4] that
 *   - histograms the per-site coverage rs = A+C+G+T into hist[min(rs, nbins-1)] (device uint64[nbins], zeroed here): the
 *     stage's statistics (fraction covered, fraction >= min_cov, np.median / np.quantile of the non-zero coverages,
 *     :476-480,522,559) are order statistics of it;
 *   - narrows the counts to uint16 [L][4] (counts16, may be NULL) for the two code kernels;
 *   - sets *bad (device uint32) if any count is not an integer in [0, 65535].                                          */
int tracs_coverage_profile_device(const double *counts, size_t L, uint64_t *hist, size_t nbins, uint16_t *counts16,
                                  uint32_t *bad, void *stream);
/* --consensus calls (tracs/align.py:482-493): mask = the first allele with the largest count; 15 (N) where the total
 * count is below min_cov.  Same packed 4-bit output as tracs_posterior_codes_device.                                   */
int tracs_consensus_codes_device(const uint16_t *counts16, size_t L, uint32_t min_cov, uint8_t *codes, void *stream);
/* The three entry points above for per-allele counts beyond 65535 (deep amplicon / viral data; the reference works on
 * float64 counts of any depth, tracs/align.py:444-647): uint32 counts [L][4], each < 2^30.  The coverage histogram's last
 * bin then collects every total >= nbins - 1; callers needing order statistics up there take them from the counts.      */
int tracs_coverage_profile_device32(const double *counts, size_t L, uint64_t *hist, size_t nbins, uint32_t *counts32,
                                    uint32_t *bad, void *stream);
int tracs_posterior_codes_cov_device32(const uint32_t *counts, size_t L, const double *alphas_host, int keep,
                                       double threshold, uint32_t min_cov, double cov_lo, double cov_hi, uint8_t *codes,
                                       void *stream);
int tracs_consensus_codes_device32(const uint32_t *counts32, size_t L, uint32_t min_cov, uint8_t *codes, void *stream);
/* 4-bit masks -> IUPAC letters as tracs/align.py:285-323,616-622 ('X' for mask 0, 'N' for 15); ascii: device, L bytes. */
int tracs_codes_to_iupac_device(const uint8_t *codes, size_t L, uint8_t *ascii, void *stream);
/* Pack ONE sample straight from its 4-bit masks (no FASTA round trip): mask 0 ('X') is treated like every other
 * non-IUPAC letter, i.e. fully ambiguous (src/pairsnp.hpp:192-197).  codes: device, (L+1)/2 bytes.                   */
int tracs_alignment_pack_codes(tracs_alignment *a, const uint8_t *codes, size_t sample, void *stream);
/* The same for `count` samples [first, first+count) in one launch: sample k's masks start at codes + k * stride_bytes
 * (stride_bytes >= (L+1)/2; a multiple of 16 lets the kernel use 16-byte loads).                                       */
int tracs_alignment_pack_codes_batch(tracs_alignment *a, const uint8_t *codes, size_t stride_bytes, size_t first,
                                     size_t count, void *stream);

/* Connected components on device edge arrays; labels as tracs_connected_components.         */
int tracs_connected_components_device(const int32_t *I, const int32_t *J, size_t n_edges, size_t n_nodes,
                                      int32_t *labels, int32_t *n_components_host, void *stream);

/* ===================================================================================== */
/* (3) ON-DISK FORMATS either side of the path (host side)            */
/* ===================================================================================== */

/* `htsbox pileup -C -s 0` text (plain or gzip) -> allele counts, the parse loop of tracs/align.py:452-472.
 *   Per line: fields split at whitespace; contig = field 0, 1-based position = field 1, reference base = field 2,
 *   alleles = second-to-last field split at ',', strand counts = last field split at ':' (pieces 1 and 2, each split
 *   at ','); count[A|C|G|T] = forward + reverse for alleles that are exactly A/C/G/T when the reference base is one
 *   too; with require_both_strands an allele seen on one strand only counts 0; a repeated position overwrites.
 *   counts: host [sum(contig_lengths)][4] f64 in contig order (np.concatenate at :473), zeroed here first.
 *   Unknown contig, position outside its contig or a non-integer field is an error (the reference raises).      */
int tracs_pileup_counts(const char *path, const char *const *contig_names, const uint64_t *contig_lengths,
                        size_t n_contigs, int require_both_strands, double *counts, uint64_t *n_lines_out);

/* posterior [L][K] f64 -> gzip CSV exactly as np.savetxt(fmt="%0.5f", delimiter=",") + the trailing "\n" the
 * reference appends (tracs/align.py:580-596).  gzip_level 0..9.                                                   */
int tracs_write_posterior_csv(const char *path, const double *post, size_t L, size_t K, int gzip_level);

/* Starts, on a thread of its own, what a process's first call otherwise waits for: the HIP runtime, the device context and this
 * library's code object.  Returns at once; later calls wait where they need the device.  `tracs distance` calls it before it reads
 * its metadata (tracs/distance.py:161-166) -- a 10 x 100 kb alignment spends more time starting up than computing.  No-op unless
 * exactly one device is visible. */
void tracs_warm_up(void);

/* `tracs distance` for one alignment (tracs/distance.py:159-258) with the results on the device until the CSV rows.
 *   tracs_distance_open   read + pack the FASTA file(s) (1 file: all pairs; 2: file 0 x file 1, src/pairsnp.hpp:348-360); the names
 *                         (tracs_distance_nseq / _name) are what the caller looks the sampling dates up by
 *   tracs_distance_run    row panel by row panel: tracs_pairsnp_dense_thr, tracs_trans_dist_dense when `days` (host, one whole day
 *                         number per sample: delta = |day_i - day_j| x 86400 / 31556952.0 years, tracs/transcluster.py:26-33) is
 *                         given, COO extraction incl. P(direct) and E(K), one device-to-host pass in batches, rows formatted and
 *                         appended to `path` (the caller has written the header) in the reference's format and order.  days == NULL:
 *                         no metadata ("NA" for delta / P / E(K), 0 in the filtered column, :240-258); else "NA" in the filtered
 *                         column (:204) and, k_max >= 0, only rows with k_max >= E(K) (:222).  filter != 0 (--filter): the emitted
 *                         pairs go through tracs_filter_recomb_pairs (src/pairsnp.hpp:405-413), the filtered column holds their
 *                         filtered distances, and P(direct) / E(K) are those of the FILTERED distance (tracs/distance.py:183-193),
 *                         evaluated per emitted pair (tracs_trans_dist_device) instead of on the panel.                          */
typedef struct tracs_distance tracs_distance;
int tracs_distance_open(const char *const *fasta, int n_fasta, tracs_distance **out);
size_t tracs_distance_nseq(const tracs_distance *h);
const char *tracs_distance_name(const tracs_distance *h, size_t i);
int tracs_distance_run(tracs_distance *h, int dist, const int32_t *days, double lamb, double beta, double precision, double k_max,
                       const char *path, const char *ref, int filter, uint64_t *rows_written, uint64_t *n_pairs);
void tracs_distance_free(tracs_distance *h);

/* Rows of `tracs distance`'s CSV appended to path (tracs/distance.py:206-258; the caller writes the header, :157):
 *   names[rows[t]],names[cols[t]],str(delta),str(int(snpd)),str(P),str(E(K)),filtered,str(nn),ref
 * floats print exactly as Python's str(float) / str(numpy.float64).  with_dates = 0: "NA" for delta, P, E(K) (:240-258).
 * filt == NULL: "NA" in the filtered column (:204).  k_max < 0: no -K filter; else rows with k_max >= E(K) only (:222). */
int tracs_write_distance_rows(const char *path, const char *const *names, const uint64_t *rows, const uint64_t *cols,
                              const uint64_t *snpd, const uint64_t *filt, const uint64_t *ncomp, const double *delta,
                              const double *p_direct, const double *e_k, size_t n, int with_dates, double k_max,
                              const char *ref, uint64_t *rows_written);

/* `tracs cluster` input (tracs/cluster.py:100-116): distance CSV -> node names in first-appearance order (sampleA before
 * sampleB; ids continue after the n_seed names given, like the reference's function-level table) and the edges whose
 * field `column` (3 snp, 6 filter, 4 direct, 5 expectedK, :90-97) is <= threshold.  The header line is skipped; a field
 * that float() would refuse is an error carrying Python's message.                                                    */
typedef struct tracs_edge_list tracs_edge_list;
int tracs_read_distance_edges(const char *path, int column, double threshold, const char *const *seed_names,
                              size_t n_seed, tracs_edge_list **out);
size_t tracs_edges_count(const tracs_edge_list *e);
uint64_t tracs_edges_rows(const tracs_edge_list *e);          /* data lines read */
size_t tracs_edges_n_names(const tracs_edge_list *e);
const char *tracs_edges_name(const tracs_edge_list *e, size_t i);
const int32_t *tracs_edges_i(const tracs_edge_list *e);
const int32_t *tracs_edges_j(const tracs_edge_list *e);
void tracs_edges_free(tracs_edge_list *e);

/* write_alignment of tracs/combine.py:220-239: one single-record FASTA per sample -> "<ref>_combined.fasta.gz" with
 * records ">sample\nSEQUENCE\n" in input order.  Samples are compressed in parallel as separate gzip members
 * (n_threads <= 0: all cores; gzip_level < 0: 6).  frac_n[s] = count('N')/len, lengths[s] = len (the ncov dict);
 * a file with more than one record is an error ("... contains more than one sequence").                         */
int tracs_combine_fasta(const char *out_path, const char *const *sample_names, const char *const *fasta_paths,
                        size_t n, int n_threads, int gzip_level, double *frac_n, uint64_t *lengths);

/* ===================================================================================== */
/* (4) MULTI-GPU EXCHANGE -- RCCL over xGMI, one communicator rank per process             */
/* ===================================================================================== */

/* The reference is one process (OpenMP, src/pairsnp.hpp:380-382); SURVEY.md 8e / north_star: the pair space is block-
 * partitioned over the GPUs of one node, every rank holds the packed alignment, no collective during compute, the per-rank
 * distance panels are exchanged at the end.  These entry points are that exchange (csrc/comm.cpp over rccl.h; RCCL is opened
 * when the first communicator is made, so a single-GPU host needs none).  Every call enqueues on `stream` of the CURRENT
 * device (hipSetDevice before tracs_comm_create) and returns; results are valid once the stream has reached them.
 *   tracs_comm_unique_id   rank 0 draws an id (TRACS_COMM_ID_BYTES bytes) and hands it to the other ranks by any host means
 *   tracs_comm_create      collective: every rank of the job calls it with the same id and its own rank
 *   tracs_bcast_planes     the packed planes of `root` (the rank that read the FASTA) -> every rank's handle of the same
 *                          shape; the receiving handles forget every derived form (tracs_alignment_touch)
 *   tracs_allgather_panels in place: every rank lays out `base` alike, rank q's block is the `bytes` bytes at
 *                          base + offsets[q] (offsets: host array of `world` entries); afterwards every rank holds every block.
 *                          Row panels of a row-major pair matrix are such blocks (tracs_amd/partition.py)
 *   tracs_allreduce        dtype 0 int64, 1 float64, 2 uint32, 3 uint8; op 0 sum, 1 max, 2 min; in place
 *   tracs_reduce_scatter   in place: buf holds world blocks of count_per_rank elements; afterwards block `rank` of this rank's buf
 *                          is the reduction of every rank's block `rank`.  The site-sharded form of the path: d and the
 *                          compared-sites counts are sums over sites (pairsnp.hpp:398-403,417-420), so ranks that each hold a slice
 *                          of the sites compute all pairs over their slice and the sums arrive as row panels
 *   tracs_alltoall         send / recv hold `world` blocks of block_bytes bytes: block q of send -> rank q, where it becomes block
 *                          `rank` of recv (ncclAllToAll: every pair of ranks over its own xGMI link)
 *   tracs_tri_pack / _sum  the compact form of the site-sharded exchange (csrc/exchange.hip): the cells (i, j >= max(col_begin,
 *                          i + 1)) of the rows [row_begin, row_end) of a uint32 panel (mat indexed by ABSOLUTE row, leading
 *                          dimension ld) packed in `width` = 2 or 4 bytes per cell at element row_slot[i - row_begin] of `packed` (NULL: statistics only)
 *                          (device array; UINT64_MAX: skip the row), as mat[i][j] or, negate != 0, as base - mat[i][j] (the
 *                          compared-sites counts travel as their deficit below the slice's length); stats[0] = largest value
 *                          packed (atomic max), stats[1] += values that did not fit 16 bits.  _sum: mat[i][j] += add +/- the sum
 *                          over the blocks b != skip_block of recv[b * block_elems + row_slot[i - row_begin] + j - first column]
 *   tracs_send / _recv     `bytes` bytes to / from rank `peer` (variable-length COO payloads to the rank that writes the CSV)
 *   tracs_comm_rank / _world  what RCCL reports for the communicator (ncclCommUserRank / ncclCommCount); tracs_rccl_version:
 *                          ncclGetVersion (0: RCCL not available) */
#define TRACS_COMM_ID_BYTES 128
typedef struct tracs_comm tracs_comm;
int tracs_comm_unique_id(void *id, size_t cap);
int tracs_comm_create(const void *id, int rank, int world, tracs_comm **out);
void tracs_comm_free(tracs_comm *c);
int tracs_comm_rank(const tracs_comm *c);
int tracs_comm_world(const tracs_comm *c);
int tracs_bcast(tracs_comm *c, void *buf, size_t bytes, int root, void *stream);
int tracs_bcast_planes(tracs_comm *c, tracs_alignment *a, int root, void *stream);
int tracs_allgather_panels(tracs_comm *c, void *base, const size_t *offsets, size_t bytes, void *stream);
int tracs_allreduce(tracs_comm *c, void *buf, size_t count, int dtype, int op, void *stream);
int tracs_reduce_scatter(tracs_comm *c, void *buf, size_t count_per_rank, int dtype, int op, void *stream);
int tracs_alltoall(tracs_comm *c, const void *send, void *recv, size_t block_bytes, void *stream);
int tracs_tri_pack(const void *mat, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin, const uint64_t *row_slot,
                   int width, uint32_t base, int negate, void *packed, uint32_t *stats, void *stream);
int tracs_tri_sum(void *mat, size_t ld, size_t n, size_t row_begin, size_t row_end, size_t col_begin, const uint64_t *row_slot, int width,
                  const void *recv, size_t block_elems, int n_blocks, int skip_block, uint32_t add, int negate, void *stream);
int tracs_rccl_version(void);
int tracs_send(tracs_comm *c, const void *buf, size_t bytes, int peer, void *stream);
int tracs_recv(tracs_comm *c, void *buf, size_t bytes, int peer, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRACS_HIP_H */
