"""ctypes face of oracle/liboracle.so plus an independent numpy brute force.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under tracs_amd/ may import this module.

Reference lines restated by each function are cited in oracle/tracs_oracle.c.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(quiet=True):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u64p = C.POINTER(C.c_uint64)
        dp = C.POINTER(C.c_double)
        L.orc_iupac_mask.restype = C.c_uint8
        L.orc_iupac_mask.argtypes = [C.c_int]
        L.orc_words.restype = C.c_size_t
        L.orc_words.argtypes = [C.c_size_t]
        L.orc_pack.restype = None
        L.orc_pack.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, u64p]
        L.orc_pairsnp.restype = C.c_int64
        L.orc_pairsnp.argtypes = [u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                  u64p, u64p, u64p, u64p]
        L.orc_pairsnp_rows_checksum.restype = C.c_uint64
        L.orc_pairsnp_rows_checksum.argtypes = [u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        L.orc_lprob_k_given_N.restype = None
        L.orc_lprob_k_given_N.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, dp, dp]
        L.orc_lprob_k_given_N_2.restype = None
        L.orc_lprob_k_given_N_2.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, dp]
        L.orc_expected_k.restype = C.c_double
        L.orc_expected_k.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.orc_expected_k_trace.restype = C.c_int
        L.orc_expected_k_trace.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                           dp, dp, dp]
        L.orc_trans_dist.restype = None
        L.orc_trans_dist.argtypes = [C.POINTER(C.c_int), dp, C.c_size_t, C.c_double, C.c_double, C.c_double, dp, dp]
        L.orc_calculate_posteriors.restype = None
        L.orc_calculate_posteriors.argtypes = [dp, C.c_size_t, C.c_size_t, dp, C.c_int, C.c_double, dp]
        L.orc_read_fasta.restype = C.c_int
        L.orc_read_fasta.argtypes = [C.c_char_p, C.c_void_p]
        L.orc_free_fasta.restype = None
        L.orc_free_fasta.argtypes = [C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_binomial_cdf.restype = C.c_double
        L.orc_binomial_cdf.argtypes = [C.c_int, C.c_double, C.c_int]
        L.orc_filter_recomb_positions.restype = C.c_uint64
        L.orc_filter_recomb_positions.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.c_int64]
        L.orc_filter_recomb_pairs.restype = None
        L.orc_filter_recomb_pairs.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p, u64p, C.c_size_t, C.c_int, u64p]
        _LIB = L
    return _LIB


def ref_module():
    """The reference's transcluster/dmultinomial compiled in place (oracle/_ref), or None."""
    d = os.path.join(_HERE, "_ref")
    if not os.path.isdir(d):
        return None
    if d not in sys.path:
        sys.path.insert(0, d)
    try:
        import _tracs_ref
        return _tracs_ref
    except ImportError:
        return None


def kseq_dump_path():
    p = os.path.join(_HERE, "_ref", "kseq_dump")
    return p if os.path.exists(p) else None


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _Fasta(C.Structure):
    _fields_ = [("n", C.c_size_t), ("L", C.c_size_t), ("seq", C.c_void_p), ("names", C.c_void_p),
                ("names_bytes", C.c_size_t)]


_ERR = {-2: "Error reading FASTA!", -4: "Error reading FASTA, variable sequence lengths!",
        -5: "cannot open FASTA"}


def read_fasta(path):
    """-> (names: list[str], seqs: np.uint8[n, L]) with kseq semantics."""
    f = _Fasta()
    rc = lib().orc_read_fasta(os.fsencode(path), C.byref(f))
    if rc != 0:
        raise RuntimeError(_ERR.get(rc, "FASTA error %d" % rc))
    try:
        seqs = np.frombuffer(C.string_at(f.seq, f.n * f.L), dtype=np.uint8).reshape(f.n, f.L).copy() \
            if f.n * f.L else np.zeros((f.n, f.L), np.uint8)
        raw = C.string_at(f.names, f.names_bytes) if f.names_bytes else b""
        names = [x.decode("latin-1") for x in raw.split(b"\0")[:f.n]]
    finally:
        lib().orc_free_fasta(C.byref(f))
    return names, seqs


def pack(seqs):
    """seqs uint8[n, L] ASCII -> planes uint64[4, n, W] (A,C,G,T)."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    n, L = seqs.shape
    W = (L + 63) // 64
    planes = np.zeros((4, n, max(W, 1)), dtype=np.uint64)[:, :, :W].copy()
    if n and L:
        lib().orc_pack(seqs.ctypes.data_as(C.c_char_p), n, L, _u64p(planes))
    return planes


def pairsnp_arrays(seqs, n0=None, dist=2147483647, n_threads=1):
    """seqs uint8[n, L]; n0 = size of the first file in two-file mode (cross pairs only,
    src/pairsnp.hpp:352-360).  -> rows, cols, d, nn as uint64 arrays, row-major."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    n, L = seqs.shape
    planes = pack(seqs)
    i_end, j_start = (n, 0) if n0 is None else (n0, n0)
    null = C.POINTER(C.c_uint64)()
    cnt = lib().orc_pairsnp(_u64p(planes), n, L, i_end, j_start, int(dist), n_threads, null, null, null, null)
    rows = np.zeros(cnt, np.uint64); cols = np.zeros(cnt, np.uint64)
    d = np.zeros(cnt, np.uint64); nn = np.zeros(cnt, np.uint64)
    if cnt:
        lib().orc_pairsnp(_u64p(planes), n, L, i_end, j_start, int(dist), n_threads,
                          _u64p(rows), _u64p(cols), _u64p(d), _u64p(nn))
    return rows, cols, d, nn


def pairsnp_planes(planes, L, n0=None, dist=2147483647, n_threads=1):
    """The pair loop on already packed planes (uint64 [4, n, W])."""
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    n = planes.shape[1]
    i_end, j_start = (n, 0) if n0 is None else (n0, n0)
    # one call: pass 1 (all d) + pass 2 (compared sites of the emitted pairs), as the reference runs them
    if j_start == 0:
        cap = n * (n - 1) // 2
    else:
        cap = i_end * (n - j_start)
    out = [np.zeros(cap, np.uint64) for _ in range(4)]
    cnt = lib().orc_pairsnp(_u64p(planes), n, L, i_end, j_start, int(dist), n_threads, *[_u64p(a) for a in out]) if cap else 0
    return [a[:cnt] for a in out]


def binomial_cdf(n, p, k):
    return lib().orc_binomial_cdf(int(n), float(p), int(k))


def filter_recomb_positions(pos, L):
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    return int(lib().orc_filter_recomb_positions(pos.ctypes.data_as(C.POINTER(C.c_int64)), len(pos), int(L)))


def filter_recomb_pairs(seqs, rows, cols, n_threads=1):
    """Filtered SNP distance (src/pairsnp.hpp:251-318) of the listed pairs.  PARITY UNPINNED (Boost)."""
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    n, L = seqs.shape
    planes = pack(seqs)
    r = np.ascontiguousarray(rows, dtype=np.uint64)
    c = np.ascontiguousarray(cols, dtype=np.uint64)
    out = np.zeros(len(r), np.uint64)
    if len(r):
        lib().orc_filter_recomb_pairs(_u64p(planes), n, L, _u64p(r), _u64p(c), len(r), n_threads, _u64p(out))
    return out


def pairsnp(fasta, n_threads, dist, filter):
    """Oracle twin of TRACS.pairsnp (src/pairsnp.hpp:320-457)."""
    if len(fasta) < 1 or len(fasta) > 2:
        raise RuntimeError("Invalid number of fasta files!")
    names, seqs = read_fasta(fasta[0])
    n0 = None
    if len(fasta) == 2:
        names2, seqs2 = read_fasta(fasta[1])
        n0 = len(names)
        names = names + names2
        if seqs.shape[0] and seqs2.shape[0] and seqs.shape[1] != seqs2.shape[1]:
            raise RuntimeError("oracle: the two files differ in alignment length")
        seqs = np.concatenate([seqs, seqs2], axis=0)
    r, c, d, nn = pairsnp_arrays(seqs, n0=n0, dist=dist, n_threads=n_threads)
    filt = filter_recomb_pairs(seqs, r, c, n_threads).tolist() if filter else [0] * len(d)
    return (r.tolist(), c.tolist(), d.tolist(), names, filt, nn.tolist())


def pairsnp_rows_checksum(planes, L, n_rows, n_threads):
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    n = planes.shape[1]
    return lib().orc_pairsnp_rows_checksum(_u64p(planes), n, L, n_rows, n_threads)


# ---- independent brute force (per site, no bit tricks) ----------------------------------
_MASK = np.full(256, 15, np.uint8)
for _ch, _m in {"A": 1, "C": 2, "G": 4, "T": 8, "M": 3, "R": 5, "W": 9, "S": 6, "Y": 10, "K": 12,
                "V": 7, "H": 11, "D": 13, "B": 14}.items():
    _MASK[ord(_ch)] = _m
    _MASK[ord(_ch.lower())] = _m


def brute_pairsnp(seqs, n0=None, dist=2147483647):
    """d = #sites whose allele sets are disjoint; nn = #sites where neither is fully
    ambiguous -- written from the definition, not from the bit-plane code."""
    m = _MASK[np.asarray(seqs, np.uint8)]
    n = m.shape[0]
    i_end, j_start = (n, 0) if n0 is None else (n0, n0)
    R, Cc, D, NN = [], [], [], []
    for i in range(i_end):
        for j in range(max(j_start, i + 1), n):
            d = int(np.count_nonzero((m[i] & m[j]) == 0))
            if d <= dist:
                R.append(i); Cc.append(j); D.append(d)
                NN.append(int(np.count_nonzero((m[i] != 15) & (m[j] != 15))))
    return (np.array(R, np.uint64), np.array(Cc, np.uint64), np.array(D, np.uint64), np.array(NN, np.uint64))


# ---- transcluster / dmultinomial --------------------------------------------------------
def lprob_k_given_N(N, k, delta, lamb, beta, lgamma):
    lg = np.ascontiguousarray(lgamma, dtype=np.float64)
    out = np.zeros(2)
    lib().orc_lprob_k_given_N(N, k, delta, lamb, beta, _dp(lg), _dp(out))
    return (float(out[0]), float(out[1]))


def lprob_k_given_N_2(N, k, delta, lamb, beta):
    out = np.zeros(2)
    lib().orc_lprob_k_given_N_2(N, k, delta, lamb, beta, _dp(out))
    return (float(out[0]), float(out[1]))


def expected_k(N, delta, lamb, beta, thr):
    ks = C.c_int(0)
    v = lib().orc_expected_k(int(N), delta, lamb, beta, thr, C.byref(ks))
    return float(v), ks.value


def expected_k_trace(N, delta, lamb, beta, thr, extra=8):
    """-> dict(k_stop, upper, partial[k], diffs[k]) -- see orc_expected_k_trace."""
    cap = 10000 + extra + 2
    partial = np.full(cap, np.nan)
    diffs = np.full(cap, np.nan)
    up = C.c_double(0)
    ks = lib().orc_expected_k_trace(int(N), float(delta), lamb, beta, thr, extra, cap, _dp(partial), _dp(diffs),
                                    C.byref(up))
    return dict(k_stop=ks, upper=up.value, partial=partial, diffs=diffs)


def ek_conditioning(N, delta, lamb, beta, thr):
    """Classify the reference's stopping rule for one key:
    'saturated' (ran to k = 10000), 'well' (the threshold crossing is far above the rounding noise of
    exp(elprob), which is ~upper * 1e-13 after the accumulation) or 'ill' (decided by rounding)."""
    t = expected_k_trace(N, delta, lamb, beta, thr, extra=0)
    ks = t["k_stop"]
    noise = abs(t["upper"]) * 1e-12
    if ks >= 10000:
        # ran to the cap.  Robust only if the bound was never approached to within rounding noise; otherwise a
        # different rounding could have ended the loop early (the stop is decided by the last bits of exp(elprob))
        closest = np.nanmin(t["diffs"][1:ks])
        return ("saturated" if closest - thr > 100 * noise else "ill"), t
    last_above = t["diffs"][ks - 2] if ks >= 3 else np.inf      # diff after the last-but-one iteration
    at_stop = t["diffs"][ks - 1]
    margin = min(abs(last_above - thr), abs(at_stop - thr))
    return ("well" if margin > 100 * noise else "ill"), t


def trans_dist(snpdiff, datediff, lamb, beta, threshold_Ek):
    n = np.ascontiguousarray(snpdiff, dtype=np.int32)
    d = np.ascontiguousarray(datediff, dtype=np.float64)
    p0 = np.zeros(len(n)); eK = np.zeros(len(n))
    if len(n):
        lib().orc_trans_dist(n.ctypes.data_as(C.POINTER(C.c_int)), _dp(d), len(n), lamb, beta, threshold_Ek,
                             _dp(p0), _dp(eK))
    return p0, eK


def calculate_posteriors(counts, alphas, keep, threshold):
    c = np.ascontiguousarray(counts, dtype=np.float64)
    a = np.ascontiguousarray(alphas, dtype=np.float64)
    out = np.zeros_like(c)
    if c.shape[0]:
        lib().orc_calculate_posteriors(_dp(c), c.shape[0], c.shape[1], _dp(a), int(bool(keep)), threshold, _dp(out))
    return out


def find_dirichlet_priors(counts, max_iter=1000, tol=1e-5, method="FPI", error_filt_threshold=None):
    """numpy/scipy restatement of tracs/dirichlet_multinomial.py:9-73 (own structure; pinned by
    tests/golden/python_reference_golden.json, which holds the reference's outputs incl. the R MGLM known answer)."""
    from scipy.special import psi
    x = np.array(counts, dtype=np.float64)
    K = x.shape[1]
    if error_filt_threshold is not None:                       # :13-15
        with np.errstate(divide="ignore", invalid="ignore"):
            freq = x / x.sum(1, keepdims=True)
        x[freq < error_filt_threshold] = 0
    poly = np.count_nonzero(x, axis=1) > 1                    # :20-35
    if poly.sum() <= 5:
        out = np.zeros(K)
        out[-1] = 1.0
        return out
    x = np.sort(x[poly], axis=1)                               # :36
    tot = x.sum(1)
    alpha = x.mean(0) + 0.5                                    # :40
    for _ in range(max_iter):
        a0 = alpha.sum()
        if method == "LOO":                                    # :43-54
            new = alpha * (x / (x - 1 + alpha)).sum(0) / (tot / (tot - 1 + a0)).sum()
            done = np.max(np.abs(new - alpha)) < tol
            alpha = new
        else:                                                  # :56-68
            new = alpha * (psi(x + alpha) - psi(alpha)).sum(0) / (psi(tot + a0) - psi(a0)).sum()
            done = np.sum(np.abs(new - alpha)) < tol
            alpha = new if done else np.maximum(new, 1e-16)
        if done:
            break
    return np.sort(alpha)[::-1]                                # :70


def connected_components(n, I, J):
    """Labels as scipy.sparse.csgraph.connected_components(directed=False) returns them
    (tracs/cluster.py:126-129): component ids in order of each component's smallest node."""
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in zip(I, J):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    labels = np.empty(n, np.int32)
    ids = {}
    for v in range(n):
        r = find(v)
        if r not in ids:
            ids[r] = len(ids)
        labels[v] = ids[r]
    return labels


# ---- align post-pileup stage and combine --------------------------------------------------------
# PARITY UNPINNED: tracs/align.py and tracs/combine.py import pyfastx (absent here) and align() shells out to minimap2 /
# htsbox, so neither can be run in this container and the reference ships no fixture for them.  These are line-by-line
# restatements of the cited blocks; calculate_posteriors / find_dirichlet_priors inside them ARE pinned (see above).

_IUPAC_LUT = np.frombuffer(b"XACMGRSVTWYHKDBN", dtype=np.uint8)     # index = packbits(little) of (A,C,G,T) > 0, align.py:285-323


def pileup_counts(lines, contigs, require_both_strands=False):
    """tracs/align.py:444-473.  lines: iterable of str; contigs: [(name, length)] in reference-FASTA order."""
    npos = {"A": 0, "C": 1, "G": 2, "T": 3}
    per = {name: np.zeros((length, 4), dtype=float) for name, length in contigs}
    for raw in lines:
        f = raw.strip().split()
        contig, pos = f[0], int(f[1]) - 1
        alleles = f[-2].split(",")
        strands = f[-1].split(":")[1:]
        row = np.zeros(4, dtype=float)
        for nuc, a, b in zip(alleles, strands[0].split(","), strands[1].split(",")):
            a, b = int(a), int(b)
            if nuc not in npos or f[2] not in npos:
                continue
            if require_both_strands and (a == 0 or b == 0):
                a = b = 0
            row[npos[nuc]] = a + b
        if pos < 0 or pos >= per[contig].shape[0]:
            raise IndexError("position outside the contig")     # (the reference wraps pos 0 to the last row; we refuse)
        per[contig][pos, :] = row
    return np.concatenate([per[name] for name, _ in contigs]) if contigs else np.zeros((0, 4))


def call_sequence(all_counts, min_cov=5, error_threshold=0.01, consensus=False, keep_all=False, keep_cov_outliers=False,
                  alphas=None):
    """tracs/align.py:476-647 from the concatenated counts to the sequence string.
    -> dict(sequence=str or None (reference skipped), alphas, threshold, band, posterior (after the coverage rules)).
    alphas: use these instead of fitting (tests pass the device's fit so that a last-bit difference in an alpha cannot
    move a site across the threshold; the fit itself is compared separately)."""
    all_counts = np.array(all_counts, dtype=float)
    rs = np.sum(all_counts, 1)
    nz_cov = np.sum(all_counts[rs > 0,], 1)
    median_cov = np.median(nz_cov)
    out = dict(sequence=None, alphas=None, threshold=None, band=None, posterior=None, csv=None)
    if consensus:                                                                  # :482-516
        one = np.zeros_like(all_counts, dtype=int)
        one[np.arange(all_counts.shape[0]), np.argmax(all_counts, axis=1)] = 1
        one[rs < min_cov,] = 1
        seq = _IUPAC_LUT[np.packbits(one > 0, axis=1, bitorder="little").flatten()].tobytes().decode()
        if seq.count("N") / float(len(seq)) > 0.75:
            return out
        out["sequence"] = seq
        return out
    thr = max(min_cov / median_cov, error_threshold)                               # :521
    if np.sum(rs >= min_cov) / all_counts.shape[0] < 0.25:                         # :522,531-535
        return out
    if alphas is None:
        alphas = find_dirichlet_priors(all_counts, method="FPI", error_filt_threshold=error_threshold)   # :537-539
    alphas = np.asarray(alphas, dtype=float)
    if thr <= alphas[1] / (median_cov + np.sum(alphas)):                           # :541-549
        thr = alphas[1] / (median_cov + np.sum(alphas)) + 0.01
    band = None
    use_band = (not keep_cov_outliers) and median_cov > 50 and alphas[1] / np.sum(alphas) > thr      # :552-556
    if use_band:
        lo = alphas[1] / thr - np.sum(alphas)                                      # :557
        lq = np.quantile(nz_cov, [0.25, 0.5])                                      # :559
        hi = lq[0] - 1.5 * (lq[1] - lq[0])                                         # :560
        band = (lo, hi)
    post = calculate_posteriors(all_counts, alphas, keep_all, thr)                 # :575-577
    out["csv"] = post.copy()                                                       # what np.savetxt writes (:580-596)
    if use_band and band[1] > band[0]:                                             # :599-611
        post[(rs <= band[1]) & (rs >= band[0]),] = 1
    post[rs < min_cov,] = 1                                                        # :613
    seq = _IUPAC_LUT[np.packbits(post > 0, axis=1, bitorder="little").flatten()].tobytes().decode()   # :616-622
    out.update(alphas=alphas, threshold=thr, band=band, posterior=post)
    if seq.count("N") / float(len(seq)) > 0.75:                                    # :626-630
        return out
    out["sequence"] = seq
    return out


def posterior_csv_text(post):
    """np.savetxt(fmt='%0.5f', delimiter=',') + the extra newline (tracs/align.py:589-596), as bytes."""
    import io
    b = io.BytesIO()
    np.savetxt(b, post, delimiter=",", newline="\n", fmt="%0.5f")
    b.write(b"\n")
    return b.getvalue()


def combined_fasta_text(records):
    """tracs/combine.py:227-238.  records: [(sample, sequence)] -> (text, {sample: (frac_N, length)})."""
    text, ncov = "", {}
    for sample, seq in records:
        text += ">%s\n%s\n" % (sample, seq)
        ncov[sample] = (seq.count("N") / len(seq), len(seq))
    return text, ncov
