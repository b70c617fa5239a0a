"""The post-pileup stage of `tracs align` on the MI355X.

What /root/reference/tracs/align.py:444-647 does between the pileup text and the per-sample FASTA that `combine` and
`distance` consume, with the same defaults and decisions:

  pileup text(.gz) --pileup_counts--> counts [L,4] --coverage profile--> thresholds
       --find_dirichlet_priors (device)--> alphas --calculate_posteriors (device)--> posterior CSV (.csv.gz)
       --coverage rules + IUPAC code (device)--> <prefix>_posterior_counts_ref_<ref>.fasta

Read mapping and pileup generation (minimap2 / htsbox, tracs/pileup.py) stay outside: this module starts from their
output.  All arithmetic runs in libtracs_hip.so; there is no CPU fallback.
"""
import ctypes as C
import gzip
import os

import numpy as np

from . import _lib

COV_BINS = 4 * 65535 + 1            # every possible total of four uint16 counts


def read_contigs(fasta_path):
    """[(name, length)] of a reference genome FASTA (plain or gzip), names cut at the first whitespace
    (what pyfastx yields at tracs/align.py:449)."""
    opener = gzip.open if _is_gzip(fasta_path) else open
    contigs, name, length = [], None, 0
    with opener(fasta_path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    contigs.append((name, length))
                head = line[1:].split()
                name, length = (head[0].decode() if head else ""), 0
            elif name is not None:
                length += len(line.strip())
    if name is not None:
        contigs.append((name, length))
    return contigs


def _is_gzip(path):
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def pileup_counts(pileup_path, contigs, require_both_strands=False):
    """`htsbox pileup -C -s 0` text (plain or gzip) -> float64 [sum(lengths), 4] allele counts (tracs/align.py:452-473)."""
    L = _lib.load()
    if not os.path.exists(pileup_path):
        raise FileNotFoundError(pileup_path)
    names = (C.c_char_p * len(contigs))(*[c[0].encode() for c in contigs])
    lens = (C.c_uint64 * len(contigs))(*[int(c[1]) for c in contigs])
    total = int(sum(int(c[1]) for c in contigs))
    out = np.empty((total, 4), dtype=np.float64)
    nl = C.c_uint64(0)
    _lib.check(L.tracs_pileup_counts(os.fsencode(pileup_path), names, lens, len(contigs), int(bool(require_both_strands)),
                                     out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nl)))
    return out


def write_posterior_csv(path, posterior, gzip_level=6):
    """np.savetxt(fmt="%0.5f", delimiter=",") + the reference's trailing newline, gzip-compressed (tracs/align.py:580-596)."""
    p = np.ascontiguousarray(posterior, dtype=np.float64)
    _lib.check(_lib.load().tracs_write_posterior_csv(os.fsencode(path), p.ctypes.data_as(C.POINTER(C.c_double)), p.shape[0],
                                                     p.shape[1], int(gzip_level)))


# ---- order statistics of the coverage histogram, with numpy's own formulas ------------------------------------------
def _kth(cum, k):
    """value (= bin index) of the k-th smallest (0-based) sample given the inclusive cumulative histogram."""
    return int(np.searchsorted(cum, k + 1, side="left"))


def _np_lerp(a, b, t):
    """numpy.lib.function_base._lerp for scalars (method='linear')."""
    a, b = float(a), float(b)
    if t >= 0.5:
        return b - (b - a) * (1.0 - t)
    return a + (b - a) * t


def _quantile(cum, n, q):
    """np.quantile(x, q) (default linear interpolation) of the n samples behind `cum`."""
    h = (n - 1) * q
    lo = int(np.floor(h))
    hi = min(lo + 1, n - 1)
    return _np_lerp(_kth(cum, lo), _kth(cum, hi), h - lo)


class CoverageProfile:
    """What tracs/align.py:476-480,522 derive from the counts: rs = per-site total, nz = the non-zero ones."""

    def __init__(self, hist, L):
        self.L = int(L)
        self.hist = hist
        self.n_nonzero = int(hist[1:].sum())
        self._cum_nz = np.cumsum(hist[1:])                     # bin b of nz coverage = index b - 1
        self._cum_all = np.cumsum(hist)

    def frac_covered(self):                                    # np.sum(rs > 0) / L
        return self.n_nonzero / self.L

    def frac_at_least(self, min_cov):                          # np.sum(rs >= min_cov) / L
        m = int(np.ceil(min_cov))
        below = int(self._cum_all[m - 1]) if m >= 1 else 0
        return (self.L - below) / self.L

    def _nz_kth(self, k):
        return _kth(self._cum_nz, k) + 1

    def median_nonzero(self):                                  # np.median(nz_cov); nan for an empty set like numpy
        n = self.n_nonzero
        if n == 0:
            return float("nan")
        if n % 2:
            return float(self._nz_kth(n // 2))
        return (self._nz_kth(n // 2 - 1) + self._nz_kth(n // 2)) / 2.0

    def quantile_nonzero(self, q):                             # np.quantile(nz_cov, q)
        n = self.n_nonzero
        h = (n - 1) * q
        lo = int(np.floor(h))
        hi = min(lo + 1, n - 1)
        return _np_lerp(self._nz_kth(lo), self._nz_kth(hi), h - lo)


class SortedCoverageProfile(CoverageProfile):
    """The same statistics when some site is deeper than the histogram reaches (coverage >= COV_BINS - 1): order statistics
    are read from the sorted non-zero coverages (a device sort) instead of from histogram bins."""

    def __init__(self, rs, L):                                 # rs: torch int64 [L] per-site totals, on the device
        import torch
        self.L = int(L)
        self._rs = rs
        nz = rs[rs > 0]
        self._sorted = torch.sort(nz).values
        self.n_nonzero = int(nz.numel())

    def frac_at_least(self, min_cov):
        return int((self._rs >= int(np.ceil(min_cov))).sum().item()) / self.L

    def _nz_kth(self, k):
        return int(self._sorted[k].item())


def call_sequence(all_counts, min_cov=5, error_threshold=0.01, consensus=False, keep_all=False, keep_cov_outliers=False,
                  want_posterior=True, log=None):
    """Counts [L,4] (numpy, as pileup_counts returns them) -> the called sequence, on the GPU.

    Returns a dict: sequence (bytes, or None when the reference is skipped, tracs/align.py:503-507,531-535,626-630),
    codes (device uint8 tensor of packed 4-bit masks, ready for Alignment.pack_codes), alphas, threshold, band,
    posterior (numpy [L,4], what the reference writes to the .csv.gz; None for consensus or want_posterior=False),
    frac_covered, frac_min_cov, median_cov."""
    import torch
    from . import device as dev
    lib = _lib.require_gpu()
    say = log or (lambda *_: None)
    counts = torch.from_numpy(np.ascontiguousarray(all_counts, dtype=np.float64)).cuda()
    L = counts.shape[0]
    if counts.ndim != 2 or counts.shape[1] != 4:
        raise ValueError("call_sequence(): counts must be [sites, 4]")
    hist = torch.empty(COV_BINS, dtype=torch.int64, device="cuda")
    c16 = torch.empty((L, 4), dtype=torch.int16, device="cuda")
    bad = torch.empty(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.tracs_coverage_profile_device(dev._ptr(counts), L, dev._ptr(hist), COV_BINS, dev._ptr(c16), dev._ptr(bad),
                                                 dev._stream()))
    wide = bool(int(bad.item()))
    if wide:
        # some count is above 65535 (deep amplicon / viral data): the uint32 forms of the same kernels
        c16 = torch.empty((L, 4), dtype=torch.int32, device="cuda")
        _lib.check(lib.tracs_coverage_profile_device32(dev._ptr(counts), L, dev._ptr(hist), COV_BINS, dev._ptr(c16), dev._ptr(bad),
                                                       dev._stream()))
        if int(bad.item()):
            raise _lib.TracsError("allele counts must be non-negative integers below 2^30")
    hist_h = hist.cpu().numpy()
    if wide and hist_h[-1] > 0:
        prof = SortedCoverageProfile(c16.to(torch.int64).sum(dim=1), L)
    else:
        prof = CoverageProfile(hist_h, L)
    median_cov = prof.median_nonzero()
    out = dict(sequence=None, codes=None, alphas=None, threshold=None, band=None, posterior=None,
               frac_covered=prof.frac_covered(), frac_min_cov=prof.frac_at_least(min_cov), median_cov=median_cov)

    def finish(codes):
        ascii_ = dev.codes_to_iupac_device(codes, L).cpu().numpy()
        if np.count_nonzero(ascii_ == ord("N")) / float(L) > 0.75:            # :503 / :626
            return out
        out["sequence"] = ascii_.tobytes()
        out["codes"] = codes
        return out

    if consensus:                                                              # :482-516
        say("Consensus requested. Skipping all coverage filters!")
        codes = torch.empty((L + 1) // 2, dtype=torch.uint8, device="cuda")
        fn = lib.tracs_consensus_codes_device32 if wide else lib.tracs_consensus_codes_device
        _lib.check(fn(dev._ptr(c16), L, int(np.ceil(min_cov)), dev._ptr(codes), dev._stream()))
        return finish(codes)

    with np.errstate(divide="ignore", invalid="ignore"):
        thr = max(np.float64(min_cov) / np.float64(median_cov), error_threshold)   # :521 (max() keeps a leading nan, like Python's)
    say("Fraction of genome with read coverage: %s" % out["frac_covered"])
    say("Fraction of genome with read coverage >= %s: %s" % (min_cov, out["frac_min_cov"]))
    say("Median non-zero coverage: %s" % median_cov)
    if out["frac_min_cov"] < 0.25:                                             # :531-535
        return out
    alphas = np.zeros(4)
    iters = C.c_int(0)
    dp = C.POINTER(C.c_double)
    _lib.check(lib.tracs_find_dirichlet_priors_device(dev._ptr(counts), L, 4, 1000, 1e-5, 0, float(error_threshold),
                                                      alphas.ctypes.data_as(dp), C.byref(iters), dev._stream()))   # :537-539
    a_sum = np.sum(alphas)
    if thr <= alphas[1] / (median_cov + a_sum):                                # :541-549
        thr = alphas[1] / (median_cov + a_sum) + 0.01
        say("WARNING: Frequency threshold is set too low! It has been increased to: %s" % thr)
    band = None
    use_band = (not keep_cov_outliers) and median_cov > 50 and alphas[1] / a_sum > thr       # :552-556
    if use_band:
        lo = alphas[1] / thr - a_sum                                           # :557
        q25, q50 = prof.quantile_nonzero(0.25), prof.quantile_nonzero(0.5)     # :559
        band = (lo, q25 - 1.5 * (q50 - q25))                                   # :560
    say("Using frequency threshold: %s" % thr)
    if want_posterior:                                                         # :575-577, written to the .csv.gz before the coverage rules
        out["posterior"] = dev.calculate_posteriors_device(counts, alphas, keep_all, thr).cpu().numpy()
    apply_band = band if (use_band and band[1] > band[0]) else None            # :599-611
    codes = dev.posterior_codes_device(c16, alphas, keep_all, thr, min_cov=int(np.ceil(min_cov)), cov_band=apply_band)   # int16 or int32 counts
    out.update(alphas=alphas, threshold=float(thr), band=band)
    return finish(codes)


def align_post(pileup_path, reference_fasta, output_dir, prefix, ref, min_cov=5, error_threshold=0.01,
               require_both_strands=False, consensus=False, keep_all=False, keep_cov_outliers=False, log=None):
    """One (sample, reference) of tracs/align.py:444-647: writes <prefix>_posterior_counts_ref_<ref>.csv.gz (posterior mode)
    and <prefix>_posterior_counts_ref_<ref>.fasta; returns the call_sequence() dict plus the FASTA path (None if skipped)."""
    output_dir = os.path.join(output_dir, "")
    contigs = read_contigs(reference_fasta)
    counts = pileup_counts(pileup_path, contigs, require_both_strands)
    res = call_sequence(counts, min_cov, error_threshold, consensus, keep_all, keep_cov_outliers, True, log)
    stem = output_dir + prefix + "_posterior_counts_ref_" + str(ref)
    if res["posterior"] is not None:
        write_posterior_csv(stem + ".csv.gz", res["posterior"])
    res["fasta"] = None
    if res["sequence"] is not None:
        with open(stem + ".fasta", "wb") as f:                                  # :636-647
            f.write(b">" + (prefix + "_" + str(ref)).encode() + b"\n" + res["sequence"] + b"\n")
        res["fasta"] = stem + ".fasta"
    return res
