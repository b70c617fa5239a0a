"""Throughput of the post-pileup stage and `combine` on one sample of L sites:
pileup text -> counts, coverage profile, alpha fit, posterior, codes, letters, CSV writer, combined-FASTA writer.
Prints one JSON object.  usage: python scripts/bench_align.py [sites] [samples_for_combine]"""
import gzip
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import align_post, combine, synth  # noqa: E402
from tracs_amd import device as dev  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out = {"sites": L}
tmp = tempfile.mkdtemp(prefix="tracs_align_")
counts = synth.allele_counts(L, seed=11, depth=60).astype(np.int64)
refb = np.frombuffer(b"ACGT", np.uint8)[np.random.default_rng(2).integers(0, 4, L)]

# ---- pileup text as htsbox writes it (one line per covered position)
t0 = time.perf_counter()
nz = counts > 0
lines = []
for i in range(L):
    ks = np.flatnonzero(nz[i])
    if not len(ks):
        continue
    c = counts[i, ks]
    f = c // 2
    lines.append("chr1\t%d\t%s\t%s\t%d:%s:%s" % (i + 1, chr(refb[i]), ",".join("ACGT"[k] for k in ks), c.sum(),
                                                ",".join(map(str, f)), ",".join(map(str, c - f))))
text = ("\n".join(lines) + "\n").encode()
del lines
out["pileup_text_bytes"] = len(text)
plain = os.path.join(tmp, "p.txt")
open(plain, "wb").write(text)
gz = os.path.join(tmp, "p.txt.gz")
with gzip.open(gz, "wb", compresslevel=6) as f:
    f.write(text)
out["make_input_s"] = time.perf_counter() - t0
contigs = [("chr1", L)]
for name, path in (("plain", plain), ("gzip", gz)):
    t0 = time.perf_counter()
    parsed = align_post.pileup_counts(path, contigs, True)
    dt = time.perf_counter() - t0
    out["pileup_counts_" + name] = {"s": dt, "text_MBps": len(text) / dt / 1e6, "lines_per_s": text.count(b"\n") / dt}

# ---- device stages, one by one (call_sequence strings the same calls together)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = align_post.call_sequence(parsed)
torch.cuda.synchronize()
out["call_sequence_first_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
res = align_post.call_sequence(parsed)
torch.cuda.synchronize()
out["call_sequence_s"] = time.perf_counter() - t0
out["call_sequence_sites_per_s"] = L / out["call_sequence_s"]
out["alphas"] = [float(a) for a in res["alphas"]]
t0 = time.perf_counter()
res2 = align_post.call_sequence(parsed, want_posterior=False)
torch.cuda.synchronize()
out["call_sequence_no_csv_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
resc = align_post.call_sequence(parsed, consensus=True)
torch.cuda.synchronize()
out["call_sequence_consensus_s"] = time.perf_counter() - t0


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for r in range(reps):
        ev[r].record()
        fn()
    ev[reps].record()
    torch.cuda.synchronize()
    return min(ev[r].elapsed_time(ev[r + 1]) for r in range(reps)) / 1e3


import ctypes as C  # noqa: E402
from tracs_amd import _lib  # noqa: E402
lib = _lib.require_gpu()
cd = torch.from_numpy(parsed).cuda()
hist = torch.empty(align_post.COV_BINS, dtype=torch.int64, device="cuda")
c16 = torch.empty((L, 4), dtype=torch.int16, device="cuda")
bad = torch.empty(1, dtype=torch.int32, device="cuda")
t = timed(lambda: _lib.check(lib.tracs_coverage_profile_device(dev._ptr(cd), L, dev._ptr(hist), align_post.COV_BINS, dev._ptr(c16),
                                                               dev._ptr(bad), dev._stream())))
out["coverage_profile_kernel"] = {"s": t, "alg_bytes_per_site": 40, "GBps": L * 40 / t / 1e9, "frac_of_8TBps": L * 40 / t / 8e12}
codes = torch.empty((L + 1) // 2, dtype=torch.uint8, device="cuda")
t = timed(lambda: _lib.check(lib.tracs_consensus_codes_device(dev._ptr(c16), L, 5, dev._ptr(codes), dev._stream())))
out["consensus_codes_kernel"] = {"s": t, "alg_bytes_per_site": 8.5, "GBps": L * 8.5 / t / 1e9, "frac_of_8TBps": L * 8.5 / t / 8e12}
al = np.zeros(4)
it = C.c_int(0)
t0 = time.perf_counter()
_lib.check(lib.tracs_find_dirichlet_priors_device(dev._ptr(cd), L, 4, 1000, 1e-5, 0, 0.01, al.ctypes.data_as(C.POINTER(C.c_double)),
                                                  C.byref(it), dev._stream()))
torch.cuda.synchronize()
out["find_dirichlet_priors_device"] = {"s": time.perf_counter() - t0, "iterations": it.value}

# ---- writers
post = res["posterior"]
t0 = time.perf_counter()
align_post.write_posterior_csv(os.path.join(tmp, "post.csv.gz"), post)
dt = time.perf_counter() - t0
out["write_posterior_csv"] = {"s": dt, "rows_per_s": L / dt, "text_MBps": L * 32 / dt / 1e6}
t0 = time.perf_counter()
with gzip.open(os.path.join(tmp, "post_np.csv.gz"), "wb") as f:           # the reference's way, on 1/20 of the rows
    np.savetxt(f, post[:L // 20], delimiter=",", newline="\n", fmt="%0.5f")
out["np_savetxt_gzip_rows_per_s"] = (L // 20) / (time.perf_counter() - t0)

alns = []
for s in range(NS):
    d_ = os.path.join(tmp, "s%d" % s)
    os.mkdir(d_)
    p = os.path.join(d_, "s%d_posterior_counts_ref_R.fasta" % s)
    with open(p, "wb") as f:
        f.write(b">s\n" + res["sequence"] + b"\n")
    alns.append(("s%d" % s, p))
t0 = time.perf_counter()
combine.write_alignment("R", alns, tmp + os.sep, n_threads=0)
dt = time.perf_counter() - t0
out["combine_write_alignment"] = {"samples": NS, "s": dt, "sequence_MBps": NS * L / dt / 1e6}
t0 = time.perf_counter()
with gzip.open(os.path.join(tmp, "ref_way.fasta.gz"), "wt") as f:          # the reference's way (serial Python gzip), 4 samples
    for s in range(4):
        f.write(">s%d\n%s\n" % (s, res["sequence"].decode()))
out["python_gzip_sequence_MBps"] = 4 * L / (time.perf_counter() - t0) / 1e6
print(json.dumps(out))
