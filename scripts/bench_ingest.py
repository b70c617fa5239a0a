"""Ingest row: FASTA(.gz) on disk -> packed planes resident in HBM, MB/s of FASTA text, next to the
reference's own reader (src/kseq.h via oracle/_ref/kseq_dump -q: parse only, no bit packing) on the same files.
usage: python scripts/bench_ingest.py [samples] [sites]"""
import gzip
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import combine  # noqa: E402
from tracs_amd import device as dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
tmp = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(3)
base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
plain, gz = os.path.join(tmp, "ingest.fa"), os.path.join(tmp, "ingest.fa.gz")
parts = []                                               # per-sample files for `combine` (what tracs align leaves behind)
with open(plain, "wb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
    for s in range(n):
        row = base.copy()
        pos = rng.integers(0, L, 50)
        row[pos] = ord("N")
        rec = b">s%d\n" % s + b"\n".join(row[o:o + 80].tobytes() for o in range(0, L, 80)) + b"\n"
        f.write(rec)
        if s < n // 8:
            g.write(rec)
        if s < n // 2:
            pp = os.path.join(tmp, "ingest_part%d_posterior_counts_ref_R.fasta" % s)
            with open(pp, "wb") as h:
                h.write(rec)
            parts.append(("s%d" % s, pp))
out = {"samples": n, "sites": L}
t0 = time.perf_counter()
combine.write_alignment("ingest_R", parts, tmp + os.sep, n_threads=0)
out["combine_s"] = time.perf_counter() - t0
members = os.path.join(tmp, "ingest_R_combined.fasta.gz")
for _, pp in parts:
    os.remove(pp)
for label, path in (("plain", plain), ("gzip", gz), ("gzip_indexed_members", members)):
    mb = os.path.getsize(path) / 1e6
    t0 = time.perf_counter()
    a = dev.Alignment.from_fasta([path])
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    text_mb = a.n * (a.L + a.L // 80 + 8) / 1e6
    out[label] = {"file_MB": mb, "text_MB": text_mb, "s": t, "text_MBps": text_mb / t, "samples": a.n}
    a.close()
    k = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "kseq_dump")
    if os.path.exists(k):
        t0 = time.perf_counter()
        subprocess.run([k, path, "-q"], check=True, capture_output=True)
        tk = time.perf_counter() - t0
        out[label]["reference_kseq_parse_only_MBps"] = text_mb / tk
print(json.dumps(out))
os.remove(plain)
os.remove(gz)
os.remove(members)
