"""End to end through the command line at config-3 scale (parse / pack / H2D / CSV "reported separately"):
`python -m tracs_amd distance --msa X.fasta --meta dates.csv -o out.csv` on a synthetic alignment written to local disk, with the
stage times the library and the driver print under TRACS_STAGE_TRACE=1.

usage: python scripts/bench_e2e.py <samples> <sites> [-D snp_threshold] [--dir /tmp]
Prints one JSON object: sizes, wall time of the command, the stage table, rates (FASTA GB/s, pairs/s, CSV rows/s)."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("samples", type=int)
ap.add_argument("sites", type=int)
ap.add_argument("-D", dest="thr", type=int, default=None)
ap.add_argument("--filter", action="store_true", help="tracs distance --filter (the recombination filter on every emitted pair)")
ap.add_argument("--dir", default=None)
ap.add_argument("--keep", action="store_true")
args = ap.parse_args()
n, L = args.samples, args.sites
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="tracs_e2e_", dir=args.dir)
fa = os.path.join(tmp, "bench_combined.fasta")
names = ["sample_%05d" % i for i in range(n)]
t0 = time.perf_counter()
with open(fa, "wb") as fh:                                     # SURVEY 8d's workload, generated on the device, one line per record
    def emit(rows, first):
        host = rows.cpu().numpy()
        for b in range(host.shape[0]):
            fh.write(b">" + names[first + b].encode() + b"\n")
            fh.write(host[b].tobytes())
            fh.write(b"\n")
    synth.generate_device(n, L, 20241022 + 2, emit, mu_lineage=0.0, mu_sample=1e-4, n_lineages=1, p_n=0.01)
t_write = time.perf_counter() - t0
iso, _ = synth.dates(n, seed=20241022 + 2)
meta = os.path.join(tmp, "dates.csv")
with open(meta, "w") as f:
    f.write("sample,date\n")
    for a, b in zip(names, iso):
        f.write("%s,%s\n" % (a, b))
torch.cuda.empty_cache()
csv = os.path.join(tmp, "out.csv")
cmd = [sys.executable, "-m", "tracs_amd", "distance", "--msa", fa, "--meta", meta, "-o", csv, "--loglevel", "ERROR"]
if args.thr is not None:
    cmd += ["-D", str(args.thr)]
if args.filter:
    cmd += ["--filter"]
t0 = time.perf_counter()
rc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, env=dict(os.environ, TRACS_STAGE_TRACE="1"))
wall_first = time.perf_counter() - t0                           # the first run on a fresh box pages the libraries in
assert rc.returncode == 0, rc.stderr[-3000:]
t0 = time.perf_counter()
rc = subprocess.run(cmd, cwd=root, capture_output=True, text=True, env=dict(os.environ, TRACS_STAGE_TRACE="1"))
wall = time.perf_counter() - t0
assert rc.returncode == 0, rc.stderr[-3000:]
stages = []
for ln in rc.stderr.splitlines():
    m = re.match(r"\[stage\] (.*?) ([0-9.]+) s(?: \(([0-9.]+) GB/s\))?$", ln)
    if m:
        stages.append({"stage": m.group(1), "seconds": float(m.group(2)), **({"GBps": float(m.group(3))} if m.group(3) else {})})
rows = 0
with open(csv, "rb") as fh:
    for blk in iter(lambda: fh.read(1 << 24), b""):
        rows += blk.count(b"\n")
rows -= 1
pairs = n * (n - 1) // 2
fasta_bytes = os.path.getsize(fa)
out = {"samples": n, "sites": L, "pairs": pairs, "snp_threshold": args.thr, "filter": bool(args.filter), "fasta_GB": fasta_bytes / 1e9, "csv_GB": os.path.getsize(csv) / 1e9,
       "csv_rows": rows, "command_seconds": wall, "command_seconds_first_run_on_the_box": wall_first, "pairs_per_s_end_to_end": pairs / wall, "stages": stages,
       "accounted_seconds": sum(s["seconds"] for s in stages if not s["stage"].startswith(("pairsnp (total", "[sum]"))),
       "fasta_write_seconds_setup": t_write,
       "note": "stage lines: library (tracs_pairsnp: read, allocate, pack, panels, COO, D2H) then driver (pairsnp total, transcluster, CSV); "
               "the rest of command_seconds is interpreter + torch-free library start-up and reading the dates"}
print(json.dumps(out))
if not args.keep:
    for f in (fa, meta, csv):
        os.remove(f)
    os.rmdir(tmp)
