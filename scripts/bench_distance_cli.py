"""End-to-end `python -m tracs_amd distance` on a synthetic alignment with dates: FASTA(.gz) in, CSV out.
usage: python scripts/bench_distance_cli.py [samples] [sites]"""
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tracs_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="tracs_cli_")
seqs = synth.alignment(n, L, seed=5, mu_lineage=2e-4, mu_sample=2e-5, n_lineages=20, p_n=0.01)
names = ["sample_%05d" % i for i in range(n)]
fa = os.path.join(tmp, "bench_combined.fasta")
synth.write_fasta(fa, seqs, names=names)
iso, _ = synth.dates(n, seed=6)
with open(os.path.join(tmp, "dates.csv"), "w") as f:
    f.write("sample,date\n")
    for a, b in zip(names, iso):
        f.write("%s,%s\n" % (a, b))
out = {"samples": n, "sites": L, "pairs": n * (n - 1) // 2}
for label, extra in (("all_pairs", []), ("snp_threshold_20", ["-D", "20"])):
    csv = os.path.join(tmp, label + ".csv")
    t0 = time.perf_counter()
    rc = subprocess.run([sys.executable, "-m", "tracs_amd", "distance", "--msa", fa, "--meta", os.path.join(tmp, "dates.csv"), "-o", csv,
                         "--loglevel", "ERROR"] + extra, cwd=root, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert rc.returncode == 0, rc.stderr
    rows = sum(1 for _ in open(csv)) - 1
    out[label] = {"s": dt, "rows": rows, "csv_MB": os.path.getsize(csv) / 1e6, "pairs_per_s_end_to_end": out["pairs"] / dt}
for label, argv in (("cluster_snp_10", ["-c", "10", "-D", "snp"]), ("cluster_expectedK_5", ["-c", "5", "-D", "expectedK"])):
    t0 = time.perf_counter()
    rc = subprocess.run([sys.executable, "-m", "tracs_amd", "cluster", "-d", os.path.join(tmp, "all_pairs.csv"), "-o",
                         os.path.join(tmp, label + ".csv"), "--loglevel", "ERROR"] + argv, cwd=root, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert rc.returncode == 0, rc.stderr
    labs = [int(x.split(",")[1]) for x in open(os.path.join(tmp, label + ".csv")).read().split("\n")[1:] if x]
    out[label] = {"s": dt, "csv_rows_per_s": out["all_pairs"]["rows"] / dt, "clusters": max(labs) + 1, "samples": len(labs)}
print(json.dumps(out))
