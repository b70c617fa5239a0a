#!/usr/bin/env python3
"""E(K) stopping-rule classes (tests/ek_parity.py) quantified, CPU only (oracle):
  (1) the golden grid (tests/golden/transcluster_golden.json, E(K) from oracle/_ref = the reference headers compiled with
      setup.py's flags): per class, how far the restated algorithm lands from the reference's value;
  (2) the class histogram of the bench workload's keys (N = SNP distance ~ 2 mu L with mu = 1e-4, L = 5 Mbp;
      delta = |day_i - day_j| over 730 days) at the CLI defaults lambda = 29.903, beta = 73, precision 0.01, and of
      outbreak-scale keys (N <= 80) with the same dates.
Writes profiles/r02/ek_class_histogram.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    out = {}
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "transcluster_golden.json")))
    per = {}
    for grid in g["trans_dist"]:
        N, delta = np.array(grid["N"], np.int32), np.array(grid["delta"])
        _, ek = O.trans_dist(N, delta, grid["lamb"], grid["beta"], grid["thr"])
        for i, cls in enumerate(grid["conditioning"]):
            rel = abs(ek[i] - grid["eK"][i]) / abs(grid["eK"][i])
            per.setdefault(cls, []).append(rel)
    out["golden_grid_oracle_vs_ref"] = {c: {"keys": len(v), "max_rel": float(np.max(v)), "median_rel": float(np.median(v)),
                                            "n_above_1e-6": int(np.sum(np.array(v) > 1e-6))} for c, v in per.items()}
    rng = np.random.default_rng(20241022 + 2)
    lamb, beta, thr = 1e-3 * 29903, 73.0, 0.01
    for name, Ns in (("bench_workload_keys_d~1000", rng.poisson(1000, 400)), ("outbreak_keys_N<=80", rng.integers(0, 81, 400))):
        days = rng.integers(0, 730, size=(400, 2))
        delta = np.abs(days[:, 0] - days[:, 1]) * 86400.0 / 31556952.0
        hist = {}
        for n, d in zip(Ns.tolist(), delta.tolist()):
            cls, _ = O.ek_conditioning(int(n), float(d), lamb, beta, thr)
            hist[cls] = hist.get(cls, 0) + 1
        out[name] = {"lamb": lamb, "beta": beta, "precision": thr, "keys": 400, "classes": hist}
    os.makedirs(os.path.join(ROOT, "profiles", "r02"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02", "ek_class_histogram.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
