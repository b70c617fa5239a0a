"""E(K) of the outbreak-scale golden grid (tests/golden/transcluster_outbreak_golden.json: 4 000 keys, N <= 80, 1..730 days, CLI
defaults) on the GPU against the reference as shipped (oracle/_ref, -ffast-math) and against the same source compiled IEEE-strict.
Prints one JSON object: per conditioning class, how many keys agree with the reference to 1e-6, the largest deviation, and on how
many of the others the reference's two builds already disagree with each other."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tracs_amd import api  # noqa: E402

g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "transcluster_outbreak_golden.json")))
N = np.array(g["N"], np.int32)
delta = np.array(g["days"], np.float64) * 86400.0 / 31556952.0
p0, ek = api.trans_dist_arrays(N, delta, g["lamb"], g["beta"], g["thr"])
ref = np.array([np.nan if v is None else v for v in g["eK"]])
strict = np.array([np.nan if v is None else v for v in g["eK_strict_build"]])
cls = np.array(g["conditioning"])
out = {"keys": len(N), "p0_max_rel_vs_ref": float(np.max(np.abs(p0 - np.array(g["p0"])) / np.abs(np.array(g["p0"])))), "classes": {}}
for c in ("well", "ill", "saturated"):
    m = cls == c
    fin = m & np.isfinite(ref)
    rel = np.abs(ek[fin] - ref[fin]) / np.abs(ref[fin])
    relb = np.abs(strict[fin] - ref[fin]) / np.abs(ref[fin])          # the reference's two builds against each other
    relb[~np.isfinite(relb)] = np.inf                                  # (the strict build returns nan where the shipped one returns a number)
    off = rel > 1e-6
    agree = relb <= 1e-6
    out["classes"][c] = {"keys": int(m.sum()), "fraction_of_grid": float(m.mean()), "reference_not_finite": int((m & ~np.isfinite(ref)).sum()),
                         "within_1e-6_of_reference": int((~off).sum()), "max_rel_vs_reference": float(rel.max()) if rel.size else None,
                         "median_rel_of_the_others": float(np.median(rel[off])) if off.any() else None,
                         "others": int(off.sum()),
                         "others_where_the_two_reference_builds_disagree": int((off & (relb > 1e-6)).sum()),
                         "reference_builds_disagree": int((relb > 1e-6).sum()),
                         "max_rel_between_reference_builds": float(np.max(relb[np.isfinite(relb)])) if np.isfinite(relb).any() else None,
                         "strict_build_not_finite": int((m & ~np.isfinite(strict)).sum()),
                         "keys_where_both_builds_are_finite_and_agree": int(agree.sum()),
                         "of_those_within_1e-6": int((agree & ~off).sum()),
                         "of_those_max_rel": float(rel[agree].max()) if agree.any() else None,
                         "within_1e-6_of_strict_build": int((np.abs(ek[fin] - strict[fin]) <= 1e-6 * np.abs(strict[fin])).sum())}
print(json.dumps(out))
