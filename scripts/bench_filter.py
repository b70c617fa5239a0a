"""`tracs distance --filter` at config-3 size: the recombination filter (src/pairsnp.hpp:251-318, called per emitted pair :405-413)
on every pair the dense call emits, packed planes and the distance matrix resident in HBM.

    python scripts/bench_filter.py [--samples 10000 --sites 5000000] [--workload sparse] [--partial 0] [--out profiles/r06/bench_filter.json]

Legs (each in seconds per call over ALL emitted pairs):
  lists_first   first filter call on a freshly packed handle: departure lists + N bitmaps built, threshold table built, pairs filtered
  lists_warm    the same call again (index and table kept on the handle)
  threshold     the same with `-D <--snp-threshold>` (only the pairs within the threshold are emitted)
  scan_sample   the round-1..5 route (every pair's SNP bits re-derived from the planes, tracs_filter_recomb_device) on a bounded
                sample of the emitted pairs, extrapolated by pair count -- "today's figure" before the lists
  oracle_check  first --check samples: GPU filtered distances == oracle (full length)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--sites", type=int, default=5000000)
    ap.add_argument("--workload", default="sparse")
    ap.add_argument("--partial", type=float, default=0.0)
    ap.add_argument("--snp-threshold", type=int, default=100)
    ap.add_argument("--scan-sample", type=int, default=200000)
    ap.add_argument("--check", type=int, default=24)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench as B
    from tracs_amd import device as dev
    from tracs_amd import synth

    n, L = args.samples, args.sites
    seed = 20241022 + 2
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    aln = dev.Alignment(n, L)
    kw = B.synth_kw(args.partial, args.workload)
    synth.pack_synthetic_device(aln, seed=seed, **kw)
    dmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    nmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    dev.pairsnp_dense(aln, dmat, nmat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    aln.mark_packed()
    dev.pairsnp_dense(aln, dmat, nmat)
    torch.cuda.synchronize()
    dense_s = time.perf_counter() - t0
    rows, cols, d, _ = dev.coo_from_dense(dmat, nmat, n)
    pairs = rows.numel()
    out = {"samples": n, "sites": L, "workload": args.workload, "partial": args.partial, "pairs": pairs,
           "mean_d": float(d.to(torch.float64).mean().item()), "max_d": int(d.max().item()), "dense_call_s": dense_s}

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t

    # first call on a freshly packed handle: everything built
    aln.mark_packed()
    filt, t_first = timed(lambda: dev.filter_recomb_pairs(aln, rows, cols, d))
    info = dev.filter_index_info(aln)
    out["index"] = info
    warm = []
    for _ in range(args.reps):
        f2, t = timed(lambda: dev.filter_recomb_pairs(aln, rows, cols, d))
        warm.append(t)
        assert torch.equal(f2, filt)
    # index built, threshold table not (a second handle state: touch + rebuild measures build alone)
    out["lists_first_s"] = t_first
    out["lists_warm_s"] = min(warm)
    out["lists_warm_pairs_per_s"] = pairs / min(warm)
    out["index_build_ms"] = None if info is None else sum(info["build_ms"].values())
    out["filtered_mean"] = float(filt.to(torch.float64).mean().item())
    out["filtered_lt_d"] = int((filt < d).sum().item())
    out["checksum_filt"] = int(filt.to(torch.int64).sum().item())
    # -D threshold
    (r2, c2, d2, _), t_coo = timed(lambda: dev.coo_from_dense(dmat, nmat, n, args.snp_threshold))
    if r2.numel():
        _, t_thr = timed(lambda: dev.filter_recomb_pairs(aln, r2, c2, d2))
    else:
        t_thr = 0.0
    out["threshold"] = {"D": args.snp_threshold, "pairs": int(r2.numel()), "coo_s": t_coo, "filter_s": t_thr}
    # the scan route on a sample (rounds 1-5: tracs_filter_recomb_device)
    m = min(pairs, args.scan_sample)
    if m:
        sel = torch.arange(0, pairs, max(1, pairs // m), device=device)[:m]
        sel, _ = torch.sort(sel)
        rs, cs, ds = rows[sel].contiguous(), cols[sel].contiguous(), d[sel].contiguous()
        (fs, found, _, _), t_scan = timed(lambda: dev.filter_recomb_device(aln, rs, cs, ds))
        assert torch.equal(found, ds) and torch.equal(fs, filt[sel])
        out["scan_sample"] = {"pairs": int(m), "seconds": t_scan, "pairs_per_s": m / t_scan, "all_pairs_s": t_scan * pairs / m,
                              "bytes_per_pair": 8 * (L / 8.0), "achieved_GBps": m * L / t_scan / 1e9}
        # and through the new entry point with the lists switched off (wave per pair, fused extract + test)
        os.environ["TRACS_FILTER_LISTS"] = "0"
        aln.mark_packed()
        m2 = min(m, 20000)
        f3, t_scan2 = timed(lambda: dev.filter_recomb_pairs(aln, rs[:m2], cs[:m2], ds[:m2]))
        del os.environ["TRACS_FILTER_LISTS"]
        aln.mark_packed()
        assert torch.equal(f3, fs[:m2])
        out["scan_fallback_sample"] = {"pairs": int(m2), "seconds": t_scan2, "all_pairs_s": t_scan2 * pairs / m2}
    # oracle
    k = min(n, args.check)
    if k >= 2:
        from oracle import oracle as O
        seqs = synth.first_samples_host(n, L, seed, k, **kw)
        r, c, dd, _ = O.pairsnp_arrays(seqs)
        t0 = time.perf_counter()
        ef = O.filter_recomb_pairs(seqs, r, c, os.cpu_count() or 1)
        t_or = time.perf_counter() - t0
        gd = dmat[:k, :k].cpu().numpy()[r.astype(np.int64), c.astype(np.int64)]
        assert np.array_equal(gd, dd.astype(gd.dtype))
        fm = torch.zeros((n, n), dtype=torch.int32, device=device)
        fm[rows.long(), cols.long()] = filt
        gf = fm[:k, :k].cpu().numpy()[r.astype(np.int64), c.astype(np.int64)]
        del fm
        ok = bool(np.array_equal(gf, ef.astype(gf.dtype)))
        out["oracle_check"] = {"samples": k, "pairs": int(len(r)), "equal": ok, "oracle_pairs_per_s": len(r) / t_or,
                               "oracle_threads": os.cpu_count() or 1}
        if not ok:
            print(json.dumps(out, indent=1))
            raise SystemExit("PARITY FAILURE: filtered distances differ from the oracle")
    print(json.dumps(out, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
