"""`tracs distance --filter` at config-3 size (bench.py's `filter` entry on its own):

    python scripts/bench_filter.py [--samples 10000 --sites 5000000] [--workload sparse] [--partial 0] [--out profiles/r06/bench_filter.json]

See bench.filter_leg for the legs."""
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
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
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
    out = {"samples": n, "sites": L, "workload": args.workload, "partial": args.partial, "dense_call_s": time.perf_counter() - t0}
    out.update(B.filter_leg(n, L, seed, kw, aln, dmat, nmat, dev, synth, torch, device, snp_threshold=args.snp_threshold,
                            scan_sample=args.scan_sample, check=args.check))
    blk = out.pop("_block", None)
    if blk is not None:                                         # (bench.py proper leaves this to its cpu_baseline leg)
        import numpy as np
        from oracle import oracle as O
        seqs = synth.first_samples_host(n, L, seed, blk["samples"], **kw)
        t1 = time.perf_counter()
        ef = O.filter_recomb_pairs(seqs, blk["rows"].astype(np.uint64), blk["cols"].astype(np.uint64), os.cpu_count() or 1)
        ok = bool(np.array_equal(blk["filt"].astype(np.int64), ef.astype(np.int64)))
        out["oracle_check"] = {"samples": blk["samples"], "pairs": int(len(ef)), "equal": ok, "cpu_pairs_per_s": len(ef) / (time.perf_counter() - t1)}
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
