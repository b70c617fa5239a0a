#!/usr/bin/env python3
"""bench.py -- sample-pairs/sec of the all-pairs SNP + transcluster distance path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 10 000 samples x
5 000 000 sites, synthetic, packed planes RESIDENT IN HBM before the timed region.  One step =
one full pass: pairsnp (d and compared sites for all N(N-1)/2 pairs) + transcluster (P(direct),
E(K) for every pair from SNP distance and sampling-date gap).  With N ranks the row panels of the
pair matrix are dealt to the ranks (fold pairing: chunk r and chunk 2N-1-r, equal work), every rank
holds the whole packed alignment, and the per-rank result panels are exchanged with RCCL all-gathers
that overlap the next step's pair kernel (strong scaling: the problem is fixed, `value` = total pairs / time;
every step's panels have arrived on every rank before the clock stops).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel from HIP events on the launch stream:
pairsnp_mfma_kernel (consensus alignments: exact fp4 Gram products on the matrix cores, bound "mfma", with the HBM
view in roofline.hbm) or pairsnp_tile_kernel (general IUPAC alignments: integer VALU, reported against HBM); `cpu_baseline` is the oracle (CPU port of the reference algorithm) timed
on the host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md "Chip-level parameters")
VALU_PEAK = 256 * 4 * 32 * 2.4e9   # 32-bit lane-ops/s: 256 CU x 4 SIMD32 x 2.4 GHz (nominal FP32-vector issue rate)
MFMA_FP4_PEAK = 10.0e15      # flop/s, dense fp4 MFMA (MI355X_MICROARCH.md: "~10 PF dense")
MFMA_FP4_MEASURED = 7.1e15   # bare v_mfma_scale_f32_32x32x64_f8f6f4 loop on +-1 operands (clock-limited; profiles/r01/mfma_fp4_rate.txt)
# Per encoding: VALU ops per 32 sites and pair, algorithmic bytes per pair as a fraction of L, and the
# rate a register-only loop of exactly that instruction mix sustains on MI355X (scripts/micro/valu_ops.hip,
# profiles/r01/valu_ops_microbench.txt) -- the practical issue ceiling of the kernel.
ENCODINGS = {"general": {"ops": 7, "bytes_per_site": 1.0, "mix_ceiling": 44.3e12},
             "consensus": {"ops": 6, "bytes_per_site": 0.75, "mix_ceiling": 49.7e12}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=int(os.environ.get("TRACS_BENCH_SAMPLES", 10000)))
    ap.add_argument("--sites", type=int, default=int(os.environ.get("TRACS_BENCH_SITES", 5000000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--lamb", type=float, default=1e-3 * 29903)     # tracs distance defaults (distance.py:76-90)
    ap.add_argument("--beta", type=float, default=73.0)
    ap.add_argument("--precision", type=float, default=0.01)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from tracs_amd import device as dev
    from tracs_amd import partition, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (several ranks on one GPU only happens in the gloo smoke test)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("TRACS_BENCH_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n, L = args.samples, args.sites
    seed = 20241022 + 2
    # ---- setup (untimed): packed alignment resident in HBM, sampling days -------------------
    t0 = time.time()
    aln = dev.Alignment(n, L)
    p_partial = float(os.environ.get("TRACS_BENCH_PARTIAL", "0"))
    synth.pack_synthetic_device(aln, seed=seed, mu_lineage=1e-5, mu_sample=1e-6, p_n=0.01, p_partial=p_partial)
    _, days_np = synth.dates(n, seed=seed)
    days = torch.from_numpy(days_np).to(device)
    setup_s = time.time() - t0

    cs, nchunk = partition.row_chunks(n, world)
    rows_pad = cs * nchunk
    ranges = partition.rank_ranges(n, rank, world)        # this rank's row panels (one launch each)
    # Two sets of result matrices when the panels travel: step s writes set s % 2 and its all-gathers are only waited for
    # before that set is written again (and at the end of the timed region), so the exchange of one step overlaps the pair
    # kernel of the next -- the way consecutive batches run in production.  One set on a single GPU.
    nsets = 2 if world > 1 else 1
    sets = [(torch.zeros((rows_pad, n), dtype=torch.int32, device=device), torch.zeros((rows_pad, n), dtype=torch.int32, device=device),
             torch.zeros((rows_pad, n), dtype=torch.float64, device=device), torch.zeros((rows_pad, n), dtype=torch.float64, device=device))
            for _ in range(nsets)]
    pending = [[] for _ in range(nsets)]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + args.warmup)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + args.warmup)]

    def step(it):
        k = it % nsets
        dmat, nmat, pmat, emat = sets[k]
        for w in pending[k]:                                  # this set's previous exchange must be over before it is rewritten
            w.wait()
        # pairsnp: the dominant kernel, bracketed by HIP events on the launch stream
        ev0[it].record()
        for r0, r1 in ranges:
            dev.pairsnp_dense(aln, dmat, nmat, row_begin=r0, row_end=r1)
        ev1[it].record()
        # the SNP panels travel (RCCL, own stream) while transcluster runs on this rank's panels
        works = partition.gather_panels((dmat, nmat), n, rank, world, dist, async_op=True)
        dev.trans_dist_dense_ranges(dmat, n, days, args.lamb, args.beta, args.precision, pmat, emat, ranges, exp_p0=True)
        works += partition.gather_panels((pmat, emat), n, rank, world, dist, async_op=True)
        pending[k] = works

    def drain():
        for k in range(nsets):
            for w in pending[k]:
                w.wait()
            pending[k] = []

    for it in range(args.warmup):
        step(it)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        step(it)
    drain()                                                   # every step's panels have arrived on every rank
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pairs_total = n * (n - 1) // 2
    my_pairs = sum(partition.pairs_in_rows(n, r0, r1) for r0, r1 in ranges)
    kern_ms = [ev0[i].elapsed_time(ev1[i]) for i in range(args.warmup, args.warmup + args.steps)]
    kern_s = sum(kern_ms) / len(kern_ms) / 1e3 / len(ranges)      # average duration of ONE launch
    my_pairs_per_launch = my_pairs / len(ranges)

    # sanity: spot-check a few cells against first principles is done in tests; here only a checksum
    dmat, nmat, pmat, emat = sets[(args.warmup + args.steps - 1) % nsets]        # the last step's results
    checksum = int(dmat[:n].sum().item()) if rank == 0 else 0
    if os.environ.get("TRACS_BENCH_VERIFY") and rank == 0:
        # the gathered matrices must equal a single-pass recomputation on this rank
        d1, n1 = torch.zeros_like(dmat), torch.zeros_like(nmat)
        p1, e1 = torch.zeros_like(pmat), torch.zeros_like(emat)
        dev.pairsnp_dense(aln, d1, n1)
        dev.trans_dist_dense_ranges(d1, n, days, args.lamb, args.beta, args.precision, p1, e1, [(0, n)], exp_p0=True)
        ok = bool(torch.equal(d1, dmat) and torch.equal(n1, nmat) and torch.equal(p1, pmat) and torch.equal(e1, emat))
        print("VERIFY gathered == single-pass:", ok, file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit("VERIFY FAILED")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = pairs_total * args.steps / elapsed
        enc = aln.encoding or "general"
        E = ENCODINGS[enc]
        alg_bytes = float(my_pairs_per_launch) * L * E["bytes_per_site"]   # L (general) / 0.75 L (consensus) per pair
        lane_ops = float(my_pairs_per_launch) * ((L + 127) // 128) * 4 * E["ops"]
        hbm = {"achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_bytes / kern_s / HBM_PEAK,
               "algorithmic_bytes_per_pair": L * E["bytes_per_site"]}
        traffic = _traffic_from_profiles(n, L, world, aln.kernel)
        if aln.kernel in ("mfma", "mfma-general"):
            # matrix-core kernel: every site is four fp4 operand values (x, y, z, v) per sample -> 4 MACs = 8 flop per pair and site
            flop = float(my_pairs_per_launch) * L * (8.0 if aln.kernel == "mfma" else 10.0)
            roof = {"bound": "mfma", "achieved": flop / kern_s / 1e12, "peak": MFMA_FP4_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": flop / kern_s / MFMA_FP4_PEAK, "traffic": traffic, "kernel": "pairsnp_mfma_kernel",
                    "kernel_ms": kern_s * 1e3, "encoding": enc, "algorithmic_flop_per_pair": L * 8.0,
                    "measured_fp4_ceiling": MFMA_FP4_MEASURED / 1e12,
                    "frac_of_measured_fp4_ceiling": flop / kern_s / MFMA_FP4_MEASURED,
                    "note": "v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 operands; peak = dense fp4 (MI355X_MICROARCH.md); the measured "
                            "ceiling is the bare instruction rate with this kernel's +-1 operand data (scripts/micro/mfma_fp4_rate.hip)",
                    "hbm": hbm}
        else:
            roof = dict(hbm, bound="hbm", traffic=traffic, kernel="pairsnp_tile_kernel", kernel_ms=kern_s * 1e3, encoding=enc,
                        note="algorithmic bytes are re-used from LDS/L2 tiles, so achieved > HBM peak is expected; "
                             "the binding limit is integer VALU (see valu)",
                        valu={"achieved": lane_ops / kern_s / 1e12, "peak": VALU_PEAK / 1e12, "unit": "Tlane-op/s",
                              "frac": lane_ops / kern_s / VALU_PEAK, "ops_per_32_sites_per_pair": E["ops"],
                              "measured_mix_ceiling": E["mix_ceiling"] / 1e12,
                              "frac_of_measured_mix_ceiling": lane_ops / kern_s / E["mix_ceiling"]})
        out = {"metric": "sample-pairs/sec for 10kx5Mbp SNP+transcluster distance", "value": value,
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "config": {"workload": "%d samples x %d sites, pairsnp (d + compared sites) + transcluster (P, E(K)), "
                                      "all %d pairs" % (n, L, pairs_total),
                          "samples": n, "sites": L, "pairs": pairs_total, "clock_rate": args.lamb,
                          "trans_rate": args.beta, "precision": args.precision,
                          "partition": "row panels, fold pairing, %d rank(s); RCCL all-gather of result panels, overlapped with the next step" % world,
                          "setup_seconds": round(setup_s, 1), "checksum_d": checksum},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, L, seed, days_np, args, dmat, nmat)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _traffic_from_profiles(n, L, world, kernel):
    """HBM bytes per launch of the kernel that ran, from the committed PMC summary (profiles/pmc_summary.json), if one matches."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as fh:
            d = json.load(fh)
        e = d.get("%dx%d@%d" % (n, L, world), {})
        if ("mfma" in e.get("kernel", "")) != (kernel == "mfma") or kernel == "mfma-general":
            return None
        return e.get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(n, L, seed, days_np, args, dmat, nmat):
    """The oracle (C/OpenMP port of the reference algorithm) on the host cores, on a bounded sample of
    the same workload: the first m samples of the SAME synthetic alignment, all m(m-1)/2 pairs at full
    length L through the pair loop (both passes, as the reference runs them; planes already packed, like
    the GPU's timed region), then trans_dist on those pairs (serial and memoised per (N, delta) key, as
    in the reference).  m is sized so the whole leg is ~cpu-seconds."""
    import numpy as np
    from oracle import oracle as O
    from tracs_amd import synth
    cores = O.lib().orc_num_threads()
    # trans_dist costs ~ms per DISTINCT key and dominates small samples: bound the pair count first
    m = int(max(16, min(n, 96)))
    seqs = synth.first_samples_host(n, L, seed, m, mu_lineage=1e-5, mu_sample=1e-6, p_n=0.01,
                                    p_partial=float(os.environ.get("TRACS_BENCH_PARTIAL", "0")))
    planes = O.pack(seqs)                                   # untimed, like the GPU side's resident planes
    t0 = time.perf_counter()
    r, c, d, nn = O.pairsnp_planes(planes, L, dist=2147483647, n_threads=cores)
    t_snp = time.perf_counter() - t0
    delta = np.abs(days_np[r.astype(np.int64)] - days_np[c.astype(np.int64)]).astype(np.float64) * 86400.0 / 31556952.0
    t1 = time.perf_counter()
    O.trans_dist(d.astype(np.int32), delta, args.lamb, args.beta, args.precision)
    t_tc = time.perf_counter() - t1
    pairs = m * (m - 1) // 2
    nkeys = len(set(zip(d.tolist(), delta.tolist())))
    # the sample doubles as a full-size parity check: the GPU's d / nn for these pairs must be bit-equal
    ri, ci = r.astype(np.int64), c.astype(np.int64)
    gd = dmat[:m, :m].cpu().numpy().astype(np.int64)[ri, ci]
    gn = nmat[:m, :m].cpu().numpy().astype(np.int64)[ri, ci]
    if not (np.array_equal(gd, d.astype(np.int64)) and np.array_equal(gn, nn.astype(np.int64))):
        raise SystemExit("PARITY FAILURE: GPU d/nn differ from the oracle on the %d x %d sample block" % (m, m))
    return {"value": pairs / (t_snp + t_tc), "unit": "pairs/s", "cores": cores, "kind": "port",
            "pairsnp_pairs_per_s": pairs / t_snp, "trans_dist_keys_per_s": nkeys / t_tc,
            "sample": "first %d samples x %d sites of the same alignment = %d pairs: oracle pair loop, two passes, "
                      "%d OpenMP threads (%.2f s) + serial memoised trans_dist over %d distinct (N, delta) keys "
                      "(%.2f s); GPU d/nn bit-equal on this block" % (m, L, pairs, cores, t_snp, nkeys, t_tc)}


if __name__ == "__main__":
    main()
