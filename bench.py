#!/usr/bin/env python3
"""bench.py -- sample-pairs/sec of the all-pairs SNP + transcluster distance path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 10 000 samples x
5 000 000 sites, synthetic -- one ancestor of iid uniform bases, every sample a copy with a different base at
Bernoulli(mu = 1e-4) sites (E[d] ~ 2 mu L ~ 1 000 before masking) and 'N' at Bernoulli(0.01) sites: a consensus (ACGTN)
alignment -- packed planes RESIDENT IN HBM before the timed region.  One step = one full pass: pairsnp (d and compared
sites for all N(N-1)/2 pairs) + transcluster (P(direct), E(K) for every pair from SNP distance and sampling-date gap).
`--gpus N` with N > 1 and no launcher around it starts its own N ranks (`python -m torch.distributed.run --nproc-per-node N bench.py
...` as a child process, before anything here has touched a GPU) and relays their line; under a launcher WORLD_SIZE must equal N.
With N ranks (one per GPU) the default partition is by SITES (--partition sites): rank r holds a contiguous 1 / N of the packed
planes (whole 128-site groups) and runs the single-GPU call on it for ALL pairs -- classification, lists, walks: every stage of a
call works on 1 / N of the data --; d and the compared-sites counts are sums over sites, so the partial matrices are summed by the
compact exchange (tracs_amd/partition.py TriExchange, csrc/exchange.hip: the upper-triangle cells only, 16 bits per cell where a
slice's values fit, RCCL all-to-all over every xGMI link at once) and rank q ends up with the rows it owns (fold pairing) of d, nn,
P and E(K): nothing is gathered, transcluster runs on a rank's own rows.  `--partition pairs` is north_star's wording: every rank
holds the whole alignment and computes its row panels (fold pairing), RCCL all-gather of the d / nn panels (16 bits per cell where
the values allow), P and E(K) re-derived from the gathered d on every rank.  Strong scaling either way: the problem is fixed,
`value` = total pairs / time (max over ranks, barrier + synchronize on both sides).

Prints ONE JSON line (rank 0):
  roofline          the dominant kernel of the timed region from HIP events on the launch stream (recorded by the library around
                    each part of a dense call).  With site classes in use (the default on this workload, csrc/site_classes.hip) a call
                    is: the four-operand pair kernel over the dense sites (matrix cores), the minority lists' kernel, the one-operand
                    counting pass over the sites with many N samples (matrix cores) and the N co-occurrence walk over the lists of the
                    sites with few (nn_rows_kernel, memory-bound: `bound` "hbm") -- whichever takes longest is reported, the others
                    beside it (`other_kernels`, `kernels_ms`), with the class sizes (`site_classes`).  With TRACS_SITE_CLASSES=0 it is
                    the pair kernel over every site
  roofline_general  the same pass over the SAME alignment with 0.5 % partial IUPAC codes added (SURVEY.md 8d's C4 mix):
                    one-hot matrix-core kernel + sparse partial-code correction, with the once-per-call stages and their bytes
                    (N = 1 only; not part of `value`; the same alignment is the `partial` leg of `sensitivity`)
  dm_frontend       counts -> posterior filter -> 4-bit codes -> packed planes for a batch of samples (SURVEY.md 8d C3's
                    "counts->posterior->code fused front-end"), timed separately (N = 1 only)
  cpu_baseline      the oracle (CPU port of the reference algorithm) on the host cores on a bounded sample of the same
                    workload, pair loop and trans_dist legs reported separately, the trans_dist leg also through the reference's
                    own source compiled in place (oracle/_ref) (rank 0, every N); its sample block doubles as the ALWAYS-ON
                    result check: the GPU's d / nn of those pairs -- at N > 1: rows rank 0 owns after the exchange -- must be bit-equal
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
# bare v_mfma_scale_f32_32x32x64_f8f6f4 loop (clock-limited; scripts/micro/mfma_fp4_rate.hip, profiles/r01/mfma_fp4_rate.txt): the
# rate depends on the operand data -- 7.1 PFLOP/s on +-1 operands (the four-operand pair kernel's), 7.9 on mostly-zero operands
# (the counting pass's N plane: 99 % zeros on this workload)
MFMA_FP4_MEASURED = 7.1e15
MFMA_FP4_MEASURED_ZEROS = 7.9e15
# Per encoding: VALU ops per 32 sites and pair, algorithmic bytes per pair as a fraction of L, and the
# rate a register-only loop of exactly that instruction mix sustains on MI355X (scripts/micro/valu_ops.hip,
# profiles/r01/valu_ops_microbench.txt) -- the practical issue ceiling of the VALU kernel.
ENCODINGS = {"general": {"ops": 7, "bytes_per_site": 1.0, "mix_ceiling": 44.3e12},
             "consensus": {"ops": 6, "bytes_per_site": 0.75, "mix_ceiling": 49.7e12}}
# matrix-core forms: fp4 operand values per site and sample (x, y, z, v / one-hot A, C, G, T + N) -> flop per pair and site
MFMA_FLOP_PER_SITE = {"mfma": 8.0, "mfma-general": 10.0}
COUNT_FLOP_PER_SITE = 2.0    # the counting pass over invariant sites (site classes): one operand plane
MU = 1e-4                    # per-sample substitution probability
P_N = 0.01
P_PARTIAL_C4 = 0.005         # SURVEY.md 8d, config 4 mix: partial-ambiguity IUPAC codes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=int(os.environ.get("TRACS_BENCH_SAMPLES", 10000)))
    ap.add_argument("--sites", type=int, default=int(os.environ.get("TRACS_BENCH_SITES", 5000000)))
    ap.add_argument("--partial", type=float, default=float(os.environ.get("TRACS_BENCH_PARTIAL", "0")),
                    help="fraction of partial IUPAC codes in the TIMED alignment (default 0: consensus, the metric's workload)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="sparse",
                    help="synthetic alignment of the TIMED steps (default: sparse = SURVEY 8d's, the metric's workload)")
    ap.add_argument("--partition", choices=["sites", "pairs"], default=os.environ.get("TRACS_BENCH_PARTITION", "sites"),
                    help="N > 1: `sites` -- every rank holds a slice of the SITES and counts all pairs over it, the sums arrive as row panels "
                         "(reduce-scatter): every stage of a call shrinks with N; `pairs` -- every rank holds the whole alignment and computes "
                         "its row panels (all-gather): what is built once per pack is repeated on every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_general and dm_frontend")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="minimum wall time of each CPU baseline leg")
    ap.add_argument("--lamb", type=float, default=1e-3 * 29903)     # tracs distance defaults (distance.py:76-90)
    ap.add_argument("--beta", type=float, default=73.0)
    ap.add_argument("--precision", type=float, default=0.01)
    ap.add_argument("--launch-check", action="store_true",
                    help="only check the launch: the ranks rendezvous (gloo, no GPU touched), rank 0 prints {\"n_gpus\": ranks that answered}")
    return ap.parse_args()


def in_launcher():
    """True when a launcher (torch.distributed.run: ours -- launch_ranks marks its children -- or the driver's) started this process as
    one rank of a job: RANK / WORLD_SIZE alone (a SLURM step's ambient variables) do not count, the launcher's rendezvous does."""
    if os.environ.get("TRACS_BENCH_WORKER") == "1":
        return True
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"))


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD process -- never a re-exec, and before this
    process has touched a GPU -- with the same arguments; their stdout (rank 0's JSON line) is this process's, their exit code too."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), TRACS_BENCH_WORKER="1")
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, world, rank):
    """--launch-check: the ranks find each other (gloo; no GPU is touched) and rank 0 reports how many answered."""
    import torch
    import torch.distributed as dist
    n_seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        n_seen = int(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": n_seen, "world_size_env": world, "gpus_arg": args.gpus}), flush=True)
    return 0 if n_seen == args.gpus else 1


# The metric's workload ("sparse": SURVEY.md 8d's generator -- star phylogeny, mu per sample, 1 % N iid) and four others that
# move the alignment across the site classes' cost model (--workload; `sensitivity` on the default line):
WORKLOADS = {
    "sparse": dict(mu_lineage=0.0, mu_sample=MU, n_lineages=1, p_n=P_N),
    # 20 lineages (founders 1e-4 from the ancestor, samples 1e-5 from their founder), the same amount of N in 1 / 21 of the samples
    "lineage": dict(mu_lineage=1e-4, mu_sample=1e-5, n_lineages=20, p_n=P_N, n_every=21),
    "divergent": dict(mu_lineage=0.0, mu_sample=1e-3, n_lineages=1, p_n=P_N),
    "clean": dict(mu_lineage=0.0, mu_sample=MU, n_lineages=1, p_n=0.0),
    "gappy": dict(mu_lineage=0.0, mu_sample=MU, n_lineages=1, p_n=0.10),
    # coverage gaps as real consensus alignments have them: N in runs of consecutive sites (geometric lengths, mean 500), each lost by
    # a random 0.5-30 % of the samples, 1 % N overall (tracs_amd/synth.py: coverage_runs) -- the reference's cost does not care
    # (src/pairsnp.hpp:417-420), the site classes' does
    "runs": dict(mu_lineage=0.0, mu_sample=MU, n_lineages=1, p_n=0.0, runs=dict(p_n=P_N, mean_len=500, frac_lo=0.005, frac_hi=0.30)),
    # what `tracs align` itself writes (tracs/align.py:599-622): every sample N wherever ITS coverage is below the thresholds -- here
    # 30 % of the sites, in runs of geometric length (mean 5 kb) whose boundaries are the sample's own --, partial IUPAC codes where
    # two alleles pass the posterior filter (0.5 % of the sites; the default, without --consensus), two lineages
    "coverage": dict(mu_lineage=1e-4, mu_sample=MU, n_lineages=2, p_n=0.0, gaps=dict(frac=0.30, mean_len=5000), p_partial=P_PARTIAL_C4),
}


def synth_kw(p_partial=0.0, workload="sparse"):
    kw = dict(WORKLOADS[workload])
    if p_partial or "p_partial" not in kw:
        kw["p_partial"] = p_partial
    return kw


def pair_split_ms(lib, calls=1):
    """(pair kernel, sparse partial-code correction + minority lists, counting pass on the matrix cores, N co-occurrence lists):
    mean over the last `calls` dense calls (the library keeps the events of the last 64), from HIP events the library records on
    the launch stream around each part (tracs_debug_pair_timing) -- over the timed steps, what rocprofv3's average reports."""
    import ctypes as C
    out = (C.c_float * 4)()
    return [float(x) for x in out] if lib.tracs_debug_pair_ms_mean(max(1, int(calls)), out) == 0 else None


def hbm_physical(traffic, kern_s, compulsory):
    """What the kernel really moves: `traffic` (PMC FETCH_SIZE / WRITE_SIZE bytes per launch, gfx950-corrected) over its launch
    time against the 8 TB/s peak, and over the bytes it has to move at least once.  SURVEY 8d's per-pair byte count assumes no
    tile reuse and reads >> 1 of the HBM peak by construction on an LDS-tiled kernel, so it is not reported as a fraction."""
    if traffic is None:
        return None
    return {"GBps": traffic / kern_s / 1e9, "frac_of_hbm_peak": traffic / kern_s / HBM_PEAK, "peak_GBps": HBM_PEAK / 1e9,
            "compulsory_bytes": compulsory, "traffic_over_compulsory": traffic / compulsory if compulsory else None}


def per_pack_roofline(stages):
    """The once-per-pack stages of one call (HIP events on the launch stream, recorded by the library): milliseconds, the bytes
    the stage reads and writes by the library's own accounting of its arrays (algorithmic: each array once per pass over it),
    and that over the stage's time against the 8 TB/s HBM peak."""
    if not stages:
        return None
    rows, tot_ms, tot_b = [], 0.0, 0.0
    for name, ms, rd, wr in stages:
        b = rd + wr
        rows.append({"stage": name, "ms": round(ms, 3), "read_GB": round(rd / 1e9, 3), "written_GB": round(wr / 1e9, 3),
                     "GBps": round(b / max(ms, 1e-6) / 1e6, 1), "frac": round(b / max(ms, 1e-6) / 1e-3 / HBM_PEAK, 3) if b else None})
        tot_ms += ms
        tot_b += b
    return {"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s", "stages": rows, "per_pack_ms": round(tot_ms, 3),
            "bytes_GB": round(tot_b / 1e9, 2), "achieved": round(tot_b / max(tot_ms, 1e-6) / 1e6, 1),
            "frac": round(tot_b / max(tot_ms, 1e-6) / 1e-3 / HBM_PEAK, 3)}


def count_roofline(pairs_per_launch, sites, count_s, n, in_place):
    flop = float(pairs_per_launch) * sites * COUNT_FLOP_PER_SITE
    return {"kernel": "pairsnp_mfma_kernel<COUNT>", "kernel_ms": count_s * 1e3, "sites": sites, "bound": "mfma", "traffic": None,
            "source": "the stored N plane of every site, read in place" if in_place else "the counted sites' N plane, re-packed",
            "algorithmic_flop_per_pair": sites * COUNT_FLOP_PER_SITE, "measured_fp4_ceiling": MFMA_FP4_MEASURED_ZEROS / 1e12,
            "frac_of_measured_fp4_ceiling": flop / count_s / MFMA_FP4_MEASURED_ZEROS,
            "achieved": flop / count_s / 1e12, "peak": MFMA_FP4_PEAK / 1e12, "unit": "TFLOP/s", "frac": flop / count_s / MFMA_FP4_PEAK,
            "compulsory_bytes": float(n) * sites / 8.0 + float(pairs_per_launch) * 4.0,      # the plane once + nn once
            "algorithmic_bytes_per_pair_no_reuse": sites * 0.25,
            "note": "nn = sites - c_i - c_j + sum n_i n_j: one fp4 operand plane (n = is N here; mostly zeros: the measured ceiling is "
                    "the bare instruction's rate on zero operands), %g flop per pair and site" % COUNT_FLOP_PER_SITE}


def nn_list_roofline(ls, sites, kern_s, rows, n, L=None):
    """nn_rows_kernel (csrc/site_lists.hip): the N co-occurrences NN = sum n_i n_j of the sites with few N samples, from their N
    lists -- memory-bound.  Algorithmic bytes per launch: every list entry a walk decodes, one byte each (n8 lines: byte deltas;
    cN entries per walk, one walk per N sample and site: the sum of cN^2 over those sites) + the rows' N bitmaps (L / 8 bytes per
    row), scaled by the rows of the launch (a rank's panel walks its rows' lists only).  `frac_needed` counts only the entries
    with j > i (half of them: row i needs no other).  What the kernel physically moves is whole 128-byte lines: one per walk and
    ~115 samples of the list (`lines_bytes`; measured: `traffic`)."""
    frac_rows = rows / float(n)
    bitmap = rows * (L / 8.0) if L else 0.0
    alg = ls["nn_visits"] * ls["n_entry_bytes"] * frac_rows + bitmap
    needed = 0.5 * ls["nn_visits"] * ls["n_entry_bytes"] * frac_rows + bitmap
    lines = ls["nn_walks"] * 128.0 * frac_rows + bitmap
    # `frac` / `achieved` count the NEEDED bytes only -- the list entries with j > i (row i needs no other: half of every list) + the
    # rows' bitmaps --, the strictest of the three counts; `frac_visited` counts every entry a walk decodes, `frac_lines` the whole
    # 128-byte lines the memory system has to move at least (the figure on the line was the builder's own byte count)
    return {"kernel": "nn_rows_kernel", "kernel_ms": kern_s * 1e3, "sites": sites, "bound": "hbm", "traffic": None,
            "achieved": needed / kern_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": needed / kern_s / HBM_PEAK,
            "algorithmic_bytes": needed, "frac_needed": needed / kern_s / HBM_PEAK, "needed_bytes": needed,
            "frac_visited": alg / kern_s / HBM_PEAK, "visited_bytes": alg,
            "lines_bytes": lines, "frac_lines": lines / kern_s / HBM_PEAK,
            "formulation_floor": "9.1 ms at this shape with the decode cut out (the lines alone, 8.8 TB/s out of the Infinity Cache: "
                                 "profiles/r04/nn_rows_n8.txt) -- the kernel runs within 15 % of what one 128-byte line per walk allows; "
                                 "profiles/r05/nn_rows_rejected.txt",
            "list_entries_visited": ls["nn_visits"] * frac_rows, "list_walks": ls["nn_walks"] * frac_rows,
            "bytes_per_list_entry": ls["n_entry_bytes"], "bitmap_bytes": bitmap,
            "entries_per_s": ls["nn_visits"] * frac_rows / kern_s,
            "note": "row i of the pair matrix in LDS; every N site of sample i (a set bit of its N bitmap) is a walk of the site's N list "
                    "-- byte deltas in sample order, 124 per 128-byte line: a site of ~100 N samples among 10 000 is ONE line --, scanned by "
                    "four lanes (byte sums, a DPP prefix) and decoded where it lies (one SDWA add and one ds_add per byte, for j > i): sum "
                    "of cN^2 list entries instead of n^2 / 2 pairs per site on the matrix cores.  Random reads of whole 128-byte lines "
                    "(`frac_lines`, `traffic`, `hbm_physical`); instruction issue is the first wall (DESIGN.md 3.1)"}


def roofline_of(kernel, enc, pairs_per_launch, L, kern_s, traffic, n):
    E = ENCODINGS[enc]
    compulsory = float(n) * L * E["bytes_per_site"] / 2.0 + float(pairs_per_launch) * 8.0     # the planes once + d, nn once
    if kernel in MFMA_FLOP_PER_SITE:
        fps = MFMA_FLOP_PER_SITE[kernel]
        flop = float(pairs_per_launch) * L * fps
        return {"bound": "mfma", "achieved": flop / kern_s / 1e12, "peak": MFMA_FP4_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": flop / kern_s / MFMA_FP4_PEAK, "traffic": traffic,
                "kernel": "pairsnp_mfma_kernel" + ("<general> + general_fixup_kernel" if kernel == "mfma-general" else ""),
                "kernel_ms": kern_s * 1e3, "encoding": enc, "algorithmic_flop_per_pair": L * fps,
                "measured_fp4_ceiling": MFMA_FP4_MEASURED / 1e12, "frac_of_measured_fp4_ceiling": flop / kern_s / MFMA_FP4_MEASURED,
                "compulsory_bytes": compulsory, "algorithmic_bytes_per_pair_no_reuse": L * E["bytes_per_site"],
                "hbm_physical": hbm_physical(traffic, kern_s, compulsory),
                "note": "v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 operands; %g flop per pair and site is the flop count of THIS "
                        "formulation (%s), not an algorithm-intrinsic number; peak = dense fp4 (MI355X_MICROARCH.md); the "
                        "measured ceiling is the bare instruction rate on +-1 operand data (scripts/micro/mfma_fp4_rate.hip)"
                        % (fps, "operand planes x, y, z = x*y, v" if kernel == "mfma" else "one-hot planes A, C, G, T and N")}
    lane_ops = float(pairs_per_launch) * ((L + 127) // 128) * 4 * E["ops"]
    alg_bytes = float(pairs_per_launch) * L * E["bytes_per_site"]
    return {"bound": "hbm", "achieved": alg_bytes / kern_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_bytes / kern_s / HBM_PEAK,
            "traffic": traffic, "kernel": "pairsnp_tile_kernel", "kernel_ms": kern_s * 1e3, "encoding": enc,
            "compulsory_bytes": compulsory, "hbm_physical": hbm_physical(traffic, kern_s, compulsory),
            "valu": {"achieved": lane_ops / kern_s / 1e12, "peak": VALU_PEAK / 1e12, "unit": "Tlane-op/s",
                     "frac": lane_ops / kern_s / VALU_PEAK, "ops_per_32_sites_per_pair": E["ops"],
                     "measured_mix_ceiling": E["mix_ceiling"] / 1e12,
                     "frac_of_measured_mix_ceiling": lane_ops / kern_s / E["mix_ceiling"]}}


def dense_call_roofline(aln, split, n, L, world, last_pairs, last_rows, my_pairs_per_launch, kern_s, split_calls, overlap=False):
    """`roofline` of a dense call: whichever of its kernels takes longest (HIP events the library records around its parts on the
    launch stream: `split`), the others beside it.  last_pairs / last_rows: the cells and rows of the call `split` belongs to (a rank
    of the pair partition: its last panel; a rank of the site partition: all rows over its slice of L sites)."""
    enc = aln.encoding or "general"
    classes = aln.site_classes
    traffic = _traffic_from_profiles(n, L, world, aln.kernel + ("+classes" if classes else ""))
    if classes and split:
        # site classes (csrc/site_classes.hip): pair kernel over the dense sites, lists for the minority sites, one-operand
        # counting pass over the counted sites.  `roofline` = whichever matrix-core kernel takes longer; the other beside it.
        dense, counted, minority, full = classes
        count_sites, in_place, nn_listed = aln.count_source or (counted, False, 0)
        ls = aln.list_stats
        main = roofline_of(aln.kernel, enc, last_pairs, dense, max(split[0], 1e-3) / 1e3, None, n)
        cnt = count_roofline(last_pairs, count_sites, max(split[2], 1e-3) / 1e3, n, in_place)
        nnl = nn_list_roofline(ls, nn_listed, max(split[3], 1e-3) / 1e3, last_rows, n, L) if nn_listed else None
        cands = [(split[0], main, ""), (split[2], cnt, "+classes"), (split[3], nnl, "+nnlists")]
        cands = sorted([c for c in cands if c[1] is not None], key=lambda c: -c[0])
        roof, tag = cands[0][1], cands[0][2]
        traffic = _traffic_from_profiles(n, L, world, aln.kernel + tag) if tag else None
        roof["traffic"] = traffic
        roof["hbm_physical"] = hbm_physical(traffic, roof["kernel_ms"] / 1e3, roof.get("compulsory_bytes") or roof.get("algorithmic_bytes"))
        roof["other_kernels"] = [{k: c[1][k] for k in ("kernel", "kernel_ms", "bound", "achieved", "frac", "unit", "sites") if k in c[1]} for c in cands[1:]]
        roof["kernel_ms_over"] = ("mean over the %d dense calls of the timed steps (HIP events around each part on the launch stream); "
                                  "transcluster runs beside them on a second stream%s" % (split_calls, "" if overlap else " -- not in this run"))
        roof["minority_lists_ms"] = split[1]
        roof["dense_call_ms"] = kern_s * 1e3
        roof["kernels_ms"] = {"pair kernel (dense sites)": split[0], "general_fixup_kernel (minority lists)": split[1],
                              "counting pass (matrix cores)": split[2], "nn_rows_kernel (N co-occurrence lists)": split[3]}
        roof["site_classes"] = {"dense": dense, "counted": counted, "minority": minority, "full": full,
                                "empty": L - dense - counted - full, "counting_pass_sites": count_sites, "counting_pass_in_place": in_place,
                                "nn_list_sites": nn_listed, "lists": ls,
                                "note": "decided once per pack, results bit-identical (csrc/site_classes.hip): the pair kernel reads the dense "
                                        "sites only; sites at which <= a few samples differ from the others (minority) add their distances "
                                        "from sparse lists (general_fixup_kernel<MINOR>); nn of every non-dense site with an N comes from "
                                        "the N lists of the sites with few N samples (nn_rows_kernel) and a one-operand matrix-core pass over the "
                                        "others (counted = both + the sites with a single N), sites without any N add a constant (full). "
                                        "TRACS_SITE_CLASSES=0 reads every site with the pair kernel, TRACS_MINORITY=0 keeps the minority sites dense"}
    else:
        roof = roofline_of(aln.kernel, enc, my_pairs_per_launch, L, kern_s, traffic, n)
    roof["traffic_source"] = ("profiles/pmc_summary.json: FETCH_SIZE / WRITE_SIZE (gfx950-corrected) from separate rocprofv3 --pmc passes of "
                              "`bench.py` itself on this workload and shape (scripts/gpu_pmc_bench.sh), committed with the profiles -- not "
                              "measured by this run" if roof.get("traffic") is not None else None)
    return roof


def main():
    global _WORKLOAD_OF_PROFILES
    args = parse()
    _WORKLOAD_OF_PROFILES = args.workload if args.partial == 0 else "partial"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not in_launcher():
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args))              # N ranks as a child process; nothing here has touched a GPU
        world, rank, local = 1, 0, 0
    else:
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:                                # never a silent run on a different number of GPUs than the line says
            raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.launch_check:
        raise SystemExit(launch_check(args, world, rank))
    import torch
    import torch.distributed as dist
    from tracs_amd import _lib
    from tracs_amd import device as dev
    from tracs_amd import partition, synth

    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (several ranks on one GPU only happens in the gloo tests)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    comm_info = None
    if world > 1:
        # the exchange: the library's own RCCL entry points (include/tracs_hip.h part 4, csrc/comm.cpp) behind tracs_amd.rccl.RcclDist
        # -- torch only launched the processes and lends the rendezvous store --; TRACS_BENCH_BACKEND=nccl: torch.distributed over
        # RCCL; gloo: torch.distributed over gloo (several ranks on one GPU: the tests; the default when GPUs < ranks).
        # NO fallback: a communicator that cannot be made or fails its self-test ends this rank non-zero (the launcher ends the rest).
        from tracs_amd import rccl
        backend = rccl.backend_choice(world, ndev, "TRACS_BENCH_BACKEND")
        if world > ndev and os.environ.get("TRACS_BENCH_BACKEND") != "gloo":
            # more ranks than GPUs only ever makes sense for the gloo smoke tests: never a line that says n_gpus = world by accident
            raise SystemExit("--gpus %d but %d GPU(s) visible (TRACS_BENCH_BACKEND=gloo lets ranks share a GPU: tests)" % (world, ndev))
        if backend == "rccl":
            try:
                cand = rccl.RcclDist(device)
                ok = cand.self_test()
            except Exception as e:                                # noqa: BLE001
                print("bench: rank %d: no RCCL communicator through libtracs_hip (%s)" % (rank, e), file=sys.stderr, flush=True)
                raise SystemExit(3)
            if not ok:
                print("bench: rank %d: the RCCL communicator failed its self-test" % rank, file=sys.stderr, flush=True)
                raise SystemExit(3)
            dist = cand
            seen = cand.ranks_seen()
            comm_info = {"ranks_seen": seen[1], "rank_seen": seen[0], "rccl_version": cand.rccl_version(), "self_test": "passed"}
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
            comm_info = {"ranks_seen": dist.get_world_size(), "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version())}
        else:
            dist.init_process_group(backend)
            comm_info = {"ranks_seen": dist.get_world_size(), "rccl_version": None}
        exchange = {"rccl": "libtracs_hip.so: RCCL behind the C ABI (csrc/comm.cpp)",
                    "nccl": "torch.distributed over RCCL", "gloo": "torch.distributed over gloo"}.get(backend, backend)
        comm_info["gpus_visible"] = ndev
    else:
        exchange = None

    n, L = args.samples, args.sites
    seed = 20241022 + 2
    if world > 1 and args.partition == "sites":
        return site_sharded(args, n, L, seed, world, rank, device, dist, exchange, comm_info)
    # ---- setup (untimed): packed alignment resident in HBM, sampling days -------------------
    t0 = time.time()
    aln = dev.Alignment(n, L)
    synth.pack_synthetic_device(aln, seed=seed, **synth_kw(args.partial, args.workload))
    _, days_np = synth.dates(n, seed=seed)
    days = torch.from_numpy(days_np).to(device)
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    cs, nchunk = partition.row_chunks(n, world)
    rows_pad = cs * nchunk
    ranges = partition.rank_ranges(n, rank, world)        # this rank's row panels (one launch each)
    if world > 1:
        aln.hint_rows(ranges)                             # what is built per row once per pack: for this rank's rows only
    # Two sets of d / nn matrices when the panels travel: step s writes set s % 2 and its all-gathers are only waited for
    # before that set is written again (and at the end of the timed region), so the exchange of one step overlaps the pair
    # kernel of the next -- the way consecutive batches run in production.  One set on a single GPU.  P and E(K) never
    # travel: each rank derives them for the whole matrix from the gathered d (one key table, ~1 ms of gather per matrix).
    nsets = 2 if world > 1 else 1
    sets = [(torch.zeros((rows_pad, n), dtype=torch.int32, device=device), torch.zeros((rows_pad, n), dtype=torch.int32, device=device))
            for _ in range(nsets)]
    pmat = torch.zeros((rows_pad, n), dtype=torch.float64, device=device)
    emat = torch.zeros((rows_pad, n), dtype=torch.float64, device=device)
    pending = [None for _ in range(nsets)]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps + args.warmup)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps + args.warmup)]
    tc_ev = []
    keys = [0]

    key_bounds = [None, int((days.max() - days.min()).item())]      # largest SNP distance (first N > 1 pass), largest day gap

    def finish(k):
        """transcluster over the WHOLE matrix of set k (all rows are present: own panels + gathered ones).  With N ranks the key
        evaluations are split: every rank evaluates its hash class of the distinct (N, day gap) keys into a dense key table, one
        all-reduce of the tables (16 B per key) completes them, and every rank reads P / E(K) of all cells from the table."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if world > 1:
            if key_bounds[0] is None:
                key_bounds[0] = int(sets[k][0][:n].max().item())     # same data every step: taken once (untimed warm-up)
            dev.trans_dist_dense_partitioned(sets[k][0], n, days, args.lamb, args.beta, args.precision, pmat, emat, rank, world,
                                             lambda t: dist.all_reduce(t), exp_p0=True, n_max=key_bounds[0], d_max=key_bounds[1])
        else:
            dev.trans_dist_dense_ranges(sets[k][0], n, days, args.lamb, args.beta, args.precision, pmat, emat, [(0, n)], exp_p0=True)
        b.record()
        tc_ev.append((a, b))
        keys[0] = int(_lib.load().tracs_debug_last_trans_dist_keys())

    # the panels travel in 16 bits per cell where the values allow it (partition.CompactPanels: decided from the first pass)
    cp = partition.CompactPanels(n, rank, world, dist) if world > 1 else None

    # One GPU: transcluster (f64 key evaluations: compute-bound) only reads the distances, which are final before the compared-
    # sites counts are -- the rest of the dense call (the N co-occurrence walk over the lists: memory-bound) runs beside it on a
    # second stream (tracs_pairsnp_notify_distances).  Off by default: the step is 4 % shorter with it (21.6 vs 22.5 ms), but
    # every kernel of the dense call then runs stretched beside the other stream's (the walk 17 instead of 14 ms) and the
    # per-kernel figures of `roofline` stop meaning the kernel.  TRACS_BENCH_OVERLAP=1: two streams.
    overlap = world == 1 and os.environ.get("TRACS_BENCH_OVERLAP", "0") == "1"
    side = torch.cuda.Stream(device=device) if overlap else None
    d_ready = torch.cuda.Event() if overlap else None
    main_stream = torch.cuda.current_stream()

    def step(it, per_call=True):
        k = it % nsets
        dmat, nmat = sets[k]
        if per_call:
            # ONE CALL: the planes count as freshly packed -- the dense call below decides the encoding and the site classes and
            # builds every list again (what src/pairsnp.hpp:320-457 does per alignment: one pairsnp call, nothing kept between calls)
            aln.mark_packed()
        if pending[k] is not None:                            # this set's exchange (two steps ago) must be over, and consumed
            cp.finish(k, dmat, nmat)
            finish(k)
            pending[k] = None
        # pairsnp: the dominant kernel, bracketed by HIP events on the launch stream
        if overlap:
            main_stream.wait_stream(side)                     # the previous step's transcluster has read the distances
            dev.notify_distances(d_ready)
        ev0[it].record()
        for r0, r1 in ranges:
            dev.pairsnp_dense(aln, dmat, nmat, row_begin=r0, row_end=r1)
        ev1[it].record()
        if world > 1:                                         # the d / nn panels travel while the next step's pair kernel runs
            pending[k] = cp.post(k, dmat, nmat, async_op=True)
        elif overlap:
            with torch.cuda.stream(side):
                side.wait_event(d_ready)
                finish(k)
        else:
            finish(k)

    def drain(next_it):
        for j in range(nsets):                                # oldest set first
            k = (next_it + j) % nsets
            if pending[k] is not None:
                cp.finish(k, sets[k][0], sets[k][1])
                finish(k)
                pending[k] = None

    lib = _lib.load()
    lib.tracs_debug_pair_timing(1)
    lib.tracs_debug_pack_timing(1)
    if overlap:
        dev.set_stream_policy(True)                           # this script orders its two streams with events
    # ---- ONE pass per alignment is the reference's unit of work (src/pairsnp.hpp:320-457: one pairsnp call per alignment): packed
    # planes resident -> d, nn, P, E(K), INCLUDING what the library decides and builds once per pack (encoding, site classes, the
    # counting pass's source, minority lists; csrc/site_classes.hip).  cold = the first pass of the process (this alignment),
    # warm = the same on a second, freshly packed handle.  Result matrices are allocated beforehand.  N = 1 only.
    single = None
    if world == 1:
        def one_pass(a):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if overlap:
                dev.notify_distances(d_ready)
            dev.pairsnp_dense(a, sets[0][0], sets[0][1])
            with torch.cuda.stream(side if overlap else main_stream):
                if overlap:
                    side.wait_event(d_ready)
                dev.trans_dist_dense_ranges(sets[0][0], n, days, args.lamb, args.beta, args.precision, pmat, emat, [(0, n)], exp_p0=True)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) * 1e3
        cold_ms = one_pass(aln)
        cold_stages = dev.pack_stages()
        aln2 = dev.Alignment(n, L)
        synth.pack_synthetic_device(aln2, seed=seed, **synth_kw(args.partial, args.workload))
        warm_ms = one_pass(aln2)
        warm_stages = dev.pack_stages()
        aln2.close()
        del aln2
        single = {"cold_ms": cold_ms, "warm_ms": warm_ms, "value_single_pass": n * (n - 1) // 2 / (warm_ms / 1e3), "unit": "pairs/s",
                  "per_pack_ms": sum(ms for _, ms in warm_stages),
                  "stages_ms": {k: round(v, 3) for k, v in warm_stages}, "cold_stages_ms": {k: round(v, 3) for k, v in cold_stages},
                  "note": "packed planes resident -> d, nn, P, E(K) for ONE pass over a freshly packed alignment, with everything the library "
                          "decides and builds once per pack (stages: HIP events on the launch stream); cold = first pass of the process, warm = "
                          "a second handle packed afterwards (wall clock around one call, host synchronisation included); `value` times K such calls back to back"}
        t_first = cold_ms / 1e3
    else:
        # first pass over this rank's panels (untimed): the once-per-pack work, and what the exchange needs to know -- do the
        # distances fit 16 bits, do the compared-sites counts span less than 65 536
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        for r0, r1 in ranges:
            dev.pairsnp_dense(aln, sets[0][0], sets[0][1], row_begin=r0, row_end=r1)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t_first
        cp.decide(sets[0][0], sets[0][1])
    def timed_run(first_it, count, per_call):
        """`count` steps bracketed by barrier + synchronize on both sides -> seconds (max over ranks)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(first_it, first_it + count):
            step(it, per_call)
        drain(first_it + count)                               # every step's panels have arrived and every matrix is complete on every rank
        torch.cuda.synchronize()                              # (both streams)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- the timed region: K CALLS, each over a freshly touched alignment (once-per-pack work included) -------------------------
    for it in range(args.warmup):
        step(it)
    drain(args.warmup)
    torch.cuda.synchronize()
    del tc_ev[:]
    elapsed = timed_run(args.warmup, args.steps, True)
    split_calls = args.steps if len(ranges) == 1 else 1
    split = pair_split_ms(lib, split_calls)                       # the timed calls' dense kernels (mean; two ranges per step: the last call)
    call_stages = dev.pack_stages_bytes()                         # the once-per-pack stages of the last timed call
    timed = range(args.warmup, args.warmup + args.steps)
    kern_ms = [ev0[i].elapsed_time(ev1[i]) for i in timed]
    tc_ms = [a.elapsed_time(b) for a, b in tc_ev] or [0.0]
    # ---- steady state: the same K steps over the alignment as it stands (nothing rebuilt) -- round 1-3's `value` --------------------
    steady_first = args.warmup + args.steps
    elapsed_steady = timed_run(steady_first, args.steps, False)
    steady_kern_ms = [ev0[i].elapsed_time(ev1[i]) for i in range(steady_first, steady_first + args.steps)]

    pairs_total = n * (n - 1) // 2
    my_pairs = sum(partition.pairs_in_rows(n, r0, r1) for r0, r1 in ranges)
    kern_s = sum(kern_ms) / len(kern_ms) / 1e3 / len(ranges)      # average duration of ONE dense call (all its kernels)
    my_pairs_per_launch = my_pairs / len(ranges)
    classes = aln.site_classes                                    # (variable, invariant) sites, or None: whole alignment read

    dmat, nmat = sets[(args.warmup + 2 * args.steps - 1) % nsets]    # the last step's results
    checksum = int(dmat[:n].sum().item()) if rank == 0 else 0
    if os.environ.get("TRACS_BENCH_VERIFY") and rank == 0:
        # the gathered matrices must equal a single-pass recomputation on this rank
        d1, n1 = torch.zeros_like(dmat), torch.zeros_like(nmat)
        p1, e1 = torch.zeros_like(pmat), torch.zeros_like(emat)
        aln.hint_rows([])                                     # every row again (the lists are re-built)
        dev.pairsnp_dense(aln, d1, n1)
        dev.trans_dist_dense_ranges(d1, n, days, args.lamb, args.beta, args.precision, p1, e1, [(0, n)], exp_p0=True)
        # (cells (i, j > i) only: the exchange does not carry what sits on or below the diagonal)
        up = torch.triu(torch.ones((n, n), dtype=torch.bool, device=device), diagonal=1)
        ok = bool(torch.equal(d1[:n][up], dmat[:n][up]) and torch.equal(n1[:n][up], nmat[:n][up]) and
                  torch.equal(p1[:n][up], pmat[:n][up]) and torch.equal(e1[:n][up], emat[:n][up]) and cp.check(dmat, nmat))
        print("VERIFY gathered == single-pass:", ok, file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit("VERIFY FAILED")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = pairs_total * args.steps / elapsed
        enc = aln.encoding or "general"
        roof = dense_call_roofline(aln, split, n, L, world, partition.pairs_in_rows(n, *ranges[-1]), ranges[-1][1] - ranges[-1][0],
                                   my_pairs_per_launch, kern_s, split_calls, overlap)
        enc_name = "consensus (ACGTN) alignment" if enc == "consensus" else "general IUPAC alignment (%.2g partial codes)" % args.partial
        W = WORKLOADS[args.workload]
        out = {"metric": "sample-pairs/sec for 10kx5Mbp SNP+transcluster distance", "value": value,
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "step": "ONE CALL per step: packed planes resident in HBM -> d, compared sites, P(direct), E(K) of all pairs, with everything "
                       "the library decides and builds per alignment (encoding, site classes, lists) redone in every step -- the reference's "
                       "unit of work, one pairsnp call per alignment (src/pairsnp.hpp:320-457).  value_steady_state: repeated passes over "
                       "one packed alignment, nothing rebuilt (what rounds 1-3 reported as `value`)",
               "value_steady_state": pairs_total * args.steps / elapsed_steady,
               "ms_per_step_steady_state": elapsed_steady / args.steps * 1e3,
               "roofline_per_pack": per_pack_roofline(call_stages),
               "config": {"workload": "%d samples x %d sites, %s, mu = %g per sample + %g N (%s): pairsnp (d + compared "
                                      "sites) + transcluster (P, E(K)), all %d pairs"
                                      % (n, L, enc_name, W["mu_sample"], W["p_n"], "SURVEY 8d" if args.workload == "sparse" else
                                         "workload '%s': %s" % (args.workload, W), pairs_total),
                          "samples": n, "sites": L, "pairs": pairs_total, "encoding": enc, "kernel": aln.kernel,
                          "mean_d": checksum / float(pairs_total), "distinct_keys": keys[0],
                          "clock_rate": args.lamb, "trans_rate": args.beta, "precision": args.precision,
                          "transcluster_ms_per_step": sum(tc_ms) / len(tc_ms),
                          "partition": ("one rank: no exchange" if world == 1 else
                                        "row panels, fold pairing, %d ranks; RCCL all-gather of the d / nn panels (%d bytes per cell: 16 bits where the "
                                        "values fit); P and E(K) derived on every rank from the gathered d, key evaluations split over the ranks "
                                        "(key-table all-reduce)" % (world, cp.bytes_per_cell())),
                          "result_placement": "whole matrices on every rank",
                          "workload_name": args.workload, "exchange": exchange,
                          "streams": ("2: transcluster on a second stream beside the rest of the dense call, from the moment the distances "
                                      "are final (tracs_pairsnp_notify_distances)") if overlap else "1",
                          "setup_seconds": round(setup_s, 1), "first_call_ms": round(t_first * 1e3, 1), "checksum_d": checksum},
               "roofline": roof}
        if single is not None:
            out["single_pass"] = single
        if world == 1 and not args.no_extras and args.partial == 0 and args.workload == "sparse":
            out["sensitivity"] = sensitivity(args, n, L, seed, days, dev, synth, torch, device, lib, value)
            out["value_worst_workload"] = min([value] + [w["pairs_per_s"] for w in out["sensitivity"]["workloads"].values()])   # per call, like `value`
        filter_block = None
        if world == 1 and not args.no_extras:
            # `tracs distance --filter`: the recombination filter over every emitted pair of the timed alignment (not part of `value`)
            out["filter"] = filter_leg(n, L, seed, synth_kw(args.partial, args.workload), aln, dmat, nmat, dev, synth, torch, device)
            filter_block = out["filter"].pop("_block", None)
        if world == 1 and not args.no_extras and args.partial == 0:
            out["roofline_general"] = general_pass(args, n, L, seed, dev, synth, torch, device)
            out["dm_frontend"] = dm_frontend(args, L, dev, torch, device)
        if world > 1:
            out["config"]["communicator"], out["config"]["ranks_seen"] = comm_info, comm_info["ranks_seen"]
            if not args.no_cpu_baseline:                       # (the gathered matrix is whole on every rank)
                keys[0] = distinct_keys(torch, dmat, n, days, [(0, n)])
                out["config"]["distinct_keys"] = keys[0]
        # ALWAYS: the first 128 samples' block against the oracle (full length, bit-equal), with or without the timed CPU legs
        out["cpu_baseline"] = cpu_baseline(n, L, seed, days_np, args, dmat, nmat, keys[0], check_only=args.no_cpu_baseline,
                                           filter_block=filter_block)
        if filter_block is not None and "filter" in out:
            out["filter"]["oracle_check"] = out["cpu_baseline"].get("filter")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def distinct_keys(torch, dmat, n, days, ranges, dist=None, world=1):
    """Distinct (SNP distance, day gap) keys over the cells (i, j > i) of the rows `ranges` of dmat -- over every rank's rows when
    `dist` is given (each rank marks its rows' keys in a bitmap indexed by the key, one all-reduce MAX merges them).  Untimed: what
    the cpu_baseline leg needs to know about the whole matrix when the ranks each hold only their rows."""
    device = dmat.device
    gmax = int((days.max() - days.min()).item())
    top = torch.zeros(1, dtype=torch.int64, device=device)
    step = 1024
    col = torch.arange(n, device=device)[None, :]

    def panels():
        for r0, r1 in ranges:
            for a in range(r0, r1, step):
                b = min(r1, a + step)
                row = torch.arange(a, b, device=device)[:, None]
                yield dmat[a:b, :n].to(torch.int64), (days[a:b, None] - days[None, :]).abs().to(torch.int64), col > row
    for d, _, up in panels():
        if bool(up.any()):
            top = torch.maximum(top, d[up].max().reshape(1))
    if dist is not None and world > 1:
        dist.all_reduce(top, op=dist.ReduceOp.MAX)
    dmax = int(top.item())
    marks = torch.zeros((dmax + 1) * (gmax + 1), dtype=torch.uint8, device=device)
    for d, gap, up in panels():
        marks[(d * (gmax + 1) + gap)[up]] = 1
    if dist is not None and world > 1:
        dist.all_reduce(marks, op=dist.ReduceOp.MAX)
    return int(marks.sum(dtype=torch.int64).item())


def site_sharded(args, n, L, seed, world, rank, device, dist, exchange, comm_info):
    """N ranks, each with a slice of the SITES (whole 128-site groups: a contiguous 1 / N of the packed planes).  d(i, j) and the
    compared-sites count nn(i, j) are sums over sites (src/pairsnp.hpp:398-403,417-420), so a rank runs the single-GPU call on its
    slice for ALL pairs -- classification, lists, walks: every stage works on 1 / N of the sites -- and the N partial matrices are
    summed by the compact exchange (tracs_amd/partition.py TriExchange, csrc/exchange.hip): the cells (i, j > i) only, 16 bits per
    cell where the slice's values fit (d as it is, nn as its deficit below the slice's length), one all-to-all (every pair of ranks
    over its own xGMI link), summed by the receiver.  The result STAYS distributed: rank q owns the rows of chunks q and 2N-1-q (fold
    pairing: equal cell counts) of d, nn, P and E(K) -- what `tracs distance --gpus N` does with the rows it then extracts -- so
    there is no all-gather, and transcluster runs on a rank's own rows (its keys evaluated there; no table exchange).
    One step = one call: the slice counts as freshly packed, everything once-per-pack is redone."""
    import torch
    from tracs_amd import _lib
    from tracs_amd import device as dev
    from tracs_amd import partition, synth
    groups = (L + 127) // 128
    g0, g1 = groups * rank // world, groups * (rank + 1) // world
    l0, l1 = g0 * 128, min(L, g1 * 128)
    t0 = time.time()
    aln = dev.Alignment(n, l1 - l0)
    synth.generate_device(n, L, seed, lambda rows, first: aln.pack(rows[:, l0:l1].contiguous(), first=first), **synth_kw(args.partial, args.workload))
    _, days_np = synth.dates(n, seed=seed)
    days = torch.from_numpy(days_np).to(device)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    rows_pad = (n + 63) // 64 * 64
    dmat = torch.zeros((rows_pad, n), dtype=torch.int32, device=device)
    nmat = torch.zeros((rows_pad, n), dtype=torch.int32, device=device)
    pmat = torch.zeros((rows_pad, n), dtype=torch.float64, device=device)
    emat = torch.zeros((rows_pad, n), dtype=torch.float64, device=device)
    lib = _lib.load()
    lib.tracs_debug_pair_timing(1)
    lib.tracs_debug_pack_timing(1)
    ex = partition.TriExchange(n, 0, n, 0, rank, world, dist, device)
    own = ex.own_ranges                                        # the rows this rank owns of every result (at most two ranges)
    # transcluster over the own rows with the key evaluations split over the ranks (partition.KeySplit: every distinct key of the whole
    # matrix evaluated by ONE rank, two small all-gathers); TRACS_KEY_SPLIT=0: every rank evaluates the keys of its own rows
    ks = partition.KeySplit(n, rank, world, dist, device) if os.environ.get("TRACS_KEY_SPLIT", "1") != "0" else None

    def step(per_call=True):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if per_call:
            aln.mark_packed()
        marks[0].record()
        dev.pairsnp_dense(aln, dmat, nmat)                     # all pairs over this rank's sites
        marks[1].record()
        ex.run(dmat, nmat, l1 - l0, L)                         # pack -> all-to-all -> sum into the own rows
        marks[2].record()
        if ks is not None:
            ks.run(dmat, days, own, args.lamb, args.beta, args.precision, pmat, emat, exp_p0=True)
        elif own:
            dev.trans_dist_dense_ranges(dmat, n, days, args.lamb, args.beta, args.precision, pmat, emat, own, exp_p0=True)
        marks[3].record()
        return marks

    def timed_run(count, per_call):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = time.perf_counter()
        got = [step(per_call) for _ in range(count)]
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t], dtype=torch.float64, device=device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), got

    # first call (untimed): the once-per-pack work, and what the exchange needs to know -- do a slice's partial distances and the
    # deficits of its compared-sites counts fit 16 bits (agreed over the ranks; a later call that does not fit is caught: ex.check)
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    aln.mark_packed()
    dev.pairsnp_dense(aln, dmat, nmat)
    ex.decide(dmat, nmat, l1 - l0)
    ex.run(dmat, nmat, l1 - l0, L)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    for _ in range(args.warmup):
        step()
    elapsed, marks = timed_run(args.steps, True)
    stages = dev.pack_stages_bytes()
    split = pair_split_ms(lib, args.steps)
    kern_ms = sum(m[0].elapsed_time(m[1]) for m in marks) / len(marks)
    elapsed_steady, _ = timed_run(args.steps, False)
    if not ex.check():
        raise SystemExit("the exchange overflowed its 16-bit cells in a later call (rank %d)" % rank)
    pairs_total = n * (n - 1) // 2
    # (every rank holds the sums of its own rows only: the checksum of d is a sum over the ranks)
    mine = torch.zeros((rows_pad, n), dtype=torch.bool, device=device)
    for q0, q1 in own:
        mine[q0:q1] = True
    mine &= torch.triu(torch.ones((rows_pad, n), dtype=torch.bool, device=device), diagonal=1)
    cks = torch.tensor([int(dmat[mine].sum().item())], dtype=torch.int64, device=device)
    dist.all_reduce(cks)
    checksum = int(cks.item())
    n_keys = distinct_keys(torch, dmat, n, days, own, dist, world) if not args.no_cpu_baseline else 0
    if os.environ.get("TRACS_BENCH_VERIFY"):
        # this rank's rows must equal the same rows of a single call over the whole alignment
        full = dev.Alignment(n, L)
        synth.pack_synthetic_device(full, seed=seed, **synth_kw(args.partial, args.workload))
        d1, n1 = torch.zeros_like(dmat), torch.zeros_like(nmat)
        p1, e1 = torch.zeros_like(pmat), torch.zeros_like(emat)
        dev.pairsnp_dense(full, d1, n1)
        dev.trans_dist_dense_ranges(d1, n, days, args.lamb, args.beta, args.precision, p1, e1, [(0, n)], exp_p0=True)
        ok = bool(torch.equal(d1[mine], dmat[mine]) and torch.equal(n1[mine], nmat[mine]) and
                  torch.equal(p1[mine], pmat[mine]) and torch.equal(e1[mine], emat[mine]))
        okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
        dist.all_reduce(okt)
        if rank == 0:
            print("VERIFY site shards == single call:", int(okt.item()) == world, file=sys.stderr, flush=True)
        full.close()
        if int(okt.item()) != world:
            raise SystemExit("VERIFY FAILED")
    if rank == 0:
        def mean_ms(a, b):
            return sum(m[a].elapsed_time(m[b]) for m in marks) / len(marks)
        W = WORKLOADS[args.workload]
        # rank 0's dense call over its slice: all pairs, all rows, L = the slice's sites
        roof = dense_call_roofline(aln, split, n, l1 - l0, world, pairs_total, n, pairs_total, kern_ms / 1e3, args.steps)
        roof["scope"] = ("rank 0's dense call over its slice of the sites (%d of %d): every rank runs the same kernels on 1 / %d of the "
                         "sites for all pairs; the step is shared between the slice's call, the exchange and transcluster (config.rank0_ms)"
                         % (l1 - l0, L, world))
        out = {"metric": "sample-pairs/sec for 10kx5Mbp SNP+transcluster distance", "value": pairs_total * args.steps / elapsed,
               "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "step": "ONE CALL per step (as at N = 1): every rank's slice counts as freshly packed, the once-per-pack work is redone in every "
                       "step; value_steady_state: repeated passes, nothing rebuilt",
               "value_steady_state": pairs_total * args.steps / elapsed_steady,
               "ms_per_step_steady_state": elapsed_steady / args.steps * 1e3,
               "config": {"workload": "%d samples x %d sites, mu = %g per sample + %g N (%s): pairsnp (d + compared sites) + transcluster (P, E(K)), "
                                      "all %d pairs" % (n, L, W["mu_sample"], W["p_n"], "SURVEY 8d" if args.workload == "sparse" else
                                                         "workload '%s'" % args.workload, pairs_total),
                          "samples": n, "sites": L, "pairs": pairs_total, "workload_name": args.workload, "exchange": exchange,
                          "communicator": comm_info, "ranks_seen": comm_info["ranks_seen"],
                          "result_placement": "distributed rows: rank q ends with d, nn, P and E(K) of the rows it owns (no all-gather)",
                          "partition": "SITE shards: rank r holds groups [%d r / %d, ..) of the packed planes (%d of %d sites on rank 0) and counts all "
                                       "pairs over them; d and nn summed by the compact exchange -- upper-triangle cells, %d + %d bytes per cell "
                                       "(d; nn as its deficit below the slice's length), all-to-all, summed by the receiver --: rank q owns the rows "
                                       "of chunks q and %d - q (%d rows each; no all-gather; transcluster on a rank's own rows)"
                                       % (groups, world, l1 - l0, L, ex.widths[0], ex.widths[1], 2 * world - 1, ex.cs),
                          "transcluster_keys": ({"route": ks.last_route, "distinct_keys_whole_matrix": ks.last_info[0],
                                                 "evaluated_by_rank0": ks.last_evaluated if ks.last_route == "split" else None,
                                                 "bytes_gathered_per_rank_per_call": ks.bytes_gathered_per_call(),
                                                 "how": "every rank marks its rows' (N, day gap) keys in a bitmap, all-gather + OR, rank r evaluates the "
                                                        "keys of ordinal r mod P, all-gather of the compact (log p0, E(K)) arrays, every rank fills its table"}
                                                if ks is not None and ks.last_info is not None else
                                                {"route": "every rank evaluates the keys of its own rows"}),
                          "rank0_ms": {"dense call over the slice (once-per-pack work included)": mean_ms(0, 1),
                                       "exchange (pack, all-to-all, sum)": mean_ms(1, 2), "transcluster over the rank's own rows": mean_ms(2, 3)},
                          "exchange_bytes_per_rank_per_call": ex.bytes_sent_per_call(), "exchange_bytes_per_cell": ex.bytes_per_cell(),
                          "kernels_ms": None if not split else dict(zip(("pair", "lists", "count", "nn_lists"), split)),
                          "mean_d": checksum / float(pairs_total), "checksum_d": checksum, "distinct_keys": n_keys or None,
                          "clock_rate": args.lamb, "trans_rate": args.beta, "precision": args.precision,
                          "setup_seconds": round(setup_s, 1), "first_call_ms": round(t_first * 1e3, 1)},
               "roofline_per_pack": per_pack_roofline(stages),
               "roofline": roof}
        # ALWAYS: the first samples' block of the rows rank 0 owns against the oracle (full length, bit-equal), with or without the
        # timed CPU legs
        m = min(128, own[0][1] - own[0][0]) if own and own[0][0] == 0 else 0
        if m >= 2:
            out["cpu_baseline"] = cpu_baseline(n, L, seed, days_np, args, dmat, nmat, n_keys, m_max=m, check_only=args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


_WORKLOAD_OF_PROFILES = None        # set by main(): the workload of this run


def _traffic_from_profiles(n, L, world, kernel):
    """HBM bytes per launch of the kernel that ran, from the COMMITTED PMC summary (profiles/pmc_summary.json) if it holds an
    entry for this size and kernel: a figure from separate --pmc passes of the same kernel on the same shape (FETCH_SIZE /
    WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), not from this run; None when there is no such entry."""
    if _WORKLOAD_OF_PROFILES is not None and _WORKLOAD_OF_PROFILES != "sparse":
        return None                                           # (the committed passes are of the default workload)
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as fh:
            e = json.load(fh).get("%dx%d@%d/%s" % (n, L, world, kernel))
        return None if e is None else e.get("hbm_bytes_per_launch")
    except Exception:
        return None


def sensitivity(args, n, L, seed, days, dev, synth, torch, device, lib, value_default):
    """The same CALL on seven other synthetic alignments of the same shape (WORKLOADS + `partial`), on the headline's unit: ONE CALL
    per step -- the planes count as freshly packed before every call (mark_packed, like step()), so the encoding, the site classes
    and every list are rebuilt inside the timed region -- pairsnp + transcluster, 2 calls after one untimed first call, HIP events.
    Beside it the steady-state pass over the alignment as it stands (rounds 1-5's figure), the class sizes, the kernels' split, and
    the same handle with every site through the pair kernel (tracs_debug_force_site_classes(0)), whose checksums must agree.  The
    reference's cost does not depend on the data (src/pairsnp.hpp:395-420 visits every site of every pair); this path's does:
    `value_worst_workload` and `spread` come from the PER-CALL figures only."""
    pairs = n * (n - 1) // 2
    dmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    nmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    pmat = torch.zeros((n, n), dtype=torch.float64, device=device)
    emat = torch.zeros((n, n), dtype=torch.float64, device=device)

    def timed(a, per_call=True):
        def one(fresh):
            if fresh:
                a.mark_packed()
            dev.pairsnp_dense(a, dmat, nmat)
            dev.trans_dist_dense_ranges(dmat, n, days, args.lamb, args.beta, args.precision, pmat, emat, [(0, n)], exp_p0=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one(False)
        torch.cuda.synchronize()
        first = (time.perf_counter() - t0) * 1e3
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        if per_call:
            one(True); one(True)
        e[1].record()
        one(False); one(False)
        e[2].record()
        torch.cuda.synchronize()
        return {"first": first, "call": e[0].elapsed_time(e[1]) / 2 if per_call else None, "steady": e[1].elapsed_time(e[2]) / 2,
                "cd": int(dmat.sum().item()), "cn": int(nmat.sum().item())}
    out = {}
    for name in ("lineage", "divergent", "clean", "gappy", "runs", "partial", "coverage"):
        a = dev.Alignment(n, L)
        # ("partial": the metric's alignment + 0.5 % partial IUPAC codes -- what `tracs align` writes without --consensus, tracs/align.py:616-622)
        synth.pack_synthetic_device(a, seed=seed, **(synth_kw(P_PARTIAL_C4, "sparse") if name == "partial" else synth_kw(0.0, name)))
        r = timed(a)
        stages = dev.pack_stages()
        classes, kernel, split, a_count, gram = a.site_classes, a.kernel, pair_split_ms(lib), a.count_source, a.nw_form
        lib.tracs_debug_force_site_classes(0)
        a.mark_packed()
        r0 = timed(a, per_call=False)
        lib.tracs_debug_force_site_classes(-2)
        a.close()
        if (r["cd"], r["cn"]) != (r0["cd"], r0["cn"]):
            raise SystemExit("PARITY FAILURE: workload %s, site classes change the result (%d, %d) vs (%d, %d)"
                             % (name, r["cd"], r["cn"], r0["cd"], r0["cn"]))
        out[name] = {"ms_per_call": r["call"], "pairs_per_s": pairs / (r["call"] / 1e3),
                     "ms_per_pass_steady_state": r["steady"], "pairs_per_s_steady_state": pairs / (r["steady"] / 1e3),
                     "first_call_ms": r["first"], "once_per_call_ms": sum(ms for _, ms in stages),
                     "once_per_call_stages_ms": {k: round(v, 3) for k, v in stages},
                     "ms_per_pass_classes_off": r0["steady"], "first_call_ms_classes_off": r0["first"],
                     "site_classes": None if classes is None else dict(zip(("dense", "counted", "minority", "full"), classes)),
                     "kernel": kernel, "kernels_ms": None if not split else dict(zip(("pair", "lists", "count", "nn_lists"), split)),
                     "count_source": a_count,
                     "n_x_listed_terms": {"u-pass": "two one-plane passes on the matrix cores (U U^T - n n^T: nw_gram)",
                                          "ns-rows": "rows of the site-major N matrix summed per listed sample (nw_rows)"}.get(gram, "walks of the sites' N lists"),
                     "mean_d": r["cd"] / float(pairs), "checksum_d": r["cd"], "checksum_nn": r["cn"],
                     "generator": dict(WORKLOADS["sparse"], p_partial=P_PARTIAL_C4) if name == "partial" else WORKLOADS[name]}
    worst = min(out, key=lambda k: out[k]["pairs_per_s"])
    rates = [value_default] + [w["pairs_per_s"] for w in out.values()]
    return {"workloads": out, "worst": worst, "spread": max(rates) / min(rates), "unit": "ONE CALL per step, like `value`",
            "note": "10k x 5 Mbp each; ms_per_call / pairs_per_s: the planes freshly packed before every call (everything the library builds "
                    "per alignment inside the timed region, pairsnp + transcluster); *_steady_state: passes over the alignment as it stands; "
                    "classes_off: every site through the pair kernel (same handle, re-decided); checksums of d and nn equal"}


def filter_leg(n, L, seed, kw, aln, dmat, nmat, dev, synth, torch, device, snp_threshold=100, scan_sample=200000, check=24, reps=2):
    """`tracs distance --filter` at the bench's size: the recombination filter (src/pairsnp.hpp:251-318) on every pair the dense call
    emits (:405-413), planes and distance matrix resident in HBM.  Seconds per call over ALL emitted pairs:
      first_call_s   on a freshly packed handle: departure lists + N bitmaps built (index_build_ms), thresholds built, pairs filtered
      warm_call_s    the same call again (index and thresholds kept on the handle)
      threshold      the same with -D snp_threshold (only the pairs within the threshold are emitted)
      scan_route     rounds 1-5's route (a pair's SNP bits re-derived from the planes) on a bounded sample, extrapolated by pair count
    The first `check` samples' filtered distances go back under "_block" for cpu_baseline to hold against the oracle (`filter` there)."""
    import numpy as np
    rows, cols, d, _ = dev.coo_from_dense(dmat, nmat, n)
    pairs = rows.numel()
    out = {"pairs": pairs, "max_d": int(d.max().item()) if pairs else 0}

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t
    aln.mark_packed()
    probe = min(pairs, 4096)
    _, t_probe = timed(lambda: dev.filter_recomb_pairs(aln, rows[:probe], cols[:probe], d[:probe]))       # builds the index
    info = dev.filter_index_info(aln)
    listed = bool(info and info["lists"] and info["longest_list"] <= 4096)
    out.update({"index": info, "index_build_ms": None if info is None else sum(info["build_ms"].values()), "first_probe_call_s": t_probe,
                "route": "departure lists" if listed else "scan of the planes (lists too long for a wave's LDS, or not built)"})
    if not listed:
        # every pair would take the scan (minutes at this size): a bounded sample, extrapolated by pair count
        m = min(pairs, scan_sample)
        sel = torch.arange(0, pairs, max(1, pairs // m), device=device)[:m]
        rows, cols, d = rows[sel].contiguous(), cols[sel].contiguous(), d[sel].contiguous()
        out["sampled_pairs"] = int(m)
    aln.mark_packed()
    filt, t_first = timed(lambda: dev.filter_recomb_pairs(aln, rows, cols, d))
    warm = []
    for _ in range(reps):
        f2, t = timed(lambda: dev.filter_recomb_pairs(aln, rows, cols, d))
        warm.append(t)
        if not torch.equal(f2, filt):
            raise SystemExit("PARITY FAILURE: two filter calls over the same pairs differ")
    scale = pairs / float(rows.numel())
    out.update({"first_call_s": t_first * scale, "warm_call_s": min(warm) * scale, "pairs_per_s": rows.numel() / min(warm),
                "mean_filtered_d": float(filt.to(torch.float64).mean().item()), "pairs_with_fewer_snps": int((filt < d).sum().item()),
                "checksum_filt": int(filt.to(torch.int64).sum().item())})
    pairs_run = rows.numel()
    (r2, c2, d2, _), t_coo = timed(lambda: dev.coo_from_dense(dmat, nmat, n, snp_threshold))
    t_thr = timed(lambda: dev.filter_recomb_pairs(aln, r2, c2, d2))[1] if r2.numel() else 0.0
    out["threshold"] = {"D": snp_threshold, "pairs": int(r2.numel()), "coo_s": t_coo, "filter_s": t_thr}
    m = min(pairs_run, scan_sample)
    if m:
        sel = torch.arange(0, pairs_run, max(1, pairs_run // m), device=device)[:m]
        rs, cs, ds = rows[sel].contiguous(), cols[sel].contiguous(), d[sel].contiguous()
        (fs, found, _, _), t_scan = timed(lambda: dev.filter_recomb_device(aln, rs, cs, ds))
        if not (torch.equal(found, ds) and torch.equal(fs, filt[sel])):
            raise SystemExit("PARITY FAILURE: the list route and the scan route of the filter differ")
        out["scan_route"] = {"pairs": int(m), "seconds": t_scan, "pairs_per_s": m / t_scan, "all_pairs_s": t_scan * pairs / m,
                             "bytes_per_pair": float(L), "achieved_GBps": m * float(L) / t_scan / 1e9,
                             "note": "tracs_filter_recomb_device: 8 planes x L / 8 bytes per pair; extrapolated to all pairs"}
    # the first `check` samples' pairs, for cpu_baseline (the one place that may touch oracle/) to hold against the oracle's filter_recomb
    k = min(n, check)
    if k >= 2:
        iu = np.triu_indices(k, 1)                              # row-major pair order (src/pairsnp.hpp:451-455)
        ri, ci = torch.from_numpy(iu[0].astype(np.int32)).to(device), torch.from_numpy(iu[1].astype(np.int32)).to(device)
        gd = dmat[ri.long(), ci.long()].contiguous()
        out["_block"] = {"samples": k, "rows": iu[0], "cols": iu[1], "d": gd.cpu().numpy(), "filt": dev.filter_recomb_pairs(aln, ri, ci, gd).cpu().numpy()}
    return out


def general_pass(args, n, L, seed, dev, synth, torch, device):
    """The general-encoding path on the same workload with SURVEY 8d's C4 mix of partial codes: its own alignment handle,
    2 passes timed with HIP events (not part of `value`)."""
    aln = dev.Alignment(n, L)
    synth.pack_synthetic_device(aln, seed=seed, **synth_kw(P_PARTIAL_C4))
    dmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    nmat = torch.zeros((n, n), dtype=torch.int32, device=device)
    dev.pairsnp_dense(aln, dmat, nmat)                        # warm-up: decides the encoding, builds the sparse lists (untimed setup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 2
    e0.record()
    for _ in range(reps):
        dev.pairsnp_dense(aln, dmat, nmat)
    e1.record()
    torch.cuda.synchronize()
    kern_s = e0.elapsed_time(e1) / 1e3 / reps
    # ONE CALL on this alignment (what `value` measures on the default one): the planes count as freshly packed, the once-per-pack
    # work is redone -- with partial codes every sample is listed at tens of thousands of sites, so the lists weigh more here
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(reps):
        aln.mark_packed()
        dev.pairsnp_dense(aln, dmat, nmat)
    c1.record()
    torch.cuda.synchronize()
    call_s = c0.elapsed_time(c1) / 1e3 / reps
    call_stages = dev.pack_stages_bytes()                     # the once-per-call stages of the last of those calls
    from tracs_amd import _lib
    split, classes, pairs = pair_split_ms(_lib.load()), aln.site_classes, n * (n - 1) // 2
    traffic = _traffic_from_profiles(n, L, 1, aln.kernel + ("+classes" if classes else ""))
    if classes and split:
        count_sites, in_place, nn_listed = aln.count_source or (classes[1], False, 0)
        main = roofline_of(aln.kernel, "general", pairs, classes[0], max(split[0], 1e-3) / 1e3, None, n)
        cnt = count_roofline(pairs, count_sites, max(split[2], 1e-3) / 1e3, n, in_place)
        nnl = nn_list_roofline(aln.list_stats, nn_listed, max(split[3], 1e-3) / 1e3, n, n, L) if nn_listed else None
        cands = sorted([c for c in ((split[0], main), (split[2], cnt), (split[3], nnl)) if c[1] is not None], key=lambda c: -c[0])
        r = cands[0][1]
        r["other_kernels"] = [{k: c[1][k] for k in ("kernel", "kernel_ms", "bound", "achieved", "frac", "unit", "sites") if k in c[1]} for c in cands[1:]]
        r["lists_ms"] = split[1]
        r["site_classes"] = {"dense": classes[0], "counted": classes[1], "minority": classes[2], "full": classes[3],
                             "empty": L - classes[0] - classes[1] - classes[3]}
    else:
        r = roofline_of(aln.kernel, "general", pairs, L, kern_s, traffic, n)
    r["dense_call_ms"] = kern_s * 1e3
    r["one_call_ms"] = call_s * 1e3                          # (pairsnp only: once-per-pack work + the dense call, no transcluster)
    r["roofline_per_pack"] = per_pack_roofline(call_stages)
    if split:
        r["kernels_ms"] = {"pairsnp_mfma_kernel": split[0], "general_fixup_kernel (partial codes of the dense sites + minority lists)": split[1],
                           "count_pass": split[2], "nn_rows_kernel": split[3]}
    r["workload"] = "the same alignment + %.3g partial IUPAC codes per site (uniformly random sites and codes)" % P_PARTIAL_C4
    r["mean_d"] = float(dmat.sum().item()) / (n * (n - 1) // 2)
    aln.close()
    return r


def dm_frontend(args, L, dev, torch, device):
    """counts -> per-site Dirichlet-multinomial posterior filter -> 4-bit allele masks -> packed planes, for a batch of
    samples (what `tracs align` does per sample before the FASTA exists, tracs/align.py:536-577,613-622, fused on the
    device), then the pair kernel over that batch.  HBM-bound: 8 B in (4 x uint16 counts) + 0.5 B out per site-row."""
    import numpy as np
    batch = 32
    g = torch.Generator(device=device)
    g.manual_seed(99)
    alphas = np.array([20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1])
    # SURVEY 8d, config 4 counts: depth ~ Poisson(30) on a random major allele, eps = 0.01 errors, 1 % two-allele sites
    counts = torch.zeros((batch, L, 4), dtype=torch.int16, device=device)
    ar = torch.arange(4, device=device)[None, :]
    for b in range(batch):
        major = torch.randint(0, 4, (L,), generator=g, device=device)
        depth = torch.poisson(torch.full((L,), 30.0, device=device), generator=g).to(torch.int16)
        minor = (major + 1 + torch.randint(0, 3, (L,), generator=g, device=device)) & 3
        err = (torch.rand(L, generator=g, device=device) < 0.25).to(torch.int16)
        two = (torch.rand(L, generator=g, device=device) < 0.01).to(torch.int16) * 9
        c = (ar == major[:, None]).to(torch.int16) * depth[:, None]
        c += (ar == minor[:, None]).to(torch.int16) * err[:, None]
        c += (ar == ((major + 2) & 3)[:, None]).to(torch.int16) * two[:, None]
        counts[b] = c
    del c
    aln = dev.Alignment(batch, L)
    stride = ((L + 1) // 2 + 15) // 16 * 16
    codes = torch.zeros((batch, stride), dtype=torch.uint8, device=device)
    dmat = torch.zeros((batch, batch), dtype=torch.int32, device=device)
    nmat = torch.zeros((batch, batch), dtype=torch.int32, device=device)

    thr = max(5.0 / 30.0, 0.01)          # tracs align's rule at its defaults: max(min_cov / median coverage, error threshold), align.py:521

    one_launch = stride * 2 == L         # the batch's codes are one contiguous run of site-rows: one launch for the batch

    def posterior():
        if one_launch:
            dev.posterior_codes_device(counts.view(batch * L, 4), alphas, False, thr, out=codes.view(-1))
            return
        for b in range(batch):
            dev.posterior_codes_device(counts[b], alphas, False, thr, out=codes[b])     # straight into the batch buffer

    def chain():
        posterior()
        aln.pack_codes(codes, 0)
        dev.pairsnp_dense(aln, dmat, nmat)
    chain()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    reps = 5                              # (one launch of ~0.3 ms: timed five times over)
    e[0].record()
    for _ in range(reps):
        posterior()
    e[1].record()
    aln.pack_codes(codes, 0)
    e[2].record()
    dev.pairsnp_dense(aln, dmat, nmat)
    e[3].record()
    torch.cuda.synchronize()
    t_post, t_pack, t_pair = e[0].elapsed_time(e[1]) / reps, e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])
    sites = float(batch) * L
    out = {"samples": batch, "sites_per_sample": L,
           "posterior_codes_ms": t_post, "pack_codes_ms": t_pack, "pairsnp_ms": t_pair, "encoding": aln.encoding,
           "posterior_codes_GBps": sites * 8.5 / (t_post / 1e3) / 1e9, "pack_codes_GBps": sites * (0.5 + 0.625) / (t_pack / 1e3) / 1e9,
           "site_rows_per_s": sites / ((t_post + t_pack) / 1e3),
           "posterior_threshold": thr,
           "launches": 1 if one_launch else batch,
           "note": "posterior_codes over the batch's site-rows (one launch when a sample's codes fill whole 16-byte rows of the batch "
                   "buffer, else one per sample; scripts/bench_config4.py streams 250 samples per launch), codes written straight into "
                   "the batch buffer; algorithmic bytes: 8.5 B per site-row (posterior), 1.125 B (pack)"}
    aln.close()
    return out


def cpu_baseline(n, L, seed, days_np, args, dmat, nmat, n_keys_full, m_max=128, check_only=False, filter_block=None):
    """The oracle (C/OpenMP port of the reference algorithm) on the host cores, on a bounded sample of the same workload.
    Two legs, reported separately and never extrapolated through each other:
      pair loop   the first m samples of the SAME alignment, all m(m-1)/2 pairs at full length L, both passes, as the
                  reference runs them (planes already packed, like the GPU's timed region), repeated until >= cpu-seconds;
      trans_dist  serial and memoised per (N, delta) key, as in the reference, over the distinct keys of those pairs (bounded
                  to the first ones that fit the time budget) -> keys/s.
    `value` = pairs / (pairs / pair-loop rate + KEYS OF THE FULL MATRIX / key rate), the full matrix's distinct-key count being
    the one the GPU's dedup table reported -- distinct keys saturate with the pair count, pairs do not."""
    import numpy as np
    from oracle import oracle as O
    from tracs_amd import synth
    # (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks: under a launcher the leg takes the cores this process may run on)
    cores = len(os.sched_getaffinity(0)) if in_launcher() else O.lib().orc_num_threads()
    m = int(max(2, min(n, m_max)))
    if check_only:
        m = min(m, 64)
    seqs = synth.first_samples_host(n, L, seed, m, **synth_kw(args.partial, args.workload))
    planes = O.pack(seqs)                                   # untimed, like the GPU side's resident planes
    pairs = m * (m - 1) // 2
    reps, t_snp = 0, 0.0
    while (t_snp < args.cpu_seconds and reps < 1000) and not (check_only and reps >= 1):
        t0 = time.perf_counter()
        r, c, d, nn = O.pairsnp_planes(planes, L, dist=2147483647, n_threads=cores)
        t_snp += time.perf_counter() - t0
        reps += 1
    pair_rate = pairs * reps / t_snp
    # the sample doubles as a full-size parity check: the GPU's d / nn for these pairs must be bit-equal
    ri, ci = r.astype(np.int64), c.astype(np.int64)
    gd = dmat[:m, :m].cpu().numpy().astype(np.int64)[ri, ci]
    gn = nmat[:m, :m].cpu().numpy().astype(np.int64)[ri, ci]
    if not (np.array_equal(gd, d.astype(np.int64)) and np.array_equal(gn, nn.astype(np.int64))):
        raise SystemExit("PARITY FAILURE: GPU d/nn differ from the oracle on the %d x %d sample block" % (m, m))
    # the recombination filter (src/pairsnp.hpp:251-318) of the first samples' pairs: the oracle's scan of the planes against the GPU's
    # filtered distances bench.py's `filter` leg handed over -- a result check (always) and the CPU rate beside the GPU's
    flt = None
    if filter_block is not None and filter_block["samples"] <= m:
        kf = filter_block["samples"]
        t0 = time.perf_counter()
        ef = O.filter_recomb_pairs(seqs[:kf], filter_block["rows"].astype(np.uint64), filter_block["cols"].astype(np.uint64), cores)
        t_f = time.perf_counter() - t0
        ok = bool(np.array_equal(filter_block["d"].astype(np.int64), dmat[:kf, :kf].cpu().numpy().astype(np.int64)[filter_block["rows"], filter_block["cols"]]) and
                  np.array_equal(filter_block["filt"].astype(np.int64), ef.astype(np.int64)))
        if not ok:
            raise SystemExit("PARITY FAILURE: filtered distances differ from the oracle on the first %d samples" % kf)
        flt = {"samples": kf, "pairs": int(len(ef)), "equal": True, "cpu_pairs_per_s": len(ef) / t_f, "cores": cores, "kind": "port"}
    if check_only:
        return {"value": None, "kind": "port", "cores": cores, "unit": "pairs/s", "filter": flt,
                "sample": "--no-cpu-baseline: only the result check ran -- first %d samples x %d sites, GPU d/nn bit-equal to the oracle" % (m, L)}
    delta = np.abs(days_np[ri] - days_np[ci]).astype(np.float64) * 86400.0 / 31556952.0
    # distinct keys in first-appearance order; time them in growing batches until the budget is spent
    _, first = np.unique(np.stack([d.astype(np.float64), delta]), axis=1, return_index=True)
    order = np.sort(first)
    kd, kdel = d[order].astype(np.int32), delta[order]
    done, t_tc, batch = 0, 0.0, 64
    while done < len(order) and t_tc < args.cpu_seconds:
        hi = min(len(order), done + batch)
        t1 = time.perf_counter()
        O.trans_dist(kd[done:hi], kdel[done:hi], args.lamb, args.beta, args.precision)
        t_tc += time.perf_counter() - t1
        done, batch = hi, batch * 2
    key_rate = done / t_tc
    pairs_total = n * (n - 1) // 2
    value = pairs_total / (pairs_total / pair_rate + n_keys_full / key_rate)
    ref = _reference_trans_dist(kd[:done], kdel[:done], args)
    return {"value": value, "unit": "pairs/s", "cores": cores, "kind": "port", "filter": flt,
            "pairsnp_pairs_per_s": pair_rate, "pairsnp_threads": cores, "pairsnp_seconds": t_snp,
            "trans_dist_keys_per_s": key_rate, "trans_dist_threads": 1, "trans_dist_seconds": t_tc,
            "trans_dist_reference": ref,
            "distinct_keys_full_matrix": n_keys_full,
            "sample": "pair loop: first %d samples x %d sites of the same alignment = %d pairs, oracle pair loop (two passes), %d OpenMP "
                      "threads, %d repeats = %.1f s; trans_dist: serial memoised, %d distinct (N, delta) keys of those pairs = %.1f s; "
                      "value = pairs / (pairs / pair rate + %d distinct keys of the full matrix (GPU dedup) / key rate); GPU d/nn "
                      "bit-equal on this block" % (m, L, pairs, cores, reps, t_snp, done, t_tc, n_keys_full)}


def _reference_trans_dist(kd, kdel, args):
    """The same distinct keys through oracle/_ref (the reference's own transcluster source compiled in place by oracle/Makefile
    with the reference's flags), serial, in a child process because that build's -ffast-math sets flush-to-zero for the whole
    process.  Reported beside the port's key rate, not folded into `value`; None when oracle/_ref was not built."""
    import subprocess, tempfile
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    if not os.path.isdir(os.path.join(here, "oracle", "_ref")) or len(kd) == 0:
        return None
    code = ("import sys, time, json, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\n"
            "R = O.ref_module()\n"
            "if R is None: print('null'); sys.exit(0)\n"
            "z = np.load(sys.argv[1]); N = z['N'].tolist(); D = z['D'].tolist()\n"
            "t0 = time.perf_counter(); e = R.ref_trans_dist(N, D, %r, %r, %r); t = time.perf_counter() - t0\n"
            "print(json.dumps({'keys': len(N), 'seconds': t, 'keys_per_s': len(N) / t, 'kind': 'reference', 'threads': 1}))\n"
            % (here, args.lamb, args.beta, args.precision))
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "keys.npz")
        np.savez(f, N=np.asarray(kd, dtype=np.int64), D=np.asarray(kdel, dtype=np.float64))
        try:
            out = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True,
                                 timeout=max(120.0, 20 * args.cpu_seconds))
            return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else None
        except (subprocess.TimeoutExpired, ValueError):
            return None


if __name__ == "__main__":
    main()
