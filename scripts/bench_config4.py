"""Config 4 at its shape (BASELINE.json configs[3]): 50 000 samples x 1 Mbp of per-site 4-allele uint16 counts
-> per-site Dirichlet-multinomial posterior filter -> 4-bit allele masks (posterior_codes_kernel, csrc/dmultinomial.hip;
reference: src/dmultinomial.hpp:8-86 applied per sample by tracs/align.py:536-577,613-622).

5 x 10^10 site-rows x 8 B = 400 GB of counts do not exist at once: they are streamed in batches of `--batch` samples.  The
batches are generated on the device (SURVEY 8d's C4 mix: depth ~ Poisson(30) on a random major allele, 1 % errors, 1 % two-allele
sites at 0.7 / 0.3), a small pool of distinct batches is cycled -- generating every batch afresh would time torch's RNG, not
the kernel -- and every batch's codes are written to their own place in a [pool] ring (0.5 B per site-row).  Timed region: all
launches of the stream, counts resident in HBM (through PCIe the same stream is bounded by the link: 400 GB at ~60 GB/s).
N ranks (launched under torch.distributed.run: WORLD_SIZE / RANK): replicas only -- rank r takes the samples r, r + N, ..; no collective in the data path (torch.distributed only for the
barrier and the max-over-ranks time).

usage: python scripts/bench_config4.py [--samples 50000] [--sites 1000000] [--batch 250]   (N GPUs: python -m torch.distributed.run --nproc-per-node N scripts/bench_config4.py ...)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=50000)
ap.add_argument("--sites", type=int, default=1000000)
ap.add_argument("--batch", type=int, default=250)
ap.add_argument("--pool", type=int, default=4)
ap.add_argument("--check", type=int, default=200000, help="site-rows of the first batch checked against the f64 kernel")
args = ap.parse_args()

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
d = torch.device("cuda", local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("TRACS_DIST_BACKEND", "nccl"), **({"device_id": d} if os.environ.get("TRACS_DIST_BACKEND", "nccl") == "nccl" else {}))

alphas = [20.8156311152126, 4.38181182238621, 0.889048781117318, 0.1]
L, B = args.sites, args.batch
rows = B * L                                                  # site-rows per batch
g = torch.Generator(device=d)
g.manual_seed(20241022 + 3 + rank)


def make_batch():
    """[B * L, 4] int16 counts (bit pattern uint16), SURVEY 8d's C4 mix."""
    major = torch.randint(0, 4, (rows,), generator=g, device=d)
    depth = torch.poisson(torch.full((rows,), 30.0, device=d), generator=g)
    two = torch.rand(rows, generator=g, device=d) < 0.01
    err = torch.poisson(depth * 0.01, generator=g)                         # errors, all on one other allele
    minor_n = torch.where(two, torch.floor(depth * 0.3), err)
    minor = (major + 1 + torch.randint(0, 3, (rows,), generator=g, device=d)) & 3
    c = torch.zeros((rows, 4), dtype=torch.int16, device=d)
    c.scatter_(1, major[:, None], (depth - minor_n).clamp_(min=0).to(torch.int16)[:, None])
    c.scatter_add_(1, minor[:, None], minor_n.to(torch.int16)[:, None])
    return c


t0 = time.time()
pool = [make_batch() for _ in range(args.pool)]
codes = [torch.empty((rows + 1) // 2, dtype=torch.uint8, device=d) for _ in range(args.pool)]
torch.cuda.synchronize()
setup_s = time.time() - t0

my_samples = len(range(rank, args.samples, world))
n_batches = (my_samples + B - 1) // B
lib = dev._lib.load()
import ctypes as C  # noqa: E402
a = np.ascontiguousarray(alphas, dtype=np.float64)
ap_ = a.ctypes.data_as(C.POINTER(C.c_double))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def launch(k, nrows):
    dev._lib.check(lib.tracs_posterior_codes_cov_device(C.c_void_p(pool[k].data_ptr()), nrows, ap_, 0, 5.0 / 30.0, 0, 1.0, 0.0,
                                                         C.c_void_p(codes[k].data_ptr()), stream))


for k in range(args.pool):                                   # warm-up
    launch(k, rows)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
done = 0
for b in range(n_batches):
    cnt = min(B, my_samples - b * B)
    launch(b % args.pool, cnt * L)
    done += cnt * L
e1.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([wall], dtype=torch.float64, device=d)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    tot = torch.tensor([done], dtype=torch.float64, device=d)
    dist.all_reduce(tot)
    done_all = int(tot.item())
else:
    done_all = done

# properties + a sample against the f64 kernel (the exact route of tests/test_gpu_golden.py::test_posteriors_golden)
chk = min(args.check, rows) // 2 * 2
f64 = dev.calculate_posteriors_device(pool[0][:chk].to(torch.float64), alphas, False, 5.0 / 30.0)
m = ((f64 > 0).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=d)).sum(1).to(torch.uint8)
launch(0, rows)
torch.cuda.synchronize()
ok = bool(torch.equal(m[0::2] | (m[1::2] << 4), codes[0][:chk // 2]))
lo = codes[0] & 15
hist = torch.bincount(lo.to(torch.int64), minlength=16).tolist()

if rank == 0:
    gpu_s = e0.elapsed_time(e1) / 1e3
    print(json.dumps({"config": "config 4: %d samples x %d sites x 4 uint16 counts -> posterior filter -> 4-bit codes" % (args.samples, L),
                      "n_gpus": world, "site_rows": done_all, "batches_per_rank": n_batches, "batch_samples": B, "pool": args.pool,
                      "seconds": wall, "site_rows_per_s": done_all / wall, "alg_bytes_per_site_row": 8.5,
                      "GBps": done_all * 8.5 / wall / 1e9, "frac_of_hbm_peak": done_all * 8.5 / wall / 8e12 / world,
                      "rank0_kernel_seconds": gpu_s, "rank0_GBps": done * 8.5 / gpu_s / 1e9,
                      "ms_per_batch": gpu_s / n_batches * 1e3, "setup_seconds": round(setup_s, 1),
                      "check_vs_f64_kernel": ok, "mask_histogram_even_sites_batch0": hist,
                      "pcie_note": "counts resident in HBM; the same stream through PCIe is link-bound: %.0f GB at ~60 GB/s = %.0f s"
                                   % (args.samples * L * 8 / 1e9, args.samples * L * 8 / 60e9)}))
if not ok:
    raise SystemExit("codes differ from the f64 kernel's mask")
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
