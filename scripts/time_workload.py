"""One workload of bench.py's `sensitivity` on its own: per-call and steady-state ms, classes, kernels' split, once-per-call stages.
   WORKLOAD=coverage [N=10000 SITES=5000000 PARTIAL=0] python scripts/time_workload.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench as B  # noqa: E402
from tracs_amd import _lib, device as dev, synth  # noqa: E402

n, L = int(os.environ.get("N", "10000")), int(os.environ.get("SITES", "5000000"))
wl, partial = os.environ.get("WORKLOAD", "coverage"), float(os.environ.get("PARTIAL", "0"))
lib = _lib.load()
lib.tracs_debug_pair_timing(1)
lib.tracs_debug_pack_timing(1)
aln = dev.Alignment(n, L)
synth.pack_synthetic_device(aln, seed=20241022 + 2, **B.synth_kw(partial, wl))
_, days_np = synth.dates(n, seed=20241022 + 2)
days = torch.from_numpy(days_np).cuda()
d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
nn = torch.zeros_like(d)
p = torch.zeros((n, n), dtype=torch.float64, device="cuda")
e = torch.zeros_like(p)


def one(fresh):
    if fresh:
        aln.mark_packed()
    dev.pairsnp_dense(aln, d, nn)
    dev.trans_dist_dense_ranges(d, n, days, 1e-3 * 29903, 73.0, 0.01, p, e, [(0, n)], exp_p0=True)


torch.cuda.synchronize(); t = time.perf_counter(); one(False); torch.cuda.synchronize()
print("first call %.1f ms" % ((time.perf_counter() - t) * 1e3))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record(); one(True); one(True); ev[1].record(); one(False); one(False); ev[2].record(); torch.cuda.synchronize()
print("%s: per call %.2f ms, steady %.2f ms; classes %s kernel %s nw_gram %s count_source %s" % (
    wl, ev[0].elapsed_time(ev[1]) / 2, ev[1].elapsed_time(ev[2]) / 2, aln.site_classes, aln.kernel, aln.nw_form, aln.count_source))
print("kernels ms (pair, lists, count, nn_lists):", B.pair_split_ms(lib))
print("stages:", [(k, round(v, 2)) for k, v in dev.pack_stages()])
print("list stats:", aln.list_stats, "mean d %.1f" % (float(d.sum().item()) / (n * (n - 1) / 2)))
