"""Lean profiling target: random ACGT(+1% N) alignment packed in a few big batches, then `reps`
launches of the dense pair kernel.  Few torch kernels, so rocprofv3 --pmc passes stay short.
usage: prof_target.py <samples> <sites> [reps] [with_nn]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tracs_amd import device as dev  # noqa: E402

n, L = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
with_nn = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(1)
d = torch.device("cuda", 0)
aln = dev.Alignment(n, L)
lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=d)
base = torch.randint(0, 4, (L,), device=d, dtype=torch.uint8)
batch = max(1, min(n, (1 << 28) // L))
for s0 in range(0, n, batch):
    cnt = min(batch, n - s0)
    idx = base.unsqueeze(0).repeat(cnt, 1)
    mut = torch.rand((cnt, L), device=d) < 2e-4
    idx[mut] = (idx[mut] + 1) & 3
    idx[torch.rand((cnt, L), device=d) < 0.01] = 4
    aln.pack(lut[idx.long()], first=s0)
    del idx, mut
dm = torch.zeros((n, n), dtype=torch.int32, device=d)
nm = torch.zeros((n, n), dtype=torch.int32, device=d) if with_nn else None
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
for r in range(reps):
    ev[r].record()
    dev.pairsnp_dense(aln, dm, nm)
ev[reps].record()
torch.cuda.synchronize()
ms = [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]
pairs = n * (n - 1) // 2
ops = pairs * ((L + 127) // 128) * 4 * (7 if with_nn else 5)
print("n=%d L=%d launches(ms)=%s  best %.3f ms  %.2f Tlane-op/s (%.1f%% of 78.6)  checksum %d" %
      (n, L, ["%.2f" % m for m in ms], min(ms), ops / min(ms) / 1e9, 100 * ops / min(ms) / 1e9 / 78.6432, int(dm.sum().item())))
