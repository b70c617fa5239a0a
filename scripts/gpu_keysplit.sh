timeout 1500 python -m pytest tests/test_gpu_keysplit.py tests/test_gpu_exchange.py -x -q 2>&1 | tail -15
python - <<'PY'
# timing at bench size: 8 ranks played in one process, rank 0's share
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from tracs_amd import device as dev, partition, synth
n = 10000
aln = dev.Alignment(n, 500000)
synth.pack_synthetic_device(aln, seed=20241024, mu_sample=1e-3, p_n=0.02)     # ~ the bench's distances at a tenth of the length
dm = torch.zeros((n, n), dtype=torch.int32, device="cuda"); nm = torch.zeros_like(dm)
dev.pairsnp_dense(aln, dm, nm)
_, days = synth.dates(n, seed=20241024)
dy = torch.from_numpy(days.astype(np.int32)).cuda()
pm = torch.zeros((n, n), dtype=torch.float64, device="cuda"); em = torch.zeros_like(pm)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / reps * 1e3
for P in (1, 2, 4, 8):
    own = partition.own_row_ranges(0, n, 0, P)
    whole = t(lambda: dev.trans_dist_dense_ranges(dm, n, dy, 5.3, 6.0, 0.01, pm, em, own, exp_p0=True))
    words = dev.trans_keys_words()
    g = torch.zeros(P * words, dtype=torch.int32, device="cuda")
    for q in range(P): dev.trans_keys_mark(dm, n, dy, partition.own_row_ranges(0, n, q, P), g[q * words:(q + 1) * words])
    u = torch.empty(words, dtype=torch.int32, device="cuda")
    dev.trans_keys_merge(u, g, P); info = dev.trans_keys_info(u)
    per = -(-info[0] // P); va = torch.zeros(P * per * 2, dtype=torch.float64, device="cuda")
    for q in range(P): dev.trans_keys_evaluate(u, info, q, P, 5.3, 6.0, 0.01, va[q * per * 2:(q + 1) * per * 2])
    mark = t(lambda: dev.trans_keys_mark(dm, n, dy, own, g[:words]))
    merge = t(lambda: (dev.trans_keys_merge(u, g, P), dev.trans_keys_info(u)))
    ev = t(lambda: dev.trans_keys_evaluate(u, info, 0, P, 5.3, 6.0, 0.01, va[:per * 2]))
    ga = t(lambda: dev.trans_keys_gather(dm, n, dy, own, u, info, va, P, pm, em, exp_p0=True))
    print("P=%d keys %d: own rows whole %.3f ms | split: mark %.3f merge+info %.3f evaluate %.3f gather %.3f = %.3f ms (+ two all-gathers)" %
          (P, info[0], whole, mark, merge, ev, ga, mark + merge + ev + ga))
PY
