"""BASELINE.json's parity configs at (or near) full size, through exact oracle checks where the oracle finishes in
seconds on the GPU box's host cores and through size-independent properties otherwise."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev(hiplib):
    import torch
    assert torch.cuda.is_available()
    from tracs_amd import device
    return device


def test_config2_full_size_bit_exact(dev, oracle):
    """configs[1]: 1 000 samples x 1 Mbp, every one of the 499 500 (d, nn) pairs bit-exact vs the CPU oracle."""
    import torch
    from tracs_amd import synth
    n, L = 1000, 1000000
    seqs = synth.first_samples_host(n, L, 20241023, n, mu_lineage=1e-4, mu_sample=1e-5, p_n=0.01)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    assert aln.encoding == "consensus"
    threads = max(1, min(128, os.cpu_count() or 1))
    r, c, ed, enn = oracle.pairsnp_planes(oracle.pack(seqs), L, n_threads=threads)
    assert len(r) == 499500
    ri, ci = r.astype(np.int64), c.astype(np.int64)
    assert np.array_equal(d.cpu().numpy()[ri, ci], ed.astype(np.int32))
    assert np.array_equal(nn.cpu().numpy()[ri, ci], enn.astype(np.int32))
    # the general (5-plane) kernel on the same data gives the same matrix
    seqs[0, 0] = ord("R")
    aln.pack(seqs[:1], first=0)
    d2 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d2, None)
    assert aln.encoding == "general"
    assert bool(torch.equal(d2[1:], d[1:]))


def test_config4_shape_posterior_codes(dev, oracle):
    """configs[3] shape: per-site 4-allele uint16 counts -> posterior filter -> 4-bit codes, 10^9 site-rows in one launch (1 000
    samples' worth of 1 Mbp: 8 GB of counts; scripts/bench_config4.py streams the full 5 x 10^10).  The oracle checks the
    1 M-row block the input repeats (a strided sample of the launch: every 1000th block is that block), the rest by properties:
    equal input blocks give equal output blocks, and rotating the allele columns rotates the mask bits (alphas go by rank, not
    by column: src/dmultinomial.hpp:45-64); the f64 kernel agrees on 2 M rows."""
    import torch
    from tracs_amd import synth
    reps = 1000
    L = reps * 1_000_000
    base = synth.allele_counts(1_000_000, seed=44, depth=30, p_two=0.01)
    base[:1000] = 0                                                     # uncovered sites
    base[1000:1200, 0] = 5000                                           # totals beyond the kernel's per-total table
    counts = torch.from_numpy(base.view(np.int16)).cuda().repeat(reps, 1)
    alphas = [20.8156, 4.3818, 0.8890, 0.1]
    for thr, keep, min_cov in ((0.01, False, 0), (5.0 / 30.0, True, 5)):
        codes = dev.posterior_codes_device(counts, alphas, keep, thr, min_cov=min_cov)
        assert codes.shape[0] == L // 2
        per = codes.view(reps, -1)
        assert bool((per == per[0]).all())                              # same input block -> same output block
        post = oracle.calculate_posteriors(base.astype(np.float64), alphas, keep, thr)
        mask = ((post > 0).astype(np.uint8) * np.array([1, 2, 4, 8], np.uint8)).sum(1).astype(np.uint8)
        mask[base.astype(np.int64).sum(1) < min_cov] = 15               # tracs/align.py:613
        exp = (mask[0::2] | (mask[1::2] << 4)).astype(np.uint8)
        assert np.array_equal(per[0].cpu().numpy(), exp)
        del per
        # allele columns rotated by one (A -> C -> G -> T -> A): every nibble's bits rotate the same way
        part = counts[:100_000_000]
        rot = dev.posterior_codes_device(torch.roll(part, 1, dims=1).contiguous(), alphas, keep, thr, min_cov=min_cov)
        c0 = codes[:50_000_000]
        lo, hi = c0 & 15, c0 >> 4
        assert bool(torch.equal(rot, (((lo << 1) | (lo >> 3)) & 15) | ((((hi << 1) | (hi >> 3)) & 15) << 4)))
        del rot, part
    codes = dev.posterior_codes_device(counts, alphas, False, 0.01)
    f64 = dev.calculate_posteriors_device(counts[:2_000_000].to(torch.float64), alphas, False, 0.01)
    m2 = ((f64 > 0).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device="cuda")).sum(1).to(torch.uint8)
    assert bool(torch.equal(m2[0::2] | (m2[1::2] << 4), codes[:1_000_000]))


def test_config5_shape_transcluster_and_clustering(dev, oracle):
    """configs[4] shape: 100 000 samples; 50 M candidate pairs -> transcluster -> threshold -> single linkage."""
    import torch
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    from ek_parity import check_ek
    n_nodes, P = 100000, 50_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    I = torch.randint(0, n_nodes, (P,), generator=g, device="cuda", dtype=torch.int32)
    J = (I + 1 + torch.randint(0, 50, (P,), generator=g, device="cuda", dtype=torch.int32)) % n_nodes
    close = torch.rand(P, generator=g, device="cuda") < 0.01
    N = torch.where(close, torch.poisson(torch.full((P,), 2.0, device="cuda"), generator=g),
                    torch.clamp(torch.poisson(torch.full((P,), 60.0, device="cuda"), generator=g), max=100)).to(torch.int32)
    days = torch.randint(0, 400, (P,), generator=g, device="cuda")
    delta = days.to(torch.float64) * 86400.0 / 31556952.0
    p0, ek = dev.trans_dist_device(N, delta, 5.3, 6.0, 0.01, exp_p0=False)
    # every element equals the value of its key: check 300 random elements against the oracle
    idx = torch.randint(0, P, (300,), generator=g, device="cuda")
    Nh, dh, ph, eh = N[idx].cpu().numpy(), delta[idx].cpu().numpy(), p0[idx].cpu().numpy(), ek[idx].cpu().numpy()
    ep0, _ = oracle.trans_dist(Nh, dh, 5.3, 6.0, 0.01)
    assert np.allclose(ph, ep0, rtol=1e-6, atol=0)
    for t in range(0, 300, 6):
        check_ek(oracle, int(Nh[t]), float(dh[t]), 5.3, 6.0, 0.01, float(eh[t]))
    # consistency at full size: equal keys -> bit-equal outputs
    key = N.to(torch.int64) * 1000 + days
    order = torch.argsort(key)
    same = key[order][1:] == key[order][:-1]
    assert bool((ek[order][1:][same] == ek[order][:-1][same]).all()) and bool((p0[order][1:][same] == p0[order][:-1][same]).all())
    # threshold on E(K) and cluster; labels must be SciPy's
    keep = ek <= 3.0
    Ik, Jk = I[keep].contiguous(), J[keep].contiguous()
    assert 10_000 < Ik.numel() < P
    nc, lab = dev.connected_components_device(Ik, Jk, n_nodes)
    G = csr_matrix((np.ones(Ik.numel(), np.int8), (Ik.cpu().numpy(), Jk.cpu().numpy())), shape=(n_nodes, n_nodes))
    enc, elab = connected_components(csgraph=G, directed=False, return_labels=True)
    assert nc == enc and np.array_equal(lab.cpu().numpy(), elab)


@pytest.mark.parametrize("part", ["sites"])
def test_multirank_driver_path_on_one_gpu(part):
    """bench.py under the driver's N > 1 launcher (python -m torch.distributed.run ... bench.py --gpus 2; the form without a launcher and
    the pair partition: tests/test_gpu_exchange.py) with two gloo ranks sharing the GPU (TRACS_BENCH_VERIFY: the ranks' d / nn / P / E(K) must equal a
    single call over the whole alignment).  `sites`: every rank holds a slice of the sites and counts all pairs over it, the sums
    arrive as row panels (reduce-scatter; summed whole over gloo); `pairs`: row-panel partition, every rank holds the alignment,
    two-panel transcluster pass, async panel all-gathers in 16 bits per cell."""
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TRACS_BENCH_BACKEND="gloo", TRACS_BENCH_VERIFY="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--samples", "1501", "--sites", "100000", "--partition", part], capture_output=True, text=True, env=env,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '"n_gpus": 2' in out.stdout and '"scaling": "strong"' in out.stdout
    if part == "pairs":
        assert "VERIFY gathered == single-pass: True" in out.stderr
        assert "4 bytes per cell" in out.stdout                    # d and nn both fit the 16-bit exchange at this size
    else:
        assert "VERIFY site shards == single call: True" in out.stderr
        assert "SITE shards" in out.stdout and '"value_steady_state"' in out.stdout


def test_rccl_communicator_world_one():
    """The collectives of the N > 1 paths through the RCCL backend itself ("nccl" on ROCm), on a world of one rank: the only
    RCCL configuration a single-GPU box can run.  A communicator is created, and the calls partition.py / bench.py make --
    all_gather into row slices of a matrix (int32 panels and the byte views of the 16-bit exchange), all_reduce (SUM of a key
    table, MAX of the exchange's three scalars), send-free gather_coo -- execute on it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from tracs_amd import partition
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = torch.arange(128 * 50, dtype=torch.int32, device=dev).reshape(128, 50)
out = torch.zeros_like(m)
w = dist.all_gather([out[0:64]], m[0:64], async_op=True); w.wait()
dist.all_gather([out[64:128].view(torch.uint8)], m[64:128].view(torch.uint8))
assert bool(torch.equal(out, m))
t = torch.ones((2, 7, 9), dtype=torch.float64, device=dev)
dist.all_reduce(t); assert float(t.sum()) == 126.0
v = torch.tensor([3, -5, 9], dtype=torch.int64, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX); assert v.tolist() == [3, -5, 9]
cp = partition.CompactPanels(100, 0, 1, dist)
d = torch.triu(torch.randint(0, 60000, (128, 100), dtype=torch.int32, device=dev), 1)
nn = torch.triu(torch.randint(4000000, 4060000, (128, 100), dtype=torch.int32, device=dev), 1)
assert cp.decide(d, nn)[:2] == (True, True) and cp.check(d, nn)
got = partition.gather_coo({0: (torch.arange(4, device=dev),), 1: (torch.arange(3, device=dev),)}, 1, 0, dist)
assert got[0].tolist() == [0, 1, 2, 3, 0, 1, 2]
dist.barrier()
dist.destroy_process_group()
print("RCCL OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-c", code % (root, port)], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert out.returncode == 0 and "RCCL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_rccl_through_the_c_abi_world_one():
    """The same calls through the LIBRARY's RCCL entry points (include/tracs_hip.h part 4: tracs_comm_create,
    tracs_allgather_panels, tracs_allreduce, tracs_bcast_planes) behind tracs_amd.rccl.RcclDist, on a world of one rank with the
    communicator id carried by a store -- what bench.py --gpus N and `tracs distance --gpus N` use when every rank has a GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as tdist
sys.path.insert(0, %r)
from tracs_amd import partition, rccl, device as dev_, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist = rccl.RcclDist(dev, store=tdist.HashStore(), rank=0, world=1)
assert dist.self_test()
m = torch.arange(128 * 50, dtype=torch.int32, device=dev).reshape(128, 50)
out = torch.zeros_like(m)
w = dist.all_gather([out[0:64]], m[0:64], async_op=True); w.wait()
dist.all_gather([out[64:128].view(torch.uint8)], m[64:128].view(torch.uint8))
torch.cuda.synchronize()
assert bool(torch.equal(out, m))
t = torch.ones((2, 7, 9), dtype=torch.float64, device=dev)
dist.all_reduce(t); torch.cuda.synchronize(); assert float(t.sum()) == 126.0
v = torch.tensor([3, -5, 9], dtype=torch.int64, device=dev)
dist.all_reduce(v, op=dist.ReduceOp.MAX); torch.cuda.synchronize(); assert v.tolist() == [3, -5, 9]
cp = partition.CompactPanels(100, 0, 1, dist)
d = torch.triu(torch.randint(0, 60000, (128, 100), dtype=torch.int32, device=dev), 1)
nn = torch.triu(torch.randint(4000000, 4060000, (128, 100), dtype=torch.int32, device=dev), 1)
assert cp.decide(d, nn)[:2] == (True, True) and cp.check(d, nn)
got = partition.gather_coo({0: (torch.arange(4, device=dev),), 1: (torch.arange(3, device=dev),)}, 1, 0, dist)
assert got[0].tolist() == [0, 1, 2, 3, 0, 1, 2]
objs = [("n", 5), None]
dist.broadcast_object_list(objs, src=0); assert objs == [("n", 5), None]
# the packed planes through tracs_bcast_planes (root = this rank: the handle stays as it is), then a dense call on it
seqs = synth.alignment(70, 3000, seed=3, mu_lineage=1e-3, mu_sample=1e-3, p_n=0.02)
aln = dev_.Alignment(70, 3000); aln.pack(seqs)
dist.broadcast_planes(aln, src=0)
dm = torch.zeros((70, 70), dtype=torch.int32, device=dev); dev_.pairsnp_dense(aln, dm, None)
from oracle import oracle as O
er, ec, ed, enn = O.pairsnp_arrays(seqs)
assert np.array_equal(dm.cpu().numpy()[er.astype(np.int64), ec.astype(np.int64)], ed.astype(np.int32))
# the compact exchange of the site shards on a world of one: ranks_seen / version from RCCL itself, an all-to-all of one block, and
# TriExchange end to end (nothing travels: the rank owns every row; nn = nn + (L - L_own) - 0)
assert dist.ranks_seen() == (0, 1) and dist.rccl_version() > 20000
a = torch.arange(256, dtype=torch.uint8, device=dev); b = torch.zeros_like(a)
dist.all_to_all_blocks(a, b, 256); torch.cuda.synchronize(); assert bool(torch.equal(a, b))
ex = partition.TriExchange(70, 0, 70, 0, 0, 1, dist, dev)
nm = torch.zeros((70, 70), dtype=torch.int32, device=dev); dev_.pairsnp_dense(aln, dm, nm)
assert ex.decide(dm, nm, 3000) == (2, 2)
ex.run(dm, nm, 3000, 3000); torch.cuda.synchronize()
assert ex.check() and ex.own_ranges == [(0, 70)] and ex.bytes_sent_per_call() == 0
assert np.array_equal(dm.cpu().numpy()[er.astype(np.int64), ec.astype(np.int64)], ed.astype(np.int32))
assert np.array_equal(nm.cpu().numpy()[er.astype(np.int64), ec.astype(np.int64)], enn.astype(np.int32))
dist.barrier()
dist.destroy_process_group()
print("RCCL C ABI OK")
"""
    out = subprocess.run([sys.executable, "-c", code % root], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
    assert out.returncode == 0 and "RCCL C ABI OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("mode", ["plain", "thresholded+db", "thresholded+filter+db", "empty+db"])
def test_distance_cli_two_ranks_equals_one(mode, tmp_path):
    """`tracs distance --gpus 2` (one process per rank; two gloo ranks sharing the GPU here) writes byte for byte the CSV of the
    single-GPU run: site shards (every rank counts all pairs over its slice of the sites, the sums arrive as row panels) in the
    first two modes, the row-chunk partition of the pair matrix with --filter; an EMPTY first file against a database (no row to compare:
    the header alone, from both).  The single-GPU leg takes the array path as the ranks do (TRACS_DISTANCE_ARRAYS: the same exp() for
    P(direct); the device-resident path of the single-GPU command is compared with it, and with the reference's CSVs, in
    tests/test_gpu_golden.py::test_cli_end_to_end_vs_reference_driver)."""
    import subprocess
    from tracs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, L = 203, 30000
    seqs = synth.alignment(n, L, seed=31, mu_lineage=3e-3, mu_sample=4e-4, p_n=0.02, p_partial=0.003)
    names = ["s%03d" % i for i in range(n)]
    fa = tmp_path / "aln_combined.fasta"
    if mode == "empty+db":
        fa.write_text("")
    else:
        synth.write_fasta(str(fa), seqs[:150], names=names[:150])
    db = tmp_path / "db.fasta"
    synth.write_fasta(str(db), seqs[150:], names=names[150:])
    iso, _ = synth.dates(n, seed=31, span_days=300)
    meta = tmp_path / "dates.csv"
    meta.write_text("name,date\n" + "".join("%s,%s\n" % (a, b) for a, b in zip(names, iso)))
    extra = {"plain": [], "thresholded+db": ["-D", "200", "--msa-db", str(db), "-K", "400"],
             "thresholded+filter+db": ["-D", "200", "--filter", "--msa-db", str(db), "-K", "400"],
             "empty+db": ["--msa-db", str(db)]}[mode]
    outs = []
    for gpus in (1, 2):
        out = tmp_path / ("out%d.csv" % gpus)
        env = dict(os.environ, TRACS_DIST_BACKEND="gloo", TRACS_DISTANCE_ARRAYS="1")
        rc = subprocess.run([sys.executable, "-m", "tracs_amd", "distance", "--msa", str(fa), "--meta", str(meta), "-o", str(out),
                             "--gpus", str(gpus)] + extra, capture_output=True, text=True, cwd=root, env=env, timeout=600)
        assert rc.returncode == 0, rc.stdout[-2000:] + rc.stderr[-3000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1]
    assert outs[0].count("\n") > (1000 if mode == "plain" else 100) or (mode == "empty+db" and outs[0].count("\n") == 1)


def test_cli_device_path_over_several_row_panels(tmp_path):
    """27 000 samples: the dense matrices of `tracs distance` are cut into row panels of ~9 900 rows (1 GiB per uint32 matrix) -- three
    of them here.  The device-resident path (tracs_distance_run: transcluster and COO extraction per panel, rows in batches) writes what
    the array path writes: the same rows in the same order (text-identical but for the last bits of P(direct): exp on the device / in
    numpy).  A SNP threshold keeps the CSV small."""
    import subprocess
    from tracs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, L = 27000, 384
    seqs = synth.alignment(n, L, seed=41, mu_lineage=4e-2, mu_sample=2e-3, n_lineages=400, p_n=0.01)
    names = ["s%05d" % i for i in range(n)]
    fa, meta = tmp_path / "big_combined.fasta", tmp_path / "dates.csv"
    synth.write_fasta(str(fa), seqs, names=names)
    iso, _ = synth.dates(n, seed=41, span_days=400)
    meta.write_text("name,date\n" + "".join("%s,%s\n" % (a, b) for a, b in zip(names, iso)))
    outs = {}
    for path in ("device", "arrays"):
        out = tmp_path / ("%s.csv" % path)
        env = dict(os.environ, TRACS_DISTANCE_BATCH_ROWS="50000")
        if path == "arrays":
            env["TRACS_DISTANCE_ARRAYS"] = "1"
        rc = subprocess.run([sys.executable, "-m", "tracs_amd", "distance", "--msa", str(fa), "--meta", str(meta), "-o", str(out), "-D", "3",
                             "--loglevel", "ERROR"], capture_output=True, text=True, cwd=root, env=env, timeout=900)
        assert rc.returncode == 0, rc.stdout[-2000:] + rc.stderr[-3000:]
        outs[path] = open(out).read().split("\n")
    a, b = outs["device"], outs["arrays"]
    assert len(a) == len(b) and a[0] == b[0] and len(a) > 100000, (len(a), len(b))
    rows_seen = set()
    for x, y in zip(a[1:], b[1:]):
        if x != y:
            fx, fy = x.split(","), y.split(",")
            assert fx[:4] == fy[:4] and fx[6:] == fy[6:], (x, y)
            for c in (4, 5):
                assert abs(float(fx[c]) - float(fy[c])) <= 1e-9 * abs(float(fy[c])) + 1e-300, (x, y)
        if x:
            rows_seen.add(int(x[1:6]) // 9900)
    assert len(rows_seen) >= 3                                   # pairs from every panel


def test_general_path_beyond_one_lds_row(dev, oracle):
    """33 100 samples with partial codes: the sparse correction's row no longer fits one 32 768-column LDS chunk, the sample ids
    in the per-site lists pass 2^15, and the matrix holds 5.5 x 10^8 pairs; 80 samples spread over the whole range (and all
    their cross pairs) against the oracle, plus the thresholded pass on the same handle."""
    import torch
    from tracs_amd import synth
    n, L = 33100, 1500
    seqs = synth.alignment(n, L, seed=77, mu_lineage=0.03, mu_sample=0.01, n_lineages=40, p_n=0.02, p_partial=0.01)
    aln = dev.Alignment(n, L)
    aln.pack(seqs)
    d = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    nn = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d, nn)
    assert aln.encoding == "general" and aln.kernel == "mfma-general"
    sub = np.sort(np.concatenate([np.arange(0, 20), np.arange(16380, 16400), np.arange(32758, 32778), np.arange(n - 20, n)]))
    er, ec, ed, enn = oracle.pairsnp_arrays(seqs[sub], n_threads=8)
    t = torch.from_numpy(sub).cuda()
    dsub, nsub = d[t][:, t].cpu().numpy(), nn[t][:, t].cpu().numpy()
    li, lj = er.astype(np.int64), ec.astype(np.int64)
    assert np.array_equal(dsub[li, lj], ed.astype(np.int32)) and np.array_equal(nsub[li, lj], enn.astype(np.int32))
    thr = int(np.percentile(ed, 20))
    d2 = torch.zeros((n, n), dtype=torch.int32, device="cuda")
    dev.pairsnp_dense(aln, d2, None, dist_threshold=thr)
    d2s = d2[t][:, t].cpu().numpy()
    keep = ed <= thr
    assert keep.sum() > 100 and np.array_equal(d2s[li[keep], lj[keep]], ed[keep].astype(np.int32))
    assert ((d2s[li[~keep], lj[~keep]].astype(np.int64) > thr) | (d2s[li[~keep], lj[~keep]] < 0)).all()


def test_config5_edge_clustering_two_ranks_equals_one():
    """scripts/bench_config5.py (per-rank panels -> transcluster -> E(K) threshold edges -> edge gather -> components) with two
    gloo ranks sharing the GPU gives the components of the single-process run; the single-process run checks itself against SciPy."""
    import json
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "bench_config5.py")
    one = subprocess.run([sys.executable, script, "--samples", "6100", "--check", "6100", "--clusters", "1500"], capture_output=True, text=True, cwd=root, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r1["scipy_check"] is True and r1["edges_rank0_chunks"] > 1000 and 1 < r1["components"] < 6100
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TRACS_DIST_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), script, "--samples", "6100", "--gpus", "2", "--clusters", "1500"], capture_output=True, text=True,
                         cwd=root, env=env, timeout=600)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert r2["n_gpus"] == 2 and r2["components"] == r1["components"]


def test_bench_line_contract_small():
    """bench.py on one GPU at a small size: one JSON line with the fields the driver reads -- metric / value / unit / n_gpus /
    steps / ms_per_step / scaling / dtype / config.workload, a `roofline` object (bound, achieved, peak, unit, frac, traffic) for
    the dominant kernel with the site-class report beside it, `roofline_general`, `dm_frontend`, and a `cpu_baseline` (kind,
    cores, sample, the two legs) whose block check -- GPU d / nn bit-equal to the oracle on the timed alignment -- has passed."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--samples", "700", "--sites",
                          "200000", "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "pairs/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert abs(j["value"] - 700 * 699 / 2 / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["encoding"] == "consensus"
    # `value` is ONE CALL per step (once-per-pack work redone in every step); the repeated pass is beside it and cannot be slower;
    # the stages of the once-per-call work carry their bytes and their fraction of the HBM peak, none above 1
    assert j["value_steady_state"] >= j["value"] * 0.98 and j["ms_per_step_steady_state"] <= j["ms_per_step"] * 1.02
    assert "ONE CALL per step" in j["step"]
    rp = j["roofline_per_pack"]
    assert rp["bound"] == "hbm" and rp["per_pack_ms"] > 0 and rp["per_pack_ms"] <= j["ms_per_step"] * 1.05
    assert {"classify", "lists: per site"} <= {st["stage"] for st in rp["stages"]}
    assert all(st["ms"] >= 0 and st["read_GB"] >= 0 and st["written_GB"] >= 0 and (st["frac"] is None or st["frac"] <= 1.0) for st in rp["stages"])
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac"] <= 1.0
    sc = r["site_classes"]
    assert sc["dense"] + sc["counted"] + sc["full"] + sc["empty"] == 200000 and sc["minority"] > 0
    assert r.get("frac_of_measured_fp4_ceiling", 0.0) <= 1.0 and "hbm" not in r    # no fraction above 1 on the line
    # one pass per alignment (the reference's unit of work), cold and warm, with the once-per-pack stages
    sp = j["single_pass"]
    assert sp["cold_ms"] > 0 and sp["warm_ms"] > 0 and sp["per_pack_ms"] > 0 and "classify" in sp["stages_ms"]
    assert sp["warm_ms"] >= j["ms_per_step"] * 0.8 and abs(sp["value_single_pass"] - 700 * 699 / 2 / (sp["warm_ms"] / 1e3)) < 1e-6 * sp["value_single_pass"]
    # the other workloads: classes on / off agree (bench.py exits non-zero otherwise), the worst one is beside `value`
    sw = j["sensitivity"]["workloads"]
    assert set(sw) == {"lineage", "divergent", "clean", "gappy", "runs", "partial", "coverage"}
    # every leg on the headline's unit (ONE CALL per step: ms_per_call), the steady-state pass beside it; worst / spread from the per-call figures
    assert all(w["ms_per_call"] > 0 and w["ms_per_pass_steady_state"] > 0 and abs(w["pairs_per_s"] - 700 * 699 / 2 / (w["ms_per_call"] / 1e3)) < 1e-6 * w["pairs_per_s"]
               for w in sw.values())
    assert j["value_worst_workload"] <= j["value"] and j["value_worst_workload"] == min([j["value"]] + [w["pairs_per_s"] for w in sw.values()])
    # `tracs distance --filter` over every emitted pair of the timed alignment: the list route, checked against the oracle
    f = j["filter"]
    assert f["pairs"] == 700 * 699 // 2 and f["route"] == "departure lists" and f["oracle_check"]["equal"] is True and f["warm_call_s"] > 0
    assert f["scan_route"]["all_pairs_s"] > 0 and f["index"]["lists"] is True
    assert all(o["kernel_ms"] >= 0 for o in r["other_kernels"]) and r["minority_lists_ms"] >= 0 and len(r["kernels_ms"]) == 4
    g = j["roofline_general"]
    assert g["bound"] in ("mfma", "hbm") and g["mean_d"] > j["config"]["mean_d"] and g["roofline_per_pack"]["per_pack_ms"] > 0
    assert j["dm_frontend"]["encoding"] in ("general", "consensus")
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "pairs/s" and "sample" in c
    assert c["pairsnp_pairs_per_s"] > 0 and c["trans_dist_keys_per_s"] > 0 and "bit-equal" in c["sample"]
