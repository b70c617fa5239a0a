"""The compact exchange of the site-sharded runs on the GPU: csrc/exchange.hip's two kernels against their numpy restatement
(tests/tri_standin.py), byte for byte, and the N > 1 bench line launched the way the driver launches it: `python bench.py --gpus 2`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n,rb,re,cb,wd,wn", [(3, 301, 0, 301, 0, 2, 2), (2, 200, 64, 199, 0, 2, 4), (4, 333, 0, 120, 120, 4, 2), (8, 70, 0, 70, 0, 4, 4)])
def test_tri_pack_and_sum_equal_numpy(world, n, rb, re, cb, wd, wn):
    """`world` partial d / nn matrices in one process: every "rank" packs with tracs_tri_pack, the blocks are shuffled as the
    all-to-all would, every rank sums with tracs_tri_sum -- the packed bytes, the statistics and the summed rows equal the numpy
    restatement's, and the own rows equal the plain sums (d) and L - sum of deficits (nn).  Panels that start at a row > 0, two-file
    geometry (col_begin), mixed widths, values that overflow 16 bits (counted, not hidden)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tri_standin
    from tracs_amd import device as dev
    from tracs_amd import partition
    rng = np.random.default_rng(world * 1000 + n)
    cuda = torch.device("cuda", 0)
    Ls = rng.integers(50000, 90000, size=world)
    d_parts = [rng.integers(0, 40000 if wd == 2 else 200000, size=(re - rb, n)).astype(np.uint32) for _ in range(world)]
    n_parts = [(Ls[r] - rng.integers(0, 30000 if wn == 2 else 48000, size=(re - rb, n))).astype(np.uint32) for r in range(world)]
    if wd == 2:
        d_parts[0][3, n - 1] = 70000                                   # one value beyond 16 bits: counted in stats[1]
    cs, owner, off, be = partition.tri_layout(rb, re, n, cb, world)
    block_bytes = (wd + wn) * be
    sd, sn = block_bytes // wd, block_bytes // wn
    sends, sends_np, stats_all = [], [], []
    for r in range(world):
        slot_d = np.where(owner == r, -1, owner * sd + off).astype(np.int64)
        slot_n = np.where(owner == r, -1, owner * sn + off).astype(np.int64)
        send = torch.zeros(world * block_bytes, dtype=torch.uint8, device=cuda)
        stats = torch.zeros((2, 2), dtype=torch.int32, device=cuda)
        dm = torch.from_numpy(d_parts[r].view(np.int32)).to(cuda)
        nm = torch.from_numpy(n_parts[r].view(np.int32)).to(cuda)
        dev.tri_pack(dm, n, rb, re, cb, torch.from_numpy(slot_d).to(cuda), wd, 0, 0, send.data_ptr(), stats[0], base_row=rb)
        dev.tri_pack(nm, n, rb, re, cb, torch.from_numpy(slot_n).to(cuda), wn, int(Ls[r]), 1, send.data_ptr() + wd * be, stats[1], base_row=rb)
        ref = np.zeros(world * block_bytes, dtype=np.uint8)
        st = np.zeros((2, 2), dtype=np.uint32)
        tri_standin.tri_pack(d_parts[r], n, rb, re, cb, slot_d, wd, 0, 0, ref.ctypes.data, world * sd, st[0], base_row=rb)
        tri_standin.tri_pack(n_parts[r], n, rb, re, cb, slot_n, wn, int(Ls[r]), 1, ref.ctypes.data + wd * be, world * sn, st[1], base_row=rb)
        torch.cuda.synchronize()
        assert np.array_equal(send.cpu().numpy(), ref), "packed bytes differ (rank %d)" % r
        assert np.array_equal(stats.cpu().numpy().view(np.uint32), st), (stats, st)
        sends.append(send); sends_np.append(ref); stats_all.append(st)
    assert int(stats_all[0][0, 1]) == (1 if wd == 2 and owner[3] != 0 and n - 1 >= max(cb, rb + 3 + 1) else 0)
    L_total = int(Ls.sum())
    for q in range(world):
        recv = torch.zeros(world * block_bytes, dtype=torch.uint8, device=cuda)
        for p in range(world):                                         # block q of rank p's send -> block p of rank q's recv
            recv[p * block_bytes:(p + 1) * block_bytes] = sends[p][q * block_bytes:(q + 1) * block_bytes]
        recv_slot = np.where(owner == q, off, -1).astype(np.int64)
        dm = torch.from_numpy(d_parts[q].view(np.int32)).to(cuda)
        nm = torch.from_numpy(n_parts[q].view(np.int32)).to(cuda)
        rs = torch.from_numpy(recv_slot).to(cuda)
        dev.tri_sum(dm, n, rb, re, cb, rs, wd, recv.data_ptr(), sd, world, q, 0, 0, base_row=rb)
        dev.tri_sum(nm, n, rb, re, cb, rs, wn, recv.data_ptr() + wd * be, sn, world, q, L_total - int(Ls[q]), 1, base_row=rb)
        torch.cuda.synchronize()
        gd, gn = dm.cpu().numpy().view(np.uint32), nm.cpu().numpy().view(np.uint32)
        for i in range(rb, re):
            jb = max(cb, i + 1)
            if jb >= n:
                continue
            if owner[i - rb] != q:                                     # not this rank's row: left as it was
                assert np.array_equal(gd[i - rb, jb:], d_parts[q][i - rb, jb:]) and np.array_equal(gn[i - rb, jb:], n_parts[q][i - rb, jb:])
                continue
            mask = np.uint32(0xFFFF) if wd == 2 else np.uint32(0xFFFFFFFF)
            want_d = d_parts[q][i - rb, jb:].copy()
            want_n = n_parts[q][i - rb, jb:].astype(np.int64) + (L_total - int(Ls[q]))
            for p in range(world):
                if p != q:
                    want_d = want_d + (d_parts[p][i - rb, jb:] & mask)           # (a value beyond its width arrives truncated: stats[1] said so)
                    want_n = want_n - (int(Ls[p]) - n_parts[p][i - rb, jb:].astype(np.int64))
            assert np.array_equal(gd[i - rb, jb:], want_d.astype(np.uint32)), (q, i)
            assert np.array_equal(gn[i - rb, jb:].astype(np.int64), want_n), (q, i)
            # on or below the diagonal / left of col_begin: untouched
            assert np.array_equal(gd[i - rb, :jb], d_parts[q][i - rb, :jb])


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


@pytest.mark.parametrize("part", ["sites", "pairs"])
def test_bench_gpus_2_launches_its_own_ranks(part):
    """`python bench.py --gpus 2` with NO launcher around it (the driver's single-command form): bench.py starts its two ranks itself
    (they share this box's GPU: gloo), and the N > 1 line is complete -- n_gpus 2 as the communicator saw it, a roofline with a number
    for the dominant kernel of rank 0's call, a cpu_baseline whose block check (rows rank 0 owns, against the oracle) has passed;
    TRACS_BENCH_VERIFY: every rank's d / nn / P / E(K) equal a single call over the whole alignment."""
    out, lines = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--samples", "1501", "--sites", "100000", "--partition", part,
                         "--cpu-seconds", "0.5"], {"TRACS_BENCH_BACKEND": "gloo", "TRACS_BENCH_VERIFY": "1"})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["ranks_seen"] == 2
    r = j["roofline"]
    assert isinstance(r["frac"], float) and 0.0 < r["frac"] <= 1.0 and r["achieved"] > 0 and r["kernel_ms"] > 0 and "kernel" in r
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "bit-equal" in c["sample"]
    assert j["config"]["distinct_keys"] > 0
    assert abs(j["value"] - 1501 * 1500 / 2 / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]
    if part == "pairs":
        assert "VERIFY gathered == single-pass: True" in out.stderr
        assert "4 bytes per cell" in j["config"]["partition"]      # d and nn both fit the 16-bit exchange at this size
    else:
        assert "VERIFY site shards == single call: True" in out.stderr
        assert "SITE shards" in j["config"]["partition"] and j["value_steady_state"] > 0
        assert j["config"]["exchange_bytes_per_cell"] == 4         # 16 bits for d, 16 for the deficit of nn
        # (P - 1) blocks of this rank's share of the upper triangle: a quarter of two full uint32 matrices, give or take the chunking
        assert j["config"]["exchange_bytes_per_rank_per_call"] <= 0.3 * 2 * 4 * 1501 * 1501 / 2
        assert j["roofline_per_pack"]["per_pack_ms"] > 0
        # transcluster's distinct keys split over the two ranks (partition.KeySplit): rank 0 evaluated half of the whole matrix's keys
        tk = j["config"]["transcluster_keys"]
        assert tk["route"] == "split" and tk["distinct_keys_whole_matrix"] == j["config"]["distinct_keys"]
        assert tk["evaluated_by_rank0"] == -(-tk["distinct_keys_whole_matrix"] // 2)
        assert tk["bytes_gathered_per_rank_per_call"] == (2 ** 24 // 32 + 4) * 4 + 16 * -(-tk["distinct_keys_whole_matrix"] // 2)


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus N under a launcher that started another number of ranks is an error, never a silent run on a different number of GPUs."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--samples", "200", "--sites", "2000"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
