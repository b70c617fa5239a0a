"""One process per GPU: the product-level multi-GPU path of `tracs distance` (and of distance -> threshold clustering).

north_star / SURVEY.md 8e: the N x N pair space is block-partitioned over the GPUs of one node (tracs_amd/partition.py: 2P row
chunks, fold pairing), every rank holds the whole packed alignment, there is no collective during compute; what travels are
the per-rank results -- here the thresholded COO lists (rows, cols, d, nn[, filtered d]) in row-major chunk order, i.e. exactly
the tuple src/pairsnp.hpp:451-457 returns, assembled on rank 0 -- or, for clustering, only the (i, j) of the edges that pass
the threshold (tracs/cluster.py:110-112).  P(direct) and E(K) never travel: rank 0 derives them from the gathered d and the
dates (tc gather is ~1 ms; the f64 values would be 2/3 of the bytes).

`spawn()` re-launches the current command under torch.distributed.run BEFORE anything has touched the GPU (a process that has
initialised HIP must not exec); the workers find RANK / LOCAL_RANK / WORLD_SIZE in the environment.
"""
import os
import socket
import subprocess
import sys

from . import partition


def in_worker():
    """True in a process spawn() started.  The ambient WORLD_SIZE / RANK of somebody else's torchrun or SLURM job do not
    count: a plain `tracs distance` run inside such a job stays a single-GPU run."""
    return os.environ.get("TRACS_MULTIGPU_WORKER") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ


def spawn(module, argv, gpus):
    """python -m torch.distributed.run --nproc-per-node gpus -m <module> <argv...>; returns the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(gpus)), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", module] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), TRACS_MULTIGPU_WORKER="1")
    return subprocess.run(cmd, env=env).returncode


def init():
    """-> (dist, rank, world, device).  `dist` is the exchange: the library's own RCCL entry points behind tracs_amd.rccl.RcclDist
    when every rank has its own GPU ("rccl": include/tracs_hip.h part 4 -- torch only launched the processes), torch.distributed
    over gloo when ranks share a device (smoke tests), or whatever TRACS_DIST_BACKEND says ("rccl", "nccl" = torch.distributed over
    RCCL, "gloo").  A communicator that cannot be made or fails its self-test ends the rank non-zero (no silent fallback)."""
    import torch
    import torch.distributed as dist
    from . import rccl
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % max(ndev, 1))
    device = torch.device("cuda", local % max(ndev, 1))
    backend = rccl.backend_choice(world, ndev, "TRACS_DIST_BACKEND")
    if backend == "rccl":
        # No fallback: a communicator that cannot be made, or fails its self-test, ends this rank non-zero -- the launcher
        # (torch.distributed.run) then ends the others --, so ranks never disagree about the exchange and no result is ever computed
        # over an exchange that was not checked.  TRACS_DIST_BACKEND=nccl|gloo asks for torch.distributed's explicitly.
        try:
            d = rccl.RcclDist(device)
            ok = d.self_test()
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write("tracs: rank %d: no RCCL communicator through libtracs_hip (%s); TRACS_DIST_BACKEND=nccl|gloo asks for "
                             "torch.distributed's exchange instead\n" % (rank, e))
            raise SystemExit(3)
        if not ok:
            sys.stderr.write("tracs: rank %d: the RCCL communicator failed its self-test; TRACS_DIST_BACKEND=nccl|gloo asks for "
                             "torch.distributed's exchange instead\n" % rank)
            raise SystemExit(3)
        return d, rank, world, device
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return dist, rank, world, device


class _DeviceBytes:
    """A raw device allocation as a `__cuda_array_interface__` object, so torch can view it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


def shared_alignment(paths, dist, rank, world, device):
    """One rank reads and packs the FASTA(s); the packed planes (5 L / 8 bytes per sample) reach the other ranks with an RCCL
    broadcast (31 GB at 10 000 x 5 Mbp: a fraction of a second over xGMI, against parsing 50 GB of text on every rank).
    -> Alignment with .names and .n_first on every rank."""
    import torch
    from . import device as dev
    aln = dev.Alignment.from_fasta(paths) if rank == 0 else None
    meta = [(aln.n, aln.L, aln.n_first, aln.names)] if rank == 0 else [None]
    dist.broadcast_object_list(meta, src=0)
    n, L, n_first, names = meta[0]
    if rank != 0:
        aln = dev.Alignment(n, L)
        aln.names, aln.n_first = names, n_first
    nbytes = aln.nbytes
    if nbytes:
        view = torch.as_tensor(_DeviceBytes(aln.planes_ptr(), nbytes), device=device)
        step = 1 << 30
        for o in range(0, nbytes, step):
            dist.broadcast(view[o:o + step], src=0)
        torch.cuda.synchronize()
        if rank != 0:
            aln.mark_packed()
    return aln


def site_sharded_alignment(paths, dist, rank, world, device):
    """One rank reads and packs the FASTA(s); every rank then receives ITS SLICE OF THE SITES -- the whole 128-site groups
    [groups r / P, groups (r + 1) / P), a contiguous 1 / P of the packed planes (group-major layout: csrc/common.h) -- as an
    alignment of its own.  d and the compared-sites counts are sums over sites (src/pairsnp.hpp:398-403,417-420), so a rank
    runs the whole single-GPU call -- classification, lists, walks -- on 1 / P of the sites for ALL pairs (pairs_site_sharded).
    -> Alignment (n samples x this rank's sites) with .names, .n_first and .L_total on every rank."""
    import torch
    from . import device as dev
    full = dev.Alignment.from_fasta(paths) if rank == 0 else None
    meta = [(full.n, full.L, full.n_first, full.names)] if rank == 0 else [None]
    dist.broadcast_object_list(meta, src=0)
    n, L, n_first, names = meta[0]
    groups = (L + 127) // 128
    n_pad = (n + 63) // 64 * 64
    gbytes = 5 * n_pad * 16                                   # one group of the packed planes
    cut = [groups * q // world for q in range(world + 1)]
    g0, g1 = cut[rank], cut[rank + 1]
    mine = dev.Alignment(n, max(0, min(L, g1 * 128) - g0 * 128))
    mine.names, mine.n_first, mine.L_total = names, n_first, L
    step = 1 << 30
    if rank == 0:
        src = torch.as_tensor(_DeviceBytes(full.planes_ptr(), full.nbytes), device=device)
        for q in range(world):
            lo, hi = cut[q] * gbytes, cut[q + 1] * gbytes
            if q == 0:
                if hi > lo:
                    torch.as_tensor(_DeviceBytes(mine.planes_ptr(), mine.nbytes), device=device)[:hi - lo].copy_(src[lo:hi])
            else:
                for o in range(lo, hi, step):
                    dist.send(src[o:min(hi, o + step)], dst=q)
        torch.cuda.synchronize()
        full.close()
    elif g1 > g0:
        dst = torch.as_tensor(_DeviceBytes(mine.planes_ptr(), mine.nbytes), device=device)
        nbytes = (g1 - g0) * gbytes
        for o in range(0, nbytes, step):
            dist.recv(dst[o:min(nbytes, o + step)], src=0)
        torch.cuda.synchronize()
    mine.mark_packed()
    return mine


def pairs_site_sharded(aln, i_end, j_start, dist_threshold, rank, world, dist):
    """pairsnp's output (rows, cols, d, nn: int32 device tensors, row-major) on rank 0, None elsewhere, from ranks that each hold
    a slice of the sites (site_sharded_alignment).  Row panel by row panel: every rank counts the panel's pairs over its sites,
    the partial panels are summed by the compact exchange (partition.TriExchange: the upper-triangle cells only, 16 bits per cell
    where the slice's values fit, all-to-all; a rank receives the rows it owns under the fold pairing), every rank extracts the
    pairs within the threshold from its rows, and the pieces reach rank 0 in row order."""
    import torch
    from . import device as dev
    n = aln.n
    dev_ = torch.device("cuda", torch.cuda.current_device())
    empty = [torch.empty(0, dtype=torch.int32, device=dev_) for _ in range(4)]
    if i_end <= 0 or n == 0:                                   # nothing to compare (an empty first file): pairsnp.hpp:383 runs no row
        return empty if rank == 0 else None
    R = min(panel_rows(n), (i_end + 63) // 64 * 64)            # rows of a panel
    dpan = torch.zeros((R, n), dtype=torch.int32, device=dev_)
    npan = torch.zeros_like(dpan)
    L_total = getattr(aln, "L_total", aln.L)
    out = [[] for _ in range(4)]
    for r0 in range(0, i_end, R):
        r1 = min(i_end, r0 + R)
        if aln.L > 0:
            dev.pairsnp_dense(aln, dpan, npan, row_begin=r0, row_end=r1, col_begin=j_start, base_row=r0)
        else:
            dpan.zero_(); npan.zero_()
        ex = partition.TriExchange(n, r0, r1, j_start, rank, world, dist, dev_)
        ex.decide(dpan, npan, aln.L, base_row=r0)
        ex.run(dpan, npan, aln.L, L_total, base_row=r0)
        if not ex.check():                                      # a value left the width decided for this panel: never a truncated sum
            raise RuntimeError("tracs: the compact exchange overflowed its cell width (rank %d, rows %d..%d)" % (rank, r0, r1))
        got = [[] for _ in range(4)]
        for q0, q1 in ex.own_ranges:
            g = dev.coo_from_dense(dpan, npan, n, dist_threshold, row_begin=q0, row_end=q1, col_begin=j_start, base_row=r0)
            for t in range(4):
                got[t].append(g[t])
        # the panel's pieces in row order: (first row, owner, index among the owner's ranges)
        pieces = sorted((rng[0], q, k) for q in range(world) for k, rng in enumerate(partition.own_row_ranges(r0, r1, q, world)))
        counts = torch.zeros(len(pieces), dtype=torch.int64, device=dev_)
        for p, (_, q, k) in enumerate(pieces):
            if q == rank:
                counts[p] = got[0][k].numel()
        dist.all_reduce(counts)
        counts = [int(x) for x in counts.cpu().tolist()]
        for p, (_, q, k) in enumerate(pieces):
            if counts[p] == 0:
                continue
            if q == 0:
                if rank == 0:
                    for t in range(4):
                        out[t].append(got[t][k])
            elif rank == q:
                for t in range(4):
                    dist.send(got[t][k].contiguous(), dst=0)
            elif rank == 0:
                for t in range(4):
                    buf = torch.empty(counts[p], dtype=torch.int32, device=dev_)
                    dist.recv(buf, src=q)
                    out[t].append(buf)
    if rank != 0:
        return None
    return [torch.cat(o) if o else empty[0] for o in out]


def panel_rows(n, budget_bytes=1 << 30):
    """Rows per dense panel so that one uint32 panel stays within budget_bytes (a multiple of 64, at least 64)."""
    return max(64, (budget_bytes // (4 * max(n, 1))) // 64 * 64)


def pairs_of_rank(aln, i_end, j_start, dist_threshold, rank, world, recomb_filter=False, align=64):
    """This rank's share of pairsnp's output: {chunk: (rows, cols, d, nn[, filt])} int32 device tensors, row-major inside
    each chunk.  Rows i < i_end, columns j >= max(j_start, i + 1) (one file: i_end = n, j_start = 0; two files: i_end =
    j_start = n0, src/pairsnp.hpp:348-360)."""
    import torch
    from . import device as dev
    n = aln.n
    rows_max = panel_rows(n)
    cs, _ = partition.row_chunks(i_end, world, align)
    dev_ = torch.device("cuda", torch.cuda.current_device())
    dpan = torch.empty((min(rows_max, max(cs, 64)), n), dtype=torch.int32, device=dev_)
    npan = torch.empty_like(dpan)
    parts = {}
    k = 5 if recomb_filter else 4
    own = [(c * cs, min(i_end, (c + 1) * cs)) for c in sorted(set(partition.rank_chunks(rank, world))) if c * cs < min(i_end, (c + 1) * cs)]
    # per-row structures of the site classes: this rank's rows only -- set on entry whatever the handle was used for before, and
    # lifted on the way out (a handle reused with another rank / world, or for other rows, must not keep a stale promise)
    aln.hint_rows(own if (world > 1 and 1 <= len(own) <= 2) else [])
    try:
        for c in sorted(set(partition.rank_chunks(rank, world))):
            r0c, r1c = c * cs, min(i_end, (c + 1) * cs)
            acc = [[] for _ in range(k)]
            for r0 in range(r0c, max(r0c, r1c), dpan.shape[0]):
                r1 = min(r1c, r0 + dpan.shape[0])
                dev.pairsnp_dense(aln, dpan, npan, row_begin=r0, row_end=r1, col_begin=j_start, dist_threshold=dist_threshold, base_row=r0)
                got = dev.coo_from_dense(dpan, npan, n, dist_threshold, row_begin=r0, row_end=r1, col_begin=j_start, base_row=r0)
                if recomb_filter:
                    got = list(got) + [dev.filter_recomb_pairs(aln, got[0], got[1], got[2]).clone()]
                for t in range(k):
                    acc[t].append(got[t])
            parts[c] = tuple(torch.cat(a) if a else torch.empty(0, dtype=torch.int32, device=dev_) for a in acc)
    finally:
        aln.hint_rows([])
    return parts


def edges_of_rank(dist_fn, n, days, lamb, beta, precision, column, threshold, rank, world, dist_threshold=2147483647, align=64):
    """Threshold edges of this rank's row chunks for clustering on a transmission column (`tracs cluster -D direct|expectedK`):
    dist_fn(dpan, npan, r0, r1) fills the SNP panel rows [r0, r1) (pairsnp_dense on an alignment, or a synthetic source),
    transcluster runs on the panel, and only the (i, j) with value <= threshold leave the device.
    -> {chunk: (rows, cols)} int32 device tensors."""
    import torch
    from . import device as dev
    rows_max = panel_rows(n, 1 << 29)
    cs, _ = partition.row_chunks(n, world, align)
    dev_ = torch.device("cuda", torch.cuda.current_device())
    rows_p = min(rows_max, max(cs, 64))
    dpan = torch.empty((rows_p, n), dtype=torch.int32, device=dev_)
    npan = torch.empty_like(dpan)
    ppan = torch.empty((rows_p, n), dtype=torch.float64, device=dev_)
    epan = torch.empty_like(ppan)
    parts = {}
    for c in sorted(set(partition.rank_chunks(rank, world))):
        r0c, r1c = c * cs, min(n, (c + 1) * cs)
        acc = [[], []]
        for r0 in range(r0c, max(r0c, r1c), rows_p):
            r1 = min(r1c, r0 + rows_p)
            dist_fn(dpan, npan, r0, r1)
            dev.trans_dist_dense(dpan, n, days, lamb, beta, precision, ppan, epan, exp_p0=True, dist_threshold=dist_threshold,
                                 row_begin=r0, row_end=r1, base_row=r0)
            val = epan if column == "expectedK" else ppan
            e = dev.edges_from_dense_f64(val, dpan, n, threshold, dist_threshold, row_begin=r0, row_end=r1, base_row=r0)
            acc[0].append(e[0]); acc[1].append(e[1])
        parts[c] = tuple(torch.cat(a) if a else torch.empty(0, dtype=torch.int32, device=dev_) for a in acc)
    return parts


def cluster_edges(parts, n_nodes, rank, world, dist):
    """Gather the per-rank edge lists to rank 0 and label the connected components there (SciPy's numbering,
    tracs/cluster.py:126-129).  -> (n_components, labels int32 device tensor) on rank 0, None elsewhere."""
    from . import device as dev
    got = partition.gather_coo(parts, world, rank, dist)
    if rank != 0:
        return None
    return dev.connected_components_device(got[0], got[1], n_nodes)
