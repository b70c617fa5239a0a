"""Row-panel partition of the N x N pair matrix over the ranks of one node, and the panel exchange.

Every (i, j) cell is independent, so the path shards with no data-path collective during compute:
the upper triangle is cut into 2*P equal row chunks and rank r owns chunk r and chunk 2P-1-r
("fold" pairing: row i has N-1-i cells, so the two chunks together are 1/P of the work).  Each rank
holds the whole packed alignment.  At the end of a pass the per-rank result panels are exchanged with
one all-gather per half per matrix (RCCL over xGMI when the tensors live on GPUs; gloo in the CPU tests).
Row ranges of a row-major matrix are contiguous, so the gather writes straight into the full matrix.
"""


def row_chunks(n, world, align=64):
    """-> (chunk_rows, n_chunks): 2*world equal chunks of `chunk_rows` rows (a multiple of `align`)."""
    nchunk = 2 * world
    cs = (n + nchunk - 1) // nchunk
    cs = (cs + align - 1) // align * align
    return cs, nchunk


def rank_chunks(rank, world):
    """The two chunk indices owned by `rank`."""
    return [rank, 2 * world - 1 - rank]


def rank_ranges(n, rank, world, align=64):
    """Row ranges [(r0, r1), ...] of `rank`, clipped to n, adjacent chunks merged (one launch when world == 1)."""
    cs, nchunk = row_chunks(n, world, align)
    out = []
    for c in sorted(rank_chunks(rank, world)):
        r0, r1 = c * cs, min(n, (c + 1) * cs)
        if r0 < r1:
            if out and out[-1][1] == r0:
                out[-1] = (out[-1][0], r1)
            else:
                out.append((r0, r1))
    return out


def pairs_in_rows(n, r0, r1):
    """Number of cells (i, j), r0 <= i < r1, i < j < n."""
    cnt = max(0, r1 - r0)
    return cnt * (n - 1) - (r0 + r1 - 1) * cnt // 2


def gather_panels(mats, n, rank, world, dist, align=64, async_op=False):
    """All-gather the row panels of every matrix in `mats` (each [rows_pad, ld], rows_pad = 2*world*chunk_rows)
    so that afterwards every rank holds all rows.  `dist` is torch.distributed.  With async_op the collectives are
    only enqueued (they run on the communicator's stream, overlapping later kernels); wait on the returned works."""
    works = []
    if world == 1:
        return works
    cs, nchunk = row_chunks(n, world, align)
    mine = rank_chunks(rank, world)
    for m in mats:
        assert m.shape[0] >= cs * nchunk, "matrix must have 2*world*chunk_rows rows"
        for half in (0, 1):
            outs = []
            for q in range(world):
                c = rank_chunks(q, world)[half]
                outs.append(m[c * cs:(c + 1) * cs])
            c = mine[half]
            w = dist.all_gather(outs, m[c * cs:(c + 1) * cs], async_op=async_op)
            if async_op:
                works.append(w)
    return works


class CompactPanels:
    """The same exchange in 16 bits per cell where the values allow it (half the bytes of the d / nn panels: at 8 ranks the
    exchange is as long as the compute it hides behind).  d travels as uint16 when every distance is below 65 536; nn travels as
    uint16 offsets from the smallest count when the counts span less than 65 536 (the compared-sites counts of an alignment
    cluster near L - 2 p_N L).  Decided once by `decide()` from the first pass's panels (an all-reduce of three scalars), int32
    whenever a matrix does not fit -- and a later pass that no longer fits is caught (`check`) rather than truncated.

    Two staging buffers of int16 per matrix and set; `post(k)` narrows this rank's rows and enqueues the all-gathers,
    `finish(k)` waits and widens the other ranks' rows into the int32 matrices."""

    def __init__(self, n, rank, world, dist, align=64):
        self.n, self.rank, self.world, self.dist, self.align = n, rank, world, dist, align
        self.cs, self.nchunk = row_chunks(n, world, align)
        self.mode = None                       # (d16, nn16, nn_base) after decide()
        self.stage = {}
        self.pending = {}

    def _mine(self, m):
        return [m[c * self.cs:(c + 1) * self.cs] for c in rank_chunks(self.rank, self.world)]

    def _valid_rows(self, c):
        return max(0, min(self.n, (c + 1) * self.cs) - c * self.cs)

    def decide(self, dmat, nmat):
        """From this rank's panels (upper-triangle cells only): d16 / nn16 / nn_base, agreed over the ranks."""
        import torch
        big = 2 ** 31 - 1
        dmax = torch.zeros((), dtype=torch.int64, device=dmat.device)
        nmin = torch.full((), big, dtype=torch.int64, device=dmat.device)
        nmax = torch.zeros((), dtype=torch.int64, device=dmat.device)
        for c in rank_chunks(self.rank, self.world):
            rows = self._valid_rows(c)
            if rows <= 0:
                continue
            r0 = c * self.cs
            upper = torch.triu(torch.ones((rows, self.n), dtype=torch.bool, device=dmat.device), diagonal=r0 + 1)
            if bool(upper.any()):
                dmax = torch.maximum(dmax, dmat[r0:r0 + rows][upper].max().to(torch.int64))
                nmin = torch.minimum(nmin, nmat[r0:r0 + rows][upper].min().to(torch.int64))
                nmax = torch.maximum(nmax, nmat[r0:r0 + rows][upper].max().to(torch.int64))
        v = torch.stack([dmax, -nmin, nmax])
        if self.world > 1:
            self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
        dmax, nmin, nmax = int(v[0]), -int(v[1]), int(v[2])
        d16 = 0 <= dmax < 65536
        nn16 = nmin <= nmax and nmax - nmin < 65536
        self.mode = (d16, nn16, nmin if nn16 else 0)
        return self.mode

    def bytes_per_cell(self):
        d16, nn16, _ = self.mode
        return (2 if d16 else 4) + (2 if nn16 else 4)

    def _buf(self, k, which, like):
        import torch
        key = (k, which)
        if key not in self.stage:
            self.stage[key] = torch.zeros((self.cs * self.nchunk, like.shape[1]), dtype=torch.int16, device=like.device)
        return self.stage[key]

    def post(self, k, dmat, nmat, async_op=True):
        """Enqueue the exchange of set k's panels; matrices that do not fit 16 bits go as they are (gather_panels)."""
        import torch
        d16, nn16, base = self.mode
        works, wide = [], []
        for which, m, narrow, off in (("d", dmat, d16, 0), ("n", nmat, nn16, base)):
            if not narrow:
                wide.append(m)
                continue
            st = self._buf(k, which, m)
            for c in rank_chunks(self.rank, self.world):
                rows = self._valid_rows(c)
                if rows > 0:
                    blk = m[c * self.cs:c * self.cs + rows]
                    # cells on or below the diagonal are never written by the kernels: whatever they hold is masked to 16 bits
                    st[c * self.cs:c * self.cs + rows] = ((blk - off) & 0xFFFF).to(torch.int16)
            # (neither RCCL nor gloo has a 16-bit integer type: the panels travel as bytes -- an all-gather only copies)
            works += gather_panels((st.view(torch.uint8),), self.n, self.rank, self.world, self.dist, self.align, async_op=async_op)
        if wide:
            works += gather_panels(tuple(wide), self.n, self.rank, self.world, self.dist, self.align, async_op=async_op)
        self.pending[k] = works
        return works

    def finish(self, k, dmat, nmat):
        """Wait for set k's exchange and widen the other ranks' rows into the int32 matrices."""
        import torch
        for w in self.pending.pop(k, []):
            if w is not None:
                w.wait()
        d16, nn16, base = self.mode
        mine = set(rank_chunks(self.rank, self.world))
        for which, m, narrow, off in (("d", dmat, d16, 0), ("n", nmat, nn16, base)):
            if not narrow:
                continue
            st = self._buf(k, which, m)
            for c in range(self.nchunk):
                rows = self._valid_rows(c)
                if c in mine or rows <= 0:
                    continue
                m[c * self.cs:c * self.cs + rows] = (st[c * self.cs:c * self.cs + rows].to(torch.int32) & 0xFFFF) + off

    def check(self, dmat, nmat):
        """True while this rank's panels still fit the decided widths (upper-triangle cells)."""
        import torch
        d16, nn16, base = self.mode
        ok = True
        for c in rank_chunks(self.rank, self.world):
            rows = self._valid_rows(c)
            if rows <= 0:
                continue
            r0 = c * self.cs
            upper = torch.triu(torch.ones((rows, self.n), dtype=torch.bool, device=dmat.device), diagonal=r0 + 1)
            if not bool(upper.any()):
                continue
            if d16:
                ok = ok and int(dmat[r0:r0 + rows][upper].max()) < 65536
            if nn16:
                v = nmat[r0:r0 + rows][upper]
                ok = ok and int(v.min()) >= base and int(v.max()) - base < 65536
        return ok


def chunk_owner(c, world):
    """Rank that owns row chunk c (0 <= c < 2 * world) under the fold pairing."""
    return c if c < world else 2 * world - 1 - c


def rank_chunk_ranges(n, rank, world, align=64):
    """[(chunk index, r0, r1), ...] of `rank`, clipped to n, empty chunks dropped, ascending."""
    cs, _ = row_chunks(n, world, align)
    out = []
    for c in sorted(rank_chunks(rank, world)):
        r0, r1 = c * cs, min(n, (c + 1) * cs)
        if r0 < r1:
            out.append((c, r0, r1))
    return out


def gather_coo(parts, world, rank, dist, dst=0):
    """Variable-length gather of per-chunk COO lists to rank `dst`, in CHUNK ORDER (= row-major order of the whole pair
    matrix, src/pairsnp.hpp:451-455): the multi-GPU form of `tracs distance` and of the thresholded edge lists that feed the
    clustering (the counts first, then the variable-length payload).

    parts: {chunk index: tuple of K equally long 1-D tensors}, one entry for EVERY chunk this rank owns (empty tensors for a
    chunk without pairs; same K, dtypes and device on every rank): rows/cols/d/nn, or the i/j of threshold edges.
    Returns on dst a list of K concatenated tensors, None elsewhere.  One all-reduce carries the 2*world counts (each chunk
    has one owner, so the sum IS the gather), then point-to-point payloads (dist.send / dist.recv: RCCL over xGMI between
    GPUs, gloo in the CPU tests)."""
    import torch
    nchunk = 2 * world
    mine = sorted(set(rank_chunks(rank, world)))
    assert sorted(parts) == mine, "gather_coo: pass one entry per owned chunk"
    proto = parts[mine[0]]
    k = len(proto)
    counts = torch.zeros(nchunk, dtype=torch.int64, device=proto[0].device)
    for c, tensors in parts.items():
        assert len(tensors) == k and len({int(t.numel()) for t in tensors}) == 1
        counts[c] = tensors[0].numel()
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = [int(x) for x in counts.cpu().tolist()]
    out = [[] for _ in range(k)]
    for c in range(nchunk):
        q = chunk_owner(c, world)
        if counts[c] == 0:
            continue
        if q == dst:
            if rank == dst:
                for t in range(k):
                    out[t].append(parts[c][t])
        elif rank == q:
            for t in range(k):
                dist.send(parts[c][t].contiguous(), dst=dst)
        elif rank == dst:
            for t in range(k):
                buf = torch.empty(counts[c], dtype=proto[t].dtype, device=proto[t].device)
                dist.recv(buf, src=q)
                out[t].append(buf)
    if rank != dst:
        return None
    return [torch.cat(out[t]) if out[t] else torch.empty(0, dtype=proto[t].dtype, device=proto[t].device) for t in range(k)]


# ---- SITE shards: the compact exchange --------------------------------------------------------------------------------------------
# Ranks that each hold a slice of the SITES compute partial d / nn matrices for ALL pairs (both are sums over sites,
# src/pairsnp.hpp:398-403,417-420) and rank q needs the sums of the rows it owns.  Rows are owned under the fold pairing (chunk q and
# chunk 2P-1-q of 2P equal chunks: equal cell counts), only the cells (i, j >= max(col_begin, i + 1)) travel, in 16 bits where a
# slice's values fit (d as it is; nn as its deficit L_slice - nn), point to point (all-to-all: every xGMI link at once), and the
# receiver sums in 32 bits (csrc/exchange.hip, tracs_alltoall).  Against the reduce-scatter of two full uint32 matrices of round 4:
# half the cells, half the bytes per cell, and P - 1 links instead of one.

def tri_layout(row_begin, row_end, n, col_begin, world, align=64):
    """The packed layout of the rows [row_begin, row_end) of an n-column pair matrix over `world` ranks.
    -> (cs, owner, off, block_elems): chunk rows; per row (numpy, index i - row_begin) the owning rank and the element offset of the
    row's first cell (column max(col_begin, i + 1)) inside the owner's block -- the rows of its lower chunk first, then of its upper
    chunk --; elements per block (the largest block, rounded up to `align`: every block has this stride)."""
    import numpy as np
    R = max(0, row_end - row_begin)
    cs = max(1, -(-R // (2 * world)))
    cs = -(-cs // align) * align
    i = np.arange(row_begin, row_begin + R, dtype=np.int64)
    cells = np.maximum(0, n - np.maximum(col_begin, i + 1))
    chunk = (i - row_begin) // cs
    owner = np.where(chunk < world, chunk, 2 * world - 1 - chunk).astype(np.int64)
    off = np.zeros(R, dtype=np.int64)
    most = 0
    for q in range(world):
        sel = np.nonzero(owner == q)[0]                   # ascending rows: the lower chunk first
        c = cells[sel]
        off[sel] = np.cumsum(c) - c
        most = max(most, int(c.sum()))
    block_elems = max(align, -(-most // align) * align)
    return cs, owner, off, block_elems


def own_row_ranges(row_begin, row_end, rank, world, align=64):
    """[(r0, r1), ..] (at most two, ascending, clipped to row_end, empty ones dropped): the rows of [row_begin, row_end) that
    `rank` owns under tri_layout's fold pairing."""
    R = max(0, row_end - row_begin)
    cs = max(1, -(-R // (2 * world)))
    cs = -(-cs // align) * align
    out = []
    for c in sorted(rank_chunks(rank, world)):
        r0, r1 = row_begin + c * cs, min(row_end, row_begin + (c + 1) * cs)
        if r0 < r1:
            if out and out[-1][1] == r0:
                out[-1] = (out[-1][0], r1)
            else:
                out.append((r0, r1))
    return out


class TriExchange:
    """The compact exchange of one panel geometry (rows [row_begin, row_end) x columns [col_begin, n)): `decide()` once per data
    (the widths: an all-reduce of two maxima), `run()` per call: pack -> all-to-all -> sum into this rank's own rows of d and nn.
    d / nn: torch.int32 device panels [rows, ld] holding rows base_row.. of the partial matrices."""

    def __init__(self, n, row_begin, row_end, col_begin, rank, world, dist, device, align=64):
        import numpy as np
        import torch
        self.n, self.rb, self.re, self.cb, self.rank, self.world, self.dist, self.device = n, row_begin, row_end, col_begin, rank, world, dist, device
        self.cs, owner, off, self.block_elems = tri_layout(row_begin, row_end, n, col_begin, world, align)
        self._owner = torch.from_numpy(owner).to(device)
        self._off = torch.from_numpy(off).to(device)
        mine = owner == rank
        self.own_ranges = own_row_ranges(row_begin, row_end, rank, world, align)
        self.own_cells = int(np.maximum(0, n - np.maximum(col_begin, np.arange(row_begin, row_end, dtype=np.int64) + 1))[mine].sum())
        self._recv_slot = torch.where(self._owner == rank, self._off, torch.full_like(self._off, -1))
        self.widths = None                       # (bytes per cell of d, of nn) after decide()
        self._slots = {}
        self._buf = {}
        self.stats = torch.zeros((2, 2), dtype=torch.int32, device=device)      # per matrix: largest value packed, values that did not fit

    # ---- the kernels (libtracs_hip.so: csrc/exchange.hip) -----------------------------------------------------------------------
    def _pack(self, mat, base_row, slots, width, base, negate, packed_ptr, stats):
        from . import device as dev
        dev.tri_pack(mat, self.n, self.rb, self.re, self.cb, slots, width, base, negate, packed_ptr, stats, base_row)

    def _sum(self, mat, base_row, slots, width, recv_ptr, block_elems, add, negate):
        from . import device as dev
        dev.tri_sum(mat, self.n, self.rb, self.re, self.cb, slots, width, recv_ptr, block_elems, self.world, self.rank, add, negate, base_row)

    def _max(self, t):
        """all-reduce MAX of a small int64 tensor"""
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t

    def _alltoall(self, send, recv, block_bytes):
        """block q of `send` -> rank q, landing as block `rank` of its `recv` (uint8 tensors of world * block_bytes)."""
        import torch
        d = self.dist
        if hasattr(d, "all_to_all_blocks"):                # tracs_amd.rccl.RcclDist: tracs_alltoall
            d.all_to_all_blocks(send, recv, block_bytes)
        elif send.is_cuda and d.get_backend() == "nccl":
            d.all_to_all_single(recv, send)
        else:                                              # gloo (ranks sharing a GPU, CPU tests): P - 1 rounds of send / recv
            # (through host tensors: gloo moves device tensors at a few tens of MB/s)
            recv[self.rank * block_bytes:(self.rank + 1) * block_bytes].copy_(send[self.rank * block_bytes:(self.rank + 1) * block_bytes])
            for k in range(1, self.world):
                to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
                out = send[to * block_bytes:(to + 1) * block_bytes].cpu().contiguous()
                w = d.isend(out, dst=to)
                got = torch.empty(block_bytes, dtype=torch.uint8)
                d.recv(got, src=frm)
                recv[frm * block_bytes:(frm + 1) * block_bytes].copy_(got)
                w.wait()

    # ---- protocol ---------------------------------------------------------------------------------------------------------------
    def _send_slots(self, stride_elems):
        import torch
        if stride_elems not in self._slots:
            s = self._owner * stride_elems + self._off
            self._slots[stride_elems] = torch.where(self._owner == self.rank, torch.full_like(s, -1), s).contiguous()
        return self._slots[stride_elems]

    def decide(self, dmat, nmat, L_own, base_row=0):
        """The widths from this call's partial matrices, agreed over the ranks: 2 bytes per cell where every rank's largest value
        (d; L_own - nn) stays below 65 536."""
        import torch
        self.stats.zero_()
        slots = self._send_slots(self.block_elems)
        self._pack(dmat, base_row, slots, 4, 0, 0, None, self.stats[0])
        self._pack(nmat, base_row, slots, 4, int(L_own), 1, None, self.stats[1])
        m = self._max((self.stats[:, 0].to(torch.int64) & 0xFFFFFFFF).clone())
        dmax, nmax = int(m[0].item()), int(m[1].item())
        self.widths = (2 if dmax < 65536 else 4, 2 if nmax < 65536 else 4)
        self.stats.zero_()
        return self.widths

    def bytes_per_cell(self):
        return self.widths[0] + self.widths[1]

    def bytes_sent_per_call(self):
        """bytes this rank hands to the other ranks per call (P - 1 blocks of both matrices)"""
        return (self.world - 1) * self.block_elems * self.bytes_per_cell()

    def run(self, dmat, nmat, L_own, L_total, base_row=0):
        """pack -> all-to-all -> sum: afterwards the rows `own_ranges` of dmat / nmat hold the sums over the ranks.  A value that no
        longer fits the decided width is counted in `stats` (see check()), never truncated silently into a result that passes."""
        import torch
        wd, wn = self.widths
        be = self.block_elems
        block_bytes = (wd + wn) * be
        key = (wd, wn)
        if key not in self._buf:
            self._buf = {key: (torch.empty(self.world * block_bytes, dtype=torch.uint8, device=self.device),
                               torch.empty(self.world * block_bytes, dtype=torch.uint8, device=self.device))}
        send, recv = self._buf[key]
        sd, sn = block_bytes // wd, block_bytes // wn          # block stride in elements of each matrix's cell type
        self._pack(dmat, base_row, self._send_slots(sd), wd, 0, 0, send.data_ptr(), self.stats[0])
        self._pack(nmat, base_row, self._send_slots(sn), wn, int(L_own), 1, send.data_ptr() + wd * be, self.stats[1])
        self._alltoall(send, recv, block_bytes)
        self._sum(dmat, base_row, self._recv_slot, wd, recv.data_ptr(), sd, 0, 0)
        self._sum(nmat, base_row, self._recv_slot, wn, recv.data_ptr() + wd * be, sn, int(L_total) - int(L_own), 1)

    def check(self):
        """True while no value of any call since decide() overflowed its width, on any rank."""
        import torch
        bad = self._max((self.stats[:, 1].to(torch.int64) & 0xFFFFFFFF).clone() * torch.tensor(
            [1 if self.widths[0] == 2 else 0, 1 if self.widths[1] == 2 else 0], dtype=torch.int64, device=self.device))
        return int(bad.sum().item()) == 0


# ---- SITE shards: transcluster over a rank's own rows, the key evaluations split over the ranks ------------------------------------
# After the compact exchange rank q holds the rows it owns of the summed d.  trans_dist memoises per (N, delta) key
# (src/transcluster.hpp:245-246,265-282) and a rank's rows see most of the matrix's distinct keys -- evaluating "its" keys is nearly
# all of them on every rank (0.9 of the 1.3 ms predicted for P = 8).  Here every distinct key of the WHOLE matrix is
# evaluated by exactly one rank: mark the own rows' keys in the (N, day gap) bitmap -> all-gather + OR (2 MB per rank) -> the keys
# numbered by their position in the union, rank r evaluates ordinals r, r + P, .. -> all-gather of the compact (log p0, E(K)) arrays
# (16 bytes per distinct key in total) -> every rank fills its table and gathers its rows (csrc/transcluster.hip, tracs_trans_keys_*).

class KeySplit:
    """`run()` per call: P / E(K) of the cells of this rank's own row ranges, bit-identical to tracs_trans_dist_dense2 on them."""

    def __init__(self, n, rank, world, dist, device):
        self.n, self.rank, self.world, self.dist, self.device = n, rank, world, dist, device
        self._bufs = {}
        self.last_info = None                    # (distinct keys of the whole matrix, largest distance, span of the days, fits)
        self.last_route = None                   # "split" | "whole" (the keys do not fit the grid: every rank evaluates its rows' keys)
        self.last_evaluated = None               # keys this rank evaluated in the last split call

    # ---- the kernels (libtracs_hip.so: csrc/transcluster.hip) -------------------------------------------------------------------
    def _words(self):
        from . import device as dev
        return dev.trans_keys_words()

    def _mark(self, dmat, days, ranges, keys, dist_threshold, col_begin):
        from . import device as dev
        dev.trans_keys_mark(dmat, self.n, days, ranges, keys, dist_threshold, col_begin)

    def _merge(self, keys, gathered):
        from . import device as dev
        dev.trans_keys_merge(keys, gathered, self.world)

    def _info(self, keys):
        from . import device as dev
        return dev.trans_keys_info(keys)

    def _evaluate(self, keys, info, lamb, beta, precision, vals):
        from . import _lib
        from . import device as dev
        dev.trans_keys_evaluate(keys, info, self.rank, self.world, lamb, beta, precision, vals)
        self.last_evaluated = int(_lib.load().tracs_debug_last_trans_dist_keys())      # the keys THIS rank evaluated

    def _gather(self, dmat, days, ranges, keys, info, vals_all, pmat, emat, exp_p0, dist_threshold, col_begin):
        from . import device as dev
        dev.trans_keys_gather(dmat, self.n, days, ranges, keys, info, vals_all, self.world, pmat, emat, exp_p0, dist_threshold, col_begin)

    def _whole(self, dmat, days, ranges, lamb, beta, precision, pmat, emat, exp_p0, dist_threshold, col_begin):
        from . import device as dev
        if ranges:
            dev.trans_dist_dense_ranges(dmat, self.n, days, lamb, beta, precision, pmat, emat, ranges, exp_p0=exp_p0,
                                        dist_threshold=dist_threshold, col_begin=col_begin)

    # ---- protocol ---------------------------------------------------------------------------------------------------------------
    def _buf(self, name, numel, dtype):
        import torch
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.dtype != dtype:
            b = torch.empty(numel, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b[:numel]

    def _all_gather(self, flat, k):
        """flat: `world` blocks of k elements, this rank's already in place -> every rank's block in place"""
        import torch
        d, w, r = self.dist, self.world, self.rank
        if w == 1:
            return
        mine = flat[r * k:(r + 1) * k]
        if hasattr(d, "all_to_all_blocks"):                # tracs_amd.rccl.RcclDist: tracs_allgather_panels, in place
            d.all_gather([flat[q * k:(q + 1) * k] for q in range(w)], mine)
        elif flat.is_cuda and d.get_backend() == "nccl":
            d.all_gather_into_tensor(flat, mine.clone())
        else:                                              # gloo (ranks sharing a GPU, CPU tests): through host tensors
            host = mine.cpu().contiguous()
            outs = [torch.empty_like(host) for _ in range(w)]
            d.all_gather(outs, host)
            for q in range(w):
                if q != r:
                    flat[q * k:(q + 1) * k].copy_(outs[q])

    def run(self, dmat, days, ranges, lamb, beta, precision, pmat, emat, exp_p0=True, dist_threshold=2147483647, col_begin=0):
        """-> True when the keys were split, False when every rank evaluated the keys of its own rows (they did not fit the grid: the
        same decision on every rank, from the merged bitmap)."""
        import torch
        words = self._words()
        gathered = self._buf("keys_all", self.world * words, torch.int32)
        keys = gathered[self.rank * words:(self.rank + 1) * words] if self.world > 1 else self._buf("keys", words, torch.int32)
        self._mark(dmat, days, ranges, keys, dist_threshold, col_begin)
        if self.world > 1:
            self._all_gather(gathered, words)
            keys = self._buf("keys", words, torch.int32)
            self._merge(keys, gathered)
        info = self._info(keys)
        self.last_info = info
        if not info[3]:
            self.last_route = "whole"
            self._whole(dmat, days, ranges, lamb, beta, precision, pmat, emat, exp_p0, dist_threshold, col_begin)
            return False
        self.last_route = "split"
        per = max(1, -(-info[0] // self.world))
        vals_all = self._buf("vals_all", self.world * per * 2, torch.float64)
        self._evaluate(keys, info, lamb, beta, precision, vals_all[self.rank * per * 2:(self.rank + 1) * per * 2])
        self._all_gather(vals_all, per * 2)
        self._gather(dmat, days, ranges, keys, info, vals_all, pmat, emat, exp_p0, dist_threshold, col_begin)
        return True

    def bytes_gathered_per_call(self):
        """bytes this rank receives per call: the other ranks' key bitmaps and their compact value arrays"""
        if self.last_info is None or self.world == 1:
            return 0
        per = max(1, -(-self.last_info[0] // self.world))
        return (self.world - 1) * (self._words() * 4 + (per * 16 if self.last_info[3] else 0))
