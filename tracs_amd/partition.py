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
