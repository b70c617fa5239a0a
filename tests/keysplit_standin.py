"""numpy stand-ins for the HIP kernels behind partition.KeySplit (csrc/transcluster.hip tracs_trans_keys_*): the same bitmap format
(bit b of word w = grid index 32 w + b = N x (span of the days + 1) + day gap; four trailer words), the same numbering of the keys,
the same slot of every key in the compact arrays.  CPU tests run KeySplit's protocol over gloo with these (the oracle evaluates the
keys); tests/test_gpu_keysplit.py holds the HIP kernels against them word for word."""
import numpy as np

GRID_BITS = 1 << 24
WORDS = GRID_BITS // 32
TRAILER = 4


def words():
    return WORDS + TRAILER


def _rows(ranges, n):
    for r0, r1 in ranges:
        for i in range(min(r0, n), min(r1, n)):
            yield i


def mark(dist, n, days, ranges, keys, thr=2147483647, col_begin=0):
    """dist: uint32 / int32 [rows, >= n] numpy; days: int32 [n]; keys: uint32 [words()] (overwritten)"""
    keys[:] = 0
    keys[WORDS + 1] = 0xFFFFFFFF
    if n == 0:
        return
    d64 = days.astype(np.int64)
    lo, hi = int(d64.min()) + (1 << 31), int(d64.max()) + (1 << 31)
    keys[WORDS + 1], keys[WORDS + 2] = lo, hi
    stride = hi - lo + 1
    bits = np.zeros(GRID_BITS, dtype=np.uint8)
    top, beyond = 0, False
    for i in _rows(ranges, n):
        j = np.arange(max(i + 1, col_begin), n)
        if not len(j):
            continue
        d = dist[i, j].astype(np.int64) & 0xFFFFFFFF
        ok = d <= thr
        if not ok.any():
            continue
        d, j = d[ok], j[ok]
        top = max(top, int(d.max()))
        key = d * stride + np.abs(d64[i] - d64[j])
        inside = key < GRID_BITS
        beyond = beyond or not bool(inside.all())
        bits[key[inside]] = 1
    keys[:WORDS] = np.packbits(bits, bitorder="little").view(np.uint32)
    keys[WORDS] = top
    keys[WORDS + 3] = 1 if beyond else 0


def merge(keys, gathered, parts):
    g = gathered.reshape(parts, WORDS + TRAILER)
    keys[:WORDS] = np.bitwise_or.reduce(g[:, :WORDS], axis=0)
    keys[WORDS] = g[:, WORDS].max()
    keys[WORDS + 1] = g[:, WORDS + 1].min()
    keys[WORDS + 2] = g[:, WORDS + 2].max()
    keys[WORDS + 3] = np.bitwise_or.reduce(g[:, WORDS + 3])


def indices(keys):
    """grid indices of the marked keys, ascending: position = the key's ordinal"""
    return np.flatnonzero(np.unpackbits(keys[:WORDS].view(np.uint8), bitorder="little"))


def info(keys):
    nk = int(np.unpackbits(keys[:WORDS].view(np.uint8)).sum())
    top, lo, hi, bad = (int(x) for x in keys[WORDS:WORDS + 4])
    days_ok = hi >= lo
    span = hi - lo if days_ok else 0
    fits = days_ok and not bad and (top + 1) * (span + 1) <= GRID_BITS
    return nk, top, span, 1 if fits else 0


def evaluate(keys, inf, part, parts, lamb, beta, thr_ek, vals, trans_dist):
    """vals: float64 [per, 2]; trans_dist(N int32[], delta f64[], lamb, beta, thr) -> (log p0, E(K)): the oracle's"""
    idx = indices(keys)[part::parts]
    stride = inf[2] + 1
    N = (idx // stride).astype(np.int32)
    delta = ((idx % stride) * 86400).astype(np.float64) / 31556952.0        # tracs/transcluster.py:26-33: whole days in seconds
    p0, ek = trans_dist(N, delta, lamb, beta, thr_ek)
    vals[:len(idx), 0] = p0
    vals[:len(idx), 1] = ek


def gather(dist, n, days, ranges, keys, inf, vals_all, parts, exp_p0, p0, eK, thr=2147483647, col_begin=0):
    """vals_all: float64 [parts, per, 2]"""
    idx = indices(keys)
    stride = inf[2] + 1
    o = np.arange(len(idx))
    vals = vals_all[o % parts, o // parts].copy()          # by ordinal = by position in idx
    if exp_p0:
        vals[:, 0] = np.exp(vals[:, 0])
    d64 = days.astype(np.int64)
    for i in _rows(ranges, n):
        j = np.arange(max(i + 1, col_begin), n)
        if not len(j):
            continue
        d = dist[i, j].astype(np.int64) & 0xFFFFFFFF
        ok = d <= thr
        key = d[ok] * stride + np.abs(d64[i] - d64[j[ok]])
        at = np.searchsorted(idx, key)
        assert np.array_equal(idx[at], key)                    # every cell's key was marked
        p0[i, j[ok]] = vals[at, 0]
        eK[i, j[ok]] = vals[at, 1]
