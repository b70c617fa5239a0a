"""numpy restatement of csrc/exchange.hip's two kernels (tracs_tri_pack / tracs_tri_sum) over raw buffer addresses: the checker of
the GPU kernels (tests/test_gpu_exchange.py) and the stand-in for them in the CPU gloo tests (tests/test_multirank_gloo.py), where
partition.TriExchange's layout and protocol run for real.  Test infrastructure only."""
import ctypes

import numpy as np


def _view(ptr, count, width):
    ct = ctypes.c_uint16 if width == 2 else ctypes.c_uint32
    return np.ctypeslib.as_array((ct * count).from_address(ptr))


def tri_pack(mat, n, rb, re, cb, slots, width, base, negate, packed_ptr, packed_elems, stats, base_row=0):
    """mat: uint32 numpy [rows, ld] holding rows base_row..; slots: int64 per row; stats: numpy uint32[2] (max, overflow count)."""
    out = _view(packed_ptr, packed_elems, width) if packed_ptr else None
    for i in range(rb, re):
        s = int(slots[i - rb])
        if s < 0:
            continue
        jb = max(cb, i + 1)
        if jb >= n:
            continue
        v = mat[i - base_row, jb:n].astype(np.uint32)
        if negate:
            v = (np.uint32(base) - v).astype(np.uint32)
        if len(v):
            stats[0] = max(int(stats[0]), int(v.max()))
            if width == 2:
                stats[1] += int((v > 0xFFFF).sum())
            if out is not None:
                out[s:s + len(v)] = v.astype(out.dtype)


def tri_sum(mat, n, rb, re, cb, slots, width, recv_ptr, recv_elems, block_elems, n_blocks, skip_block, add, negate, base_row=0):
    src = _view(recv_ptr, recv_elems, width)
    for i in range(rb, re):
        s = int(slots[i - rb])
        if s < 0:
            continue
        jb = max(cb, i + 1)
        if jb >= n:
            continue
        acc = np.zeros(n - jb, dtype=np.uint32)
        for b in range(n_blocks):
            if b != skip_block:
                acc += src[b * block_elems + s:b * block_elems + s + (n - jb)].astype(np.uint32)
        row = mat[i - base_row, jb:n].astype(np.uint32)
        mat[i - base_row, jb:n] = (row + np.uint32(add & 0xFFFFFFFF) + (np.uint32(0) - acc if negate else acc)).astype(np.uint32)
