"""Python face of libtracs_hip.so: the four functions of the reference's pybind11 module `TRACS`
(/root/reference/src/python_bindings.cpp:12-25) with the same names, keyword names, return shapes
and error behaviour, plus array-returning variants for callers that do not want Python lists.

All arithmetic runs in HIP kernels (tracs_amd/csrc/*.hip).  No CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _paths(fasta):
    if isinstance(fasta, (str, bytes, os.PathLike)):
        raise TypeError("pairsnp(): fasta must be a list of paths")
    return [os.fsencode(p) for p in fasta]


def pairsnp_arrays(fasta, n_threads=1, dist=2147483647, filter=False):
    """Like pairsnp() but returns numpy arrays: (rows, cols, distances, names, filt_distances, n_compared)."""
    paths = _paths(fasta)
    if len(paths) < 1 or len(paths) > 2:
        raise RuntimeError("Invalid number of fasta files!")      # src/pairsnp.hpp:340-343
    for p in paths:
        if not os.path.exists(p):
            # the reference passes a NULL gzFile on (src/pairsnp.hpp:75-76); we diagnose instead
            raise FileNotFoundError(os.fsdecode(p))
    L = _lib.require_gpu()
    arr = (C.c_char_p * len(paths))(*paths)
    h = C.c_void_p()
    _lib.check(L.tracs_pairsnp(arr, len(paths), int(n_threads), int(dist), int(bool(filter)), C.byref(h)))
    owner = _ResultOwner(L, h)
    n = L.tracs_pairsnp_len(h)
    nseq = L.tracs_pairsnp_nseq(h)

    def grab(fn):
        # a VIEW of the library's result (five arrays of 8 bytes per emitted pair: copying them was a second of a 10 000-sample
        # run); the result handle lives as long as any of the views does
        if n == 0:
            return np.zeros(0, np.uint64)
        v = np.ctypeslib.as_array(fn(h), shape=(n,)).view(_OwnedArray)
        v._tracs_owner = owner
        return v
    rows = grab(L.tracs_pairsnp_rows)
    cols = grab(L.tracs_pairsnp_cols)
    d = grab(L.tracs_pairsnp_distances)
    filt = grab(L.tracs_pairsnp_filt_distances)
    nn = grab(L.tracs_pairsnp_ncompared)
    names = [L.tracs_pairsnp_name(h, i).decode("utf-8", "replace") for i in range(nseq)]
    return rows, cols, d, names, filt, nn


class _ResultOwner:
    """Frees a tracs_pairsnp_result when the last array that views it is gone."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def __del__(self):
        try:
            if self._h:
                self._lib.tracs_pairsnp_free(self._h)
                self._h = None
        except Exception:
            pass


class _OwnedArray(np.ndarray):
    """ndarray view that keeps its owner alive (views and slices of it inherit the reference through `base`)."""
    _tracs_owner = None

    def __array_finalize__(self, obj):
        if obj is not None and getattr(obj, "_tracs_owner", None) is not None:
            self._tracs_owner = obj._tracs_owner


def pairsnp(fasta, n_threads, dist, filter):
    """pairsnp(fasta, n_threads, dist, filter) -> (rows, cols, distances, seq_names, filt_distances,
    n_compared_sites), six Python lists, row-major.  src/python_bindings.cpp:12-13."""
    r, c, d, names, f, nn = pairsnp_arrays(fasta, n_threads, dist, filter)
    return (r.tolist(), c.tolist(), d.tolist(), names, f.tolist(), nn.tolist())


def trans_dist_arrays(snpdiff, datediff, lamb, beta, threshold_Ek):
    n = np.ascontiguousarray(snpdiff, dtype=np.int32)
    d = np.ascontiguousarray(datediff, dtype=np.float64)
    if n.ndim != 1 or d.ndim != 1 or n.shape != d.shape:
        raise ValueError("trans_dist(): snpdiff and datediff must be 1-D and of equal length")
    p0 = np.empty(n.shape[0], np.float64)
    eK = np.empty(n.shape[0], np.float64)
    if n.shape[0]:
        L = _lib.require_gpu()
        _lib.check(L.tracs_trans_dist(n.ctypes.data_as(C.POINTER(C.c_int32)), d.ctypes.data_as(C.POINTER(C.c_double)),
                                      n.shape[0], float(lamb), float(beta), float(threshold_Ek),
                                      p0.ctypes.data_as(C.POINTER(C.c_double)), eK.ctypes.data_as(C.POINTER(C.c_double))))
    return p0, eK


def trans_dist(snpdiff, datediff, lamb, beta, threshold_Ek):
    """trans_dist(snpdiff, datediff, lamb, beta, threshold_Ek) -> (p0_log: list, eK: list).
    src/python_bindings.cpp:19-21 -> src/transcluster.hpp:240-287 (note the order: p0 first)."""
    p0, eK = trans_dist_arrays(snpdiff, datediff, lamb, beta, threshold_Ek)
    return (p0.tolist(), eK.tolist())


def lprob_k_given_N(N, k, delta, lamb, beta, lgamma):
    """lprob_k_given_N(N, k, delta, lamb, beta, lgamma) -> (lprob, lhs).  src/python_bindings.cpp:15-17."""
    if int(N) < 0 or int(k) < 0:
        raise TypeError("lprob_k_given_N(): N and k are unsigned (size_t)")
    lg = np.ascontiguousarray(lgamma, dtype=np.float64)
    Ns = np.array([int(N)], np.uint64)
    ks = np.array([int(k)], np.uint64)
    ds = np.array([float(delta)], np.float64)
    out = np.empty(1), np.empty(1)
    L = _lib.require_gpu()
    u64p, dp = C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    _lib.check(L.tracs_lprob_k_given_N(Ns.ctypes.data_as(u64p), ks.ctypes.data_as(u64p), ds.ctypes.data_as(dp), 1,
                                       float(lamb), float(beta), lg.ctypes.data_as(dp), lg.shape[0],
                                       out[0].ctypes.data_as(dp), out[1].ctypes.data_as(dp)))
    return (float(out[0][0]), float(out[1][0]))


def calculate_posteriors(counts, alphas, keep, threshold):
    """calculate_posteriors(counts[L,K], alphas[K], keep, threshold) -> float64[L,K].
    src/python_bindings.cpp:23-25 -> src/dmultinomial.hpp:8-86."""
    c = np.ascontiguousarray(counts, dtype=np.float64)      # py::array_t<double> casts silently
    if c.ndim != 2:
        raise ValueError("calculate_posteriors(): counts must be 2-D [sites, alleles]")
    a = np.ascontiguousarray(alphas, dtype=np.float64)
    if a.ndim != 1 or a.shape[0] != c.shape[1]:
        raise ValueError("calculate_posteriors(): len(alphas) must equal counts.shape[1]")
    out = np.empty_like(c)
    if c.shape[0]:
        L = _lib.require_gpu()
        dp = C.POINTER(C.c_double)
        _lib.check(L.tracs_calculate_posteriors(c.ctypes.data_as(dp), c.shape[0], c.shape[1], a.ctypes.data_as(dp),
                                                int(bool(keep)), float(threshold), out.ctypes.data_as(dp)))
    return out


def connected_components(n_nodes, I, J):
    """Labels of scipy.sparse.csgraph.connected_components(directed=False) (tracs/cluster.py:126-129)."""
    i = np.ascontiguousarray(I, dtype=np.int32)
    j = np.ascontiguousarray(J, dtype=np.int32)
    labels = np.empty(int(n_nodes), np.int32)
    ncomp = C.c_int32(0)
    if n_nodes:
        L = _lib.require_gpu()
        ip = C.POINTER(C.c_int32)
        _lib.check(L.tracs_connected_components(i.ctypes.data_as(ip), j.ctypes.data_as(ip), i.shape[0], int(n_nodes),
                                                labels.ctypes.data_as(ip), C.byref(ncomp)))
    return int(ncomp.value), labels
