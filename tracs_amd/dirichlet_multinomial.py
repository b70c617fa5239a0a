"""find_dirichlet_priors -- fit of the Dirichlet-multinomial alphas, on the GPU.

Same signature and return value as /root/reference/tracs/dirichlet_multinomial.py:9 (numpy array of K alphas sorted
descending; (0,..,0,1) when there are at most 5 polymorphic sites; prints "Calculated alphas: ").
"""
import ctypes as C

import numpy as np

from . import _lib


def find_dirichlet_priors(counts, max_iter=1000, tol=1e-5, method="FPI", error_filt_threshold=None):
    c = np.ascontiguousarray(counts, dtype=np.float64)
    if c.ndim != 2:
        raise ValueError("find_dirichlet_priors(): counts must be 2-D [sites, alleles]")
    K = c.shape[1]
    out = np.zeros(K)
    iters = C.c_int(0)
    L = _lib.require_gpu()
    dp = C.POINTER(C.c_double)
    _lib.check(L.tracs_find_dirichlet_priors(c.ctypes.data_as(dp), c.shape[0], K, int(max_iter), float(tol),
                                             1 if method == "LOO" else 0,
                                             -1.0 if error_filt_threshold is None else float(error_filt_threshold),
                                             out.ctypes.data_as(dp), C.byref(iters)))
    if not (out[:-1] == 0).all() or out[-1] != 1.0:
        print("Calculated alphas: ", out)          # the reference returns before printing in the degenerate case (:26-29)
    return out
