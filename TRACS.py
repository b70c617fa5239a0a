"""Drop-in for the reference's pybind11 extension module `TRACS`
(/root/reference/src/python_bindings.cpp:8-25): same four functions, same keyword names, so
`from TRACS import pairsnp`, `trans_dist`, `calculate_posteriors`, `lprob_k_given_N`
(tracs/distance.py:8, tracs/transcluster.py:2, tracs/align.py:21, tests/test_llk.py:3) resolve here.
Everything runs in HIP kernels on the MI355X through tracs_amd/lib/libtracs_hip.so.
"""
from tracs_amd.api import (calculate_posteriors, lprob_k_given_N, pairsnp, trans_dist)  # noqa: F401

__doc__ = "Meta Transmission Clustering"          # m.doc(), src/python_bindings.cpp:10
__all__ = ["pairsnp", "lprob_k_given_N", "trans_dist", "calculate_posteriors"]
