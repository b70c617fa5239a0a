"""tracs_amd -- MI355X-native implementation of the TRACS all-pairs distance path.

    from tracs_amd import pairsnp, trans_dist, lprob_k_given_N, calculate_posteriors

or, as a drop-in for the reference extension module, `import TRACS` (repo root).
The command lines `tracs distance` / `tracs cluster` are `python -m tracs_amd distance|cluster`.
"""
__version__ = "0.1.0"

_API = ("calculate_posteriors", "connected_components", "lprob_k_given_N", "pairsnp", "pairsnp_arrays", "trans_dist", "trans_dist_arrays")
__all__ = list(_API)


def __getattr__(name):
    """The array API (tracs_amd/api.py: numpy in, numpy / lists out) is imported on first use: the command lines' own path
    (`python -m tracs_amd distance`: device-resident, ctypes only) does not pay numpy's import (~0.15 s of a 0.4 s command)."""
    if name in _API:
        from . import api
        return getattr(api, name)
    raise AttributeError("module 'tracs_amd' has no attribute %r" % name)
