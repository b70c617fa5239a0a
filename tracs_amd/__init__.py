"""tracs_amd -- MI355X-native implementation of the TRACS all-pairs distance path.

    from tracs_amd import pairsnp, trans_dist, lprob_k_given_N, calculate_posteriors

or, as a drop-in for the reference extension module, `import TRACS` (repo root).
The command lines `tracs distance` / `tracs cluster` are `python -m tracs_amd distance|cluster`.
"""
__version__ = "0.1.0"

from .api import (calculate_posteriors, connected_components, lprob_k_given_N, pairsnp,  # noqa: F401
                  pairsnp_arrays, trans_dist, trans_dist_arrays)
