"""calculate_trans_prob -- host glue between the SNP pairs and trans_dist.

Mirrors /root/reference/tracs/transcluster.py:8-41: sampling dates -> seconds since 1970-01-01 ->
|dt| / 31556952.0 years (:5,26-33); trans_dist (:36); exp(p0) unless log (:38-39).
"""
from datetime import date

import numpy as np

from .api import trans_dist_arrays

SECONDS_IN_YEAR = 31556952.0


def calculate_trans_prob(sparse_snp_dist, sample_dates, K, lamb, beta, samplenames=None, log=False, precision=0.01):
    i = np.asarray(sparse_snp_dist[0], dtype=np.int64)
    j = np.asarray(sparse_snp_dist[1], dtype=np.int64)
    d = np.asarray(sparse_snp_dist[2]).astype(int)
    # the reference indexes every sample 0..max(i, j), so a missing date among them is a KeyError (:23-32)
    nsamples = int(max(i.max(), j.max()))
    reftime = date.fromisoformat("1970-01-01")
    time_array = np.array([(sample_dates[samplenames[s]][1] - reftime).total_seconds() for s in range(nsamples + 1)])
    time_diff = np.abs(time_array[i] - time_array[j]) / SECONDS_IN_YEAR
    p0, eK = trans_dist_arrays(d, time_diff, lamb, beta, precision)
    if not log:
        p0 = np.exp(p0)
    return p0, eK, time_diff
