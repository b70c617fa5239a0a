"""`tracs cluster` -- threshold single-linkage clustering of a distance CSV, GPU connected components.

Follows /root/reference/tracs/cluster.py: flags :24-79, distance column by -D (:90-97), node ids in
first-appearance order with sampleA before sampleB (:11-21,108-109), edges kept iff value <= threshold
(:110-112), labels as scipy connected_components(directed=False) gives them (:126-129), output
`sample,cluster` in node-id order (:134-137).
"""
import argparse
import ctypes as C
import logging
import os

import numpy as np

from . import _lib
from .api import connected_components

COLUMN = {"snp": 3, "filter": 6, "direct": 4, "expectedK": 5}

# The reference keeps its name->id table on the function object (index_count.dict, :12-19), so ids
# persist across cluster() calls in one process.  Same here, for drop-in behaviour.
_ids = {}


def index_count(name):
    if name not in _ids:
        _ids[name] = len(_ids)
    return _ids[name]


def cluster_parser(parser):
    parser.description = "Groups samples into putative transmission clusters using single linkage clustering"
    io = parser.add_argument_group("Input/output")
    io.add_argument("-d", "--distances", dest="distance_file", required=True, type=os.path.abspath,
                    help="Pairwise distance estimates obtained from running the 'distance' function")
    io.add_argument("-o", "--output", dest="output_file", required=True, type=str,
                    help="name of the output file to store the resulting cluster assignments")
    opts = parser.add_argument_group("Cluster options")
    opts.add_argument("-c", "--threshold", dest="threshold", type=float, required=True,
                      help="Distance threshold. Samples will be grouped together if the distance between them is "
                           "below this threshold.")
    opts.add_argument("-D", "--distance", dest="distance", choices=list(COLUMN), type=str, required=True,
                      help="The type of transmission distance to use. Can be one of 'snp', 'filter', 'direct', "
                           "'expectedK'")
    parser.add_argument("--loglevel", type=str.upper, default="INFO",
                        choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"], help="Set the logging threshold.")
    parser.set_defaults(func=cluster)
    return parser


def read_edges(distance_file, col, threshold):
    """The reference's CSV scan (:100-116) in libtracs_hip.so's host code: parallel parse, ids merged in file order.
    -> (number of data lines, I, J); new names are appended to the persistent id table."""
    L = _lib.load()
    seed = list(_ids.keys())
    arr = (C.c_char_p * len(seed))(*[x.encode() for x in seed])
    h = C.c_void_p()
    rc = L.tracs_read_distance_edges(os.fsencode(distance_file), int(col), float(threshold), arr, len(seed), C.byref(h))
    if rc:
        msg = L.tracs_last_error().decode("utf-8", "replace")
        if msg.startswith("could not convert string to float"):
            raise ValueError(msg)
        if msg == "list index out of range":
            raise IndexError(msg)
        if msg == "StopIteration":
            raise StopIteration
        if msg.startswith("cannot open"):
            raise FileNotFoundError(distance_file)
        _lib.check(rc)
    try:
        for i in range(len(seed), L.tracs_edges_n_names(h)):
            index_count(L.tracs_edges_name(h, i).decode("utf-8", "replace"))
        ne = L.tracs_edges_count(h)
        I = np.ctypeslib.as_array(L.tracs_edges_i(h), shape=(ne,)).copy() if ne else np.zeros(0, np.int32)
        J = np.ctypeslib.as_array(L.tracs_edges_j(h), shape=(ne,)).copy() if ne else np.zeros(0, np.int32)
        return int(L.tracs_edges_rows(h)), I, J
    finally:
        L.tracs_edges_free(h)


def cluster(args):
    logging.basicConfig(level=args.loglevel, format="%(asctime)s - %(levelname)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S")
    count, I, J = read_edges(args.distance_file, COLUMN[args.distance], args.threshold)
    if count <= 0:
        logging.warning("No distances available! Abandoning clustering.")
        return
    names = list(_ids.keys())
    logging.info("Clustering %d samples...", len(names))
    n_components, labels = connected_components(len(names), I, J)
    logging.info("%d putative transmission clusters found!", n_components)
    with open(args.output_file, "w") as out:
        out.write("sample,cluster\n")
        out.write("".join("%s,%d\n" % (names[i], int(lab)) for i, lab in enumerate(labels)))


def main():
    parser = cluster_parser(argparse.ArgumentParser())
    args = parser.parse_args()
    args.func(args)


if __name__ == "__main__":
    main()
