"""`tracs cluster` -- threshold single-linkage clustering of a distance CSV, GPU connected components.

Follows /root/reference/tracs/cluster.py: flags :24-79, distance column by -D (:90-97), node ids in
first-appearance order with sampleA before sampleB (:11-21,108-109), edges kept iff value <= threshold
(:110-112), labels as scipy connected_components(directed=False) gives them (:126-129), output
`sample,cluster` in node-id order (:134-137).
"""
import argparse
import logging
import os

import numpy as np

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


def cluster(args):
    logging.basicConfig(level=args.loglevel, format="%(asctime)s - %(levelname)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S")
    col = COLUMN[args.distance]
    I, J = [], []
    count = 0
    with open(args.distance_file, "r") as fh:
        next(fh)
        for line in fh:
            f = line.strip().split(",")
            a = index_count(f[0])
            b = index_count(f[1])
            if float(f[col]) <= args.threshold:
                I.append(a)
                J.append(b)
            count += 1
    if count <= 0:
        logging.warning("No distances available! Abandoning clustering.")
        return
    names = list(_ids.keys())
    logging.info("Clustering %d samples...", len(names))
    n_components, labels = connected_components(len(names), np.asarray(I, np.int32), np.asarray(J, np.int32))
    logging.info("%d putative transmission clusters found!", n_components)
    with open(args.output_file, "w") as out:
        out.write("sample,cluster\n")
        for i, lab in enumerate(labels):
            out.write(names[i] + "," + str(int(lab)) + "\n")


def main():
    parser = cluster_parser(argparse.ArgumentParser())
    args = parser.parse_args()
    args.func(args)


if __name__ == "__main__":
    main()
