"""`tracs <command>` for the commands on the GPU distance path (reference: tracs/__main__.py:15-57).

Only `distance` and `cluster` exist here; align/combine/threshold/build-db/pipe/plot are outside
the scope of this repository and are reported as such.
"""
import argparse
import sys

from . import __version__
from .cluster import cluster_parser
from .distance import distance_parser

OUT_OF_SCOPE = ["align", "combine", "threshold", "build-db", "pipe", "plot"]


def main():
    parser = argparse.ArgumentParser(prog="tracs")
    parser.add_argument("--version", action="version", version="%(prog)s (tracs_amd) " + __version__)
    sub = parser.add_subparsers(title="subcommands", dest="command")
    distance_parser(sub.add_parser("distance"))
    cluster_parser(sub.add_parser("cluster"))
    if len(sys.argv) > 1 and sys.argv[1] in OUT_OF_SCOPE:
        parser.error("'%s' is not part of the MI355X distance path; use the reference TRACS for it" % sys.argv[1])
    args = parser.parse_args()
    if not hasattr(args, "func"):
        parser.print_help()
        sys.exit(0)
    args.func(args)


if __name__ == "__main__":
    main()
