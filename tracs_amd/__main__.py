"""`tracs <command>` for the commands on the GPU distance path (reference: tracs/__main__.py:15-57).

`distance` and `cluster` are the hot path; `combine` and `align-post` are the data formats either side of it (SURVEY.md 8f
row 4: `align-post` is the part of `tracs align` after the pileup, tracs/align.py:444-647, with that command's option names).
Read mapping (`align` proper), threshold/build-db/pipe/plot are outside the scope of this repository and are reported as such.
"""
import argparse
import sys

from . import __version__

OUT_OF_SCOPE = ["align", "threshold", "build-db", "pipe", "plot"]


def align_post_parser(parser):
    parser.description = "The post-pileup stage of `tracs align`: pileup text -> posterior counts CSV + FASTA (on the GPU)"
    parser.add_argument("--pileup", required=True, help="`htsbox pileup -C -s 0` output (plain or gzip)")
    parser.add_argument("--reference", required=True, help="reference genome FASTA the reads were mapped to")
    parser.add_argument("--ref-id", dest="ref", required=True, help="reference genome id used in the output file names")
    parser.add_argument("-o", "--output", dest="output_dir", required=True, help="output directory")
    parser.add_argument("-p", "--prefix", dest="prefix", required=True, help="sample prefix of the output file names")
    parser.add_argument("--consensus", dest="consensus", action="store_true", default=False)
    parser.add_argument("--min-cov", dest="min_cov", default=5, type=int, help="Minimum read coverage (default=5).")
    parser.add_argument("--keep-cov-outliers", dest="keep_cov_outliers", action="store_true", default=False)
    parser.add_argument("--error-perc", dest="error_threshold", default=0.01, type=float)
    parser.add_argument("--either-strand", dest="require_both_strands", action="store_false", default=True)
    parser.add_argument("--keep-all", dest="keep_all", action="store_true", default=False)

    def run(args):
        import logging
        import os
        from .align_post import align_post
        logging.basicConfig(level="INFO", format="%(asctime)s - %(levelname)s - %(message)s")
        os.makedirs(args.output_dir, exist_ok=True)
        res = align_post(args.pileup, args.reference, args.output_dir, args.prefix, args.ref, args.min_cov,
                         args.error_threshold, args.require_both_strands, args.consensus, args.keep_all,
                         args.keep_cov_outliers, log=logging.info)
        if res["fasta"] is None:
            logging.info("Skipping reference: %s (insufficient coverage or more than 75%% N)" % args.ref)

    parser.set_defaults(func=run)
    return parser


def main():
    parser = argparse.ArgumentParser(prog="tracs")
    parser.add_argument("--version", action="version", version="%(prog)s (tracs_amd) " + __version__)
    sub = parser.add_subparsers(title="subcommands", dest="command")
    # (a command's module is imported when that command -- or the help -- is asked for: `tracs distance` on ten isolates is 0.4 s, of
    # which the interpreter and the imports are most; the cluster command's numpy is not its business)
    want = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("distance", "cluster", "combine", "align-post") else None

    def register(name, get):
        if want is None or want == name:
            get()(sub.add_parser(name))

    def _distance():
        from .distance import distance_parser
        return distance_parser

    def _cluster():
        from .cluster import cluster_parser
        return cluster_parser

    def _combine():
        from .combine import combine_parser
        return combine_parser
    register("distance", _distance)
    register("cluster", _cluster)
    register("combine", _combine)
    register("align-post", lambda: align_post_parser)
    if len(sys.argv) > 1 and sys.argv[1] in OUT_OF_SCOPE:
        parser.error("'%s' is not part of the MI355X distance path; use the reference TRACS for it" % sys.argv[1])
    args = parser.parse_args()
    if not hasattr(args, "func"):
        parser.print_help()
        sys.exit(0)
    args.func(args)


if __name__ == "__main__":
    main()
