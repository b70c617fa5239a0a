"""`tracs combine` -- gather per-sample `tracs align` outputs into one multi-FASTA per reference genome.

Same command line, file discovery, output names and file contents as /root/reference/tracs/combine.py:15-239
(`<ref>_combined.fasta.gz` is the input of `tracs distance`); the FASTA reading, N counting and gzip compression run in
libtracs_hip.so's host code, one gzip member per sample, compressed in parallel (the reference compresses serially
through Python's gzip module).
"""
import argparse
import ctypes as C
import glob
import logging
import os
import re
import sys
from collections import defaultdict

from . import _lib


def combine_parser(parser):
    parser.description = "Combine runs of TRACS'm align ready for distance estimation"
    io_opts = parser.add_argument_group("Input/output")
    io_opts.add_argument("-i", "--input", dest="directories", required=True,
                         help="Paths to each directory containing the output of the TRACS align function",
                         type=os.path.abspath, nargs="+")
    io_opts.add_argument("-o", "--output", dest="output_dir", required=True,
                         help="name of the output driectory to store the combined alignments.", type=str)
    parser.add_argument("-t", "--threads", dest="n_cpu", help="number of threads to use (default=1)", type=int, default=1)
    parser.add_argument("--loglevel", type=str.upper, choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"],
                        default="INFO", help="Set the logging threshold.")
    parser.set_defaults(func=combine)
    return parser


def find_ref(filename):
    """reference genome id from '<prefix>_posterior_counts_ref_<ref>.fasta[.gz]' (tracs/combine.py:61-73)."""
    m = re.search(r"posterior_counts_ref_(.+)\.fasta", filename)
    if not m:
        logging.error("ERROR: {} is not the expected output of TRACS align".format(filename))
        sys.exit(1)
    return m.group(1)


def write_alignment(ref, alns, output_dir, n_threads=0, gzip_level=6):
    """alns: [(sample, fasta_path)] -> writes output_dir + ref + '_combined.fasta.gz'; returns the reference's ncov dict
    {(sample, ref): (fraction of N, length)} (tracs/combine.py:220-239)."""
    L = _lib.load()
    output_file = output_dir + ref + "_combined.fasta.gz"
    logging.info("Writing combined alignment for {} to {}".format(ref, output_file))
    n = len(alns)
    names = (C.c_char_p * n)(*[a[0].encode() for a in alns])
    paths = (C.c_char_p * n)(*[os.fsencode(a[1]) for a in alns])
    frac = (C.c_double * n)()
    lens = (C.c_uint64 * n)()
    rc = L.tracs_combine_fasta(os.fsencode(output_file), names, paths, n, int(n_threads), int(gzip_level), frac, lens)
    if rc:
        msg = L.tracs_last_error().decode("utf-8", "replace")
        if "contains more than one sequence" in msg:          # tracs/combine.py:233-237
            logging.error(msg)
            sys.exit(1)
        _lib.check(rc)
    return {(alns[i][0], ref): (frac[i], int(lens[i])) for i in range(n) if frac[i] >= 0.0}


METADATA_HEADER = ("sample", "accession", "intersect_bp", "f_orig_query", "f_match", "f_unique_to_query", "coverage",
                   "mean_depth", "mean_nonzero_depth", "frac_N", "species")


def _sample_name(directory):
    return os.path.basename(os.path.normpath(directory))


def _resolve_directories(arg):
    """-i takes the directories themselves, or ONE file listing them one per line (tracs/combine.py:111-113)."""
    dirs = list(arg)
    if len(dirs) == 1:
        with open(dirs[0], "r") as fh:
            dirs = [ln.strip() for ln in fh]
    missing = [d for d in dirs if not os.path.isdir(d)]
    if missing:
        logging.error("ERROR: {} is not a directory".format(missing[0]))
        sys.exit(1)
    return dirs


def _alignments_by_reference(dirs):
    """{reference id: [(sample, fasta path), ...]} over every `*posterior_counts_ref_*.fasta*` of every directory."""
    by_ref = defaultdict(list)
    for d in dirs:
        for path in glob.iglob(os.path.join(d, "*posterior_counts_ref_*.fasta*")):
            by_ref[find_ref(path)].append((_sample_name(d), path))
    return by_ref


def _hit_rows(hits_csv):
    """(accession, species, first four columns) per data line of a `*_sourmash_hits.csv`; column 9 is '"<accession> <species>"'."""
    with open(hits_csv, "r") as fh:
        fh.readline()
        for ln in fh:
            cols = ln.strip().split(",")
            label = cols[9]
            accession = label.split()[0].strip('"')
            yield accession, label.replace(accession, "").replace('"', "").strip(), cols[:4]


def _write_metadata(path, dirs, frac_n):
    """combined_metadata.csv: one row per sourmash hit; the three coverage columns are "NA" (the reference's coverage pass is
    commented out, tracs/combine.py:141-163), frac_N comes from the combined alignments."""
    with open(path, "w") as out:
        out.write(",".join(METADATA_HEADER) + "\n")
        for d in dirs:
            sample = _sample_name(d)
            for hits in glob.iglob(os.path.join(d, "*_sourmash_hits.csv")):
                for accession, species, head in _hit_rows(hits):
                    known = frac_n.get((sample, accession))
                    out.write(",".join([sample, accession, *head, "NA", "NA", "NA", "NA" if known is None else str(known[0]), species]) + "\n")


def combine(args):
    logging.basicConfig(level=args.loglevel, format="%(asctime)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    dirs = _resolve_directories(args.directories)
    os.makedirs(args.output_dir, exist_ok=True)
    prefix = os.path.join(args.output_dir, "")
    frac_n = {}
    for ref, alns in _alignments_by_reference(dirs).items():
        frac_n.update(write_alignment(ref, alns, prefix, n_threads=args.n_cpu))
    _write_metadata(prefix + "combined_metadata.csv", dirs, frac_n)


def main():
    args = combine_parser(argparse.ArgumentParser()).parse_args()
    args.func(args)


if __name__ == "__main__":
    main()
