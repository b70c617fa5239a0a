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


def combine(args):
    logging.basicConfig(level=args.loglevel, format="%(asctime)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    if len(args.directories) == 1:                             # a single argument is a file listing the directories (:111-113)
        with open(args.directories[0], "r") as infile:
            args.directories = [line.strip() for line in infile.readlines()]
    for directory in args.directories:
        if not os.path.isdir(directory):
            logging.error("ERROR: {} is not a directory".format(directory))
            sys.exit(1)
    if not os.path.exists(args.output_dir):
        os.mkdir(args.output_dir)
    args.output_dir = os.path.join(args.output_dir, "")

    alignments = defaultdict(list)                             # by reference genome (:127-132)
    for directory in args.directories:
        sample = os.path.basename(os.path.normpath(directory))
        for aln in glob.iglob(os.path.join(directory, "*posterior_counts_ref_*.fasta*")):
            alignments[find_ref(aln)].append((sample, aln))

    ncovs = {}
    for ref, alns in alignments.items():
        ncovs.update(write_alignment(ref, alns, args.output_dir, n_threads=args.n_cpu))

    # sourmash hits -> combined_metadata.csv; the coverage columns are "NA" in the reference too (its coverage pass is
    # commented out, tracs/combine.py:141-163)
    with open(args.output_dir + "combined_metadata.csv", "w") as outfile:
        outfile.write("sample,accession,intersect_bp,f_orig_query,f_match,f_unique_to_query,coverage,mean_depth,"
                      "mean_nonzero_depth,frac_N,species\n")
        for directory in args.directories:
            sample = os.path.basename(os.path.normpath(directory))
            for hits in glob.iglob(os.path.join(directory, "*_sourmash_hits.csv")):
                with open(hits, "r") as infile:
                    next(infile)
                    for line in infile:
                        f = line.strip().split(",")
                        accession = f[9].split()[0].strip('"')
                        species = f[9].replace(accession, "").replace('"', "").strip()
                        frac_n = str(ncovs[(sample, accession)][0]) if (sample, accession) in ncovs else "NA"
                        outfile.write(",".join([sample, accession] + f[:4] + ["NA", "NA", "NA", frac_n, species]) + "\n")
    return


def main():
    parser = argparse.ArgumentParser()
    parser = combine_parser(parser)
    args = parser.parse_args()
    args.func(args)
    return


if __name__ == "__main__":
    main()
