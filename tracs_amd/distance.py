"""`tracs distance` -- pairwise SNP + transmission distances, GPU path.

Command line, CSV schema and row filtering follow /root/reference/tracs/distance.py:
flags :15-131, dates CSV :145-151, per-MSA pairsnp :159-176, transmission block :179-204,
CSV rows :206-258 (header :157).  The pair loop and the transcluster integral run on the MI355X.
"""
import argparse
import ctypes as C
import logging
import os
import sys
from datetime import date

from . import _lib
from .utils import check_positive_float, check_positive_int


def pairsnp_arrays(*args, **kwargs):
    """tracs_amd.api.pairsnp_arrays, imported with numpy on first use (the array route only: --gpus N, incomplete metadata, TRACS_DISTANCE_ARRAYS)"""
    from .api import pairsnp_arrays as f
    return f(*args, **kwargs)


def calculate_trans_prob(*args, **kwargs):
    from .transcluster import calculate_trans_prob as f
    return f(*args, **kwargs)

HEADER = ("sampleA,sampleB,date difference,SNP distance,transmission distance,expected K,"
          "filtered SNP distance,sites considered,MSA file\n")


def distance_parser(parser):
    parser.description = ("Estimates pairwise SNP and transmission distances between each pair of samples "
                          "aligned to the same reference genome.")
    io = parser.add_argument_group("Input/output")
    io.add_argument("--msa", dest="msa_files", required=True, type=os.path.abspath, nargs="+",
                    help="Input fasta files formatted by the align and merge functions")
    io.add_argument("--msa-db", dest="msa_db", type=os.path.abspath, default=None,
                    help="A database MSA used to compare each sequence to. By default all pairwise comparisons "
                         "within each MSA are considered.")
    io.add_argument("--meta", dest="metadata", default=None, type=os.path.abspath,
                    help="Location of metadata in csv format. The first column must include the sequence names "
                         "and the second column must include sampling dates.")
    io.add_argument("-o", "--output", dest="output_file", required=True, type=str,
                    help="name of the output file to store the pairwise distance estimates.")
    snp = parser.add_argument_group("SNP distance options")
    snp.add_argument("-D", "--snp_threshold", dest="snp_threshold", type=check_positive_int, default=2147483647,
                     help="Only output those transmission pairs with a SNP distance <= D")
    snp.add_argument("--filter", dest="recomb_filter", action="store_true", default=False,
                     help="Filter out regions with unusually high SNP distances often caused by HGT")
    tr = parser.add_argument_group("Transmission distance options")
    tr.add_argument("--clock_rate", dest="clock_rate", type=check_positive_float, default=1e-3 * 29903,
                    help="clock rate as defined in the transcluster paper (SNPs/genome/year) default=1e-3 * 29903")
    tr.add_argument("--trans_rate", dest="trans_rate", type=check_positive_float, default=73.0,
                    help="transmission rate as defined in the transcluster paper (transmissions/year) default=73")
    tr.add_argument("-K", "--trans_threshold", dest="trans_threshold", type=check_positive_int, default=None,
                    help="Only outputs those pairs where the most likely number of intermediate hosts <= K")
    tr.add_argument("--precision", dest="precision", type=check_positive_float, default=0.01,
                    help="The precision used to calculate E(K) (default=0.01).")
    parser.add_argument("-t", "--threads", dest="n_cpu", type=check_positive_int, default=1,
                        help="number of threads to use (default=1; the pair loop runs on the GPU)")
    parser.add_argument("--loglevel", type=str.upper, default="INFO",
                        choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"], help="Set the logging threshold.")
    parser.add_argument("--gpus", dest="gpus", type=check_positive_int, default=1,
                        help="number of GPUs of this node to spread the work over (default=1; one process per GPU, each with a slice "
                             "of the sites -- with --filter: with its row panels of the pair matrix --, results gathered on the "
                             "first; not in the reference)")
    parser.set_defaults(func=distance)
    return parser


def _read_dates(path):
    dates = {}
    with open(path, "r") as fh:
        next(fh)                                   # header line is skipped (:148)
        for line in fh:
            f = line.strip().split(",")
            dates[f[0]] = (f[1], date.fromisoformat(f[1]))
    return dates


def _append_rows(path, names, rows, cols, snpd, filt, ncomp, ddiff, tdist, ek, kmax, ref):
    """CSV rows in the reference's format (:206-258), formatted and written by libtracs_hip.so's host code; floats print as
    Python's str(float).  ddiff is None without metadata; filt is None for the "NA" column."""
    import numpy as np
    L = _lib.load()
    u64p, dp = C.POINTER(C.c_uint64), C.POINTER(C.c_double)

    def u64(a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return a, a.ctypes.data_as(u64p)

    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(dp)

    n = len(rows)
    keep = [u64(rows), u64(cols), u64(snpd), u64(ncomp)]
    kf = u64(filt) if filt is not None else (None, None)
    with_dates = ddiff is not None
    fl = [f64(ddiff), f64(tdist), f64(ek)] if with_dates else [(None, None)] * 3
    cnames = (C.c_char_p * len(names))(*[x.encode() for x in names])
    written = C.c_uint64(0)
    _lib.check(L.tracs_write_distance_rows(os.fsencode(path), cnames, keep[0][1], keep[1][1], keep[2][1], kf[1], keep[3][1],
                                           fl[0][1], fl[1][1], fl[2][1], n, int(with_dates),
                                           -1.0 if kmax is None else float(kmax), ref.encode(), C.byref(written)))
    return written.value


def _rows_on_device(msas, args, dates, ref, stage):
    """One alignment through libtracs_hip.so's device-resident path (tracs_distance_open / _run: include/tracs_hip.h): FASTA -> packed
    planes -> dense panels -> transcluster on the panels -> the pairs within the threshold with their P and E(K) -> ONE device-to-host
    pass, in batches -> the CSV rows, formatted and appended by the library's host threads.  Nothing comes back to Python but the
    sample names (to look the dates up).  -> False when the path does not apply: a sample without a date (the reference raises KeyError
    only if that sample's index is at most the largest index among the emitted pairs, tracs/transcluster.py:23-32: left to the
    array path below, which reproduces that)."""
    L = _lib.require_gpu()
    arr = (C.c_char_p * len(msas))(*[os.fsencode(p) for p in msas])
    h = C.c_void_p()
    _lib.check(L.tracs_distance_open(arr, len(msas), C.byref(h)))
    try:
        stage("[sum] tracs_distance_open (read FASTA, allocate, H2D + pack)")
        days = None
        if dates is not None:
            n = L.tracs_distance_nseq(h)
            epoch = date(1970, 1, 1)
            try:
                days = (C.c_int32 * max(n, 1))(*[(dates[L.tracs_distance_name(h, i).decode("utf-8", "replace")][1] - epoch).days for i in range(n)])
            except KeyError:
                return False
        written, pairs = C.c_uint64(0), C.c_uint64(0)
        if dates is not None:
            logging.info("Inferring transmission probabilities for %s", msas[0])
        kmax = -1.0 if (args.trans_threshold is None or dates is None) else float(args.trans_threshold)
        _lib.check(L.tracs_distance_run(h, int(args.snp_threshold), days, float(args.clock_rate), float(args.trans_rate), float(args.precision),
                                        kmax, os.fsencode(args.output_file), ref.encode(), int(bool(args.recomb_filter)), C.byref(written),
                                        C.byref(pairs)))
        stage("[sum] tracs_distance_run (dense panels, transcluster, rows: %d pairs, %d rows written)" % (pairs.value, written.value))
        return True
    finally:
        L.tracs_distance_free(h)


def _cli_of(args):
    """The command line that reproduces `args` (the multi-GPU path re-launches itself, one process per GPU)."""
    argv = ["distance", "--msa"] + list(args.msa_files) + ["-o", args.output_file, "-D", str(args.snp_threshold),
                                                            "--clock_rate", repr(float(args.clock_rate)), "--trans_rate", repr(float(args.trans_rate)),
                                                            "--precision", repr(float(args.precision)), "-t", str(args.n_cpu),
                                                            "--loglevel", args.loglevel, "--gpus", str(args.gpus)]
    if args.msa_db is not None:
        argv += ["--msa-db", args.msa_db]
    if args.metadata is not None:
        argv += ["--meta", args.metadata]
    if args.recomb_filter:
        argv += ["--filter"]
    if args.trans_threshold is not None:
        argv += ["-K", str(args.trans_threshold)]
    return argv


def _pairs_multi_gpu(msas, args, ctx):
    """pairsnp's six outputs for one MSA, computed by all ranks and assembled on rank 0 (None on the others)."""
    import numpy as np
    from . import device as dev
    from . import multigpu, partition
    dist, rank, world, device = ctx
    if not args.recomb_filter and os.environ.get("TRACS_DIST_PARTITION", "sites") == "sites":
        # SITE shards (the default): every rank holds 1 / P of the sites and counts all pairs over them -- every stage of the call,
        # what is built once per alignment included, works on 1 / P of the data; the sums arrive as row panels (reduce-scatter)
        aln = multigpu.site_sharded_alignment(msas, dist, rank, world, device)
        n = aln.n
        i_end, j_start = (n, 0) if len(msas) == 1 else (aln.n_first, aln.n_first)      # src/pairsnp.hpp:348-360
        got = multigpu.pairs_site_sharded(aln, i_end, j_start, args.snp_threshold, rank, world, dist)
        names = aln.names
        aln.close()
        if rank != 0:
            return None
        host = [g.cpu().numpy().astype(np.uint32).astype(np.uint64) for g in got]
        return host[0], host[1], host[2], names, np.zeros(len(host[0]), np.uint64), host[3]      # filter off: `len` zeros (:452)
    # PAIR partition (with --filter: the recombination filter wants every site of a pair in one place): every rank holds the whole
    # alignment and computes its row panels
    if os.environ.get("TRACS_DIST_PARSE_ALL"):
        aln = dev.Alignment.from_fasta(msas)              # every rank parses and packs
    else:
        aln = multigpu.shared_alignment(msas, dist, rank, world, device)      # rank 0 parses, the packed planes are broadcast
    n = aln.n
    i_end, j_start = (n, 0) if len(msas) == 1 else (aln.n_first, aln.n_first)      # src/pairsnp.hpp:348-360
    parts = multigpu.pairs_of_rank(aln, i_end, j_start, args.snp_threshold, rank, world, args.recomb_filter)
    got = partition.gather_coo(parts, world, rank, dist)
    names = aln.names
    aln.close()
    if rank != 0:
        return None
    host = [g.cpu().numpy().astype(np.uint32).astype(np.uint64) for g in got]
    filt = host[4] if args.recomb_filter else np.zeros(len(host[0]), np.uint64)     # filter off: `len` zeros (:452)
    return host[0], host[1], host[2], names, filt, host[3]


def distance(args):
    from . import multigpu
    if getattr(args, "gpus", 1) > 1 and not multigpu.in_worker():
        rc = multigpu.spawn("tracs_amd", _cli_of(args), args.gpus)     # before anything here has touched the GPU
        if rc:
            raise SystemExit(rc)
        return
    ctx = multigpu.init() if multigpu.in_worker() else None
    if ctx is None:
        # the HIP runtime, the context and the library's kernels come up beside the metadata and the FASTA read (csrc/capi.hip)
        from . import _lib
        try:
            _lib.load().tracs_warm_up()
        except Exception:                                        # (no library / no GPU: the first real call says so)
            pass
    lead = ctx is None or ctx[1] == 0
    logging.basicConfig(level=args.loglevel, format="%(asctime)s - %(levelname)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S")
    logging.info("Loading metadata...")
    dates = _read_dates(args.metadata) if args.metadata is not None else None
    logging.info("Estimating transmission distances...")
    if lead:
        with open(args.output_file, "w") as out:
            out.write(HEADER)
    import time
    trace = os.environ.get("TRACS_STAGE_TRACE") is not None      # "[stage] name seconds" lines on stderr (scripts/bench_e2e.py)
    t_stage = [time.perf_counter()]
    if trace and lead:
        try:
            import psutil
            sys.stderr.write("[stage] process start -> first alignment (interpreter, imports, metadata) %.4f s\n"
                             % (time.time() - psutil.Process().create_time()))
        except Exception:
            pass

    def stage(name):
        if trace and lead:
            now = time.perf_counter()
            sys.stderr.write("[stage] %s %.4f s\n" % (name, now - t_stage[0]))
            t_stage[0] = now
    for msa in args.msa_files:
        logging.info("Calculating pairwise snp distances for %s", msa)
        msas = [msa, args.msa_db] if args.msa_db is not None else [msa]
        t_stage[0] = time.perf_counter()
        ref = os.path.basename(msa).split(".")[0].replace("_combined", "")      # (:208-209)
        if ctx is None and os.environ.get("TRACS_DISTANCE_ARRAYS") is None:
            # one GPU: the results stay on the device until the CSV rows (with --filter: the filtered distances and the transmission
            # model they drive too)
            for p in msas:
                if not os.path.exists(p):
                    raise FileNotFoundError(p)               # (api.pairsnp_arrays's diagnosis; the reference passes a NULL gzFile on)
            if _rows_on_device(msas, args, dates, ref, stage):
                logging.info("Saving distances for %s", msa)
                continue
        if ctx is None:
            res = pairsnp_arrays(fasta=msas, n_threads=args.n_cpu, dist=args.snp_threshold, filter=args.recomb_filter)
        else:
            res = _pairs_multi_gpu(msas, args, ctx)
        stage("pairsnp (total, incl. the copy of the result into numpy arrays)")
        if not lead:
            continue
        rows, cols, snpd, names, filt, ncomp = res
        with_dates = dates is not None and len(rows) > 0
        tdist = ek = ddiff = None
        if with_dates:
            logging.info("Inferring transmission probabilities for %s", msa)
            # with --filter the transmission model is driven by the FILTERED distance (:183-193)
            drive = filt if args.recomb_filter else snpd
            tdist, ek, ddiff = calculate_trans_prob([rows, cols, drive], sample_dates=dates, K=100,
                                                    lamb=args.clock_rate, beta=args.trans_rate, samplenames=names,
                                                    log=False, precision=args.precision)
            if not args.recomb_filter:
                filt = None                                                     # a column of "NA" (:204)
            stage("transcluster (dates -> delta, H2D, keys, gather, D2H)")
        logging.info("Saving distances for %s", msa)
        _append_rows(args.output_file, names, rows, cols, snpd, filt, ncomp, ddiff, tdist, ek,
                     args.trans_threshold if with_dates else None, ref)
        stage("CSV rows (format + write, %d rows)" % len(rows))
    if ctx is not None:
        ctx[0].barrier()
        ctx[0].destroy_process_group()


def main():
    parser = distance_parser(argparse.ArgumentParser())
    args = parser.parse_args()
    args.func(args)


if __name__ == "__main__":
    main()
