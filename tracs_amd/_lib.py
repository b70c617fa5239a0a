"""ctypes binding of libtracs_hip.so (include/tracs_hip.h).

The HIP library IS the product: there is no CPU fallback.  If the shared library is missing,
does not load, or no GPU is visible, calls raise -- they never route through oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtracs_hip.so")
_lib = None

# every symbol include/tracs_hip.h declares (tests/test_cabi.py checks the .so exports them all)
SYMBOLS = [
    "tracs_last_error", "tracs_abi_version", "tracs_device_count",
    "tracs_pairsnp", "tracs_pairsnp_len", "tracs_pairsnp_nseq", "tracs_pairsnp_seqlen", "tracs_pairsnp_rows",
    "tracs_pairsnp_cols", "tracs_pairsnp_distances", "tracs_pairsnp_filt_distances", "tracs_pairsnp_ncompared",
    "tracs_pairsnp_name", "tracs_pairsnp_free",
    "tracs_trans_dist", "tracs_lprob_k_given_N", "tracs_calculate_posteriors", "tracs_connected_components",
    "tracs_find_dirichlet_priors", "tracs_find_dirichlet_priors_device",
    "tracs_alignment_create", "tracs_alignment_free", "tracs_alignment_n", "tracs_alignment_len",
    "tracs_alignment_bytes", "tracs_alignment_planes", "tracs_alignment_touch", "tracs_alignment_hint_rows", "tracs_pairsnp_notify_distances", "tracs_set_stream_policy", "tracs_alignment_pack", "tracs_alignment_from_fasta",
    "tracs_free",
    "tracs_pairsnp_dense", "tracs_pairsnp_dense_thr", "tracs_coo_count", "tracs_coo_fill", "tracs_filter_recomb_device", "tracs_filter_recomb_pairs",
    "tracs_trans_dist_device", "tracs_trans_dist_dense", "tracs_trans_dist_dense2", "tracs_trans_table_dense", "tracs_trans_table_gather",
    "tracs_trans_keys_words", "tracs_trans_keys_mark", "tracs_trans_keys_merge", "tracs_trans_keys_info", "tracs_trans_keys_evaluate", "tracs_trans_keys_gather",
    "tracs_calculate_posteriors_device", "tracs_posterior_codes_device", "tracs_posterior_codes_cov_device",
    "tracs_codes_to_iupac_device", "tracs_alignment_pack_codes", "tracs_alignment_pack_codes_batch", "tracs_edges_count_f64", "tracs_edges_fill_f64",
    "tracs_coverage_profile_device32", "tracs_posterior_codes_cov_device32", "tracs_consensus_codes_device32", "tracs_coverage_profile_device",
    "tracs_consensus_codes_device",
    "tracs_connected_components_device",
    "tracs_pileup_counts", "tracs_write_posterior_csv", "tracs_combine_fasta", "tracs_write_distance_rows",
    "tracs_read_distance_edges", "tracs_edges_count", "tracs_edges_rows", "tracs_edges_n_names", "tracs_edges_name", "tracs_edges_i",
    "tracs_edges_j", "tracs_edges_free",
    "tracs_comm_unique_id", "tracs_comm_create", "tracs_comm_free", "tracs_comm_rank", "tracs_comm_world", "tracs_bcast",
    "tracs_bcast_planes", "tracs_allgather_panels", "tracs_allreduce", "tracs_reduce_scatter", "tracs_send", "tracs_recv",
    "tracs_alltoall", "tracs_tri_pack", "tracs_tri_sum", "tracs_rccl_version",
    "tracs_coo_fill_f64", "tracs_distance_open", "tracs_distance_nseq", "tracs_distance_name", "tracs_distance_run", "tracs_distance_free",
    "tracs_warm_up",
]


class TracsError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, but torch links it as
    `libamdhip64.so`).  If our library pulled in /opt/rocm's copy first, a later `import torch` would load a
    SECOND HIP runtime into the process and one of the two would see no device.  So when torch is installed,
    its copy is loaded first (no torch import needed); our DT_NEEDED libamdhip64.so.7 then resolves to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load():
    """Load the library (building nothing: run `python -m tracs_amd.build` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise TracsError("libtracs_hip.so is missing (%s): build it with `python -m tracs_amd.build`; "
                         "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, i64, dbl = C.c_void_p, C.c_size_t, C.c_int32, C.c_int64, C.c_double
    u64p, dp = C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    L.tracs_last_error.restype = C.c_char_p
    L.tracs_abi_version.restype = C.c_int
    L.tracs_device_count.restype = C.c_int
    L.tracs_pairsnp.restype = C.c_int
    L.tracs_pairsnp.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    for name in ("len", "nseq", "seqlen"):
        f = getattr(L, "tracs_pairsnp_" + name)
        f.restype = sz
        f.argtypes = [vp]
    for name in ("rows", "cols", "distances", "filt_distances", "ncompared"):
        f = getattr(L, "tracs_pairsnp_" + name)
        f.restype = u64p
        f.argtypes = [vp]
    L.tracs_pairsnp_name.restype = C.c_char_p
    L.tracs_pairsnp_name.argtypes = [vp, sz]
    L.tracs_pairsnp_free.restype = None
    L.tracs_pairsnp_free.argtypes = [vp]
    L.tracs_trans_dist.restype = C.c_int
    L.tracs_trans_dist.argtypes = [C.POINTER(i32), dp, sz, dbl, dbl, dbl, dp, dp]
    L.tracs_lprob_k_given_N.restype = C.c_int
    L.tracs_lprob_k_given_N.argtypes = [u64p, u64p, dp, sz, dbl, dbl, dp, sz, dp, dp]
    L.tracs_calculate_posteriors.restype = C.c_int
    L.tracs_calculate_posteriors.argtypes = [dp, sz, sz, dp, C.c_int, dbl, dp]
    L.tracs_find_dirichlet_priors.restype = C.c_int
    L.tracs_find_dirichlet_priors.argtypes = [dp, sz, sz, C.c_int, dbl, C.c_int, dbl, dp, C.POINTER(C.c_int)]
    L.tracs_find_dirichlet_priors_device.restype = C.c_int
    L.tracs_find_dirichlet_priors_device.argtypes = [vp, sz, sz, C.c_int, dbl, C.c_int, dbl, dp, C.POINTER(C.c_int), vp]
    L.tracs_connected_components.restype = C.c_int
    L.tracs_connected_components.argtypes = [C.POINTER(i32), C.POINTER(i32), sz, sz, C.POINTER(i32), C.POINTER(i32)]
    L.tracs_alignment_create.restype = C.c_int
    L.tracs_alignment_create.argtypes = [sz, sz, C.POINTER(vp)]
    L.tracs_alignment_free.restype = None
    L.tracs_alignment_free.argtypes = [vp]
    for name in ("n", "len", "bytes"):
        f = getattr(L, "tracs_alignment_" + name)
        f.restype = sz
        f.argtypes = [vp]
    L.tracs_alignment_planes.restype = vp
    L.tracs_alignment_planes.argtypes = [vp]
    L.tracs_alignment_touch.restype = C.c_int
    L.tracs_alignment_touch.argtypes = [vp]
    L.tracs_alignment_pack.restype = C.c_int
    L.tracs_alignment_pack.argtypes = [vp, vp, sz, sz, C.c_int, vp]
    L.tracs_alignment_from_fasta.restype = C.c_int
    L.tracs_alignment_from_fasta.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp), C.POINTER(vp),
                                             C.POINTER(sz), C.POINTER(sz)]
    L.tracs_free.restype = None
    L.tracs_free.argtypes = [vp]
    L.tracs_pairsnp_dense.restype = C.c_int
    L.tracs_pairsnp_dense.argtypes = [vp, sz, sz, sz, vp, vp, sz, vp]
    L.tracs_pairsnp_dense_thr.restype = C.c_int
    L.tracs_pairsnp_dense_thr.argtypes = [vp, sz, sz, sz, vp, vp, sz, i32, vp]
    L.tracs_coo_count.restype = C.c_int
    L.tracs_coo_count.argtypes = [vp, sz, sz, sz, sz, sz, i32, vp, vp]
    L.tracs_coo_fill.restype = C.c_int
    L.tracs_coo_fill.argtypes = [vp, vp, sz, sz, sz, sz, sz, i32, vp, vp, vp, vp, vp, vp]
    L.tracs_coo_fill_f64.restype = C.c_int
    L.tracs_coo_fill_f64.argtypes = [vp, sz, sz, sz, sz, sz, i32, vp, vp, vp, vp, vp, vp]
    L.tracs_distance_open.restype = C.c_int
    L.tracs_distance_open.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(vp)]
    L.tracs_distance_nseq.restype = sz
    L.tracs_distance_nseq.argtypes = [vp]
    L.tracs_distance_name.restype = C.c_char_p
    L.tracs_distance_name.argtypes = [vp, sz]
    L.tracs_distance_run.restype = C.c_int
    L.tracs_distance_run.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), dbl, dbl, dbl, dbl, C.c_char_p, C.c_char_p, C.c_int, u64p, u64p]
    L.tracs_warm_up.restype = None
    L.tracs_warm_up.argtypes = []
    L.tracs_distance_free.restype = None
    L.tracs_distance_free.argtypes = [vp]
    L.tracs_edges_count_f64.restype = C.c_int
    L.tracs_edges_count_f64.argtypes = [vp, vp, sz, sz, sz, sz, sz, C.c_int32, dbl, vp, vp]
    L.tracs_edges_fill_f64.restype = C.c_int
    L.tracs_edges_fill_f64.argtypes = [vp, vp, sz, sz, sz, sz, sz, C.c_int32, dbl, vp, vp, vp, vp, vp]
    L.tracs_filter_recomb_device.restype = C.c_int
    L.tracs_filter_recomb_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, vp]
    L.tracs_filter_recomb_pairs.restype = C.c_int
    L.tracs_filter_recomb_pairs.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    L.tracs_debug_filter_index.restype = C.c_int
    L.tracs_debug_filter_index.argtypes = [vp, C.POINTER(C.c_double)]
    L.tracs_trans_dist_device.restype = C.c_int
    L.tracs_trans_dist_device.argtypes = [vp, vp, sz, dbl, dbl, dbl, C.c_int, vp, vp, vp]
    L.tracs_trans_dist_dense.restype = C.c_int
    L.tracs_trans_dist_dense.argtypes = [vp, sz, sz, sz, sz, sz, i32, vp, dbl, dbl, dbl, C.c_int, vp, vp, vp]
    L.tracs_trans_dist_dense2.restype = C.c_int
    L.tracs_trans_dist_dense2.argtypes = [vp, sz, sz, C.POINTER(sz), C.c_int, sz, i32, vp, dbl, dbl, dbl, C.c_int, vp, vp, vp]
    L.tracs_trans_table_dense.restype = C.c_int
    L.tracs_trans_table_dense.argtypes = [vp, sz, sz, sz, sz, sz, C.c_int32, vp, dbl, dbl, dbl, C.c_int, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.tracs_trans_table_gather.restype = C.c_int
    L.tracs_trans_table_gather.argtypes = [vp, sz, sz, sz, sz, sz, C.c_int32, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int, vp, vp, vp, vp]
    L.tracs_trans_keys_words.restype = sz
    L.tracs_trans_keys_words.argtypes = []
    L.tracs_trans_keys_mark.restype = C.c_int
    L.tracs_trans_keys_mark.argtypes = [vp, sz, sz, C.POINTER(sz), C.c_int, sz, C.c_int32, vp, vp, vp]
    L.tracs_trans_keys_merge.restype = C.c_int
    L.tracs_trans_keys_merge.argtypes = [vp, vp, C.c_int, vp]
    L.tracs_trans_keys_info.restype = C.c_int
    L.tracs_trans_keys_info.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    L.tracs_trans_keys_evaluate.restype = C.c_int
    L.tracs_trans_keys_evaluate.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int, C.c_int, dbl, dbl, dbl, vp, sz, vp]
    L.tracs_trans_keys_gather.restype = C.c_int
    L.tracs_trans_keys_gather.argtypes = [vp, sz, sz, C.POINTER(sz), C.c_int, sz, C.c_int32, vp, vp, C.POINTER(C.c_uint64), vp, C.c_int, sz, C.c_int, vp, vp, vp]
    L.tracs_calculate_posteriors_device.restype = C.c_int
    L.tracs_calculate_posteriors_device.argtypes = [vp, sz, sz, dp, C.c_int, dbl, vp, vp]
    L.tracs_posterior_codes_device.restype = C.c_int
    L.tracs_posterior_codes_device.argtypes = [vp, sz, dp, C.c_int, dbl, vp, vp]
    L.tracs_posterior_codes_cov_device.restype = C.c_int
    L.tracs_posterior_codes_cov_device.argtypes = [vp, sz, dp, C.c_int, dbl, C.c_uint32, dbl, dbl, vp, vp]
    L.tracs_codes_to_iupac_device.restype = C.c_int
    L.tracs_codes_to_iupac_device.argtypes = [vp, sz, vp, vp]
    L.tracs_alignment_pack_codes.restype = C.c_int
    L.tracs_alignment_pack_codes.argtypes = [vp, vp, sz, vp]
    L.tracs_alignment_pack_codes_batch.restype = C.c_int
    L.tracs_alignment_pack_codes_batch.argtypes = [vp, vp, sz, sz, sz, vp]
    L.tracs_connected_components_device.restype = C.c_int
    L.tracs_connected_components_device.argtypes = [vp, vp, sz, sz, vp, C.POINTER(i32), vp]
    cpp = C.POINTER(C.c_char_p)
    L.tracs_coverage_profile_device.restype = C.c_int
    L.tracs_coverage_profile_device.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    L.tracs_consensus_codes_device.restype = C.c_int
    L.tracs_consensus_codes_device.argtypes = [vp, sz, C.c_uint32, vp, vp]
    L.tracs_coverage_profile_device32.restype = C.c_int
    L.tracs_coverage_profile_device32.argtypes = [vp, sz, vp, sz, vp, vp, vp]
    L.tracs_posterior_codes_cov_device32.restype = C.c_int
    L.tracs_posterior_codes_cov_device32.argtypes = [vp, sz, dp, C.c_int, dbl, C.c_uint32, dbl, dbl, vp, vp]
    L.tracs_consensus_codes_device32.restype = C.c_int
    L.tracs_consensus_codes_device32.argtypes = [vp, sz, C.c_uint32, vp, vp]
    L.tracs_pileup_counts.restype = C.c_int
    L.tracs_pileup_counts.argtypes = [C.c_char_p, cpp, u64p, sz, C.c_int, dp, u64p]
    L.tracs_write_posterior_csv.restype = C.c_int
    L.tracs_write_posterior_csv.argtypes = [C.c_char_p, dp, sz, sz, C.c_int]
    L.tracs_write_distance_rows.restype = C.c_int
    L.tracs_write_distance_rows.argtypes = [C.c_char_p, cpp, u64p, u64p, u64p, u64p, u64p, dp, dp, dp, sz, C.c_int, dbl, C.c_char_p, u64p]
    L.tracs_debug_format_floats.restype = C.c_long
    L.tracs_debug_format_floats.argtypes = [dp, sz, C.c_char_p, sz]
    L.tracs_read_distance_edges.restype = C.c_int
    L.tracs_read_distance_edges.argtypes = [C.c_char_p, C.c_int, dbl, cpp, sz, C.POINTER(vp)]
    L.tracs_edges_count.restype = sz
    L.tracs_edges_count.argtypes = [vp]
    L.tracs_edges_rows.restype = C.c_uint64
    L.tracs_edges_rows.argtypes = [vp]
    L.tracs_edges_n_names.restype = sz
    L.tracs_edges_n_names.argtypes = [vp]
    L.tracs_edges_name.restype = C.c_char_p
    L.tracs_edges_name.argtypes = [vp, sz]
    L.tracs_edges_i.restype = C.POINTER(i32)
    L.tracs_edges_i.argtypes = [vp]
    L.tracs_edges_j.restype = C.POINTER(i32)
    L.tracs_edges_j.argtypes = [vp]
    L.tracs_edges_free.restype = None
    L.tracs_edges_free.argtypes = [vp]
    L.tracs_combine_fasta.restype = C.c_int
    L.tracs_combine_fasta.argtypes = [C.c_char_p, cpp, cpp, sz, C.c_int, C.c_int, dp, u64p]
    L.tracs_debug_read_fasta.restype = C.c_int
    L.tracs_debug_read_fasta.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_uint64)]
    L.tracs_debug_alignment_encoding.restype = C.c_int
    L.tracs_debug_alignment_encoding.argtypes = [vp]
    L.tracs_debug_alignment_kernel.restype = C.c_int
    L.tracs_debug_alignment_kernel.argtypes = [vp]
    L.tracs_debug_alignment_site_classes.restype = C.c_int
    L.tracs_debug_alignment_site_classes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.tracs_pairsnp_notify_distances.restype = None
    L.tracs_pairsnp_notify_distances.argtypes = [vp]
    L.tracs_set_stream_policy.restype = None
    L.tracs_set_stream_policy.argtypes = [C.c_int]
    L.tracs_alignment_hint_rows.restype = C.c_int
    L.tracs_alignment_hint_rows.argtypes = [vp, C.POINTER(sz), C.c_int]
    L.tracs_debug_alignment_nw_gram.restype = C.c_int
    L.tracs_debug_alignment_nw_gram.argtypes = [vp]
    L.tracs_debug_alignment_count_source.restype = C.c_int
    L.tracs_debug_alignment_count_source.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.tracs_debug_force_site_classes.restype = None
    L.tracs_debug_force_site_classes.argtypes = [C.c_int]
    L.tracs_debug_pack_timing.restype = None
    L.tracs_debug_pack_timing.argtypes = [C.c_int]
    L.tracs_debug_pack_stages.restype = C.c_int
    L.tracs_debug_pack_stages.argtypes = [C.c_char_p, sz, C.POINTER(C.c_float), C.c_int]
    L.tracs_debug_pack_stage_bytes.restype = C.c_int
    L.tracs_debug_pack_stage_bytes.argtypes = [dp, dp, C.c_int]
    L.tracs_comm_unique_id.restype = C.c_int
    L.tracs_comm_unique_id.argtypes = [vp, sz]
    L.tracs_comm_create.restype = C.c_int
    L.tracs_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.tracs_comm_free.restype = None
    L.tracs_comm_free.argtypes = [vp]
    L.tracs_comm_rank.restype = C.c_int
    L.tracs_comm_rank.argtypes = [vp]
    L.tracs_comm_world.restype = C.c_int
    L.tracs_comm_world.argtypes = [vp]
    L.tracs_bcast.restype = C.c_int
    L.tracs_bcast.argtypes = [vp, vp, sz, C.c_int, vp]
    L.tracs_bcast_planes.restype = C.c_int
    L.tracs_bcast_planes.argtypes = [vp, vp, C.c_int, vp]
    L.tracs_allgather_panels.restype = C.c_int
    L.tracs_allgather_panels.argtypes = [vp, vp, C.POINTER(sz), sz, vp]
    L.tracs_allreduce.restype = C.c_int
    L.tracs_allreduce.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp]
    L.tracs_reduce_scatter.restype = C.c_int
    L.tracs_reduce_scatter.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp]
    L.tracs_alltoall.restype = C.c_int
    L.tracs_alltoall.argtypes = [vp, vp, vp, sz, vp]
    L.tracs_tri_pack.restype = C.c_int
    L.tracs_tri_pack.argtypes = [vp, sz, sz, sz, sz, sz, vp, C.c_int, C.c_uint32, C.c_int, vp, vp, vp]
    L.tracs_tri_sum.restype = C.c_int
    L.tracs_tri_sum.argtypes = [vp, sz, sz, sz, sz, sz, vp, C.c_int, vp, sz, C.c_int, C.c_int, C.c_uint32, C.c_int, vp]
    L.tracs_rccl_version.restype = C.c_int
    L.tracs_rccl_version.argtypes = []
    L.tracs_send.restype = C.c_int
    L.tracs_send.argtypes = [vp, vp, sz, C.c_int, vp]
    L.tracs_recv.restype = C.c_int
    L.tracs_recv.argtypes = [vp, vp, sz, C.c_int, vp]
    L.tracs_debug_lists.restype = sz
    L.tracs_debug_lists.argtypes = [vp, C.c_int, vp, sz]
    L.tracs_debug_pair_timing.restype = None
    L.tracs_debug_pair_timing.argtypes = [C.c_int]
    L.tracs_debug_last_pair_ms.restype = C.c_int
    L.tracs_debug_pair_ms_mean.restype = C.c_int
    L.tracs_debug_pair_ms_mean.argtypes = [C.c_int, C.POINTER(C.c_float)]
    L.tracs_debug_last_pair_ms.argtypes = [C.POINTER(C.c_float)]
    L.tracs_debug_tile_variant.restype = C.c_char_p
    L.tracs_debug_mfma_shape.restype = C.c_char_p
    L.tracs_debug_last_trans_dist_keys.restype = C.c_uint64
    L.tracs_debug_iupac_mask.restype = C.c_int
    L.tracs_debug_iupac_mask.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc):
    """Raise RuntimeError with the library's message (the reference raises RuntimeError with the
    same strings: src/pairsnp.hpp:86,96-97,342) on a non-zero return code."""
    if rc == -8:                                   # TRACS_E_INTERRUPTED: Ctrl-C arrived while the library held the call
        raise KeyboardInterrupt("Interrupted by user!")
    if rc != 0:
        msg = load().tracs_last_error()
        raise RuntimeError(msg.decode("utf-8", "replace") if msg else "libtracs_hip error %d" % rc)


def require_gpu():
    L = load()
    if L.tracs_device_count() <= 0:
        raise TracsError("no MI355X/HIP device visible: the TRACS distance path runs on the GPU only "
                         "(there is no CPU fallback)")
    return L
