"""Device-resident entry points (include/tracs_hip.h part 2) over torch tensors.

torch is used for device memory, streams and torch.distributed only -- plumbing; every kernel is
in libtracs_hip.so.  Import torch BEFORE this module's first call so that both share one HIP
runtime (torch ships its own libamdhip64.so with the same soname).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _panel_ptr(t, base_row):
    """Pointer the dense entry points can index with ABSOLUTE row numbers when `t` ([rows, ld]) only holds rows
    base_row.. of the matrix (row panels of matrices too large to allocate whole)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr() - int(base_row) * t.stride(0) * t.element_size())


class Alignment:
    """A packed alignment resident in HBM (tracs_alignment)."""

    def __init__(self, n, L):
        self._L = _lib.require_gpu()
        self._h = C.c_void_p()
        _lib.check(self._L.tracs_alignment_create(int(n), int(L), C.byref(self._h)))
        self.n, self.L = int(n), int(L)

    @classmethod
    def from_fasta(cls, paths):
        import os
        L = _lib.require_gpu()
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h, names, nb, n0 = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        _lib.check(L.tracs_alignment_from_fasta(arr, len(paths), C.byref(h), C.byref(names), C.byref(nb), C.byref(n0)))
        self = cls.__new__(cls)
        self._L, self._h = L, h
        self.n, self.L = L.tracs_alignment_n(h), L.tracs_alignment_len(h)
        raw = C.string_at(names, nb.value) if nb.value else b""
        L.tracs_free(names)
        self.names = [x.decode("utf-8", "replace") for x in raw.split(b"\0")[:self.n]]
        self.n_first = n0.value
        return self

    def pack(self, ascii_u8, first=0):
        """ascii_u8: torch.uint8 [count, L] on the GPU, or a numpy uint8 array on the host."""
        if isinstance(ascii_u8, torch.Tensor):
            assert ascii_u8.dtype == torch.uint8 and ascii_u8.is_contiguous() and ascii_u8.shape[1] == self.L
            on_dev = 1 if ascii_u8.is_cuda else 0
            ptr, count = C.c_void_p(ascii_u8.data_ptr()), ascii_u8.shape[0]
        else:
            a = np.ascontiguousarray(ascii_u8, dtype=np.uint8)
            assert a.ndim == 2 and a.shape[1] == self.L
            on_dev, ptr, count = 0, C.c_void_p(a.ctypes.data), a.shape[0]
        _lib.check(self._L.tracs_alignment_pack(self._h, ptr, int(first), int(count), on_dev, _stream()))

    def mark_packed(self):
        """The planes were written from outside the library (a broadcast from another rank): drop every cached derived form."""
        _lib.check(self._L.tracs_alignment_touch(self._h))

    def hint_rows(self, ranges):
        """A multi-GPU rank's promise: this handle will only be asked for rows inside `ranges` ([(begin, end), ..], at most two;
        [] lifts it), so what is built per row once per pack is built for those rows only.  Dense calls for other rows fail."""
        flat = [int(x) for r in ranges for x in r]
        arr = (C.c_size_t * max(len(flat), 1))(*flat)
        _lib.check(self._L.tracs_alignment_hint_rows(self._h, arr, len(ranges)))

    def pack_codes(self, codes, sample):
        """Pack samples straight from their packed 4-bit allele masks (posterior_codes_device output):
        codes uint8 [(L+1)//2] -> one sample, or uint8 [count, stride >= (L+1)//2] -> samples sample..sample+count-1."""
        assert codes.dtype == torch.uint8 and codes.is_cuda
        if codes.dim() == 1:
            assert codes.numel() == (self.L + 1) // 2
            _lib.check(self._L.tracs_alignment_pack_codes(self._h, _ptr(codes), int(sample), _stream()))
        else:
            assert codes.dim() == 2 and codes.stride(1) == 1 and codes.shape[1] >= (self.L + 1) // 2
            _lib.check(self._L.tracs_alignment_pack_codes_batch(self._h, _ptr(codes), int(codes.stride(0)), int(sample),
                                                                int(codes.shape[0]), _stream()))

    @property
    def encoding(self):
        """'consensus' (3 planes) / 'general' (5 planes) as used by the last dense call, None before the first."""
        e = self._L.tracs_debug_alignment_encoding(self._h)
        return {0: "general", 1: "consensus"}.get(e)

    @property
    def kernel(self):
        """'mfma' (matrix-core kernel, consensus operands) / 'mfma-general' (matrix-core kernel, one-hot operands + sparse
        partial-code correction) / 'valu' (tile kernel) as used by the last dense call, None before the first."""
        return {0: "valu", 1: "mfma", 2: "mfma-general"}.get(self._L.tracs_debug_alignment_kernel(self._h))

    @property
    def site_classes(self):
        """(dense, counted, minority, full) sites when the last dense call ran on site classes (csrc/site_classes.hip): the
        pair kernel read the dense sites only, a one-operand pass over the counted sites (and a constant for the full ones:
        nobody is N there) completed the compared-sites counts, and the minority sites -- counted or full sites at which a few
        samples differ -- added their distances from sparse lists; None when the whole alignment was read."""
        import ctypes as C
        out = (C.c_uint64 * 4)()
        state = self._L.tracs_debug_alignment_site_classes(self._h, out)
        return tuple(int(x) for x in out) if state == 1 else None

    @property
    def nw_gram(self):
        """True when the minority sites' N x listed terms of the last decided classes come from two one-plane passes on the matrix cores
        (U U^T - n n^T, csrc/site_classes.hip) instead of walks of the sites' N lists."""
        return bool(self._L.tracs_debug_alignment_nw_gram(self._h))

    @property
    def nw_form(self):
        """None / 'u-pass' (U U^T - n n^T: two one-plane matrix passes) / 'ns-rows' (the rows of the site-major N matrix summed per listed
        sample inside the fix-up: one matrix pass, for the compared sites alone) -- how the second form of the classes gets its terms."""
        return {0: None, 1: "u-pass", 2: "ns-rows"}.get(self._L.tracs_debug_alignment_nw_gram(self._h))

    @property
    def count_source(self):
        """(sites the counting pass reads on the matrix cores, in_place, sites whose N co-occurrences come from lists) for an
        alignment on site classes: in_place = the stored N plane of every site, read where it lies (the pair kernels then write d
        only); else the N plane of the sites with many N samples, re-packed.  None without classes."""
        out = (C.c_uint64 * 8)()
        return (int(out[0]), bool(out[1]), int(out[2])) if self._L.tracs_debug_alignment_count_source(self._h, out) else None

    @property
    def list_stats(self):
        """{nn_visits, nn_walks, n_lines, p_entries, n_entry_bytes[, max_row_n, bitmaps, row_splits]}: list entries one pass of the N
        co-occurrence walk decodes (sum of cN^2 over the sites whose co-occurrences come from lists) in how many list walks (sum of
        cN), 128-byte lines reserved for the per-site N lists, entries of the listed-sample lists, bytes per N list entry; the most N
        sites any sample has and the workgroups its row is cut over; None without classes."""
        out = (C.c_uint64 * 8)()
        if not self._L.tracs_debug_alignment_count_source(self._h, out):
            return None
        st = {"nn_visits": int(out[3]), "n_lines": int(out[4]), "p_entries": int(out[5]), "n_entry_bytes": int(out[6]),
              "nn_walks": int(out[7])}
        sizes = (C.c_uint64 * 8)()
        if self._L.tracs_debug_lists(self._h, 0, sizes, 64) == 64:
            # (nn_rows_kernel cuts a row over up to 32 workgroups by its N sites: csrc/site_lists.hip)
            target = max(8192, st["nn_walks"] // 2048)
            st.update(max_row_n=int(sizes[7]), bitmaps=bool(sizes[5]), row_splits=min(32, max(1, -(-int(sizes[7]) // target))))
        return st

    @property
    def nbytes(self):
        return self._L.tracs_alignment_bytes(self._h)

    def planes_ptr(self):
        return self._L.tracs_alignment_planes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tracs_alignment_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def notify_distances(event):
    """The next pairsnp_dense call records `event` (torch.cuda.Event) on its stream once the distances are final -- before the
    compared-sites counts are (include/tracs_hip.h, "Two streams")."""
    if not event.cuda_event:
        # torch creates the hipEvent lazily, on the first record: force it into existence (the handle of an unrecorded
        # torch.cuda.Event is 0 -- the library would then record nothing and a wait on the event would not wait)
        event.record(torch.cuda.current_stream())
    if not event.cuda_event:
        raise _lib.TracsError("notify_distances: the event has no HIP handle")
    _lib.load().tracs_pairsnp_notify_distances(C.c_void_p(event.cuda_event))


def set_stream_policy(caller_orders_streams):
    _lib.load().tracs_set_stream_policy(1 if caller_orders_streams else 0)


def pack_stages():
    """[(stage, ms), ..] of the last once-per-pack build (encoding decision, site classes, lists), from HIP events on the launch
    stream; recorded when tracs_debug_pack_timing(1) was set before the dense call (diagnostics: bench.py's single_pass)."""
    lib = _lib.load()
    names = C.create_string_buffer(1024)
    ms = (C.c_float * 16)()
    k = lib.tracs_debug_pack_stages(names, 1024, ms, 16)
    return list(zip(names.value.decode().split("\n"), [float(x) for x in ms[:k]])) if k else []


def pack_stages_bytes():
    """[(stage, ms, bytes read, bytes written), ..] of the last once-per-pack build: pack_stages() with the library's own
    accounting of what each stage reads and writes (bench.py's roofline_per_pack)."""
    lib = _lib.load()
    st = pack_stages()
    rd, wr = (C.c_double * 16)(), (C.c_double * 16)()
    k = lib.tracs_debug_pack_stage_bytes(rd, wr, 16)
    return [(name, ms, float(rd[i]) if i < k else 0.0, float(wr[i]) if i < k else 0.0) for i, (name, ms) in enumerate(st)]


def pairsnp_dense(aln, dist, ncomp=None, row_begin=0, row_end=None, col_begin=0, dist_threshold=None, base_row=0):
    """dist/ncomp: torch.int32 (bit pattern uint32) device matrices [rows, ld >= n] holding rows base_row.. of the pair matrix,
    written for the cells rows [row_begin,row_end) x cols [max(col_begin,i+1), n).  With dist_threshold, pairs beyond it may
    come back negative (bit 31 set) with an unspecified ncomp: tiles stop early once all their pairs are past the threshold."""
    row_end = aln.n if row_end is None else row_end
    ld = dist.stride(0)
    assert base_row <= row_begin and row_end - base_row <= dist.shape[0]
    if dist_threshold is None:
        _lib.check(aln._L.tracs_pairsnp_dense(aln._h, int(row_begin), int(row_end), int(col_begin), _panel_ptr(dist, base_row),
                                              _panel_ptr(ncomp, base_row), int(ld), _stream()))
    else:
        _lib.check(aln._L.tracs_pairsnp_dense_thr(aln._h, int(row_begin), int(row_end), int(col_begin), _panel_ptr(dist, base_row),
                                                  _panel_ptr(ncomp, base_row), int(ld), int(dist_threshold), _stream()))


def coo_from_dense(dist, ncomp, n, dist_threshold=2147483647, row_begin=0, row_end=None, col_begin=0, base_row=0):
    """-> rows, cols, d, nn (torch.int32 on device), row-major like the reference's output."""
    L = _lib.require_gpu()
    row_end = n if row_end is None else row_end
    nrows = max(0, row_end - row_begin)
    off = torch.zeros(nrows + 1, dtype=torch.int64, device=dist.device)
    ld = dist.stride(0)
    _lib.check(L.tracs_coo_count(_panel_ptr(dist, base_row), ld, n, row_begin, row_end, col_begin, int(dist_threshold), _ptr(off), _stream()))
    total = int(off[nrows].item())
    out = [torch.empty(total, dtype=torch.int32, device=dist.device) for _ in range(4)]
    if total:
        _lib.check(L.tracs_coo_fill(_panel_ptr(dist, base_row), _panel_ptr(ncomp, base_row), ld, n, row_begin, row_end, col_begin,
                                    int(dist_threshold), _ptr(off), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _stream()))
    return out


def edges_from_dense_f64(val, dist, n, threshold, dist_threshold=2147483647, row_begin=0, row_end=None, col_begin=0, base_row=0,
                         with_values=False):
    """Cells (i, j > i) with dist <= dist_threshold and val <= threshold, row-major -> rows, cols (torch.int32)[, values f64]:
    the edges `tracs cluster -D expectedK|direct -c threshold` keeps (tracs/cluster.py:110-112), straight from device panels."""
    L = _lib.require_gpu()
    row_end = n if row_end is None else row_end
    nrows = max(0, row_end - row_begin)
    off = torch.zeros(nrows + 1, dtype=torch.int64, device=val.device)
    ld = val.stride(0)
    assert dist.stride(0) == ld
    _lib.check(L.tracs_edges_count_f64(_panel_ptr(val, base_row), _panel_ptr(dist, base_row), ld, n, row_begin, row_end, col_begin,
                                       int(dist_threshold), float(threshold), _ptr(off), _stream()))
    total = int(off[nrows].item())
    rows = torch.empty(total, dtype=torch.int32, device=val.device)
    cols = torch.empty(total, dtype=torch.int32, device=val.device)
    vals = torch.empty(total, dtype=torch.float64, device=val.device) if with_values else None
    if total:
        _lib.check(L.tracs_edges_fill_f64(_panel_ptr(val, base_row), _panel_ptr(dist, base_row), ld, n, row_begin, row_end, col_begin,
                                          int(dist_threshold), float(threshold), _ptr(off), _ptr(rows), _ptr(cols), _ptr(vals), _stream()))
    return (rows, cols, vals) if with_values else (rows, cols)


def filter_recomb_device(aln, rows, cols, d):
    """rows/cols/d: torch.int32 device vectors of emitted pairs -> filtered distances (torch.int32)."""
    L = _lib.require_gpu()
    n = rows.numel()
    off = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
    off[1:] = torch.cumsum(d.to(torch.int64), 0)
    total = int(off[n].item()) if n else 0
    pos = torch.empty(max(total, 1), dtype=torch.int32, device=rows.device)
    found = torch.empty(max(n, 1), dtype=torch.int32, device=rows.device)
    filt = torch.empty(max(n, 1), dtype=torch.int32, device=rows.device)
    _lib.check(L.tracs_filter_recomb_device(aln._h, _ptr(rows), _ptr(cols), n, _ptr(off), _ptr(pos), _ptr(found), _ptr(filt),
                                            _stream()))
    return filt[:n], found[:n], pos[:total], off


def filter_recomb_pairs(aln, rows, cols, d):
    """Filtered distances (src/pairsnp.hpp:251-318) of the emitted pairs (rows, cols) with SNP distances d: torch.int32 device
    vectors in, torch.int32 out.  The pairs' SNP sites come from the samples' departure lists (csrc/filter_lists.hip), built on the
    alignment's first filter call after a pack."""
    L = _lib.require_gpu()
    n = rows.numel()
    filt = torch.empty(max(n, 1), dtype=torch.int32, device=rows.device)
    if n:
        _lib.check(L.tracs_filter_recomb_pairs(aln._h, _ptr(rows), _ptr(cols), _ptr(d), n, _ptr(filt), _stream()))
    return filt[:n]


def filter_index_info(aln):
    """Diagnostics of the alignment's filter index (None before the first filter call)."""
    import ctypes as C
    L = _lib.require_gpu()
    out = (C.c_double * 10)()
    if not L.tracs_debug_filter_index(aln._h, out):
        return None
    keys = ("ref", "count", "offsets", "fill", "nt", "ns")
    return {"lists": bool(out[0]), "entries": int(out[1]), "longest_list": int(out[2]), "alloc_ms": out[3],
            "build_ms": {k: out[4 + i] for i, k in enumerate(keys)}}


def trans_dist_device(snpdiff, datediff, lamb, beta, threshold_Ek, exp_p0=False):
    L = _lib.require_gpu()
    n = snpdiff.numel()
    p0 = torch.empty(n, dtype=torch.float64, device=snpdiff.device)
    eK = torch.empty(n, dtype=torch.float64, device=snpdiff.device)
    _lib.check(L.tracs_trans_dist_device(_ptr(snpdiff), _ptr(datediff), n, float(lamb), float(beta), float(threshold_Ek),
                                         int(exp_p0), _ptr(p0), _ptr(eK), _stream()))
    return p0, eK


def trans_dist_dense(dist, n, days, lamb, beta, threshold_Ek, p0, eK, exp_p0=True, dist_threshold=2147483647,
                     row_begin=0, row_end=None, col_begin=0, base_row=0):
    L = _lib.require_gpu()
    row_end = n if row_end is None else row_end
    ld = dist.stride(0)
    assert p0.stride(0) == ld and eK.stride(0) == ld
    _lib.check(L.tracs_trans_dist_dense(_panel_ptr(dist, base_row), ld, n, row_begin, row_end, col_begin, int(dist_threshold), _ptr(days),
                                        float(lamb), float(beta), float(threshold_Ek), int(exp_p0), _panel_ptr(p0, base_row),
                                        _panel_ptr(eK, base_row), _stream()))


def trans_dist_dense_ranges(dist, n, days, lamb, beta, threshold_Ek, p0, eK, ranges, exp_p0=True,
                            dist_threshold=2147483647, col_begin=0):
    """One pass (one key table) over one or two row panels: ranges = [(b0, e0)] or [(b0, e0), (b1, e1)]."""
    L = _lib.require_gpu()
    ld = dist.stride(0)
    assert p0.stride(0) == ld and eK.stride(0) == ld and 1 <= len(ranges) <= 2
    flat = [int(x) for r in ranges for x in r]
    arr = (C.c_size_t * len(flat))(*flat)
    _lib.check(L.tracs_trans_dist_dense2(_ptr(dist), ld, n, arr, len(ranges), col_begin, int(dist_threshold), _ptr(days),
                                         float(lamb), float(beta), float(threshold_Ek), int(exp_p0), _ptr(p0), _ptr(eK),
                                         _stream()))


def trans_dist_dense_partitioned(dist, n, days, lamb, beta, threshold_Ek, p0, eK, part, parts, all_reduce, exp_p0=True,
                                 dist_threshold=2147483647, n_max=None, d_max=None):
    """trans_dist_dense over the WHOLE matrix on every rank, with the key evaluations split over the ranks: this rank evaluates
    the keys of hash class `part` of `parts` into a dense key table, `all_reduce(tensor)` sums the tables over the ranks
    (torch.distributed: RCCL), and p0 / eK of every cell are read from the completed table."""
    L = _lib.require_gpu()
    ld = dist.stride(0)
    assert p0.stride(0) == ld and eK.stride(0) == ld
    if n_max is None:
        valid = torch.triu(dist[:n, :n], diagonal=1)
        n_max = int(torch.where(valid <= dist_threshold, valid, torch.zeros_like(valid)).max().item())
    if d_max is None:
        d_max = int((days.max() - days.min()).item())
    table = torch.zeros((2, n_max + 1, d_max + 1), dtype=torch.float64, device=dist.device)
    flag = torch.zeros(1, dtype=torch.int32, device=dist.device)
    _lib.check(L.tracs_trans_table_dense(_ptr(dist), ld, n, 0, n, 0, int(dist_threshold), _ptr(days), float(lamb), float(beta),
                                         float(threshold_Ek), int(part), int(parts), n_max, d_max, _ptr(table[0]), _ptr(table[1]),
                                         _ptr(flag), _stream()))
    if parts > 1:
        all_reduce(table)
    _lib.check(L.tracs_trans_table_gather(_ptr(dist), ld, n, 0, n, 0, int(dist_threshold), _ptr(days), n_max, d_max, _ptr(table[0]),
                                          _ptr(table[1]), int(exp_p0), _ptr(p0), _ptr(eK), _ptr(flag), _stream()))
    if int(flag.item()):
        raise _lib.TracsError("trans_dist key table too small (n_max / d_max)")


# ---- the distinct keys split over ranks that each hold rows of the matrix (partition.KeySplit; csrc/transcluster.hip) ---------------
def trans_keys_words():
    return int(_lib.require_gpu().tracs_trans_keys_words())


def _ranges_arg(ranges):
    flat = [int(x) for r in ranges for x in r]
    return (C.c_size_t * max(1, len(flat)))(*flat), len(ranges)


def trans_keys_mark(dist, n, days, ranges, keys, dist_threshold=2147483647, col_begin=0):
    """keys (torch.int32 [trans_keys_words()], overwritten) <- the (SNP distance, day gap) keys of the cells of the row ranges"""
    L = _lib.require_gpu()
    assert len(ranges) <= 2 and keys.is_contiguous() and keys.numel() == trans_keys_words()
    arr, k = _ranges_arg(ranges)
    _lib.check(L.tracs_trans_keys_mark(_ptr(dist), dist.stride(0), n, arr, k, col_begin, int(dist_threshold), _ptr(days), _ptr(keys), _stream()))


def trans_keys_merge(keys, gathered, parts):
    L = _lib.require_gpu()
    assert gathered.is_contiguous() and gathered.numel() == parts * keys.numel()
    _lib.check(L.tracs_trans_keys_merge(_ptr(keys), _ptr(gathered), int(parts), _stream()))


def trans_keys_info(keys):
    """-> (distinct keys, largest distance, span of the days, fits the grid) of a (merged) key bitmap; synchronises"""
    L = _lib.require_gpu()
    info = (C.c_uint64 * 4)()
    _lib.check(L.tracs_trans_keys_info(_ptr(keys), info, _stream()))
    return tuple(int(x) for x in info)


def trans_keys_evaluate(keys, info, part, parts, lamb, beta, threshold_Ek, vals):
    """vals (torch.float64 [per, 2]) <- (log p0, E(K)) of the keys with ordinal = part mod parts, at slot ordinal // parts"""
    L = _lib.require_gpu()
    assert vals.is_contiguous() and vals.dtype == torch.float64
    arr = (C.c_uint64 * 4)(*info)
    _lib.check(L.tracs_trans_keys_evaluate(_ptr(keys), arr, int(part), int(parts), float(lamb), float(beta), float(threshold_Ek),
                                           _ptr(vals), vals.numel() // 2, _stream()))


def trans_keys_gather(dist, n, days, ranges, keys, info, vals_all, parts, p0, eK, exp_p0=True, dist_threshold=2147483647, col_begin=0):
    L = _lib.require_gpu()
    ld = dist.stride(0)
    assert p0.stride(0) == ld and eK.stride(0) == ld and len(ranges) <= 2 and vals_all.is_contiguous()
    arr, k = _ranges_arg(ranges)
    inf = (C.c_uint64 * 4)(*info)
    _lib.check(L.tracs_trans_keys_gather(_ptr(dist), ld, n, arr, k, col_begin, int(dist_threshold), _ptr(days), _ptr(keys), inf,
                                         _ptr(vals_all), int(parts), vals_all.numel() // (2 * parts), int(exp_p0), _ptr(p0), _ptr(eK), _stream()))


def calculate_posteriors_device(counts, alphas, keep, threshold):
    L = _lib.require_gpu()
    a = np.ascontiguousarray(alphas, dtype=np.float64)
    out = torch.empty_like(counts)
    _lib.check(L.tracs_calculate_posteriors_device(_ptr(counts), counts.shape[0], counts.shape[1],
                                                   a.ctypes.data_as(C.POINTER(C.c_double)), int(bool(keep)),
                                                   float(threshold), _ptr(out), _stream()))
    return out


def posterior_codes_device(counts_u16, alphas, keep, threshold, min_cov=0, cov_band=None, out=None):
    """counts_u16: torch.int16/uint16 [L,4] (or int32 [L,4] for depths above 65535) -> torch.uint8 [(L+1)//2] packed allele
    masks (low nibble = even site).  min_cov / cov_band=(lo, hi): the align stage's coverage rules (sites below min_cov or
    inside the band become N).  out: a contiguous uint8 view of at least (L+1)//2 bytes to write into (e.g. the sample's row
    of a batch buffer for Alignment.pack_codes)."""
    L = _lib.require_gpu()
    a = np.ascontiguousarray(alphas, dtype=np.float64)
    n = counts_u16.shape[0]
    if out is None:
        out = torch.empty((n + 1) // 2, dtype=torch.uint8, device=counts_u16.device)
    else:
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= (n + 1) // 2 and out.device == counts_u16.device
    lo, hi = cov_band if cov_band is not None else (1.0, 0.0)
    assert counts_u16.is_contiguous() and counts_u16.element_size() in (2, 4)
    fn = L.tracs_posterior_codes_cov_device32 if counts_u16.element_size() == 4 else L.tracs_posterior_codes_cov_device
    _lib.check(fn(_ptr(counts_u16), n, a.ctypes.data_as(C.POINTER(C.c_double)), int(bool(keep)), float(threshold), int(min_cov),
                  float(lo), float(hi), _ptr(out), _stream()))
    return out


def codes_to_iupac_device(codes, L):
    """packed 4-bit masks -> torch.uint8 [L] IUPAC letters ('X' for an empty mask), tracs/align.py:285-323."""
    lib = _lib.require_gpu()
    out = torch.empty(L, dtype=torch.uint8, device=codes.device)
    _lib.check(lib.tracs_codes_to_iupac_device(_ptr(codes), L, _ptr(out), _stream()))
    return out


def connected_components_device(I, J, n_nodes):
    L = _lib.require_gpu()
    labels = torch.empty(n_nodes, dtype=torch.int32, device=I.device)
    nc = C.c_int32(0)
    _lib.check(L.tracs_connected_components_device(_ptr(I), _ptr(J), I.numel(), n_nodes, _ptr(labels), C.byref(nc), _stream()))
    return int(nc.value), labels


def tri_pack(mat, n, row_begin, row_end, col_begin, slots, width, base, negate, packed_ptr, stats, base_row=0):
    """csrc/exchange.hip: the cells (i, j >= max(col_begin, i + 1)) of rows [row_begin, row_end) of the uint32 panel `mat` (rows
    base_row..) -> `width` bytes per cell at element slots[i - row_begin] of the buffer at packed_ptr (None: only the statistics);
    stats: int32[2] (largest value, values beyond 16 bits)."""
    L = _lib.require_gpu()
    assert slots.dtype == torch.int64 and slots.is_contiguous() and slots.numel() == row_end - row_begin
    _lib.check(L.tracs_tri_pack(_panel_ptr(mat, base_row), int(mat.stride(0)), int(n), int(row_begin), int(row_end), int(col_begin), _ptr(slots),
                                int(width), int(base) & 0xFFFFFFFF, int(negate), C.c_void_p(packed_ptr or 0), _ptr(stats), _stream()))


def tri_sum(mat, n, row_begin, row_end, col_begin, slots, width, recv_ptr, block_elems, n_blocks, skip_block, add, negate, base_row=0):
    """csrc/exchange.hip: mat[i][j] += add +/- the sum over the blocks b != skip_block of recv[b * block_elems + slots[i - row_begin] + j - first
    column] for the rows whose slot is not -1."""
    L = _lib.require_gpu()
    assert slots.dtype == torch.int64 and slots.is_contiguous() and slots.numel() == row_end - row_begin
    _lib.check(L.tracs_tri_sum(_panel_ptr(mat, base_row), int(mat.stride(0)), int(n), int(row_begin), int(row_end), int(col_begin), _ptr(slots),
                               int(width), C.c_void_p(recv_ptr), int(block_elems), int(n_blocks), int(skip_block), int(add) & 0xFFFFFFFF,
                               int(negate), _stream()))
