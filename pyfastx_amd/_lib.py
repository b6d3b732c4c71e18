"""ctypes binding of libfxgpu.so (include/fxgpu.h) -- the only way pyfastx_amd
reaches the GPU.  There is no CPU fallback: if the library is missing, or no
gfx950 device is usable, the calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("FX_LIBFXGPU") or os.path.join(_HERE, "csrc", "libfxgpu.so")   # the override is for kernel experiments (tools/)
_LIB = None

FX_HOST, FX_DEVICE = 0, 1
FX_UPPER, FX_REVERSE, FX_COMPLEMENT, FX_RAW = 1, 2, 4, 8
FX_OK, FX_ENOENT, FX_EFORMAT, FX_EIO, FX_EDEVICE, FX_ENOMEM, FX_ERANGE, FX_EINVAL, FX_ESTATE = \
    0, -1, -2, -3, -4, -5, -6, -7, -8


class FxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class FastaSummary(C.Structure):
    _fields_ = [("n_seq", C.c_int64), ("seq_len", C.c_int64), ("n_lines", C.c_int64), ("n_bytes", C.c_int64)]


class FastqSummary(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("size", C.c_int64), ("n_lines", C.c_int64), ("n_bytes", C.c_int64),
                ("first_id", C.c_int64)]


class LenStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_seq", "sum_len", "longest_id", "longest_len", "shortest_id", "shortest_len",
                                         "count_ge", "med_lo", "med_hi", "nx_len", "nx_count")]


class ShardSummary(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "base", "n_bytes", "is_last", "n_nl", "first_nl", "second_nl", "last_nl", "first_nl_prev",
        "first_byte", "last_byte", "n_hdr", "first_hdr", "last_hdr", "lead_nl", "lead_ws",
        "lead_v1", "lead_c1", "lead_v2", "lead_c2", "tail_e", "tail_first_end", "tail_nl_after", "tail_bad",
        "tail_elen", "tail_dlen", "tail_name_len", "lead_prev_nl", "second_last_nl")]


SYMBOLS = [
    "fx_last_error", "fx_version", "fx_device_count", "fx_open_file", "fx_open_laps", "fx_build_laps", "fx_open_file_indexed", "fx_gz_checkpoints", "fx_stream_size", "fx_open_file_range", "fx_open_host", "fx_open_device",
    "fx_set_shard", "fx_close", "fx_release_scratch", "fx_pinned_alloc", "fx_pinned_free", "fx_pinned_holds", "fx_pinned_trim", "fx_size", "fx_device_memory", "fx_is_gzip", "fx_device_ptr", "fx_read_bytes", "fx_first_byte",
    "fx_fasta_build", "fx_fasta_build_begin", "fx_fasta_build_end", "fx_fasta_table", "fx_fasta_set_table", "fx_fasta_line_regular", "fx_fasta_len_stats", "fx_fasta_comp", "fx_fasta_comp_shard", "fx_fasta_comp_sparse", "fx_fastq_build", "fx_fastq_build_comp", "fx_fastq_comp_info", "fx_set_halo", "fx_fastq_scan", "fx_fastq_build_ctx", "fx_fastq_table", "fx_fastq_comp",
    "fx_fetch_ranges", "fx_fetch_slices", "fx_fetch_one", "fx_fasta_fetch", "fx_fasta_fetch_alloc", "fx_fetch_phases", "fx_fastq_fetch", "fx_fastq_fetch_alloc", "fx_names_build", "fx_names_lookup", "fx_names_sort", "fx_names_pack", "fx_revcomp", "fx_shard_summary_get",
    "fx_fasta_set_row", "fx_shard_route", "fx_shard_summary_dev", "fx_fasta_stitch_dev", "fx_stream", "fx_read_fetch", "fx_gz_points", "fx_fxi_bulk_rows", "fx_fxi_bulk_index", "fx_fxi_bulk_index_int", "fx_fxi_dev_sort", "fx_fxi_dev_write", "fx_fxi_dev_build", "fx_fxi_presize_begin", "fx_fxi_presize_end", "fx_fxi_part_shape", "fx_fxi_part_firsts", "fx_fxi_part_names", "fx_fxi_part_leaves", "fx_fxi_join_grow", "fx_fxi_join_begin", "fx_fxi_join_write", "fx_fxi_join_end", "fx_scratch_policy", "fx_open_file_async", "fx_stage_wait", "fx_sync", "fx_prof_default", "fx_prof_enable", "fx_prof_reset", "fx_prof_count", "fx_prof_name", "fx_prof_read",
    "fx_comm_unique_id", "fx_comm_init", "fx_comm_destroy", "fx_comm_rank", "fx_comm_world", "fx_comm_allgather", "fx_fasta_build_sharded_begin",
    "fx_fasta_build_sharded", "fx_comm_summaries", "fx_fastq_build_sharded", "fx_bgzf_counts", "fx_sort_packed_names", "fx_gunzip_parallel", "fx_gz_open_mode", "fx_kseq_scan", "fx_kseq_records", "fx_kseq_fetch", "fx_kseq_prefix_lines",
]


def so_path():
    return _SO


def open_laps():
    """(device allocation, page cache -> HBM) seconds of this thread's last fx_open_file of a plain file."""
    a, b = C.c_double(0.0), C.c_double(0.0)
    lib().fx_open_laps(C.byref(a), C.byref(b))
    return float(a.value), float(b.value)


def build_laps():
    """The parts of this thread's last fx_fastq_build, seconds (fx_build_laps)."""
    a = (C.c_double * 8)()
    lib().fx_build_laps(a)
    return {"sample": float(a[0]), "count_pass_enqueue": float(a[1]), "count_pass_wait": float(a[2]), "table_alloc": float(a[4]), "rows": float(a[5])}


def fxi_presize_begin(path, nbytes, device=-1):
    """A library thread (on the CPUs next to `device`) grows the (just created) index file to nbytes with fallocate -> token for
    fxi_presize_end."""
    tok = C.c_void_p(None)
    check(lib().fx_fxi_presize_begin(os.fsencode(path), int(nbytes), int(device), C.byref(tok)))
    return tok


def fxi_join_grow(path, root_table, nleaf_table, device, extra_bytes=0):
    """Room for the table's new pages and extra_bytes behind them (best effort) -> the page number the parts' leaves begin at."""
    first = C.c_int64(0)
    check(lib().fx_fxi_join_grow(os.fsencode(path), int(root_table), int(nleaf_table), int(extra_bytes), int(device), C.byref(first)))
    return int(first.value)


class FxiJoin:
    """The writer's side of an index file that several handles fill (fx_fxi_join_*): the names of all parts, in part order,
    back to back in device memory (d_names, d_lens: pointers as integers), sorted once."""
    LAPS = ("file_grown", "index_shape", "index_kernels", "index_to_file", "host_levels_and_header", "table_interior")

    def __init__(self, device, d_names, d_lens, n):
        self._j, nd = C.c_void_p(), C.c_int64(0)
        check(lib().fx_fxi_join_begin(int(device), C.c_void_p(int(d_names or 0)), C.c_void_p(int(d_lens or 0)), int(n), C.byref(self._j), C.byref(nd)))
        self.n_dup = int(nd.value)

    def write(self, path, root_table, root_index, n_rows, first_rows, first_new_page):
        first_rows = np.ascontiguousarray(first_rows, dtype=np.int64)
        laps = (C.c_double * 6)()
        check(lib().fx_fxi_join_write(self._j, os.fsencode(path), int(root_table), int(root_index), int(n_rows), int(first_rows.size),
                                      _ptr(first_rows) if first_rows.size else None, int(first_new_page), laps))
        return dict(zip(self.LAPS, (float(x) for x in laps)))

    def close(self):
        if self._j:
            lib().fx_fxi_join_end(self._j)
            self._j = C.c_void_p()

    __del__ = close


def fxi_presize_end(token, cancel=False):
    if token is not None and token.value:
        lib().fx_fxi_presize_end(token, 1 if cancel else 0)
        token.value = None


def fxi_bulk_rows(path, rootpage, packed_names, name_off, cols):
    """Host-side bulk load of an index table (fx_fxi_bulk_rows): packed_names uint8, name_off int64[n+1],
    cols: list of int64 arrays.  packed_names = name_off = None: a table without a TEXT column (comp).
    The database file must have no open connection."""
    arrs = [np.ascontiguousarray(c, dtype=np.int64) for c in cols]
    ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
    if name_off is None:
        check(lib().fx_fxi_bulk_rows(os.fsencode(path), int(rootpage), arrs[0].size if arrs else 0, None, None, len(arrs), ptrs))
        return
    names = np.ascontiguousarray(packed_names, dtype=np.uint8)
    offs = np.ascontiguousarray(name_off, dtype=np.int64)
    check(lib().fx_fxi_bulk_rows(os.fsencode(path), int(rootpage), offs.size - 1, names.ctypes.data if names.size else None,
                                 offs.ctypes.data, len(arrs), ptrs))


def fxi_bulk_index(path, rootpage, packed_names, name_off, order):
    """Host-side bulk load of the UNIQUE INDEX on the name column (fx_fxi_bulk_index): names in row order,
    order[i] = row of the i-th smallest name."""
    names = np.ascontiguousarray(packed_names, dtype=np.uint8)
    offs = np.ascontiguousarray(name_off, dtype=np.int64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    if order.size != offs.size - 1:
        raise ValueError("order must have one entry per row")
    check(lib().fx_fxi_bulk_index(os.fsencode(path), int(rootpage), offs.size - 1, names.ctypes.data if names.size else None,
                                  offs.ctypes.data, order.ctypes.data if order.size else None))


def fxi_bulk_index_int(path, rootpage, key, order):
    """Host-side bulk load of a non-unique INDEX on an INTEGER column (fx_fxi_bulk_index_int)."""
    key = np.ascontiguousarray(key, dtype=np.int64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    if key.size != order.size:
        raise ValueError("one key and one order entry per row")
    check(lib().fx_fxi_bulk_index_int(os.fsencode(path), int(rootpage), key.size, key.ctypes.data if key.size else None,
                                      order.ctypes.data if order.size else None))


def lib():
    """Load libfxgpu.so (after torch, if torch is importable, so both share one HIP runtime)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise ImportError(
            "pyfastx_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pyfastx_amd/csrc`.  There is no CPU fallback." % _SO)
    if not os.environ.get("FX_NO_TORCH"):                  # FX_NO_TORCH=1: the library alone, on the system's ROCm (no torch in the process)
        try:
            import torch  # noqa: F401  (loads torch's bundled libamdhip64 first; ours then binds to the same runtime)
        except Exception:
            pass
    L = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.fx_last_error.restype = C.c_char_p
    L.fx_version.restype = C.c_char_p
    L.fx_device_count.restype = i32
    L.fx_open_file.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.fx_open_host.argtypes = [vp, i64, i32, C.POINTER(vp)]
    L.fx_stream_size.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(i32)]
    L.fx_open_file_indexed.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, i64, C.POINTER(vp)]
    L.fx_gz_checkpoints.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    L.fx_open_file_range.argtypes = [C.c_char_p, i64, i64, i64, i32, C.POINTER(vp)]
    L.fx_open_device.argtypes = [vp, i64, i32, C.POINTER(vp)]
    L.fx_shard_route.argtypes = [i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp] + [vp] * 8
    L.fx_set_shard.argtypes = [vp, i64, i32, i32]
    L.fx_close.argtypes = [vp]
    L.fx_size.restype = i64
    L.fx_size.argtypes = [vp]
    L.fx_is_gzip.argtypes = [vp]
    L.fx_device_memory.argtypes = [i32, C.POINTER(i64), C.POINTER(i64)]
    L.fx_device_ptr.restype = vp
    L.fx_device_ptr.argtypes = [vp]
    L.fx_read_bytes.argtypes = [vp, i64, i64, vp]
    L.fx_first_byte.argtypes = [vp, C.POINTER(i32)]
    L.fx_fasta_build.argtypes = [vp, i32, C.POINTER(FastaSummary)]
    L.fx_fasta_build_begin.argtypes = [vp, i32]
    L.fx_fasta_build_end.argtypes = [vp, C.POINTER(FastaSummary)]
    L.fx_fasta_table.argtypes = [vp, i32] + [vp] * 9
    L.fx_fasta_set_table.argtypes = [vp, i64] + [vp] * 6
    L.fx_fasta_line_regular.argtypes = [vp, i32, vp]
    L.fx_fasta_len_stats.argtypes = [vp, i64, C.c_double, C.POINTER(LenStats)]
    L.fx_fasta_comp.argtypes = [vp, i32, vp]
    L.fx_fasta_comp_shard.argtypes = [vp, i32, vp, i64, vp]
    L.fx_fasta_comp_sparse.argtypes = [vp, i32, i64, vp, vp, vp, C.POINTER(i64), vp]
    L.fx_fastq_build.argtypes = [vp, C.POINTER(FastqSummary)]
    L.fx_fastq_build_comp.argtypes = [vp, C.POINTER(FastqSummary)]
    L.fx_fastq_table.argtypes = [vp, i32] + [vp] * 6
    L.fx_set_halo.argtypes = [vp, i64]
    L.fx_fastq_scan.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.fx_fastq_build_ctx.argtypes = [vp, i64, i64, C.POINTER(FastqSummary)]
    L.fx_fastq_comp.argtypes = [vp, vp, vp]
    L.fx_fetch_one.argtypes = [vp, i64, i64, i64, i64, i32, vp, C.POINTER(i64)]
    L.fx_fetch_ranges.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp, vp, vp]
    L.fx_fasta_fetch.argtypes = [vp, i32, i64, vp, vp, vp, i32, vp, vp, vp, vp]
    L.fx_fetch_slices.argtypes = [vp, i32, i64, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.fx_fastq_fetch.argtypes = [vp, i32, i64, vp, i32, i32, vp, vp, vp, vp]
    L.fx_pinned_alloc.restype = vp
    L.fx_pinned_alloc.argtypes = [i64]
    L.fx_pinned_free.restype = None
    L.fx_pinned_free.argtypes = [vp]
    L.fx_pinned_holds.argtypes = [vp, i64]
    L.fx_fasta_fetch_alloc.argtypes = [vp, i64, vp, vp, vp, i32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.fx_fastq_fetch_alloc.argtypes = [vp, i64, vp, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.fx_fetch_phases.argtypes = [C.POINTER(C.c_double), i32]
    L.fx_names_build.argtypes = [vp, i32]
    L.fx_names_lookup.argtypes = [vp, i32, i64, vp, vp, vp]
    L.fx_revcomp.argtypes = [i32, i32, vp, i64, i32]
    L.fx_read_fetch.argtypes = [vp, i32, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.fx_shard_summary_get.argtypes = [vp, C.POINTER(ShardSummary)]
    L.fx_fasta_set_row.argtypes = [vp, i64, i64, i64, i64, i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.fx_shard_summary_dev.argtypes = [vp, vp]
    L.fx_fasta_stitch_dev.argtypes = [vp, vp, i32, i32, i32]
    L.fx_stream.restype = vp
    L.fx_stream.argtypes = [vp]
    L.fx_gz_points.argtypes = [vp, i64, vp, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.fx_names_sort.argtypes = [vp, i32, i32, vp, C.POINTER(i64)]
    L.fx_names_pack.argtypes = [vp, i32, vp, i64, vp, C.POINTER(i64)]
    L.fx_fxi_bulk_rows.argtypes = [C.c_char_p, i32, i64, vp, vp, i32, vp]
    L.fx_fxi_bulk_index.argtypes = [C.c_char_p, i32, i64, vp, vp, vp]
    L.fx_fxi_bulk_index_int.argtypes = [C.c_char_p, i32, i64, vp, vp]
    L.fx_open_laps.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.fx_build_laps.argtypes = [C.POINTER(C.c_double)]
    L.fx_fastq_comp_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.fx_fxi_dev_sort.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    L.fx_fxi_dev_write.argtypes = [vp, i32, C.c_char_p, i32, i32, C.POINTER(C.c_double)]
    L.fx_fxi_dev_build.argtypes = [vp, i32, C.c_char_p, i32, i32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.fx_fxi_presize_begin.argtypes = [C.c_char_p, i64, i32, C.POINTER(vp)]
    L.fx_fxi_presize_end.argtypes = [vp, i32]
    L.fx_fxi_part_shape.argtypes = [vp, i32, i64, C.POINTER(C.c_int64)]
    L.fx_fxi_part_firsts.argtypes = [vp, vp]
    L.fx_fxi_part_names.argtypes = [vp, i32, vp, vp]
    L.fx_fxi_part_leaves.argtypes = [vp, i32, C.c_char_p, i64, i64, C.POINTER(C.c_double)]
    L.fx_fxi_join_grow.argtypes = [C.c_char_p, i32, i64, i64, i32, C.POINTER(C.c_int64)]
    L.fx_fxi_join_begin.argtypes = [i32, vp, vp, i64, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.fx_fxi_join_write.argtypes = [vp, C.c_char_p, i32, i32, i64, i64, vp, i64, C.POINTER(C.c_double)]
    L.fx_fxi_join_end.argtypes = [vp]
    L.fx_fxi_join_end.restype = None
    L.fx_scratch_policy.argtypes = [vp, i32, i64, i64, vp]
    L.fx_open_file_async.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.fx_stage_wait.argtypes = [vp, i64]
    L.fx_sync.argtypes = [vp]
    L.fx_prof_enable.argtypes = [vp, i32]
    L.fx_prof_default.argtypes = [i32]
    L.fx_prof_reset.argtypes = [vp]
    L.fx_prof_name.restype = C.c_char_p
    L.fx_prof_name.argtypes = [i32]
    L.fx_prof_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    L.fx_comm_unique_id.argtypes = [vp]
    L.fx_comm_init.argtypes = [i32, i32, vp, i32, C.POINTER(vp)]
    L.fx_comm_destroy.argtypes = [vp]
    L.fx_comm_rank.argtypes = [vp]
    L.fx_comm_world.argtypes = [vp]
    L.fx_comm_allgather.argtypes = [vp, vp, vp, i64]
    L.fx_fasta_build_sharded_begin.argtypes = [vp, vp, i32]
    L.fx_fasta_build_sharded.argtypes = [vp, vp, i32, vp]
    L.fx_comm_summaries.argtypes = [vp, vp, vp]
    L.fx_fastq_build_sharded.argtypes = [vp, vp, vp]
    L.fx_bgzf_counts.argtypes = [vp, vp]
    L.fx_gz_open_mode.argtypes = [vp]
    L.fx_kseq_scan.argtypes = [vp, vp, vp, vp, vp]
    L.fx_kseq_records.argtypes = [vp, i64, i64, vp]
    L.fx_kseq_prefix_lines.argtypes = [vp]
    L.fx_kseq_prefix_lines.restype = i64
    L.fx_kseq_fetch.argtypes = [vp, i32, i64, i64, i32, vp, vp, vp]
    L.fx_gunzip_parallel.argtypes = [vp, i64, i32, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.fx_sort_packed_names.argtypes = [i32, vp, vp, i64, vp, C.POINTER(i64)]
    for s in SYMBOLS:
        if getattr(L, s).restype is C.c_int:
            pass
    _LIB = L
    return L


def sort_packed_names(packed, name_off, device=0):
    """fx_sort_packed_names: BINARY-collation order of names from several handles -> (order int64[n], duplicates)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    name_off = np.ascontiguousarray(name_off, dtype=np.int64)
    n = name_off.size - 1
    order = np.empty(max(n, 0), dtype=np.int64)
    nd = C.c_int64(0)
    check(lib().fx_sort_packed_names(int(device), _ptr(packed) if packed.size else None, _ptr(name_off), n, _ptr(order) if n > 0 else None, C.byref(nd)))
    return order, int(nd.value)


def gunzip_parallel(gz, threads=8):
    """fx_gunzip_parallel (host only): the inflated bytes of ONE gzip member and the number of restart points it found, or
    None when the parallel decoder declines (small input, several members, anything it is not sure of)."""
    src = np.frombuffer(gz, dtype=np.uint8)
    n, npts = C.c_int64(0), C.c_int64(0)
    cap = max(len(gz) * 8, 1 << 20)
    for _ in range(2):
        out = np.empty(cap, dtype=np.uint8)
        rc = lib().fx_gunzip_parallel(src.ctypes.data, src.size, int(threads), out.ctypes.data, cap, C.byref(n), C.byref(npts))
        if rc == 1:
            return None
        if rc == FX_ERANGE and n.value > cap:
            cap = int(n.value)
            continue
        check(rc)
        return out[:n.value].tobytes(), int(npts.value)
    raise FxError(FX_ERANGE, "output did not fit twice")


def device_memory(device=0):
    """-> (free, total) bytes of the device's HBM now."""
    f, t = C.c_int64(0), C.c_int64(0)
    check(lib().fx_device_memory(int(device), C.byref(f), C.byref(t)))
    return f.value, t.value


def stream_size(path):
    """-> (bytes of the uncompressed stream or -1, kind): kind 0 plain, 1 BGZF, 2 single-stream gzip (fx_stream_size)."""
    n, k = C.c_int64(0), C.c_int(0)
    check(lib().fx_stream_size(os.fsencode(path), C.byref(n), C.byref(k)))
    return int(n.value), int(k.value)


def shard_route(ids, starts, stops, cols, bases, ends, flags=0, flags_per_query=None):
    """fx_shard_route: a batch of (record, start, stop) over the byte-range shards [bases[r], ends[r]) -> dict in ROUTED
    order (by answering shard, then by position in the batch): order, shard_start[n_shard + 1], off, len, skip, take,
    fl, cnt.  cols: the global table as contiguous arrays boff, blen, llen, elen (int64) and reg (uint8)."""
    ids, a, b = (np.ascontiguousarray(x, dtype=np.int64) for x in (ids, starts, stops))
    n, G = ids.size, len(bases)
    bases, ends = np.ascontiguousarray(bases, dtype=np.int64), np.ascontiguousarray(ends, dtype=np.int64)
    fpq = None if flags_per_query is None else np.ascontiguousarray(flags_per_query, dtype=np.uint8)
    out = {k: np.empty(n, dtype=np.int64) for k in ("order", "off", "len", "skip", "take")}
    out["shard_start"] = np.zeros(G + 1, dtype=np.int64)
    out["fl"], out["cnt"] = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.int32)
    check(lib().fx_shard_route(n, _ptr(ids), _ptr(a), _ptr(b), cols["boff"].size, _ptr(cols["boff"]), _ptr(cols["blen"]),
                               _ptr(cols["llen"]), _ptr(cols["elen"]), _ptr(cols["reg"]), G, _ptr(bases), _ptr(ends),
                               int(flags), _ptr(fpq), _ptr(out["order"]), _ptr(out["shard_start"]), _ptr(out["off"]),
                               _ptr(out["len"]), _ptr(out["skip"]), _ptr(out["take"]), _ptr(out["fl"]), _ptr(out["cnt"])))
    return out


def pinned_array(ptr, count, dtype=np.uint8):
    """A numpy array over `count` items of pinned memory that fx_pinned_alloc (or an fx_*_alloc entry) handed out; the
    block goes back to the library's pool when the last view of the array is gone (_fxobj.PinnedBuf owns it)."""
    from . import _fxobj
    L = lib()
    dt = np.dtype(dtype)
    owner = _fxobj.PinnedBuf(int(ptr), int(count) * dt.itemsize, C.cast(L.fx_pinned_free, C.c_void_p).value)
    return np.frombuffer(owner, dtype=dt, count=int(count))


def pinned_empty(count, dtype=np.uint8):
    """numpy array of `count` items in pinned host memory (fx_pinned_alloc): query arrays built in it go to the device
    without a staging copy, answers written into it arrive by DMA."""
    dt = np.dtype(dtype)
    p = lib().fx_pinned_alloc(max(int(count) * dt.itemsize, 1))
    if not p:
        raise FxError(FX_ENOMEM, lib().fx_last_error().decode())
    return pinned_array(p, count, dt)


def fetch_phases():
    """Host-side phases of this thread's last fx_*_fetch_alloc call, ms: (stage + upload, offsets, pinned blocks, launch, answers back, total)."""
    v = (C.c_double * 6)()
    check(lib().fx_fetch_phases(v, 6))
    return dict(zip(("stage_upload_ms", "offsets_wait_ms", "pinned_blocks_ms", "launch_ms", "answers_wait_ms", "call_ms"), [round(x, 3) for x in v]))


def check(rc):
    if rc != 0:
        raise FxError(rc, lib().fx_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data


class Comm:
    """The library's own communicator (fx_comm: RCCL bound at run time) -- one per process, one process per GPU.
    The 128-byte id comes from rank 0 (Comm.unique_id()) over whatever channel the launcher offers; share_id() uses the
    torch.distributed process group when there is one."""

    def __init__(self, rank, world, uid, device=0):
        h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(uid))
        check(lib().fx_comm_init(int(rank), int(world), buf, int(device), C.byref(h)))
        self._c, self.rank, self.world, self.device = h, int(rank), int(world), int(device)

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        check(lib().fx_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device):
        """rank / world of torch.distributed's default group; the id travels over it (one broadcast at setup)."""
        import torch.distributed as dist
        # Every collective below is entered by EVERY rank whatever happens on one of them (a rank that failed early and went on
        # to the next collective while the others still sat in this one hung the job): rank 0 always broadcasts something --
        # the id, or None when it could not make one -- and ncclCommInitRank, itself collective, is only reached when an id exists.
        uid = None
        if dist.get_rank() == 0:
            try:
                uid = cls.unique_id()
            except Exception:                                   # noqa: BLE001  (no RCCL to bind)
                uid = None
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise FxError(FX_EDEVICE, "rank 0 could not create a communicator id (RCCL not bound)")
        return cls(dist.get_rank(), dist.get_world_size(), box[0], device)

    def allgather(self, arr):
        """A small host array of every rank to every rank -> [world, ...] (fx_comm_allgather)."""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        check(lib().fx_comm_allgather(self._c, a.ctypes.data, out.ctypes.data, a.nbytes))
        return out

    def summaries(self, blob):
        from .shard import Summary, NWORDS
        out = np.zeros((self.world, NWORDS), dtype=np.int64)
        check(lib().fx_comm_summaries(self._c, blob._h, out.ctypes.data))
        return [Summary.from_array(r) for r in out]

    def close(self):
        if self._c:
            lib().fx_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


KSEQ_REC = np.dtype([("hdr_off", "<i8"), ("hdr_line", "<i8"), ("seq_len", "<i8"), ("seq_cum", "<i8"),
                     ("hdr_len", "<u4"), ("s_n", "<u4"), ("q_n", "<u4"), ("flags", "<u4")])


class Blob:
    """Owning wrapper of an fx_handle: one staged stream resident in HBM."""

    def __init__(self, handle):
        self.on_close = []
        self._h = handle

    # -- constructors -------------------------------------------------------
    @classmethod
    def from_file(cls, path, device=0, gzindex=None):
        """gzindex: the restart points of a single-stream gzip file (fxi.read_gzindex) -> the segments between them are
        inflated in parallel (fx_open_file_indexed)."""
        h = C.c_void_p()
        if gzindex and len(gzindex["cmp"]) and gzindex["windows"] is not None:
            a = [np.ascontiguousarray(gzindex[k], dtype=np.int64) for k in ("cmp", "uncmp")]
            b = [np.ascontiguousarray(gzindex[k], dtype=np.uint8) for k in ("bits", "has")]
            w = np.ascontiguousarray(gzindex["windows"], dtype=np.uint8)
            check(lib().fx_open_file_indexed(os.fsencode(path), device, a[0].size, _ptr(a[0]), _ptr(a[1]), _ptr(b[0]), _ptr(b[1]),
                                             _ptr(w) if w.size else None, int(gzindex["uncompressed_size"]), C.byref(h)))
        else:
            check(lib().fx_open_file(os.fsencode(path), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_file_range(cls, path, off, length, halo=0, device=0):
        """One byte-range shard of a file: bytes [off, off + length + halo) of the uncompressed stream (fx_open_file_range);
        the shard context (base, previous byte, is_last, halo) comes from the file."""
        h = C.c_void_p()
        check(lib().fx_open_file_range(os.fsencode(path), int(off), int(length), int(halo), device, C.byref(h)))
        b = cls(h)
        b.base = int(off)
        return b

    @classmethod
    def from_bytes(cls, data, device=0):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().fx_open_host(a.ctypes.data if a.size else None, a.size, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_file_async(cls, path, device=0):
        """A plain file staged in the background (fx_open_file_async): work on prefixes through views until stage_wait(-1)."""
        h = C.c_void_p()
        check(lib().fx_open_file_async(os.fsencode(path), device, C.byref(h)))
        return cls(h)

    def stage_wait(self, upto=-1):
        check(lib().fx_stage_wait(self._h, int(upto)))

    @classmethod
    def from_device(cls, dptr, nbytes, device=0, keepalive=None):
        h = C.c_void_p()
        check(lib().fx_open_device(C.c_void_p(dptr), nbytes, device, C.byref(h)))
        b = cls(h)
        b._keep = keepalive
        return b

    def close(self):
        if self._h:
            for f in self.__dict__.get("on_close", ()):          # owners that cached the raw handle (the C getters of api.Fasta / Fastq) let go of it first
                try:
                    f()
                except Exception:                                # noqa: BLE001
                    pass
            lib().fx_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- basics -------------------------------------------------------------
    @property
    def size(self):
        return lib().fx_size(self._h)

    @property
    def is_gzip(self):
        return bool(lib().fx_is_gzip(self._h))

    @property
    def device_ptr(self):
        return lib().fx_device_ptr(self._h)

    def set_shard(self, base, prev_byte, is_last):
        check(lib().fx_set_shard(self._h, base, prev_byte, int(is_last)))

    def read_bytes(self, off, n):
        out = np.empty(max(int(n), 0), dtype=np.uint8)
        if n > 0:
            check(lib().fx_read_bytes(self._h, int(off), int(n), out.ctypes.data))
        return out.tobytes()

    def first_byte(self):
        v = C.c_int(-1)
        check(lib().fx_first_byte(self._h, C.byref(v)))
        return v.value

    def gz_points(self, spacing=1048576):
        """-> (cmp_off int64[], uncmp_off int64[], compressed_size) restart points for the gzindex table."""
        n, cs = C.c_int64(0), C.c_int64(0)
        check(lib().fx_gz_points(self._h, spacing, None, None, 0, C.byref(n), C.byref(cs)))
        a = np.zeros(n.value, dtype=np.int64)
        b = np.zeros(n.value, dtype=np.int64)
        if n.value:
            check(lib().fx_gz_points(self._h, spacing, a.ctypes.data, b.ctypes.data, n.value, C.byref(n), C.byref(cs)))
        return a, b, cs.value

    @property
    def gz_open_mode(self):
        """0 plain, 1 BGZF on the device, 2 one gzip stream serially, 3 one stream on all host cores, 4 from restart points."""
        return int(lib().fx_gz_open_mode(self._h))

    # -- Fastx: the kseq walk -------------------------------------------------
    def kseq_scan(self):
        """kseq_read over the whole stream (fx_kseq_scan) -> (records, lines, sequence bytes, end code -1 / -2)."""
        nr, nl, sb, code = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(lib().fx_kseq_scan(self._h, C.byref(nr), C.byref(nl), C.byref(sb), C.byref(code)))
        return int(nr.value), int(nl.value), int(sb.value), int(code.value)

    def kseq_prefix_lines(self):
        """Lines of the last kseq_scan that the parallel passes took (the walk did the rest)."""
        return int(lib().fx_kseq_prefix_lines(self._h))

    def kseq_records(self, first, count):
        """Records [first, first + count) of the walk as a structured array (fx_kseq_rec)."""
        out = np.zeros(int(count), dtype=KSEQ_REC)
        if count:
            check(lib().fx_kseq_records(self._h, int(first), int(count), _ptr(out)))
        return out

    def kseq_fetch(self, first, count, nbytes, upper=False, want_qual=True):
        """The sequence (and quality) strings of records [first, first + count), one behind the other: nbytes =
        seq_cum + seq_len of the last minus seq_cum of the first."""
        seq = np.empty(max(int(nbytes), 1), dtype=np.uint8)
        qual = np.empty(max(int(nbytes), 1), dtype=np.uint8) if want_qual else None
        got = C.c_int64(0)
        check(lib().fx_kseq_fetch(self._h, FX_HOST, int(first), int(count), FX_UPPER if upper else 0, _ptr(seq), _ptr(qual), C.byref(got)))
        if got.value != int(nbytes):
            raise RuntimeError("fx_kseq_fetch: %d bytes, expected %d" % (got.value, int(nbytes)))
        return seq[:int(nbytes)], (qual[:int(nbytes)] if want_qual else None)

    def bgzf_counts(self):
        """(members, members handed over to the serial decoder, reason of the first) of the open that made this blob."""
        a = (C.c_int64 * 3)()
        check(lib().fx_bgzf_counts(self._h, a))
        return int(a[0]), int(a[1]), int(a[2])

    def gz_checkpoints(self):
        """Restart points captured while a single gzip stream was inflated (fx_gz_checkpoints) -> dict cmp, uncmp (int64),
        bits, has (uint8), windows (uint8 [n_with_data * 32768])."""
        n, nw = C.c_int64(0), C.c_int64(0)
        check(lib().fx_gz_checkpoints(self._h, 0, None, None, None, None, None, C.byref(n), C.byref(nw)))
        out = {"cmp": np.zeros(n.value, dtype=np.int64), "uncmp": np.zeros(n.value, dtype=np.int64),
               "bits": np.zeros(n.value, dtype=np.uint8), "has": np.zeros(n.value, dtype=np.uint8),
               "windows": np.zeros(nw.value * 32768, dtype=np.uint8)}
        if n.value:
            check(lib().fx_gz_checkpoints(self._h, n.value, _ptr(out["cmp"]), _ptr(out["uncmp"]), _ptr(out["bits"]), _ptr(out["has"]),
                                          _ptr(out["windows"]) if nw.value else None, C.byref(n), C.byref(nw)))
        return out

    def sync(self):
        check(lib().fx_sync(self._h))

    def prof_enable(self, on=1):
        """0 off, 1 every kernel, 2 only the dominant scan kernel (k_span_scan / k_scan)."""
        check(lib().fx_prof_enable(self._h, int(on)))

    def prof_reset(self):
        check(lib().fx_prof_reset(self._h))

    def prof_read(self):
        """-> {kernel name: (total ms, launches)} for kernels that ran."""
        out = {}
        for i in range(lib().fx_prof_count()):
            ms, cnt = C.c_double(0), C.c_int64(0)
            check(lib().fx_prof_read(self._h, i, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[lib().fx_prof_name(i).decode()] = (ms.value, cnt.value)
        return out

    # device-array variants (pointers are raw device addresses, e.g. tensor.data_ptr())
    def fasta_fetch_dev(self, n, seq_id, start, stop, dst, dst_off, flags=0, flags_per_query=0, out_len=0):
        check(lib().fx_fasta_fetch(self._h, FX_DEVICE, n, seq_id, start, stop, int(flags),
                                   flags_per_query or None, dst, dst_off, out_len or None))

    def fasta_table_dev(self, **ptrs):
        order = ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len")
        check(lib().fx_fasta_table(self._h, FX_DEVICE, *[ptrs.get(k) or None for k in order]))

    def fasta_comp_dev(self, ptr):
        check(lib().fx_fasta_comp(self._h, FX_DEVICE, ptr))

    def shard_summary(self):
        from .shard import Summary, FIELDS
        s = ShardSummary()
        check(lib().fx_shard_summary_get(self._h, C.byref(s)))
        return Summary((k, int(getattr(s, k))) for k in FIELDS)

    def shard_summary_dev(self, d_out):
        """Enqueue the 28-word boundary summary into a device buffer (no host round trip)."""
        check(lib().fx_shard_summary_dev(self._h, d_out))

    def fasta_stitch_dev(self, d_all, world, rank, full_name=False):
        """Enqueue the completion of this shard's last record from the gathered summaries (device)."""
        check(lib().fx_fasta_stitch_dev(self._h, d_all, int(world), int(rank), int(bool(full_name))))

    @property
    def stream(self):
        return lib().fx_stream(self._h)

    def fasta_set_row(self, k, boff, blen, slen, llen, elen, norm, dlen, name_len, reg=0):
        check(lib().fx_fasta_set_row(self._h, int(k), int(boff), int(blen), int(slen), int(llen), int(elen),
                                     (int(norm) & 1) | (int(bool(reg)) << 1), int(dlen), int(name_len)))

    # -- FASTA --------------------------------------------------------------
    def fasta_set_table(self, boff, blen, slen, llen, elen, norm):
        """Install the record table of an existing .fxi (no scan); fasta_fetch by id works afterwards."""
        a = [np.ascontiguousarray(x, dtype=np.int64) for x in (boff, blen, slen, llen)]
        b = [np.ascontiguousarray(x, dtype=np.int32) for x in (elen, norm)]
        check(lib().fx_fasta_set_table(self._h, a[0].size, *[_ptr(x) for x in a + b]))
        self._table_ready = True
        self._n_fasta = int(a[0].size)

    def fasta_build_begin(self, full_name=False, comp=False):
        """Enqueue the build and return (no host synchronisation); see fasta_build_end.  comp: the composition counters
        ride on the scan (one read of the stream for index + composition; fasta_comp* then only attribute them)."""
        self._table_ready = True
        self._n_fasta = None
        check(lib().fx_fasta_build_begin(self._h, int(bool(full_name)) | (2 if comp else 0)))

    def fasta_build_sharded_begin(self, comm, full_name=False, comp=False):
        """fx_fasta_build_sharded_begin: scan + tables + summary + ncclAllGather + stitch, enqueued on the handle's stream."""
        self._table_ready = True
        self._n_fasta = None
        check(lib().fx_fasta_build_sharded_begin(self._h, comm._c, int(bool(full_name)) | (2 if comp else 0)))

    def fastq_build_sharded(self, comm):
        s = FastqSummary()
        check(lib().fx_fastq_build_sharded(self._h, comm._c, C.byref(s)))
        return s

    def fasta_build_end(self):
        s = FastaSummary()
        check(lib().fx_fasta_build_end(self._h, C.byref(s)))
        self._n_fasta = int(s.n_seq)
        return s

    def fasta_build(self, full_name=False, comp=False):
        self._table_ready = True
        s = FastaSummary()
        check(lib().fx_fasta_build(self._h, int(bool(full_name)) | (2 if comp else 0), C.byref(s)))
        self._n_fasta = int(s.n_seq)
        return s

    def _check_rows(self, n, kind):
        """The C entries fill n_seq / n_reads rows whatever the caller allocated: refuse a wrong n here."""
        have = getattr(self, "_n_fasta" if kind == 0 else "_n_fastq", None)
        if have is None and kind == 0:
            have = int(self.fasta_build_end().n_seq)         # a build was only enqueued: wait for it, learn the count
        if have is not None and int(n) != have:
            raise ValueError("the table holds %d records, the caller asked for %d" % (have, int(n)))

    def fasta_table(self, n):
        self._check_rows(n, 0)
        cols = {k: np.empty(n, dtype=np.int64) for k in ("hoff", "boff", "blen", "slen", "llen")}
        cols.update({k: np.empty(n, dtype=np.int32) for k in ("elen", "norm", "dlen", "name_len")})
        check(lib().fx_fasta_table(self._h, FX_HOST, *[_ptr(cols[k]) for k in (
            "hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len")]))
        return cols

    def fasta_len_stats(self, count_min=0, half=0.0):
        """Statistics of the record lengths from the table in HBM (fx_fasta_len_stats) -> LenStats."""
        st = LenStats()
        check(lib().fx_fasta_len_stats(self._h, int(count_min), float(half), C.byref(st)))
        return st

    def fasta_line_regular(self, n):
        """int32[n]: 1 where slices of the record may use the line arithmetic (fx_fasta_line_regular)."""
        self._check_rows(n, 0)
        reg = np.zeros(n, dtype=np.int32)
        if n:
            check(lib().fx_fasta_line_regular(self._h, FX_HOST, reg.ctypes.data))
        return reg

    def fasta_comp(self, n):
        self._check_rows(n, 0)
        comp = np.zeros((n, 128), dtype=np.int64)
        check(lib().fx_fasta_comp(self._h, FX_HOST, comp.ctypes.data))
        return comp

    def fasta_comp_sparse(self, guess=0):
        """-> (seqid, abc, num int64 arrays: the non-zero bins in record order, seqid 1-based; total int64[128])."""
        total = np.zeros(128, dtype=np.int64)
        cap = int(guess)
        for _ in range(2):
            arrs = [np.empty(max(cap, 1), dtype=np.int64) for _ in range(3)]
            nout = C.c_int64(0)
            rc = lib().fx_fasta_comp_sparse(self._h, FX_HOST, cap, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                            C.byref(nout), total.ctypes.data)
            if rc == FX_ERANGE and nout.value > cap:
                cap = int(nout.value)
                continue
            check(rc)
            return arrs[0][:nout.value], arrs[1][:nout.value], arrs[2][:nout.value], total
        raise FxError(FX_ERANGE, "composition did not fit twice")

    def fasta_comp_shard(self, n, lead_from):
        """-> (comp int64[n,128] of the records that start in this shard, lead int64[128]: the bytes before the
        shard's first header line from global offset lead_from on; lead_from < 0: not counted)."""
        self._check_rows(n, 0)
        comp = np.zeros((max(n, 1), 128), dtype=np.int64)
        lead = np.zeros(128, dtype=np.int64)
        check(lib().fx_fasta_comp_shard(self._h, FX_HOST, comp.ctypes.data, int(lead_from), lead.ctypes.data))
        return comp[:n], lead

    # -- FASTQ --------------------------------------------------------------
    def fastq_build(self, comp=False):
        """comp: count the composition on the way (fx_fastq_build_comp: index and base / meta in one read of the stream)."""
        s = FastqSummary()
        check((lib().fx_fastq_build_comp if comp else lib().fx_fastq_build)(self._h, C.byref(s)))
        self._n_fastq = int(s.n_reads)
        return s

    def set_halo(self, halo):
        check(lib().fx_set_halo(self._h, int(halo)))

    def fastq_scan(self):
        """phase 1 of a sharded FASTQ build -> (newlines in the core, offset of the last one or -1)"""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().fx_fastq_scan(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fastq_build_ctx(self, line_offset, prev_nl):
        s = FastqSummary()
        check(lib().fx_fastq_build_ctx(self._h, int(line_offset), int(prev_nl), C.byref(s)))
        self._n_fastq = int(s.n_reads)
        return s

    def fastq_table(self, n):
        self._check_rows(n, 1)
        cols = {k: np.empty(n, dtype=np.int64) for k in ("name_off", "rlen", "soff", "qoff")}
        cols.update({k: np.empty(n, dtype=np.int32) for k in ("name_len", "dlen")})
        check(lib().fx_fastq_table(self._h, FX_HOST, _ptr(cols["name_off"]), _ptr(cols["name_len"]),
                                   _ptr(cols["dlen"]), _ptr(cols["rlen"]), _ptr(cols["soff"]), _ptr(cols["qoff"])))
        return cols

    def fastq_comp(self):
        base = np.zeros(5, dtype=np.int64)
        meta = np.zeros(5, dtype=np.int64)
        check(lib().fx_fastq_comp(self._h, base.ctypes.data, meta.ctypes.data))
        return base, meta

    # -- names ----------------------------------------------------------------
    def names_build(self, kind):
        """kind 0: FASTA sequence names, 1: FASTQ read names -> id table in HBM."""
        check(lib().fx_names_build(self._h, int(kind)))

    def names_lookup(self, names):
        """list of str / bytes -- or the names pre-packed as a (bytes, int64 offsets[n + 1]) pair -- -> int64 ids (0-based,
        -1 when absent), one kernel launch."""
        if isinstance(names, tuple) and len(names) == 2 and not isinstance(names[1], (str, int)) and \
                isinstance(names[0], (bytes, bytearray, memoryview, np.ndarray)):
            offs = np.ascontiguousarray(np.frombuffer(names[1], dtype=np.int64) if not isinstance(names[1], np.ndarray) else names[1], dtype=np.int64)
            n = offs.size - 1
            out = np.empty(max(n, 0), dtype=np.int64)
            if n <= 0:
                return out
            raw = np.frombuffer(names[0], dtype=np.uint8) if not isinstance(names[0], np.ndarray) else names[0]
            total = int(offs[n])
            packed = raw if raw.size >= total + 8 else np.concatenate([raw[:total], np.zeros(16, dtype=np.uint8)])
            check(lib().fx_names_lookup(self._h, FX_HOST, n, _ptr(packed), _ptr(offs), _ptr(out)))
            return out
        n = len(names)
        out = np.empty(n, dtype=np.int64)
        if not n:
            return out
        from . import _fxobj
        try:                                                    # one C pass over the list: sizes, then the bytes back to back (+ 16 zero bytes)
            pb, po = _fxobj.pack_names(names)
            packed, offs = np.frombuffer(pb, dtype=np.uint8), np.frombuffer(po, dtype=np.int64)
        except ValueError:                                      # a name with surrogate escapes: encode one by one
            enc = [x if isinstance(x, bytes) else x.encode("utf-8", "surrogateescape") for x in names]
            offs = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=n), out=offs[1:])
            packed = np.frombuffer(b"".join(enc) + b"\0" * 16, dtype=np.uint8)
        check(lib().fx_names_lookup(self._h, FX_HOST, n, _ptr(packed), _ptr(offs), _ptr(out)))
        return out

    def names_pack(self, kind, n, guess=0):
        """-> (packed uint8, name_off int64[n+1]): the names of the n records back to back, from the table in HBM."""
        offs = np.zeros(int(n) + 1, dtype=np.int64)
        total = C.c_int64(0)
        cap = int(guess)
        for _ in range(2):
            buf = np.empty(max(cap, 1), dtype=np.uint8)
            rc = lib().fx_names_pack(self._h, int(kind), buf.ctypes.data, cap, offs.ctypes.data, C.byref(total))
            if rc == FX_ERANGE and total.value > cap:
                cap = int(total.value)
                continue
            check(rc)
            return buf[:total.value], offs
        raise FxError(FX_ERANGE, "names did not fit twice")

    def fastq_comp_info(self):
        """How the last fastq_build(comp=True) counted -> (runs, runs counted again from the prefixes or -1, one-read result in use)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(lib().fx_fastq_comp_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), bool(c.value)

    def fxi_dev_sort(self, kind):
        """The sorted order of the record names, computed and KEPT on the device for fxi_dev_write -> number of adjacent
        equal pairs (0: the names are distinct and the UNIQUE INDEX may be written)."""
        ndup = C.c_int64(0)
        check(lib().fx_fxi_dev_sort(self._h, int(kind), C.byref(ndup)))
        return int(ndup.value)

    FXI_LAPS = ("table_shape", "table_kernels", "table_to_file", "file_grown", "index_shape", "index_kernels", "index_to_file", "host_levels_and_header")

    def fxi_dev_write(self, kind, path, root_table, root_index=0):
        """The big table of a new index file (and, root_index != 0, the UNIQUE INDEX on its name column) as b-tree pages
        formatted on the device (fx_fxi_dev_write) -> dict of the phases in seconds."""
        laps = (C.c_double * 8)()
        check(lib().fx_fxi_dev_write(self._h, int(kind), os.fsencode(path), int(root_table), int(root_index), laps))
        return dict(zip(self.FXI_LAPS, (float(x) for x in laps)))

    # -- one index file from several handles (fx_fxi_part_*: this handle's part) --
    def fxi_part_shape(self, kind, row_base):
        """-> (rows, table leaves, bytes of names) of this handle's part of the table; rowids start at row_base + 1."""
        out = (C.c_int64 * 3)()
        check(lib().fx_fxi_part_shape(self._h, int(kind), int(row_base), out))
        return int(out[0]), int(out[1]), int(out[2])

    def fxi_part_firsts(self, nleaf):
        first = np.empty(int(nleaf), dtype=np.int64)
        check(lib().fx_fxi_part_firsts(self._h, _ptr(first) if nleaf else None))
        return first

    def fxi_part_names(self, kind, d_names, d_lens):
        """Names back to back + their lengths into DEVICE memory (pointers as integers: torch data_ptr())."""
        check(lib().fx_fxi_part_names(self._h, int(kind), C.c_void_p(int(d_names)), C.c_void_p(int(d_lens))))

    def fxi_part_leaves(self, kind, path, first_new_page, leaf_base):
        laps = (C.c_double * 2)()
        check(lib().fx_fxi_part_leaves(self._h, int(kind), os.fsencode(path), int(first_new_page), int(leaf_base), laps))
        return {"table_kernels": float(laps[0]), "table_to_file": float(laps[1])}

    def fxi_dev_build(self, kind, path, root_table, root_index):
        """fxi_dev_sort + fxi_dev_write in one call, the sort and the index shape beside the table's copy-out (fx_fxi_dev_build)
        -> (n_dup, phases in seconds); n_dup > 0: the table alone was written, drop the empty index."""
        laps = (C.c_double * 8)()
        nd = C.c_int64(0)
        check(lib().fx_fxi_dev_build(self._h, int(kind), os.fsencode(path), int(root_table), int(root_index), C.byref(nd), laps))
        d = dict(zip(self.FXI_LAPS, (float(x) for x in laps)))
        d["sort_and_index_shape_not_hidden"] = d.pop("index_shape")
        return int(nd.value), d

    def names_sort(self, kind, n):
        """-> (order int64[n], n_dup): sorted order of the n record names (BINARY collation) computed on the GPU."""
        n = int(n)
        order = np.empty(n, dtype=np.int64)
        ndup = C.c_int64(0)
        check(lib().fx_names_sort(self._h, int(kind), FX_HOST, _ptr(order) if n else None, C.byref(ndup)))
        return order, int(ndup.value)

    # -- fetch (host arrays) ------------------------------------------------
    @staticmethod
    def _i64(a):
        return np.ascontiguousarray(a, dtype=np.int64)

    def fetch_one(self, off, blen, slen, flags=0, skip=0):
        """One range -> bytes (fx_fetch_one: one launch, one wait, no staging copies)."""
        slen = int(slen)
        if slen <= 65536:                                   # the getter path: one buffer per Blob, no allocation per call
            ob = self.__dict__.get("_one")
            if ob is None:
                ob = self.__dict__["_one"] = (C.create_string_buffer(65536), C.c_int64(0))
            buf, got = ob
        else:
            buf, got = C.create_string_buffer(slen), C.c_int64(0)
        rc = lib().fx_fetch_one(self._h, int(off), int(blen), int(skip), slen, int(flags), buf, C.byref(got))
        if rc:
            check(rc)
        return C.string_at(buf, got.value)

    def fetch_ranges(self, off, blen, slen, flags=0, flags_per_query=None, skip=None):
        """-> (uint8 buffer, offsets int64[n+1] (exclusive cumsum of slen), out_len int64[n]).  skip: kept bytes dropped
        in front of each answer (fx_fetch_slices: slice after despacing)."""
        off, blen, slen = self._i64(off), self._i64(blen), self._i64(slen)
        n = off.size
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(slen, 0), out=offs[1:])
        dst = np.empty(max(int(offs[-1]), 1), dtype=np.uint8)     # the copy back fills all of it (np.zeros would touch every page first)
        out_len = np.zeros(n, dtype=np.int64)
        fpq = None if flags_per_query is None else np.ascontiguousarray(flags_per_query, dtype=np.uint8)
        if n and skip is not None:
            skip = self._i64(skip)
            check(lib().fx_fetch_slices(self._h, FX_HOST, n, _ptr(off), _ptr(blen), _ptr(skip), _ptr(slen), int(flags),
                                        _ptr(fpq), _ptr(dst), _ptr(offs), _ptr(out_len)))
        elif n:
            check(lib().fx_fetch_ranges(self._h, FX_HOST, n, _ptr(off), _ptr(blen), _ptr(slen), int(flags),
                                        _ptr(fpq), _ptr(dst), _ptr(offs), _ptr(out_len)))
        return dst[:int(offs[-1])], offs, out_len

    def fasta_fetch(self, seq_id, start, stop, flags=0, flags_per_query=None):
        seq_id, start, stop = self._i64(seq_id), self._i64(start), self._i64(stop)
        n = seq_id.size
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(stop - start, 0), out=offs[1:])
        dst = np.empty(max(int(offs[-1]), 1), dtype=np.uint8)     # the copy back fills all of it (np.zeros would touch every page first)
        out_len = np.zeros(n, dtype=np.int64)
        fpq = None if flags_per_query is None else np.ascontiguousarray(flags_per_query, dtype=np.uint8)
        if n:
            check(lib().fx_fasta_fetch(self._h, FX_HOST, n, _ptr(seq_id), _ptr(start), _ptr(stop), int(flags),
                                       _ptr(fpq), _ptr(dst), _ptr(offs), _ptr(out_len)))
        return dst[:int(offs[-1])], offs, out_len

    def fasta_fetch_alloc(self, seq_id, start, stop, flags=0, flags_per_query=None):
        """(record id, start, stop) batches with the layout left to the library (fx_fasta_fetch_alloc): intervals checked on the
        device, answers and offsets in pinned memory -> (uint8 buffer, int64 offsets[n+1]); an invalid query raises
        FxError(FX_ERANGE) whose .first_bad is its index."""
        seq_id, start, stop = self._i64(seq_id), self._i64(start), self._i64(stop)
        n = seq_id.size
        if start.size != n or stop.size != n:
            raise ValueError("ids, starts and stops differ in length")
        fpq = None if flags_per_query is None else np.ascontiguousarray(flags_per_query, dtype=np.uint8)
        dst, offs, bad = C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        rc = lib().fx_fasta_fetch_alloc(self._h, n, _ptr(seq_id), _ptr(start), _ptr(stop), int(flags), _ptr(fpq),
                                        C.byref(dst), C.byref(offs), C.byref(bad))
        if rc:
            e = FxError(rc, lib().fx_last_error().decode())
            e.first_bad = int(bad.value)
            raise e
        o = pinned_array(offs.value, n + 2, np.int64)[:n + 1]
        return pinned_array(dst.value, max(int(o[n]), 1))[:int(o[n])], o

    def fastq_fetch_alloc(self, read_id, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        """Reads by id with the layout left to the library (fx_fastq_fetch_alloc) -> (seq, qual, quali, offsets), pinned."""
        read_id = self._i64(read_id)
        n = read_id.size
        w = (1 if "seq" in want else 0) | (2 if "qual" in want else 0) | (4 if "quali" in want else 0)
        ps, pq, pi, po, bad = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(-1)
        rc = lib().fx_fastq_fetch_alloc(self._h, n, _ptr(read_id), int(phred), int(seq_flags), w, C.byref(ps), C.byref(pq),
                                        C.byref(pi), C.byref(po), C.byref(bad))
        if rc:
            e = FxError(rc, lib().fx_last_error().decode())
            e.first_bad = int(bad.value)
            raise e
        o = pinned_array(po.value, n + 2, np.int64)[:n + 1]
        tot = int(o[n])
        seq = pinned_array(ps.value, max(tot, 1))[:tot] if ps.value else None
        qual = pinned_array(pq.value, max(tot, 1))[:tot] if pq.value else None
        qi = pinned_array(pi.value, max(tot, 1), np.int8)[:tot] if pi.value else None
        return seq, qual, qi, o

    def fastq_fetch(self, read_id, rlen, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        read_id = self._i64(read_id)
        rlen = self._i64(rlen)
        n = read_id.size
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rlen, out=offs[1:])
        tot = max(int(offs[-1]), 1)
        seq = np.zeros(tot, dtype=np.uint8) if "seq" in want else None
        qual = np.zeros(tot, dtype=np.uint8) if "qual" in want else None
        qi = np.zeros(tot, dtype=np.int8) if "quali" in want else None
        if n:
            check(lib().fx_fastq_fetch(self._h, FX_HOST, n, _ptr(read_id), int(phred), int(seq_flags),
                                       _ptr(seq), _ptr(qual), _ptr(qi), _ptr(offs)))
        return seq, qual, qi, offs

    def read_fetch(self, soff, qoff, rlen, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        soff, qoff, rlen = self._i64(soff), self._i64(qoff), self._i64(rlen)
        n = soff.size
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(rlen, out=offs[1:])
        tot = max(int(offs[-1]), 1)
        seq = np.zeros(tot, dtype=np.uint8) if "seq" in want else None
        qual = np.zeros(tot, dtype=np.uint8) if "qual" in want else None
        qi = np.zeros(tot, dtype=np.int8) if "quali" in want else None
        if n:
            check(lib().fx_read_fetch(self._h, FX_HOST, n, _ptr(soff), _ptr(qoff), _ptr(rlen), int(phred),
                                      int(seq_flags), _ptr(seq), _ptr(qual), _ptr(qi), _ptr(offs)))
        return seq, qual, qi, offs


def _noop():
    pass


def revcomp_bytes(b, mode=FX_REVERSE | FX_COMPLEMENT, device=0):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    if a.size:
        check(lib().fx_revcomp(device, FX_HOST, a.ctypes.data, a.size, mode))
    return a.tobytes()
