"""supersampler_amd -- Python host mirror of libspsp (include/spsp.h).

A thin ctypes layer over the C-ABI: every GPU entry point goes straight to the
HIP kernels in libspsp.so; there is no Python or CPU fallback.  Loading fails
loudly if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C supersampler_amd/csrc`).

If the process also uses PyTorch, import torch BEFORE this package: torch
ships its own libamdhip64.so.7 and the first HIP runtime loaded is the one both
share.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPSP_LIB: A/B experiments with another build (tools/ab_small.sh).  Whatever is loaded is reported: library_info() goes
# into bench.py's JSON line and the pytest header, and an override is announced on stderr when it is loaded.
LIB_OVERRIDDEN = bool(os.environ.get("SPSP_LIB"))
LIB_PATH = os.environ.get("SPSP_LIB") or os.path.join(_HERE, "libspsp.so")

SPSP_SCAN_DEFAULT = 0
SPSP_SCAN_DIRECT_HASH = 1
SPSP_SCAN_LDS_FILTER = 2
SPSP_SCAN_PAIR_FILTER = 4
SPSP_SCAN_STATS = 8
SPSP_SCAN_BLOOM_FILTER = 16
SPSP_SCAN_PACKED_INPUT = 32


class SpspError(RuntimeError):
    code = 0


ERR_OVERFLOW = -7
KEYS_UNORDERED = 1
TIME_DENSE, TIME_SCAN, TIME_ACCUMULATE, TIME_COMPARE, TIME_PARTS, TIME_ALL = 1, 2, 4, 8, 16, 31


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("m", C.c_uint32), ("threshold", C.c_uint64),
                ("abundance", C.c_uint32), ("flags", C.c_uint32)]


class SketchView(C.Structure):
    _fields_ = [("minimizer", C.c_void_p), ("kmer_lo", C.c_void_p), ("kmer_hi", C.c_void_p), ("n", C.c_uint64)]


class SketchStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("read_kmer", "selected_kmer_number", "selected_superkmer_number", "count_maximal_skmer",
                 "seen_kmers_at_reconstruction", "seen_superkmers_at_reconstruction",
                 "seen_max_superkmers_at_reconstruction", "actual_minimizer_number", "nb_mmer_selected",
                 "total_kmer_number", "total_superkmer_number")]


class Timing(C.Structure):
    _fields_ = [("dense_ms", C.c_double), ("dense_launches", C.c_uint64), ("scan_ms", C.c_double),
                ("scan_calls", C.c_uint64), ("accumulate_ms", C.c_double), ("accumulate_launches", C.c_uint64),
                ("compare_ms", C.c_double), ("compare_calls", C.c_uint64),
                ("scatter_ms", C.c_double), ("scatter_launches", C.c_uint64),
                ("group_ms", C.c_double), ("group_launches", C.c_uint64)]


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("read_s", "ingest_s", "scan_s", "gather_s", "build_s", "gzip_s", "load_s",
                                          "compare_s", "csv_s", "csv_gzip_s")] + [("sketch_files", C.c_uint64), ("compare_calls", C.c_uint64)]


class HbmRates(C.Structure):
    _fields_ = [("copy_GBps", C.c_double), ("copy_ms", C.c_double), ("read_GBps", C.c_double), ("read_ms", C.c_double),
                ("bytes", C.c_uint64), ("reps", C.c_uint32), ("n_cu", C.c_uint32)]


FILE_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(SketchStats), C.c_char_p)

SUPERKMER_DTYPE = np.dtype([("rec", "<u4"), ("minimizer", "<u4"), ("start", "<u8"), ("len", "<u4"), ("rev", "<u4")])

# every symbol include/spsp.h declares (tests check the .so exports them all)
ABI_SYMBOLS = [
    "spsp_device_count", "spsp_create", "spsp_destroy", "spsp_last_error", "spsp_version", "spsp_free", "spsp_copy_to_host",
    "spsp_stream_create_cus", "spsp_stream_destroy", "spsp_set_cu_count", "spsp_pack_bases_device",
    "spsp_timing_enable", "spsp_timing_sample", "spsp_timing_read", "spsp_threshold_host", "spsp_scan", "spsp_scan_device", "spsp_scan_device_begin", "spsp_scan_device_end", "spsp_scan_tail_stream", "spsp_wait_dense", "spsp_wait_stream", "spsp_scan_hits_device", "spsp_count_superkmers_device", "spsp_compare",
    "spsp_compare_device", "spsp_slot_bytes", "spsp_partition_keys_device", "spsp_compare_slots_device", "spsp_compare_device_begin", "spsp_compare_slots_device_begin", "spsp_compare_end", "spsp_fasta_clean_host", "spsp_fasta_clean_device", "spsp_fasta_clean_packed_device", "spsp_sketch_text", "spsp_sketch_build_host", "spsp_sketch_parse_host", "spsp_sketch_decode_device", "spsp_sketch_keys_device", "spsp_sketch_keys_device_begin", "spsp_sketch_keys_device_end", "spsp_sketch_keys_big_genomes", "spsp_scan_output_wait", "spsp_compare_keys_unordered", "spsp_compare_forget", "spsp_sketch_chain_host",
    "spsp_csv_host", "spsp_csv_cells_host", "spsp_csv_cells_gz_host", "spsp_sort_csv_host", "spsp_read_file_host", "spsp_write_gz_host", "spsp_sketch_file", "spsp_compare_files", "spsp_compare_files_chatty", "spsp_stage_times_read", "spsp_measure_hbm_device", "spsp_sketch_files", "spsp_sketch_files_multi", "spsp_sketch_files_release", "spsp_compare_files_multi", "spsp_matrix_cells_device", "spsp_matrix_add_cells_device", "spsp_compare_cells_device", "spsp_compare_slots_cells_device",
]

_lib = None


def lib():
    """Load libspsp.so (once).  Raises SpspError if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpspError("%s not found: build the HIP extension first (make -C supersampler_amd/csrc)" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if LIB_OVERRIDDEN:
        import sys
        print("[supersampler_amd] SPSP_LIB is set: loaded %s instead of the in-tree libspsp.so" % LIB_PATH, file=sys.stderr, flush=True)
    u64, u32, vp, cp, dbl, i32 = C.c_uint64, C.c_uint32, C.c_void_p, C.c_char_p, C.c_double, C.c_int
    P = C.POINTER
    L.spsp_device_count.restype = i32; L.spsp_device_count.argtypes = []
    L.spsp_create.restype = i32; L.spsp_create.argtypes = [i32, vp, P(vp)]
    L.spsp_destroy.restype = None; L.spsp_destroy.argtypes = [vp]
    L.spsp_last_error.restype = cp; L.spsp_last_error.argtypes = []
    L.spsp_version.restype = cp; L.spsp_version.argtypes = []
    L.spsp_free.restype = None; L.spsp_free.argtypes = [vp]
    L.spsp_copy_to_host.restype = i32; L.spsp_copy_to_host.argtypes = [vp, vp, vp, u64]
    L.spsp_stream_create_cus.restype = i32; L.spsp_stream_create_cus.argtypes = [i32, u32, u32, P(vp)]
    L.spsp_stream_destroy.restype = i32; L.spsp_stream_destroy.argtypes = [i32, vp]
    L.spsp_set_cu_count.restype = i32; L.spsp_set_cu_count.argtypes = [vp, u32, u32]
    L.spsp_pack_bases_device.restype = i32; L.spsp_pack_bases_device.argtypes = [vp, vp, u64, P(vp)]
    L.spsp_timing_enable.restype = i32; L.spsp_timing_enable.argtypes = [vp, i32]
    L.spsp_timing_read.restype = i32; L.spsp_timing_read.argtypes = [vp, P(Timing)]
    L.spsp_timing_sample.restype = i32; L.spsp_timing_sample.argtypes = [vp, u32]
    L.spsp_threshold_host.restype = u64; L.spsp_threshold_host.argtypes = [u32, u32, dbl]
    L.spsp_scan.restype = i32; L.spsp_scan.argtypes = [vp, P(Params), vp, vp, u32, P(vp), P(u64)]
    L.spsp_scan_device.restype = i32
    L.spsp_scan_device.argtypes = [vp, P(Params), vp, u64, vp, u32, P(vp), P(u64)]
    L.spsp_scan_device_begin.restype = i32
    L.spsp_scan_device_begin.argtypes = [vp, P(Params), vp, u64, vp, u32]
    L.spsp_scan_device_end.restype = i32; L.spsp_scan_device_end.argtypes = [vp, P(vp), P(u64)]
    L.spsp_wait_dense.restype = i32; L.spsp_wait_dense.argtypes = [vp, vp]
    L.spsp_wait_stream.restype = i32; L.spsp_wait_stream.argtypes = [vp, vp]
    L.spsp_compare_device_begin.restype = i32
    L.spsp_compare_device_begin.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32, u32, u32, vp]
    L.spsp_compare_slots_device_begin.restype = i32
    L.spsp_compare_slots_device_begin.argtypes = [vp, u32, vp, u32, u32, u32, vp]
    L.spsp_compare_end.restype = i32; L.spsp_compare_end.argtypes = [vp]
    L.spsp_scan_hits_device.restype = i32
    L.spsp_scan_hits_device.argtypes = [vp, P(Params), vp, u64, P(u64)]
    L.spsp_compare.restype = i32; L.spsp_compare.argtypes = [vp, P(SketchView), u32, u32, vp, vp]
    L.spsp_compare_device.restype = i32
    L.spsp_compare_device.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32, u32, u32, vp]
    L.spsp_slot_bytes.restype = u64; L.spsp_slot_bytes.argtypes = [u32, u32, u32]
    L.spsp_partition_keys_device.restype = i32
    L.spsp_partition_keys_device.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32, u32, vp]
    L.spsp_compare_slots_device.restype = i32
    L.spsp_compare_slots_device.argtypes = [vp, u32, vp, u32, u32, u32, vp]
    L.spsp_fasta_clean_host.restype = i32
    L.spsp_fasta_clean_host.argtypes = [cp, u64, P(vp), P(vp), P(u32)]
    L.spsp_fasta_clean_device.restype = i32
    L.spsp_fasta_clean_device.argtypes = [vp, vp, u64, P(vp), P(u64), P(vp), P(u32)]
    L.spsp_fasta_clean_packed_device.restype = i32
    L.spsp_fasta_clean_packed_device.argtypes = [vp, vp, u64, P(vp), P(u64), P(vp), P(u32)]
    L.spsp_sketch_text.restype = i32
    L.spsp_sketch_text.argtypes = [vp, P(Params), dbl, cp, u64, P(vp), P(u64), P(SketchStats)]
    L.spsp_sketch_build_host.restype = i32
    L.spsp_sketch_build_host.argtypes = [P(Params), dbl, vp, vp, u32, vp, u64, P(vp), P(u64), P(SketchStats)]
    L.spsp_sketch_parse_host.restype = i32
    L.spsp_sketch_parse_host.argtypes = [cp, u64, P(u32), P(u32), P(vp), P(vp), P(vp), P(u64)]
    L.spsp_sketch_chain_host.restype = i32
    L.spsp_sketch_chain_host.argtypes = [cp, u64, u32, u32, C.c_char_p, P(i32), P(u32), P(u64), P(u64)]
    L.spsp_csv_host.restype = i32
    L.spsp_csv_host.argtypes = [i32, P(cp), u32, u32, vp, vp, i32, dbl, P(vp), P(u64)]
    L.spsp_csv_cells_host.restype = i32
    L.spsp_csv_cells_host.argtypes = [i32, P(cp), u32, u32, vp, u64, vp, i32, dbl, P(vp), P(u64)]
    L.spsp_csv_cells_gz_host.restype = i32
    L.spsp_csv_cells_gz_host.argtypes = [i32, P(cp), u32, u32, vp, u64, vp, i32, dbl, cp]
    L.spsp_sort_csv_host.restype = i32
    L.spsp_sort_csv_host.argtypes = [cp, u64, cp, u64, P(vp), P(u64)]
    L.spsp_read_file_host.restype = i32; L.spsp_read_file_host.argtypes = [cp, P(vp), P(u64)]
    L.spsp_write_gz_host.restype = i32; L.spsp_write_gz_host.argtypes = [cp, cp, u64, i32]
    L.spsp_sketch_file.restype = i32
    L.spsp_sketch_file.argtypes = [vp, P(Params), dbl, cp, cp, P(SketchStats)]
    L.spsp_compare_files.restype = i32
    L.spsp_compare_files.argtypes = [vp, P(cp), u32, u32, i32, dbl, cp]
    L.spsp_sketch_keys_device.restype = i32
    L.spsp_sketch_keys_device.argtypes = [vp, P(Params), vp, u64, vp, vp, u64, vp, u32, u32, P(vp), P(vp), P(vp), vp]
    L.spsp_sketch_keys_device_begin.restype = i32
    L.spsp_sketch_keys_device_begin.argtypes = [vp, P(Params), vp, u64, vp, vp, u64, vp, u32, u32]
    L.spsp_compare_keys_unordered.restype = i32; L.spsp_compare_keys_unordered.argtypes = [vp, i32]
    L.spsp_compare_forget.restype = i32; L.spsp_compare_forget.argtypes = [vp]
    L.spsp_sketch_keys_device_end.restype = i32
    L.spsp_sketch_keys_device_end.argtypes = [vp, P(vp), P(vp), P(vp), vp]
    L.spsp_sketch_keys_big_genomes.restype = u32; L.spsp_sketch_keys_big_genomes.argtypes = [vp]
    L.spsp_scan_output_wait.restype = i32; L.spsp_scan_output_wait.argtypes = [vp, vp]
    L.spsp_sketch_decode_device.restype = i32
    L.spsp_sketch_decode_device.argtypes = [vp, P(cp), P(u64), u32, P(u32), P(u32), P(vp), P(vp), P(vp), P(u64)]
    L.spsp_count_superkmers_device.restype = i32
    L.spsp_count_superkmers_device.argtypes = [vp, P(Params), vp, u64, vp, u32, P(u64)]
    L.spsp_scan_tail_stream.restype = i32; L.spsp_scan_tail_stream.argtypes = [vp, i32, vp]
    L.spsp_stage_times_read.restype = i32; L.spsp_stage_times_read.argtypes = [vp, P(StageTimes), i32]
    L.spsp_sketch_files.restype = i32
    L.spsp_sketch_files.argtypes = [i32, P(Params), dbl, P(cp), P(cp), u32, u32, FILE_CALLBACK, vp, P(StageTimes)]
    L.spsp_sketch_files_multi.restype = i32
    L.spsp_sketch_files_multi.argtypes = [P(i32), u32, P(Params), dbl, P(cp), P(cp), u32, u32, FILE_CALLBACK, vp, P(StageTimes)]
    L.spsp_sketch_files_release.restype = None; L.spsp_sketch_files_release.argtypes = [i32]
    L.spsp_measure_hbm_device.restype = i32; L.spsp_measure_hbm_device.argtypes = [vp, u64, u32, P(HbmRates)]
    L.spsp_compare_files_multi.restype = i32
    L.spsp_compare_files_multi.argtypes = [P(i32), u32, P(cp), u32, u32, i32, dbl, cp, i32, P(StageTimes)]
    L.spsp_matrix_cells_device.restype = i32; L.spsp_matrix_cells_device.argtypes = [vp, vp, u32, u32, u32, vp, u64, P(u64)]
    L.spsp_compare_cells_device.restype = i32; L.spsp_compare_cells_device.argtypes = [vp, u32, vp, vp, vp, vp, u32, u32, vp, vp, u64, P(u64)]
    L.spsp_compare_slots_cells_device.restype = i32; L.spsp_compare_slots_cells_device.argtypes = [vp, u32, vp, u32, u32, u32, vp, vp, u64, P(u64)]
    L.spsp_matrix_add_cells_device.restype = i32; L.spsp_matrix_add_cells_device.argtypes = [vp, vp, u32, vp, u64]
    _lib = L
    return L


def library_info():
    """which shared library this process computes with (path, version string, whether SPSP_LIB replaced the in-tree build)"""
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)) if not LIB_OVERRIDDEN else LIB_PATH,
            "version": lib().spsp_version().decode(), "overridden_by_SPSP_LIB": LIB_OVERRIDDEN}


def _check(rc):
    if rc != 0:
        e = SpspError("libspsp error %d: %s" % (rc, lib().spsp_last_error().decode(errors="replace")))
        e.code = rc
        raise e


def _take(ptr, nbytes):
    data = C.string_at(ptr, nbytes) if nbytes else b""
    lib().spsp_free(ptr)
    return data


def slot_bytes(n, slot_cap, k):
    return int(lib().spsp_slot_bytes(n, slot_cap, k))


def threshold(k, m, s):
    """selection threshold for -s (Subsampler::compute_threshold)."""
    return lib().spsp_threshold_host(k, m, float(s))


def make_params(k=31, m=11, s=1000.0, abundance=1, flags=SPSP_SCAN_DEFAULT, threshold_value=None):
    p = Params()
    p.k, p.m, p.abundance, p.flags = k, m, abundance, flags
    p.threshold = threshold(k, m, s) if threshold_value is None else threshold_value
    return p


def clean_fasta(text):
    """FASTA bytes (already gunzipped) -> (bases uint8[n], rec_off uint64[n_rec+1])."""
    bases, offs, n_rec = C.c_void_p(), C.c_void_p(), C.c_uint32()
    _check(lib().spsp_fasta_clean_host(text, len(text), C.byref(bases), C.byref(offs), C.byref(n_rec)))
    off = np.frombuffer(_take(offs, 8 * (n_rec.value + 1)), dtype=np.uint64).copy()
    b = np.frombuffer(_take(bases, int(off[-1])), dtype=np.uint8).copy()
    return b, off


def sketch_build(params, rate, bases, rec_off, superkmers):
    """super-k-mer stream -> (uncompressed sketch payload, stats dict)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    sk = np.ascontiguousarray(superkmers, dtype=SUPERKMER_DTYPE)
    out, n, st = C.c_void_p(), C.c_uint64(), SketchStats()
    _check(lib().spsp_sketch_build_host(C.byref(params), float(rate), bases.ctypes.data, rec_off.ctypes.data,
                                        len(rec_off) - 1, sk.ctypes.data, len(sk), C.byref(out), C.byref(n),
                                        C.byref(st)))
    return _take(out, n.value), {f: getattr(st, f) for f, _ in SketchStats._fields_}


class Sketch:
    """Parsed sketch: sorted distinct (minimizer, canonical k-mer) keys."""

    def __init__(self, k, m, minimizer, kmer_lo, kmer_hi):
        self.k, self.m = k, m
        self.minimizer, self.kmer_lo, self.kmer_hi = minimizer, kmer_lo, kmer_hi

    def __len__(self):
        return len(self.minimizer)

    def key_set(self):
        return {(int(a), (int(h) << 64) | int(l)) for a, l, h in zip(self.minimizer, self.kmer_lo, self.kmer_hi)}


def sketch_parse(payload):
    k, m, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
    mn, lo, hi = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _check(lib().spsp_sketch_parse_host(payload, len(payload), C.byref(k), C.byref(m), C.byref(mn), C.byref(lo),
                                        C.byref(hi), C.byref(n)))
    c = n.value
    a = np.frombuffer(_take(mn, 4 * c), dtype=np.uint32).copy()
    b = np.frombuffer(_take(lo, 8 * c), dtype=np.uint64).copy()
    d = np.frombuffer(_take(hi, 8 * c), dtype=np.uint64).copy()
    return Sketch(k.value, m.value, a, b, d)


def csv(jaccard, names, inter, card, n_query=None, precision=6, min_threshold=0.0):
    n = len(names)
    nq = n if n_query is None else n_query
    inter = np.ascontiguousarray(inter, dtype=np.uint32)
    card = np.ascontiguousarray(card, dtype=np.uint64)
    arr = (C.c_char_p * n)(*[s.encode() for s in names])
    out, ln = C.c_void_p(), C.c_uint64()
    _check(lib().spsp_csv_host(1 if jaccard else 0, arr, n, nq, inter.ctypes.data, card.ctypes.data, precision,
                               float(min_threshold), C.byref(out), C.byref(ln)))
    return _take(out, ln.value)


def csv_cells(jaccard, names, cells, card, n_query=None, precision=6, min_threshold=0.0):
    """the printers from the sparse form of the pair matrix: cells = uint64 array of i << 48 | j << 32 | count (i < j)"""
    n = len(names)
    nq = n if n_query is None else n_query
    cells = np.ascontiguousarray(cells, dtype=np.uint64)
    card = np.ascontiguousarray(card, dtype=np.uint64)
    arr = (C.c_char_p * n)(*[s.encode() for s in names])
    out, ln = C.c_void_p(), C.c_uint64()
    _check(lib().spsp_csv_cells_host(1 if jaccard else 0, arr, n, nq, cells.ctypes.data, len(cells), card.ctypes.data, precision,
                                     float(min_threshold), C.byref(out), C.byref(ln)))
    return _take(out, ln.value)


def csv_cells_gz(jaccard, names, cells, card, gz_path, n_query=None, precision=6, min_threshold=0.0):
    """spsp_csv_cells_gz_host: the same matrix straight into a .csv.gz (gzip members made without the text in between)"""
    n = len(names)
    nq = n if n_query is None else n_query
    cells = np.ascontiguousarray(cells, dtype=np.uint64)
    card = np.ascontiguousarray(card, dtype=np.uint64)
    arr = (C.c_char_p * n)(*[s.encode() for s in names])
    _check(lib().spsp_csv_cells_gz_host(1 if jaccard else 0, arr, n, nq, cells.ctypes.data, len(cells), card.ctypes.data, precision,
                                        float(min_threshold), gz_path.encode()))


def sketches_from_payloads(payloads):
    """Parse sketch payloads IN FILE ORDER, as the comparator reads them: besides sketch_parse this applies the
    merge's first-read rule (spsp_sketch_chain_host), which gives an empty sketch one phantom key when k == m."""
    out = []
    buf = None
    for pl in payloads:
        sk = sketch_parse(pl)
        if buf is None:
            buf = C.create_string_buffer(b"A" * 16, 16)
        has, mn, lo, hi = C.c_int32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(lib().spsp_sketch_chain_host(pl, len(pl), out[0].k if out else sk.k, out[0].m if out else sk.m, buf,
                                            C.byref(has), C.byref(mn), C.byref(lo), C.byref(hi)))
        if has.value and len(sk) == 0:
            sk = Sketch(sk.k, sk.m, np.array([mn.value], np.uint32), np.array([lo.value], np.uint64),
                        np.array([hi.value], np.uint64))
        out.append(sk)
    return out


def sort_csv(csv_text, fof_text):
    """sortCSV on gunzipped CSV bytes + file-of-files bytes -> reordered CSV bytes"""
    out, n = C.c_void_p(), C.c_uint64()
    _check(lib().spsp_sort_csv_host(csv_text, len(csv_text), fof_text, len(fof_text), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def read_file(path):
    out, ln = C.c_void_p(), C.c_uint64()
    _check(lib().spsp_read_file_host(path.encode(), C.byref(out), C.byref(ln)))
    return _take(out, ln.value)


def write_gz(path, data, level=9):
    _check(lib().spsp_write_gz_host(path.encode(), data, len(data), level))


def sketch_files(fasta_paths, out_paths, k=31, m=11, s=1000.0, abundance=1, threads=8, device=0, flags=SPSP_SCAN_DEFAULT, devices=None):
    """many FASTA files -> many sketch files on `threads` workers, one context (HIP stream) each (spsp_sketch_files).
    Returns (per-file list of (rc, stats dict or None, error text or None), stage seconds summed over the workers)."""
    n = len(fasta_paths)
    p = make_params(k, m, s, abundance, flags)
    ins = (C.c_char_p * n)(*[x.encode() for x in fasta_paths])
    outs = (C.c_char_p * n)(*[x.encode() for x in out_paths])
    res = [None] * n
    started = []

    def on_file(user, i, phase, rc, st, err):
        if phase == 0:
            started.append(i)
        else:
            res[i] = (rc, {f: getattr(st.contents, f) for f, _ in SketchStats._fields_} if rc == 0 else None, err.decode() if err else None)
    cb = FILE_CALLBACK(on_file)
    times = StageTimes()
    if devices is not None:         # spsp_sketch_files_multi: the batches dealt over these devices
        devs = (C.c_int * len(devices))(*devices)
        rc = lib().spsp_sketch_files_multi(devs, len(devices), C.byref(p), float(s), ins, outs, n, threads, cb, None, C.byref(times))
    else:
        rc = lib().spsp_sketch_files(device, C.byref(p), float(s), ins, outs, n, threads, cb, None, C.byref(times))
    if rc != 0 and not any(r is not None and r[0] != 0 for r in res):
        _check(rc)
    return res, {f: getattr(times, f) for f, _ in StageTimes._fields_}, started


def sketch_files_release(device=-1):
    """release the contexts and pinned buffers spsp_sketch_files keeps between calls"""
    lib().spsp_sketch_files_release(device)


def device_count():
    """spsp_device_count: gfx950 devices visible to this process (initialises HIP)"""
    return int(lib().spsp_device_count())


def _paths_array(paths):
    """a `const char* const*` over `paths` for the C side, and what must stay alive while it is used.  Thousands of paths: ONE
    encode of the joined names and pointer arithmetic in numpy (10 000 c_char_p objects made one by one were 5 ms of a 37 ms
    spsp_compare_files call); anything that is not plain ASCII takes the plain road."""
    n = len(paths)
    if n >= 256:
        joined = "\0".join(paths)
        blob = joined.encode() + b"\0"
        if len(blob) == len(joined) + 1:                                          # ASCII: a character is a byte
            lens = np.fromiter(map(len, paths), dtype=np.uint64, count=n)
            if int(lens.sum()) + n == len(blob):                                  # (no name holds a NUL)
                buf = C.create_string_buffer(blob, len(blob))
                ptrs = np.empty(n, dtype=np.uint64)
                ptrs[0] = 0
                np.cumsum(lens[:-1] + np.uint64(1), out=ptrs[1:])
                ptrs += np.uint64(C.addressof(buf))
                return ptrs.ctypes.data_as(C.POINTER(C.c_char_p)), (buf, ptrs)
    arr = (C.c_char_p * n)(*[x.encode() for x in paths])
    return arr, (arr,)


def compare_files_multi(devices, paths, out_prefix, n_query=None, precision=6, min_threshold=0.0):
    """spsp_compare_files_multi: the comparator split by key over one context per entry of `devices` -> stage seconds"""
    n = len(paths)
    devs = (C.c_int * len(devices))(*devices)
    arr, _alive = _paths_array(paths)
    st = StageTimes()
    _check(lib().spsp_compare_files_multi(devs, len(devices), arr, n, n if n_query is None else n_query, precision, float(min_threshold),
                                          out_prefix.encode(), 0, C.byref(st)))
    return {f: getattr(st, f) for f, _ in StageTimes._fields_}


def stream_create_cus(device, first_cu, n_cu):
    """a HIP stream (handle as int) whose kernels run on logical CUs [first_cu, first_cu + n_cu) only"""
    h = C.c_void_p()
    _check(lib().spsp_stream_create_cus(device, first_cu, n_cu, C.byref(h)))
    return h.value


def stream_destroy(device, stream):
    _check(lib().spsp_stream_destroy(device, C.c_void_p(stream)))


class Context:
    """One HIP device + one stream (spsp_create).  Raises SpspError without a gfx950 GPU."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().spsp_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().spsp_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_host(self, d_ptr, count, dtype):
        """copy `count` items of numpy dtype from a device buffer returned by this context."""
        out = np.zeros(count, dtype=dtype)
        _check(lib().spsp_copy_to_host(self._h, out.ctypes.data, d_ptr, out.nbytes))
        return out

    # ---- measurement
    def timing_enable(self, on=True, kinds=TIME_ALL):
        """HIP-event brackets for the regions in `kinds` (TIME_DENSE | TIME_SCAN | TIME_ACCUMULATE | TIME_COMPARE)"""
        _check(lib().spsp_timing_enable(self._h, int(kinds) if on else 0))

    def timing_sample(self, every):
        """bracket only every `every`-th region (the event packets themselves delay a pipelined stream)"""
        _check(lib().spsp_timing_sample(self._h, every))

    def timing_read(self):
        """HIP-event totals since the last read (synchronises the stream) as a dict."""
        t = Timing()
        _check(lib().spsp_timing_read(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in Timing._fields_}

    def measure_hbm(self, nbytes=1 << 30, reps=10):
        """streaming copy / read rates of this device on this context's stream (roofline denominator) as a dict"""
        r = HbmRates()
        _check(lib().spsp_measure_hbm_device(self._h, nbytes, reps, C.byref(r)))
        return {f: getattr(r, f) for f, _ in HbmRates._fields_}

    # ---- path A
    def scan(self, params, bases, rec_off):
        """cleaned ASCII records -> structured array of selected super-k-mers (genome order)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        out, n = C.c_void_p(), C.c_uint64()
        _check(lib().spsp_scan(self._h, C.byref(params), bases.ctypes.data, rec_off.ctypes.data, len(rec_off) - 1,
                               C.byref(out), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=SUPERKMER_DTYPE)
        return np.frombuffer(_take(out, n.value * SUPERKMER_DTYPE.itemsize), dtype=SUPERKMER_DTYPE).copy()

    def scan_device(self, params, d_bases, n_bases, d_rec_off, n_rec):
        """device pointers in, device pointer out: returns (d_out, n_out); d_out is owned by the
        context and valid until its next scan call."""
        out, n = C.c_void_p(), C.c_uint64()
        _check(lib().spsp_scan_device(self._h, C.byref(params), d_bases, n_bases, d_rec_off, n_rec, C.byref(out),
                                      C.byref(n)))
        return out.value, n.value

    def pack_bases_device(self, d_bases, n_bases):
        """cleaned ASCII bases on the device -> 2-bit packed buffer owned by the context (for SPSP_SCAN_PACKED_INPUT)"""
        out = C.c_void_p()
        _check(lib().spsp_pack_bases_device(self._h, d_bases, n_bases, C.byref(out)))
        return out.value

    def scan_device_begin(self, params, d_bases, n_bases, d_rec_off, n_rec):
        """queue the scan on the context's stream and return; collect with scan_device_end()"""
        _check(lib().spsp_scan_device_begin(self._h, C.byref(params), d_bases, n_bases, d_rec_off, n_rec))

    def scan_device_end(self):
        out, n = C.c_void_p(), C.c_uint64()
        _check(lib().spsp_scan_device_end(self._h, C.byref(out), C.byref(n)))
        return out.value, n.value

    def set_cu_count(self, n_cu, dense_blocks_per_cu=1):
        """how many CUs the context's stream owns (0 = the whole device) and how many dense workgroups go on each
        (2 for a stream that has its CUs to itself): sizes the dense pass's grid"""
        _check(lib().spsp_set_cu_count(self._h, n_cu, dense_blocks_per_cu))

    def scan_tail_stream(self, on=True, stream=None):
        """sparse stages of the scan on a second stream (None = one the context creates)"""
        _check(lib().spsp_scan_tail_stream(self._h, 1 if on else 0, stream))

    def wait_dense(self, scanner):
        """work queued on this context from now on starts behind `scanner`'s latest dense pass"""
        _check(lib().spsp_wait_dense(self._h, scanner._h))

    def wait_stream(self, other):
        """work queued on this context from now on starts behind everything queued on `other` so far"""
        _check(lib().spsp_wait_stream(self._h, other._h))

    def scan_hits_device(self, params, d_bases, n_bases):
        n = C.c_uint64()
        _check(lib().spsp_scan_hits_device(self._h, C.byref(params), d_bases, n_bases, C.byref(n)))
        return n.value

    def count_superkmers_device(self, params, d_bases, n_bases, d_rec_off, n_rec):
        """total_superkmer_number of print_stat: every super-k-mer of the input, selected or not"""
        n = C.c_uint64()
        _check(lib().spsp_count_superkmers_device(self._h, C.byref(params), d_bases, n_bases, d_rec_off, n_rec, C.byref(n)))
        return n.value

    def clean_fasta_device(self, d_text, n_text):
        """raw FASTA text in HBM -> (d_bases, n_bases, d_rec_off, n_rec), context-owned device buffers."""
        b, o, nb, nr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint32()
        _check(lib().spsp_fasta_clean_device(self._h, d_text, n_text, C.byref(b), C.byref(nb), C.byref(o), C.byref(nr)))
        return b.value, nb.value, o.value, nr.value

    def clean_fasta_packed_device(self, d_text, n_text):
        """raw FASTA text in HBM -> (d_packed 2-bit words, n_bases, d_rec_off, n_rec), context-owned device buffers"""
        b, o, nb, nr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint32()
        _check(lib().spsp_fasta_clean_packed_device(self._h, d_text, n_text, C.byref(b), C.byref(nb), C.byref(o), C.byref(nr)))
        return b.value, nb.value, o.value, nr.value

    def sketch_text(self, text, k=31, m=11, s=1000.0, abundance=1, flags=SPSP_SCAN_DEFAULT):
        """FASTA bytes -> (payload, stats) with ingest, scan and gather on the GPU."""
        p = make_params(k, m, s, abundance, flags)
        out, n, st = C.c_void_p(), C.c_uint64(), SketchStats()
        _check(lib().spsp_sketch_text(self._h, C.byref(p), float(s), text, len(text), C.byref(out), C.byref(n), C.byref(st)))
        return _take(out, n.value), {f: getattr(st, f) for f, _ in SketchStats._fields_}

    def sketch_fasta(self, text, k=31, m=11, s=1000.0, abundance=1, flags=SPSP_SCAN_DEFAULT):
        """FASTA bytes -> (payload, stats): clean -> GPU scan -> sketch builder."""
        p = make_params(k, m, s, abundance, flags)
        bases, off = clean_fasta(text)
        sk = self.scan(p, bases, off)
        return sketch_build(p, s, bases, off, sk)

    def sketch_file(self, fasta_path, out_path, k=31, m=11, s=1000.0, abundance=1):
        p = make_params(k, m, s, abundance)
        st = SketchStats()
        _check(lib().spsp_sketch_file(self._h, C.byref(p), float(s), fasta_path.encode(), out_path.encode(),
                                      C.byref(st)))
        return {f: getattr(st, f) for f, _ in SketchStats._fields_}

    def sketch_decode_device(self, payloads):
        """list of gunzipped sketch payloads -> (k, m, d_minimizer, d_kmer_lo, d_kmer_hi or None, sk_off np.uint64[n+1]);
        the device pointers belong to the context (valid until its next decode / compare call)"""
        n = len(payloads)
        arr = (C.c_char_p * n)(*payloads)
        lens = (C.c_uint64 * n)(*[len(p) for p in payloads])
        k, m = C.c_uint32(), C.c_uint32()
        d_mn, d_lo, d_hi = C.c_void_p(), C.c_void_p(), C.c_void_p()
        sk_off = np.zeros(n + 1, dtype=np.uint64)
        _check(lib().spsp_sketch_decode_device(self._h, arr, lens, n, C.byref(k), C.byref(m), C.byref(d_mn), C.byref(d_lo),
                                                C.byref(d_hi), sk_off.ctypes.data_as(C.POINTER(C.c_uint64))))
        return k.value, m.value, d_mn.value, d_lo.value, d_hi.value, sk_off

    def scan_output_wait(self, reader):
        """this context's next scan writes its output only behind `reader`'s latest sketch_keys_device_begin"""
        _check(lib().spsp_scan_output_wait(self._h, reader._h))

    def compare_forget(self):
        """forget what earlier comparisons taught this context about its inputs (scheduling only: spsp_compare_forget)"""
        _check(lib().spsp_compare_forget(self._h))

    def compare_keys_unordered(self, on=True):
        """device-form comparisons of this context accept sketches whose keys are distinct but unsorted"""
        _check(lib().spsp_compare_keys_unordered(self._h, 1 if on else 0))

    def sketch_keys_device_begin(self, params, d_bases, n_bases, d_rec_off, d_sk, n_sk, first_rec, unordered=False):
        """queue: scan output -> comparator keys of the genomes whose record ranges `first_rec` (n_genomes + 1) gives"""
        fr = np.ascontiguousarray(first_rec, dtype=np.uint32)
        _check(lib().spsp_sketch_keys_device_begin(self._h, C.byref(params), d_bases, n_bases, d_rec_off, d_sk, n_sk, fr.ctypes.data, len(fr) - 1,
                                                   KEYS_UNORDERED if unordered else 0))
        self._keys_n = len(fr) - 1

    def sketch_keys_device_end(self):
        """-> (d_minimizer, d_kmer_lo, d_kmer_hi or None, sk_off np.uint64[n_genomes + 1]); device pointers belong to the context"""
        d_mn, d_lo, d_hi = C.c_void_p(), C.c_void_p(), C.c_void_p()
        sk_off = np.zeros(self._keys_n + 1, dtype=np.uint64)
        _check(lib().spsp_sketch_keys_device_end(self._h, C.byref(d_mn), C.byref(d_lo), C.byref(d_hi), sk_off.ctypes.data))
        return d_mn.value, d_lo.value, d_hi.value, sk_off

    def sketch_keys_big_genomes(self):
        """genomes of the last collected key extraction that were beyond the per-genome LDS forms (table in HBM instead)"""
        return int(lib().spsp_sketch_keys_big_genomes(self._h))

    def sketch_keys_device(self, params, d_bases, n_bases, d_rec_off, d_sk, n_sk, first_rec, unordered=False):
        self.sketch_keys_device_begin(params, d_bases, n_bases, d_rec_off, d_sk, n_sk, first_rec, unordered)
        return self.sketch_keys_device_end()

    def stage_times(self, reset=True):
        """wall seconds the whole-file drivers (sketch_file / compare_files) spent per stage on this context"""
        st = StageTimes()
        _check(lib().spsp_stage_times_read(self._h, C.byref(st), 1 if reset else 0))
        return {f: getattr(st, f) for f, _ in StageTimes._fields_}

    # ---- path B
    def compare(self, sketches, n_query=None):
        """list of Sketch -> (inter uint32[n,n] upper triangle, card uint64[n])."""
        n = len(sketches)
        nq = n if n_query is None else n_query
        views = (SketchView * max(n, 1))()
        use_hi = any(s.k > 32 for s in sketches)
        for i, s in enumerate(sketches):
            views[i].minimizer = s.minimizer.ctypes.data
            views[i].kmer_lo = s.kmer_lo.ctypes.data
            views[i].kmer_hi = s.kmer_hi.ctypes.data if use_hi else None
            views[i].n = len(s)
        inter = np.zeros((n, n), dtype=np.uint32)
        card = np.zeros(n, dtype=np.uint64)
        _check(lib().spsp_compare(self._h, views, n, nq, inter.ctypes.data, card.ctypes.data))
        return inter, card

    def compare_device(self, k, d_min, d_lo, d_hi, sk_off, n, row_first, row_stride, d_inter, n_query=None):
        sk_off = np.ascontiguousarray(sk_off, dtype=np.uint64)
        _check(lib().spsp_compare_device(self._h, k, d_min, d_lo, d_hi, sk_off.ctypes.data, n,
                                         n if n_query is None else n_query, row_first, row_stride, d_inter))

    def compare_device_begin(self, k, d_min, d_lo, d_hi, sk_off, n, row_first, row_stride, d_inter, n_query=None):
        sk_off = np.ascontiguousarray(sk_off, dtype=np.uint64)
        _check(lib().spsp_compare_device_begin(self._h, k, d_min, d_lo, d_hi, sk_off.ctypes.data, n,
                                               n if n_query is None else n_query, row_first, row_stride, d_inter))

    def compare_slots_device_begin(self, k, d_slots, parts, n, slot_cap, d_inter):
        _check(lib().spsp_compare_slots_device_begin(self._h, k, d_slots, parts, n, slot_cap, d_inter))

    def compare_end(self):
        _check(lib().spsp_compare_end(self._h))

    def partition_keys_device(self, k, d_min, d_lo, d_hi, sk_off, n, parts, slot_cap, d_slots):
        """Scatter this rank's sketch keys into `parts` exchange slots (include/spsp.h); asynchronous."""
        sk_off = np.ascontiguousarray(sk_off, dtype=np.uint64)
        _check(lib().spsp_partition_keys_device(self._h, k, d_min, d_lo, d_hi, sk_off.ctypes.data, n, parts, slot_cap,
                                                d_slots))

    def compare_slots_device(self, k, d_slots, parts, n, slot_cap, d_inter):
        """Partial pair matrix of the hash class this rank received (one slot per source rank)."""
        _check(lib().spsp_compare_slots_device(self._h, k, d_slots, parts, n, slot_cap, d_inter))

    def matrix_cells_device(self, d_inter, n, d_cells, cap, row_first=0, row_limit=None):
        """non-zero cells (i < j) of a dense pair matrix on the device as packed words i << 48 | j << 32 | count -> how many"""
        cnt = C.c_uint64()
        _check(lib().spsp_matrix_cells_device(self._h, d_inter, n, row_first, n if row_limit is None else row_limit, d_cells, cap, C.byref(cnt)))
        return cnt.value

    def compare_cells_device(self, k, d_min, d_lo, d_hi, sk_off, n, d_scratch, d_cells, cap, n_query=None):
        """all-vs-all with the pair matrix returned as packed non-zero cells (i << 48 | j << 32 | count) -> how many"""
        sk_off = np.ascontiguousarray(sk_off, dtype=np.uint64)
        cnt = C.c_uint64()
        try:
            _check(lib().spsp_compare_cells_device(self._h, k, d_min, d_lo, d_hi, sk_off.ctypes.data, n, n if n_query is None else n_query, d_scratch,
                                                   d_cells, cap, C.byref(cnt)))
        except SpspError as e:
            e.cells_needed = cnt.value
            raise
        return cnt.value

    def compare_slots_cells_device(self, k, d_slots, parts, n, slot_cap, d_scratch, d_cells, cap):
        """this rank's partial matrix of the key-partitioned split as packed non-zero cells -> how many"""
        cnt = C.c_uint64()
        try:
            _check(lib().spsp_compare_slots_cells_device(self._h, k, d_slots, parts, n, slot_cap, d_scratch, d_cells, cap, C.byref(cnt)))
        except SpspError as e:
            e.cells_needed = cnt.value          # ERR_OVERFLOW from too little room: how much there must be (spsp.h)
            raise
        return cnt.value

    def matrix_add_cells_device(self, d_inter, n, d_cells, n_cells):
        _check(lib().spsp_matrix_add_cells_device(self._h, d_inter, n, d_cells, n_cells))

    def compare_files(self, paths, out_prefix, n_query=None, precision=6, min_threshold=0.0):
        n = len(paths)
        nq = n if n_query is None else n_query
        arr, _alive = _paths_array(paths)
        _check(lib().spsp_compare_files(self._h, arr, n, nq, precision, float(min_threshold), out_prefix.encode()))
