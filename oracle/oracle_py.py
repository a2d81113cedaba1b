"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/spsp_oracle.cpp header).  The product package
supersampler_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


class Superkmer(C.Structure):
    _fields_ = [("rec", C.c_uint32), ("minimizer", C.c_uint32), ("start", C.c_uint64),
                ("len", C.c_uint32), ("rev", C.c_uint32)]


SUPERKMER_DTYPE = np.dtype([("rec", "<u4"), ("minimizer", "<u4"), ("start", "<u8"),
                            ("len", "<u4"), ("rev", "<u4")])


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("read_kmer", "total_kmer_number", "total_superkmer_number", "nb_mmer_selected")]


class SketchStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("selected_kmer_number", "selected_superkmer_number", "count_maximal_skmer",
                 "seen_kmers_at_reconstruction", "seen_superkmers_at_reconstruction",
                 "seen_max_superkmers_at_reconstruction", "actual_minimizer_number",
                 "read_kmer", "total_kmer_number", "total_superkmer_number", "nb_mmer_selected")]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "spsp_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None
BUILD_FLAGS = None     # set by use_native(): the flags of the library in use, for bench.py's report


def use_native():
    """bench.py's timed leg: (re)build the oracle with BASELINE.md's flags for THIS host (-O3 -march=native -fopenmp)
    and switch to it.  Returns the flags in use; falls back to the shipped portable build if the box has no compiler."""
    global _lib, _SO, BUILD_FLAGS
    flags = None
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        native = os.path.join(_HERE, "liboracle_native.so")
        if os.path.exists(native):
            _SO, _lib = native, None
            flags = "-O3 -march=native -fopenmp -std=c++17 (built on this host)"
    except Exception:  # noqa: BLE001
        pass
    if flags is None:
        for line in open(os.path.join(_HERE, "Makefile")):
            if line.startswith("CXXFLAGS"):
                flags = line.split("=", 1)[1].strip() + " (portable build shipped with the repository)"
    BUILD_FLAGS = flags
    return flags


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u64, u32, dbl, vp, cp = C.c_uint64, C.c_uint32, C.c_double, C.c_void_p, C.c_char_p
    L.orc_xxh64_u64.restype = u64; L.orc_xxh64_u64.argtypes = [u64, u64]
    L.orc_threshold.restype = u64; L.orc_threshold.argtypes = [u32, u32, dbl]
    L.orc_rc64.restype = u64; L.orc_rc64.argtypes = [u64, u32]
    L.orc_canon64.restype = u64; L.orc_canon64.argtypes = [u64, u32]
    L.orc_canon128.restype = None
    L.orc_canon128.argtypes = [u64, u64, u32, C.POINTER(u64), C.POINTER(u64)]
    L.orc_str2num64.restype = u64; L.orc_str2num64.argtypes = [cp, u32]
    L.orc_free.restype = None; L.orc_free.argtypes = [vp]
    L.orc_compress.restype = vp; L.orc_compress.argtypes = [cp, u64, C.POINTER(u64)]
    L.orc_decompress.restype = vp; L.orc_decompress.argtypes = [cp, u64, C.POINTER(u64)]
    L.orc_rescan.restype = u64
    L.orc_rescan.argtypes = [u32, u32, cp, C.POINTER(u64), C.POINTER(u32)]
    L.orc_clean_fasta.restype = u64
    L.orc_clean_fasta.argtypes = [cp, u64, C.POINTER(vp), C.POINTER(vp)]
    L.orc_scan.restype = u64
    L.orc_scan.argtypes = [u32, u32, u64, vp, vp, u32, C.POINTER(vp), C.POINTER(ScanStats)]
    L.orc_scan_timed.restype = dbl
    L.orc_scan_timed.argtypes = [u32, u32, u64, vp, vp, u32, C.POINTER(u64), C.POINTER(u64)]
    L.orc_sketch_fasta.restype = vp
    L.orc_sketch_fasta.argtypes = [cp, u64, u32, u32, dbl, u32, C.POINTER(u64), C.POINTER(SketchStats)]
    L.orc_compare.restype = C.c_int
    L.orc_compare.argtypes = [C.POINTER(cp), C.POINTER(u64), u32, u32, vp, vp,
                              C.POINTER(u32), C.POINTER(u32)]
    L.orc_compare_timed.restype = dbl
    L.orc_compare_timed.argtypes = [C.POINTER(cp), C.POINTER(u64), u32, u32, vp, vp]
    L.orc_sketch_keys.restype = u64
    L.orc_sketch_keys.argtypes = [cp, u64, C.POINTER(u32), C.POINTER(u32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.orc_sort_csv.restype = vp
    L.orc_sort_csv.argtypes = [cp, u64, cp, u64, C.POINTER(u64)]
    L.orc_csv.restype = vp
    L.orc_csv.argtypes = [C.c_int, cp, u32, u32, vp, vp, C.c_int, dbl, C.POINTER(u64)]
    _lib = L
    return L


def _take(ptr, n):
    data = C.string_at(ptr, n)
    lib().orc_free(ptr)
    return data


def xxh64(x, seed=1312):
    return lib().orc_xxh64_u64(x, seed)


def threshold(k, m, s):
    return lib().orc_threshold(k, m, float(s))


def rc64(x, n):
    return lib().orc_rc64(x, n)


def canon64(x, n):
    return lib().orc_canon64(x, n)


def canon128(x, n):
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().orc_canon128(x & (2**64 - 1), x >> 64, n, C.byref(lo), C.byref(hi))
    return (hi.value << 64) | lo.value


def str2num(s):
    b = s.encode() if isinstance(s, str) else s
    return lib().orc_str2num64(b, len(b))


def compress(s):
    b = s.encode() if isinstance(s, str) else s
    n = C.c_uint64()
    return _take(lib().orc_compress(b, len(b), C.byref(n)), n.value)


def decompress(b):
    n = C.c_uint64()
    return _take(lib().orc_decompress(b, len(b), C.byref(n)), n.value)


def rescan(k, m, kmer_ascii):
    b = kmer_ascii.encode() if isinstance(kmer_ascii, str) else kmer_ascii
    pos, rev = C.c_uint64(), C.c_uint32()
    mini = lib().orc_rescan(k, m, b, C.byref(pos), C.byref(rev))
    return mini, pos.value, rev.value


def clean_fasta(text):
    """FASTA bytes -> (bases: np.uint8[n], offsets: np.uint64[n_rec+1])."""
    bases, offs = C.c_void_p(), C.c_void_p()
    n_rec = lib().orc_clean_fasta(text, len(text), C.byref(bases), C.byref(offs))
    off = np.frombuffer(C.string_at(offs, 8 * (n_rec + 1)), dtype=np.uint64).copy()
    total = int(off[-1])
    b = np.frombuffer(C.string_at(bases, total), dtype=np.uint8).copy()
    lib().orc_free(bases); lib().orc_free(offs)
    return b, off


def scan(k, m, thr, bases, offsets):
    """Literal reference scan -> (structured array of super-k-mers, stats dict)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = C.c_void_p()
    st = ScanStats()
    n = lib().orc_scan(k, m, thr, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                       C.byref(out), C.byref(st))
    arr = np.frombuffer(C.string_at(out, n * SUPERKMER_DTYPE.itemsize), dtype=SUPERKMER_DTYPE).copy()
    lib().orc_free(out)
    return arr, {f: getattr(st, f) for f, _ in ScanStats._fields_}


def scan_timed(k, m, thr, bases, offsets):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    kmers, nem = C.c_uint64(), C.c_uint64()
    sec = lib().orc_scan_timed(k, m, thr, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                               C.byref(kmers), C.byref(nem))
    return sec, kmers.value, nem.value


def sketch_fasta(text, k, m, s, abundance=1):
    """FASTA bytes -> (uncompressed sketch payload bytes, stats dict)."""
    n = C.c_uint64()
    st = SketchStats()
    p = lib().orc_sketch_fasta(text, len(text), k, m, float(s), abundance, C.byref(n), C.byref(st))
    return _take(p, n.value), {f: getattr(st, f) for f, _ in SketchStats._fields_}


def compare(payloads, n_query=None, timed=False):
    """list of payload bytes -> (inter np.uint32[n,n], card np.uint64[n], k, m[, seconds])."""
    n = len(payloads)
    nq = n if n_query is None else n_query
    arr = (C.c_char_p * n)(*payloads)
    sizes = (C.c_uint64 * n)(*[len(p) for p in payloads])
    inter = np.zeros((n, n), dtype=np.uint32)
    card = np.zeros(n, dtype=np.uint64)
    if timed:
        sec = lib().orc_compare_timed(arr, sizes, n, nq, inter.ctypes.data, card.ctypes.data)
        return inter, card, sec
    k, m = C.c_uint32(), C.c_uint32()
    rc = lib().orc_compare(arr, sizes, n, nq, inter.ctypes.data, card.ctypes.data, C.byref(k), C.byref(m))
    if rc != 0:
        raise RuntimeError("orc_compare failed")
    return inter, card, k.value, m.value


def sketch_keys(payload):
    """gunzipped sketch payload -> (k, m, minimizer u32[n], kmer_lo u64[n], kmer_hi u64[n]): the sorted distinct keys the
    comparator's bucket walk (Comparator.cpp:186-260) yields for this file"""
    k, m = C.c_uint32(), C.c_uint32()
    mn, lo, hi = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = lib().orc_sketch_keys(payload, len(payload), C.byref(k), C.byref(m), C.byref(mn), C.byref(lo), C.byref(hi))
    a = np.frombuffer(C.string_at(mn, 4 * n), dtype=np.uint32).copy()
    b = np.frombuffer(C.string_at(lo, 8 * n), dtype=np.uint64).copy()
    d = np.frombuffer(C.string_at(hi, 8 * n), dtype=np.uint64).copy()
    for ptr in (mn, lo, hi):
        lib().orc_free(ptr)
    return k.value, m.value, a, b, d


def csv(jaccard, names, inter, card, n_query=None, precision=6, min_threshold=0.0):
    n = len(names)
    nq = n if n_query is None else n_query
    inter = np.ascontiguousarray(inter, dtype=np.uint32)
    card = np.ascontiguousarray(card, dtype=np.uint64)
    ln = C.c_uint64()
    p = lib().orc_csv(1 if jaccard else 0, "\n".join(names).encode(), n, nq, inter.ctypes.data,
                      card.ctypes.data, precision, float(min_threshold), C.byref(ln))
    return _take(p, ln.value)


def sort_csv(csv_text, fof_text):
    """sortCSV (sort_csv.cpp:26-111) on gunzipped CSV bytes + fof bytes -> output file bytes, or None where the
    reference's behaviour is undefined (see orc_sort_csv)."""
    ln = C.c_uint64()
    p = lib().orc_sort_csv(csv_text, len(csv_text), fof_text, len(fof_text), C.byref(ln))
    if not p:
        return None
    return _take(p, ln.value)
