"""CPU tests: the C-ABI library loads and exports what include/spsp.h declares,
the GPU entry points fail loudly without a device, and the host side of the
CLIs (ingest, sketch builder/reader, CSV, gz I/O) matches the oracle byte for
byte.  No GPU compute is called here."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import bruteforce as bf
import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=60).stdout
        return "gfx950" in out
    except Exception:
        return False


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "spsp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(spsp_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(sp.ABI_SYMBOLS)
    L = sp.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.spsp_version()


def test_library_carries_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + sp.LIB_PATH], capture_output=True, text=True)
    blob = open(sp.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and (b"k_dense" in blob), out.stderr


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present; the loud-failure path is for GPU-less hosts")
def test_gpu_entry_points_fail_loudly_without_device():
    with pytest.raises(sp.SpspError) as e:
        sp.Context(0)
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_threshold_matches_oracle():
    for (k, m, s) in [(31, 11, 1000), (63, 15, 100), (31, 11, 2), (21, 11, 50), (31, 15, 10000), (15, 15, 3),
                      (31, 11, 1.0), (31, 11, 0.3), (31, 11, float(np.float32(0.1) * 0 + 1.5))]:
        assert sp.threshold(k, m, s) == orc.threshold(k, m, s)


FASTA_CASES = [
    b">r1 desc\nACGTNNacgt\r\nGG\n>r2\n\nTTTT\n>empty\n>r4\nAC",
    b"ACGT\nGGGG\n",
    b"",
    b">only header",
    b">h\n",
    b"\n\n>x\nAC\n\n\nGT\n>y\nNNNN\n>z\nacgtn",
    b">a\nAC\xffGT\n\xff\nTT\n",
    b">a\nACGT",
]


@pytest.mark.parametrize("text", FASTA_CASES)
def test_fasta_clean_matches_oracle(text):
    b1, o1 = sp.clean_fasta(text)
    b2, o2 = orc.clean_fasta(text)
    assert o1.tolist() == o2.tolist()
    assert b1.tobytes() == b2.tobytes()


def _stream_and_payload(text, k, m, s, abundance=1):
    bases, offs = orc.clean_fasta(text)
    T = orc.threshold(k, m, s)
    em, _ = orc.scan(k, m, T, bases, offs)
    payload, st = orc.sketch_fasta(text, k, m, s, abundance)
    return bases, offs, em, payload, st


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 30, 1), (21, 11, 4, 1), (63, 15, 12, 1), (31, 11, 6, 2),
                                      (15, 15, 3, 1), (31, 11, 1, 1), (33, 13, 9, 3)])
def test_sketch_builder_matches_oracle_bytes(k, m, s, ab):
    rng = np.random.default_rng(k * 100 + m + ab)
    anc = bf.random_dna(rng, 3000)
    recs = [anc, bf.mutate(rng, anc, 0.01), anc[500:1500], bf.random_dna(rng, 40), "A" * 120, "ACGTTGCA" * 30]
    text = bf.fasta(recs)
    bases, offs, em, payload, st = _stream_and_payload(text, k, m, s, ab)
    p = sp.make_params(k, m, s, ab)
    got, gst = sp.sketch_build(p, s, bases, offs, em)
    assert got == payload
    for f in ("selected_kmer_number", "selected_superkmer_number", "count_maximal_skmer",
              "seen_kmers_at_reconstruction", "seen_superkmers_at_reconstruction",
              "seen_max_superkmers_at_reconstruction", "actual_minimizer_number", "read_kmer", "nb_mmer_selected"):
        assert gst[f] == st[f], f


def test_sketch_builder_count_wraps_at_256():
    """H4: a k-mer seen 256 times has count 0 and fails `count >= abundance`."""
    k, m, s = 21, 11, 1.0
    unit = bf.random_dna(np.random.default_rng(4), 60)
    text = bf.fasta([unit] * 256 + [unit[:40]])
    bases, offs, em, payload, st = _stream_and_payload(text, k, m, s)
    got, _ = sp.sketch_build(sp.make_params(k, m, s), s, bases, offs, em)
    assert got == payload
    full = bf.sketch_set([unit], k, m, 2**64 - 1)
    kept = bf.payload_set(payload)
    assert kept < full and len(kept) > 0  # the k-mers past position 40 were seen exactly 256 times and vanish


@pytest.mark.parametrize("k,m,s", [(31, 11, 25), (63, 15, 8), (21, 11, 3), (15, 15, 2)])
def test_sketch_reader_matches_bruteforce_and_oracle(k, m, s):
    rng = np.random.default_rng(k + m)
    anc = bf.random_dna(rng, 4000)
    gs = [[anc], [bf.mutate(rng, anc, 0.02), bf.random_dna(rng, 300)], [bf.random_dna(rng, 2500)]]
    payloads = [orc.sketch_fasta(bf.fasta(g), k, m, s)[0] for g in gs]
    inter, card, kk, mm = orc.compare(payloads)
    for i, pl in enumerate(payloads):
        sk = sp.sketch_parse(pl)
        assert (sk.k, sk.m) == (k, m)
        assert sk.key_set() == bf.payload_set(pl)
        assert len(sk) == card[i]
        keys = list(zip(sk.minimizer.tolist(), sk.kmer_hi.tolist(), sk.kmer_lo.tolist()))
        assert keys == sorted(set(keys))


@pytest.mark.parametrize("k,m,s", [(31, 11, 25), (63, 15, 8), (21, 11, 3), (15, 15, 2)])
def test_oracle_key_enumeration_matches_bruteforce_and_host_parser(k, m, s):
    """orc_sketch_keys (the oracle's merge + bucket walk over ONE file) is what the GPU decoder is held against in
    tests/test_gpu.py: here it is pinned itself -- equal to the brute-force reading of the payload (tests/bruteforce.py)
    and, array for array, to the product's host parser."""
    rng = np.random.default_rng(100 + k + m)
    anc = bf.random_dna(rng, 5000)
    for g in ([anc, anc[:2000]], [bf.mutate(rng, anc, 0.03)], [bf.random_dna(rng, 40)]):
        pl = orc.sketch_fasta(bf.fasta(g), k, m, s)[0]
        kk, mm, mn, lo, hi = orc.sketch_keys(pl)
        assert (kk, mm) == (k, m)
        assert {(int(a), (int(h) << 64) | int(l)) for a, l, h in zip(mn, lo, hi)} == bf.payload_set(pl)
        sk = sp.sketch_parse(pl)
        assert (sk.minimizer == mn).all() and (sk.kmer_lo == lo).all() and (sk.kmer_hi == hi).all()


def test_direct_sketch_synthesis_is_consistent():
    """synth.direct_family_sketches (the workload of bench.py's configs[3] leg): the key arrays it returns are what the
    oracle and the host parser read out of the payloads it writes for the same sketches, and pair counts by numpy set
    algebra equal the oracle's comparison of those payloads."""
    from supersampler_amd import synth
    D = synth.direct_family_sketches(40, fam_size=20, seed=11, skm_range=(30, 60))
    pls = [D.payload(i) for i in range(D.n)]
    mn_all, lo_all = D.minimizer.numpy().view(np.uint32), D.kmer_lo.numpy().view(np.uint64)
    sets = []
    for i in range(D.n):
        a, b = int(D.sk_off[i]), int(D.sk_off[i + 1])
        kk, mm, mn, lo, hi = orc.sketch_keys(pls[i])
        assert (kk, mm) == (31, 11) and (mn == mn_all[a:b]).all() and (lo == lo_all[a:b]).all() and not hi.any()
        sk = sp.sketch_parse(pls[i])
        assert (sk.minimizer == mn).all() and (sk.kmer_lo == lo).all()
        sets.append(set(zip(mn.tolist(), lo.tolist())))
    inter, card, _, _ = orc.compare(pls)
    assert [int(c) for c in card] == [len(x) for x in sets]
    for i in range(D.n):
        for j in range(i + 1, D.n):
            assert inter[i, j] == len(sets[i] & sets[j]), (i, j)
    assert inter[0, 1] > 0 and inter[0, 20] == 0 and inter[20, 39] > 0


def test_sketch_reader_empty_and_bad_input():
    payload, _ = orc.sketch_fasta(b">x\nACGT\n", 31, 11, 1000)
    sk = sp.sketch_parse(payload)
    assert len(sk) == 0 and (sk.k, sk.m) == (31, 11)
    with pytest.raises(sp.SpspError):
        sp.sketch_parse(b"no newline here")
    with pytest.raises(sp.SpspError):
        sp.sketch_parse(b"51 11 3 1000.000000\nACGTACGTACG\xff\xff\xff\x7f")


@pytest.mark.parametrize("prec,thr", [(6, 0.0), (3, 0.0), (12, 0.0), (6, 0.4), (1, 0.0), (0, 0.0)])
def test_csv_matches_oracle(prec, thr):
    rng = np.random.default_rng(prec + 17)
    n = 7
    card = rng.integers(50, 5000, size=n).astype(np.uint64)
    inter = np.zeros((n, n), dtype=np.uint32)
    for i in range(n):
        for j in range(i + 1, n):
            if rng.random() < 0.7:
                inter[i, j] = rng.integers(1, int(min(card[i], card[j])) + 1)
    inter[0, 1] = card[0] = card[1]  # J = C = 1
    names = ["dir/sk_%d.gz" % i for i in range(n)]
    for jac in (True, False):
        for nq in (n, 3):
            assert sp.csv(jac, names, inter, card, nq, prec, thr) == orc.csv(jac, names, inter, card, nq, prec, thr)


def test_csv_large_matrix_threaded_rows_match_oracle():
    rng = np.random.default_rng(3)
    n = 700                                           # 490 000 cells: formatted by several threads
    card = rng.integers(1000, 9000, size=n).astype(np.uint64)
    inter = np.triu(rng.integers(0, 1000, size=(n, n)), 1).astype(np.uint32)
    inter[rng.random((n, n)) < 0.5] = 0
    names = ["s%d" % i for i in range(n)]
    for jac in (True, False):
        assert sp.csv(jac, names, inter, card, n, 6, 0.0) == orc.csv(jac, names, inter, card, n, 6, 0.0)
        assert sp.csv(jac, names, inter, card, 10, 3, 0.05) == orc.csv(jac, names, inter, card, 10, 3, 0.05)


def test_csv_sparse_matrix_long_zero_runs_match_oracle():
    """a matrix of thousands of sketches is nearly all zeros: rows are written as runs of "0," copied from a constant
    (4096 cells at a time), broken by the diagonal, a few non-zero partners and the threshold rule -- bytes equal the
    oracle's cell-by-cell printer, also when a zero run is longer than the constant and for a row that is all zeros"""
    rng = np.random.default_rng(8)
    n = 4500
    card = rng.integers(3000, 9000, size=n).astype(np.uint64)
    inter = np.zeros((n, n), dtype=np.uint32)
    ii, jj = rng.integers(0, n, 3000), rng.integers(0, n, 3000)
    keep = ii < jj
    inter[ii[keep], jj[keep]] = rng.integers(1, 3000, int(keep.sum()))
    inter[7, :] = 0; inter[:, 7] = 0                   # sketch 7 shares nothing with anybody
    inter[0, n - 1] = 1234                             # first row: one partner at the far end behind 4 498 zeros
    names = ["g%d" % i for i in range(n)]
    for jac in (True, False):
        assert sp.csv(jac, names, inter, card, 12, 6, 0.0) == orc.csv(jac, names, inter, card, 12, 6, 0.0)
    assert sp.csv(True, names, inter, card, n, 4, 0.2) == orc.csv(True, names, inter, card, n, 4, 0.2)
    # ... and from the sparse form of the same matrix (what a large comparison returns): no n x n matrix at all
    ii, jj = np.nonzero(np.triu(inter, 1))
    cells = (ii.astype(np.uint64) << np.uint64(48)) | (jj.astype(np.uint64) << np.uint64(32)) | inter[ii, jj].astype(np.uint64)
    cells = cells[rng.permutation(len(cells))]
    for jac in (True, False):
        assert sp.csv_cells(jac, names, cells, card, 12, 6, 0.0) == orc.csv(jac, names, inter, card, 12, 6, 0.0)
        assert sp.csv_cells(jac, names, cells, card, n, 4, 0.2) == orc.csv(jac, names, inter, card, n, 4, 0.2)
    assert sp.csv_cells(True, names[:3], np.zeros(0, np.uint64), card[:3]) == orc.csv(True, names[:3], np.zeros((3, 3), np.uint32), card[:3])
    with pytest.raises(sp.SpspError):
        sp.csv_cells(True, names, np.concatenate([cells, cells[:1]]), card)                      # a pair twice
    with pytest.raises(sp.SpspError):
        sp.csv_cells(True, names[:10], np.array([(3 << 48) | (2 << 32) | 5], np.uint64), card[:10])   # not i < j


def test_csv_cells_straight_into_gzip_members(tmp_path):
    """spsp_csv_cells_gz_host (what spsp_compare_files writes its matrices with): rows go into gzip members without ever being
    text -- zero runs as deflate matches of distance 2 in a fixed-Huffman block, their CRC-32 by GF(2) operators.  Python's
    gzip (zlib: every member's CRC-32 and ISIZE are checked on the way) must read back exactly the text path's bytes: zero
    runs of every length class (1, 2, 3, 129, 130, 131, 4 097, 65 534 cells; runs that end the row and runs that do not),
    a last column that is a number / the diagonal / a zero, numbers with bytes >= 144 nowhere but names, query mode,
    the threshold rule, 1 x 1 and 2 x 2 matrices, and the containment header's blank line."""
    import gzip
    rng = np.random.default_rng(80)

    def check(n, pairs, nq=None, thr=0.0, prec=6, tag=""):
        card = rng.integers(3000, 9000, size=n).astype(np.uint64)
        cells = np.array([(i << 48) | (j << 32) | c for (i, j, c) in pairs], dtype=np.uint64)
        names = ["génome_%d.fa" % i for i in range(n)]            # (a byte >= 144 in every name: the 9-bit literals)
        for jac in (True, False):
            want = sp.csv_cells(jac, names, cells, card, nq, prec, thr)
            pth = str(tmp_path / ("m%s_%d_%d.csv.gz" % (tag, n, jac)))
            sp.csv_cells_gz(jac, names, cells, card, pth, nq, prec, thr)
            assert gzip.open(pth, "rb").read() == want, (tag, n, jac)
            assert sp.read_file(pth) == want                        # ... and through the library's own reader (zstr's member loop)

    check(1, [], tag="one")
    check(2, [(0, 1, 7)], tag="two")
    check(2, [], tag="two0")
    for n in (3, 4, 5, 131, 132, 133, 260, 4099):
        check(n, [(0, n - 1, 5), (1, 2, 9)] if n > 3 else [(0, 2, 5)], tag="ends")
        check(n, [(0, 1, 5)], tag="head")
        check(n, [], tag="empty")
    n = 70000 // 2                                                  # rows of 35 000 cells: runs of up to 34 999
    check(n, [(0, 17, 3), (5, n - 2, 8), (n - 2, n - 1, 4)], nq=6, tag="long")
    check(65535, [(0, 65534, 11), (1, 30000, 2)], nq=2, tag="max")
    # a random sparse matrix with the threshold rule and query rows
    n = 3000
    ii, jj = rng.integers(0, n, 5000), rng.integers(0, n, 5000)
    keep = ii < jj
    seen = set()
    pairs = []
    for i, j in zip(ii[keep].tolist(), jj[keep].tolist()):
        if (i, j) not in seen:
            seen.add((i, j)); pairs.append((i, j, int(rng.integers(1, 3000))))
    check(n, pairs, tag="rand")
    check(n, pairs, nq=40, thr=0.2, prec=4, tag="randq")
    # a dense block (one species): every cell a number
    n = 60
    check(n, [(i, j, 100 + i + j) for i in range(n) for j in range(i + 1, n)], tag="dense")


def test_gz_io_roundtrip(tmp_path):
    data = os.urandom(1000) + b"ACGT" * 100000
    p = str(tmp_path / "x.gz")
    sp.write_gz(p, data, 9)
    assert gzip.open(p, "rb").read() == data       # a real gzip container
    assert sp.read_file(p) == data
    q = str(tmp_path / "plain.txt")
    open(q, "wb").write(data)
    assert sp.read_file(q) == data                  # zstr autodetect: plain passes through
    r = str(tmp_path / "multi.gz")
    with open(r, "wb") as f:                        # concatenated members
        f.write(gzip.compress(b"hello "))
        f.write(gzip.compress(b"world"))
    assert sp.read_file(r) == b"hello world"
    big = os.urandom(1 << 16) * 600                  # 37.5 MiB -> three 16 MiB members, compressed on threads
    b = str(tmp_path / "big.gz")
    sp.write_gz(b, big, 1)
    assert gzip.open(b, "rb").read() == big and sp.read_file(b) == big
    import zlib
    z = str(tmp_path / "wrapped.z")
    open(z, "wb").write(zlib.compress(data, 6))     # zstr autodetect: a zlib wrapper inflates too
    assert sp.read_file(z) == data
    # the per-thread inflator is reset per member and per file: a tiny file after a big one, many members of ragged sizes
    assert sp.read_file(r) == b"hello world"
    parts = [os.urandom(n) for n in (0, 1, 4095, 4096, 4097, 70000, 3)]
    mz = str(tmp_path / "ragged.gz")
    open(mz, "wb").write(b"".join(gzip.compress(x, 1) for x in parts))
    assert sp.read_file(mz) == b"".join(parts)
    e = str(tmp_path / "empty")
    open(e, "wb").close()
    assert sp.read_file(e) == b""
    with pytest.raises(sp.SpspError):
        sp.read_file(str(tmp_path / "missing"))


def test_synth_generators_are_seeded():
    a = synth.family_genomes(7, 6, 2000, 2, [0.0, 0.01, 0.05])
    b = synth.family_genomes(7, 6, 2000, 2, [0.0, 0.01, 0.05])
    assert all((x == y).all() for x, y in zip(a, b))
    assert (a[0] != a[1]).sum() > 0 and (a[0] != a[3]).mean() > 0.5
    fa = synth.to_fasta(a[0], "g0", n_records=3)
    bases, off = sp.clean_fasta(fa)
    assert bases.tobytes() == a[0].tobytes() and len(off) == 4


def _jaccard_csv(rng, n, precision=6):
    """a symmetric all-vs-all Jaccard CSV as bin/comparator writes it (oracle printer), names in shuffled order"""
    card = rng.integers(500, 5000, size=n).astype(np.uint64)
    inter = np.zeros((n, n), dtype=np.uint32)
    for i in range(n):
        for j in range(i + 1, n):
            if rng.random() < 0.6:
                inter[i, j] = rng.integers(1, int(min(card[i], card[j])) + 1)
    names = ["out/sk_genome_%d.gz" % i for i in range(n)]
    order = rng.permutation(n)
    shuffled = [names[i] for i in order]
    inter_s = np.zeros_like(inter)
    for a in range(n):
        for b in range(a + 1, n):
            i, j = sorted((order[a], order[b]))
            inter_s[a, b] = inter[i, j]
    csv = orc.csv(True, shuffled, inter_s, card[order], n, precision, 0.0)
    return names, shuffled, csv


@pytest.mark.parametrize("n,precision", [(1, 6), (2, 6), (9, 6), (40, 3), (17, 12)])
def test_sort_csv_matches_oracle_and_restores_fof_order(n, precision):
    rng = np.random.default_rng(n * 31 + precision)
    names, shuffled, csv = _jaccard_csv(rng, n, precision)
    for fof in ("\n".join(names) + "\n", "\n".join(names), "\n".join(["unrelated.gz"] + names + ["tail.gz"]) + "\n"):
        want = orc.sort_csv(csv, fof.encode())
        assert want is not None
        got = sp.sort_csv(csv, fof.encode())
        assert got == want
        lines = got.decode().split("\n")
        assert lines[0].split(",") == names and len(lines) == n + 2 and lines[-1] == ""
    # sorting an already sorted file is the identity up to the six-digit reprint
    again = sp.sort_csv(got, ("\n".join(names) + "\n").encode())
    assert again == orc.sort_csv(got, ("\n".join(names) + "\n").encode())
    if precision <= 6:
        assert again == got


def test_sort_csv_large_sparse_matrix_matches_oracle():
    """sortCSV at the scale it is for (N3: the matrix of thousands of sketches, nearly all "0"): rows are parsed and printed by
    several threads, cells that are not "0" kept sparsely, zero runs copied -- bytes equal the oracle's cell-by-cell sortCSV;
    values that are not plain ("1e-05", "0.0", "-0") pass through strtod and %g as in the reference"""
    rng = np.random.default_rng(21)
    n = 1500
    names = ["d/s%04d.gz" % i for i in range(n)]
    card = rng.integers(3000, 9000, n).astype(np.uint64)
    ii, jj = rng.integers(0, n, 8 * n), rng.integers(0, n, 8 * n)
    keep = ii < jj
    cells = np.unique((ii[keep].astype(np.uint64) << np.uint64(48)) | (jj[keep].astype(np.uint64) << np.uint64(32))) | np.uint64(41)
    txt = sp.csv_cells(True, names, cells, card)
    lines = txt.split(b"\n")
    row = lines[3].split(b",")
    row[7], row[9], row[11] = b"1e-05", b"0.0", b"-0"                     # (cells (2, 7), (2, 9), (2, 11) of an otherwise sparse row)
    lines[3] = b",".join(row)
    txt = b"\n".join(lines)
    fof = ("\n".join(names[i] for i in rng.permutation(n)) + "\n").encode()
    want = orc.sort_csv(txt, fof)
    assert want is not None and sp.sort_csv(txt, fof) == want


def test_sort_csv_rejects_what_the_reference_mishandles():
    rng = np.random.default_rng(5)
    names, shuffled, csv = _jaccard_csv(rng, 6)
    fof = ("\n".join(names) + "\n").encode()
    assert sp.sort_csv(csv, fof) == orc.sort_csv(csv, fof)
    text = csv.decode()
    rows = text.split("\n")
    cases = {
        "name not in fof": (csv, ("\n".join(names[:-1]) + "\n").encode()),
        "duplicate column": ((",".join([shuffled[0]] * 2 + shuffled[2:]) + "\n" + "\n".join(rows[1:])).encode(), fof),
        "missing row": ("\n".join(rows[:-2]).encode() + b"\n", fof),
        "not a number": (text.replace(rows[2], "x" + rows[2][1:]).encode(), fof),
        "containment (blank line after the header)": ((rows[0] + "\n\n" + "\n".join(rows[1:])).encode(), fof),
        "diagonal not 1": (text.replace(rows[1], "0.5" + rows[1][1:], 1).encode() if rows[1].startswith("1,") else csv[:0], fof),
    }
    for what, (c, f) in cases.items():
        assert orc.sort_csv(c, f) is None, what
        with pytest.raises(sp.SpspError):
            sp.sort_csv(c, f)


def test_sort_csv_cli(tmp_path):
    rng = np.random.default_rng(12)
    names, shuffled, csv = _jaccard_csv(rng, 12)
    src, dst, fof = str(tmp_path / "j.csv.gz"), str(tmp_path / "sorted.csv"), str(tmp_path / "fof.txt")
    sp.write_gz(src, csv, 1)
    open(fof, "w").write("\n".join(names) + "\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "sortCSV"), src, dst, fof], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "The end" in r.stdout, (r.stdout, r.stderr)
    assert open(dst, "rb").read() == orc.sort_csv(csv, open(fof, "rb").read())
    r = subprocess.run([os.path.join(ROOT, "bin", "sortCSV"), src], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Need input" in r.stdout


@pytest.mark.parametrize("k,m", [(7, 7), (15, 15), (11, 11), (21, 11)])
def test_empty_sketches_follow_the_merges_first_read_rule(k, m):
    """Comparator.cpp:294,316-319: the N-way merge reads every file's first minimizer into one shared buffer without
    an end-of-file check, so with k == m an empty sketch inherits its predecessor's first minimizer as a phantom
    k-mer (a leading empty sketch gets AAA...).  sketches_from_payloads must reproduce what the oracle's literal
    merge counts; with m < k nothing changes."""
    rng = np.random.default_rng(k * 3 + m)
    anc = synth.random_genome(rng, 6000)
    short = synth.random_genome(rng, max(1, k - 2))                       # shorter than k: an empty sketch
    genomes = [short, anc, short, synth.mutate(rng, anc, 0.02), synth.random_genome(rng, 3000), short, short, anc[:2000]]
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, 3.0)[0] for i, g in enumerate(genomes)]
    assert sum(1 for pl in payloads if pl.count(b"\n") == 1) >= 4          # header-only payloads
    inter, card, kk, mm = orc.compare(payloads)
    sks = sp.sketches_from_payloads(payloads)
    sets = [s.key_set() for s in sks]
    assert [len(s) for s in sets] == [int(c) for c in card]
    for i in range(len(genomes)):
        for j in range(i + 1, len(genomes)):
            assert len(sets[i] & sets[j]) == inter[i, j], (i, j)
    plain = [len(sp.sketch_parse(pl)) for pl in payloads]
    if k == m:
        assert [len(s) for s in sets] != plain and len(sets[0]) == 1     # the phantom keys are there
    else:
        assert [len(s) for s in sets] == plain


def test_host_parsers_under_address_and_ub_sanitizers():
    """tests/tools/host_asan: spsp_host.cpp compiled with -fsanitize=address,undefined and fed valid and thousands of
    randomly corrupted sketch payloads, CSVs, file-of-files and FASTA text: every call returns, none crashes or reads
    out of bounds; the threaded host functions (CSV rows, parallel gzip members) run under ThreadSanitizer (sanitizers
    cannot run on the GPU side of this pool, so the host side is where they run)."""
    r = subprocess.run([os.path.join(ROOT, "tests", "tools", "host_asan", "run.sh"), "2500"], env=dict(os.environ, SPSP_ROOT=ROOT),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no crash" in r.stdout and "threaded sketch build OK" in r.stdout and r.stdout.count("the oracle's bytes") == 2, (r.stdout[-1500:], r.stderr[-3000:])
