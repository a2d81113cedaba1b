"""Pins for the CPU oracle (oracle/spsp_oracle.cpp) -- CPU only.

The reference has no tests or golden vectors for this path and cannot be
executed, so the oracle is pinned by first principles:
third-party python-xxhash, hand-derived packings, brute-force window minima
and Python set intersections (tests/bruteforce.py).
"""
import numpy as np
import pytest
from mpmath import mp, mpf

import bruteforce as bf
from oracle import oracle_py as orc

# XXH64(LE64(x), seed 1312) computed with python-xxhash
XXH_PINS = {
    0: 0x1cfdb444767b5459, 1: 0x42b41e8ff385b129, 2: 0xd1dd4292dc209b01, 3: 0xd330fd591c2f3422,
    2**22 - 1: 0xe98a81a0431c49e1, 2**22: 0xe66380dda3f126fe, 2**30 - 1: 0x7146ec6f33dc5b33,
    2**30: 0xdd60f3da31d4677f, 493447: 0x9bdb02e755c180db, 1216274: 0xc396683a0871c352,
    126322567: 0x202fe01db4b8d9e7,
}


def test_xxh64_pins():
    for x, h in XXH_PINS.items():
        assert orc.xxh64(x) == h
        assert bf.h64(x) == h


def test_xxh64_random_vs_python_xxhash():
    rng = np.random.default_rng(7)
    for x in rng.integers(0, 2**30, size=2000):
        assert orc.xxh64(int(x)) == bf.h64(int(x))
    for x in rng.integers(0, 2**63, size=200):
        assert orc.xxh64(int(x)) == bf.h64(int(x))


def test_encoding_pins():
    assert orc.str2num("ACGTACGTACG") == 493447
    assert orc.rc64(493447, 11) == 1973790 == bf.val("CGTACGTACGT")
    assert orc.canon64(493447, 11) == 493447
    assert orc.str2num("CATTAGGACAT") == 1216274
    assert orc.rc64(1216274, 11) == 759307 == bf.val("ATGTCCTAATG")
    assert orc.canon64(1216274, 11) == 759307
    assert orc.str2num("GATTACAGATTACAG") == 847563283
    assert orc.rc64(847563283, 15) == 461532681


def test_rc_random():
    rng = np.random.default_rng(3)
    for n in (11, 15, 7, 31, 32):
        for _ in range(50):
            s = bf.random_dna(rng, n)
            assert orc.rc64(bf.val(s), n) == bf.val(bf.rc_str(s))
    for n in (31, 33, 63):
        for _ in range(50):
            s = bf.random_dna(rng, n)
            assert orc.canon128(bf.val(s), n) == bf.canon_val(s)


def test_blob_packing_pins():
    assert orc.compress("ACGT") == bytes([0x00, 0x1e])
    assert orc.compress("ACGTA") == bytes([0x01, 0x1e, 0x00])
    assert orc.compress("ACGTAC") == bytes([0x02, 0x1e, 0x04])
    assert orc.compress("ACGTACG") == bytes([0x03, 0x1e, 0x1c])
    assert orc.compress("GGGGTTTTCCCCAAAA") == bytes([0x00, 0xff, 0xaa, 0x55, 0x00])
    assert orc.compress("") == b""
    rng = np.random.default_rng(5)
    for n in (4, 8, 40, 5, 6, 7, 41):
        s = bf.random_dna(rng, n)
        assert orc.decompress(orc.compress(s)) == s.encode()


def test_threshold_against_high_precision():
    mp.prec = 200
    for (k, m, s) in [(31, 11, 1000), (63, 15, 100), (31, 11, 2), (21, 11, 50), (31, 15, 10000)]:
        w = k - m + 1
        exact = (1 - (1 - mpf(1) / mpf(s)) ** (mpf(1) / w)) * mpf(2) ** 64
        got = orc.threshold(k, m, s)
        # 80-bit cancellation in (1 - root) plus the floor-and-double leave a few units
        assert abs(mpf(got) - exact) <= 8, (k, m, s, got)
        assert got % 2 == 0
    # SURVEY.md 8a A4 decimal anchors (low bits may differ by a few units)
    assert abs(orc.threshold(31, 11, 1000) - 878834950402620) < 64
    assert abs(orc.threshold(63, 15, 100) - 3783203295155380) < 64
    assert orc.threshold(31, 11, 1.0) == 2**64 - 1
    assert orc.threshold(31, 11, 0.5) == 2**64 - 1


def test_rescan_equals_bruteforce_minimum():
    """check (1): without a repeated m-mer in the window, the rescan returns the
    canonical m-mer of minimum hash, its position and strand (rightmost-reverse
    quirk SubSampler.cpp:89-93 aside)."""
    rng = np.random.default_rng(11)
    for (k, m) in [(31, 11), (21, 11), (63, 15), (15, 15)]:
        for _ in range(200):
            s = bf.random_dna(rng, k)
            cv, hs = bf.mmer_hashes(s, m)
            if len(set(cv)) != len(cv):
                continue
            best = min(range(len(cv)), key=lambda p: hs[p])
            mini, pos, rev = orc.rescan(k, m, s)
            assert mini == cv[best]
            is_rev = bf.val(s[best:best + m]) != cv[best]
            assert rev == int(is_rev)
            if best == k - m and is_rev:
                assert pos == 0
            else:
                assert pos == best


def _repeat_rich(rng, n):
    """DNA in which m-mers recur inside one window, on both strands: short tandem units, a small library of
    pieces and their reverse complements, low-complexity stretches."""
    lib = [bf.random_dna(rng, int(rng.integers(4, 40))) for _ in range(6)]
    out = []
    while sum(map(len, out)) < n:
        r = rng.random()
        piece = lib[int(rng.integers(0, len(lib)))]
        if r < 0.3:
            out.append(piece * int(rng.integers(2, 6)))
        elif r < 0.5:
            out.append(bf.rc_str(piece))
        elif r < 0.6:
            out.append("ACGT"[int(rng.integers(0, 4))] * int(rng.integers(5, 60)))
        elif r < 0.7:
            out.append(bf.mutate(rng, piece, 0.1))
        elif r < 0.8:
            out.append(bf.random_dna(rng, int(rng.integers(1, 30))))
        else:
            out.append(piece)
    return "".join(out)[:n]


def test_tie_rules_second_witness_rescan():
    """H5: windows that hold their minimizer more than once, on either strand.  The oracle's regular_minimizer_pos
    against bruteforce.model_rescan, a description of SubSampler.cpp:81-169 over occurrences (not a transcription of
    the loop) -- ~7 000 windows, ~2 000 of them with a repeated m-mer."""
    rng = np.random.default_rng(2718)
    checked = repeated = 0
    for (k, m) in [(31, 11), (21, 11), (15, 9), (33, 13), (63, 15), (15, 15), (17, 5)]:
        text = _repeat_rich(rng, 3000)
        for j in range(0, len(text) - k + 1, 3):
            s = text[j:j + k]
            cv = [bf.canon_val(s[p:p + m]) for p in range(k - m + 1)]
            repeated += len(set(cv)) != len(cv)
            assert orc.rescan(k, m, s) == bf.model_rescan(s, k, m), (k, m, s)
            checked += 1
    assert checked > 6000 and repeated > 1500


def test_tie_rules_second_witness_scan():
    """the whole scan loop on repeat-rich records: selected super-k-mers (start, length, minimizer, strand) and the
    count of ALL super-k-mers (the `dump` cuts of the believed position) against bruteforce.model_scan."""
    rng = np.random.default_rng(31415)
    for (k, m, s) in [(31, 11, 3), (21, 11, 1.0), (15, 9, 2), (33, 13, 5), (63, 15, 4), (15, 15, 2), (17, 5, 1.5)]:
        T = orc.threshold(k, m, s)
        recs = [_repeat_rich(rng, int(rng.integers(200, 1500))) for _ in range(6)] + [bf.random_dna(rng, 600), "A" * 200, "ACG" * 100]
        bases, offs = orc.clean_fasta(bf.fasta(recs))
        got, st = orc.scan(k, m, T, bases, offs)
        want, total = [], 0
        for r, seq in enumerate(recs):
            em, n_all = bf.model_scan(seq, k, m, T)
            want += [(r, a, b, c, d) for (a, b, c, d) in em]
            total += n_all
        assert [(int(e["rec"]), int(e["start"]), int(e["len"]), int(e["minimizer"]), int(e["rev"])) for e in got] == want, (k, m, s)
        assert st["total_superkmer_number"] == total, (k, m, s)


def _records(rng, lens):
    return [bf.random_dna(rng, n) for n in lens]


@pytest.mark.parametrize("k,m,s", [(31, 11, 20), (21, 11, 5), (63, 15, 10), (31, 11, 1), (15, 11, 3)])
def test_scan_covers_exactly_the_selected_kmers(k, m, s):
    """check (2): the union of emitted super-k-mers is exactly the set of k-mers
    whose window minimum hash is <= T, each emitted once, in genome order,
    labelled with that minimum's canonical m-mer."""
    rng = np.random.default_rng(100 + k + m)
    recs = _records(rng, [900, 10, k, 400, k + 1])
    T = orc.threshold(k, m, s)
    text = bf.fasta(recs)
    bases, offs = orc.clean_fasta(text)
    assert [int(x) for x in np.diff(offs)] == [len(r) for r in recs]
    em, st = orc.scan(k, m, T, bases, offs)
    got = []
    for e in em:
        for j in range(int(e["start"]), int(e["start"]) + int(e["len"]) - k + 1):
            got.append((int(e["rec"]), j, int(e["minimizer"])))
    want = []
    for r, seq in enumerate(recs):
        want += [(r, j, b) for j, b in bf.selected_kmers(seq, k, m, T)]
    assert got == want
    assert st["read_kmer"] == sum(max(0, len(r) - k + 1) for r in recs)
    assert st["total_kmer_number"] == st["read_kmer"]


def test_scan_orientation_flag():
    """rev flag == the minimizer occurrence reads reverse-complemented in the genome."""
    rng = np.random.default_rng(21)
    k, m = 31, 11
    seq = bf.random_dna(rng, 3000)
    T = orc.threshold(k, m, 8)
    bases, offs = orc.clean_fasta(bf.fasta([seq]))
    em, _ = orc.scan(k, m, T, bases, offs)
    assert len(em) > 8
    for e in em:
        sk = seq[int(e["start"]):int(e["start"]) + int(e["len"])]
        ms = bf.to_str(int(e["minimizer"]), m)
        if e["rev"]:
            assert ms in bf.rc_str(sk)
        else:
            assert ms in sk


@pytest.mark.parametrize("k,m,s", [(31, 11, 30), (21, 11, 4), (63, 15, 12)])
def test_sketch_roundtrip_recovers_selected_kmer_set(k, m, s):
    """check (3): payload -> k-mers equals the brute-force selected set."""
    rng = np.random.default_rng(k * 7 + m)
    recs = _records(rng, [2500, 1200, 30])
    payload, st = orc.sketch_fasta(bf.fasta(recs), k, m, s)
    T = orc.threshold(k, m, s)
    want = bf.sketch_set(recs, k, m, T)
    assert bf.payload_set(payload) == want
    hdr, buckets, order = bf.parse_payload(payload)
    assert hdr["k"] == k and hdr["m"] == m and hdr["skmer"] == 2 * k - m
    assert hdr["n"] == st["selected_kmer_number"]
    assert hdr["rate"] == "%f" % s
    assert [bf.val(x) for x in order] == sorted(bf.val(x) for x in order)
    assert st["seen_kmers_at_reconstruction"] == len(want)


def test_compare_equals_set_intersections():
    """check (4)+(5): inter/card equal Python set algebra; duplicates give J=C=1."""
    rng = np.random.default_rng(99)
    k, m, s = 31, 11, 25
    anc = bf.random_dna(rng, 6000)
    genomes = [[anc], [bf.mutate(rng, anc, 0.01)], [bf.mutate(rng, anc, 0.05), bf.random_dna(rng, 800)],
               [anc], [bf.random_dna(rng, 5000)]]
    T = orc.threshold(k, m, s)
    payloads = [orc.sketch_fasta(bf.fasta(g), k, m, s)[0] for g in genomes]
    sets = [bf.sketch_set(g, k, m, T) for g in genomes]
    inter, card, kk, mm = orc.compare(payloads)
    assert (kk, mm) == (k, m)
    n = len(genomes)
    for i in range(n):
        assert card[i] == len(sets[i])
        for j in range(i + 1, n):
            assert inter[i, j] == len(sets[i] & sets[j]), (i, j)
    assert inter[0, 3] == card[0] == card[3]
    names = ["g%d.gz" % i for i in range(n)]
    jac = orc.csv(True, names, inter, card).decode().split("\n")
    con = orc.csv(False, names, inter, card).decode().split("\n")
    assert jac[0] == ",".join(names) and con[0] == jac[0] and con[1] == ""
    row0 = jac[1].split(",")
    assert row0[0] == "1" and row0[3] == "1"
    x = int(inter[0, 1]) / (int(card[0]) + int(card[1]) - int(inter[0, 1]))
    assert row0[1] == "%.6g" % x
    crow = con[2 + 1].split(",")
    assert crow[0] == "%.6g" % (int(inter[0, 1]) / int(card[1]))


def test_compare_query_mode_rows():
    rng = np.random.default_rng(5)
    k, m, s = 21, 11, 10
    anc = bf.random_dna(rng, 3000)
    gs = [[bf.mutate(rng, anc, mu)] for mu in (0.0, 0.02, 0.03, 0.1)]
    T = orc.threshold(k, m, s)
    payloads = [orc.sketch_fasta(bf.fasta(g), k, m, s)[0] for g in gs]
    sets = [bf.sketch_set(g, k, m, T) for g in gs]
    inter, card, _, _ = orc.compare(payloads, n_query=1)
    for j in range(1, 4):
        assert inter[0, j] == len(sets[0] & sets[j])
    for i in range(4):
        assert card[i] == len(sets[i])


def test_fasta_cleaning_quirks():
    # first line always dropped, N's vanish and flanks join, lower-case folded, CRLF tolerated
    text = b">r1 desc\nACGTNNacgt\r\nGG\n>r2\n\nTTTT\n>empty\n>r4\nAC"
    bases, offs = orc.clean_fasta(text)
    recs = [bytes(bases[int(offs[i]):int(offs[i + 1])]).decode() for i in range(len(offs) - 1)]
    assert recs == ["ACGTACGTGG", "TTTT", "", "AC"]
    bases, offs = orc.clean_fasta(b"ACGT\nGGGG\n")  # no '>' header: first line still dropped
    assert bytes(bases).decode() == "GGGG"


def test_low_complexity_and_repeats_do_not_lose_kmers():
    """homopolymers / tandem repeats exercise the duplicate-m-mer tie rules
    (SubSampler.cpp:132-166): coverage must still be exact."""
    k, m = 31, 11
    rng = np.random.default_rng(8)
    unit = bf.random_dna(rng, 13)
    seqs = ["A" * 200, "ACGT" * 60, unit * 30, bf.random_dna(rng, 100) + "T" * 90 + bf.random_dna(rng, 100)]
    T = 2**64 - 1
    for seq in seqs:
        bases, offs = orc.clean_fasta(bf.fasta([seq]))
        em, st = orc.scan(k, m, T, bases, offs)
        cover = []
        for e in em:
            cover += list(range(int(e["start"]), int(e["start"]) + int(e["len"]) - k + 1))
        assert cover == list(range(len(seq) - k + 1))
        payload, _ = orc.sketch_fasta(bf.fasta([seq]), k, m, 1.0)
        assert bf.payload_set(payload) == bf.sketch_set([seq], k, m, T)


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 20.0, 1), (21, 11, 3.0, 1), (15, 7, 1.0, 1), (31, 11, 5.0, 2), (63, 15, 10.0, 1), (15, 15, 2.0, 1), (21, 9, 1.0, 3)])
def test_payload_second_witness(k, m, s, ab):
    """the oracle's sketch payload (handle_superkmer + emission + strCompressor restated in C++) against an independent
    description of the same reference text over Python strings and dicts (tests/bruteforce.py::model_payload, which
    takes its super-k-mers from model_scan): byte for byte, on random records, a mutated copy (shared k-mers: counts
    above 1), a reverse-complemented copy, a tandem repeat and -- for the uint8 count -- a unit repeated 257 times."""
    rng = np.random.default_rng(100 * k + m + ab)
    a = bf.random_dna(rng, 2500)
    unit = bf.random_dna(rng, 55)
    records = [a, bf.mutate(rng, a, 0.02), bf.rc_str(a[300:1500]), bf.random_dna(rng, 17) * 40, bf.random_dna(rng, k - 1), unit * 257 + unit[:30], a[:1200]]
    T = orc.threshold(k, m, s)
    want = bf.model_payload(records, k, m, T, s, ab)
    got, _ = orc.sketch_fasta(bf.fasta(records), k, m, s, ab)
    assert got == want
