"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the
C-ABI of libspsp.so and is compared with the CPU oracle on the same seeded
inputs; full-size inputs are checked through size-independent properties."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

import bruteforce as bf
import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = sp.Context(0)
    yield c
    c.close()


def _oracle_stream(k, m, thr, bases, offs):
    em, _ = orc.scan(k, m, thr, bases, offs)
    return em


def _assert_stream_equal(got, want):
    assert len(got) == len(want), (len(got), len(want))
    for f in ("rec", "minimizer", "start", "len", "rev"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, (f, int(bad[0]), got[bad[0]], want[bad[0]])


# (the blocked-Bloom variant exists for m = 13 and m = 15; with other m the flag leaves the choice to the library)
MODES = [sp.SPSP_SCAN_DIRECT_HASH, sp.SPSP_SCAN_LDS_FILTER, sp.SPSP_SCAN_PAIR_FILTER, sp.SPSP_SCAN_DEFAULT, sp.SPSP_SCAN_BLOOM_FILTER]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,m,s", [(31, 11, 1000), (31, 11, 20), (21, 11, 5), (63, 15, 10), (63, 15, 100),
                                   (15, 11, 3), (15, 15, 4), (33, 13, 50), (31, 11, 1.5)])
def test_scan_stream_equals_oracle(ctx, k, m, s, mode):
    rng = np.random.default_rng(1000 + k + m)
    gen = [synth.random_genome(rng, n) for n in (300_000, 17, k, 50_001, k - 1, 0, 120_000)]
    bases, offs = synth.concat_records(gen)
    p = sp.make_params(k, m, s, flags=mode)
    assert p.threshold == orc.threshold(k, m, s)
    got = ctx.scan(p, bases, offs)
    want = _oracle_stream(k, m, p.threshold, bases, offs)
    assert len(want) > 0
    _assert_stream_equal(got, want)


@pytest.mark.parametrize("mode", MODES)
def test_scan_select_all_and_low_complexity(ctx, mode):
    """s <= 1 selects every k-mer; homopolymers / tandem repeats drive the
    duplicate-minimizer tie rules (SubSampler.cpp:89-93, 132-166)."""
    rng = np.random.default_rng(5)
    unit = synth.random_genome(rng, 13)
    recs = [np.frombuffer(b"A" * 500, np.uint8), np.frombuffer(b"ACGT" * 200, np.uint8), np.tile(unit, 60),
            np.concatenate([synth.random_genome(rng, 300), np.frombuffer(b"T" * 200, np.uint8),
                            synth.random_genome(rng, 300)]),
            synth.random_genome(rng, 20_000)]
    bases, offs = synth.concat_records(recs)
    for (k, m, s) in [(31, 11, 1.0), (31, 11, 3), (21, 11, 1.0), (63, 15, 2)]:
        p = sp.make_params(k, m, s, flags=mode)
        _assert_stream_equal(ctx.scan(p, bases, offs), _oracle_stream(k, m, p.threshold, bases, offs))


def test_scan_repeat_rich_genome(ctx):
    """genome built from a small repeat library: many windows hold the same
    m-mer twice, on both strands."""
    rng = np.random.default_rng(77)
    lib = [synth.random_genome(rng, int(n)) for n in rng.integers(15, 60, size=12)]
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    parts = []
    for _ in range(4000):
        u = lib[int(rng.integers(0, len(lib)))]
        if rng.random() < 0.5:
            u = np.array([comp[int(c)] for c in u[::-1]], dtype=np.uint8)
        parts.append(u)
    g = np.concatenate(parts)
    bases, offs = synth.concat_records([g])
    for (k, m, s) in [(31, 11, 4), (21, 11, 1.0), (63, 15, 3)]:
        for mode in MODES:
            p = sp.make_params(k, m, s, flags=mode)
            _assert_stream_equal(ctx.scan(p, bases, offs), _oracle_stream(k, m, p.threshold, bases, offs))


def test_scan_randomised_configs(ctx):
    """40 seeded random configurations: odd k in 11..63, odd m in 3..15 (m <= k), s from select-all to sparse,
    genomes mixing random sequence, mutated copies, inverted repeats, homopolymer runs and short records."""
    rng = np.random.default_rng(31337)
    comp = np.zeros(256, np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    for it in range(40):
        m = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        k = int(rng.choice([x for x in range(max(m, 11), 64, 2)]))
        s = float(rng.choice([1.0, 1.5, 2, 3, 7, 20, 100, 1000]))
        base = synth.random_genome(rng, int(rng.integers(2_000, 60_000)))
        parts = [base, synth.mutate(rng, base, 0.02)[: len(base) // 2], comp[base[::-1]][: len(base) // 3]]
        if rng.random() < 0.5:
            parts.append(np.full(int(rng.integers(10, 300)), int(rng.choice([65, 67, 71, 84])), np.uint8))
        if rng.random() < 0.5:
            parts.append(np.tile(synth.random_genome(rng, int(rng.integers(2, 20))), int(rng.integers(5, 60))))
        rng.shuffle(parts)
        genome = np.concatenate(parts)
        cuts = sorted(set([0, len(genome)] + [int(x) for x in rng.integers(0, len(genome), size=int(rng.integers(0, 5)))]))
        recs = [genome[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        bases, offs = synth.concat_records(recs)
        mode = [sp.SPSP_SCAN_DEFAULT, sp.SPSP_SCAN_DIRECT_HASH, sp.SPSP_SCAN_LDS_FILTER, sp.SPSP_SCAN_PAIR_FILTER, sp.SPSP_SCAN_BLOOM_FILTER][it % 5]
        p = sp.make_params(k, m, s, flags=mode)
        got = ctx.scan(p, bases, offs)
        want = _oracle_stream(k, m, p.threshold, bases, offs)
        try:
            _assert_stream_equal(got, want)
        except AssertionError as e:
            raise AssertionError("config it=%d k=%d m=%d s=%g mode=%d n=%d: %s" % (it, k, m, s, mode, len(bases), e))


def test_scan_select_all_through_long_runs_without_a_reset(ctx):
    """-s 1 by segments (k_seg_scan) where a chain of rescans leaves its tile's halo -- 1 024 iterations without an m-mer that
    beats the window's minimum: the owning lane goes on alone over the bases.  Runs that end with a reset (random sequence
    behind a homopolymer), with the record (a homopolymer record, a short-period tandem repeat), inside the halo (500), and one
    longer than the lane walks (40 kb: the call goes to the dense + sparse passes); records that start inside such a run."""
    rng = np.random.default_rng(909)
    A = lambda c, n: np.full(n, ord(c), np.uint8)   # noqa: E731
    recs = [np.concatenate([synth.random_genome(rng, 5000), A("G", 3000), synth.random_genome(rng, 5000)]),
            A("A", 5000), np.tile(np.frombuffer(b"ACGGT", np.uint8), 2000),
            np.concatenate([A("T", 500), synth.random_genome(rng, 3000), A("C", 1500), A("A", 1500)]),
            synth.random_genome(rng, 30_000)]
    long_run = recs + [np.concatenate([synth.random_genome(rng, 2000), A("C", 40_000), synth.random_genome(rng, 2000)])]
    for group in (recs, long_run):
        bases, offs = synth.concat_records(group)
        for (k, m) in [(31, 11), (63, 15), (21, 11), (15, 15)]:
            p = sp.make_params(k, m, 1.0)
            _assert_stream_equal(ctx.scan(p, bases, offs), _oracle_stream(k, m, p.threshold, bases, offs))


def test_scan_by_segments_and_by_hits_agree_with_the_oracle():
    """the scan by segments (spsp_stats.hip::k_seg_scan: what a threshold that selects nearly every m-mer takes, -s 1) pinned
    for EVERY threshold (SPSP_DEBUG_SEG_SCAN=1: random configurations, the repeat library, low complexity -- chains that
    leave their tile hand the call to the product scan) and switched off (=0: -s 1 through the dense + sparse passes, as
    until round 5); each in a process of its own (the switch is read once)."""
    import subprocess
    import sys
    for val, names in (("1", ("test_scan_randomised_configs", "test_scan_repeat_rich_genome", "test_scan_edge_inputs")),
                       ("0", ("test_scan_every_mmer_selected_over_megabases",))):
        code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport supersampler_amd as sp, test_gpu\nctx = sp.Context(0)\n" % (ROOT, os.path.join(ROOT, "tests"))
                + "".join("test_gpu.%s(ctx)\n" % n for n in names)
                + "for mode in test_gpu.MODES: test_gpu.test_scan_select_all_and_low_complexity(ctx, mode)\nprint('ok')\n")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SPSP_DEBUG_SEG_SCAN=val), timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, (val, r.stderr[-2000:])


def test_scan_dense_selection_many_hits(ctx):
    """a dense selection over 3 Mbp: several hundred thousand hits and super-k-mers, so the write pass of the
    sparse stage sums more than 64 chunk totals and several 16-Ki-position tiles carry thousands of hits each"""
    rng = np.random.default_rng(77)
    gen = [synth.random_genome(rng, n) for n in (2_000_000, 999_983, 40)]
    bases, offs = synth.concat_records(gen)
    for k, m, s in [(15, 11, 2), (21, 15, 1.2)]:
        p = sp.make_params(k, m, s)
        got = ctx.scan(p, bases, offs)
        want = _oracle_stream(k, m, p.threshold, bases, offs)
        assert len(want) > 300_000
        _assert_stream_equal(got, want)


def test_scan_every_mmer_selected_over_megabases(ctx):
    """s = 1 (no sub-sampling): every m-mer is a hit, a record is ONE cluster of millions of hits -- replayed in pieces
    cut in front of every strict window minimum (head_kind), offsets over three levels of sums (> 8 x 10^6 hits need 40 Mbp:
    test_scan_40_mbp_stream_equals_oracle's shape at s = 1 below).  Random records, a repeat-rich one (equal m-mers inside one
    window: no strict minimum, long pieces) and a homopolymer run, k - m + 1 = 21 and 49."""
    rng = np.random.default_rng(2718)
    unit = synth.random_genome(rng, 37)
    gen = [synth.random_genome(rng, 1_500_000), np.tile(unit, 3000), synth.random_genome(rng, 700_001),
           np.concatenate([synth.random_genome(rng, 5000), np.frombuffer(b"G" * 3000, np.uint8), synth.random_genome(rng, 5000)])]
    bases, offs = synth.concat_records(gen)
    for k, m, s in [(31, 11, 1.0), (63, 15, 1.0)]:
        p = sp.make_params(k, m, s)
        got = ctx.scan(p, bases, offs)
        want = _oracle_stream(k, m, p.threshold, bases, offs)
        assert len(want) > 80_000
        _assert_stream_equal(got, want)
    # 10^7 hits: the third level of the write pass's offsets
    g = synth.random_genome(rng, 10_000_000)
    bases, offs = synth.concat_records([g, g[:1_000_000]])
    p = sp.make_params(21, 11, 1.0)
    _assert_stream_equal(ctx.scan(p, bases, offs), _oracle_stream(21, 11, p.threshold, bases, offs))


def test_scan_40_mbp_stream_equals_oracle(ctx):
    """40 Mbp in 23 records: 2 441 tiles, i.e. several scan segments, tens of thousands of waves in the expand pass and
    (at s = 5) more than 64 chunk sums in the write pass -- the full stream against the oracle, not just invariants."""
    rng = np.random.default_rng(4040)
    lens = [int(x) for x in rng.integers(20, 4_000_000, size=22)] + [31]
    total = 40_000_000
    lens = [max(1, int(x * (total - 31) / sum(lens[:-1]))) for x in lens[:-1]] + [31]
    bases, offs = synth.concat_records([synth.random_genome(rng, n) for n in lens])
    for k, m, s, mode in [(31, 11, 1000, sp.SPSP_SCAN_DEFAULT), (31, 11, 50, sp.SPSP_SCAN_PAIR_FILTER), (31, 11, 5, sp.SPSP_SCAN_DEFAULT),
                          (63, 15, 100, sp.SPSP_SCAN_DEFAULT), (21, 9, 30, sp.SPSP_SCAN_DIRECT_HASH)]:
        p = sp.make_params(k, m, s, flags=mode)
        got = ctx.scan(p, bases, offs)
        want = _oracle_stream(k, m, p.threshold, bases, offs)
        assert len(want) > 1000
        _assert_stream_equal(got, want)


def test_scan_edge_inputs(ctx):
    p = sp.make_params(31, 11, 10)
    assert len(ctx.scan(p, np.zeros(0, np.uint8), np.zeros(1, np.uint64))) == 0
    short = np.frombuffer(b"ACGTACGTAC", np.uint8)
    assert len(ctx.scan(p, short, np.array([0, 10], np.uint64))) == 0
    rng = np.random.default_rng(3)
    g = synth.random_genome(rng, 16384 * 3)  # exact tile multiple: halo of the last tile is past the end
    bases, offs = synth.concat_records([g])
    _assert_stream_equal(ctx.scan(p, bases, offs), _oracle_stream(31, 11, p.threshold, bases, offs))
    with pytest.raises(sp.SpspError):
        ctx.scan(sp.make_params(31, 17, 10), bases, offs)
    with pytest.raises(sp.SpspError):
        ctx.scan(p, bases, np.array([5, 10], np.uint64))


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 200, 1), (63, 15, 40, 1), (21, 11, 6, 2)])
def test_sketch_payload_bytes_equal_oracle(ctx, k, m, s, ab):
    gs = synth.family_genomes(11, 3, 150_000, 1, [0.0, 0.01, 0.05])
    for i, g in enumerate(gs):
        text = synth.to_fasta(g, "g%d" % i, n_records=1 + i)
        got, gst = ctx.sketch_fasta(text, k, m, s, ab)
        want, wst = orc.sketch_fasta(text, k, m, s, ab)
        assert got == want
        assert gst["selected_kmer_number"] == wst["selected_kmer_number"]
        assert gst["nb_mmer_selected"] == wst["nb_mmer_selected"]


@pytest.mark.parametrize("k,m,s", [(31, 11, 100), (63, 15, 30), (21, 11, 10)])
def test_compare_equals_oracle(ctx, k, m, s):
    gs = synth.family_genomes(21, 9, 60_000, 3, [0.0, 0.005, 0.03])
    gs.append(gs[0].copy())                       # duplicate: J = C = 1
    gs.append(np.frombuffer(b"ACGT", np.uint8))   # empty sketch
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)]
    want_inter, want_card, _, _ = orc.compare(payloads)
    sketches = [sp.sketch_parse(p) for p in payloads]
    inter, card = ctx.compare(sketches)
    assert (card == want_card).all()
    assert (inter == want_inter).all()
    n = len(gs)
    assert inter[0, n - 2] == card[0] == card[n - 2] > 0
    assert want_inter.sum() > 0 and card[n - 1] == 0


def test_compare_query_mode_rows_only(ctx):
    """-q mode: only the query rows (the ones the printers emit) are computed, and they equal the reference's."""
    k, m, s = 31, 11, 30
    gs = synth.family_genomes(8, 12, 30_000, 2, [0.0, 0.01, 0.03])
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)]
    nq = 3
    want, wcard, _, _ = orc.compare(payloads, n_query=nq)
    inter, card = ctx.compare([sp.sketch_parse(p) for p in payloads], n_query=nq)
    assert (card == wcard).all()
    assert (inter[:nq] == want[:nq]).all() and want[:nq].sum() > 0
    assert (inter[nq:] == 0).all()
    names = ["s%d" % i for i in range(len(gs))]
    for jac in (True, False):
        assert sp.csv(jac, names, inter, card, nq) == orc.csv(jac, names, want, wcard, nq)


def test_compare_many_sketches_multiword_columns(ctx):
    """N > 64 so the colour matrix has several words per row and the triangle
    skips leading words."""
    k, m, s = 31, 11, 20
    gs = synth.family_genomes(5, 150, 6_000, 15, [0.0, 0.01, 0.03, 0.08])
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)]
    want_inter, want_card, _, _ = orc.compare(payloads)
    inter, card = ctx.compare([sp.sketch_parse(p) for p in payloads])
    assert (card == want_card).all() and (inter == want_inter).all()
    assert np.count_nonzero(want_inter) > 500


@pytest.mark.parametrize("n,use_hi", [(3, False), (64, False), (65, True), (130, False), (200, True)])
def test_compare_synthetic_key_sets(ctx, n, use_hi):
    """spsp_compare on hand-made key sets (no sketching involved): intersections must equal Python set algebra,
    for 1, 2 and 4 colour words per row, with and without the high k-mer word (k > 32), empty and identical sets."""
    rng = np.random.default_rng(n * 7 + use_hi)
    universe = [(int(rng.integers(0, 2**22)), int(rng.integers(0, 2**62)) if use_hi else 0, int(rng.integers(0, 2**62)))
                for _ in range(3000)]
    # same k-mer under two different minimizers must stay two different keys
    universe += [(u[0] ^ 1, u[1], u[2]) for u in universe[:50]]
    sets = []
    for i in range(n):
        if i % 17 == 5:
            sets.append(set())
        elif i % 11 == 3 and i > 0:
            sets.append(set(sets[i - 1]))
        else:
            pick = rng.random(len(universe)) < rng.choice([0.01, 0.1, 0.4])
            sets.append({universe[j] for j in np.nonzero(pick)[0]})
    sketches = []
    for st in sets:
        keys = sorted(st)
        sketches.append(sp.Sketch(63 if use_hi else 31, 11, np.array([x[0] for x in keys], np.uint32),
                                  np.array([x[2] for x in keys], np.uint64), np.array([x[1] for x in keys], np.uint64)))
    inter, card = ctx.compare(sketches)
    for i in range(n):
        assert card[i] == len(sets[i])
        for j in range(n):
            want = len(sets[i] & sets[j]) if j > i else 0
            assert inter[i, j] == want, (i, j, int(inter[i, j]), want)


def test_compare_fingerprint_collision_retry():
    """fingerprints cut to 10 bits on the first attempt: distinct keys collide, the full-key check in k_fill
    notices and the dictionary is rebuilt with a new seed -- results unchanged."""
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "import supersampler_amd as sp\n"
        "rng = np.random.default_rng(1)\n"
        "sets = [set(map(int, rng.integers(0, 5000, size=2000))) for _ in range(6)]\n"
        "sk = [sp.Sketch(31, 11, np.zeros(len(s), np.uint32) + 7, np.array(sorted(s), np.uint64), np.zeros(len(s), np.uint64)) for s in sets]\n"
        "ctx = sp.Context(0)\n"
        "inter, card = ctx.compare(sk)\n"
        "assert all(inter[i, j] == len(sets[i] & sets[j]) for i in range(6) for j in range(i + 1, 6))\n"
        "assert [int(c) for c in card] == [len(s) for s in sets]\n"
        "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_FP_BITS="10"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_compare_rejects_unsorted_keys(ctx):
    sk = sp.sketch_parse(orc.sketch_fasta(synth.to_fasta(synth.random_genome(np.random.default_rng(1), 20000)),
                                          31, 11, 10)[0])
    assert len(sk) > 10
    sk.kmer_lo[[2, 3]] = sk.kmer_lo[[3, 2]]
    sk.minimizer[[2, 3]] = sk.minimizer[[3, 2]]
    with pytest.raises(sp.SpspError):
        ctx.compare([sk, sk])


def test_cli_end_to_end_matches_oracle(tmp_path):
    """bin/sub_sampler + bin/comparator: the drop-in CLIs produce the sketch
    payloads and CSVs of the reference algorithm."""
    k, m, s = 31, 11, 50
    gs = synth.family_genomes(42, 6, 80_000, 2, [0.0, 0.01, 0.04])
    names = []
    for i, g in enumerate(gs):
        path = tmp_path / ("genome%d.fa" % i)
        data = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
        if i % 2:
            path = tmp_path / ("genome%d.fa.gz" % i)
            path.write_bytes(gzip.compress(data))
        else:
            path.write_bytes(data)
        names.append(str(path))
    fof = tmp_path / "genomes.txt"
    fof.write_text("\n".join(names) + "\n")
    env = dict(os.environ)
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", str(fof), "-k", str(k), "-m", str(m), "-s",
                        str(s), "-t", "1", "-p", "sk_"], cwd=tmp_path, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    listed = (tmp_path / "sk_genomes.txt").read_text().split()
    assert listed == ["sk_genome%d.gz" % i for i in range(len(gs))]
    payloads = []
    for i, nm in enumerate(listed):
        got = gzip.open(tmp_path / nm, "rb").read()
        text = sp.read_file(names[i])
        want, _ = orc.sketch_fasta(text, k, m, float(np.float32(s)))
        assert got == want, nm
        payloads.append(want)
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "sk_genomes.txt", "-o", "res", "-p", "5"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    inter, card, _, _ = orc.compare(payloads)
    for jac, fn in ((True, "res_jaccard.csv.gz"), (False, "res_containment.csv.gz")):
        assert gzip.open(tmp_path / fn, "rb").read() == orc.csv(jac, listed, inter, card, None, 5, 0.0)


def _fmt_commas(n):
    return "{:,}".format(int(n))


def test_count_all_superkmers_matches_oracle(ctx):
    """total_superkmer_number of print_stat (SubSampler.cpp:430,452): every super-k-mer of the input, cut by the
    literal state machine incl. its `dump` cuts on repeated m-mers -- random, mutated-repeat, tandem-repeat and
    homopolymer records (no state-resetting event for whole chunks: the second pass), short and empty records."""
    import torch
    rng = np.random.default_rng(321)
    unit = synth.random_genome(rng, 37)
    lib = synth.random_genome(rng, 300)
    pieces = [synth.random_genome(rng, 40_000), np.tile(unit, 400), np.full(9_000, ord("A"), np.uint8),
              np.concatenate([lib if i % 3 else synth.mutate(rng, lib, 0.02) for i in range(40)]),
              synth.random_genome(rng, 10), np.zeros(0, np.uint8), np.tile(np.frombuffer(b"ACG", np.uint8), 2000),
              synth.random_genome(rng, 3_000), np.full(33, ord("C"), np.uint8)]
    bases, off = synth.concat_records(pieces)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    for k, m in ((31, 11), (21, 11), (63, 15), (15, 15), (33, 13), (15, 9)):
        p = sp.make_params(k, m, 50)
        _, st = orc.scan(k, m, p.threshold, bases, off)
        got = ctx.count_superkmers_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(pieces))
        assert got == st["total_superkmer_number"], (k, m, got, st)
    # one long random record: chunks far from the record start
    g = synth.random_genome(rng, 700_000)
    b2, o2 = synth.concat_records([g])
    d_b2 = torch.from_numpy(b2).cuda()
    d_o2 = torch.from_numpy(o2.view(np.int64)).cuda()
    p = sp.make_params(31, 11, 1000)
    _, st = orc.scan(31, 11, p.threshold, b2, o2)
    assert ctx.count_superkmers_device(p, d_b2.data_ptr(), len(b2), d_o2.data_ptr(), 1) == st["total_superkmer_number"]


@pytest.mark.gpu
def test_count_all_superkmers_chunk_replay_form_matches_oracle():
    """the statistics pass's first form (one lane replays a chunk of 512 iterations; SPSP_DEBUG_STATS=chunks, and the
    fallback of the segment form when thousands of chains leave their tiles) on the same records, in a process of its own"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport supersampler_amd as sp, test_gpu\n"
            "ctx = sp.Context(0)\ntest_gpu.test_count_all_superkmers_matches_oracle(ctx)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for form in ("chunks", "tiny"):          # tiny: the segment form hands a call over when more than ONE chain leaves its tile
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SPSP_DEBUG_STATS=form), timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (form, r.stderr[-2000:])


def test_cli_default_verbose_stdout_matches_reference_lines(tmp_path):
    """-v 1 (the default): print_stat's lines (SubSampler.cpp:633-665) with the numbers of the oracle, in the
    reference's wording and order, and the comparator's progress lines (Comparator.cpp:56,69,364,414,494,498,503,509)."""
    k, m, s = 31, 11, 20
    gs = synth.family_genomes(5, 2, 60_000, 1, [0.0, 0.02])
    names = []
    for i, g in enumerate(gs):
        path = tmp_path / ("v%d.fa" % i)
        path.write_bytes(synth.to_fasta(g, "g%d" % i, n_records=2))
        names.append(str(path))
    (tmp_path / "g.txt").write_text("\n".join(names) + "\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "g.txt", "-k", str(k), "-m", str(m), "-s", str(s), "-t", "1"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert out.startswith(" I use k=31 m=11 s=20\nMaximal super kmer are of length 51 or 21 kmers\n")
    at = 0
    for i, nm in enumerate(names):
        _, st = orc.sketch_fasta(open(nm, "rb").read(), k, m, float(np.float32(s)))
        want = [nm,
                "I have seen %s kmers and I selected %s kmers" % (_fmt_commas(st["total_kmer_number"]), _fmt_commas(st["selected_kmer_number"])),
                "After removing duplicate kmers, I selected %s kmers" % _fmt_commas(st["seen_kmers_at_reconstruction"]),
                "This means a practical subsampling rate of ", "This means a practical subsampling rate of ",
                "I have seen %s superkmers and I selected %s superkmers" % (_fmt_commas(st["total_superkmer_number"]), _fmt_commas(st["selected_superkmer_number"])),
                "After reconstruction and filtering with abundance, I have selected %s superkmers" % _fmt_commas(st["seen_superkmers_at_reconstruction"]),
                "This means a practical subsampling rate of ", "This means a practical subsampling rate of ",
                "This means a mean superkmer size of ", "This means a mean superkmer size of ", "This means a mean superkmer size of ",
                "Actual output file size is ", "This mean ", "Minimizer number: %s Skmer/minimizer:  " % _fmt_commas(st["actual_minimizer_number"]),
                "Minimizer number: %s Skmer/minimizer without duplicates: " % _fmt_commas(st["actual_minimizer_number"]),
                "Density is: ", "Number of maximal skmer was:       %s" % _fmt_commas(st["count_maximal_skmer"]),
                "Actual number of maximal skmer is: %s" % _fmt_commas(st["seen_max_superkmers_at_reconstruction"]),
                "Proportion of max skmers:        ", "Actual proportion of max skmers: "]
        for line in want:
            nxt = out.find(line, at)
            assert nxt >= 0, (line, out[at:at + 400])
            at = nxt + len(line)
    # a number printed with cout's default formatting: total / selected super-k-mers, 6 significant digits
    _, st0 = orc.sketch_fasta(open(names[0], "rb").read(), k, m, float(np.float32(s)))
    assert ("This means a mean superkmer size of %g kmer per superkmer in the input" % (st0["total_kmer_number"] / st0["total_superkmer_number"])) in out
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "subsampled_g.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == "No query file, I will perform a all versus all comparison" and lines[1] == "I found 2 documents"
    assert lines[2] == "kmers evaluated are of length: 31 minimizer size is 11" and lines[3] == "Comparisons done"
    assert lines[4].startswith("Comparisons lasted ") and lines[4].endswith(" sec")
    assert lines[5] == "Containement index dump " and lines[6] == "Jackard index dump"
    assert lines[7].startswith("Jaccard output lasted ") and lines[7].endswith(" sec")


def test_cli_k_equals_m_with_empty_sketches(tmp_path):
    """k == m = 15 and genomes shorter than k in the list: their sketches are header-only, and the comparator's merge
    gives each of them its predecessor's first minimizer as a phantom k-mer (Comparator.cpp:294,316-319).  The
    drop-in CLIs must print the reference algorithm's matrices here too."""
    k = m = 15
    s = 4
    rng = np.random.default_rng(99)
    anc = synth.random_genome(rng, 30_000)
    gs = [synth.random_genome(rng, 9), anc, synth.mutate(rng, anc, 0.01), synth.random_genome(rng, 14),
          synth.random_genome(rng, 3), synth.random_genome(rng, 20_000), synth.mutate(rng, anc, 0.05)]
    names = []
    for i, g in enumerate(gs):
        path = tmp_path / ("genome%d.fa" % i)
        path.write_bytes(synth.to_fasta(g, "g%d" % i))
        names.append(str(path))
    (tmp_path / "genomes.txt").write_text("\n".join(names) + "\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "genomes.txt", "-k", str(k), "-m", str(m), "-s", str(s),
                        "-t", "1", "-p", "sk_"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    listed = (tmp_path / "sk_genomes.txt").read_text().split()
    payloads = []
    for i, nm in enumerate(listed):
        got = gzip.open(tmp_path / nm, "rb").read()
        want, _ = orc.sketch_fasta(sp.read_file(names[i]), k, m, float(np.float32(s)))
        assert got == want, nm
        payloads.append(want)
    assert sum(1 for pl in payloads if pl.count(b"\n") == 1) == 3
    inter, card, _, _ = orc.compare(payloads)
    assert int(card[0]) == 1 and int(card[3]) == 1 and int(card[4]) == 1 and int(inter[2, 3]) == 1   # the phantoms count
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "sk_genomes.txt", "-o", "res"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for jac, fn in ((True, "res_jaccard.csv.gz"), (False, "res_containment.csv.gz")):
        assert gzip.open(tmp_path / fn, "rb").read() == orc.csv(jac, listed, inter, card, None, 6, 0.0)


def test_cli_query_mode_abundance_and_flag_rules(tmp_path):
    """-q query fof (rows = queries, columns = queries then index), -a abundance, -m/-p of the comparator, and the
    sketcher's flag rules: even k and m are bumped to odd, m is clamped to 15 (SubSampler.cpp:732-746)."""
    gs = synth.family_genomes(77, 5, 60_000, 1, [0.0, 0.01, 0.03])
    # every genome twice in the FASTA so that abundance 2 keeps (almost) everything
    idx, qry = [], []
    for i, g in enumerate(gs):
        path = tmp_path / ("s%d.fasta" % i)
        path.write_bytes(synth.to_fasta(g, "a%d" % i) + synth.to_fasta(g, "b%d" % i))
        (qry if i < 2 else idx).append(str(path))
    (tmp_path / "q.txt").write_text("\n".join(qry) + "\n")
    (tmp_path / "i.txt").write_text("\n".join(idx) + "\n")
    exe = os.path.join(ROOT, "bin", "sub_sampler")
    for fof in ("q.txt", "i.txt"):
        r = subprocess.run([exe, "-f", fof, "-k", "30", "-m", "16", "-s", "40", "-a", "2", "-t", "2", "-v", "0"],
                           cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Kmer size must be odd" in r.stdout and "Minimizer size must be odd" in r.stdout
        assert "Minimizer size can't be greater than 15." in r.stdout and " I use k=31 m=15 s=40" in r.stdout
    k, m, s = 31, 15, float(np.float32(40))
    payloads, names = [], []
    for fof in ("subsampled_q.txt", "subsampled_i.txt"):
        for nm in sorted((tmp_path / fof).read_text().split()):
            names.append(nm)
    # the .txt lists are written in processing order (2 threads); sort both sides the same way
    q_names = sorted((tmp_path / "subsampled_q.txt").read_text().split())
    i_names = sorted((tmp_path / "subsampled_i.txt").read_text().split())
    (tmp_path / "sq.txt").write_text("\n".join(q_names) + "\n")
    (tmp_path / "si.txt").write_text("\n".join(i_names) + "\n")
    order = q_names + i_names
    for nm in order:
        src = tmp_path / (nm[len("subsampled_"):-3] + ".fasta")
        want, _ = orc.sketch_fasta(src.read_bytes(), k, m, s, 2)
        assert gzip.open(tmp_path / nm, "rb").read() == want, nm
        payloads.append(want)
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-q", "sq.txt", "-f", "si.txt", "-o", "qres", "-p", "4",
                        "-m", "0.2"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "I query 2 file(s) against the bank" in r.stdout
    inter, card, _, _ = orc.compare(payloads, n_query=2)
    for jac, fn in ((True, "qres_jaccard.csv.gz"), (False, "qres_containment.csv.gz")):
        assert gzip.open(tmp_path / fn, "rb").read() == orc.csv(jac, order, inter, card, 2, 4, 0.2)


FASTA_TEXTS = [
    b">r1 desc\nACGTNNacgt\r\nGG\n>r2\n\nTTTT\n>empty\n>r4\nAC",
    b"ACGT\nGGGG\n", b"", b">only header", b">h\n", b"\n\n>x\nAC\n\n\nGT\n>y\nNNNN\n>z\nacgtn",
    b">a\nAC\xffGT\n\xff\nTT\n", b">a\nACGT", b"\n", b">\n>\n>", b"ACGT",
]


def _random_fasta_text(rng, n_lines, max_len):
    out = []
    for _ in range(n_lines):
        r = rng.random()
        if r < 0.08:
            out.append(b">" + bytes(rng.integers(32, 127, size=int(rng.integers(0, 40))).astype(np.uint8)))
        elif r < 0.10:
            out.append(b"")
        elif r < 0.12:
            out.append(b"\xff" + bytes(synth.random_genome(rng, int(rng.integers(0, 30)))))
        else:
            L = int(rng.integers(1, max_len))
            line = synth.random_genome(rng, L).copy()
            if rng.random() < 0.3:
                line[rng.integers(0, L, size=max(1, L // 10))] = ord("N")
            if rng.random() < 0.3:
                line = np.frombuffer(bytes(line).lower(), dtype=np.uint8).copy()
            if rng.random() < 0.1:
                line = np.concatenate([line, np.frombuffer(b"\r", np.uint8)])
            out.append(bytes(line))
    sep = b"\n"
    return sep.join(out) + (b"\n" if rng.random() < 0.5 else b"")


def test_gpu_ingest_matches_oracle(ctx):
    """row N1: getLineFasta + clean_dna on the GPU == the oracle's cleaned records and offsets, including the
    first-line rule, empty records, CR/LF, lower case, N's, 0xFF and texts spanning many 4 KiB tiles."""
    import torch
    rng = np.random.default_rng(2024)
    texts = list(FASTA_TEXTS)
    texts += [_random_fasta_text(rng, n, ml) for (n, ml) in ((30, 80), (400, 120), (3000, 90), (50, 20000), (5, 70000))]
    texts.append(synth.to_fasta(synth.random_genome(rng, 300_000), "g", n_records=3))
    texts.append(b">x\n" + bytes(synth.random_genome(rng, 4096 * 3 - 3)))          # newline exactly at tile seams
    texts.append(b"A" * 4095 + b"\n>" + b"C" * 4094 + b"\nG" * 3000)
    # 38 MB: more than two rounds of the single-workgroup tile scan (4 096 tiles x 4 tiles per lane per round), with
    # lower case, N runs and headers sprinkled in
    big = np.frombuffer(synth.to_fasta(synth.random_genome(rng, 37_000_000), "big", n_records=7), dtype=np.uint8).copy()
    for a in rng.integers(0, len(big) - 200, size=2000):
        big[a:a + int(rng.integers(1, 150))] = ord("N") if rng.random() < 0.5 else ord("a")
    texts.append(bytes(big))
    for text in texts:
        want_b, want_o = orc.clean_fasta(text)
        d = torch.from_numpy(np.frombuffer(text + b"\0" * 16, dtype=np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        db, nb, do, nr = ctx.clean_fasta_device(d.data_ptr(), len(text))
        assert nr == len(want_o) - 1 and nb == len(want_b), (len(text), nr, nb, len(want_o) - 1, len(want_b))
        got_b = ctx.to_host(db, nb, np.uint8)
        got_o = ctx.to_host(do, nr + 1, np.uint64)
        assert got_o.tolist() == want_o.tolist()
        assert got_b.tobytes() == want_b.tobytes()
        # the same ingest writing 2-bit words (16 bases per dword, first base in bits 31:30, zero tail): the oracle's
        # cleaned bases, packed with numpy
        dp, nb2, do2, nr2 = ctx.clean_fasta_packed_device(d.data_ptr(), len(text))
        assert (nb2, nr2) == (nb, nr) and ctx.to_host(do2, nr + 1, np.uint64).tolist() == want_o.tolist()
        n_dw = (nb + 15) // 16
        codes = np.zeros(n_dw * 16, dtype=np.uint32)
        codes[:nb] = (want_b.astype(np.uint32) >> 1) & 3
        want_w = (codes.reshape(-1, 16) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
        got_w = ctx.to_host(dp, n_dw + 64, np.uint32)
        assert (got_w[:n_dw] == want_w).all() and not got_w[n_dw:].any()


@pytest.mark.parametrize("k,m,s", [(31, 11, 100), (31, 11, 1000), (63, 15, 20), (21, 11, 1.0)])
def test_sketch_text_gpu_ingest_equals_oracle(ctx, k, m, s):
    rng = np.random.default_rng(k + m)
    g = synth.random_genome(rng, 200_000)
    text = synth.to_fasta(g[:120_000], "a", n_records=2) + b">short\nACGT\n>n\nNNNNNN\n" + synth.to_fasta(g[120_000:], "b")
    text = text.replace(b"ACGTA", b"acgNa", 50)
    got, gst = ctx.sketch_text(text, k, m, s, flags=sp.SPSP_SCAN_STATS)
    want, wst = orc.sketch_fasta(text, k, m, s)
    assert got == want
    for f in ("selected_kmer_number", "read_kmer", "nb_mmer_selected", "seen_kmers_at_reconstruction", "total_superkmer_number"):
        assert gst[f] == wst[f], f


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 20, 2), (31, 11, 1.0, 3), (63, 15, 5, 2), (21, 11, 1.0, 2), (15, 15, 1.0, 2), (33, 13, 3, 300)])
def test_abundance_filter_on_device_equals_oracle(ctx, k, m, s, ab):
    """N4: -a > 1 is counted on the GPU (spsp_abund.hip) and the host builder indexes the usable k-mers only.  Segments
    occur once, twice, three times and (a tandem unit) 256+ times -- the uint8 count wraps (H4) -- on both strands;
    payload bytes and every statistic the reference prints must equal the oracle's (SubSampler.cpp:243-302,587,608)."""
    rng = np.random.default_rng(1000 * k + ab)
    a, b, c = (synth.random_genome(rng, n) for n in (30_000, 20_000, 8_000))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    unit = synth.random_genome(rng, 70).tobytes()
    recs = [a.tobytes() + b.tobytes(), b.tobytes()[::-1].translate(comp) + c.tobytes(), c.tobytes() + a.tobytes()[:10_000] + c.tobytes(),
            unit * 257 + unit[:50], synth.random_genome(rng, 5_000).tobytes()]
    text = b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(recs))
    got, gst = ctx.sketch_text(text, k, m, s, abundance=ab)
    want, wst = orc.sketch_fasta(text, k, m, s, ab)
    assert got == want
    for f in ("selected_kmer_number", "selected_superkmer_number", "count_maximal_skmer", "seen_kmers_at_reconstruction",
              "seen_superkmers_at_reconstruction", "seen_max_superkmers_at_reconstruction", "actual_minimizer_number",
              "read_kmer", "nb_mmer_selected"):
        assert gst[f] == wst[f], f
    if ab < 300 and k != m:   # (k == m: a bucket is its minimizer, nothing else is written either way)
        assert 0 < wst["seen_superkmers_at_reconstruction"] and len(want) < len(orc.sketch_fasta(text, k, m, s, 1)[0])


_BUILDER = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth
import test_gpu
ctx = sp.Context(0)
fields = ("selected_kmer_number", "selected_superkmer_number", "count_maximal_skmer", "seen_kmers_at_reconstruction", "seen_superkmers_at_reconstruction",
          "seen_max_superkmers_at_reconstruction", "actual_minimizer_number", "read_kmer", "nb_mmer_selected")
comp = bytes.maketrans(b"ACGT", b"TGCA")
n_cases = 0
for (k, m, s, ab) in [(31, 11, 20.0, 1), (31, 11, 3.0, 2), (63, 15, 10.0, 1), (63, 15, 4.0, 3), (21, 11, 1.0, 1), (15, 15, 3.0, 1), (15, 15, 2.0, 2), (33, 13, 50.0, 1), (13, 9, 2.0, 1)]:
    rng = np.random.default_rng(31 * k + ab)
    a, b, c = (synth.random_genome(rng, n).tobytes() for n in (40_000, 20_000, 8_000))
    unit = synth.random_genome(rng, k + 7).tobytes()
    texts = [
        b">one\n" + a + b"\n",                                                                        # plain
        b"".join(b">r%%d\n%%s\n" %% (i, r) for i, r in enumerate([a + b, b[::-1].translate(comp) + c, c + a[:10_000] + c, unit * 257 + unit[:50], unit[::-1].translate(comp) * 256])),
        b">hp\n" + b"A" * 3000 + c[:500] + b"T" * 3000 + b"\n>again\n" + b"A" * 3000 + b"\n",         # homopolymers: one k-mer thousands of times
        b">short\n" + a[: k - 1] + b"\n>exact\n" + a[:k] + b"\n>plus1\n" + a[: k + 1] + b"\n",        # no k-mer / one / two
        b">inv\n" + c + c[::-1].translate(comp) + c + b"\n",                                            # an inverted repeat: both strands of every k-mer
        b">empty\n",
    ]
    for i, text in enumerate(texts):
        got, gst = ctx.sketch_text(text, k, m, s, abundance=ab)
        want, wst = orc.sketch_fasta(text, k, m, s, ab)
        assert got == want, (k, m, s, ab, i, len(got), len(want))
        for f in fields:
            assert gst[f] == wst[f], (k, m, s, ab, i, f, gst[f], wst[f])
        n_cases += 1
print("ok", n_cases)
"""


def test_sketch_builder_on_the_device_equals_the_oracle(tmp_path):
    """spsp_build.hip (SPSP_BUILD=device: handle_superkmer's index, the greedy emission walk and strCompressor on the GPU;
    SubSampler.cpp:243-302, 458-620, utils.cpp:48-68) against the oracle's payload BYTES and every counter of print_stat,
    on the shapes that exercise the walk's order rules: plain genomes, a segment on both strands, a k-mer seen twice and
    three times (-a 1 / 2 / 3), a unit repeated 257 and 256 times (the uint8 count wraps to 1 / 0), homopolymers (one k-mer
    thousands of times, ties of the minimizer inside a k-mer), records of k - 1 / k / k + 1 bases, an inverted repeat, an
    empty file; k <= 32 and k > 32, k == m, -s 1 (every k-mer selected: buckets of thousands of k-mers).  Then the batched
    file pipeline's own tests with the device builder pinned, and the same payloads with SPSP_BUILD=host."""
    for build in ("device", "host"):
        r = subprocess.run([sys.executable, "-c", _BUILDER % (ROOT, os.path.join(ROOT, "tests"))], env=dict(os.environ, SPSP_BUILD=build), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and "ok 54" in r.stdout, (build, r.stdout[-2000:], r.stderr[-3000:])
    code = ("import sys, pathlib, tempfile\nsys.path.insert(0, %r); sys.path.insert(0, %r)\nimport test_gpu\n"
            "d = pathlib.Path(tempfile.mkdtemp(dir=sys.argv[1]))\ntest_gpu.test_sketch_files_pipeline_equals_oracle(d)\n"
            "for (k, m, s, ab) in [(31, 11, 40.0, 2), (63, 15, 15.0, 3)]:\n"
            "    d = pathlib.Path(tempfile.mkdtemp(dir=sys.argv[1]))\n    test_gpu.test_abundance_through_the_batched_pipeline(d, k, m, s, ab)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path)], env=dict(os.environ, SPSP_BUILD="device"), capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_scan_buffer_overflow_retries():
    """the sparse stages are launched with capacity-sized buffers; a call that overflows them
    re-runs with room (hits: from the dense pass, super-k-mers: the write pass only)."""
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import supersampler_amd as sp\n"
        "from supersampler_amd import synth\n"
        "from oracle import oracle_py as orc\n"
        "g = synth.random_genome(np.random.default_rng(8), 400_000)\n"
        "b, o = synth.concat_records([g[:250_000], g[250_000:]])\n"
        "ctx = sp.Context(0)\n"
        "for (k, m, s) in [(31, 11, 50), (21, 11, 1.0)]:\n"
        "    p = sp.make_params(k, m, s)\n"
        "    got = ctx.scan(p, b, o)\n"
        "    want, _ = orc.scan(k, m, p.threshold, b, o)\n"
        "    assert len(got) == len(want) > 50 and all((got[f] == want[f]).all() for f in got.dtype.names)\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    # HITS_CAP also shrinks the per-wave hit lists of the table variants (list overflow -> grown lists);
    # LIST_BUDGET=0 forbids growing them, so the bitmap form of the dense pass takes over
    for env_extra in ({"SPSP_DEBUG_HITS_CAP": "16"}, {"SPSP_DEBUG_OUT_CAP": "8"},
                      {"SPSP_DEBUG_HITS_CAP": "100", "SPSP_DEBUG_OUT_CAP": "3"},
                      {"SPSP_DEBUG_HITS_CAP": "16", "SPSP_DEBUG_LIST_BUDGET": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (env_extra, r.stdout[-2000:], r.stderr[-2000:])


def test_scan_across_the_4gib_seam(ctx):
    """one call over > 2^32 bases: a 1 Mbp record that straddles position 2^32 and another one after it must
    give the same super-k-mers as when scanned alone (queued survivor positions carry only their low 32 bits)."""
    import torch
    k, m, s = 31, 11, 100
    rng = np.random.default_rng(99)
    a = synth.random_genome(rng, 1_000_000)
    b = synth.random_genome(rng, 300_000)
    filler_len = (1 << 32) - 400_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    total = filler_len + len(a) + len(b)
    buf = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for off in range(0, filler_len, step):
        n = min(step, filler_len - off)
        buf[off:off + n] = acgt[torch.randint(0, 4, (n,), device=dev, generator=gen).to(torch.int64)]
    buf[filler_len:filler_len + len(a)] = torch.from_numpy(a).to(dev)
    buf[filler_len + len(a):total] = torch.from_numpy(b).to(dev)
    rec_off = np.array([0, filler_len, filler_len + len(a), total], dtype=np.uint64)
    d_off = torch.from_numpy(rec_off.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    for mode in (sp.SPSP_SCAN_DEFAULT, sp.SPSP_SCAN_LDS_FILTER):
        p = sp.make_params(k, m, s, flags=mode)
        d_out, n_out = ctx.scan_device(p, buf.data_ptr(), total, d_off.data_ptr(), 3)
        got = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
        assert (np.diff(got["rec"].astype(np.int64)) >= 0).all()
        for r, g in ((1, a), (2, b)):
            want, _ = orc.scan(k, m, p.threshold, *synth.concat_records([g]))
            mine = got[got["rec"] == r]
            assert len(mine) == len(want) > 100
            for f in ("minimizer", "start", "len", "rev"):
                assert (mine[f] == want[f]).all(), (mode, r, f)
    del buf


# ----------------------------------------------------------------- full size --
def _np_xxh64(x):
    """vectorised XXH64 of uint64 words, seed 1312 (first principles, not the oracle)."""
    P1, P2, P3, P4, P5 = (np.uint64(v) for v in (11400714785074694791, 14029467366897019727, 1609587929392839161,
                                                  9650029242287828579, 2870177450012600261))

    def rotl(v, b):
        return (v << np.uint64(b)) | (v >> np.uint64(64 - b))
    with np.errstate(over="ignore"):
        h = np.uint64(1312) + P5 + np.uint64(8)
        h = h ^ (rotl(x * P2, 31) * P1)
        h = rotl(h, 27) * P1 + P4
        h ^= h >> np.uint64(33); h *= P2
        h ^= h >> np.uint64(29); h *= P3
        h ^= h >> np.uint64(32)
    return h


def _np_hit_count(genome, m, thr):
    codes = ((genome >> 1) & 3).astype(np.uint64)
    n = len(codes) - m + 1
    f = np.zeros(n, np.uint64)
    r = np.zeros(n, np.uint64)
    for j in range(m):
        f = (f << np.uint64(2)) | codes[j:j + n]
        r |= (codes[j:j + n] ^ np.uint64(2)) << np.uint64(2 * j)
    return int(np.count_nonzero(_np_xxh64(np.minimum(f, r)) <= np.uint64(thr)))


@pytest.mark.parametrize("mode", MODES)
def test_full_size_genome_properties(ctx, mode):
    """one BASELINE-size 5 Mbp genome, k31 m11 s1000: hit count equals a
    first-principles numpy count; super-k-mers are disjoint, in order, inside
    their record, each holds its minimizer with hash <= T; revcomp of the
    genome selects the mirrored k-mers; a duplicate record repeats the stream."""
    import torch  # device memory only
    k, m, s = 31, 11, 1000
    rng = np.random.default_rng(2)
    g = synth.random_genome(rng, 5_000_000)
    p = sp.make_params(k, m, s, flags=mode)
    d = torch.from_numpy(g.copy()).cuda()
    hits = ctx.scan_hits_device(p, d.data_ptr(), d.numel())
    assert hits == _np_hit_count(g, m, p.threshold)
    bases, offs = synth.concat_records([g, g])
    em = ctx.scan(p, bases, offs)
    a, b = em[em["rec"] == 0], em[em["rec"] == 1]
    assert len(a) == len(b) > 100
    for f in ("minimizer", "start", "len", "rev"):
        assert (a[f] == b[f]).all()
    ends = a["start"] + a["len"]
    assert (a["len"] >= k).all() and (a["len"] <= 2 * k - m).all() and (ends <= len(g)).all()
    # disjoint k-mer ranges in genome order
    assert (a["start"][1:] >= a["start"][:-1] + a["len"][:-1] - k + 1).all()
    for e in a[:: max(1, len(a) // 200)]:
        seg = g[int(e["start"]):int(e["start"]) + int(e["len"])].tobytes().decode()
        ms = bf.to_str(int(e["minimizer"]), m)
        assert (ms in bf.rc_str(seg)) if e["rev"] else (ms in seg)
        assert orc.xxh64(int(e["minimizer"])) <= p.threshold
    comp = np.zeros(256, np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    rc = comp[g[::-1]]
    em_rc = ctx.scan(p, *synth.concat_records([rc]))
    sel = lambda e: int((e["len"] - k + 1).sum())
    assert sel(em_rc) == sel(a)
    sk_f = sp.sketch_parse(sp.sketch_build(p, s, bases[:len(g)], offs[:2], a)[0])
    sk_r = sp.sketch_parse(sp.sketch_build(p, s, rc, np.array([0, len(rc)], np.uint64), em_rc)[0])
    assert sk_f.key_set() == sk_r.key_set()  # canonical k-mers do not depend on the strand read
    inter, card = ctx.compare([sk_f, sk_r])
    assert inter[0, 1] == card[0] == card[1]


def test_c5_shape_one_gbp_properties(ctx):
    """BASELINE configs[4] shape (k63 m15 s100, records of 10^6 bp) at 1 Gbp, generated on the GPU: the first 12
    records' super-k-mers equal the oracle's; over the whole gigabase the stream is ordered, disjoint and inside
    its records, and the number of selected k-mers is n/s within 2 %; the dense pass alone finds as many hits as the
    prefix-table and direct-hash variants (three independent dense kernels)."""
    import torch
    k, m, s = 63, 15, 100.0
    rec_len, n_rec = 1_000_000, 1000
    n = rec_len * n_rec
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(55)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    bases = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    for a in range(0, n, 1 << 27):
        b = min(n, a + (1 << 27))
        bases[a:b] = lut[torch.randint(0, 4, (b - a,), device=dev, generator=gen, dtype=torch.int64)]
    off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * rec_len
    torch.cuda.synchronize()
    p = sp.make_params(k, m, s)
    d_out, n_out = ctx.scan_device(p, bases.data_ptr(), n, off.data_ptr(), n_rec)
    sk = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
    rec, start, ln = sk["rec"].astype(np.int64), sk["start"].astype(np.int64), sk["len"].astype(np.int64)
    assert (np.diff(rec) >= 0).all() and (ln >= k).all() and (ln <= 2 * k - m).all() and (start + ln <= rec_len).all()
    same = rec[1:] == rec[:-1]
    assert (start[1:][same] >= start[:-1][same] + ln[:-1][same] - k + 1).all()
    sel = int((ln - k + 1).sum())
    assert abs(sel / ((n - n_rec * (k - 1)) / s) - 1.0) < 0.02
    head = bases[:12 * rec_len].cpu().numpy()
    want, _ = orc.scan(k, m, p.threshold, head, np.arange(13, dtype=np.uint64) * rec_len)
    mine = sk[sk["rec"] < 12]
    _assert_stream_equal(mine, want)
    hits = [ctx.scan_hits_device(sp.make_params(k, m, s, flags=f), bases.data_ptr(), 200_000_000)
            for f in (sp.SPSP_SCAN_BLOOM_FILTER, sp.SPSP_SCAN_LDS_FILTER, sp.SPSP_SCAN_DIRECT_HASH)]
    assert hits[0] == hits[1] == hits[2] > 30_000
    del bases


def test_gpu_sketch_decode_matches_oracle_keys(ctx):
    """N2: bulk decode of sketch payloads on the GPU (blob + text super-k-mers -> canonical keys -> sort -> unique) gives
    the keys the ORACLE's comparator enumerates for each file (orc_sketch_keys: its merge + walk_bucket, Comparator.cpp:
    39-74, 186-260 restated), sketch by sketch: k <= 32 and k > 32, k == m (bare minimizers), empty sketches, a sketch too
    large for the LDS sort (the table in HBM + the merge sort of spsp_bigkeys.hip inside the same call) and duplicate k-mers across super-k-mers."""
    rng = np.random.default_rng(606)
    for (k, m, s, sizes) in [(31, 11, 20, [60_000, 0, 25_000, 300, 700_000]), (63, 15, 10, [40_000, 9_000, 120_000]),
                             (15, 15, 3, [20_000, 5, 8_000]), (21, 9, 4, [30_000, 30_000])]:
        payloads = []
        for i, L in enumerate(sizes):
            g = synth.random_genome(rng, L) if L else np.zeros(0, np.uint8)
            if i == 1 and L:
                g = np.concatenate([g, g[: L // 2], g])          # repeated sequence: the same k-mers in several super-k-mers
            text = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 2) if len(g) else b">empty\n"
            payloads.append(orc.sketch_fasta(text, k, m, s)[0])
        kk, mm, d_mn, d_lo, d_hi, sk_off = ctx.sketch_decode_device(payloads)
        assert (kk, mm) == (k, m)
        total = int(sk_off[-1])
        mn = ctx.to_host(d_mn, total, np.uint32)
        lo = ctx.to_host(d_lo, total, np.uint64)
        hi = ctx.to_host(d_hi, total, np.uint64) if k > 32 else np.zeros(total, np.uint64)
        for i, pl in enumerate(payloads):
            ok, om, w_mn, w_lo, w_hi = orc.sketch_keys(pl)
            a, b = int(sk_off[i]), int(sk_off[i + 1])
            if k == m and b == a and len(w_mn) == 1:
                # a sketch without buckets at k == m: the comparator's merge reads its "first minimizer" past the end of the
                # file and counts the buffer's content as one k-mer (Comparator.cpp:294,316-319).  That key is not in the
                # payload: the decoder returns none and spsp_sketch_chain_host supplies it (include/spsp.h) -- the same one
                ph = sp.sketches_from_payloads([pl])[0]
                assert (ph.minimizer == w_mn).all() and (ph.kmer_lo == w_lo).all()
                continue
            assert (ok, om) == (k, m) and b - a == len(w_mn), (k, m, i, b - a, len(w_mn))
            assert (mn[a:b] == w_mn).all() and (lo[a:b] == w_lo).all(), (k, m, i)
            if k > 32:
                assert (hi[a:b] == w_hi).all(), (k, m, i)
        if k == 31:
            assert int(sk_off[5] - sk_off[4]) > 8192               # the large sketch went through the table in HBM and the merge sort


def _decode_sort_cases(ctx):
    """sketches whose buckets are small (tens of k-mers: every key finds its place by counting inside its bucket), a sketch
    with buckets of thousands of k-mers (a mutated tandem repeat at -s 2: that workgroup takes the bitonic network) and an
    empty one, k <= 32 and k > 32: the decoder's keys = the oracle's, sketch by sketch"""
    rng = np.random.default_rng(707)
    unit = synth.random_genome(rng, 57)
    rep = synth.mutate(rng, np.tile(unit, 700), 0.01)
    out = []
    for (k, m, s) in [(31, 11, 2.0), (63, 15, 3.0)]:
        gs = [synth.random_genome(rng, 30_000), rep, np.concatenate([synth.random_genome(rng, 5_000), rep[:9_000]]), synth.random_genome(rng, 12_000)]
        payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)] + [orc.sketch_fasta(b">e\n", k, m, s)[0]]
        kk, mm, d_mn, d_lo, d_hi, sk_off = ctx.sketch_decode_device(payloads)
        total = int(sk_off[-1])
        mn, lo = ctx.to_host(d_mn, total, np.uint32), ctx.to_host(d_lo, total, np.uint64)
        hi = ctx.to_host(d_hi, total, np.uint64) if k > 32 else np.zeros(total, np.uint64)
        biggest = []
        for i, pl in enumerate(payloads):
            _, _, w_mn, w_lo, w_hi = orc.sketch_keys(pl)
            a, b = int(sk_off[i]), int(sk_off[i + 1])
            assert b - a == len(w_mn) and (mn[a:b] == w_mn).all() and (lo[a:b] == w_lo).all() and (hi[a:b] == w_hi).all(), (k, i)
            biggest.append(int(np.bincount(np.unique(w_mn, return_inverse=True)[1]).max()) if len(w_mn) else 0)
        out.append(biggest)
        assert biggest[1] > 256 and biggest[0] <= 256 and biggest[3] <= 256, biggest   # (both sorts ran, whatever the order the workgroups came in)
    return out


@pytest.mark.gpu
def test_decode_sort_by_bucket_and_by_network_give_the_oracle_keys(ctx):
    """k_decode_sort: counting inside the buckets (payloads as the sketcher writes them: minimizers ascend) and the bitonic
    network (buckets of more than 256 k-mers; every sketch with SPSP_DEBUG_DECODE_SORT=network, in a process of its own)."""
    import subprocess
    import sys
    _decode_sort_cases(ctx)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\nimport supersampler_amd as sp, test_gpu\n"
            "ctx = sp.Context(0)\ntest_gpu._decode_sort_cases(ctx)\ntest_gpu.test_gpu_sketch_decode_matches_oracle_keys(ctx)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, SPSP_DEBUG_DECODE_SORT="network")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


_DECODE_WALKS = r"""
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth
import test_gpu
ctx = sp.Context(0)
test_gpu.test_gpu_sketch_decode_matches_oracle_keys(ctx)            # the oracle's keys, sketch by sketch, through this walk
rng = np.random.default_rng(99)
k, m, s = 31, 11, 12
good = [orc.sketch_fasta(synth.to_fasta(synth.random_genome(rng, int(L)), "g%%d" %% i), k, m, s)[0] for i, L in enumerate(rng.integers(2000, 30000, 40))]
def mutants(pl):
    nl = pl.index(b"\n")
    yield "truncated", pl[: len(pl) * 2 // 3]
    yield "no header", pl.replace(b"\n", b" ")
    yield "header words", b"x" + pl
    yield "other m", pl[:nl].replace(b" 11 ", b" 13 ") + pl[nl:]
    b = bytearray(pl); b[nl + 1 + m : nl + 1 + m + 4] = (2**31).to_bytes(4, "little"); yield "blob past the end", bytes(b)
    b = bytearray(pl); b[nl + 1 + m + 4] = 2; yield "partial blob byte", bytes(b)
    yield "long line", pl[:-2] + b"A" * 300 + b"\n" + b"C" * 5 + b"\n\n\n"
    yield "no last newline", pl[:-1]
    yield "only header", pl[: nl + 1]
    yield "empty", b""
digest = hashlib.sha256()
for name, bad in mutants(good[3]):
    batch = list(good); batch[7] = bad
    try:
        kk, mm, d_mn, d_lo, d_hi, off = ctx.sketch_decode_device(batch)
        tot = int(off[-1])
        digest.update(name.encode() + off.tobytes() + ctx.to_host(d_mn, tot, np.uint32).tobytes() + ctx.to_host(d_lo, tot, np.uint64).tobytes())
    except sp.SpspError as e:
        digest.update(name.encode() + b"error %%d " %% e.code + str(e).encode())
print("digest", digest.hexdigest())
"""


def test_sketch_decode_structure_walk_on_the_device_equals_the_host_walk():
    """k_decode_parse (the payload structure walk on the device, what collections of >= 256 sketch files take) against
    sketch_parse_structure_host: the oracle's keys for every shape of test_gpu_sketch_decode_matches_oracle_keys through BOTH
    walks, and ten malformed payloads (truncated, no header, another m, a blob that runs past the end, a partial blob byte, a
    300-base line, ...) in a batch of 40: the two walks return the same keys or the same error, text included."""
    out = []
    for walk in ("host", "device"):
        r = subprocess.run([sys.executable, "-c", _DECODE_WALKS % (ROOT, os.path.join(ROOT, "tests"))], env=dict(os.environ, SPSP_DEBUG_DECODE_WALK=walk),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "digest" in r.stdout, (walk, r.stdout[-2000:], r.stderr[-3000:])
        out.append(r.stdout.strip().split()[-1])
    assert out[0] == out[1]


def test_cu_partitioned_streams_and_sampled_timing():
    """spsp_stream_create_cus / spsp_set_cu_count (bench.py's step schedule): a scan whose dense pass runs on a stream
    that owns 192 CUs (two workgroups each) with its sparse stages on a 64-CU stream, next to a comparison on
    another 64-CU stream, gives the same stream and matrix as the oracle; bad CU ranges are refused; with
    spsp_timing_sample(4) every fourth dense pass is bracketed."""
    import torch
    k, m, s = 31, 11, 200.0
    gs = synth.family_genomes(5, 6, 300_000, 2, [0.0, 0.01])
    bases, offs = synth.concat_records(gs)
    p = sp.make_params(k, m, s)
    with pytest.raises(sp.SpspError):
        sp.stream_create_cus(0, 250, 64)
    with pytest.raises(sp.SpspError):
        sp.stream_create_cus(0, 0, 0)
    s_d, s_t, s_c = sp.stream_create_cus(0, 64, 192), sp.stream_create_cus(0, 0, 64), sp.stream_create_cus(0, 0, 64)
    scan, cmpc = sp.Context(0, s_d), sp.Context(0, s_c)
    try:
        with pytest.raises(sp.SpspError):
            scan.set_cu_count(10_000, 1)
        scan.set_cu_count(192, 2)
        scan.scan_tail_stream(True, s_t)
        cmpc.set_cu_count(64)
        d_b = torch.from_numpy(bases).cuda()
        d_o = torch.from_numpy(offs.view(np.int64)).cuda()
        payloads = [orc.sketch_fasta(synth.to_fasta(g, "g"), k, m, s)[0] for g in gs]
        sk = [sp.sketch_parse(pl) for pl in payloads]
        sk_off = np.zeros(len(sk) + 1, dtype=np.uint64)
        sk_off[1:] = np.cumsum([len(x) for x in sk])
        d_mn = torch.from_numpy(np.concatenate([x.minimizer for x in sk]).astype(np.uint32).view(np.int32)).cuda()
        d_lo = torch.from_numpy(np.concatenate([x.kmer_lo for x in sk]).astype(np.uint64).view(np.int64)).cuda()
        d_inter = torch.zeros((len(sk), len(sk)), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        scan.timing_enable(True, sp.TIME_DENSE)
        scan.timing_sample(4)
        scan.timing_read()
        want = _oracle_stream(k, m, p.threshold, bases, offs)
        for _ in range(8):
            scan.scan_device_begin(p, d_b.data_ptr(), d_b.numel(), d_o.data_ptr(), len(gs))
            cmpc.compare_device_begin(k, d_mn.data_ptr(), d_lo.data_ptr(), None, sk_off, len(sk), 0, 1, d_inter.data_ptr())
            d_out, n_out = scan.scan_device_end()
            cmpc.compare_end()
            got = scan.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
            _assert_stream_equal(got, want)
        t = scan.timing_read()
        assert t["dense_launches"] in (2, 3) and t["dense_ms"] > 0   # (3: a first call that grew its buffers)
        scan.timing_sample(1)
        scan.timing_enable(False)
        inter, _, _, _ = orc.compare(payloads)
        got_inter = d_inter.cpu().numpy().astype(np.uint32)
        iu = np.triu_indices(len(sk), 1)
        assert (got_inter[iu] == np.asarray(inter, dtype=np.uint32).reshape(len(sk), len(sk))[iu]).all()
    finally:
        scan.close(); cmpc.close()
        for h in (s_d, s_t, s_c):
            sp.stream_destroy(0, h)


def test_config2_full_size_stream_and_matrix_equal_oracle(ctx):
    """BASELINE configs[1] at FULL size (100 genomes x 5 Mbp in 10 families, k31 m11 s1000 -- the workload of bench.py):
    the whole super-k-mer stream of the 5 x 10^8-base batch, field by field, and the 100 x 100 pair matrix of its
    sketches equal the oracle's (the oracle needs ~4 s for the scan of the batch)."""
    import torch
    k, m, s = 31, 11, 1000.0
    genomes = synth.family_genomes(2, 100, 5_000_000, 10, [0.001, 0.01])
    recs = []
    for i, g in enumerate(genomes):          # 1-3 records per genome, as in bench.py
        nr = 1 + i % 3
        cuts = [len(g) * j // nr for j in range(nr + 1)]
        recs += [g[cuts[j]:cuts[j + 1]] for j in range(nr)]
    bases, offs = synth.concat_records(recs)
    p = sp.make_params(k, m, s)
    d_b = torch.from_numpy(bases).cuda()
    d_o = torch.from_numpy(offs.view(np.int64)).cuda()
    d_out, n_out = ctx.scan_device(p, d_b.data_ptr(), d_b.numel(), d_o.data_ptr(), len(recs))
    got = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
    want = _oracle_stream(k, m, p.threshold, bases, offs)
    _assert_stream_equal(got, want)
    assert len(got) > 15_000
    # sketches of the genomes (host builder over the GPU stream) -> all-vs-all on the GPU vs the oracle's comparator
    payloads, r0 = [], 0
    for i, g in enumerate(genomes):
        nr = 1 + i % 3
        sel = got[(got["rec"] >= r0) & (got["rec"] < r0 + nr)].copy()
        sel["rec"] -= r0
        gb, go = synth.concat_records(recs[r0:r0 + nr])
        r0 += nr
        payloads.append(sp.sketch_build(p, s, gb, go, sel)[0])
    inter, card = ctx.compare(sp.sketches_from_payloads(payloads))
    want_inter, want_card, _, _ = orc.compare(payloads)
    assert (inter == want_inter).all() and [int(c) for c in card] == [int(c) for c in want_card]
    assert int((want_inter > 0).sum()) >= 400          # the family structure is there


def test_config3_thousand_sketches_equal_oracle(ctx):
    """BASELINE configs[2] at full count: 1000 sketches in 50 families of 20 (2-8 k keys each, as bench.py's `compare`
    object builds them), all 499 500 pairs and every cardinality against the oracle's comparator."""
    n, k, m, s = 1000, 31, 11, 50.0
    rng = np.random.default_rng(3)
    payloads = []
    p = sp.make_params(k, m, s)
    for f in range(n // 20):
        anc = synth.random_genome(rng, int(rng.integers(100_000, 400_000)))
        for j in range(20):
            b, o = synth.concat_records([synth.mutate(rng, anc, [0.001, 0.01, 0.05][j % 3])])
            payloads.append(sp.sketch_build(p, s, b, o, ctx.scan(p, b, o))[0])
    inter, card = ctx.compare(sp.sketches_from_payloads(payloads))
    want_inter, want_card, _, _ = orc.compare(payloads)
    assert (inter == want_inter).all()
    assert [int(c) for c in card] == [int(c) for c in want_card]
    assert int((want_inter > 0).sum()) >= 9_000


@pytest.mark.parametrize("k,m,s,mode", [(31, 11, 1000, 0), (31, 11, 150, 0), (21, 9, 300, sp.SPSP_SCAN_PAIR_FILTER), (31, 11, 20, 0), (63, 15, 100, 0), (31, 11, 1.0, 0),
                                        (33, 13, 50, 0), (31, 11, 1000, sp.SPSP_SCAN_PAIR_FILTER)])
def test_scan_packed_input_equals_oracle(ctx, k, m, s, mode):
    """SPSP_SCAN_PACKED_INPUT: the same records as 2-bit words (spsp_pack_bases_device) give the oracle's stream -- read
    directly by the pair-table pass (both flavours of its second bit, m = 9 and m >= 10), through an ASCII copy made on the
    device by every other variant; record lengths that are no multiple of 16, inputs shorter than a wave-row, the
    dense-only entry point."""
    import torch
    rng = np.random.default_rng(31 * k + m)
    for lens in ([700_003, 5, 40_000, 1_234_567, 17], [900], [64 * 16 + 3], [130_001]):
        recs = [synth.random_genome(rng, n) for n in lens]
        if len(lens) > 1:
            recs[2] = np.tile(synth.random_genome(rng, 37), 1100)[:40_000]      # a tandem repeat: duplicate minimizers
        bases, offs = synth.concat_records(recs)
        p = sp.make_params(k, m, s, flags=mode | sp.SPSP_SCAN_PACKED_INPUT)
        d_b = torch.from_numpy(bases).cuda()
        d_o = torch.from_numpy(offs.view(np.int64)).cuda()
        torch.cuda.synchronize()
        d_pk = ctx.pack_bases_device(d_b.data_ptr(), d_b.numel())
        d_out, n_out = ctx.scan_device(p, d_pk, d_b.numel(), d_o.data_ptr(), len(recs))
        got = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE) if n_out else np.zeros(0, sp.SUPERKMER_DTYPE)
        _assert_stream_equal(got, _oracle_stream(k, m, p.threshold, bases, offs))
        pa = sp.make_params(k, m, s, flags=mode)
        assert ctx.scan_hits_device(p, d_pk, d_b.numel()) == ctx.scan_hits_device(pa, d_b.data_ptr(), d_b.numel())
    with pytest.raises(sp.SpspError):            # the host-buffer form takes ASCII only
        ctx.scan(sp.make_params(k, m, s, flags=sp.SPSP_SCAN_PACKED_INPUT), bases, offs)


def test_sketch_files_pipeline_equals_oracle(tmp_path):
    """spsp_sketch_files (the library's form of the reference's OpenMP loop over the file of files, SubSampler.cpp:771-793;
    several files share one GPU job): thirteen FASTA files -- plain, gzip, multi-record, Ns and lower case, no k-mers, EMPTY,
    first line not a header (the reference drops it all the same), no newline at the end, CRLF, exactly one ingest tile long --
    on 1, 3 and 16 workers; every output gunzips to the oracle's payload, the statistics (with the count of ALL
    super-k-mers, -v 1) equal the oracle's, the "started" reports come in list order, a file that does not exist fails
    alone; -a 2 takes the one-job-per-file form."""
    import gzip
    k, m, s = 31, 11, 50.0
    rng = np.random.default_rng(77)
    texts = []
    for i in range(13):
        g = synth.random_genome(rng, int(rng.integers(30_000, 400_000)))
        t = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
        if i == 4:
            t = t.replace(b"ACG", b"NnG", 50).replace(b"TT", b"tt", 500)
        if i == 7:
            t = b">tiny\nACGTACGT\n"
        if i == 8:
            t = b""
        if i == 9:
            t = t[1:]                                     # no '>' in front: line 0 is dropped anyway (getLineFasta)
        if i == 10:
            t = t.rstrip(b"\n")
        if i == 11:
            t = t.replace(b"\n", b"\r\n")
        if i == 12:
            t = t[:4095] + b"\n"                          # file + its newline fill one 4 KiB tile exactly
        texts.append(t)
    ins = []
    for i, t in enumerate(texts):
        pth = str(tmp_path / ("in%d.fa" % i)) + (".gz" if i % 2 else "")
        open(pth, "wb").write(gzip.compress(t, 1) if i % 2 else t)
        ins.append(pth)
    want = [orc.sketch_fasta(t, k, m, s) for t in texts]
    fields = ("selected_kmer_number", "selected_superkmer_number", "seen_kmers_at_reconstruction", "actual_minimizer_number", "read_kmer",
              "count_maximal_skmer", "nb_mmer_selected", "total_superkmer_number", "total_kmer_number")
    for threads in (1, 3, 16):
        outs = [str(tmp_path / ("out_t%d_%d.gz" % (threads, i))) for i in range(len(ins))]
        res, times, started = sp.sketch_files(ins, outs, k, m, s, threads=threads, flags=sp.SPSP_SCAN_STATS)
        assert started == list(range(len(ins)))
        assert times["sketch_files"] == len(ins)
        for i, (rc, st, err) in enumerate(res):
            assert rc == 0 and err is None, (threads, i, rc, err)
            assert sp.read_file(outs[i]) == want[i][0], (threads, i)
            for f in fields:
                assert st[f] == want[i][1][f], (threads, i, f, st[f], want[i][1][f])
    # the default sampling (-s 1000): the pair-table dense pass, so the ingest writes 2-bit words and the gather and the
    # count of all super-k-mers read them
    want1k = [orc.sketch_fasta(t, k, m, 1000.0) for t in texts]
    outs = [str(tmp_path / ("out_s1000_%d.gz" % i)) for i in range(len(ins))]
    res, _, _ = sp.sketch_files(ins, outs, k, m, 1000.0, threads=4, flags=sp.SPSP_SCAN_STATS)
    for i, (rc, st, err) in enumerate(res):
        assert rc == 0 and sp.read_file(outs[i]) == want1k[i][0], (i, rc, err)
        for f in fields:
            assert st[f] == want1k[i][1][f], (i, f, st[f], want1k[i][1][f])
    assert sum(w[1]["selected_kmer_number"] for w in want1k) > 500
    bad = ins[:2] + [str(tmp_path / "missing.fa")] + ins[2:4]
    outs = [str(tmp_path / ("out_bad_%d.gz" % i)) for i in range(len(bad))]
    res, _, started = sp.sketch_files(bad, outs, k, m, s, threads=2)
    assert [r[0] == 0 for r in res] == [True, True, False, True, True] and "missing.fa" in res[2][2]
    assert sp.read_file(outs[4]) == want[3][0]
    # -a 2: k-mers are counted per file on the device (since round 5 in one pass per batch: test_abundance_through_the_batched_pipeline)
    dup = texts[0] + texts[0].replace(b">g0", b">again")
    open(str(tmp_path / "dup.fa"), "wb").write(dup)
    outs = [str(tmp_path / "dup.gz"), str(tmp_path / "single.gz")]
    res, _, _ = sp.sketch_files([str(tmp_path / "dup.fa"), ins[0]], outs, k, m, s, abundance=2, threads=2)
    assert all(r[0] == 0 for r in res)
    assert sp.read_file(outs[0]) == orc.sketch_fasta(dup, k, m, s, 2)[0] and sp.read_file(outs[1]) == orc.sketch_fasta(texts[0], k, m, s, 2)[0]


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 40.0, 2), (63, 15, 15.0, 3), (31, 11, 40.0, 1)])
def test_abundance_through_the_batched_pipeline(tmp_path, k, m, s, ab):
    """-a > 1 in spsp_sketch_files: the k-mer occurrences of a whole batch of files are counted in ONE device pass with the
    file as part of the key (the reference's index is per file: one Subsampler per file, SubSampler.cpp:787; its count is a
    uint8, :283-300, and a k-mer is used only from `count >= abundance` on, :587,608).  100 files: every file holds segments
    seen once, twice and three times (some on the other strand), a unit repeated 257 times (the count wraps to 1), and
    shares its whole sequence with its neighbour file -- k-mers of ANOTHER file must not count.  Payload bytes and statistics
    equal the oracle's, with 1, 4 and 16 workers, and the one-job-per-file form (SPSP_DEBUG_ABUND_PER_FILE) writes the same."""
    rng = np.random.default_rng(500 + k + ab)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    base = [synth.random_genome(rng, 30_000) for _ in range(50)]
    texts = []
    for i in range(100):
        g = base[i // 2]                                              # files 2j and 2j + 1 hold the same genome ...
        once, twice = g[:10_000], g[10_000:20_000]
        rc_twice = np.array([comp[c] for c in twice[::-1].tolist()], dtype=np.uint8)
        thrice = g[20_000:30_000]
        unit = synth.random_genome(rng, k + 5)
        recs = [once, twice, rc_twice if i % 2 else twice, thrice, thrice, thrice, np.tile(unit, 257)]
        if i % 2:
            recs = recs[::-1]                                         # ... in another record order
        texts.append(b"".join(synth.to_fasta(r, "f%d_r%d" % (i, j)) for j, r in enumerate(recs)))
    ins = []
    for i, t in enumerate(texts):
        pth = str(tmp_path / ("a%03d.fa" % i))
        open(pth, "wb").write(t)
        ins.append(pth)
    want = [orc.sketch_fasta(t, k, m, s, ab) for t in texts]
    assert want[0][0] != orc.sketch_fasta(texts[0], k, m, s, 1)[0] or ab == 1
    fields = ("selected_kmer_number", "selected_superkmer_number", "seen_kmers_at_reconstruction", "actual_minimizer_number", "read_kmer",
              "count_maximal_skmer", "nb_mmer_selected")
    for threads in (1, 4, 16):
        outs = [str(tmp_path / ("o%d_%03d.gz" % (threads, i))) for i in range(len(ins))]
        res, times, _ = sp.sketch_files(ins, outs, k, m, s, abundance=ab, threads=threads)
        for i, (rc, st, err) in enumerate(res):
            assert rc == 0 and err is None, (threads, i, rc, err)
            assert sp.read_file(outs[i]) == want[i][0], (threads, i)
            for f in fields:
                assert st[f] == want[i][1][f], (threads, i, f, st[f], want[i][1][f])
    if ab > 1:
        code = ("import sys\nsys.path.insert(0, %r)\nimport supersampler_amd as sp\n"
                "ins = [l.strip() for l in open(sys.argv[1])]\nouts = [x + '.perfile.gz' for x in ins]\n"
                "res, _, _ = sp.sketch_files(ins, outs, %d, %d, %r, abundance=%d, threads=4)\nassert all(r[0] == 0 for r in res)\nprint('ok')\n") % (ROOT, k, m, s, ab)
        (tmp_path / "fof.txt").write_text("\n".join(ins[:12]) + "\n")
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "fof.txt")], env=dict(os.environ, SPSP_DEBUG_ABUND_PER_FILE="1"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
        for i in range(12):
            assert sp.read_file(ins[i] + ".perfile.gz") == want[i][0], i


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 20.0, 1), (31, 11, 1000.0, 1), (63, 15, 10.0, 1), (21, 11, 3.0, 2), (33, 13, 4.0, 3), (11, 11, 4.0, 3)])
def test_sketch_keys_on_device_equal_the_file_path(ctx, k, m, s, ab):
    """spsp_sketch_keys_device: from ONE scan over the records of several genomes to the comparator's keys, without
    sketch files -- equal, genome by genome, to what the ORACLE's comparator enumerates (orc_sketch_keys) from the sketch
    the oracle writes for that genome (handle_superkmer's uint8 counts and the -a rule included): random and mutated
    genomes, a genome followed by its reverse complement (both orientations of every k-mer), a unit repeated 257 and 256
    times (the count wraps), a record too short for a k-mer; ASCII and 2-bit input; a genome too large for the
    per-genome LDS forms goes through the table in HBM (and the merge sort) inside the same call."""
    import torch
    rng = np.random.default_rng(1000 + k)
    L = min(900_000, int((1500 if k > 32 else 3000) * s))     # a genome's selected k-mer occurrences stay inside the per-genome sort
    a = synth.random_genome(rng, L)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    rc = np.array([comp[c] for c in a[::-1].tolist()], dtype=np.uint8)
    unit = synth.random_genome(rng, k + 2)
    genomes = [[a[: L // 2], a[L // 2:]], [synth.mutate(rng, a, 0.02)], [a[: L // 2], rc],                 # (every k-mer of the first half twice, once per strand)
               [np.tile(unit, 257), synth.random_genome(rng, 500)], [np.tile(unit[::-1].copy(), 256)], [synth.random_genome(rng, k - 1)],
               [synth.random_genome(rng, L // 2)], [a[: L // 4]] * ab + [a[L // 4: L // 3]] * max(1, ab - 1)]     # seen ab and ab - 1 times
    recs, first_rec, texts = [], [0], []
    for i, g in enumerate(genomes):
        recs += g
        first_rec.append(len(recs))
        texts.append(b"".join(synth.to_fasta(r, "g%d_%d" % (i, j)) for j, r in enumerate(g)))
    bases, off = synth.concat_records(recs)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    want = [orc.sketch_keys(orc.sketch_fasta(t, k, m, s, ab)[0]) for t in texts]
    assert sum(len(w[2]) for w in want) > (500 if ab == 1 else 100)
    for packed in (False, True):
        p = sp.make_params(k, m, s, abundance=ab, flags=sp.SPSP_SCAN_PACKED_INPUT if packed else 0)
        src = ctx.pack_bases_device(d_b.data_ptr(), len(bases)) if packed else d_b.data_ptr()
        d_sk, n_sk = ctx.scan_device(p, src, len(bases), d_o.data_ptr(), len(recs))
        d_mn, d_lo, d_hi, sk_off = ctx.sketch_keys_device(p, src, len(bases), d_o.data_ptr(), d_sk, n_sk, first_rec)
        total = int(sk_off[-1])
        mn, lo = ctx.to_host(d_mn, total, np.uint32), ctx.to_host(d_lo, total, np.uint64)
        hi = ctx.to_host(d_hi, total, np.uint64) if k > 32 else np.zeros(total, np.uint64)
        for g, (_, _, w_mn, w_lo, w_hi) in enumerate(want):
            x, y = int(sk_off[g]), int(sk_off[g + 1])
            if k == m and y == x and len(w_mn) == 1:
                continue                                  # the merge's phantom key of a sketch without buckets: in no file (spsp_sketch_chain_host)
            assert y - x == len(w_mn), (packed, g, y - x, len(w_mn))
            assert (mn[x:y] == w_mn).all() and (lo[x:y] == w_lo).all() and (hi[x:y] == w_hi).all(), (packed, g)
    # the keys feed the comparison as they are: pair counts equal the oracle's comparison of the sketch files
    w_inter, w_card, _, _ = orc.compare([orc.sketch_fasta(t, k, m, s, ab)[0] for t in texts])
    w_inter = w_inter.astype(np.int64)
    if k == m:       # (a sketch without buckets gets the merge's phantom key there, which no file holds: leave its pairs out)
        empty = [g for g in range(len(genomes)) if sk_off[g + 1] == sk_off[g]]
        w_inter[empty, :] = 0
        w_inter[:, empty] = 0
    d_inter = torch.zeros((len(genomes), len(genomes)), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.compare_device(k, d_mn, d_lo, d_hi, sk_off, len(genomes), 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    assert (np.triu(d_inter.cpu().numpy(), 1) == np.triu(w_inter, 1)).all()
    # SPSP_KEYS_UNORDERED: the same key SETS from an LDS table per genome instead of a sort; a comparison that has been told
    # so takes them, one that has not refuses them (its order check is its duplicate check)
    # (k == m: one k-mer per super-k-mer, thousands of super-k-mers per genome -- more than the unordered form stages in LDS:
    # those genomes go through the table in HBM, same keys)
    d_mn, d_lo, d_hi, sk_off2 = ctx.sketch_keys_device(p, src, len(bases), d_o.data_ptr(), d_sk, n_sk, first_rec, unordered=True)
    assert (sk_off2 == sk_off).all()
    total = int(sk_off2[-1])
    mn, lo = ctx.to_host(d_mn, total, np.uint32), ctx.to_host(d_lo, total, np.uint64)
    hi = ctx.to_host(d_hi, total, np.uint64) if k > 32 else np.zeros(total, np.uint64)
    for g, (_, _, w_mn, w_lo, w_hi) in enumerate(want):
        x, y = int(sk_off2[g]), int(sk_off2[g + 1])
        if k == m and y == x and len(w_mn) == 1:
            continue
        assert sorted(zip(mn[x:y].tolist(), hi[x:y].tolist(), lo[x:y].tolist())) == list(zip(w_mn.tolist(), w_hi.tolist(), w_lo.tolist())), g
    d_inter.zero_()
    torch.cuda.synchronize()
    if k != m:
        with pytest.raises(sp.SpspError):
            ctx.compare_device(k, d_mn, d_lo, d_hi, sk_off2, len(genomes), 0, 1, d_inter.data_ptr())
    ctx.compare_keys_unordered(True)
    d_inter.zero_()
    torch.cuda.synchronize()
    ctx.compare_device(k, d_mn, d_lo, d_hi, sk_off2, len(genomes), 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    ctx.compare_keys_unordered(False)
    assert (np.triu(d_inter.cpu().numpy(), 1) == np.triu(w_inter, 1)).all()
    if k == 31 and s == 20.0:
        big = synth.random_genome(rng, 400_000)
        bb, bo = synth.concat_records([big])
        d_b2 = torch.from_numpy(np.concatenate([bb, np.zeros(64, np.uint8)])).cuda()
        d_o2 = torch.from_numpy(bo.view(np.int64)).cuda()
        torch.cuda.synchronize()
        p = sp.make_params(k, m, s)
        d_sk, n_sk = ctx.scan_device(p, d_b2.data_ptr(), len(bb), d_o2.data_ptr(), 1)
        # a genome beyond the per-genome LDS forms (20 000 selected k-mers): the table in HBM inside the call (+ the merge
        # sort in the sorted form), the keys the file would give
        for un in (False, True):
            d_mn, d_lo, _, koff = ctx.sketch_keys_device(p, d_b2.data_ptr(), len(bb), d_o2.data_ptr(), d_sk, n_sk, [0, 1], unordered=un)
            assert ctx.sketch_keys_big_genomes() == 1
            _, _, w_mn, w_lo, _ = orc.sketch_keys(orc.sketch_fasta(synth.to_fasta(big, "big"), k, m, s)[0])
            assert int(koff[1]) == len(w_mn) > 8192
            g_mn, g_lo = ctx.to_host(d_mn, len(w_mn), np.uint32), ctx.to_host(d_lo, len(w_lo), np.uint64)
            if un:
                o = np.lexsort((g_lo, g_mn))
                g_mn, g_lo = g_mn[o], g_lo[o]
            assert (g_mn == w_mn).all() and (g_lo == w_lo).all(), un


def test_sketch_keys_api_misuse(ctx):
    """a second _begin on a context with a job pending and an _end without a job are refused; spsp_measure_hbm_device
    returns rates a streaming kernel can have on this part."""
    import torch
    rng = np.random.default_rng(6)
    g = synth.random_genome(rng, 40_000)
    bases, off = synth.concat_records([g])
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    p = sp.make_params(31, 11, 20.0)
    d_sk, n_sk = ctx.scan_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), 1)
    with pytest.raises(sp.SpspError):
        ctx._keys_n = 1
        ctx.sketch_keys_device_end()
    ctx.sketch_keys_device_begin(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, [0, 1])
    with pytest.raises(sp.SpspError):
        ctx.sketch_keys_device_begin(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, [0, 1])
    _, _, _, koff = ctx.sketch_keys_device_end()
    assert int(koff[1]) > 1000
    with pytest.raises(sp.SpspError):
        ctx.sketch_keys_device_begin(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, [1, 0])     # decreasing record ranges
    rates = ctx.measure_hbm(256 << 20, 3)
    assert 1000 < rates["copy_GBps"] < 8000 and 1000 < rates["read_GBps"] < 8000 and rates["bytes"] == 256 << 20


def _keys_want(texts, k, m, s, ab=1):
    return [orc.sketch_keys(orc.sketch_fasta(t, k, m, s, ab)[0]) for t in texts]


def _check_keys(ctx, got, want, k, sorted_form, tag):
    d_mn, d_lo, d_hi, sk_off = got
    total = int(sk_off[-1])
    mn, lo = ctx.to_host(d_mn, total, np.uint32), ctx.to_host(d_lo, total, np.uint64)
    hi = ctx.to_host(d_hi, total, np.uint64) if k > 32 else np.zeros(total, np.uint64)
    for g, (_, _, w_mn, w_lo, w_hi) in enumerate(want):
        x, y = int(sk_off[g]), int(sk_off[g + 1])
        assert y - x == len(w_mn), (tag, g, y - x, len(w_mn))
        a, b, c = mn[x:y], lo[x:y], hi[x:y]
        if not sorted_form:
            o = np.lexsort((b, c, a))
            a, b, c = a[o], b[o], c[o]
        assert (a == w_mn).all() and (b == w_lo).all() and (c == w_hi).all(), (tag, g)


@pytest.mark.parametrize("k,m,s,ab", [(31, 11, 25.0, 1), (63, 15, 12.0, 1), (31, 11, 30.0, 2), (21, 11, 8.0, 1)])
def test_sketch_keys_of_any_size_stay_on_the_device(ctx, k, m, s, ab):
    """The reference's k-mer index is unbounded (SubSampler.h:62, SubSampler.cpp:274-300).  Genomes beyond the per-genome
    LDS forms (8192 / 6144 k-mers) and genomes inside them in ONE call, both forms, ASCII and 2-bit input: the large ones go
    through the table in HBM (spsp_bigkeys.hip), the others keep their workgroup; every genome equals the oracle's
    comparator walk over the oracle's sketch.  A large genome that repeats a unit 257 / 256 times (the uint8 count wraps),
    one that holds a segment on both strands, and with -a 2 segments seen once and twice."""
    import torch
    rng = np.random.default_rng(77 + k)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    big = synth.random_genome(rng, int(14_000 * s))                      # ~14 000 selected k-mers
    mid = synth.random_genome(rng, int(7_000 * s)) if k <= 32 else synth.random_genome(rng, int(5_000 * s))   # between the two LDS limits
    small = synth.random_genome(rng, int(1_500 * s))
    rcbig = np.array([comp[c] for c in big[: len(big) // 3][::-1].tolist()], dtype=np.uint8)
    unit = synth.random_genome(rng, k + 3)
    genomes = [[small], [big[: len(big) // 2], big[len(big) // 2:]], [synth.mutate(rng, small, 0.02)], [mid],
               [big[: len(big) // 3], rcbig, np.tile(unit, 257), big[len(big) // 3:]],       # both strands + a wrapped count, large
               [np.tile(unit[::-1].copy(), 256), synth.mutate(rng, big, 0.01)],
               [synth.random_genome(rng, k - 1)], [small[: len(small) // 2]] * ab,
               [big] + [big[: len(big) // 2]] * (ab - 1)]
    recs, first_rec, texts = [], [0], []
    for i, g in enumerate(genomes):
        recs += g
        first_rec.append(len(recs))
        texts.append(b"".join(synth.to_fasta(r, "g%d_%d" % (i, j)) for j, r in enumerate(g)))
    bases, off = synth.concat_records(recs)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    want = _keys_want(texts, k, m, s, ab)
    n_big = 4                                                            # genomes 1, 4, 5 and 8 hold ~14 000 k-mer occurrences or more
    if ab == 1:
        assert sum(1 for w in want if len(w[2]) > 8192) >= n_big and sum(1 for w in want if 0 < len(w[2]) < 2048) >= 2
    assert sum(len(w[2]) for w in want) > 5000
    for packed in (False, True):
        p = sp.make_params(k, m, s, abundance=ab, flags=sp.SPSP_SCAN_PACKED_INPUT if packed else 0)
        src = ctx.pack_bases_device(d_b.data_ptr(), len(bases)) if packed else d_b.data_ptr()
        d_sk, n_sk = ctx.scan_device(p, src, len(bases), d_o.data_ptr(), len(recs))
        for un in (False, True):
            got = ctx.sketch_keys_device(p, src, len(bases), d_o.data_ptr(), d_sk, n_sk, first_rec, unordered=un)
            assert ctx.sketch_keys_big_genomes() >= n_big
            _check_keys(ctx, got, want, k, not un, (packed, un))
    # the keys feed the comparison as they are
    w_inter, _, _, _ = orc.compare([orc.sketch_fasta(t, k, m, s, ab)[0] for t in texts])
    d_inter = torch.zeros((len(genomes), len(genomes)), dtype=torch.int32, device="cuda")
    ctx.compare_keys_unordered(True)
    torch.cuda.synchronize()
    ctx.compare_device(k, got[0], got[1], got[2], got[3], len(genomes), 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    ctx.compare_keys_unordered(False)
    assert (np.triu(d_inter.cpu().numpy(), 1) == np.triu(w_inter.astype(np.int64), 1)).all()


def test_sketch_keys_survive_the_next_scan(ctx):
    """The pipelining pattern of bench.py's closed step with genomes beyond the LDS table: the key extraction of scan t is
    queued, scan t + 1 is queued on the SAME scan context behind spsp_scan_output_wait and rewrites the super-k-mer buffer,
    and only then is the extraction collected.  Everything the extraction reads of the caller's buffers it reads in the
    work _begin queued (there is no host path behind _end any more), so the keys are those of scan t."""
    import torch
    k, m, s = 31, 11, 20.0
    rng = np.random.default_rng(123)
    sets = []
    for r in range(2):
        gs = [[synth.random_genome(rng, 300_000)], [synth.random_genome(rng, 40_000)], [synth.random_genome(rng, 200_000), synth.random_genome(rng, 150_000)]]
        recs = [x for g in gs for x in g]
        first = np.cumsum([0] + [len(g) for g in gs]).astype(np.uint32)
        bases, off = synth.concat_records(recs)
        texts = [b"".join(synth.to_fasta(x, "r%d" % j) for j, x in enumerate(g)) for g in gs]
        sets.append((torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), len(bases), len(recs), first, texts))
    torch.cuda.synchronize()
    p = sp.make_params(k, m, s)
    reader = sp.Context(0)
    try:
        for un in (True, False):
            d_b, d_o, nb, nr, first, texts = sets[0]
            d_sk, n_sk = ctx.scan_device(p, d_b.data_ptr(), nb, d_o.data_ptr(), nr)
            reader.wait_stream(ctx)
            reader.sketch_keys_device_begin(p, d_b.data_ptr(), nb, d_o.data_ptr(), d_sk, n_sk, first, unordered=un)
            ctx.scan_output_wait(reader)
            d_b2, d_o2, nb2, nr2, _, _ = sets[1]
            ctx.scan_device(p, d_b2.data_ptr(), nb2, d_o2.data_ptr(), nr2)        # rewrites the buffer the extraction read
            got = reader.sketch_keys_device_end()
            assert reader.sketch_keys_big_genomes() == 2
            _check_keys(reader, got, _keys_want(texts, k, m, s), k, not un, un)
    finally:
        reader.close()


def test_key_extraction_keeps_its_gate_words_across_a_large_comparison():
    """the key extraction's two device words (big-genome gate and count) were words 12 / 13 of the comparison's
    flag block, where k_parts_group leaves its list statistics after any comparison beyond the small form -- the next
    extraction on the same context then reported them as `big genomes` and ran the table kernels ungated.  They are words of
    the extraction's own now: extraction -> comparison of 2 000 sketches -> extraction on ONE context, big_genomes == 0
    both times and the same keys."""
    import torch
    dev = torch.device("cuda", 0)
    k, m, s = 31, 11, 200.0
    rng = np.random.default_rng(9)
    gs = [synth.random_genome(rng, 150_000) for _ in range(6)]
    bases, off = synth.concat_records(gs)
    texts = [synth.to_fasta(g, "g%d" % i) for i, g in enumerate(gs)]
    want = _keys_want(texts, k, m, s)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    n = 2000
    D = synth.direct_family_sketches(n, fam_size=20, seed=5, device=dev, skm_range=(20, 40))
    dense = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    p = sp.make_params(k, m, s)
    c = sp.Context(0)
    try:
        for rnd in range(2):
            for un in (True, False):
                d_sk, n_sk = c.scan_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(gs))
                got = c.sketch_keys_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, list(range(len(gs) + 1)), unordered=un)
                assert c.sketch_keys_big_genomes() == 0, (rnd, un)
                _check_keys(c, got, want, k, not un, (rnd, un))
            c.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, dense.data_ptr())
            torch.cuda.synchronize()
            assert int((dense != 0).sum().item()) >= (n // 20) * 100
    finally:
        c.close()


def test_configs4_shape_record_set_keys_equal_the_oracle(ctx):
    """BASELINE configs[4]'s shape (k63 m15 s100) at the size of one streamed segment: 60 records of 2 Mbp as ONE genome,
    ~1.2 x 10^6 selected k-mers -- two hundred times the LDS table -- from the scan to the comparator's keys on the device,
    equal to the oracle's sketch read back by the oracle's comparator (orc_sketch_keys), sorted form element by element."""
    import torch
    k, m, s = 63, 15, 100.0
    rng = np.random.default_rng(5)
    recs = [synth.random_genome(rng, 2_000_000) for _ in range(60)]
    recs[7][:500_000] = recs[3][:500_000]                                # a repeated stretch: k-mers seen twice
    bases, off = synth.concat_records(recs)
    text = b"".join(synth.to_fasta(r, "r%d" % j) for j, r in enumerate(recs))
    want = _keys_want([text], k, m, s)
    assert len(want[0][2]) > 1_000_000
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    p = sp.make_params(k, m, s)
    d_sk, n_sk = ctx.scan_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(recs))
    for un in (False, True):
        got = ctx.sketch_keys_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, [0, len(recs)], unordered=un)
        assert ctx.sketch_keys_big_genomes() == 1
        _check_keys(ctx, got, want, k, not un, un)


def test_config1_single_genome_sketch_and_self_compare(ctx, tmp_path):
    """BASELINE configs[0]: ONE 4 641 652-bp genome (the E. coli K-12 length stand-in of SURVEY.md 8d "C1", seed 1, three
    records), k31 m11 s1000 by default flags: `sub_sampler -i` writes the oracle's payload bytes and prints print_stat's
    numbers (SubSampler.cpp:306-510, 633-665, 749-760); `comparator -f` on that ONE sketch takes the single-file route of
    the merge -- every bucket goes through skip_bucket (Comparator.cpp:60-63) -- and prints the two 1 x 1 matrices
    (:362-460); spsp_compare with n = 1, host and device forms."""
    import torch
    k, m, s = 31, 11, 1000
    g = synth.random_genome(np.random.default_rng(1), 4_641_652)
    text = synth.to_fasta(g, "ecoli_stand_in", n_records=3)
    (tmp_path / "ecoli.fa").write_bytes(text)
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-i", "ecoli.fa"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    want, st = orc.sketch_fasta(text, k, m, float(np.float32(s)))
    got = gzip.open(tmp_path / "subsampled_ecoli.gz", "rb").read()
    assert got == want
    assert 3_000 < st["selected_kmer_number"] < 7_000
    out = r.stdout
    assert out.startswith(" I use k=31 m=11 s=1000\nMaximal super kmer are of length 51 or 21 kmers\n")
    assert not (tmp_path / "subsampled_ecoli.txt").exists()            # -i writes no list of sketches (SubSampler.cpp:749-760)
    at = 0
    for line in ["I have seen %s kmers and I selected %s kmers" % (_fmt_commas(st["total_kmer_number"]), _fmt_commas(st["selected_kmer_number"])),
                 "After removing duplicate kmers, I selected %s kmers" % _fmt_commas(st["seen_kmers_at_reconstruction"]),
                 "I have seen %s superkmers and I selected %s superkmers" % (_fmt_commas(st["total_superkmer_number"]), _fmt_commas(st["selected_superkmer_number"])),
                 "After reconstruction and filtering with abundance, I have selected %s superkmers" % _fmt_commas(st["seen_superkmers_at_reconstruction"]),
                 "Minimizer number: %s Skmer/minimizer:  " % _fmt_commas(st["actual_minimizer_number"]),
                 "Number of maximal skmer was:       %s" % _fmt_commas(st["count_maximal_skmer"]),
                 "Actual number of maximal skmer is: %s" % _fmt_commas(st["seen_max_superkmers_at_reconstruction"])]:
        nxt = out.find(line, at)
        assert nxt >= 0, (line, out[at:at + 400])
        at = nxt + len(line)
    (tmp_path / "one.txt").write_text("subsampled_ecoli.gz\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "one.txt", "-o", "self"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split("\n")[1] == "I found 1 documents"
    inter, card, _, _ = orc.compare([want])
    assert inter.shape == (1, 1) and int(card[0]) == len(orc.sketch_keys(want)[2]) > 3_000
    for jac, fn in ((True, "self_jaccard.csv.gz"), (False, "self_containment.csv.gz")):
        assert gzip.open(tmp_path / fn, "rb").read() == orc.csv(jac, ["subsampled_ecoli.gz"], inter, card, None, 6, 0.0)
    # n = 1 through the C-ABI: host form, device form, and the device keys of the one genome
    sk = sp.sketches_from_payloads([got])
    i1, c1 = ctx.compare(sk)
    assert i1.shape == (1, 1) and int(i1[0, 0]) == 0 and int(c1[0]) == int(card[0])
    bases, off = sp.clean_fasta(text)
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    p = sp.make_params(k, m, float(s))
    d_sk, n_sk = ctx.scan_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), len(off) - 1)
    d_mn, d_lo, d_hi, koff = ctx.sketch_keys_device(p, d_b.data_ptr(), len(bases), d_o.data_ptr(), d_sk, n_sk, [0, len(off) - 1])
    assert int(koff[1]) == int(card[0])
    assert (ctx.to_host(d_mn, int(koff[1]), np.uint32) == sk[0].minimizer).all() and (ctx.to_host(d_lo, int(koff[1]), np.uint64) == sk[0].kmer_lo).all()
    d_inter = torch.full((1, 1), 7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.compare_device(k, d_mn, d_lo, None, koff, 1, 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    assert int(d_inter[0, 0].item()) == 7                                # a 1 x 1 problem has no cell (i, j > i): nothing is written


def test_config3_true_shape_sketched_on_device_equals_oracle(ctx):
    """BASELINE configs[2] at its TRUE shape, reduced count (bench.py's `compare` leg runs all 1000): 60 genomes of
    L ~ U[2, 8] Mbp in 3 families of 20 at mu 0.001 / 0.01 / 0.05, generated on the device, k31 m11 s1000 -- ~100 minimizer
    buckets shared by every genome, ~L / 1000 keys per sketch, a third of the genomes beyond the per-genome LDS table.
    Sketched by the HIP path (one scan + one key extraction per batch, no file, no host path), all pairs and every
    cardinality against the oracle's sketcher + comparator over the same bases."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    n = 60
    dev = torch.device("cuda", 0)
    want = card = None
    for un in (False, True):
        T = bench.config3_true_shape(ctx, dev, n, want is None, seed=3, unordered=un)
        # unordered form: genomes above ~6.7 Mbp hold more than its 6144 k-mer places -> the table in HBM; the sorted form's 8192 are not reached
        assert (T["big_genomes"] >= 5) if un else (T["big_genomes"] == 0)
        assert 2_000 * n < int(T["sk_off"][-1]) < 8_000 * n
        d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.compare_keys_unordered(un)
        ctx.compare_device(T["k"], T["d_min"].data_ptr(), T["d_lo"].data_ptr(), None, T["sk_off"], n, 0, 1, d_inter.data_ptr())
        torch.cuda.synchronize()
        ctx.compare_keys_unordered(False)
        if want is None:
            payloads = T["payloads"]
            want, card, _, _ = orc.compare(payloads)
        assert (np.triu(d_inter.cpu().numpy().astype(np.uint32), 1) == np.triu(want, 1)).all(), un
        assert (card.astype(np.int64) == T["cnt"]).all(), un
        assert int(np.count_nonzero(np.triu(want, 1))) >= 3 * 190        # every pair inside a family shares k-mers
        # the same keys, genome by genome, as the oracle's comparator enumerates them from the oracle's sketch files
        mn = T["d_min"].cpu().numpy().view(np.uint32)
        lo = T["d_lo"].cpu().numpy().view(np.uint64)
        for g in (0, 7, 23, 41, 59):
            _, _, w_mn, w_lo, _ = orc.sketch_keys(payloads[g])
            a, b = int(T["sk_off"][g]), int(T["sk_off"][g + 1])
            x, y = mn[a:b], lo[a:b]
            if un:
                o = np.lexsort((y, x))
                x, y = x[o], y[o]
            assert (x == w_mn).all() and (y == w_lo).all(), (un, g)


def test_sketch_files_dealt_over_a_device_list(tmp_path):
    """spsp_sketch_files_multi: the batches of the file pipeline dealt over several devices (here device 0 named twice and
    three times: one pipeline slot per entry and round) write the oracle's payload bytes and report its statistics, with
    -a 1 (batched pipeline) and -a 2 (one GPU job per file, workers dealt over the devices); bin/sub_sampler with
    SPSP_DEVICES="0,0" takes the same route."""
    k, m, s = 31, 11, 60.0
    gs = synth.family_genomes(77, 10, 120_000, 2, [0.0, 0.01, 0.03])
    ins, texts = [], []
    for i, g in enumerate(gs):
        t = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
        if i == 4:
            t = t + t.replace(b">g4", b">again")                 # every k-mer twice: -a 2 keeps them
        pth = tmp_path / ("d%d.fa" % i)
        pth.write_bytes(t)
        ins.append(str(pth)); texts.append(t)
    for ab in (1, 2):
        want = [orc.sketch_fasta(t, k, m, s, ab) for t in texts]
        for devs in ([0, 0], [0, 0, 0]):
            outs = [str(tmp_path / ("o%d_%d_%d.gz" % (ab, len(devs), i))) for i in range(len(ins))]
            res, _, _ = sp.sketch_files(ins, outs, k, m, s, abundance=ab, threads=4, devices=devs)
            for i, (rc, st, err) in enumerate(res):
                assert rc == 0 and sp.read_file(outs[i]) == want[i][0], (ab, devs, i, rc, err)
                assert st["selected_kmer_number"] == want[i][1]["selected_kmer_number"]
    (tmp_path / "fof.txt").write_text("\n".join(ins) + "\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "fof.txt", "-k", str(k), "-m", str(m), "-s", str(int(s)), "-t", "3", "-p", "two_"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(os.environ, SPSP_DEVICES="0,0"))
    assert r.returncode == 0, r.stdout + r.stderr
    want = [orc.sketch_fasta(t, k, m, float(np.float32(s)))[0] for t in texts]
    for i in range(len(ins)):
        assert gzip.open(tmp_path / ("two_d%d.gz" % i), "rb").read() == want[i], i


_ONE_SPECIES = r"""
import numpy as np, sys, os
sys.path.insert(0, %r)
import supersampler_amd as sp
import torch
use_hi = bool(int(sys.argv[1])); n = int(sys.argv[2])
rng = np.random.default_rng(n + use_hi)
U = 2600                                            # ancestral keys: held by 90 %% / 50 %% / 10 %% / 1 %% of the sketches
share = np.repeat([0.9, 0.5, 0.1, 0.01], U // 4)
B = rng.random((n, U)) < share[None, :]
B[n // 3] = False                                   # an empty sketch
B[n // 2] = B[n // 2 - 1]                           # two identical ones
anc = np.stack([rng.integers(0, 2**22, U), rng.integers(0, 2**62, U), rng.integers(0, 2**62, U) if use_hi else np.zeros(U, np.int64)], 1)
sketches, sizes = [], []
for i in range(n):
    own = np.stack([rng.integers(0, 2**22, 150), rng.integers(0, 2**62, 150), rng.integers(0, 2**62, 150) if use_hi else np.zeros(150, np.int64)], 1)
    keys = np.concatenate([anc[B[i]], own]) if B[i].any() else anc[:0]
    order = np.lexsort((keys[:, 1], keys[:, 2], keys[:, 0]))          # by (minimizer, kmer_hi, kmer_lo)
    keys = keys[order]
    sizes.append(len(keys))
    sketches.append(sp.Sketch(63 if use_hi else 31, 11, keys[:, 0].astype(np.uint32), keys[:, 1].astype(np.uint64), keys[:, 2].astype(np.uint64)))
want = np.triu(B.astype(np.int64) @ B.astype(np.int64).T, 1)
ctx = sp.Context(0)
for rep in range(2):                                # (the second call is queued with the spill from the start)
    inter, card = ctx.compare(sketches)
    assert [int(c) for c in card] == sizes
    assert (np.triu(inter.astype(np.int64), 1) == want).all(), rep
# a collection that does not overflow, then the species again
small = sketches[: n // 8]
inter, _ = ctx.compare(small)
assert (np.triu(inter.astype(np.int64), 1) == want[: n // 8, : n // 8]).all()
inter, _ = ctx.compare(sketches)
assert (np.triu(inter.astype(np.int64), 1) == want).all()
# query mode (rows of the first sketches only) and the result as sparse cells
q = 37
inter, _ = ctx.compare(sketches, n_query=q)
assert (np.triu(inter.astype(np.int64), 1)[:q] == want[:q]).all()
dev = torch.device("cuda", 0)
mn = torch.from_numpy(np.concatenate([s.minimizer for s in sketches]).astype(np.int32)).to(dev)
lo = torch.from_numpy(np.concatenate([s.kmer_lo for s in sketches]).view(np.int64)).to(dev)
hi = torch.from_numpy(np.concatenate([s.kmer_hi for s in sketches]).view(np.int64)).to(dev) if use_hi else None
sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(sizes)
scratch = torch.zeros((n, n), dtype=torch.int32, device=dev)
cells = torch.zeros(n * n, dtype=torch.int64, device=dev)
cnt = ctx.compare_cells_device(63 if use_hi else 31, mn.data_ptr(), lo.data_ptr(), hi.data_ptr() if use_hi else None, sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
c = cells[:cnt].cpu().numpy()
got = np.zeros((n, n), np.int64)
got[(c >> 48) & 0xffff, (c >> 32) & 0xffff] = c & 0xffffffff
assert cnt == np.count_nonzero(want) and (got == want).all()
# rows dealt i %% 3 (a rank of a row-partitioned comparison)
d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
ctx.compare_device(63 if use_hi else 31, mn.data_ptr(), lo.data_ptr(), hi.data_ptr() if use_hi else None, sk_off, n, 1, 3, d_inter.data_ptr())
got = d_inter.cpu().numpy().astype(np.int64)
assert (got[1::3] == want[1::3]).all() and not got[0::3].any() and not got[2::3].any()
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("use_hi,n,env", [(0, 400, {}), (1, 300, {}), (0, 400, {"SPSP_DEBUG_SPILL_BITS": "0"}), (0, 300, {"SPSP_DEBUG_SPILL_BITS": "2"}),
                                          (1, 260, {"SPSP_DEBUG_SPILL": "0"}), (0, 330, {"SPSP_DEBUG_FILTER": "0"}),
                                          (0, 440, {"SPSP_DEBUG_KEY_CLASSES": "2"})])
def test_compare_one_species_collection_spills_overflowed_parts(use_hi, n, env):
    """Hundreds of sketches that share most of their keys (one species sequenced many times): a key arrives in its part
    with all its holders, parts overflow whatever their number, and their records are grouped in HBM instead (k_spill_*):
    holder lists for the keys of few sketches, columns of the bit matrix + popcounts for the keys of many
    (SPSP_DEBUG_SPILL_BITS: lists only / columns only; SPSP_DEBUG_SPILL=0: the global dictionary of before).  Every pair
    against B B^T of the incidence matrix, all-vs-all, query mode, as cells and for a rank's rows; k <= 32 and k > 32."""
    r = subprocess.run([sys.executable, "-c", _ONE_SPECIES % ROOT, str(use_hi), str(n)], env=dict(os.environ, SPSP_DEBUG_SPILL_TRACE="1", **env),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert ("spsp spill:" in r.stderr) == (env.get("SPSP_DEBUG_SPILL") != "0"), r.stderr[-2000:]
