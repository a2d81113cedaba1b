"""GPU tests of the key-partitioned multi-GPU exchange (include/spsp.h "multi-GPU exchange"), with the
all-to-all played in-process: `world` ranks are partitioned one after another on the one test GPU, their
slots transposed on the host exactly as the collective would, every rank's partial matrix computed by
spsp_compare_slots_device and the partials summed.  The sum must equal plain set algebra (and the
single-GPU spsp_compare), whatever the number of ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

MAGIC = 0x4C535053


@pytest.fixture(scope="module")
def ctx():
    c = sp.Context(0)
    yield c
    c.close()


def make_sets(rng, n, use_hi, universe_size=4000):
    universe = [(int(rng.integers(0, 2**22)), int(rng.integers(0, 2**62)) if use_hi else 0, int(rng.integers(0, 2**62)))
                for _ in range(universe_size)]
    universe += [(u[0] ^ 1, u[1], u[2]) for u in universe[:50]]      # same k-mer, another minimizer: another key
    sets = []
    for i in range(n):
        if i % 13 == 5:
            sets.append(set())
        elif i % 7 == 3:
            sets.append(set(sets[i - 1]))
        else:
            pick = rng.random(len(universe)) < rng.choice([0.02, 0.1, 0.4])
            sets.append({universe[j] for j in np.nonzero(pick)[0]})
    return sets


def rank_arrays(sets):
    """sorted key arrays of one rank's sketches, concatenated, + offsets"""
    keys = [sorted(st) for st in sets]
    off = np.zeros(len(sets) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in keys])
    flat = [x for ks in keys for x in ks]
    mn = np.array([x[0] for x in flat], dtype=np.uint32)
    hi = np.array([x[1] for x in flat], dtype=np.uint64)
    lo = np.array([x[2] for x in flat], dtype=np.uint64)
    return mn, lo, hi, off


def partition(ctx, k, mn, lo, hi, off, n_local, world, cap):
    dev = torch.device("cuda", 0)
    d_mn = torch.from_numpy(mn.view(np.int32)).to(dev)
    d_lo = torch.from_numpy(lo.view(np.int64)).to(dev)
    d_hi = torch.from_numpy(hi.view(np.int64)).to(dev)
    sb = sp.slot_bytes(n_local, cap, k)
    d_slots = torch.full((world * sb,), 0xEE, dtype=torch.uint8, device=dev)    # stale bytes must not matter
    torch.cuda.synchronize()
    ctx.partition_keys_device(k, d_mn.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr() if k > 32 else None, off, n_local,
                              world, cap, d_slots.data_ptr())
    out = ctx.to_host(d_slots.data_ptr(), world * sb, np.uint8)                 # drains the context's stream
    return out.reshape(world, sb)


def parse_slot(slot, n_local, cap, k):
    """-> (per-sketch key lists in stored order, n_keys in the header)"""
    hdr = slot[:16].view(np.uint32)
    assert hdr[0] == MAGIC and hdr[1] == n_local and hdr[3] == (3 if k > 32 else 2)
    cnt = slot[16:16 + 4 * n_local].view(np.uint32)
    words = int(hdr[3])
    rec_off = 16 + ((n_local + 1) & ~1) * 4
    rec = slot[rec_off:rec_off + cap * words * 8].view(np.uint64).reshape(cap, words)
    out, at = [], 0
    for j in range(n_local):
        c = int(cnt[j])
        rows = rec[at:min(at + c, cap)]
        assert all(int(r[-1]) >> 32 == j for r in rows)
        out.append([(int(r[-1]) & 0xFFFFFFFF, int(r[1]) if words == 3 else 0, int(r[0])) for r in rows])
        at += c
    assert at == int(hdr[2])
    return out, int(hdr[2])


def exchange_and_compare(ctx, k, per_rank_sets, cap):
    world, n_local = len(per_rank_sets), len(per_rank_sets[0])
    sent = [partition(ctx, k, *rank_arrays(sets), n_local, world, cap) for sets in per_rank_sets]
    dev = torch.device("cuda", 0)
    n_total = world * n_local
    total = np.zeros((n_total, n_total), dtype=np.int64)
    for d in range(world):
        recv = np.concatenate([sent[s][d] for s in range(world)])               # what the all-to-all delivers to rank d
        d_recv = torch.from_numpy(recv).to(dev)
        d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.compare_slots_device(k, d_recv.data_ptr(), world, n_local, cap, d_inter.data_ptr())
        torch.cuda.synchronize()
        total += d_inter.cpu().numpy()
    return sent, total


@pytest.mark.parametrize("world,n_local,use_hi", [(1, 5, False), (2, 7, False), (3, 11, True), (8, 9, False), (4, 40, True)])
def test_partitioned_compare_equals_set_algebra(ctx, world, n_local, use_hi):
    rng = np.random.default_rng(world * 100 + n_local)
    k = 63 if use_hi else 31
    sets = make_sets(rng, world * n_local, use_hi)
    per_rank = [sets[r * n_local:(r + 1) * n_local] for r in range(world)]
    biggest = max(sum(len(s) for s in pr) for pr in per_rank)
    cap = biggest if world == 1 else int(biggest / world * 1.3) + 64
    sent, total = exchange_and_compare(ctx, k, per_rank, cap)
    n_total = world * n_local
    for i in range(n_total):
        for j in range(n_total):
            want = len(sets[i] & sets[j]) if j > i else 0
            assert total[i, j] == want, (i, j, int(total[i, j]), want)
    # wire format: every key of every sketch sits in exactly one slot, grouped by sketch, in sorted order,
    # and equal keys of different ranks land in the same destination
    where = {}
    for r in range(world):
        got = [[] for _ in range(n_local)]
        for d in range(world):
            per_sketch, n_keys = parse_slot(sent[r][d], n_local, cap, k)
            assert n_keys <= cap
            for j, keys in enumerate(per_sketch):
                assert keys == sorted(keys)
                got[j] += keys
                for key in keys:
                    assert where.setdefault(key, d) == d
        for j in range(n_local):
            assert sorted(got[j]) == sorted(per_rank[r][j])
            assert len(got[j]) == len(per_rank[r][j])
    if world > 1:   # the hash spreads the keys: no slot far from its share
        sizes = [parse_slot(sent[r][d], n_local, cap, k)[1] for r in range(world) for d in range(world)]
        assert max(sizes) <= cap


def test_partitioned_compare_matches_single_gpu_compare(ctx):
    """the same sketches through spsp_compare (one GPU) and through 4 simulated ranks"""
    rng = np.random.default_rng(5)
    world, n_local = 4, 6
    sets = make_sets(rng, world * n_local, False, universe_size=20000)
    sketches = []
    for st in sets:
        keys = sorted(st)
        sketches.append(sp.Sketch(31, 11, np.array([x[0] for x in keys], np.uint32), np.array([x[2] for x in keys], np.uint64),
                                  np.array([x[1] for x in keys], np.uint64)))
    inter, _ = ctx.compare(sketches)
    per_rank = [sets[r * n_local:(r + 1) * n_local] for r in range(world)]
    cap = int(max(sum(len(s) for s in pr) for pr in per_rank) / world * 1.3) + 64
    _, total = exchange_and_compare(ctx, 31, per_rank, cap)
    assert (total == inter.astype(np.int64)).all() and total.sum() > 0


def test_slot_overflow_is_reported_and_a_larger_cap_succeeds(ctx):
    rng = np.random.default_rng(9)
    sets = make_sets(rng, 8, False)
    per_rank = [sets[:4], sets[4:]]
    with pytest.raises(sp.SpspError) as e:
        exchange_and_compare(ctx, 31, per_rank, 100)      # far too small: slots keep 100 keys and say so
    assert e.value.code == sp.ERR_OVERFLOW
    _, total = exchange_and_compare(ctx, 31, per_rank, 4000)
    assert all(total[i, j] == len(sets[i] & sets[j]) for i in range(8) for j in range(i + 1, 8))


def test_malformed_slots_are_rejected(ctx):
    rng = np.random.default_rng(10)
    sets = make_sets(rng, 6, False)
    n_local, world, cap, k = 3, 2, 3000, 31
    sent = [partition(ctx, k, *rank_arrays(sets[r * 3:(r + 1) * 3]), n_local, world, cap) for r in range(2)]
    dev = torch.device("cuda", 0)
    d_inter = torch.zeros((6, 6), dtype=torch.int32, device=dev)

    def run(recv):
        d = torch.from_numpy(recv).to(dev)
        torch.cuda.synchronize()
        ctx.compare_slots_device(k, d.data_ptr(), world, n_local, cap, d_inter.data_ptr())

    good = np.concatenate([sent[0][0], sent[1][0]])
    run(good)
    for corrupt in ("magic", "n", "count", "sketch"):
        bad = good.copy()
        h = bad[:16].view(np.uint32)
        if corrupt == "magic":
            h[0] ^= 1
        elif corrupt == "n":
            h[1] += 1
        elif corrupt == "count":
            bad[16:20].view(np.uint32)[0] += 1            # counts no longer add up to n_keys
        else:
            rec_off = 16 + ((n_local + 1) & ~1) * 4
            bad[rec_off + 8:rec_off + 16].view(np.uint64)[0] |= np.uint64(77) << np.uint64(32)   # sketch id out of range
        with pytest.raises(sp.SpspError):
            run(bad)
    with pytest.raises(sp.SpspError):
        ctx.compare_slots_device(k, 0, world, n_local, cap, d_inter.data_ptr())
    with pytest.raises(sp.SpspError):
        ctx.partition_keys_device(k, 1, 1, None, np.zeros(4, np.uint64), 3, 65, cap, 8)     # too many destinations
    bad_off = np.array([0, 50, 20, 60], np.uint64)                                          # decreasing offsets never reach a kernel
    with pytest.raises(sp.SpspError):
        ctx.partition_keys_device(k, 8, 8, None, bad_off, 3, 2, cap, 8)
    with pytest.raises(sp.SpspError):
        ctx.compare_device(k, 8, 8, None, bad_off, 3, 0, 1, d_inter.data_ptr())


def test_partitioned_compare_collision_retry():
    """fingerprints cut to 8 bits on the first attempt: the slot form detects the collisions and rebuilds"""
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import supersampler_amd as sp\n"
        "import test_exchange as tx\n"
        "ctx = sp.Context(0)\n"
        "sets = tx.make_sets(np.random.default_rng(3), 6, False)\n"
        "_, total = tx.exchange_and_compare(ctx, 31, [sets[:3], sets[3:]], 4000)\n"
        "assert all(total[i, j] == len(sets[i] & sets[j]) for i in range(6) for j in range(i + 1, 6))\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_FP_BITS="8"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


# ----------------------------------------------------------------- begin / end (pipelined) forms of the ABI
def test_begin_end_forms_on_two_streams_match_the_blocking_calls(ctx):
    """scan on one context/stream and the comparison on another, both queued before either is waited for
    (bench.py's pipelined step): results identical to the blocking calls, step after step."""
    from oracle import oracle_py as orc
    from supersampler_amd import synth
    dev = torch.device("cuda", 0)
    k, m, s = 31, 11, 50
    p = sp.make_params(k, m, s)
    genomes = synth.family_genomes(11, 6, 300_000, 2, [0.0, 0.01, 0.02])
    bases, rec_off = synth.concat_records(genomes)
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(rec_off.view(np.int64)).to(dev)
    sets = make_sets(np.random.default_rng(21), 70, False)
    mn, lo, hi, off = rank_arrays(sets)
    d_mn, d_lo = torch.from_numpy(mn.view(np.int32)).to(dev), torch.from_numpy(lo.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    want_stream = ctx.scan(p, bases, rec_off)
    want_inter = np.array([[len(sets[i] & sets[j]) if j > i else 0 for j in range(70)] for i in range(70)])
    a, b = sp.Context(0), sp.Context(0)
    try:
        for step in range(3):
            d_inter = torch.zeros((70, 70), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            a.scan_device_begin(p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), len(genomes))
            b.wait_dense(a)
            b.compare_device_begin(k, d_mn.data_ptr(), d_lo.data_ptr(), None, off, 70, 0, 1, d_inter.data_ptr())
            with pytest.raises(sp.SpspError):      # one pending job of each kind per context
                a.scan_device_begin(p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), len(genomes))
            with pytest.raises(sp.SpspError):
                b.compare_device_begin(k, d_mn.data_ptr(), d_lo.data_ptr(), None, off, 70, 0, 1, d_inter.data_ptr())
            d_out, n_out = a.scan_device_end()
            b.compare_end()
            got = a.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
            assert n_out == len(want_stream) and got.tobytes() == np.asarray(want_stream).tobytes()
            assert (d_inter.cpu().numpy() == want_inter).all()
        # stream ordering helpers and the timing mask: only the requested regions are bracketed
        a.timing_enable(True, sp.TIME_DENSE)
        b.timing_enable(True, sp.TIME_COMPARE | sp.TIME_ACCUMULATE)
        a.timing_read(); b.timing_read()
        a.wait_stream(b)
        a.scan_device_begin(p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), len(genomes))
        b.wait_stream(a)
        b.compare_device_begin(k, d_mn.data_ptr(), d_lo.data_ptr(), None, off, 70, 0, 1, d_inter.data_ptr())
        assert a.scan_device_end()[1] == len(want_stream)
        b.compare_end()
        ta, tb = a.timing_read(), b.timing_read()
        assert ta["dense_launches"] == 1 and ta["dense_ms"] > 0 and ta["scan_calls"] == 0
        assert tb["compare_calls"] == 1 and tb["accumulate_launches"] <= 1 and tb["dense_launches"] == 0   # (70 sketches: the small-problem form has no row-sum kernel)
        a.timing_enable(False); b.timing_enable(False)
        assert (d_inter.cpu().numpy() == want_inter).all()
        with pytest.raises(sp.SpspError):
            a.scan_device_end()                    # nothing pending
        with pytest.raises(sp.SpspError):
            b.compare_end()
        # empty inputs keep begin/end paired
        a.scan_device_begin(p, d_bases.data_ptr(), 5, d_off.data_ptr(), 1)
        assert a.scan_device_end() == (None, 0)
        b.compare_device_begin(k, d_mn.data_ptr(), d_lo.data_ptr(), None, np.zeros(3, np.uint64), 2, 0, 1, d_inter.data_ptr())
        b.compare_end()
    finally:
        a.close()
        b.close()
    em_oracle, _ = orc.scan(k, m, orc.threshold(k, m, s), bases, rec_off)
    assert len(em_oracle) == len(want_stream)


def test_capped_colour_matrix_runs_key_class_passes():
    """SPSP_DEBUG_MATRIX_BUDGET caps the colour matrix at a few KiB, so the comparison is built and summed in many
    key-class passes (the path taken for tens of thousands of sketches): flat form, query mode, exchange slots --
    results unchanged; with cut fingerprints on top, a colliding class restarts the whole sum."""
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import supersampler_amd as sp\n"
        "import test_exchange as tx\n"
        "ctx = sp.Context(0)\n"
        "sets = tx.make_sets(np.random.default_rng(8), 70, True)\n"
        "sk = []\n"
        "for st in sets:\n"
        "    keys = sorted(st)\n"
        "    sk.append(sp.Sketch(63, 15, np.array([x[0] for x in keys], np.uint32), np.array([x[2] for x in keys], np.uint64),\n"
        "                        np.array([x[1] for x in keys], np.uint64)))\n"
        "inter, card = ctx.compare(sk)\n"
        "assert all(inter[i, j] == (len(sets[i] & sets[j]) if j > i else 0) for i in range(70) for j in range(70))\n"
        "assert [int(c) for c in card] == [len(s) for s in sets]\n"
        "interq, _ = ctx.compare(sk, n_query=9)\n"
        "assert all(interq[i, j] == (len(sets[i] & sets[j]) if (j > i and i < 9) else 0) for i in range(70) for j in range(70))\n"
        "per_rank = [sets[r * 14:(r + 1) * 14] for r in range(5)]\n"
        "cap = int(max(sum(len(s) for s in pr) for pr in per_rank) / 5 * 1.3) + 64\n"
        "_, total = tx.exchange_and_compare(ctx, 63, per_rank, cap)\n"
        "assert all(total[i, j] == (len(sets[i] & sets[j]) if j > i else 0) for i in range(70) for j in range(70))\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for extra in ({}, {"SPSP_DEBUG_FP_BITS": "9"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_MATRIX_BUDGET="6000", **extra),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-2000:], r.stderr[-3000:])


def test_sparse_list_form_many_sketches(ctx):
    """4 300 sketches (67 colour words): partition form (default) -- with SPSP_DEBUG_PARTS=0 the global-dictionary
    comparison takes the sketch-list form by itself (test_many_sketches_global_dictionary).  Checked against a
    plain Python count of shared keys over all 9.2 million pairs."""
    rng = np.random.default_rng(123)
    n = 4300
    universe = rng.integers(1, 2**62, size=60_000, dtype=np.int64)
    fam = [rng.choice(universe, size=120, replace=False) for _ in range(200)]          # 200 families share their keys
    sketches, sets = [], []
    for i in range(n):
        base = fam[i % 200]
        keep = base[rng.random(len(base)) < 0.8]
        extra = rng.choice(universe, size=int(rng.integers(0, 15)), replace=False)
        keys = np.unique(np.concatenate([keep, extra])).astype(np.uint64) if i % 97 else np.zeros(0, np.uint64)
        sets.append(keys)
        sketches.append(sp.Sketch(31, 11, np.full(len(keys), 5, np.uint32), keys, np.zeros(len(keys), np.uint64)))
    inter, card = ctx.compare(sketches)
    assert [int(c) for c in card] == [len(s) for s in sets]
    holders = {}
    for i, keys in enumerate(sets):
        for key in keys.tolist():
            holders.setdefault(key, []).append(i)
    want = np.zeros((n, n), dtype=np.uint32)
    for hs in holders.values():
        if len(hs) > 1:
            a = np.array(hs)
            ii, jj = np.triu_indices(len(a), 1)
            np.add.at(want, (a[ii], a[jj]), 1)
    assert want.sum() > 1_000_000
    assert (inter == want).all()


def test_sparse_list_form_forced_on_small_inputs():
    """SPSP_DEBUG_SPARSE=1: the sketch-list form on the inputs of the dense tests (k > 32 keys, empty and duplicate
    sketches, 1-3 colour words) and, with cut fingerprints, through its collision retry."""
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import supersampler_amd as sp\n"
        "import test_exchange as tx\n"
        "ctx = sp.Context(0)\n"
        "for n, use_hi in ((3, False), (70, True), (150, False)):\n"
        "    sets = tx.make_sets(np.random.default_rng(n), n, use_hi)\n"
        "    sk = []\n"
        "    for st in sets:\n"
        "        keys = sorted(st)\n"
        "        sk.append(sp.Sketch(63 if use_hi else 31, 15, np.array([x[0] for x in keys], np.uint32),\n"
        "                            np.array([x[2] for x in keys], np.uint64), np.array([x[1] for x in keys], np.uint64)))\n"
        "    inter, card = ctx.compare(sk)\n"
        "    assert all(inter[i, j] == (len(sets[i] & sets[j]) if j > i else 0) for i in range(n) for j in range(n)), n\n"
        "    assert [int(c) for c in card] == [len(s) for s in sets]\n"
        "sk[4].kmer_lo[[1, 2]] = sk[4].kmer_lo[[2, 1]]; sk[4].minimizer[[1, 2]] = sk[4].minimizer[[2, 1]]\n"
        "try:\n"
        "    ctx.compare(sk); raise SystemExit('unsorted keys were accepted')\n"
        "except sp.SpspError:\n"
        "    pass\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for extra in ({}, {"SPSP_DEBUG_FP_BITS": "9"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_SPARSE="1", **extra),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-2000:], r.stderr[-3000:])


def test_comparison_forms_agree_on_random_inputs():
    """the same twelve random comparison problems (2-260 sketches, k <= 32 and k > 32, empty / duplicate / nested
    sketches, query mode) through the default form, the sketch-list form and the capped colour matrix: identical
    matrices, and equal to Python set algebra."""
    code = (
        "import sys, hashlib\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import supersampler_amd as sp\n"
        "ctx = sp.Context(0)\n"
        "rng = np.random.default_rng(2024)\n"
        "digest = hashlib.sha256()\n"
        "for case in range(12):\n"
        "    n = int(rng.choice([2, 3, 17, 64, 65, 130, 260]))\n"
        "    use_hi = bool(rng.integers(0, 2))\n"
        "    uni = [(int(rng.integers(0, 2**22)), int(rng.integers(0, 2**62)) if use_hi else 0, int(rng.integers(0, 2**62)))\n"
        "           for _ in range(int(rng.integers(50, 3000)))]\n"
        "    sets = []\n"
        "    for i in range(n):\n"
        "        r = rng.random()\n"
        "        if r < 0.1: st = set()\n"
        "        elif r < 0.2 and i: st = set(sets[int(rng.integers(0, i))])\n"
        "        elif r < 0.3 and i: st = set(list(sets[int(rng.integers(0, i))])[::2])\n"
        "        else: st = {uni[j] for j in np.nonzero(rng.random(len(uni)) < rng.choice([0.01, 0.1, 0.5]))[0]}\n"
        "        sets.append(st)\n"
        "    sk = []\n"
        "    for st in sets:\n"
        "        keys = sorted(st)\n"
        "        sk.append(sp.Sketch(63 if use_hi else 31, 11, np.array([x[0] for x in keys], np.uint32),\n"
        "                            np.array([x[2] for x in keys], np.uint64), np.array([x[1] for x in keys], np.uint64)))\n"
        "    nq = n if case %% 3 else max(1, n // 3)\n"
        "    inter, card = ctx.compare(sk, n_query=nq)\n"
        "    want = np.zeros((n, n), np.uint32)\n"
        "    for i in range(nq):\n"
        "        for j in range(i + 1, n):\n"
        "            want[i, j] = len(sets[i] & sets[j])\n"
        "    assert (inter == want).all(), case\n"
        "    assert [int(c) for c in card] == [len(s) for s in sets]\n"
        "    digest.update(inter.tobytes())\n"
        "print('digest', digest.hexdigest())\n") % (ROOT, os.path.join(ROOT, "tests"))
    digests = []
    # {} = the partition form (LDS dictionary per key class), the others force the global-dictionary forms
    # SPSP_DEBUG_SMALL=0: the general partition form also for <= 128 sketches; SPSP_DEBUG_KEY_CLASSES: the keys in that many
    # hash classes, one pass through the parts each (what inputs beyond 9 x 10^7 keys get), also with many tiny parts
    for env in ({}, {"SPSP_DEBUG_SPARSE": "1"}, {"SPSP_DEBUG_MATRIX_BUDGET": "9000"}, {"SPSP_DEBUG_SPARSE": "0"}, {"SPSP_DEBUG_PARTS": "0"},
                {"SPSP_DEBUG_SMALL": "0"}, {"SPSP_DEBUG_KEY_CLASSES": "3"}, {"SPSP_DEBUG_KEY_CLASSES": "2", "SPSP_DEBUG_PART_MEAN": "40"},
                # SPSP_DEBUG_TILES: the scatter over tiles (32 sketches x the same share of each) switched off / with tiles of ~40
                # entries / of ~9 000 (three rounds per tile) in blocks of 8 and 64 sketches
                {"SPSP_DEBUG_TILES": "0", "SPSP_DEBUG_SMALL": "0"}, {"SPSP_DEBUG_TILES": "40", "SPSP_DEBUG_SMALL": "0"},
                {"SPSP_DEBUG_TILES": "9000", "SPSP_DEBUG_SMALL": "0", "SPSP_DEBUG_TILE_SK": "64"},
                {"SPSP_DEBUG_TILES": "40", "SPSP_DEBUG_KEY_CLASSES": "3", "SPSP_DEBUG_PART_MEAN": "40", "SPSP_DEBUG_TILE_SK": "8"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "digest" in r.stdout, (env, r.stdout[-2000:], r.stderr[-3000:])
        digests.append(r.stdout.strip().split()[-1])
    assert len(set(digests)) == 1


def test_many_sketches_global_dictionary():
    """the same 4 300 sketches with the partition form switched off: sketch-list form over the global dictionary"""
    code = ("import sys\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import supersampler_amd as sp, test_exchange as tx\n"
            "ctx = sp.Context(0)\ntx.test_sparse_list_form_many_sketches(ctx)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_PARTS="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_partition_form_overflow_falls_back(ctx):
    """600 sketches holding the SAME 4 000 keys (+ a few private ones): every part receives its keys 600 at a
    time, the fixed-capacity parts overflow and their records are grouped in HBM (spill; before round 4 the
    global-dictionary form took over -- tests/test_gpu.py::test_compare_one_species_collection_spills_overflowed_parts
    runs that way too, SPSP_DEBUG_SPILL=0)."""
    rng = np.random.default_rng(77)
    n = 600
    shared = np.unique(rng.integers(1, 2**62, size=4000, dtype=np.int64)).astype(np.uint64)
    sketches, sets = [], []
    for i in range(n):
        extra = np.unique(rng.integers(1, 2**62, size=i % 5, dtype=np.int64)).astype(np.uint64)
        keys = np.unique(np.concatenate([shared, extra]))
        sets.append(keys)
        sketches.append(sp.Sketch(31, 11, np.full(len(keys), 9, np.uint32), keys, np.zeros(len(keys), np.uint64)))
    inter, card = ctx.compare(sketches)
    assert [int(c) for c in card] == [len(s) for s in sets]
    want = np.triu(np.full((n, n), len(shared), dtype=np.uint32), 1)
    assert (inter == want).all()


def test_eight_simulated_ranks_c4_shaped(ctx):
    """BASELINE configs[3] shape on one GPU: 2 048 sketches (families of 16, ~600 keys each), the all-gather form's
    per-rank call (rows i % 8 == rank over ALL keys) for each of 8 ranks; the strips together equal the single
    call and an inverted-index count."""
    import dist_worker as dw
    n, world = 2048, 8
    sets = dw.c4_shaped_sets(n, 600, seed=44)
    want = dw.expected_inter(sets)
    dev = torch.device("cuda", 0)
    cnt = np.array([len(x) for x in sets])
    sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)
    d_lo = torch.from_numpy(np.concatenate(sets).view(np.int64)).to(dev)
    d_min = torch.full((int(cnt.sum()),), 7, dtype=torch.int32, device=dev)
    merged = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for r in range(world):
        part = torch.full((n, n), -1, dtype=torch.int32, device=dev)     # cells outside the owned rows must stay untouched
        torch.cuda.synchronize()
        ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, r, world, part.data_ptr())
        torch.cuda.synchronize()
        p = part.cpu().numpy()
        own = np.zeros((n, n), bool)
        own[r::world] = np.triu(np.ones((n, n), bool), 1)[r::world]
        assert (p[~own] == -1).all()
        merged[r::world] = torch.triu(part, 1)[r::world]
    assert (merged.cpu().numpy() == want).all() and int(want.sum()) > 1_000_000


def test_eight_simulated_ranks_own_blocks_of_rows(ctx):
    """The same 2 048 sketches with the rows dealt in BLOCKS (rank r owns rows [256 r, 256 r + 256): the sketches it
    scanned itself -- dist.KeyExchange's default): row_first = first row, row_stride = 1, n_query = end of the block.
    Each call builds its dictionary from its own block's keys and the keys of later sketches that pass the Bloom
    filter; cells outside the block stay untouched, the blocks together equal the inverted-index count.  A second
    round on the same context starts from the part size the first one measured."""
    import dist_worker as dw
    n, world = 2048, 8
    sets = dw.c4_shaped_sets(n, 600, seed=46)
    want = dw.expected_inter(sets)
    dev = torch.device("cuda", 0)
    cnt = np.array([len(x) for x in sets])
    sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)
    d_lo = torch.from_numpy(np.concatenate(sets).view(np.int64)).to(dev)
    d_min = torch.full((int(cnt.sum()),), 7, dtype=torch.int32, device=dev)
    per = n // world
    for rnd in range(2):
        merged = torch.zeros((n, n), dtype=torch.int32, device=dev)
        for r in (range(world) if rnd == 0 else reversed(range(world))):
            part = torch.full((n, n), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, r * per, 1, part.data_ptr(), n_query=(r + 1) * per)
            torch.cuda.synchronize()
            p = part.cpu().numpy()
            own = np.zeros((n, n), bool)
            own[r * per:(r + 1) * per] = np.triu(np.ones((n, n), bool), 1)[r * per:(r + 1) * per]
            assert (p[~own] == -1).all()
            merged[r * per:(r + 1) * per] = torch.triu(part, 1)[r * per:(r + 1) * per]
        assert (merged.cpu().numpy() == want).all() and int(want.sum()) > 1_000_000
    # a block that is not aligned to anything, with a stride: rows 301, 304, 307, ... below 1500
    part = torch.full((n, n), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 301, 3, part.data_ptr(), n_query=1500)
    torch.cuda.synchronize()
    p = part.cpu().numpy()
    rows = np.arange(301, 1500, 3)
    own = np.zeros((n, n), bool)
    own[rows] = np.triu(np.ones((n, n), bool), 1)[rows]
    assert (p[~own] == -1).all() and (p[own] == want[own]).all()
    # misuse: a zero stride is refused; a first row behind the last sketch owns nothing and writes nothing
    part.fill_(-1)
    torch.cuda.synchronize()
    with pytest.raises(sp.SpspError):
        ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 0, 0, part.data_ptr())
    ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, n + 5, 1, part.data_ptr())
    ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 7, 3, part.data_ptr(), n_query=7)
    torch.cuda.synchronize()
    assert (part.cpu().numpy() == -1).all()


def test_filtered_rows_equal_unfiltered_rows_and_grow_their_parts():
    """Row-partitioned calls with the filter forced off (SPSP_DEBUG_FILTER=0) and on give the same strips; with keys
    that EVERY sketch shares almost everything passes the filter, the first attempt's parts (sized for the owned keys)
    overflow and the call repeats itself with the count it measured."""
    code = ("import sys, os\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch\nimport supersampler_amd as sp\n"
            "rng = np.random.default_rng(5)\nn = 64\n"
            "shared = np.unique(rng.integers(1, 2**62, size=30000, dtype=np.int64)).astype(np.uint64)\n"
            "sets = [np.unique(np.concatenate([shared[rng.random(len(shared)) < 0.9], rng.integers(1, 2**62, size=2000, dtype=np.int64).astype(np.uint64)])) for _ in range(n)]\n"
            "cnt = np.array([len(x) for x in sets]); sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)\n"
            "dev = torch.device('cuda', 0)\n"
            "d_lo = torch.from_numpy(np.concatenate(sets).view(np.int64)).to(dev)\n"
            "d_min = torch.full((int(cnt.sum()),), 7, dtype=torch.int32, device=dev)\n"
            "d_hi = d_lo >> 8                                         # k > 32: the high word is part of the key (three-word records); order kept\n"
            "ctx = sp.Context(0)\nout = []\n"
            "forms = ((0, 1, 8), (8, 1, 16), (56, 1, 64), (3, 8, 64), (0, 1, 64))\n"
            "for k in (31, 63):\n"
            "  for first, stride, limit in forms:\n"
            "    part = torch.full((n, n), -1, dtype=torch.int32, device=dev)\n"
            "    torch.cuda.synchronize()\n"
            "    ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr() if k > 32 else None, sk_off, n, first, stride, part.data_ptr(), n_query=limit)\n"
            "    torch.cuda.synchronize()\n"
            "    out.append(part.cpu().numpy())\n"
            "np.save(sys.argv[1], np.stack(out))\n"
            "for half in (out[:5], out[5:]):\n"
            "  full = half[-1]\n"
            "  assert full[0, 1] == len(np.intersect1d(sets[0], sets[1])) and full[5, 60] == len(np.intersect1d(sets[5], sets[60]))\n"
            "  for o, (first, stride, limit) in zip(half, forms[:4]):\n"
            "    rows = np.arange(first, limit, stride)\n"
            "    own = np.zeros((n, n), bool); own[rows] = np.triu(np.ones((n, n), bool), 1)[rows]\n"
            "    assert (o[~own] == -1).all() and (o[own] == full[own]).all(), (first, stride, limit)\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for flt in ("0", "1"):
            f = os.path.join(d, "o%s.npy" % flt)
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, SPSP_DEBUG_FILTER=flt), capture_output=True, text=True, timeout=900)
            assert r.returncode == 0 and "ok" in r.stdout, (flt, r.stdout[-2000:], r.stderr[-3000:])
            outs.append(np.load(f))
        assert (outs[0] == outs[1]).all()


def test_partition_form_with_tens_of_thousands_of_parts():
    """SPSP_DEBUG_PART_MEAN=40 cuts the keys into ~30 000 parts (the number BASELINE configs[3]'s 5 x 10^7 keys need):
    list references, slice offsets and the scatter's per-part counters at that scale, on 2 048 C4-shaped sketches."""
    code = ("import sys\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch\nimport supersampler_amd as sp, dist_worker as dw\n"
            "n = 2048\nsets = dw.c4_shaped_sets(n, 600, seed=45)\nwant = dw.expected_inter(sets)\n"
            "sk = [sp.Sketch(31, 11, np.full(len(x), 7, np.uint32), x, np.zeros(len(x), np.uint64)) for x in sets]\n"
            "ctx = sp.Context(0)\ninter, card = ctx.compare(sk)\n"
            "assert (inter.astype(np.int64) == want).all() and int(want.sum()) > 1_000_000\n"
            "assert [int(c) for c in card] == [len(x) for x in sets]\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_PART_MEAN="40"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_config4_full_size_on_one_gpu(ctx):
    """BASELINE configs[3] at full size on ONE GPU: 10 000 sketches, 500 families of 20, ~5 000 keys each (5 x 10^7 keys,
    ~18 000 key classes), synthesised directly.  All pairs inside five families and 2 000 random pairs
    equal numpy set intersections; cardinalities and the number of non-zero pairs are as constructed."""
    rng = np.random.default_rng(4)
    n, fam_size = 10_000, 20
    sets = []
    for f in range(n // fam_size):
        anc = np.unique(rng.integers(1, 2**62, size=int(rng.integers(2500, 10000)), dtype=np.int64))
        for j in range(fam_size):
            keep = anc[rng.random(len(anc)) < (0.97, 0.73, 0.45)[j % 3]]
            extra = rng.integers(1, 2**62, size=len(anc) - len(keep), dtype=np.int64)
            sets.append(np.unique(np.concatenate([keep, extra])).astype(np.uint64))
    cnt = np.array([len(x) for x in sets])
    assert cnt.sum() > 45_000_000
    dev = torch.device("cuda", 0)
    sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)
    d_lo = torch.from_numpy(np.concatenate(sets).view(np.int64)).to(dev)
    d_min = torch.full((int(cnt.sum()),), 7, dtype=torch.int32, device=dev)
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 0, 1, d_inter.data_ptr())
    torch.cuda.synchronize()
    got = d_inter.cpu().numpy()
    del d_inter, d_lo, d_min
    pairs = [(i, j) for f in (0, 1, 77, 250, 499) for i in range(f * fam_size, (f + 1) * fam_size) for j in range(i + 1, (f + 1) * fam_size)]
    pairs += [tuple(sorted(int(x) for x in rng.integers(0, n, size=2))) for _ in range(2000)]
    for i, j in pairs:
        if i != j:
            assert got[i, j] == len(np.intersect1d(sets[i], sets[j], assume_unique=True)), (i, j)
    assert np.count_nonzero(np.triu(got, 1)) == (n // fam_size) * fam_size * (fam_size - 1) // 2      # random 62-bit keys: no chance matches
    assert (np.tril(got) == 0).all()


def test_small_problem_form_at_bench_size():
    """k_parts_group_small: 100 sketches in families of 10, ~4 000 keys each (about 280 parts), on a whole-device
    context and on contexts that say they own 32 and 2 CUs (grids and part counts follow the CU count) -- every matrix
    equals an inverted-index count in numpy; a pair of identical sketches makes single cells large (LDS counters are
    16 bits wide), one sketch is empty."""
    rng = np.random.default_rng(17)
    n, per = 100, 4000
    fams = [np.unique(rng.integers(1, 2**62, size=per + per // 4, dtype=np.int64)) for _ in range(n // 10)]
    sets = []
    for i in range(n):
        base = fams[i // 10]
        keep = base[rng.random(len(base)) < 0.8]
        extra = rng.integers(1, 2**62, size=int(rng.integers(0, 50)), dtype=np.int64)
        sets.append(np.unique(np.concatenate([keep, extra])).astype(np.uint64))
    sets[11] = sets[10].copy()                            # identical sketches: cells of ~4 000
    sets[37] = np.zeros(0, np.uint64)                     # an empty one
    holders = {}
    for i, keys in enumerate(sets):
        for key in keys.tolist():
            holders.setdefault(key, []).append(i)
    want = np.zeros((n, n), dtype=np.int64)
    for hs in holders.values():
        if len(hs) > 1:
            a = np.array(hs)
            ii, jj = np.triu_indices(len(a), 1)
            np.add.at(want, (a[ii], a[jj]), 1)
    assert want[10, 11] == len(sets[10]) and want.sum() > 1_000_000
    sk = [sp.Sketch(31, 11, np.full(len(s), 5, np.uint32), s, None) for s in sets]
    for cus in (0, 32, 2):
        ctx = sp.Context(0)
        if cus:
            ctx.set_cu_count(cus)
        ctx.timing_enable(True)
        inter, card = ctx.compare(sk)
        tm = ctx.timing_read()
        assert tm["accumulate_launches"] == 0, "expected the small-problem form (no row-sum kernel)"
        assert (inter.astype(np.int64) == want).all(), cus
        assert [int(c) for c in card] == [len(s) for s in sets]
        ctx.close()


def test_small_problem_form_respects_the_part_limit():
    """The small-problem form (<= 128 sketches) cuts the keys into parts of half the size, so its part count reaches
    the scatter's limit (32 000 parts: LDS counters, 15 part bits) at half the keys the general form does -- two sketches
    of 3 x 10^7 keys each, say.  SPSP_DEBUG_SMALL_MEAN shrinks the planned part size so that 96 sketches of ~450 keys
    get there: at mean 1 (43 000 parts) the call must take the general form, at mean 2 (21 000 parts) the small-problem
    form runs with that many parts; both equal the inverted-index count, and the small form leaves the diagonal and the
    lower triangle of the caller's matrix untouched like every other form."""
    code = ("import sys, os\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch\nimport supersampler_amd as sp, dist_worker as dw\n"
            "n = 96\nsets = dw.c4_shaped_sets(n, 560, seed=46)\nwant = dw.expected_inter(sets)\n"
            "cnt = np.array([len(x) for x in sets]); assert cnt.sum() > 40000\n"
            "dev = torch.device('cuda', 0)\n"
            "sk_off = np.zeros(n + 1, np.uint64); sk_off[1:] = np.cumsum(cnt)\n"
            "d_lo = torch.from_numpy(np.concatenate(sets).view(np.int64)).to(dev)\n"
            "d_min = torch.full((int(cnt.sum()),), 7, dtype=torch.int32, device=dev)\n"
            "d_inter = torch.full((n, n), -7, dtype=torch.int32, device=dev)\ntorch.cuda.synchronize()\n"
            "ctx = sp.Context(0)\nctx.timing_enable(True)\n"
            "ctx.compare_device(31, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 0, 1, d_inter.data_ptr())\n"
            "tm = ctx.timing_read()\ngot = d_inter.cpu().numpy()\n"
            "up = np.triu(np.ones((n, n), bool), 1)\n"
            "assert (got[up] == want[up]).all() and int(want.sum()) > 100000\n"
            "assert (got[~up] == -7).all(), 'cells outside (i, j > i) were written'\n"
            "small = tm['accumulate_launches'] == 0\n"
            "assert small == (os.environ['SPSP_DEBUG_SMALL_MEAN'] == '2'), (small, tm)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for mean in ("1", "2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPSP_DEBUG_SMALL_MEAN=mean), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, (mean, r.stdout[-2000:], r.stderr[-3000:])


def test_compare_files_over_several_contexts_equals_one_and_the_oracle(tmp_path):
    """spsp_compare_files_multi: the comparator split by KEY over several contexts (one per device on a multi-GPU node; here
    one, two, three and five contexts on device 0) writes the CSV bytes of the single-context driver and of the oracle's
    comparator + printers (Comparator.cpp:39-74, 362-460): related and unrelated sketches, an empty sketch, a number of
    files that no context count divides, query mode (rows of the first files only), k <= 32 and k > 32; bin/comparator
    with SPSP_DEVICES="0,0" takes the same route."""
    import gzip
    import subprocess
    for (k, m, s, n_files) in ((31, 11, 40.0, 23), (63, 15, 25.0, 11)):
        rng = np.random.default_rng(900 + k)
        anc = [synth.random_genome(rng, 60_000) for _ in range(3)]
        paths, payloads = [], []
        for i in range(n_files):
            g = synth.mutate(rng, anc[i % 3], [0.0, 0.01, 0.03][(i // 3) % 3]) if i != 7 else synth.random_genome(rng, k - 2)   # file 7: no k-mer at all
            pl = orc.sketch_fasta(synth.to_fasta(g, "g%d" % i, n_records=1 + i % 2), k, m, s)[0]
            pth = tmp_path / ("k%d_s%02d.gz" % (k, i))
            sp.write_gz(str(pth), pl)
            paths.append(str(pth)); payloads.append(pl)
        inter, card, _, _ = orc.compare(payloads)
        want = {jac: orc.csv(jac, paths, inter, card, None, 6, 0.0) for jac in (True, False)}
        with sp.Context(0) as ctx:
            ctx.compare_files(paths, str(tmp_path / "one"))
        for devs in ([0], [0, 0], [0, 0, 0], [0] * 5):
            pre = str(tmp_path / ("multi%d_%d" % (k, len(devs))))
            st = sp.compare_files_multi(devs, paths, pre)
            assert st["compare_calls"] == 1
            for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
                got = gzip.open(pre + suf, "rb").read()
                assert got == want[jac], (k, devs, suf)
                assert got == gzip.open(str(tmp_path / "one") + suf, "rb").read()
        # more contexts than files: two files and ONE file over three contexts (contexts without a sketch take part in the exchange all the same)
        for few in (2, 1):
            fi, fc, _, _ = orc.compare(payloads[:few])
            pre = str(tmp_path / ("few%d_%d" % (k, few)))
            sp.compare_files_multi([0, 0, 0], paths[:few], pre)
            for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
                assert gzip.open(pre + suf, "rb").read() == orc.csv(jac, paths[:few], fi, fc, None, 6, 0.0), (k, few, suf)
        # query mode: the first four files are the queries
        qi, qc, _, _ = orc.compare(payloads, n_query=4)
        pre = str(tmp_path / ("q%d" % k))
        sp.compare_files_multi([0, 0, 0], paths, pre, n_query=4)
        for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
            assert gzip.open(pre + suf, "rb").read() == orc.csv(jac, paths, qi, qc, 4, 6, 0.0), (k, suf)
    fof = tmp_path / "fof.txt"
    fof.write_text("\n".join(paths) + "\n")
    r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", str(fof), "-o", "cli2"], cwd=tmp_path, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SPSP_DEVICES="0,0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert gzip.open(tmp_path / "cli2_jaccard.csv.gz", "rb").read() == want[True]
    assert r.stdout.split("\n")[1] == "I found %d documents" % n_files and "Comparisons done" in r.stdout


def test_pair_matrix_cells_round_trip():
    """spsp_matrix_cells_device / spsp_matrix_add_cells_device: the non-zero cells (i < j) of a dense matrix as packed words
    and back; rows outside [row_first, row_limit) stay out; too little room is reported with the number needed; a cell that
    names a sketch outside the matrix is refused."""
    import torch
    rng = np.random.default_rng(12)
    n = 1500
    dense = np.zeros((n, n), dtype=np.uint32)
    ii, jj = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    dense[ii, jj] = rng.integers(1, 1 << 20, 4000)
    up = np.triu(dense, 1)
    d = torch.from_numpy(dense.view(np.int32)).cuda()
    d_cells = torch.zeros(8000, dtype=torch.int64, device="cuda")
    with sp.Context(0) as ctx:
        cnt = ctx.matrix_cells_device(d.data_ptr(), n, d_cells.data_ptr(), 8000)
        assert cnt == int(np.count_nonzero(up))
        cells = d_cells[:cnt].cpu().numpy().view(np.uint64)
        back = np.zeros((n, n), dtype=np.uint32)
        back[(cells >> np.uint64(48)).astype(np.int64), ((cells >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)] = (cells & np.uint64(0xffffffff)).astype(np.uint32)
        assert (back == up).all()
        part = ctx.matrix_cells_device(d.data_ptr(), n, d_cells.data_ptr(), 8000, row_first=100, row_limit=700)
        assert part == int(np.count_nonzero(up[100:700]))
        with pytest.raises(sp.SpspError) as e:
            ctx.matrix_cells_device(d.data_ptr(), n, d_cells.data_ptr(), 10)
        assert e.value.code == sp.ERR_OVERFLOW and str(cnt) in str(e.value)
        acc = torch.from_numpy(np.triu(dense, 1).view(np.int32).copy()).cuda()
        ctx.matrix_cells_device(d.data_ptr(), n, d_cells.data_ptr(), 8000)
        ctx.matrix_add_cells_device(acc.data_ptr(), n, d_cells.data_ptr(), cnt)
        assert (acc.cpu().numpy().view(np.uint32) == 2 * up).all()
        bad = torch.tensor([(5 << 48) | (2000 << 32) | 1], dtype=torch.int64, device="cuda")
        with pytest.raises(sp.SpspError):
            ctx.matrix_add_cells_device(acc.data_ptr(), n, bad.data_ptr(), 1)


def test_pair_matrix_as_cells_straight_from_the_row_sums(ctx, tmp_path):
    """spsp_compare_cells_device / spsp_compare_slots_cells_device: the pair matrix as packed non-zero cells without a dense
    matrix in between (the partition form's row sums emit them), equal to the dense spsp_compare_device result cell for
    cell -- 2 048 configs[3]-shaped sketches all-vs-all, in query mode (through the dense matrix: rows limited), with too
    little room (the count needed comes back), for a small problem (the LDS pair-counter form: dense, then sparsified) and
    as the four partial matrices of a key-partitioned split; spsp_compare_files over 1 100 sketch files takes the same
    route (n >= 1024) and writes the oracle's CSV bytes."""
    import gzip
    dev = torch.device("cuda", 0)
    n = 2048
    D = synth.direct_family_sketches(n, fam_size=16, seed=11, device=dev, skm_range=(20, 40))
    dense = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, dense.data_ptr())
    torch.cuda.synchronize()
    want = np.triu(dense.cpu().numpy().view(np.uint32), 1)
    assert np.count_nonzero(want) >= (n // 16) * 100

    def unpack(cells, cnt, size):
        c = cells[:cnt].cpu().numpy().view(np.uint64)
        out = np.zeros((size, size), dtype=np.uint32)
        ii, jj = (c >> np.uint64(48)).astype(np.int64), ((c >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
        assert len(set(zip(ii.tolist(), jj.tolist()))) == cnt              # every cell once
        out[ii, jj] = (c & np.uint64(0xffffffff)).astype(np.uint32)
        return out
    scratch = torch.full((n, n), 7, dtype=torch.int32, device=dev)
    cells = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    cnt = ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
    assert cnt == np.count_nonzero(want) and (unpack(cells, cnt, n) == want).all()
    assert int((scratch != 7).sum().item()) == 0                            # the dense matrix was never written
    cnt_q = ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel(), n_query=100)
    assert (unpack(cells, cnt_q, n)[:100] == want[:100]).all() and cnt_q == np.count_nonzero(want[:100])
    with pytest.raises(sp.SpspError) as e:
        ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, scratch.data_ptr(), cells.data_ptr(), 1000)
    assert e.value.code == sp.ERR_OVERFLOW and str(cnt) in str(e.value) and e.value.cells_needed == cnt
    # a small problem (<= 128 sketches: pair counters in LDS, dense result) comes back sparse all the same
    off100 = D.sk_off[:101].copy()
    c100 = ctx.compare_cells_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, off100, 100, scratch.data_ptr(), cells.data_ptr(), cells.numel())
    assert (unpack(cells, c100, 100) == want[:100, :100]).all()
    # key-partitioned: four ranks' partial matrices as cells add up to the whole
    G, per = 4, n // 4
    off = D.sk_off.astype(np.int64)
    sends, cap = [], None
    most = max(int(off[(r + 1) * per] - off[r * per]) for r in range(G))
    cap = int(most / G * 1.3) + 1024
    slot_sz = sp.slot_bytes(per, cap, 31)
    for r in range(G):
        a, b = int(off[r * per]), int(off[(r + 1) * per])
        snd = torch.zeros(G * slot_sz, dtype=torch.uint8, device=dev)
        bm, bl = D.minimizer[a:b].contiguous(), D.kmer_lo[a:b].contiguous()
        ctx.partition_keys_device(31, bm.data_ptr(), bl.data_ptr(), None, (off[r * per:(r + 1) * per + 1] - a).astype(np.uint64), per, G, cap, snd.data_ptr())
        torch.cuda.synchronize()
        sends.append(snd)
    total = torch.zeros((n, n), dtype=torch.int32, device=dev)
    for r in range(G):
        recv = torch.cat([sends[s][r * slot_sz:(r + 1) * slot_sz] for s in range(G)])
        torch.cuda.synchronize()
        c = ctx.compare_slots_cells_device(31, recv.data_ptr(), G, per, cap, scratch.data_ptr(), cells.data_ptr(), cells.numel())
        assert 0 < c <= cnt
        ctx.matrix_add_cells_device(total.data_ptr(), n, cells.data_ptr(), c)
    torch.cuda.synchronize()
    assert (np.triu(total.cpu().numpy().view(np.uint32), 1) == want).all()
    # spsp_compare_files with n >= 1024: cells over PCIe instead of 4 n^2 bytes
    m = 1100
    paths, payloads = [], []
    for i in range(m):
        pl = D.payload(i)
        pth = str(tmp_path / ("s%04d.gz" % i))
        sp.write_gz(pth, pl, 1)
        paths.append(pth); payloads.append(pl)
    ctx.compare_files(paths, str(tmp_path / "big"))
    inter, card, _, _ = orc.compare(payloads)
    for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
        assert gzip.open(str(tmp_path / "big") + suf, "rb").read() == orc.csv(jac, paths, inter, card, None, 6, 0.0), suf


@pytest.mark.gpu
def test_compare_files_of_one_species_dense_matrix_as_cells(tmp_path):
    """1 100 sketch files of one species: from 1 024 files on the drivers take the pair matrix as CELLS -- here every cell is
    non-zero (6 x 10^5 of them: the first guess of the cell buffer is too small, the spill's result goes through the dense
    matrix and is sparsified again into a larger one) and every CSV row is dense (row blocks sized for threads, not for rows of
    "0,").  CSV bytes of one context and of two against the oracle's comparator + printers."""
    import gzip
    rng = np.random.default_rng(99)
    anc = synth.random_genome(rng, 40_000)
    paths, payloads = [], []
    for i in range(1100):
        pl = orc.sketch_fasta(synth.to_fasta(synth.mutate(rng, anc, [0.0, 0.001, 0.004][i % 3]), "g%d" % i), 31, 11, 40.0)[0]
        pth = tmp_path / ("d_%04d.gz" % i)
        sp.write_gz(str(pth), pl, 1)
        paths.append(str(pth)); payloads.append(pl)
    inter, card, _, _ = orc.compare(payloads)
    assert int((np.triu(inter, 1) > 0).sum()) == 1100 * 1099 // 2
    with sp.Context(0) as ctx:
        ctx.compare_files(paths, str(tmp_path / "one"))
    sp.compare_files_multi([0, 0], paths, str(tmp_path / "two"))
    for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
        want = orc.csv(jac, paths, inter, card, None, 6, 0.0)
        assert gzip.open(str(tmp_path / "one") + suf, "rb").read() == want
        assert gzip.open(str(tmp_path / "two") + suf, "rb").read() == want


@pytest.mark.gpu
def test_compare_files_read_into_one_block_whatever_the_container(tmp_path):
    """spsp_compare_files from 256 files on lays the payloads of every reader thread's range of files down back to back in a region
    of its own (each at the 16-byte-rounded end of the one before) and the decoder uploads them from where they lie, a copy per region.  300 sketch files in every container the reader knows -- gzip
    of one member, gzip of two members (the trailer's length is then not the payload's: that file takes the general reader
    and the decoder gathers), a zlib wrapper, plain text, an empty sketch (header only) -- as ONE kind each and mixed, in one
    context and over two, against the oracle's CSV bytes; a truncated gzip file is the reader's error, not a wrong matrix."""
    import gzip
    import zlib
    rng = np.random.default_rng(123)
    anc = [synth.random_genome(rng, 30_000) for _ in range(6)]
    payloads = []
    for i in range(300):
        g = synth.mutate(rng, anc[i % 6], 0.002 * (i % 4))
        payloads.append(orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), 31, 11, 60.0)[0])
    payloads[17] = payloads[17][:payloads[17].index(b"\n") + 1]           # a sketch with no bucket at all
    inter, card, _, _ = orc.compare(payloads)

    def container(kind, pl):
        if kind == 0:
            return gzip.compress(pl, 1)
        if kind == 1:
            cut = len(pl) // 3
            return gzip.compress(pl[:cut], 6) + gzip.compress(pl[cut:], 1)
        if kind == 2:
            return zlib.compress(pl, 6)
        return pl
    for tag, kinds in (("gz", [0] * 300), ("plain", [3] * 300), ("mixed", [int(x) for x in rng.integers(0, 4, 300)]), ("two", [1] * 300)):
        paths = []
        for i, pl in enumerate(payloads):
            pth = str(tmp_path / ("%s_%03d" % (tag, i)))
            open(pth, "wb").write(container(kinds[i], pl))
            paths.append(pth)
        with sp.Context(0) as ctx:
            ctx.compare_files(paths, str(tmp_path / (tag + "_one")))
            ctx.compare_files(paths, str(tmp_path / (tag + "_q")), n_query=7)
        sp.compare_files_multi([0, 0], paths, str(tmp_path / (tag + "_multi")))
        for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
            want = orc.csv(jac, paths, inter, card, None, 6, 0.0)
            assert gzip.open(str(tmp_path / (tag + "_one")) + suf, "rb").read() == want, (tag, suf)
            assert gzip.open(str(tmp_path / (tag + "_multi")) + suf, "rb").read() == want, (tag, suf)
            assert gzip.open(str(tmp_path / (tag + "_q")) + suf, "rb").read() == orc.csv(jac, paths, inter, card, 7, 6, 0.0), (tag, suf)
    # damaged files: the general reader's verdict
    bad = list(paths)
    blob = gzip.compress(payloads[5], 1)
    open(bad[5], "wb").write(blob[:-8] + bytes([blob[-8] ^ 0xff]) + blob[-7:])      # the trailer's CRC-32 does not match the data
    with sp.Context(0) as ctx:
        with pytest.raises(sp.SpspError):
            ctx.compare_files(bad, str(tmp_path / "bad"))
        ctx.compare_files(paths[:5] + paths[6:], str(tmp_path / "after"))      # ... and the context works on


@pytest.mark.gpu
def test_compare_files_of_one_species_over_contexts_equal_the_oracle(tmp_path):
    """320 sketch FILES of one species (one ancestor, 0-1 % substitutions): every part of the comparison overflows, so the
    file-level drivers go through the spill and the bit columns -- with one context (cells through the dense matrix) and
    split by key over three (each context spills its own hash class; partial cells added on the host).  CSV bytes against
    the oracle's comparator + printers, all-vs-all and with 9 query files."""
    import gzip
    import subprocess
    rng = np.random.default_rng(4242)
    anc = synth.random_genome(rng, 120_000)
    k, m, s = 31, 11, 30.0
    paths, payloads = [], []
    for i in range(320):
        g = synth.mutate(rng, anc, [0.0, 0.001, 0.003, 0.01][i % 4])
        pl = orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0]
        pth = tmp_path / ("sp_%03d.gz" % i)
        sp.write_gz(str(pth), pl, 1)
        paths.append(str(pth)); payloads.append(pl)
    code = ("import sys\nsys.path.insert(0, %r)\nimport supersampler_amd as sp\n"
            "paths = [l.strip() for l in open(sys.argv[1])]\n"
            "with sp.Context(0) as ctx:\n    ctx.compare_files(paths, sys.argv[2] + '_one')\n    ctx.compare_files(paths, sys.argv[2] + '_oneq', n_query=9)\n"
            "sp.compare_files_multi([0, 0, 0], paths, sys.argv[2] + '_multi')\nsp.compare_files_multi([0, 0], paths, sys.argv[2] + '_multiq', n_query=9)\nprint('ok')\n") % ROOT
    fof = tmp_path / "fof.txt"
    fof.write_text("\n".join(paths) + "\n")
    r = subprocess.run([sys.executable, "-c", code, str(fof), str(tmp_path / "out")], env=dict(os.environ, SPSP_DEBUG_SPILL_TRACE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stderr.count("spsp spill:") >= 4, r.stderr[-3000:]          # the one-context call and every context of the split
    inter, card, _, _ = orc.compare(payloads)
    qi, qc, _, _ = orc.compare(payloads, n_query=9)
    assert int((inter > 0).sum()) == 320 * 319 // 2
    for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
        want = orc.csv(jac, paths, inter, card, None, 6, 0.0)
        assert gzip.open(str(tmp_path / "out") + "_one" + suf, "rb").read() == want
        assert gzip.open(str(tmp_path / "out") + "_multi" + suf, "rb").read() == want
        want_q = orc.csv(jac, paths, qi, qc, 9, 6, 0.0)
        assert gzip.open(str(tmp_path / "out") + "_oneq" + suf, "rb").read() == want_q
        assert gzip.open(str(tmp_path / "out") + "_multiq" + suf, "rb").read() == want_q


@pytest.mark.gpu
def test_rows_ordered_by_min_hash_give_the_same_matrix(ctx):
    """1 500 sketches in families of 20, once family by family and once in a random order: the row sums of the shuffled
    collection run in the order of the sketches' min-hash signatures (k_row_signature / k_row_order: a family's rows side by
    side behind one L2) -- scheduling only, so every cell must equal the cell of the same two sketches in the first matrix.
    Several calls per context (a context that saw a well-ordered input skips the ordering for a while)."""
    n, F = 2600, 20
    dev = torch.device("cuda", 0)
    D = synth.direct_family_sketches(n, fam_size=F, seed=11, device=dev, skm_range=(40, 90))
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    off0 = D.sk_off.astype(np.int64)
    cnt = np.diff(off0)[perm]
    new_off = np.zeros(n + 1, np.int64)
    new_off[1:] = np.cumsum(cnt)
    src = torch.from_numpy(np.repeat(off0[:-1][perm] - new_off[:-1], cnt)).to(dev) + torch.arange(int(new_off[-1]), device=dev)
    mn_b, lo_b = D.minimizer[src].contiguous(), D.kmer_lo[src].contiguous()
    a = torch.zeros((n, n), dtype=torch.int32, device=dev)
    b = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, a.data_ptr())
        ctx.compare_device(31, mn_b.data_ptr(), lo_b.data_ptr(), None, new_off.astype(np.uint64), n, 0, 1, b.data_ptr())
        A = np.triu(a.cpu().numpy(), 1)
        A = A + A.T                                       # the pair count of two sketches, whichever comes first
        B = np.triu(b.cpu().numpy(), 1)
        want = np.triu(A[np.ix_(perm, perm)], 1)
        assert (B == want).all(), rep
    assert np.count_nonzero(A) >= n * (F - 1) * 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["first", "middle", "last"])
def test_one_sketch_far_longer_than_the_others(ctx, where):
    """600 sketches of a few hundred keys and ONE of 90 000 (a eukaryote among bacteria): the long row is summed in slices by a
    launch of its own, the launch of all rows passes it by and keeps its 16-bit counters (k_accumulate_sparse, long_limit).
    Every cell against numpy set algebra; dense and as cells."""
    import torch
    rng = np.random.default_rng(17)
    n = 601
    uni_mn = rng.integers(0, 2**22, 4000).astype(np.uint32)
    uni_lo = rng.integers(0, 2**62, 4000).astype(np.uint64)
    pos = {"first": 0, "middle": 300, "last": n - 1}[where]
    sets = []
    for i in range(n):
        if i == pos:
            pick = rng.random(4000) < 0.3
            extra = 90_000
        else:
            pick = rng.random(4000) < rng.choice([0.02, 0.1])
            extra = int(rng.integers(0, 50))
        mn = np.concatenate([uni_mn[pick], rng.integers(0, 2**22, extra).astype(np.uint32)])
        lo = np.concatenate([uni_lo[pick], rng.integers(2**62, 2**63, extra).astype(np.uint64)])     # (private keys: another range)
        order = np.lexsort((lo, mn))
        sets.append((mn[order], lo[order]))
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(s[0]) for s in sets])
    dev = torch.device("cuda", 0)
    d_mn = torch.from_numpy(np.concatenate([s[0] for s in sets]).view(np.int32)).to(dev)
    d_lo = torch.from_numpy(np.concatenate([s[1] for s in sets]).view(np.int64)).to(dev)
    d_inter = torch.full((n, n), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.compare_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, off, n, 0, 1, d_inter.data_ptr())
    got = d_inter.cpu().numpy()
    in_uni = np.zeros((n, 4000), bool)                     # shared keys come from the universe only
    key_id = {(int(a), int(b)): j for j, (a, b) in enumerate(zip(uni_mn, uni_lo))}
    for i, (mn, lo) in enumerate(sets):
        for a, b in zip(mn.tolist(), lo.tolist()):
            j = key_id.get((a, b))
            if j is not None:
                in_uni[i, j] = True
    want = np.triu(in_uni.astype(np.int64) @ in_uni.astype(np.int64).T, 1)
    assert (np.triu(got, 1) == want).all()
    assert (got[np.tril_indices(n)] == -1).all()           # the diagonal and the lower triangle stay the caller's
    cells = torch.zeros(n * n, dtype=torch.int64, device=dev)
    scratch = torch.zeros((n, n), dtype=torch.int32, device=dev)
    cnt = ctx.compare_cells_device(31, d_mn.data_ptr(), d_lo.data_ptr(), None, off, n, scratch.data_ptr(), cells.data_ptr(), cells.numel())
    c = cells[:cnt].cpu().numpy()
    back = np.zeros((n, n), np.int64)
    back[(c >> 48) & 0xffff, (c >> 32) & 0xffff] = c & 0xffffffff
    assert cnt == np.count_nonzero(want) and (back == want).all()


@pytest.mark.parametrize("use_hi", [False, True])
def test_scatter_tiles_of_any_shape_give_the_same_matrix(ctx, use_hi):
    """k_parts_scatter_tiles (the scatter over tiles of 32 sketches x the same share of each): scheduling only, so any
    collection gives the matrix of set algebra -- a block whose longest sketch shares nothing with the other 31, blocks with
    empty sketches, a ragged last block, a block of identical sketches, sketches of one key; all-vs-all and in query mode;
    and a sketch that is not sorted is refused whichever tile its keys fall into."""
    rng = np.random.default_rng(31 + use_hi)
    uni = [(int(rng.integers(100, 2**20)), int(rng.integers(0, 2**62)) if use_hi else 0, int(rng.integers(0, 2**62))) for _ in range(6000)]
    sets = []
    low = {(1, 0, int(x)) for x in rng.integers(0, 2**62, 6000)}                       # minimizer 1: below every universe key
    high = {(2**22 - 1, 0, int(x)) for x in rng.integers(0, 2**62, 6000)}              # above every universe key
    for i in range(32 * 5 + 9):
        blk, q = divmod(i, 32)
        pick = lambda p: {uni[j] for j in np.nonzero(rng.random(len(uni)) < p)[0]}
        if blk == 0: st = low if q == 5 else pick(0.65)                                  # longest sketch below the others
        elif blk == 1: st = high if q == 0 else pick(0.6)                                # ... above the others
        elif blk == 2: st = set() if q % 3 else pick(0.2)                                # mostly empty
        elif blk == 3: st = set(sets[32]) if q else pick(0.5)                            # (q = 0 sets the block's own sketch, the others copy sketch 32)
        elif blk == 4: st = {uni[int(rng.integers(0, len(uni)))]}                        # one key each
        else: st = pick(0.1)
        sets.append(st)
    n = len(sets)
    sk = []
    for st in sets:
        keys = sorted(st)
        sk.append(sp.Sketch(63 if use_hi else 31, 11, np.array([x[0] for x in keys], np.uint32), np.array([x[2] for x in keys], np.uint64),
                            np.array([x[1] for x in keys], np.uint64)))
    idx = {key: c for c, key in enumerate(sorted(set().union(*sets)))}
    inc = np.zeros((n, len(idx)), np.int64)
    for i, st in enumerate(sets):
        inc[i, [idx[x] for x in st]] = 1
    want = np.triu(inc @ inc.T, 1).astype(np.uint32)
    inter, card = ctx.compare(sk)
    assert (inter == want).all() and [int(c) for c in card] == [len(st) for st in sets]
    inter_q, _ = ctx.compare(sk, n_query=40)
    assert (inter_q[:40] == want[:40]).all()
    # a sketch that is not sorted is refused
    bad = sk[70]
    bad.kmer_lo[[10, 400]] = bad.kmer_lo[[400, 10]]
    bad.minimizer[[10, 400]] = bad.minimizer[[400, 10]]
    bad.kmer_hi[[10, 400]] = bad.kmer_hi[[400, 10]]
    with pytest.raises(sp.SpspError):
        ctx.compare(sk)
