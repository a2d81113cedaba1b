"""The C++ multi-device path (include/spsp.h: spsp_compare_files_multi, spsp_sketch_files_multi, spsp_device_count) over
DISTINCT devices: peer access, hipMemcpyPeerAsync of the exchange slots, one context and host thread per GPU.  These tests
need a node: they skip on a one-GPU box (where tests/test_exchange.py and tests/test_gpu.py run the same code with several
contexts on device 0) and un-skip by themselves wherever two or more gfx950 devices are visible.  What they replace in the
reference is one thread over one merge (Comparator.cpp:39-74) and an OpenMP loop over files (SubSampler.cpp:771-793), so the
bar is the same as on one device: the oracle's CSV / payload BYTES."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import supersampler_amd as sp
from oracle import oracle_py as orc
from supersampler_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SUF = ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz"))


@pytest.fixture(scope="module")
def devices():
    n = sp.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (spsp_device_count() = %d)" % n)
    return list(range(n))


def _sketch_files(tmp_path, k, m, s, n_files, seed, empty=7, length=60_000):
    rng = np.random.default_rng(seed)
    anc = [synth.random_genome(rng, length) for _ in range(3)]
    paths, payloads = [], []
    for i in range(n_files):
        g = synth.mutate(rng, anc[i % 3], [0.0, 0.01, 0.03][(i // 3) % 3]) if i != empty else synth.random_genome(rng, k - 2)   # no k-mer at all
        pl = orc.sketch_fasta(synth.to_fasta(g, "g%d" % i, n_records=1 + i % 2), k, m, s)[0]
        pth = tmp_path / ("k%d_s%03d.gz" % (k, i))
        sp.write_gz(str(pth), pl, 1)
        paths.append(str(pth)); payloads.append(pl)
    return paths, payloads


@pytest.mark.parametrize("k,m,s,n_files", [(31, 11, 40.0, 37), (63, 15, 25.0, 19)])
def test_compare_files_over_all_devices_equals_one_device_and_the_oracle(devices, tmp_path, k, m, s, n_files):
    """all-vs-all and query mode, k <= 32 and k > 32, one empty sketch, a file count no device count divides: every visible
    device, then the first two, then the last and the first (an order that is not 0, 1, ...)"""
    paths, payloads = _sketch_files(tmp_path, k, m, s, n_files, 900 + k)
    inter, card, _, _ = orc.compare(payloads)
    qi, qc, _, _ = orc.compare(payloads, n_query=5)
    with sp.Context(0) as ctx:
        ctx.compare_files(paths, str(tmp_path / "one"))
    for tag, devs in (("all", devices), ("two", devices[:2]), ("rev", [devices[-1], devices[0]])):
        pre = str(tmp_path / ("m_%s" % tag))
        st = sp.compare_files_multi(devs, paths, pre)
        assert st["compare_calls"] == 1
        sp.compare_files_multi(devs, paths, pre + "q", n_query=5)
        for jac, suf in SUF:
            got = gzip.open(pre + suf, "rb").read()
            assert got == orc.csv(jac, paths, inter, card, None, 6, 0.0), (tag, suf)
            assert got == gzip.open(str(tmp_path / "one") + suf, "rb").read()
            assert gzip.open(pre + "q" + suf, "rb").read() == orc.csv(jac, paths, qi, qc, 5, 6, 0.0), (tag, suf, "query")
    # fewer files than devices: contexts without a sketch take part in the exchange all the same
    fi, fc, _, _ = orc.compare(payloads[:1])
    sp.compare_files_multi(devices, paths[:1], str(tmp_path / "few"))
    for jac, suf in SUF:
        assert gzip.open(str(tmp_path / "few") + suf, "rb").read() == orc.csv(jac, paths[:1], fi, fc, None, 6, 0.0)


def test_compare_files_of_one_species_over_all_devices(devices, tmp_path):
    """400 sketch files of ONE species: every context's hash class overflows its parts (spill + bit columns per device), the
    partial cells of all devices add up to the oracle's dense matrix"""
    rng = np.random.default_rng(4243)
    anc = synth.random_genome(rng, 120_000)
    k, m, s = 31, 11, 30.0
    paths, payloads = [], []
    for i in range(400):
        pl = orc.sketch_fasta(synth.to_fasta(synth.mutate(rng, anc, [0.0, 0.001, 0.003, 0.01][i % 4]), "g%d" % i), k, m, s)[0]
        pth = tmp_path / ("sp_%03d.gz" % i)
        sp.write_gz(str(pth), pl, 1)
        paths.append(str(pth)); payloads.append(pl)
    inter, card, _, _ = orc.compare(payloads)
    sp.compare_files_multi(devices, paths, str(tmp_path / "all"))
    for jac, suf in SUF:
        assert gzip.open(str(tmp_path / "all") + suf, "rb").read() == orc.csv(jac, paths, inter, card, None, 6, 0.0), suf


def test_comparator_cli_uses_every_visible_device(devices, tmp_path):
    """bin/comparator WITHOUT SPSP_DEVICES: spsp_device_count() devices once there are SPSP_PER_DEVICE sketches for each
    (comparator_main.cpp; 512 by default, lowered here so that 37 files are split), same CSV bytes and stdout lines"""
    k, m, s = 31, 11, 40.0
    paths, payloads = _sketch_files(tmp_path, k, m, s, 37, 31)
    inter, card, _, _ = orc.compare(payloads)
    (tmp_path / "fof.txt").write_text("\n".join(paths) + "\n")
    env = {k_: v for k_, v in os.environ.items() if k_ != "SPSP_DEVICES"}
    for per, tag in (("4", "split"), (None, "default")):
        e = dict(env, SPSP_DEBUG_MULTI_TRACE="1")
        if per:
            e["SPSP_PER_DEVICE"] = per
        r = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "fof.txt", "-o", tag], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.split("\n")[1] == "I found 37 documents" and "Comparisons done" in r.stdout
        used = [ln for ln in r.stderr.splitlines() if ln.startswith("spsp multi:")]
        assert used and ("%d contexts" % (min(len(devices), 37 // 4) if per else 1)) in used[0], r.stderr[-2000:]
        for jac, suf in SUF:
            assert gzip.open(tmp_path / (tag + suf), "rb").read() == orc.csv(jac, paths, inter, card, None, 6, 0.0), (tag, suf)


def test_sketch_files_dealt_over_all_devices(devices, tmp_path):
    """spsp_sketch_files_multi over every visible device (-a 1: batches dealt; -a 2: per-file jobs dealt) writes the oracle's
    payload bytes; bin/sub_sampler without SPSP_DEVICES deals 40 files from SPSP_PER_DEVICE=4 files per device on"""
    k, m, s = 31, 11, 60.0
    gs = synth.family_genomes(78, 40, 120_000, 4, [0.0, 0.01, 0.03])
    ins, texts = [], []
    for i, g in enumerate(gs):
        t = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
        if i == 4:
            t = t + t.replace(b">g4", b">again")
        pth = tmp_path / ("d%02d.fa" % i)
        pth.write_bytes(t)
        ins.append(str(pth)); texts.append(t)
    for ab in (1, 2):
        want = [orc.sketch_fasta(t, k, m, s, ab) for t in texts]
        outs = [str(tmp_path / ("o%d_%02d.gz" % (ab, i))) for i in range(len(ins))]
        res, _, _ = sp.sketch_files(ins, outs, k, m, s, abundance=ab, threads=8, devices=devices)
        for i, (rc, st, err) in enumerate(res):
            assert rc == 0 and sp.read_file(outs[i]) == want[i][0], (ab, i, rc, err)
            assert st["selected_kmer_number"] == want[i][1]["selected_kmer_number"]
    (tmp_path / "fof.txt").write_text("\n".join(ins) + "\n")
    env = {k_: v for k_, v in os.environ.items() if k_ != "SPSP_DEVICES"}
    r = subprocess.run([os.path.join(ROOT, "bin", "sub_sampler"), "-f", "fof.txt", "-k", str(k), "-m", str(m), "-s", str(int(s)), "-t", "8", "-p", "all_"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600, env=dict(env, SPSP_PER_DEVICE="4"))
    assert r.returncode == 0, r.stdout + r.stderr
    want = [orc.sketch_fasta(t, k, m, float(np.float32(s)))[0] for t in texts]
    for i in range(len(ins)):
        assert gzip.open(tmp_path / ("all_d%02d.gz" % i), "rb").read() == want[i], i
