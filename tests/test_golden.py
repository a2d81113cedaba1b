"""Committed fixtures (tests/golden/golden.json, made by tests/golden/make_golden.py from the oracle after its
first-principle pins pass): the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import json
import os

import numpy as np
import pytest

import supersampler_amd as sp
from oracle import oracle_py as orc

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))
FASTAS = {k: v.encode("latin1") for k, v in G["fastas"].items()}


@pytest.mark.parametrize("case", G["cases"], ids=lambda c: "k%d_m%d_s%g_a%d" % (c["k"], c["m"], c["s"], c["abundance"]))
def test_oracle_reproduces_golden(case):
    k, m, s, ab = case["k"], case["m"], case["s"], case["abundance"]
    assert orc.threshold(k, m, s) == case["threshold"] == sp.threshold(k, m, s)
    payloads = []
    for name, text in FASTAS.items():
        g = case["genomes"][name]
        bases, offs = orc.clean_fasta(text)
        assert [int(x) for x in offs] == g["rec_off"]
        em, _ = orc.scan(k, m, case["threshold"], bases, offs)
        assert [[int(e[f]) for f in ("rec", "minimizer", "start", "len", "rev")] for e in em] == g["stream"]
        payload, st = orc.sketch_fasta(text, k, m, s, ab)
        assert payload.hex() == g["payload_hex"] and st["selected_kmer_number"] == g["selected_kmer_number"]
        # host side of the product (no GPU): builder and reader on the golden stream
        got, _ = sp.sketch_build(sp.make_params(k, m, s, ab), s, bases, offs, em)
        assert got.hex() == g["payload_hex"]
        payloads.append(payload)
    inter, card, _, _ = orc.compare(payloads)
    assert inter.tolist() == case["inter"] and [int(c) for c in card] == case["card"]
    names = [n + ".gz" for n in FASTAS]
    assert sp.csv(True, names, inter, card).decode() == case["jaccard_csv"]
    assert sp.csv(False, names, inter, card).decode() == case["containment_csv"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G["cases"], ids=lambda c: "k%d_m%d_s%g_a%d" % (c["k"], c["m"], c["s"], c["abundance"]))
def test_hip_path_matches_golden(case):
    k, m, s, ab = case["k"], case["m"], case["s"], case["abundance"]
    ctx = sp.Context(0)
    sketches = []
    for name, text in FASTAS.items():
        g = case["genomes"][name]
        p = sp.make_params(k, m, s, ab)
        bases, offs = sp.clean_fasta(text)
        em = ctx.scan(p, bases, offs)
        assert [[int(e[f]) for f in ("rec", "minimizer", "start", "len", "rev")] for e in em] == g["stream"]
        payload, _ = ctx.sketch_text(text, k, m, s, ab)           # GPU ingest + scan + gather
        assert payload.hex() == g["payload_hex"]
        sketches.append(sp.sketch_parse(payload))
    inter, card = ctx.compare(sketches)
    assert inter.tolist() == case["inter"] and [int(c) for c in card] == case["card"]
    ctx.close()
