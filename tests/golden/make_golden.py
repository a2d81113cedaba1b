#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle (oracle/spsp_oracle.cpp).

The reference ships no fixtures and cannot be run, so these are
NOT reference outputs: they freeze the oracle's answers -- after its first-principle
pins in tests/test_oracle.py pass -- on three tiny FASTAs, so that a later change to
the oracle or to the HIP path that alters any byte is caught (last row).
Inputs are stored verbatim; outputs are hex / integer lists.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bruteforce as bf  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

rng = np.random.default_rng(20261003)
anc = bf.random_dna(rng, 1400)
fastas = {
    "tiny_a": bf.fasta([anc[:900], anc[900:]], names=["a1 first", "a2"]),
    "tiny_b": bf.fasta([bf.mutate(rng, anc, 0.02)], width=60) + b">short\nACGTNNACG\n",
    "tiny_c": (">c lower+N\n" + anc[200:700].lower() + "\nNNNN\n" + anc[700:1100] + "\r\n>rep\n" + "ACGTTGCAAG" * 12 + "\n" + "A" * 80 + "\n").encode(),
}
CONFIGS = [(31, 11, 20.0, 1), (21, 11, 4.0, 1), (63, 15, 6.0, 1), (31, 11, 3.0, 2)]
out = {"note": "oracle-generated regression fixtures; see make_golden.py", "fastas": {k: v.decode("latin1") for k, v in fastas.items()}, "cases": []}
for (k, m, s, ab) in CONFIGS:
    case = {"k": k, "m": m, "s": s, "abundance": ab, "threshold": orc.threshold(k, m, s), "genomes": {}}
    payloads, names = [], []
    for name, text in fastas.items():
        bases, offs = orc.clean_fasta(text)
        em, st = orc.scan(k, m, case["threshold"], bases, offs)
        payload, sst = orc.sketch_fasta(text, k, m, s, ab)
        case["genomes"][name] = {
            "rec_off": [int(x) for x in offs],
            "stream": [[int(e[f]) for f in ("rec", "minimizer", "start", "len", "rev")] for e in em],
            "payload_hex": payload.hex(),
            "selected_kmer_number": sst["selected_kmer_number"],
        }
        payloads.append(payload); names.append(name + ".gz")
    inter, card, _, _ = orc.compare(payloads)
    case["inter"] = inter.tolist(); case["card"] = [int(c) for c in card]
    case["jaccard_csv"] = orc.csv(True, names, inter, card).decode()
    case["containment_csv"] = orc.csv(False, names, inter, card).decode()
    out["cases"].append(case)
json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=0)
print("wrote", os.path.join(HERE, "golden.json"), os.path.getsize(os.path.join(HERE, "golden.json")), "bytes")
