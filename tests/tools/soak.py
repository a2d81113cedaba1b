#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: scan (all dense variants) and comparison against the oracle for as long
as the time budget allows.  Prints the first mismatch with the configuration that produced it.
usage: python tests/tools/soak.py [seconds=300] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (before libspsp)
import supersampler_amd as sp  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from supersampler_amd import synth  # noqa: E402

COMP = np.zeros(256, np.uint8)
COMP[[65, 67, 71, 84]] = [84, 71, 67, 65]


def random_genome(rng):
    base = synth.random_genome(rng, int(rng.integers(50, 120_000)))
    parts = [base]
    for _ in range(int(rng.integers(0, 5))):
        kind = rng.integers(0, 6)
        if kind == 0:
            parts.append(synth.mutate(rng, base, float(rng.choice([0.001, 0.02, 0.2])))[: int(rng.integers(1, len(base) + 1))])
        elif kind == 1:
            parts.append(COMP[base[::-1]][: int(rng.integers(1, len(base) + 1))])
        elif kind == 2:
            parts.append(np.full(int(rng.integers(1, 400)), int(rng.choice([65, 67, 71, 84])), np.uint8))
        elif kind == 3:
            parts.append(np.tile(synth.random_genome(rng, int(rng.integers(1, 40))), int(rng.integers(2, 80))))
        elif kind == 4:
            a = int(rng.integers(0, len(base)))
            parts.append(base[a:a + int(rng.integers(1, 3000))])
        else:
            parts.append(synth.random_genome(rng, int(rng.integers(1, 70))))
    rng.shuffle(parts)
    genome = np.concatenate(parts)
    cuts = sorted(set([0, len(genome)] + [int(x) for x in rng.integers(0, len(genome) + 1, size=int(rng.integers(0, 8)))]))
    return [genome[a:b] for a, b in zip(cuts[:-1], cuts[1:])]


def random_fasta_text(rng):
    """FASTA text with everything getLineFasta / clean_dna have to cope with"""
    out = []
    repeat = synth.random_genome(rng, int(rng.integers(20, 400)))
    for _ in range(int(rng.integers(1, 120))):
        r = rng.random()
        if r < 0.08:
            out.append(b">" + bytes(rng.integers(32, 127, size=int(rng.integers(0, 40))).astype(np.uint8)))
        elif r < 0.10:
            out.append(b"")
        elif r < 0.12:
            out.append(b"\xff" + bytes(synth.random_genome(rng, int(rng.integers(0, 30)))))
        else:
            L = int(rng.integers(1, 300))
            line = (repeat[:L] if rng.random() < 0.3 and L <= len(repeat) else synth.random_genome(rng, L)).copy()
            if rng.random() < 0.3:
                line[rng.integers(0, L, size=max(1, L // 10))] = ord("N")
            if rng.random() < 0.3:
                line = np.frombuffer(bytes(line).lower(), dtype=np.uint8).copy()
            if rng.random() < 0.1:
                line = np.concatenate([line, np.frombuffer(b"\r", np.uint8)])
            out.append(bytes(line))
    return b"\n".join(out) + (b"\n" if rng.random() < 0.5 else b"")


def mark(msg):
    """what is about to run, kept in a file: a GPU fault kills the process without a Python traceback"""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "soak_last.txt"), "w") as f:
        f.write(msg + "\n")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = sp.Context(0)
    t0 = time.time()
    n_scan = n_cmp = n_sk = n_ex = n_keys = n_files = n_rows = 0
    modes = [sp.SPSP_SCAN_DEFAULT, sp.SPSP_SCAN_DIRECT_HASH, sp.SPSP_SCAN_LDS_FILTER, sp.SPSP_SCAN_PAIR_FILTER, sp.SPSP_SCAN_PAIR_FILTER, sp.SPSP_SCAN_BLOOM_FILTER]
    t_report = t0
    while time.time() - t0 < budget:
        if time.time() - t_report > 45:      # a long run must keep talking (the GPU pool kills silent commands)
            print("  ... %d scans, %d sketch payloads, %d comparisons, %d exchanges, %d key extractions, %d files after %.0f s" % (n_scan, n_sk, n_cmp, n_ex, n_keys, n_files, time.time() - t0), flush=True)
            t_report = time.time()
        m = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
        k = int(rng.choice([x for x in range(max(m, 5) | 1, 64, 2)]))
        s = float(rng.choice([1.0, 1.2, 2, 3, 7, 20, 100, 1000, 5000]))
        if n_scan % 60 == 59:    # now and then a large input: many tiles, several scan segments, many chunk sums
            big = int(rng.integers(17_000_000, 60_000_000))
            cuts = sorted(set([0, big] + [int(x) for x in rng.integers(0, big, size=int(rng.integers(0, 30)))]))
            recs = [synth.random_genome(rng, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
            if s < 3:
                s = float(rng.choice([3, 10, 100, 1000]))     # keep the oracle's stream (and the run) small
        else:
            recs = random_genome(rng)
        bases, offs = synth.concat_records(recs)
        mode = int(rng.choice(modes))
        p = sp.make_params(k, m, s, flags=mode)
        mark("scan %d: k=%d m=%d s=%g mode=%d bases=%d records=%d" % (n_scan + 1, k, m, s, mode, len(bases), len(recs)))
        if n_scan % 3 == 1 and len(bases) >= k:      # the same records as 2-bit words (SPSP_SCAN_PACKED_INPUT), device form
            d_b = torch.from_numpy(bases).cuda()
            d_o = torch.from_numpy(offs.view(np.int64)).cuda()
            torch.cuda.synchronize()
            pp = sp.make_params(k, m, s, flags=mode | sp.SPSP_SCAN_PACKED_INPUT)
            d_out, n_out = ctx.scan_device(pp, ctx.pack_bases_device(d_b.data_ptr(), d_b.numel()), d_b.numel(), d_o.data_ptr(), len(recs))
            got = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE) if n_out else np.zeros(0, sp.SUPERKMER_DTYPE)
        else:
            got = ctx.scan(p, bases, offs)
        want, _ = orc.scan(k, m, p.threshold, bases, offs)
        ok = len(got) == len(want) and all((np.asarray(got[f]) == np.asarray(want[f])).all() for f in ("rec", "minimizer", "start", "len", "rev"))
        if not ok:
            print("SCAN MISMATCH k=%d m=%d s=%g mode=%d n=%d records=%d got=%d want=%d" % (k, m, s, mode, len(bases), len(recs), len(got), len(want)))
            np.save(os.path.join(ROOT, "gpurun_out", "soak_bases.npy"), bases)
            np.save(os.path.join(ROOT, "gpurun_out", "soak_offs.npy"), offs)
            sys.exit(1)
        n_scan += 1
        if n_scan % 5 == 0:      # whole sketch payload through the GPU ingest path, bytes vs the oracle
            text = random_fasta_text(rng)
            ab = int(rng.choice([1, 1, 2, 3]))
            s2 = float(np.float32(max(s, 1.0) if rng.random() < 0.7 else float(rng.choice([1.0, 2.0, 4.0]))))   # the CLI parses -s with stof
            mark("sketch_text after scan %d: k=%d m=%d s=%g ab=%d mode=%d text=%d bytes" % (n_scan, k, m, s2, ab, mode, len(text)))
            got_pl, got_st = ctx.sketch_text(text, k, m, s2, ab, mode)
            want_pl, want_st = orc.sketch_fasta(text, k, m, float(np.float32(s2)), ab)
            if got_pl != want_pl:
                print("SKETCH MISMATCH k=%d m=%d s=%g ab=%d mode=%d text=%d bytes" % (k, m, s2, ab, mode, len(text)))
                open(os.path.join(ROOT, "gpurun_out", "soak_text.fa"), "wb").write(text)
                sys.exit(1)
            n_sk += 1
        if n_scan % 40 == 0:     # key-partitioned exchange with simulated ranks + CSV text, vs set algebra / the oracle printer
            import test_exchange as tx
            world, n_local = int(rng.integers(1, 9)), int(rng.integers(1, 12))
            use_hi = bool(rng.integers(0, 2))
            sets = tx.make_sets(rng, world * n_local, use_hi, universe_size=int(rng.integers(60, 3000)))
            per_rank = [sets[r * n_local:(r + 1) * n_local] for r in range(world)]
            biggest = max(1, max(sum(len(x) for x in pr) for pr in per_rank))
            cap = biggest if world == 1 else int(biggest / world * 1.5) + 64
            mark("exchange after scan %d: world=%d n_local=%d use_hi=%d" % (n_scan, world, n_local, use_hi))
            try:
                _, total = tx.exchange_and_compare(ctx, 63 if use_hi else 31, per_rank, cap)
            except sp.SpspError as e:
                if e.code != sp.ERR_OVERFLOW:
                    raise
                _, total = tx.exchange_and_compare(ctx, 63 if use_hi else 31, per_rank, biggest + 64)
            nt = world * n_local
            want_t = np.array([[len(sets[i] & sets[j]) if j > i else 0 for j in range(nt)] for i in range(nt)])
            if not (total == want_t).all():
                print("EXCHANGE MISMATCH world=%d n_local=%d use_hi=%d" % (world, n_local, use_hi))
                sys.exit(1)
            # the same sketches through the row-partitioned device form (a rank's block / strided rows, query limits):
            # owned cells equal set algebra, every other cell untouched
            if nt >= 2:
                mn_a, lo_a, hi_a, off_a = tx.rank_arrays(sets)
                if int(off_a[-1]) > 0:
                    dev = torch.device("cuda", 0)
                    d_mn = torch.from_numpy(mn_a.view(np.int32)).to(dev)
                    d_lo = torch.from_numpy(lo_a.view(np.int64)).to(dev)
                    d_hi = torch.from_numpy(hi_a.view(np.int64)).to(dev)
                    for _ in range(3):
                        first = int(rng.integers(0, nt))
                        stride = int(rng.choice([1, 1, 2, 3, world]))
                        limit = int(rng.integers(first + 1, nt + 1))
                        d_part = torch.full((nt, nt), -1, dtype=torch.int32, device=dev)
                        torch.cuda.synchronize()
                        mark("row partition after scan %d: nt=%d first=%d stride=%d limit=%d use_hi=%d keys=%d" % (n_scan, nt, first, stride, limit, use_hi, int(off_a[-1])))
                        ctx.compare_device(63 if use_hi else 31, d_mn.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr() if use_hi else None, off_a, nt,
                                           first, stride, d_part.data_ptr(), n_query=limit)
                        torch.cuda.synchronize()
                        got_p = d_part.cpu().numpy()
                        rows = np.arange(first, limit, stride)
                        own = np.zeros((nt, nt), bool)
                        own[rows] = np.triu(np.ones((nt, nt), bool), 1)[rows]
                        if not ((got_p[~own] == -1).all() and (got_p[own] == want_t[own]).all()):
                            print("ROW PARTITION MISMATCH nt=%d first=%d stride=%d limit=%d use_hi=%d" % (nt, first, stride, limit, use_hi))
                            sys.exit(1)
                        n_rows += 1
            cardx = np.array([len(x) for x in sets], dtype=np.uint64)
            names = ["s%d.gz" % i for i in range(nt)]
            prec, thr, nqx = int(rng.integers(0, 13)), float(rng.choice([0.0, 0.0, 0.01, 0.3])), int(rng.integers(1, nt + 1))
            for jac in (True, False):
                if sp.csv(jac, names, total.astype(np.uint32), cardx, nqx, prec, thr) != orc.csv(jac, names, total.astype(np.uint32), cardx, nqx, prec, thr):
                    print("CSV MISMATCH jac=%d prec=%d thr=%g nq=%d" % (jac, prec, thr, nqx))
                    sys.exit(1)
            n_ex += 1
        if n_scan % 6 == 3:      # scan of several genomes at once -> the comparator's keys on the device, sorted and unordered, vs the oracle's file path
            ng = int(rng.integers(1, 7))
            groups = [random_genome(rng) for _ in range(ng)]
            if rng.random() < 0.5 and ng > 1:
                groups[1] = groups[0] + groups[1][:1]             # shared records: keys seen in two genomes, some twice in one
            grecs = [r for g in groups for r in g]
            first = np.cumsum([0] + [len(g) for g in groups]).astype(np.uint32)
            gb, go = synth.concat_records(grecs)
            ab = int(rng.choice([1, 1, 2, 3]))
            pk = sp.make_params(k, m, s, abundance=ab, flags=mode)
            d_b = torch.from_numpy(np.concatenate([gb, np.zeros(64, np.uint8)])).cuda()
            d_o = torch.from_numpy(go.view(np.int64)).cuda()
            torch.cuda.synchronize()
            mark("key extraction after scan %d: k=%d m=%d s=%g ab=%d mode=%d genomes=%d bases=%d" % (n_scan, k, m, s, ab, mode, ng, len(gb)))
            d_sk, n_skm = ctx.scan_device(pk, d_b.data_ptr(), len(gb), d_o.data_ptr(), len(grecs))
            wants = []
            for g in groups:
                text = b"".join(synth.to_fasta(r, "r%d" % j) for j, r in enumerate(g)) if g else b">e\n"
                wants.append(orc.sketch_keys(orc.sketch_fasta(text, k, m, s, ab)[0]))
            for unordered in (False, True):
                try:
                    d_mn, d_lo, d_hi, koff = ctx.sketch_keys_device(pk, d_b.data_ptr(), len(gb), d_o.data_ptr(), d_sk, n_skm, first, unordered=unordered)
                except sp.SpspError:
                    raise                                         # (no size is refused any more: genomes beyond the LDS forms take the table in HBM)
                tot = int(koff[-1])
                mn, lo = ctx.to_host(d_mn, tot, np.uint32), ctx.to_host(d_lo, tot, np.uint64)
                hi = ctx.to_host(d_hi, tot, np.uint64) if k > 32 else np.zeros(tot, np.uint64)
                for gi, (_, _, w_mn, w_lo, w_hi) in enumerate(wants):
                    x, y = int(koff[gi]), int(koff[gi + 1])
                    if k == m and y == x and len(w_mn) == 1:
                        continue                                  # the merge's phantom key of an empty sketch (spsp_sketch_chain_host): not in any file
                    got_k = list(zip(mn[x:y].tolist(), hi[x:y].tolist(), lo[x:y].tolist()))
                    if unordered:
                        got_k = sorted(got_k)
                    if got_k != list(zip(w_mn.tolist(), w_hi.tolist(), w_lo.tolist())):
                        print("KEYS MISMATCH k=%d m=%d s=%g ab=%d mode=%d unordered=%d genome=%d of %d got=%d want=%d" % (k, m, s, ab, mode, unordered, gi, ng, y - x, len(w_mn)))
                        np.save(os.path.join(ROOT, "gpurun_out", "soak_keys_bases.npy"), gb)
                        np.save(os.path.join(ROOT, "gpurun_out", "soak_keys_offs.npy"), go)
                        np.save(os.path.join(ROOT, "gpurun_out", "soak_keys_first.npy"), first)
                        sys.exit(1)
            n_keys += 1
        if n_scan % 25 == 12:    # FASTA files -> sketch files through the batched pipeline, payload bytes and print_stat counters vs the oracle
            import gzip
            import shutil
            import tempfile
            tmp = tempfile.mkdtemp(prefix="soak_files_")
            try:
                nf = int(rng.integers(1, 12))
                texts = [random_fasta_text(rng) if rng.random() < 0.6 else synth.to_fasta(np.concatenate(random_genome(rng) or [np.zeros(0, np.uint8)]), "g", n_records=int(rng.integers(1, 4)))
                         for _ in range(nf)]
                ins = []
                for i, t in enumerate(texts):
                    pth = os.path.join(tmp, "f%d.fa" % i) + (".gz" if i % 3 == 1 else "")
                    open(pth, "wb").write(gzip.compress(t, 1) if i % 3 == 1 else t)
                    ins.append(pth)
                outs = [os.path.join(tmp, "o%d.gz" % i) for i in range(nf)]
                s2 = float(np.float32(max(s, 1.0)))
                ab = int(rng.choice([1, 1, 1, 2]))
                mark("files after scan %d: k=%d m=%d s=%g ab=%d mode=%d files=%d" % (n_scan, k, m, s2, ab, mode, nf))
                res, _, _ = sp.sketch_files(ins, outs, k, m, s2, abundance=ab, threads=int(rng.integers(1, 9)), flags=mode | sp.SPSP_SCAN_STATS)
                for i, (rc, st, err) in enumerate(res):
                    want_pl, want_st = orc.sketch_fasta(texts[i], k, m, s2, ab)
                    if rc != 0 or sp.read_file(outs[i]) != want_pl or st["total_superkmer_number"] != want_st["total_superkmer_number"] or st["nb_mmer_selected"] != want_st["nb_mmer_selected"]:
                        print("FILES MISMATCH k=%d m=%d s=%g ab=%d mode=%d file %d of %d rc=%d err=%s" % (k, m, s2, ab, mode, i, nf, rc, err))
                        open(os.path.join(ROOT, "gpurun_out", "soak_file.fa"), "wb").write(texts[i])
                        sys.exit(1)
                n_files += nf
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        if n_scan % 30 == 7:     # a collection of one species: hundreds of sketches sharing most keys -> parts overflow, spill, bit columns
            n = int(rng.integers(130, 760))                     # (from 512 rows on: rows dealt to the XCDs in runs, in min-hash order)
            use_hi = bool(rng.integers(0, 2))
            U = int(rng.integers(400, 3600))
            share = rng.choice([0.95, 0.6, 0.3, 0.08, 0.01], size=U)
            B = rng.random((n, U)) < share[None, :]
            if rng.random() < 0.5:
                B[int(rng.integers(0, n))] = False
            anc = np.stack([rng.integers(0, 2**22, U), rng.integers(0, 2**62, U), rng.integers(0, 2**62, U) if use_hi else np.zeros(U, np.int64)], 1)
            own_n = int(rng.integers(0, 200))
            sks, sizes = [], []
            for i in range(n):
                own = np.stack([rng.integers(0, 2**22, own_n), rng.integers(0, 2**62, own_n), rng.integers(0, 2**62, own_n) if use_hi else np.zeros(own_n, np.int64)], 1)
                keys = np.concatenate([anc[B[i]], own]) if B[i].any() else anc[:0]
                keys = keys[np.lexsort((keys[:, 1], keys[:, 2], keys[:, 0]))]
                sizes.append(len(keys))
                sks.append(sp.Sketch(63 if use_hi else 31, 11, keys[:, 0].astype(np.uint32), keys[:, 1].astype(np.uint64), keys[:, 2].astype(np.uint64)))
            nq = n if rng.random() < 0.7 else int(rng.integers(1, n + 1))
            want_sp = np.triu(B.astype(np.int64) @ B.astype(np.int64).T, 1)
            mark("one-species compare after scan %d: n=%d nq=%d use_hi=%d U=%d keys=%d" % (n_scan, n, nq, use_hi, U, sum(sizes)))
            inter, card = ctx.compare(sks, n_query=nq)
            if not ((np.triu(inter.astype(np.int64), 1)[:nq] == want_sp[:nq]).all() and [int(c) for c in card] == sizes):
                print("ONE-SPECIES COMPARE MISMATCH n=%d nq=%d use_hi=%d U=%d" % (n, nq, use_hi, U))
                np.save(os.path.join(ROOT, "gpurun_out", "soak_species_B.npy"), B)
                sys.exit(1)
            n_cmp += 1
        if n_scan % 8 == 0:      # a comparison problem from sketches of related genomes
            n = int(rng.integers(2, 40))
            anc = synth.random_genome(rng, int(rng.integers(2000, 30_000)))
            payloads = []
            for i in range(n):
                g = anc if rng.random() < 0.2 else synth.mutate(rng, anc, float(rng.choice([0.0, 0.005, 0.05])))
                if rng.random() < 0.1:
                    g = synth.random_genome(rng, int(rng.integers(10, 3000)))
                payloads.append(orc.sketch_fasta(synth.to_fasta(g, "g%d" % i, n_records=int(rng.integers(1, 4))), k, m, max(s, 1.0))[0])
            nq = n if rng.random() < 0.6 else int(rng.integers(1, n + 1))
            want_inter, want_card, _, _ = orc.compare(payloads, n_query=nq)
            sketches = sp.sketches_from_payloads(payloads)     # incl. the merge's first-read rule for empty sketches
            mark("compare after scan %d: k=%d m=%d s=%g n=%d nq=%d keys=%d" % (n_scan, k, m, s, n, nq, sum(len(x) for x in sketches)))
            inter, card = ctx.compare(sketches, n_query=nq)
            if not ((inter[:nq] == want_inter[:nq]).all() and [int(c) for c in card] == [int(c) for c in want_card]):   # printed rows
                print("COMPARE MISMATCH k=%d m=%d s=%g n=%d nq=%d" % (k, m, s, n, nq))
                bad = np.argwhere(inter[:nq] != want_inter[:nq])
                print("cells:", [(int(a), int(b), int(inter[a, b]), int(want_inter[a, b])) for a, b in bad[:10]])
                print("card got/want:", [(i, int(card[i]), int(want_card[i])) for i in range(n) if int(card[i]) != int(want_card[i])][:10])
                import pickle
                pickle.dump({"payloads": payloads, "nq": nq, "k": k, "m": m}, open(os.path.join(ROOT, "gpurun_out", "soak_cmp.pkl"), "wb"))
                sys.exit(1)
            n_cmp += 1
            # the same problem as sparse cells straight from the row sums (or through the dense matrix: small problems, query rows),
            # and -- now and then -- from sketch FILES over two or three contexts on this device (the key-partitioned split)
            tot_keys = sum(len(x) for x in sketches)
            if tot_keys:
                dev = torch.device("cuda", 0)
                use_hi = k > 32
                d_mn = torch.from_numpy(np.concatenate([x.minimizer for x in sketches]).view(np.int32)).to(dev)
                d_lo = torch.from_numpy(np.concatenate([x.kmer_lo for x in sketches]).view(np.int64)).to(dev)
                d_hi = torch.from_numpy(np.concatenate([x.kmer_hi for x in sketches]).view(np.int64)).to(dev)
                off = np.zeros(n + 1, np.uint64)
                off[1:] = np.cumsum([len(x) for x in sketches])
                scratch = torch.zeros((n, n), dtype=torch.int32, device=dev)
                cells = torch.zeros(n * n + 16, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                mark("cells after scan %d: n=%d nq=%d keys=%d" % (n_scan, n, nq, tot_keys))
                cnt = ctx.compare_cells_device(k, d_mn.data_ptr(), d_lo.data_ptr(), d_hi.data_ptr() if use_hi else None, off, n, scratch.data_ptr(),
                                               cells.data_ptr(), cells.numel(), n_query=nq)
                cw = cells[:cnt].cpu().numpy().view(np.uint64)
                back = np.zeros((n, n), np.uint32)
                back[(cw >> np.uint64(48)).astype(np.int64), ((cw >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)] = (cw & np.uint64(0xffffffff)).astype(np.uint32)
                if not (back[:nq] == np.triu(want_inter, 1)[:nq]).all() or cnt != int(np.count_nonzero(np.triu(want_inter, 1)[:nq])):
                    print("CELLS MISMATCH k=%d m=%d s=%g n=%d nq=%d" % (k, m, s, n, nq))
                    sys.exit(1)
            if n_cmp % 4 == 0:
                import gzip
                import shutil
                import tempfile
                tmp = tempfile.mkdtemp(prefix="soak_multi_")
                try:
                    paths = []
                    for i, pl in enumerate(payloads):
                        pth = os.path.join(tmp, "s%d.gz" % i)
                        sp.write_gz(pth, pl, 1)
                        paths.append(pth)
                    devs = [0] * int(rng.integers(2, 4))
                    mark("compare_files_multi after scan %d: n=%d nq=%d contexts=%d" % (n_scan, n, nq, len(devs)))
                    sp.compare_files_multi(devs, paths, os.path.join(tmp, "m"), n_query=nq)
                    for jac, suf in ((True, "_jaccard.csv.gz"), (False, "_containment.csv.gz")):
                        if gzip.open(os.path.join(tmp, "m") + suf, "rb").read() != orc.csv(jac, paths, want_inter, want_card, nq, 6, 0.0):
                            print("MULTI-CONTEXT CSV MISMATCH k=%d m=%d s=%g n=%d nq=%d contexts=%d" % (k, m, s, n, nq, len(devs)))
                            sys.exit(1)
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
    sp.sketch_files_release()
    print("soak ok: %d scans, %d sketch payloads, %d comparisons, %d exchanges + CSV pairs, %d row-partitioned comparisons, %d key extractions, %d files through the pipeline in %.0f s"
          % (n_scan, n_sk, n_cmp, n_ex, n_rows, n_keys, n_files, time.time() - t0))


if __name__ == "__main__":
    main()
