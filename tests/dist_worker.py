"""Worker for the world_size-2 tests of the multi-GPU comparison path.

mode cpu: gloo, CPU tensors; the key exchange and the row partition are real,
          the per-row intersections are plain numpy set algebra (no GPU here).
mode gpu: gloo for the exchange, then every rank calls spsp_compare_device on
          the GPU for its own rows (two processes share the one test GPU).
mode cpu_slots / gpu_slots: the key-partitioned exchange (SlotExchange): all-to-all
          of fixed-size slots + all-reduce of partial matrices.  On the CPU the
          slots are built and counted by a numpy stand-in (the layout and the
          collectives are what is tested); on the GPU by the real kernels.
Rank 0 checks the merged matrix against the oracle and exits non-zero on a
mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from supersampler_amd import dist as spd  # noqa: E402
from supersampler_amd import synth  # noqa: E402


def np_build_slots(minimizer, kmer_lo, counts, world, cap, n_local):
    """numpy stand-in for spsp_partition_keys_device (k <= 32): same wire format, any deterministic hash"""
    sb = sp.slot_bytes(n_local, cap, 31)
    rec_off = 16 + ((n_local + 1) & ~1) * 4
    out = np.zeros((world, sb), dtype=np.uint8)
    dest = ((kmer_lo * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(world)
    sk_of = np.repeat(np.arange(n_local), counts)
    for d in range(world):
        sel = dest == d
        hdr = out[d, :16].view(np.uint32)
        hdr[:] = [0x4C535053, n_local, int(sel.sum()), 2]
        out[d, 16:16 + 4 * n_local].view(np.uint32)[:] = np.bincount(sk_of[sel], minlength=n_local)
        rec = out[d, rec_off:].view(np.uint64).reshape(cap, 2)
        m = int(sel.sum())
        rec[:m, 0] = kmer_lo[sel]
        rec[:m, 1] = minimizer[sel].astype(np.uint64) | (sk_of[sel].astype(np.uint64) << np.uint64(32))
    return out.reshape(-1)


def np_partial_from_slots(recv, world, cap, n_local):
    """numpy stand-in for spsp_compare_slots_device: partial pair matrix of the received hash class"""
    sb = sp.slot_bytes(n_local, cap, 31)
    rec_off = 16 + ((n_local + 1) & ~1) * 4
    recv = recv.reshape(world, sb)
    n_total = world * n_local
    sets = [set() for _ in range(n_total)]
    for s in range(world):
        hdr = recv[s, :16].view(np.uint32)
        assert hdr[0] == 0x4C535053 and hdr[1] == n_local and hdr[2] <= cap
        rec = recv[s, rec_off:].view(np.uint64).reshape(cap, 2)[:int(hdr[2])]
        for lo, w in rec.tolist():
            sets[s * n_local + (w >> 32)].add((w & 0xFFFFFFFF, lo))
    part = torch.zeros((n_total, n_total), dtype=torch.int32)
    for i in range(n_total):
        for j in range(i + 1, n_total):
            part[i, j] = len(sets[i] & sets[j])
    return part


def main_slots(mode, rank, world, k, payloads, mine, counts, my_min, my_lo, per_rank):
    n_total = per_rank * world
    sk_off = np.zeros(per_rank + 1, dtype=np.uint64)
    sk_off[1:] = np.cumsum(counts)
    if mode == "gpu_slots":
        dev = torch.device("cuda", 0)
        ctx = sp.Context(0)
        ex = spd.SlotExchange(ctx, k, per_rank, int(sk_off[-1]), dev)
        d_min, d_lo = my_min.to(dev), my_lo.to(dev)
        d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):                                   # twice: buffers are reused step after step
            h = ex.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off)
            ex.end(h, d_inter)
        assert not ex.overflowed(d_inter)
        merged = d_inter.cpu()
        # an undersized exchange is noticed by every rank, and growing it fixes it
        small = spd.SlotExchange(ctx, k, per_rank, int(sk_off[-1]), dev)
        small.slot_cap = 8
        small._alloc()
        d2 = torch.zeros_like(d_inter)
        small.end(small.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off), d2)
        assert small.overflowed(d2)
        while small.overflowed(d2):
            small.grow()
            d2.zero_()
            small.end(small.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off), d2)
        assert bool((d2.cpu() == merged).all())
        # the partial matrices summed as sparse cells (what BASELINE configs[3] uses): every rank ends with the same matrix
        cx = spd.SlotExchange(ctx, k, per_rank, int(sk_off[-1]), dev, reduce="cells")
        d3 = torch.zeros_like(d_inter)
        for _ in range(2):
            cx.end(cx.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off), d3)
        assert not cx.overflowed(d3) and bool((d3.cpu() == merged).all())
        cx.slot_cap = 8
        cx._alloc()
        cx.end(cx.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off), d3)
        assert cx.overflowed(d3)
        ctx.close()
    else:
        ex = spd.SlotExchange(None, k, per_rank, int(sk_off[-1]), torch.device("cpu"))
        ex.send.copy_(torch.from_numpy(np_build_slots(my_min.numpy().view(np.uint32), my_lo.numpy().view(np.uint64),
                                                     np.asarray(counts), world, ex.slot_cap, per_rank)))
        dist.all_to_all_single(ex.recv, ex.send)
        merged = np_partial_from_slots(ex.recv.numpy(), world, ex.slot_cap, per_rank)
        dist.all_reduce(merged, op=dist.ReduceOp.SUM)
    ok = True
    if rank == 0:
        want, card, _, _ = orc.compare(payloads)
        ok = bool((merged.numpy().astype(np.uint32) == want).all()) and int(want.sum()) > 0
    return ok


def c4_shaped_sets(n, keys_per_sketch, seed=4):
    """BASELINE configs[3] shape in small: n sketches in families of 16, a member keeps ~80 % of its family's keys
    and adds a few of its own.  Returns sorted uint64 key arrays (one bucket: minimizer 7)."""
    rng = np.random.default_rng(seed)
    fams = [np.unique(rng.integers(1, 2**62, size=keys_per_sketch + keys_per_sketch // 4, dtype=np.int64)) for _ in range((n + 15) // 16)]
    out = []
    for i in range(n):
        base = fams[i // 16]
        keep = base[rng.random(len(base)) < 0.8]
        extra = rng.integers(1, 2**62, size=int(rng.integers(0, 4)), dtype=np.int64)
        out.append(np.unique(np.concatenate([keep, extra])).astype(np.uint64) if i % 37 else np.zeros(0, np.uint64))
    return out


def expected_inter(sets):
    """pair counts by an inverted index (plain numpy / Python)"""
    n = len(sets)
    holders = {}
    for i, keys in enumerate(sets):
        for key in keys.tolist():
            holders.setdefault(key, []).append(i)
    want = np.zeros((n, n), dtype=np.int32)
    for hs in holders.values():
        if len(hs) > 1:
            a = np.array(hs)
            ii, jj = np.triu_indices(len(a), 1)
            np.add.at(want, (a[ii], a[jj]), 1)
    return want


ROWS = os.environ.get("SPSP_TEST_ROWS", "block")     # row ownership after the all-gather: "block" (default of KeyExchange) or "strided"


def main_c4(mode, rank, world):
    """8 ranks, 2 048 sketches: all-gather of the keys, every rank's own rows (ROWS), strips collected on rank 0.
    cpu_c4: gloo, rows counted by numpy; nccl modes use the GPU kernels (see main_nccl)."""
    per_rank = 256 if world >= 8 else 64
    n_total = per_rank * world
    sets = c4_shaped_sets(n_total, 24)
    mine = sets[rank * per_rank:(rank + 1) * per_rank]
    counts = [len(x) for x in mine]
    my_lo = torch.from_numpy(np.concatenate(mine).view(np.int64))
    my_min = torch.full((int(sum(counts)),), 7, dtype=torch.int32)
    ex = spd.KeyExchange(counts, torch.device("cpu"), rows=ROWS)
    g = ex.exchange(my_min, my_lo)
    mine_rows = set(ex.own_rows())
    assert sorted(r for q in range(world) for r in ex.own_rows(rank=q)) == list(range(n_total))   # every row has one owner
    lo = g.kmer_lo.numpy().view(np.uint64)
    for i in (0, 1, n_total // 2, n_total - 1):                       # gathered keys == every sketch's keys in global order
        assert (lo[int(g.sk_off[i]):int(g.sk_off[i + 1])] == sets[i]).all(), i
    gathered = [lo[int(g.sk_off[i]):int(g.sk_off[i + 1])] for i in range(n_total)]
    holders = {}
    for i, keys in enumerate(gathered):
        for key in keys.tolist():
            holders.setdefault(key, []).append(i)
    local = torch.zeros((n_total, n_total), dtype=torch.int32)
    ln = local.numpy()
    for hs in holders.values():
        for a_i, a in enumerate(hs):
            if a in mine_rows:                                        # only the rows this rank owns
                for b in hs[a_i + 1:]:
                    ln[a, b] += 1
    own_rows = local.clone()
    ok = True
    want = expected_inter(sets) if rank == 0 else None
    for limit in (spd.KeyExchange.REDUCE_BYTES, 0, 1 << 40):          # as sized (strips at 2 048 sketches), strips, one reduce
        ex.REDUCE_BYTES = limit
        for _ in range(2):                                            # twice: the same tensor is collected step after step
            # no prepare call: collect_rows is self-contained.  "The comparison" writes the cells (i, j > i) of this
            # rank's rows only; everything else is whatever the previous collect (or a backend's scratch use) left --
            # poisoned on the ranks that are not the destination
            if rank != 0:
                ln[:] = 12345
            ln[sorted(mine_rows)] = own_rows.numpy()[sorted(mine_rows)]
            if rank != 0:
                ln[np.tril_indices(n_total)] = 12345                  # (the comparison never writes the diagonal or below)
            full = ex.collect_rows(local)
            if rank == 0:
                good = bool((full.numpy() == want).all()) and int(want.sum()) > 10_000
                if not good:
                    print("c4 collect mismatch: limit", limit, "cells", int((full.numpy() != want).sum()), flush=True)
                ok = ok and good
    return ok


def main_nccl(rank, world):
    """RCCL with more than one rank (needs >= 2 GPUs): both exchange forms with the real kernels, every collective
    on the stream the libspsp context runs on."""
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx = sp.Context(dev.index, stream.cuda_stream)
    k, per_rank = 31, 64
    n_total = per_rank * world
    sets = c4_shaped_sets(n_total, 300)
    mine = sets[rank * per_rank:(rank + 1) * per_rank]
    counts = [len(x) for x in mine]
    with torch.cuda.stream(stream):
        d_lo = torch.from_numpy(np.concatenate(mine).view(np.int64)).to(dev)
        d_min = torch.full((int(sum(counts)),), 7, dtype=torch.int32, device=dev)
        d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
    want = expected_inter(sets)
    ex = spd.KeyExchange(counts, dev, stream=stream, rows=ROWS)
    ra = ex.row_args()
    for _ in range(2):
        g = ex.exchange(d_min, d_lo)
        ctx.compare_device(k, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, g.sk_off, n_total, ra[0], ra[1], d_inter.data_ptr(), n_query=ra[2])
        full = ex.collect_rows(d_inter)
    stream.synchronize()
    ok = rank != 0 or bool((full.cpu().numpy() == want).all())
    sk_off = np.zeros(per_rank + 1, dtype=np.uint64)
    sk_off[1:] = np.cumsum(counts)
    for reduce in ("all", "scatter", "cells"):
        sx = spd.SlotExchange(ctx, k, per_rank, int(sk_off[-1]), dev, stream=stream, reduce=reduce)
        with torch.cuda.stream(stream):
            d2 = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
        sx.end(sx.begin(d_min.data_ptr(), d_lo.data_ptr(), None, sk_off), d2)
        stream.synchronize()
        got = d2.cpu().numpy()
        if reduce in ("all", "cells"):
            got[0, 0] = 0                                              # the overflow tally lives in an unused cell
            ok = ok and bool((got == want).all())
        else:
            blk = n_total // world
            rows = got[rank * blk:(rank + 1) * blk].copy()
            if rank == 0:
                rows[0, 0] = 0
            ok = ok and bool((rows == want[rank * blk:(rank + 1) * blk]).all())
    t = torch.tensor([1 if ok else 0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 1 else 1)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "nccl":
        main_nccl(rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "cpu_c4":
        ok = main_c4(mode, rank, world)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    k, m, s, per_rank = 31, 11, 20, 5
    gs = synth.family_genomes(9, per_rank * world, 8_000, 2, [0.0, 0.01, 0.03])
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)]
    mine = [sp.sketch_parse(payloads[spd.global_index(rank, j, per_rank)]) for j in range(per_rank)]
    counts = [len(x) for x in mine]
    my_min = torch.from_numpy(np.concatenate([x.minimizer for x in mine]).view(np.int32))
    my_lo = torch.from_numpy(np.concatenate([x.kmer_lo for x in mine]).view(np.int64))
    n_total = per_rank * world
    if mode.endswith("_slots"):
        ok = main_slots(mode, rank, world, k, payloads, mine, counts, my_min, my_lo, per_rank)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    if mode == "gpu":
        dev = torch.device("cuda", 0)
        ex = spd.KeyExchange(counts, torch.device("cpu"), rows=ROWS)
        g = ex.exchange(my_min, my_lo)
        ra = ex.row_args()
        ctx = sp.Context(0)
        d_min, d_lo = g.minimizer.to(dev), g.kmer_lo.to(dev)
        d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()   # torch fills on its own stream; the context has its own
        ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), None, g.sk_off, n_total, ra[0], ra[1], d_inter.data_ptr(), n_query=ra[2])
        torch.cuda.synchronize()
        local = d_inter.cpu()
        ctx.close()
    else:
        ex = spd.KeyExchange(counts, torch.device("cpu"), rows=ROWS)
        g = ex.exchange(my_min, my_lo)
        mn = g.minimizer.numpy().view(np.uint32)
        lo = g.kmer_lo.numpy().view(np.uint64)
        # gathered keys == every sketch's keys in global order
        for i in range(n_total):
            sk = sp.sketch_parse(payloads[i])
            a, b = int(g.sk_off[i]), int(g.sk_off[i + 1])
            assert (mn[a:b] == sk.minimizer).all() and (lo[a:b] == sk.kmer_lo).all(), i
        sets = [set(zip(mn[int(g.sk_off[i]):int(g.sk_off[i + 1])].tolist(), lo[int(g.sk_off[i]):int(g.sk_off[i + 1])].tolist()))
                for i in range(n_total)]
        local = torch.zeros((n_total, n_total), dtype=torch.int32)
        for i in spd.owned_rows(n_total, rank, world, ROWS):
            for j in range(i + 1, n_total):
                local[i, j] = len(sets[i] & sets[j])
    merged = spd.merge_rows(local, n_total, rank, world)
    ok = True
    if rank == 0:
        want, card, _, _ = orc.compare(payloads)
        ok = bool((merged.numpy().astype(np.uint32) == want).all()) and int(want.sum()) > 0
        # every pair is owned by exactly one rank
        owners = np.zeros((n_total, n_total), dtype=np.int32)
        for r in range(world):
            for i in spd.owned_rows(n_total, r, world, ROWS):
                owners[i, i + 1:] += 1
        ok = ok and bool((owners[np.triu_indices(n_total, 1)] == 1).all())
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
