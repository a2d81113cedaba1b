"""Worker for the world_size-2 tests of the multi-GPU comparison path.

mode cpu: gloo, CPU tensors; the key exchange and the row partition are real,
          the per-row intersections are plain numpy set algebra (no GPU here).
mode gpu: gloo for the exchange, then every rank calls spsp_compare_device on
          the GPU for its own rows (two processes share the one test GPU).
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


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, m, s, per_rank = 31, 11, 20, 5
    gs = synth.family_genomes(9, per_rank * world, 8_000, 2, [0.0, 0.01, 0.03])
    payloads = [orc.sketch_fasta(synth.to_fasta(g, "g%d" % i), k, m, s)[0] for i, g in enumerate(gs)]
    mine = [sp.sketch_parse(payloads[spd.global_index(rank, j, per_rank)]) for j in range(per_rank)]
    counts = [len(x) for x in mine]
    my_min = torch.from_numpy(np.concatenate([x.minimizer for x in mine]).view(np.int32))
    my_lo = torch.from_numpy(np.concatenate([x.kmer_lo for x in mine]).view(np.int64))
    n_total = per_rank * world
    if mode == "gpu":
        dev = torch.device("cuda", 0)
        ex = spd.KeyExchange(counts, torch.device("cpu"))
        g = ex.exchange(my_min, my_lo)
        ctx = sp.Context(0)
        d_min, d_lo = g.minimizer.to(dev), g.kmer_lo.to(dev)
        d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()   # torch fills on its own stream; the context has its own
        ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), None, g.sk_off, n_total, rank, world, d_inter.data_ptr())
        torch.cuda.synchronize()
        local = d_inter.cpu()
        ctx.close()
    else:
        ex = spd.KeyExchange(counts, torch.device("cpu"))
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
        for i in spd.owned_rows(n_total, rank, world):
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
            for i in spd.owned_rows(n_total, r, world):
                owners[i, i + 1:] += 1
        ok = ok and bool((owners[np.triu_indices(n_total, 1)] == 1).all())
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
