"""Multi-GPU plumbing for the all-vs-all comparison.

One process per GPU.  Sketching shards genomes by rank with no collective.  The
comparison has ONE exchange step: an all-gather of the packed sketch keys, after
which every rank holds all sketches and owns the pair-matrix rows
`i % world == rank` (round-robin rows balance the upper triangle).  Works with
backend "nccl" (= RCCL over xGMI, device tensors) and "gloo" (CPU tensors, used
by the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def owned_rows(n_total, rank, world):
    return list(range(rank, n_total, world))


def global_index(rank, local_idx, per_rank):
    """sketch `local_idx` of `rank` -> index in the gathered order (rank-major)."""
    return rank * per_rank + local_idx


class GatheredSketches:
    """All ranks' keys, rank-major: minimizer/kmer_lo/(kmer_hi) tensors + sk_off (numpy uint64)."""

    def __init__(self, minimizer, kmer_lo, kmer_hi, sk_off):
        self.minimizer, self.kmer_lo, self.kmer_hi, self.sk_off = minimizer, kmer_lo, kmer_hi, sk_off

    @property
    def n(self):
        return len(self.sk_off) - 1


class KeyExchange:
    """Pre-sized all-gather of sketch keys.  `counts` = keys per local sketch
    (every rank must hold the same number of sketches)."""

    def __init__(self, counts, device, use_hi=False, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.use_hi = use_hi
        counts = np.asarray(counts, dtype=np.int64)
        gathered = [torch.zeros(len(counts), dtype=torch.int64, device=device) for _ in range(self.world)]
        dist.all_gather(gathered, torch.from_numpy(counts).to(device), group=group)
        all_counts = torch.stack(gathered).cpu().numpy()           # [world, sketches per rank]
        self.per_rank = all_counts.sum(axis=1)
        self.pad = int(self.per_rank.max())
        self.sk_off = np.zeros(all_counts.size + 1, dtype=np.uint64)
        self.sk_off[1:] = np.cumsum(all_counts.reshape(-1))
        self.starts = np.concatenate([[0], np.cumsum(self.per_rank)]).astype(np.int64)
        total = int(self.per_rank.sum())
        mk = lambda dt, n: torch.zeros(n, dtype=dt, device=device)
        self._pad_min, self._pad_lo = mk(torch.int32, self.pad), mk(torch.int64, self.pad)
        self._g_min, self._g_lo = mk(torch.int32, self.world * self.pad), mk(torch.int64, self.world * self.pad)
        self.all_min, self.all_lo = mk(torch.int32, total), mk(torch.int64, total)
        if use_hi:
            self._pad_hi, self._g_hi, self.all_hi = mk(torch.int64, self.pad), mk(torch.int64, self.world * self.pad), mk(torch.int64, total)
        else:
            self.all_hi = None

    def _gather_one(self, g, padded, mine, out):
        padded[:mine.numel()] = mine
        if dist.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(padded) for _ in range(self.world)]
            dist.all_gather(parts, padded, group=self.group)
            g.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(g, padded, group=self.group)
        for r in range(self.world):
            out[self.starts[r]:self.starts[r + 1]] = g[r * self.pad:r * self.pad + int(self.per_rank[r])]

    def exchange(self, my_min, my_lo, my_hi=None):
        """int32/int64 tensors (bit patterns of the uint32/uint64 keys) of this rank -> GatheredSketches."""
        self._gather_one(self._g_min, self._pad_min, my_min, self.all_min)
        self._gather_one(self._g_lo, self._pad_lo, my_lo, self.all_lo)
        if self.use_hi:
            self._gather_one(self._g_hi, self._pad_hi, my_hi, self.all_hi)
        return GatheredSketches(self.all_min, self.all_lo, self.all_hi, self.sk_off)


def merge_rows(inter_local, n_total, rank, world, group=None):
    """Sum the per-rank pair matrices (each rank filled only its own rows) on every rank."""
    t = inter_local.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
