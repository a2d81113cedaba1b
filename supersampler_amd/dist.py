"""Multi-GPU plumbing for the all-vs-all comparison.

One process per GPU.  Sketching shards genomes by rank with no collective.  The
comparison has ONE key exchange step, in two forms:

* `KeyExchange` (default of bench.py, north_star's form): an all-gather of the
  packed keys, after which every rank holds all sketches and computes a share
  of the pair-matrix rows -- by default the BLOCK of rows of the sketches it
  scanned itself (`rows="block"`: its dictionary is then built from its own
  keys, and the other ranks' keys only pass a filter: spsp_compare_device with
  row_first = first row, row_stride = 1, n_query = end of the block), or the
  rows `i % world == rank` (`rows="strided"`); `collect_rows` returns the
  strips to one rank (each cell crosses the fabric once).
* `SlotExchange`: keys are partitioned by hash, every rank sends each of its
  keys once (all-to-all of fixed-size slots), counts its hash class for ALL
  pairs and the partial N x N matrices are summed (reduce-scatter by row
  blocks, or all-reduce).  Per-rank key traffic stays O(own keys), but every
  rank holds and reduces a full N x N partial matrix: DESIGN.md 5 prices both.

Both work with backend "nccl" (= RCCL over xGMI, device tensors) and "gloo"
(CPU tensors, used by the CPU tests).  With "nccl", pass `stream=` (the
torch.cuda.Stream the libspsp context was created on): RCCL orders itself
against torch's CURRENT stream, so every collective and every torch copy of
these classes runs inside `torch.cuda.stream(stream)` -- the same stream the
context's kernels are queued on.
"""
import contextlib
import math

import numpy as np
import torch
import torch.distributed as dist


def owned_rows(n_total, rank, world, rows="strided"):
    """the pair-matrix rows of `rank`: every world-th row, or (rows="block") its n_total / world consecutive rows"""
    if rows == "block":
        per = n_total // world
        return list(range(rank * per, n_total if rank == world - 1 else (rank + 1) * per))
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

    def __init__(self, counts, device, use_hi=False, group=None, stream=None, rows="block"):
        if rows not in ("block", "strided"):
            raise ValueError("rows must be 'block' or 'strided'")
        self.rows = rows
        self.group = group
        self.stream = stream
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
        # ONE all-gather per step for both key arrays (the step is host-bound with RCCL: a collective costs ~17 us of host
        # time): every rank's padded block is [kmer_lo: pad x 8 B | minimizer: pad x 4 B] in one byte buffer; the
        # gathered blocks are compacted by ONE gather kernel per array (index_select over the int64 / int32 view)
        self.pad += self.pad & 1                                         # blocks stay 8-byte aligned
        blk = self.pad * 12
        self._pad_buf = mk(torch.uint8, blk)
        self._pad_lo = self._pad_buf[:self.pad * 8].view(torch.int64)
        self._pad_min = self._pad_buf[self.pad * 8:].view(torch.int32)
        self._g_buf = mk(torch.uint8, self.world * blk)
        self._g_as_lo, self._g_as_min = self._g_buf.view(torch.int64), self._g_buf.view(torch.int32)
        idx_lo = np.concatenate([np.arange(int(self.per_rank[r]), dtype=np.int64) + r * (blk // 8) for r in range(self.world)]) if total else np.zeros(0, np.int64)
        idx_mn = np.concatenate([np.arange(int(self.per_rank[r]), dtype=np.int64) + r * (blk // 4) + self.pad * 2 for r in range(self.world)]) if total else np.zeros(0, np.int64)
        small = self.world * blk // 4 < 2**31
        # ONE gather kernel for both arrays: the gathered bytes as int32 words -> [kmer_lo: 2 words per key | minimizer]
        idx_all = np.concatenate([np.stack([2 * idx_lo, 2 * idx_lo + 1], axis=1).reshape(-1), idx_mn])
        self._idx_all = torch.from_numpy(idx_all.astype(np.int32 if small else np.int64)).to(device)
        self._all_buf = mk(torch.int32, 3 * total)
        self.all_lo, self.all_min = self._all_buf[:2 * total].view(torch.int64), self._all_buf[2 * total:]
        self.n_mine = int(self.per_rank[self.rank])
        if use_hi:
            idx = np.concatenate([np.arange(int(self.per_rank[r]), dtype=np.int64) + r * self.pad for r in range(self.world)]) if total else np.zeros(0, np.int64)
            self._idx = torch.from_numpy(idx.astype(np.int32 if self.world * self.pad < 2**31 else np.int64)).to(device)
            self._pad_hi, self._g_hi, self.all_hi = mk(torch.int64, self.pad), mk(torch.int64, self.world * self.pad), mk(torch.int64, total)
        else:
            self.all_hi = None

    def row_args(self, n_total=None):
        """(row_first, row_stride, n_query) of this rank's rows for spsp_compare_device over the gathered sketches"""
        n = self.n_total if n_total is None else n_total
        rows = self.own_rows(n)
        if self.rows == "block":
            return (rows[0] if rows else 0), 1, (rows[-1] + 1 if rows else 0)
        return self.rank, self.world, n

    @property
    def n_total(self):
        return len(self.sk_off) - 1

    def own_rows(self, n_total=None, rank=None):
        return owned_rows(self.n_total if n_total is None else n_total, self.rank if rank is None else rank, self.world, self.rows)

    def _all_gather(self, g, padded):
        if dist.get_backend(self.group) == "gloo":
            parts = [torch.empty_like(padded) for _ in range(self.world)]
            dist.all_gather(parts, padded, group=self.group)
            g.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(g, padded, group=self.group)

    def _on_stream(self):
        # (entering a stream context costs ~7 us of host time: skipped when the caller is on the stream already)
        if self.stream is None or torch.cuda.current_stream(self.stream.device) == self.stream:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.stream)

    @property
    def send_min(self):
        """this rank's slice of the send buffer (int32): keys written here need no copy (`exchange()` without arguments)"""
        return self._pad_min[:self.n_mine]

    @property
    def send_lo(self):
        return self._pad_lo[:self.n_mine]

    @property
    def send_hi(self):
        return self._pad_hi[:self.n_mine] if self.use_hi else None

    def exchange(self, my_min=None, my_lo=None, my_hi=None):
        """int32/int64 tensors (bit patterns of the uint32/uint64 keys) of this rank -> GatheredSketches.
        Without arguments the keys are taken from where `send_min` / `send_lo` / `send_hi` point (for a caller that
        assembles its keys there; bench.py passes them and pays the two copies)."""
        with self._on_stream():
            if my_min is not None:
                self._pad_min[:my_min.numel()] = my_min
                self._pad_lo[:my_lo.numel()] = my_lo
            self._all_gather(self._g_buf, self._pad_buf)
            torch.index_select(self._g_as_min, 0, self._idx_all, out=self._all_buf)
            if self.use_hi and my_hi is not None:
                self._pad_hi[:my_hi.numel()] = my_hi
            if self.use_hi:
                self._all_gather(self._g_hi, self._pad_hi)
                torch.index_select(self._g_hi, 0, self._idx, out=self.all_hi)
        return GatheredSketches(self.all_min, self.all_lo, self.all_hi, self.sk_off)

    def _collect_plan(self, inter_local, dst):
        """index tensors and buffers of collect_rows for one matrix shape (built once: the step is host-bound)"""
        n = inter_local.shape[0]
        key = (n, dst, inter_local.dtype, str(inter_local.device))
        plan = getattr(self, "_plans", {}).get(key)
        if plan is None:
            dev = inter_local.device
            rows = max(len(self.own_rows(n, r)) for r in range(self.world))   # rows per rank, padded
            own = torch.tensor(self.own_rows(n), dtype=torch.int64, device=dev)
            plan = {"rows": rows, "own": own, "mine": torch.zeros((rows, n), dtype=inter_local.dtype, device=dev)}
            # cells (i, j > i) of the owned rows: what the comparison wrote; the rest of a strip is not the sender's to send
            plan["upper"] = (torch.arange(n, device=dev)[None, :] > own[:, None]).to(inter_local.dtype)
            if self.rank == dst:
                plan["all"] = torch.zeros((self.world, rows, n), dtype=inter_local.dtype, device=dev)
                src, dstr = [], []
                for r in range(self.world):
                    if r == dst:
                        continue
                    theirs = self.own_rows(n, r)
                    src += [r * rows + j for j in range(len(theirs))]
                    dstr += theirs
                plan["src"] = torch.tensor(src, dtype=torch.int64, device=dev)
                plan["dst"] = torch.tensor(dstr, dtype=torch.int64, device=dev)
            self._plans = getattr(self, "_plans", {})
            self._plans[key] = plan
        return plan

    # matrices up to this many bytes are collected by ONE reduce instead of strips (see collect_rows)
    REDUCE_BYTES = 16 << 20

    def rows_by_reduce(self, inter_local):
        return inter_local.numel() * inter_local.element_size() <= self.REDUCE_BYTES

    def prepare_rows(self, inter_local, dst=0):
        """Kept for callers of earlier versions: does nothing.  `collect_rows` itself clears every cell this rank does
        not own before the sum is taken (a caller that forgot this call, or made it after queueing the comparison, used
        to get silently doubled counts in the reduce form and correct ones in the strip form)."""
        return None

    def _own_mask(self, inter_local):
        """1 in the cells this rank's comparison writes -- (i, j > i) of its own rows -- 0 elsewhere"""
        n = inter_local.shape[0]
        key = ("mask", n, inter_local.dtype, str(inter_local.device))
        self._plans = getattr(self, "_plans", {})
        m = self._plans.get(key)
        if m is None:
            i = torch.arange(n, device=inter_local.device)
            mine = torch.zeros(n, dtype=torch.bool, device=inter_local.device)
            mine[torch.tensor(self.own_rows(n), dtype=torch.int64, device=inter_local.device)] = True
            m = (mine[:, None] & (i[None, :] > i[:, None])).to(inter_local.dtype)
            self._plans[key] = m
        return m

    def collect_rows(self, inter_local, dst=0):
        """strips -> one rank: every rank sends the rows it owns (`own_rows`) of its n x n
        int32 matrix; on `dst` they are put in place in `inter_local`, which then holds the whole matrix.  Each
        cell crosses the fabric once (an all-reduce of the n x n matrices moves 2 (world-1)/world of ALL of them).
        Small matrices (<= REDUCE_BYTES: every bench.py world size; 800 x 800 cells = 2.6 MB) take a shortcut that
        costs the host two calls instead of four: everything outside the cells this rank's comparison wrote is cleared
        here (one multiply by a cached 0/1 mask: on `dst` the tensor still holds the previous step's collected rows,
        elsewhere a backend may have used it as scratch -- gloo's reduce does -- and spsp_compare_device leaves such
        cells untouched), then ONE reduce (sum) onto `dst` puts every strip in place.  Self-contained: no call before
        the comparison is needed, and the result does not depend on which of the two forms the matrix size selects."""
        n = inter_local.shape[0]
        if self.rows_by_reduce(inter_local) and not (dist.get_backend(self.group) == "gloo" and inter_local.is_cuda):
            if self.world > 1:
                with self._on_stream():
                    inter_local.mul_(self._own_mask(inter_local))
                    dist.reduce(inter_local, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
            return inter_local
        with self._on_stream():
            if dist.get_backend(self.group) == "gloo" and inter_local.is_cuda:
                # (CPU-side collective in the tests: through host copies)
                rows = max(len(self.own_rows(n, r)) for r in range(self.world))
                mine = torch.zeros((rows, n), dtype=inter_local.dtype, device=inter_local.device)
                idx = torch.tensor(self.own_rows(n), dtype=torch.int64, device=inter_local.device)
                own = inter_local[idx]
                mine[:own.shape[0]] = own * (torch.arange(n, device=own.device)[None, :] > idx[:, None]).to(own.dtype)
                torch.cuda.synchronize()
                mine = mine.cpu()
                parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == dst else None
                dist.gather(mine, parts, dst=dst, group=self.group)
                if self.rank == dst:
                    for r in range(self.world):
                        theirs = self.own_rows(n, r)
                        if r != dst:
                            inter_local[theirs] = parts[r][:len(theirs)].to(inter_local.device)
                return inter_local
            P = self._collect_plan(inter_local, dst)
            k_own = P["own"].numel()
            if k_own:
                torch.index_select(inter_local, 0, P["own"], out=P["mine"][:k_own])      # own rows, one kernel
                P["mine"][:k_own].mul_(P["upper"])
            parts = list(P["all"].unbind(0)) if self.rank == dst else None
            dist.gather(P["mine"], parts, dst=dst, group=self.group)
            if self.rank == dst and P["src"].numel():
                inter_local.index_copy_(0, P["dst"], P["all"].view(-1, n).index_select(0, P["src"]))
        return inter_local


def merge_rows(inter_local, n_total, rank, world, group=None):
    """Sum the per-rank pair matrices (each rank filled only its own rows) on every rank."""
    t = inter_local.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class SlotExchange:
    """Key-partitioned exchange (include/spsp.h "multi-GPU exchange").

    begin(): spsp_partition_keys_device -> all-to-all of one fixed-size slot per peer (asynchronous with
             nccl, so the caller can queue other GPU work -- e.g. the next scan -- behind it);
    end():   wait, spsp_compare_slots_device on the received slots (the partition-form comparison over every sketch's
             keys of this rank's hash class: 1 / world of the one-GPU work on every rank), then the partial matrices are
             summed -- reduce="cells" (what a large matrix wants): every rank turns its partial matrix into its non-zero
             cells (spsp_matrix_cells_device: 95 000 of 5 x 10^7 at BASELINE configs[3]), the cells are all-gathered and
             every rank adds the others' into its own matrix; "all" / "scatter": all-reduce / reduce-scatter of the dense
             matrices (small matrices inside a pipelined step; the receiver's spsp_compare_slots_device_begin reads the slot
             headers back before it queues anything -- one host wait per step, since round 4: the flat comparison wants the
             sketch offsets on the host).
    Every rank must hold the same number of sketches `n_local`; global sketch id = rank * n_local + local id.
    """

    def __init__(self, ctx, k, n_local, n_keys_local, device, slack=1.25, group=None, stream=None, reduce="all"):
        self.ctx, self.k, self.n_local, self.device, self.group = ctx, k, n_local, device, group
        self.stream = stream      # torch.cuda.Stream wrapping ctx's stream (nccl): see the module docstring
        self.reduce = reduce      # "all": all-reduce of the partial matrices; "scatter": reduce-scatter by row blocks
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        # all ranks must agree on the slot size: size it for the largest rank
        t = torch.tensor([int(n_keys_local), int(n_local)], dtype=torch.int64, device=self._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        if int(t[1].item()) != n_local:
            raise ValueError("every rank must hold the same number of sketches")
        self.max_keys = int(t[0].item())
        self.slot_cap = self.max_keys if self.world == 1 else int(math.ceil(self.max_keys / self.world * slack)) + 1024
        self._alloc()

    def _comm_device(self):
        return self.device if self.backend == "nccl" else torch.device("cpu")

    def _alloc(self):
        from . import slot_bytes
        self.slot_bytes = slot_bytes(self.n_local, self.slot_cap, self.k)
        self.send = torch.zeros(self.world * self.slot_bytes, dtype=torch.uint8, device=self.device)
        self.recv = torch.zeros(self.world * self.slot_bytes, dtype=torch.uint8, device=self.device)

    @property
    def n_total(self):
        return self.n_local * self.world

    def grow(self):
        """After an overflow (every rank calls this together): double the slots."""
        self.slot_cap *= 2
        self._alloc()

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def begin(self, d_min, d_lo, d_hi, sk_off):
        """device pointers of this rank's concatenated keys + host sk_off (n_local+1) -> handle for end()"""
        self.ctx.partition_keys_device(self.k, d_min, d_lo, d_hi, sk_off, self.n_local, self.world, self.slot_cap,
                                       self.send.data_ptr())
        if self.backend == "nccl":
            with self._on_stream():
                return dist.all_to_all_single(self.recv, self.send, group=self.group, async_op=True)
        if self.send.is_cuda:   # gloo with GPU compute (tests): stage through the host
            torch.cuda.synchronize()
            h_send, h_recv = self.send.cpu(), torch.empty(self.send.numel(), dtype=torch.uint8)
            dist.all_to_all_single(h_recv, h_send, group=self.group)
            self.recv.copy_(h_recv)
            torch.cuda.synchronize()
        else:
            dist.all_to_all_single(self.recv, self.send, group=self.group)
        return None

    def end(self, handle, d_inter):
        """d_inter: zero-initialised int32 [n_total, n_total] device tensor -> summed over ranks in place.
        Cell [0, 0] (unused by the pair matrix) counts the ranks whose slots overflowed: see overflowed()."""
        self.end_queue(handle, d_inter)
        self.end_collect(d_inter)

    def end_queue(self, handle, d_inter):
        """first half of end(): wait for the slots, queue the partial comparison (returns without waiting)"""
        if handle is not None:
            with self._on_stream():
                handle.wait()
        from . import SpspError, ERR_OVERFLOW
        self._overflow = False
        self._d_inter = d_inter
        if self.reduce == "cells":      # the comparison and its sparse result are one synchronous call (spsp_compare_slots_cells_device): end_collect
            return
        try:
            self.ctx.compare_slots_device_begin(self.k, self.recv.data_ptr(), self.world, self.n_local, self.slot_cap,
                                                d_inter.data_ptr())
        except SpspError as e:          # the receiver reads the slot headers first: a slot that overflowed at a sender is refused here
            if e.code != ERR_OVERFLOW:
                raise
            self._overflow = True

    def _sum_cells(self, d_inter):
        """reduce="cells": the partial matrices meet as packed non-zero cells (i << 48 | j << 32 | count), all-gathered;
        every rank ends with the summed matrix.  Cell [0, 0] = number of ranks whose slots overflowed."""
        from . import SpspError, ERR_OVERFLOW
        n = d_inter.shape[0]
        cnt = 0
        cap = getattr(self, "_cells_cap", max(1 << 16, 32 * n))
        while True:
            if getattr(self, "_cells", None) is None or self._cells.numel() < cap:
                self._cells = torch.zeros(cap, dtype=torch.int64, device=self.device)
                torch.cuda.synchronize()
            try:
                # this rank's partial matrix: its non-zero cells straight from the row sums (d_inter is scratch during the call)
                cnt = self.ctx.compare_slots_cells_device(self.k, self.recv.data_ptr(), self.world, self.n_local, self.slot_cap, d_inter.data_ptr(),
                                                          self._cells.data_ptr(), cap)
                break
            except SpspError as e:
                if e.code != ERR_OVERFLOW:
                    raise
                if "non-zero cells" in str(e):      # more cells than room: again with the room it asks for (a dense partial matrix
                    cap = max(2 * cap, int(getattr(e, "cells_needed", 0)) + 1024)   # -- one species -- would double eight times)
                    continue
                self._overflow = True               # a slot overflowed at its sender
                break
        self._cells_cap = cap
        # (torch work on the context's stream, like the library calls around it: the caller's current stream may be another)
        with self._on_stream():
            d_inter.zero_()
        if self.stream is None:
            torch.cuda.synchronize()
        if cnt:
            self.ctx.matrix_add_cells_device(d_inter.data_ptr(), n, self._cells.data_ptr(), cnt)
        staged = self.backend != "nccl" and d_inter.is_cuda             # gloo with GPU compute (tests): through the host
        comm = torch.device("cpu") if staged else self.device
        with self._on_stream():
            mine = torch.tensor([cnt, 1 if self._overflow else 0], dtype=torch.int64, device=comm)
            every = torch.zeros(2 * self.world, dtype=torch.int64, device=comm)
            dist.all_gather_into_tensor(every, mine, group=self.group)
            every = every.view(self.world, 2).cpu()
            most = int(every[:, 0].max().item())
            bad = int(every[:, 1].sum().item())
            if most:
                send = torch.zeros(most, dtype=torch.int64, device=comm)
                if cnt:
                    send[:cnt] = self._cells[:cnt].to(comm)
                got = torch.zeros(self.world * most, dtype=torch.int64, device=comm)
                dist.all_gather_into_tensor(got, send, group=self.group)
                got = got.to(self.device).view(self.world, most)
                torch.cuda.synchronize()
                for r in range(self.world):
                    c = int(every[r, 0].item())
                    if r != self.rank and c:
                        self.ctx.matrix_add_cells_device(d_inter.data_ptr(), n, got[r].data_ptr(), c)
            if bad:
                d_inter.view(-1)[0] = bad

    def end_collect(self, d_inter):
        """second half of end(): wait for the partial matrix, sum the partial matrices over the ranks"""
        if self.reduce == "cells":
            return self._sum_cells(d_inter)
        if not self._overflow:
            self.ctx.compare_end()
        if self._overflow:
            d_inter.view(-1)[0] += 1
        if self.backend == "nccl" or not d_inter.is_cuda:
            with self._on_stream():
                n = d_inter.shape[0]
                if self.reduce == "scatter" and n % self.world == 0 and self.backend == "nccl":
                    # every rank ends with the summed rows [rank n/world, (rank+1) n/world) in place; (world-1)/world
                    # of ONE matrix per rank on the wire instead of twice that
                    blk = n // self.world
                    dist.reduce_scatter_tensor(d_inter[self.rank * blk:(self.rank + 1) * blk], d_inter, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.all_reduce(d_inter, op=dist.ReduceOp.SUM, group=self.group)
        else:
            torch.cuda.synchronize()
            h = d_inter.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            d_inter.copy_(h)
            torch.cuda.synchronize()

    @staticmethod
    def overflowed(d_inter):
        """Host check (synchronises): did any rank report a slot overflow in the last end()?"""
        return int(d_inter.view(-1)[0].item()) != 0
