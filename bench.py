#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: the minimizer scan of this rank's genomes
(spsp_scan_device) followed by this rank's share of the all-vs-all sketch
comparison (spsp_compare_device).  Workload = BASELINE.json configs[1]:
100 synthetic 5 Mbp genomes per GPU, k=31 m=11 s=1000.

N > 1: genomes are sharded by rank (no data-path collective for the scan); the
comparison all-gathers the packed sketch keys over RCCL and every rank owns the
rows i % N == rank of the (100 N) x (100 N) pair matrix.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# The pipelined step keeps several HIP streams busy at once (scan, comparison, RCCL's own stream). The runtime maps
# streams onto 4 hardware queues by default; two streams sharing a queue serialise (a kernel waiting on an event blocks
# whatever is queued behind it), which cost the multi-GPU step ~0.1 ms in rehearsal. Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402  device memory, streams, torch.distributed -- plumbing (imported before libspsp, see package docstring)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import supersampler_amd as sp  # noqa: E402
from supersampler_amd import dist as spd  # noqa: E402
from supersampler_amd import synth  # noqa: E402

K, M, S = 31, 11, 1000.0
N_GENOMES, GENOME_LEN, N_FAMILIES, MUS = 100, 5_000_000, 10, [0.001, 0.01]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    # stdout carries exactly ONE JSON line: anything native libraries print there (RCCL's version banner,
    # for one) is sent to stderr by pointing fd 1 at fd 2 until the result is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--genomes", type=int, default=N_GENOMES, help="genomes per GPU (default = BASELINE config)")
    ap.add_argument("--length", type=int, default=GENOME_LEN)
    ap.add_argument("--mode", choices=["default", "direct", "filter", "pair"], default="default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1 takes the multi-GPU code path (RCCL key all-gather, strided rows) even with one rank,
    # so that path can be rehearsed on a single-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    flags = {"default": sp.SPSP_SCAN_DEFAULT, "direct": sp.SPSP_SCAN_DIRECT_HASH, "filter": sp.SPSP_SCAN_LDS_FILTER,
             "pair": sp.SPSP_SCAN_PAIR_FILTER}[args.mode]
    p = sp.make_params(K, M, S, flags=flags)
    # ONE explicit HIP stream for everything in a step: torch copies/collectives (RCCL orders itself against the
    # current stream) and libspsp's kernels (the context is created on the same stream handle)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = sp.Context(local_rank, stream.cuda_stream)
    # Pipelined step (default): the scan of one batch and the all-vs-all of the previous batch's sketches are
    # independent, so they are queued on TWO HIP streams (two contexts, the begin/end forms of the ABI): the dense
    # pass has the GPU to itself (stream order / spsp_wait_dense), then the
    # latency-bound sparse stages of the scan and the comparison fill each other's idle CUs.  The host stays one
    # step ahead: two such stream pairs alternate, step t+1 is queued before step t is collected, so the GPU never
    # waits for the host.  BENCH_PIPELINE=0 runs the two halves back to back on one stream, one step at a time.
    pipelined = os.environ.get("BENCH_PIPELINE", "1") != "0"
    strict_order = os.environ.get("BENCH_STRICT_ORDER") == "1"
    # BENCH_FREE_RUN=1 (experiment): no ordering between the streams at all -- every scan on its own stream, the
    # comparison does not wait for the dense pass.  Highest throughput when the dense kernel leaves room on the CUs
    # (SPSP_PAIR_BLOCKS_PER_CU=1), but the dense kernel then shares the GPU and its roofline figure drops.
    free_run = os.environ.get("BENCH_FREE_RUN") == "1"

    class Slot:
        pass

    slots = []
    for i in range(2 if pipelined else 1):
        sl = Slot()
        if pipelined:
            # the scans of all steps share stream A: in order, no event waits (own workspace and results per slot)
            sl.stream_a = stream if (i == 0 or not free_run) else torch.cuda.Stream(device=dev)
            sl.scan = ctx if i == 0 else sp.Context(local_rank, sl.stream_a.cuda_stream)
            sl.stream_b = torch.cuda.Stream(device=dev)
            sl.cmp = sp.Context(local_rank, sl.stream_b.cuda_stream)
        else:
            sl.stream_a = sl.stream_b = stream
            sl.scan = sl.cmp = ctx
        slots.append(sl)
    all_ctx = []
    for sl in slots:
        for c in (sl.scan, sl.cmp):
            if c not in all_ctx:
                all_ctx.append(c)

    # ------------------------------------------------------------------ setup (untimed)
    t_setup = time.time()
    genomes = synth.family_genomes(2 + 1000 * rank, args.genomes, args.length, N_FAMILIES, MUS)
    recs = []
    for i, g in enumerate(genomes):  # 1-3 records per genome
        nr = 1 + i % 3
        cuts = [len(g) * j // nr for j in range(nr + 1)]
        recs += [g[cuts[j]:cuts[j + 1]] for j in range(nr)]
    bases, rec_off = synth.concat_records(recs)
    kmers_per_step = int(sum(max(0, len(r) - K + 1) for r in recs))
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(rec_off.view(np.int64)).to(dev)
    # sketches of this rank's genomes -> packed keys (host side of the CLIs, outside the timed region)
    sketches, payloads = [], []
    r0 = 0
    for i, g in enumerate(genomes):
        nr = 1 + i % 3
        gb, go = synth.concat_records(recs[r0:r0 + nr])
        r0 += nr
        em = ctx.scan(p, gb, go)
        payload, _ = sp.sketch_build(p, S, gb, go, em)
        payloads.append(payload)
        sketches.append(sp.sketch_parse(payload))
    my_n = np.array([len(s) for s in sketches], dtype=np.int64)
    my_min = np.concatenate([s.minimizer for s in sketches]).astype(np.uint32)
    my_lo = np.concatenate([s.kmer_lo for s in sketches]).astype(np.uint64)
    d_my_min = torch.from_numpy(my_min.view(np.int32)).to(dev)
    d_my_lo = torch.from_numpy(my_lo.view(np.int64)).to(dev)
    n_total = args.genomes * world
    my_sk_off = np.zeros(args.genomes + 1, dtype=np.uint64)
    my_sk_off[1:] = np.cumsum(my_n)
    exchange_kind = os.environ.get("BENCH_EXCHANGE", "slots") if use_dist else "none"
    if exchange_kind == "slots":      # key-partitioned: all-to-all of own keys + all-reduce of partial matrices
        for sl in slots:
            sl.exchange = spd.SlotExchange(sl.cmp, K, args.genomes, int(my_sk_off[-1]), dev)
        sk_off = np.zeros(n_total + 1, dtype=np.uint64)
        sk_off[-1] = slots[0].exchange.max_keys * world           # log line only
    elif exchange_kind == "gather":   # all-gather of every rank's keys + strided row ownership
        for sl in slots:
            sl.exchange = spd.KeyExchange(my_n, dev)
        sk_off = slots[0].exchange.sk_off
    else:
        sk_off = np.zeros(n_total + 1, dtype=np.uint64)
        sk_off[1:] = np.cumsum(my_n)
        d_all_min, d_all_lo = d_my_min, d_my_lo
    for sl in slots:
        sl.d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
    d_inter = slots[0].d_inter
    pairs_per_step = n_total * (n_total - 1) // 2
    torch.cuda.synchronize()
    if rank == 0:
        log("setup %.1fs: %d genomes x %d bp per GPU, %d records, %d k-mers/step/GPU, %d sketch keys, %d pairs"
            % (time.time() - t_setup, args.genomes, args.length, len(recs), kmers_per_step, int(sk_off[-1]), pairs_per_step))

    scan_args = (p, d_bases.data_ptr(), d_bases.numel(), d_off.data_ptr(), len(recs))

    def queue_step(sl, prev, nxt=None):
        """queue one whole step on slot sl without waiting for anything; on the GPU it starts behind slot prev.
        nxt = the slot of the following step (its key exchange is started from here, behind this dense pass)"""
        # stream A: behind the previous step's scan.  The previous comparison is all but done by then (its last
        # kernel may overlap the start of this dense pass); a full spsp_wait_stream(prev.cmp) costs more in
        # cross-queue latency than that overlap (BENCH_STRICT_ORDER=1 adds it).
        if prev is not None and strict_order:
            sl.scan.wait_stream(prev.cmp)
        sl.scan.scan_device_begin(*scan_args)
        if exchange_kind == "none":
            if not free_run:
                sl.cmp.wait_dense(sl.scan)                # the comparison starts behind A's dense pass
            sl.cmp.compare_device_begin(K, d_all_min.data_ptr(), d_all_lo.data_ptr(), None, sk_off, n_total, rank, world,
                                        sl.d_inter.data_ptr())                                             # stream B
            return
        with torch.cuda.stream(sl.stream_b):              # torch ops and RCCL order themselves against stream B
            if exchange_kind == "slots":
                if getattr(sl, "handle", None) is None:   # first steps only: later ones were started a step ahead (below)
                    sl.handle = sl.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off)
                sl.cmp.wait_dense(sl.scan)
                sl.exchange.end_queue(sl.handle, sl.d_inter)
                sl.handle = None
            else:
                g = sl.exchange.exchange(d_my_min, d_my_lo)
                sl.cmp.wait_dense(sl.scan)
                sl.cmp.compare_device_begin(K, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, sk_off, n_total, rank,
                                            world, sl.d_inter.data_ptr())

        if exchange_kind == "slots" and nxt is not None and nxt is not sl and getattr(nxt, "handle", None) is None:
            # the NEXT step's key partition + RCCL all-to-all: behind this step's dense pass (so the dense kernel keeps the
            # GPU to itself), long before the next dense pass, and behind the comparison still queued on that stream
            with torch.cuda.stream(nxt.stream_b):
                nxt.cmp.wait_dense(sl.scan)
                nxt.handle = nxt.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off)

    def collect_step(sl):
        d_out, n_out = sl.scan.scan_device_end()
        if exchange_kind == "slots":
            with torch.cuda.stream(sl.stream_b):
                sl.exchange.end_collect(sl.d_inter)       # partial pair matrix done -> RCCL all-reduce
        else:
            sl.cmp.compare_end()
        last["n_out"], last["slot"] = n_out, sl

    def run_steps(n):
        """n steps = n scans + n comparisons"""
        if not pipelined:
            sl = slots[0]
            for _ in range(n):
                # one stream: the key exchange (RCCL all-to-all) is queued first and runs behind the scan kernels
                h = sl.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off) if exchange_kind == "slots" else None
                d_out, n_out = ctx.scan_device(*scan_args)
                if exchange_kind == "slots":
                    sl.exchange.end(h, sl.d_inter)        # partial pair matrix + RCCL all-reduce
                else:
                    if exchange_kind == "gather":
                        g = sl.exchange.exchange(d_my_min, d_my_lo)
                        mn_ptr, lo_ptr = g.minimizer.data_ptr(), g.kmer_lo.data_ptr()
                    else:
                        mn_ptr, lo_ptr = d_all_min.data_ptr(), d_all_lo.data_ptr()
                    ctx.compare_device(K, mn_ptr, lo_ptr, None, sk_off, n_total, rank, world, sl.d_inter.data_ptr())
                last["n_out"], last["slot"] = n_out, sl
            return
        pending = None
        for i in range(n):
            sl = slots[i % len(slots)]
            t_a = time.perf_counter()
            queue_step(sl, pending, slots[(i + 1) % len(slots)] if i + 1 < n else None)   # step i is on the GPU's queues ...
            t_b = time.perf_counter()
            if pending is not None:
                collect_step(pending)                     # ... before the host waits for step i-1
            host["queue"] += t_b - t_a
            host["collect"] += time.perf_counter() - t_b
            pending = sl
        if pending is not None:
            collect_step(pending)

    last = {"n_out": 0, "slot": slots[0]}
    host = {"queue": 0.0, "collect": 0.0}   # host seconds spent queueing / waiting (pipelined mode)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    # the roofline needs the dense kernel's duration from HIP events in the timed region; the other brackets
    # (whole pipelines, accumulate kernel) are extra packets on the streams and are only recorded on request
    kinds = sp.TIME_ALL if os.environ.get("BENCH_STAGE_TIMING", "0") == "1" or not pipelined else sp.TIME_DENSE
    for c in all_ctx:
        c.timing_enable(True, kinds)
        c.timing_read()
    fence()
    host["queue"] = host["collect"] = 0.0
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    n_out = last["n_out"]
    d_inter = last["slot"].d_inter
    exchange = getattr(last["slot"], "exchange", None)
    tm = None
    for c in all_ctx:                                     # HIP-event logs of all contexts, summed
        t = c.timing_read()
        c.timing_enable(False)
        if tm is None:
            tm = dict(t)
        else:
            for key in tm:
                tm[key] += t[key]

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    agg = torch.tensor([float(kmers_per_step), tm["dense_ms"], tm["compare_ms"], tm["scan_ms"], tm["accumulate_ms"]],
                       dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        total_kmers_per_step = float(agg[0].item())
        dense_ms, compare_ms, scan_ms, acc_ms = (float(mx[i].item()) for i in (1, 2, 3, 4))
    else:
        total_kmers_per_step = float(kmers_per_step)
        dense_ms, compare_ms, scan_ms, acc_ms = tm["dense_ms"], tm["compare_ms"], tm["scan_ms"], tm["accumulate_ms"]
    elapsed = float(el.item())

    # sanity: the comparison produced something (family structure => shared k-mers)
    inter_nonzero = int(torch.count_nonzero(d_inter).item())
    # untimed cross-check of the two exchange forms: the key-partitioned result must equal the
    # all-gather + owned-rows result on every rank count
    exchange_check = None
    if exchange_kind == "slots":
        try:
            if exchange.overflowed(d_inter):
                exchange_check = "slot overflow"
            else:
                ge = spd.KeyExchange(my_n, dev)
                g = ge.exchange(d_my_min, d_my_lo)
                d_ref = torch.zeros_like(d_inter)
                ctx.compare_device(K, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, ge.sk_off, n_total, rank, world,
                                   d_ref.data_ptr())
                dist.all_reduce(d_ref, op=dist.ReduceOp.SUM)
                exchange_check = "equal to all-gather form" if bool(torch.equal(d_ref, d_inter)) else "MISMATCH"
        except Exception as e:  # noqa: BLE001 -- the check must not take the bench line down
            exchange_check = "check failed: %r" % (e,)

    if rank == 0:
        value = total_kmers_per_step * args.steps / elapsed
        dense_avg_ms = dense_ms / max(1, tm["dense_launches"])
        achieved = (d_bases.numel() / 1e9) / (dense_avg_ms / 1e3) if dense_avg_ms > 0 else 0.0  # 1 B per position (ASCII)
        compare_avg_ms = compare_ms / max(1, tm["compare_calls"])
        out = {
            "metric": "k-mers hashed/s (sketch) + sketch-pairs/s (all-vs-all)",
            "value": value,
            "unit": "k-mers hashed/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic %d bp genomes per GPU (10 families, mu 0.001/0.01), "
                                   "k=31 m=11 s=1000, scan + all-vs-all; inputs resident in HBM" % (args.genomes, args.length),
                       "k": K, "m": M, "s": S, "genomes_per_gpu": args.genomes, "genome_len": args.length,
                       "sketches_total": n_total, "scan_mode": args.mode,
                       "parallelism": {"none": "single GPU",
                                       "slots": "genomes sharded by rank; sketch keys partitioned by hash, RCCL all-to-all behind "
                                                "the scan, per-rank partial pair matrix, RCCL all-reduce",
                                       "gather": "genomes sharded by rank; pair-matrix rows i%N==rank after RCCL all-gather of "
                                                 "sketch keys"}[exchange_kind],
                       "exchange_check": exchange_check,
                       "step": ("scan(batch t) on stream A || all-vs-all(sketches of batch t-1) on stream B; two stream pairs "
                                "alternate and the host queues step t+1 before collecting step t" if pipelined
                                else "scan then all-vs-all on one stream, one step at a time")},
            # with the comparison's own bracket: pairs / its pipeline time; otherwise the sustained rate of the whole step
            "sketch_pairs_per_s": (pairs_per_step / (compare_avg_ms / 1e3) if compare_avg_ms > 0
                                   else pairs_per_step * args.steps / elapsed),
            "sketch_pairs_per_s_basis": "compare pipeline (HIP events)" if compare_avg_ms > 0 else "whole step (scan + all-vs-all)",
            "stage_ms": {"scan_pipeline": scan_ms / tm["scan_calls"] if tm["scan_calls"] else None, "dense_kernel": dense_avg_ms,
                         "compare_pipeline": compare_avg_ms if tm["compare_calls"] else None,
                         "accumulate_kernel": acc_ms / tm["accumulate_launches"] if tm["accumulate_launches"] else None},
            "superkmers_per_step": int(n_out), "inter_nonzero": inter_nonzero,
            "host_ms_per_step": {"queueing": host["queue"] * 1e3 / args.steps, "waiting": host["collect"] * 1e3 / args.steps},
            "roofline": {"kernel": "k_dense_pair (2-bit pack + LDS pair-table test of every m-mer position; XXH64 on survivors)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args),
                         "algorithmic_bytes_per_launch": int(d_bases.numel())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(recs, payloads, p, int(n_out), d_inter)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args):
    """HBM bytes per launch of the dense kernel from the PMC counters (FETCH_SIZE/WRITE_SIZE, separate
    rocprofv3 --pmc passes of this same command, gfx950 correction applied: profiles/r01_d_pmc_hbm_traffic.json).
    Counters cannot be read from inside this process, so the figure is looked up for the matching workload."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_d_pmc_hbm_traffic.json")))
        w = d["workload"]
        if (w["genomes"], w["genome_len"], w["k"], w["m"], w["s"], w["scan_mode"]) != (args.genomes, args.length, K, M, S, args.mode):
            return None
        return d["kernels"]["k_dense_pair"]["hbm_bytes_per_launch_corrected"]
    except Exception:
        return None


def cpu_baseline(recs, payloads, p, gpu_superkmers, d_inter):
    """The reference-algorithm CPU restatement (oracle/) timed on this host, rank 0, N=1 only.
    Checker code used as a reported baseline -- never on the product path.  While it is at it, it checks the
    timed GPU step against it: super-k-mers emitted over the same records, pair matrix over the same sketches."""
    from oracle import oracle_py as orc
    budget, spent, kmers, used, emitted = 12.0, 0.0, 0, 0, 0
    for r in recs:  # bounded sample: whole records until ~12 s of single-thread CPU work
        if spent >= budget:
            break
        b, o = synth.concat_records([r])
        sec, km, nem = orc.scan_timed(K, M, p.threshold, b, o)
        spent += sec; kmers += km; used += 1; emitted += nem
    # the reference parallelises over FILES with OpenMP (SubSampler.cpp:771): same thing with host threads
    # (ctypes releases the GIL), whole records as work items, bounded the same way
    import concurrent.futures as cf
    import os as _os
    import time as _time
    cores = max(1, len(_os.sched_getaffinity(0)))
    items = [synth.concat_records([r]) for r in recs[:used]]
    t0 = _time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda bo: orc.scan_timed(K, M, p.threshold, bo[0], bo[1]), items))
    wall = _time.perf_counter() - t0
    all_cores = {"value": sum(r[1] for r in res) / wall if wall > 0 else None, "cores": cores,
                 "sample": "same records, one record per task over %d host threads, %.2f s wall" % (cores, wall)}
    n = len(payloads)
    want_inter, _, csec = orc.compare(payloads, timed=True)  # the reference comparator is single-threaded too
    parity = {"pair_matrix": bool((d_inter.cpu().numpy().astype(np.uint32) == want_inter).all())}
    if used == len(recs):
        parity["superkmers_per_step"] = bool(emitted == gpu_superkmers)
    return {"parity_vs_oracle": parity, "value": kmers / spent if spent > 0 else None, "unit": "k-mers hashed/s", "cores": 1, "kind": "port",
            "sample": "oracle scan loop (SubSampler.cpp:357-455 restated), single thread, first %d of %d records "
                      "of the same workload, %.1f s" % (used, len(recs), spent),
            "all_cores": all_cores,
            "sketch_pairs_per_s": (n * (n - 1) // 2) / csec if csec > 0 else None,
            "pairs_sample": "oracle compare_sketches (Comparator.cpp:39-287 restated) over the same %d sketches, "
                            "single thread, %.2f s" % (n, csec)}


if __name__ == "__main__":
    main()
