#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: the minimizer scan of this rank's genomes
(spsp_scan_device), the comparator's keys of an earlier scan made on the device
from its super-k-mer stream (spsp_sketch_keys_device: what the comparator would
read from the sketch files of those genomes) and this rank's share of the
all-vs-all comparison of earlier keys (spsp_compare_device) -- nothing from the
setup inside the timed loop (single GPU; with N > 1 the comparison still works
on the keys of the setup's sketches, see below).  Consecutive steps scan
different batches (a ring of 3 x 500 MB).  Workload = BASELINE.json configs[1]:
100 synthetic 5 Mbp genomes per GPU, k=31 m=11 s=1000.

N > 1: genomes are sharded by rank (no data-path collective for the scan); the
comparison all-gathers the packed sketch keys over RCCL and every rank owns the
rows of its own 100 sketches in the (100 N) x (100 N) pair matrix (SURVEY.md 8e;
its dictionary holds its own keys, the other ranks' keys pass a filter).  The
all-gather is pre-sized from the setup's key counts, so with N > 1 the keys that
travel are the setup's (the per-step key extraction would need the counts of
every rank on the host first: one more collective and a host wait per step).

Besides the timed step, rank 0 of a single-GPU run measures (untimed, after the
step loop): the same closed step over 200 steps when the timed region was
shorter (`long_run`), the open step of rounds 1-2 (`open_loop`), the comparator alone at
BASELINE configs[2] and configs[3] scale (`compare`, `compare_c4`), the scan at
the configs[4] shape (`scan_c5`) and the file-to-file drivers end to end
(`end_to_end`), each beside the oracle.

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

# The pipelined step keeps several HIP streams busy at once (scan, comparison, RCCL's own stream). The runtime maps
# streams onto 4 hardware queues by default; two streams sharing a queue serialise (a kernel waiting on an event blocks
# whatever is queued behind it), which cost the multi-GPU step ~0.1 ms in rehearsal. Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def launch_ranks_if_needed():
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the driver's plain command): this
    process starts the N ranks itself -- `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as
    a CHILD process, before anything here has touched the GPU (nothing but the standard library is imported yet; never
    an exec) -- forwards rank 0's one JSON line and leaves with the children's status."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv[1:]):
        if a == "--gpus" and i + 2 < len(sys.argv):
            n = int(sys.argv[i + 2])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:                          # a free port of this host (two benches side by side must not meet)
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")              # (one node: a gloo rehearsal must not wait for the host's name to resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: starting the ranks as child processes: %s" % (n, " ".join(cmd[1:])),
          file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)  # the children write to this process's stdout / stderr


if __name__ == "__main__":
    launch_ranks_if_needed()

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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# Knobs of the step's schedule and of analysis runs (tools/exp/*.sh): read through xenv(), which returns the
# default unless the run was started with --experiment -- a line made with --experiment is never a valid measurement
EXPERIMENT_KNOBS = ("BENCH_PIPELINE", "BENCH_SCHEDULE", "BENCH_SMALL_CUS", "BENCH_DENSE_FIRST_CU", "BENCH_DENSE_STREAMS", "BENCH_SCAN_BUFFERS", "BENCH_SLOTS", "BENCH_SMALL_STREAMS", "BENCH_BATCHES", "BENCH_SIM_WORLD", "BENCH_ROWS", "BENCH_EXCHANGE", "BENCH_SLOT_REDUCE", "BENCH_DEVICE_KEYS", "BENCH_DEBUG_SKIP_COMPARE", "BENCH_KEYS_STREAM", "BENCH_KEYS_CUS", "BENCH_KEYS_STREAMS", "BENCH_DEBUG_NOISE_KERNELS", "BENCH_COLLECT_DEPTH", "BENCH_STAGE_TIMING", "BENCH_TIMING_EVERY")
XENV = {"on": False, "ignored": []}


def xenv(name, default=None):
    if XENV["on"]:
        return os.environ.get(name, default)
    if name in os.environ and name not in XENV["ignored"]:
        XENV["ignored"].append(name)
    return default


def env_report():
    """Every BENCH_* / SPSP_* variable that is set goes into the line (`config.env`): a stale variable on the box must not
    change the headline silently.  Variables that take work out of the timed region, replace the library or the transport,
    or exist for analysis only make the line INVALID as a measurement (`valid`: false, with the reasons)."""
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("BENCH_", "SPSP_"))}
    why = []
    if XENV["on"]:
        why.append("--experiment: the schedule / analysis knobs are honoured")
    for k, v in env.items():
        if k in XENV["ignored"]:
            continue                                          # set, but this run was not started with --experiment: not read
        if "_DEBUG_" in k or "_EXP_" in k:
            why.append("%s is an analysis / test hook" % k)
        elif k == "BENCH_DEVICE_KEYS" and v == "0":
            why.append("BENCH_DEVICE_KEYS=0 leaves the key extraction out of the step")
        elif k == "BENCH_SIM_WORLD" and v not in ("", "1"):
            why.append("BENCH_SIM_WORLD simulates one rank's share, no exchange")
        elif k == "BENCH_SHARE_GPU" and v == "1":
            why.append("BENCH_SHARE_GPU=1 puts every rank on one device")
        elif k == "BENCH_BACKEND" and v != "nccl":
            why.append("BENCH_BACKEND=%s is not RCCL" % v)
        elif k == "SPSP_LIB":
            why.append("SPSP_LIB replaces the in-tree libspsp.so")
        elif k == "BENCH_BATCHES" and v in ("0", "1"):
            why.append("BENCH_BATCHES=%s lets consecutive steps read the same batch (Infinity Cache)" % v)
    return env, why


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
    ap.add_argument("--no-extras", action="store_true", help="skip the `compare` (config 3) and `end_to_end` objects")
    ap.add_argument("--experiment", action="store_true",
                    help="honour the schedule / analysis knobs (EXPERIMENT_KNOBS below); without it they are ignored, so that the "
                         "headline path is the code as written with its defaults, whatever the environment holds")
    args = ap.parse_args()
    XENV.update(on=args.experiment, ignored=[])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: run `python bench.py --gpus N` (it starts its ranks itself) or launch "
                         "N ranks with torch.distributed.run" % (args.gpus, world))
    # BENCH_SHARE_GPU=1 + BENCH_BACKEND=gloo (rehearsal on a one-GPU box): every rank computes on cuda:0 and the collectives
    # go through gloo -- the N > 1 code path of this file end to end, without RCCL's transport; the timing means nothing
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        if backend == "nccl" and world > 1:
            raise SystemExit("BENCH_SHARE_GPU=1 needs BENCH_BACKEND=gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1 takes the multi-GPU code path (RCCL key all-gather, row ownership) even with one rank,
    # so that path can be rehearsed on a single-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    flags = {"default": sp.SPSP_SCAN_DEFAULT, "direct": sp.SPSP_SCAN_DIRECT_HASH, "filter": sp.SPSP_SCAN_LDS_FILTER,
             "pair": sp.SPSP_SCAN_PAIR_FILTER}[args.mode]
    p = sp.make_params(K, M, S, flags=flags)
    # ONE explicit HIP stream for the torch side (copies, collectives: RCCL orders itself against the current stream)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = sp.Context(local_rank, stream.cuda_stream)
    # Pipelined step (default): two slots alternate.  All scans are queued in order on ONE stream (dense pass, sparse
    # stages, next dense pass ...); each slot's comparison has a stream of its own and starts behind its dense pass
    # (spsp_wait_dense), so it runs beside the sparse stages and the next dense pass -- the dense kernel occupies one
    # workgroup per CU and leaves the other half of every CU to it.  The host queues step t+1 before it collects
    # step t (scan_end / compare_end wait on per-job events, not on the streams).
    # BENCH_PIPELINE=0 runs the two halves back to back on one stream, one step at a time.
    pipelined = xenv("BENCH_PIPELINE", "1") != "0"
    # BENCH_SCHEDULE: "single" = as above (0.158-0.163 ms); "streams" = every slot's scan on a stream of its own, its dense
    # pass behind the previous slot's by an event (0.167 ms); "tail" = scans on one stream, their sparse stages on
    # spsp_scan_tail_stream streams (0.18 ms)
    # "partition" = the chip is split (spsp_stream_create_cus): every dense pass runs on a stream that owns all but
    # BENCH_SMALL_CUS compute units, the sparse stages of every scan and every comparison on streams that own the rest
    # -- the dense passes run back to back and nothing that shares a CU with them slows them down
    schedule = xenv("BENCH_SCHEDULE", "partition")
    tail_streams = schedule in ("tail", "partition")
    # CUs of the small streams: a rank's share of the comparison grows with the world size (every rank streams the other
    # ranks' keys through its filter), so its streams get more of the chip -- measured with BENCH_SIM_WORLD (rank 0's
    # share, the largest; tools/sim_world_sweep.sh, ms per step at 64 / 96 / 128 CUs, rows in blocks): W = 2: 0.115 /
    # 0.124 / 0.144, W = 4: 0.142 / 0.124 / 0.147, W = 8: 0.185 / 0.149 / 0.148 (round 2, strided rows, every key
    # dealt on every rank: W = 4: 0.156 / 0.135 / 0.150, W = 8: 0.267 / 0.197 / 0.167)
    sim_w = max(world, int(xenv("BENCH_SIM_WORLD", "1")))
    small_cus = int(xenv("BENCH_SMALL_CUS", "64" if sim_w <= 2 else "96"))
    ctx_full, full_stream = ctx, stream                   # whole-device context: setup and the extras
    if schedule == "partition" and pipelined:
        try:                                              # (a runtime without CU masks: fall back to the unpartitioned schedule)
            sp.stream_destroy(local_rank, sp.stream_create_cus(local_rank, 0, 32))
        except sp.SpspError as e:
            log("CU-masked streams unavailable (%s): schedule 'single'" % e)
            schedule, tail_streams = "single", False
    if schedule == "partition" and pipelined:
        n_dev_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        # BENCH_DENSE_FIRST_CU (experiment): the dense stream may start below the small streams' upper end, i.e. share CUs
        dense_first = int(xenv("BENCH_DENSE_FIRST_CU", str(small_cus)))
        dense_cus = n_dev_cus - dense_first
        masked = [sp.stream_create_cus(local_rank, dense_first, dense_cus)]
        stream = torch.cuda.ExternalStream(masked[0], device=dev)
        torch.cuda.set_stream(stream)
        ctx = sp.Context(local_rank, stream.cuda_stream)
        ctx.set_cu_count(dense_cus, 2)

    class Slot:
        pass

    slots = []
    dense_streams = int(xenv("BENCH_DENSE_STREAMS", "1")) if (schedule == "partition" and pipelined) else 1
    scan_buffers = int(xenv("BENCH_SCAN_BUFFERS", "1")) if pipelined else 1
    dense_extra = []
    # (four steps in flight: the closed step has more host work per visit -- two more queueing calls -- and the scans'
    # sparse stages finish later beside the key extraction: 0.134 / 0.127 / 0.124 ms per step with 2 / 3 / 4 slots)
    # (the multi-GPU step, which still compares the setup's keys, keeps the two slots it was tuned with)
    for i in range(int(xenv("BENCH_SLOTS", "2" if use_dist else "4")) if pipelined else 1):
        sl = Slot()
        if pipelined:
            sl.stream_a = stream if (i == 0 or schedule != "streams") else torch.cuda.Stream(device=dev)
            if schedule == "partition" and dense_streams > 1 and i % dense_streams:
                # BENCH_DENSE_STREAMS=2 (experiment): consecutive dense passes on two streams that own the SAME CUs, not ordered
                # against each other -- the next pass's workgroups move in as the previous one's leave
                if len(dense_extra) < i % dense_streams:
                    masked.append(sp.stream_create_cus(local_rank, dense_first, dense_cus))
                    dense_extra.append(torch.cuda.ExternalStream(masked[-1], device=dev))
                sl.stream_a = dense_extra[i % dense_streams - 1]
            sl.scan = ctx if i == 0 else sp.Context(local_rank, sl.stream_a.cuda_stream)
            if schedule == "partition":
                sl.scan.set_cu_count(dense_cus, 2)
                mode = xenv("BENCH_SMALL_STREAMS", "shared")
                if i == 0 or mode == "own":
                    # one stream for every slot's sparse stages and one for every comparison by default: every CU-masked
                    # stream is a hardware queue of its own, and a handful of them already delay each other's packets
                    # ("own": a pair per slot; "cmp": one for the sparse stages, one per slot for the comparisons)
                    masked += [sp.stream_create_cus(local_rank, 0, small_cus), sp.stream_create_cus(local_rank, 0, small_cus)]
                    tail_h = masked[-2]
                    small_b = torch.cuda.ExternalStream(masked[-1], device=dev)
                elif mode == "cmp":
                    masked += [sp.stream_create_cus(local_rank, 0, small_cus)]
                    small_b = torch.cuda.ExternalStream(masked[-1], device=dev)
                sl.scan.scan_tail_stream(True, tail_h)
                sl.stream_b = small_b
            else:
                if tail_streams:
                    sl.scan.scan_tail_stream(True)
                sl.stream_b = torch.cuda.Stream(device=dev)
            sl.scans = [sl.scan]
            if scan_buffers > 1:
                # BENCH_SCAN_BUFFERS=2 (experiment): a slot alternates between two scan contexts, so the scan queued in a visit
                # does not rewrite the super-k-mer stream the key extraction queued in the same visit still reads
                extra = sp.Context(local_rank, sl.stream_a.cuda_stream)
                if schedule == "partition":
                    extra.set_cu_count(dense_cus, 2)
                    extra.scan_tail_stream(True, tail_h)
                elif tail_streams:
                    extra.scan_tail_stream(True)
                sl.scans.append(extra)
            sl.sv = 0
            sl.cmp = sp.Context(local_rank, sl.stream_b.cuda_stream)
            if schedule == "partition":
                sl.cmp.set_cu_count(small_cus)
        else:
            sl.stream_a = sl.stream_b = stream
            sl.scan = sl.cmp = ctx
        slots.append(sl)
    all_ctx = []
    for sl in slots:
        for c in (*getattr(sl, "scans", [sl.scan]), sl.cmp):
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
    # Consecutive steps scan DIFFERENT batches: a ring of BENCH_BATCHES buffers of the same shape (default 3 x 500 MB), so
    # that no step can be served by what the previous one left in the 256 MiB Infinity Cache (FETCH_SIZE counts its hits
    # as if they were HBM reads).  Batch b > 0 = batch 0 under a fixed permutation of the alphabet (A->C->G->T->A applied b
    # times): other k-mers, other minimizers, other hits -- the same records, family structure and base composition.
    n_batches = max(1, int(xenv("BENCH_BATCHES", "3")))
    d_batches = [d_bases]
    for b in range(1, n_batches):
        lut = torch.arange(256, dtype=torch.uint8, device=dev)
        for j, c in enumerate(b"ACGT"):
            lut[c] = b"ACGT"[(j + b) % 4]
        nb = torch.empty_like(d_bases)
        for a in range(0, d_bases.numel(), 1 << 27):
            nb[a:a + (1 << 27)] = lut[d_bases[a:a + (1 << 27)].long()]
        d_batches.append(nb)
    del bases
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
    sim_world = int(xenv("BENCH_SIM_WORLD", "1")) if not use_dist else 1
    my_sk_off = np.zeros(args.genomes + 1, dtype=np.uint64)
    my_sk_off[1:] = np.cumsum(my_n)
    # rows of the pair matrix a rank owns after the all-gather: the BLOCK of the sketches it scanned (default: its
    # dictionary holds its own keys + what passes the filter) or every world-th row (BENCH_ROWS=strided)
    row_form = xenv("BENCH_ROWS", "block")
    # multi-GPU exchange: north_star's form (RCCL all-gather of the packed keys + row ownership) by default;
    # BENCH_EXCHANGE=slots selects the key-partitioned all-to-all + partial-matrix reduction (DESIGN.md 5 prices both)
    exchange_kind = xenv("BENCH_EXCHANGE", "gather") if use_dist else "none"
    if exchange_kind == "slots":      # key-partitioned: all-to-all of own keys + all-reduce of partial matrices
        for sl in slots:
            sl.exchange = spd.SlotExchange(sl.cmp, K, args.genomes, int(my_sk_off[-1]), dev, stream=sl.stream_b,
                                           reduce=xenv("BENCH_SLOT_REDUCE", "scatter"))
        sk_off = np.zeros(n_total + 1, dtype=np.uint64)
        sk_off[-1] = slots[0].exchange.max_keys * world           # log line only
    elif exchange_kind == "gather":   # all-gather of every rank's keys; a rank owns the rows of the sketches it scanned
        for sl in slots:
            sl.exchange = spd.KeyExchange(my_n, dev, stream=sl.stream_b, rows=row_form)
        sk_off = slots[0].exchange.sk_off
    elif sim_world > 1:
        # BENCH_SIM_WORLD=W on one GPU (analysis only, the line is not a measurement): this rank's share of the
        # comparison at world size W -- the keys of W x genomes sketches resident (as after the all-gather), the rows
        # of rank 0 (its own block, which sees every other rank's keys; BENCH_ROWS=strided: rows i % W == 0) -- next
        # to the usual scan, to size the CU partition for W ranks; no exchange is simulated
        mins, los, ns = [my_min], [my_lo], [my_n]
        for r in range(1, sim_world):
            for g in synth.family_genomes(2 + 1000 * r, args.genomes, args.length, N_FAMILIES, MUS):
                gb, go = synth.concat_records([g])
                sk = sp.sketch_parse(sp.sketch_build(p, S, gb, go, ctx.scan(p, gb, go))[0])
                mins.append(sk.minimizer.astype(np.uint32)), los.append(sk.kmer_lo.astype(np.uint64)), ns.append([len(sk)])
        n_total = args.genomes * sim_world
        sk_off = np.zeros(n_total + 1, dtype=np.uint64)
        sk_off[1:] = np.cumsum(np.concatenate(ns))
        d_all_min = torch.from_numpy(np.concatenate(mins).view(np.int32)).to(dev)
        d_all_lo = torch.from_numpy(np.concatenate(los).view(np.int64)).to(dev)
    else:
        sk_off = np.zeros(n_total + 1, dtype=np.uint64)
        sk_off[1:] = np.cumsum(my_n)
        d_all_min, d_all_lo = d_my_min, d_my_lo
    if exchange_kind == "gather":
        row_args = slots[0].exchange.row_args()
    elif sim_world > 1:
        row_args = (0, 1, args.genomes) if row_form == "block" else (0, sim_world, n_total)
    else:
        row_args = (0, 1, n_total)
    for sl in slots:
        sl.d_inter = torch.zeros((n_total, n_total), dtype=torch.int32, device=dev)
    d_inter = slots[0].d_inter
    pairs_per_step = n_total * (n_total - 1) // 2
    torch.cuda.synchronize()
    if rank == 0:
        log("setup %.1fs: %d genomes x %d bp per GPU, %d records, %d k-mers/step/GPU, %d sketch keys, %d pairs"
            % (time.time() - t_setup, args.genomes, args.length, len(recs), kmers_per_step, int(sk_off[-1]), pairs_per_step))

    scan_args = [(p, t.data_ptr(), t.numel(), d_off.data_ptr(), len(recs)) for t in d_batches]
    # genome i = records [first_rec[i], first_rec[i + 1]) of the batch (1 + i % 3 records each, as laid out above)
    first_rec = np.zeros(args.genomes + 1, dtype=np.uint32)
    first_rec[1:] = np.cumsum([1 + i % 3 for i in range(args.genomes)])
    # The step closes on itself (single GPU): the keys the comparison of step t works on are made ON THE DEVICE from the
    # super-k-mer stream of an earlier step's scan (spsp_sketch_keys_device: what the comparator would read from the sketch
    # files of those genomes) -- no setup-time sketches inside the timed region.  BENCH_DEVICE_KEYS=0: the round-2 step
    # (keys of the setup's sketches, the same every step).
    device_keys = (exchange_kind == "none" and sim_world == 1 and pipelined and xenv("BENCH_DEVICE_KEYS", "1") != "0"
                   and xenv("BENCH_DEBUG_SKIP_COMPARE") != "1")
    if device_keys:
        # a third stream on the small CUs for the key extraction, two key contexts per slot (one's arrays are read by the
        # comparison in flight while the other's are being rewritten)
        keys_stream = xenv("BENCH_KEYS_STREAM", "own")  # "own" | "cmp" (the comparisons' stream) | "dense" (behind every dense pass)
        if keys_stream == "cmp":
            stream_k = slots[0].stream_b
        elif keys_stream == "dense":
            stream_k = stream
        elif schedule == "partition":
            # BENCH_KEYS_CUS="first,count" (experiment): the key extraction on a part of the small CUs only
            k_first, k_cnt = (int(x) for x in xenv("BENCH_KEYS_CUS", "0,%d" % small_cus).split(","))
            masked.append(sp.stream_create_cus(local_rank, k_first, k_cnt))
            stream_k = torch.cuda.ExternalStream(masked[-1], device=dev)
        else:
            stream_k = torch.cuda.Stream(device=dev)
        # BENCH_KEYS_STREAMS=2 (experiment): the key extractions of consecutive steps on two streams over the same CUs
        streams_k = [stream_k]
        if int(xenv("BENCH_KEYS_STREAMS", "1")) > 1 and schedule == "partition" and keys_stream == "own":
            masked.append(sp.stream_create_cus(local_rank, k_first, k_cnt))
            streams_k.append(torch.cuda.ExternalStream(masked[-1], device=dev))
        for si, sl in enumerate(slots):
            stream_k = streams_k[si % len(streams_k)]
            sl.cmp.compare_keys_unordered(True)           # the step's keys come out of an LDS table per genome: distinct, not sorted
            sl.keys = [sp.Context(local_rank, stream_k.cuda_stream), sp.Context(local_rank, stream_k.cuda_stream)]
            sl.kv = 0
            sl.keys_job = [False, False]
            for c in sl.keys:
                if schedule == "partition":
                    c.set_cu_count(dense_cus if keys_stream == "dense" else small_cus)
                all_ctx.append(c)
    step_no = [0]                                             # steps queued so far: step i scans batch i % n_batches

    # analysis only (the line it prints is not a valid measurement): the step without its comparison, to see what the
    # comparison's kernels cost the dense pass they run beside
    skip_compare = xenv("BENCH_DEBUG_SKIP_COMPARE") == "1" and exchange_kind == "none"

    def on_b(sl):
        # torch ops and RCCL order themselves against torch's CURRENT stream: stream B of the slot.  Entering a stream
        # context costs ~7 us of host time, so the thread simply stays on stream B when all slots share it (below)
        if torch.cuda.current_stream(dev) == sl.stream_b:
            return contextlib.nullcontext()
        return torch.cuda.stream(sl.stream_b)

    # BENCH_DEBUG_NOISE_KERNELS=n (analysis only): n empty launches per step on a stream of the small CUs -- what kernel
    # boundaries beside a dense pass cost it, whatever the kernels do
    noise_n = int(xenv("BENCH_DEBUG_NOISE_KERNELS", "0"))
    if noise_n and schedule == "partition" and pipelined:
        masked.append(sp.stream_create_cus(local_rank, 0, small_cus))
        noise_stream = torch.cuda.ExternalStream(masked[-1], device=dev)
        noise_t = torch.zeros(64, dtype=torch.int32, device=dev)
    else:
        noise_n = 0

    def queue_step(sl, prev, nxt=None):
        """queue one whole step on slot sl without waiting for anything; on the GPU its dense pass starts behind
        the dense pass of slot prev.  nxt = the slot of the following step (slots exchange: its key partition is
        started from here, behind this dense pass)"""
        if prev is not None and prev.stream_a is not sl.stream_a and dense_streams == 1:
            sl.scan.wait_dense(prev.scan)                 # "streams": dense passes never overlap each other
        sl.batch = step_no[0] % n_batches
        step_no[0] += 1
        if len(getattr(sl, "scans", ())) > 1:
            sl.sv ^= 1
            sl.scan = sl.scans[sl.sv]                     # the context whose output nothing queued in this visit reads
        reader = getattr(sl.scan, "_reader", None)
        if device_keys and reader is not None:
            sl.scan.scan_output_wait(reader)              # this scan's last stage rewrites the buffer that key extraction read (or still reads)
        sl.scan.scan_device_begin(*scan_args[sl.batch])   # "tail" / "single": dense passes in order on the one scan stream
        if noise_n:
            with torch.cuda.stream(noise_stream):
                for _ in range(noise_n):
                    noise_t.add_(1)
        if exchange_kind == "none":
            if schedule != "partition":
                sl.cmp.wait_dense(sl.scan)                # the comparison starts behind this step's dense pass
            if not skip_compare and not (device_keys and getattr(sl, "compare_queued", False)):
                sl.cmp.compare_device_begin(K, d_all_min.data_ptr(), d_all_lo.data_ptr(), None, sk_off, n_total, row_args[0], row_args[1],
                                            sl.d_inter.data_ptr(), n_query=row_args[2])                    # stream B
                sl.compare_queued = True
            return
        with on_b(sl):
            if exchange_kind == "slots":
                if getattr(sl, "handle", None) is None:   # first steps only: later ones were started a step ahead (below)
                    sl.handle = sl.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off)
                sl.cmp.wait_dense(sl.scan)
                sl.exchange.end_queue(sl.handle, sl.d_inter)
                sl.handle = None
            else:
                g = sl.exchange.exchange(d_my_min, d_my_lo)
                sl.cmp.wait_dense(sl.scan)
                sl.cmp.compare_device_begin(K, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, sk_off, n_total, row_args[0],
                                            row_args[1], sl.d_inter.data_ptr(), n_query=row_args[2])

        if exchange_kind == "slots" and nxt is not None and nxt is not sl and getattr(nxt, "handle", None) is None:
            # the NEXT step's key partition + RCCL all-to-all: behind this step's dense pass, long before the next
            # comparison needs it, and behind the comparison still queued on that stream
            with on_b(nxt):
                nxt.cmp.wait_dense(sl.scan)
                nxt.handle = nxt.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off)

    def collect_step(sl):
        tq = time.perf_counter()
        d_out, n_out = sl.scan.scan_device_end()
        host_detail["scan_end"] = host_detail.get("scan_end", 0.0) + time.perf_counter() - tq
        if device_keys:
            # stream B of this slot, in this order: [comparison of the keys made one visit ago] -> [keys of the scan collected now].
            # The key arrays belong to the context: the comparison that reads them is queued in front of the extraction that
            # rewrites them, and nothing here waits for work queued in this same visit.
            tr0 = time.perf_counter()
            sl.cmp.compare_end()                          # the comparison queued one visit ago (it read the arrays rewritten below)
            sl.compare_queued = False
            tr1 = time.perf_counter()
            kprev, kcur = sl.kv ^ 1, sl.kv
            if sl.keys_job[kprev]:
                d_mn, d_lo, _, koff = sl.keys[kprev].sketch_keys_device_end()   # queued one visit ago: long done
                sl.keys_job[kprev] = False
                last["keys_total"] = int(koff[-1])
                tr2 = time.perf_counter()
                sl.cmp.compare_device_begin(K, d_mn, d_lo, None, koff, n_total, 0, 1, sl.d_inter.data_ptr())      # stream B
                sl.compare_queued = True
            else:
                tr2 = tr1
            tr3 = time.perf_counter()
            sl.keys[kcur].sketch_keys_device_begin(p, scan_args[sl.batch][1], scan_args[sl.batch][2], d_off.data_ptr(), d_out, n_out, first_rec,
                                                   unordered=True)                                                  # stream K
            tr4 = time.perf_counter()
            for name, dt in (("compare_end", tr1 - tr0), ("keys_end", tr2 - tr1), ("compare_begin", tr3 - tr2), ("keys_begin", tr4 - tr3)):
                host_detail[name] = host_detail.get(name, 0.0) + dt
            sl.keys_job[kcur] = True
            sl.scan._reader = sl.keys[kcur]
            sl.kv ^= 1
            last["n_out"], last["slot"] = n_out, sl
            last["n_out_batch"][sl.batch] = n_out
            return
        if exchange_kind == "slots":
            with on_b(sl):
                sl.exchange.end_collect(sl.d_inter)       # partial pair matrix done -> RCCL reduction
        else:
            if not skip_compare:
                sl.cmp.compare_end()
            if exchange_kind == "gather":
                sl.exchange.collect_rows(sl.d_inter)      # strips -> rank 0, on stream B
        last["n_out"], last["slot"] = n_out, sl
        last["n_out_batch"][sl.batch] = n_out

    def run_steps(n):
        """n steps = n scans + n comparisons"""
        if not pipelined:
            sl = slots[0]
            for _ in range(n):
                # one stream: the key exchange (RCCL all-to-all) is queued first and runs behind the scan kernels
                h = sl.exchange.begin(d_my_min.data_ptr(), d_my_lo.data_ptr(), None, my_sk_off) if exchange_kind == "slots" else None
                sl.batch = step_no[0] % n_batches
                step_no[0] += 1
                d_out, n_out = ctx.scan_device(*scan_args[sl.batch])
                if exchange_kind == "slots":
                    sl.exchange.end(h, sl.d_inter)        # partial pair matrix + RCCL reduction
                else:
                    if exchange_kind == "gather":
                        g = sl.exchange.exchange(d_my_min, d_my_lo)
                        mn_ptr, lo_ptr = g.minimizer.data_ptr(), g.kmer_lo.data_ptr()
                    else:
                        mn_ptr, lo_ptr = d_all_min.data_ptr(), d_all_lo.data_ptr()
                    ctx.compare_device(K, mn_ptr, lo_ptr, None, sk_off, n_total, row_args[0], row_args[1], sl.d_inter.data_ptr(), n_query=row_args[2])
                    if exchange_kind == "gather":
                        sl.exchange.collect_rows(sl.d_inter)
                last["n_out"], last["slot"] = n_out, sl
                last["n_out_batch"][sl.batch] = n_out
            return
        import collections
        # a step is collected when `collect_depth` younger ones have been queued (default: when its slot comes round again).
        # BENCH_COLLECT_DEPTH < slots: earlier -- fewer chains left for the drain at the end of a short run, less slack for the host
        collect_depth = min(len(slots), int(xenv("BENCH_COLLECT_DEPTH", str(len(slots)))))
        pending = collections.deque()                     # steps queued and not yet collected, oldest first
        prev = None
        for i in range(n):
            sl = slots[i % len(slots)]
            t_c = time.perf_counter()
            if len(pending) == collect_depth:
                collect_step(pending.popleft())           # the oldest step ran on `sl`: the host waits for it only now,
            t_a = time.perf_counter()                     # with the steps queued since then already on the GPU's queues
            queue_step(sl, prev, slots[(i + 1) % len(slots)] if i + 1 < n else None)
            t_b = time.perf_counter()
            host["queue"] += t_b - t_a
            host["collect"] += t_a - t_c
            pending.append(sl)
            prev = sl
        while pending:
            collect_step(pending.popleft())

    last = {"n_out": 0, "slot": slots[0], "n_out_batch": {}}
    host_detail = {}                        # closed step: host seconds per call of a visit
    host = {"queue": 0.0, "collect": 0.0}   # host seconds spent queueing / waiting (pipelined mode)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Clocks and caches: the GPU has idled through the setup above (host-side synthesis, uploads) and takes ~100 steps
    # to come back to its sustained clocks (200 timed steps: 0.124-0.127 ms each; 600: 0.117).  An untimed run-in of the
    # same steps precedes the W warm-up steps, so that short runs measure the same machine state as long ones.
    stay_on_b = use_dist and pipelined and all(sl.stream_b is slots[0].stream_b for sl in slots)
    if stay_on_b:
        torch.cuda.set_stream(slots[0].stream_b)          # see on_b(); the scans name their streams themselves
    prewarm = int(os.environ.get("BENCH_PREWARM_STEPS", "300"))
    if prewarm > 0:
        run_steps(prewarm)
    run_steps(args.warmup)
    # the roofline needs the dense kernel's duration from HIP events in the timed region; the other brackets
    # (whole pipelines, accumulate kernel) are extra packets on the streams and are only recorded on request
    kinds = sp.TIME_ALL if xenv("BENCH_STAGE_TIMING", "0") == "1" or not pipelined else sp.TIME_DENSE
    if xenv("BENCH_STAGE_TIMING") == "off":     # experiment: what the event packets themselves cost
        kinds = 0
    # every 8th dense pass is bracketed: two event packets per launch cost the pipelined step 8.6 us of 123 (measured,
    # BENCH_STAGE_TIMING=off); BENCH_TIMING_EVERY=1 brackets them all
    # (short runs -- the driver's --steps 20 -- bracket every 4th, so that a handful of launches is behind `roofline.achieved`)
    timing_every = int(xenv("BENCH_TIMING_EVERY", "8" if args.steps >= 64 else "4")) if pipelined and kinds == sp.TIME_DENSE else 1
    for c in all_ctx:
        c.timing_enable(True, kinds)
        c.timing_sample(timing_every)
        c.timing_read()
    fence()
    host["queue"] = host["collect"] = 0.0
    host_detail.clear()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    host_main = {"queueing": host["queue"] * 1e3 / args.steps, "waiting": host["collect"] * 1e3 / args.steps,
                 **{("in_" + k): v * 1e3 / args.steps for k, v in host_detail.items()}}
    if stay_on_b:
        torch.cuda.set_stream(stream)
    n_out = last["n_out_batch"].get(0, last["n_out"])       # batch 0 = the records the oracle is run over (cpu_baseline)
    d_inter = last["slot"].d_inter
    exchange = getattr(last["slot"], "exchange", None)
    tm = None
    for c in all_ctx:                                     # HIP-event logs of all contexts, summed
        t = c.timing_read()
        c.timing_enable(False)
        c.timing_sample(1)
        if tm is None:
            tm = dict(t)
        else:
            for key in tm:
                tm[key] += t[key]
    inter_nonzero_main = int(torch.count_nonzero(d_inter).item())
    # Second, shorter timed region (single GPU): the OPEN step of rounds 1-2 -- the same scans, the comparison working on the
    # keys of the setup's sketches (the same every step), no key extraction in the loop.  Reported beside the headline as
    # `open_loop`: what the two kernels chains do when the hand-off between them is left out.
    # The same closed step over a longer run (single GPU, when the timed region was short -- the driver's --steps 20): what
    # filling and draining the pipeline once costs a 20-step region.  Reported beside the headline.
    long_run = None
    if device_keys and args.steps < 200:
        try:
            for c in all_ctx:                             # the same event brackets as a --steps 200 run has (every 8th dense pass)
                c.timing_enable(True, kinds)
                c.timing_sample(8 if kinds == sp.TIME_DENSE else 1)
            run_steps(args.warmup)
            fence()
            t0 = time.perf_counter()
            run_steps(200)
            fence()
            e_long = time.perf_counter() - t0
            for c in all_ctx:
                c.timing_read()
                c.timing_enable(False)
                c.timing_sample(1)
            long_run = {"steps": 200, "ms_per_step": e_long * 1e3 / 200, "value": float(kmers_per_step) * 200 / e_long,
                        "what": "the closed step of the headline over 200 steps, measured right behind the timed region"}
        except Exception as e:  # noqa: BLE001
            long_run = {"error": repr(e)}
    open_loop = None
    if device_keys:
        try:
            for sl in slots:                              # finish what the closed steps left queued
                if getattr(sl, "compare_queued", False):
                    sl.cmp.compare_end()
                    sl.compare_queued = False
                for j in (0, 1):
                    if sl.keys_job[j]:
                        sl.keys[j].sketch_keys_device_end()
                        sl.keys_job[j] = False
                for c in getattr(sl, "scans", [sl.scan]):
                    c._reader = None
            closed_keys_total = last.get("keys_total")
            device_keys = False
            n_open = min(args.steps, 200)
            run_steps(max(args.warmup, 10))
            fence()
            t0 = time.perf_counter()
            run_steps(n_open)
            fence()
            e_open = time.perf_counter() - t0
            open_loop = {"steps": n_open, "ms_per_step": e_open * 1e3 / n_open, "value": float(kmers_per_step) * n_open / e_open,
                         "what": "scan(batch t) || all-vs-all(keys of the SETUP's sketches, the same every step): the step of rounds 1 and 2, "
                                 "without the key extraction between the two halves"}
            device_keys = True
            last["keys_total"] = closed_keys_total
        except Exception as e:  # noqa: BLE001
            open_loop = {"error": repr(e)}
            device_keys = True

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
    inter_nonzero = inter_nonzero_main
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
                ra = ge.row_args()
                ctx.compare_device(K, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, ge.sk_off, n_total, ra[0], ra[1],
                                   d_ref.data_ptr(), n_query=ra[2])
                dist.all_reduce(d_ref, op=dist.ReduceOp.SUM)
                exchange_check = "equal to all-gather form" if bool(torch.equal(d_ref, d_inter)) else "MISMATCH"
        except Exception as e:  # noqa: BLE001 -- the check must not take the bench line down
            exchange_check = "check failed: %r" % (e,)
    elif exchange_kind == "gather":
        # untimed: the strips collected on rank 0 (every rank's own rows) must equal ONE device's comparison of all rows
        # over the gathered keys (that single-device result is what the gpu tests hold against the oracle)
        try:
            with torch.cuda.stream(last["slot"].stream_b):
                g = last["slot"].exchange.exchange(d_my_min, d_my_lo)
                d_ref = torch.zeros_like(d_inter)
                last["slot"].cmp.compare_device(K, g.minimizer.data_ptr(), g.kmer_lo.data_ptr(), None, sk_off, n_total, 0, 1,
                                                d_ref.data_ptr())
            torch.cuda.synchronize()
            if rank == 0:
                exchange_check = ("collected strips equal a one-device comparison of all %d sketches" % n_total
                                  if bool(torch.equal(d_ref, d_inter)) else "MISMATCH")
        except Exception as e:  # noqa: BLE001
            exchange_check = "check failed: %r" % (e,)

    # BASELINE configs[3] over the ranks of this job (every rank takes part; rank 0 reports)
    c4_dist = None
    if use_dist and not args.no_extras:          # (BENCH_FORCE_DIST=1: also with ONE rank -- the RCCL calls of the leg on a one-GPU box)
        try:
            c4_dist = compare_config4_dist(ctx_full, dev, full_stream, rank, world, backend)
        except Exception as e:  # noqa: BLE001
            c4_dist = {"error": repr(e)}
    if rank == 0:
        env_set, env_why = env_report()
        value = total_kmers_per_step * args.steps / elapsed
        ms_per_step = elapsed * 1e3 / args.steps
        dense_avg_ms = dense_ms / max(1, tm["dense_launches"])
        achieved = (d_bases.numel() / 1e9) / (dense_avg_ms / 1e3) if dense_avg_ms > 0 else 0.0  # 1 B per position (ASCII)
        step_achieved = (d_bases.numel() / 1e9) / (ms_per_step / 1e3)
        compare_avg_ms = compare_ms / max(1, tm["compare_calls"])
        out = {
            "metric": "k-mers hashed/s (sketch) + sketch-pairs/s (all-vs-all)",
            "value": value,
            "unit": "k-mers hashed/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "run_in_steps": prewarm,   # untimed, in front of the warm-up: brings the GPU back to its sustained clocks after the setup
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "valid": not env_why, **({"invalid_because": env_why} if env_why else {}),
            "config": {"env": env_set, **({"env_ignored_without_--experiment": sorted(XENV["ignored"])} if XENV["ignored"] else {}),
                       "workload": "BASELINE configs[1]: %d synthetic %d bp genomes per GPU (10 families, mu 0.001/0.01), "
                                   "k=31 m=11 s=1000, scan + all-vs-all; inputs resident in HBM; consecutive steps scan different batches "
                                   "(ring of %d x %d MB)" % (args.genomes, args.length, n_batches, d_bases.numel() // 1000000),
                       "batches": n_batches,
                       "k": K, "m": M, "s": S, "genomes_per_gpu": args.genomes, "genome_len": args.length,
                       "sketches_total": n_total, "scan_mode": args.mode,
                       "parallelism": {"none": "single GPU",
                                       "slots": "genomes sharded by rank; sketch keys partitioned by hash, RCCL all-to-all behind "
                                                "the scan, per-rank partial pair matrix, RCCL reduction",
                                       "gather": "genomes sharded by rank; RCCL all-gather of the packed sketch keys, every rank "
                                                 "computes the pair-matrix rows " + ("of its own sketches (dictionary of its own keys; "
                                                 "the other ranks' keys pass a filter first)" if row_form == "block" else "i % N == rank") +
                                                 ", strips collected on rank 0 (SURVEY.md 8e; "
                                                 "matrices <= 16 MB by one RCCL reduce of the zero-padded strips)"}[exchange_kind],
                       "exchange_check": exchange_check,
                       **({"simulation": "BENCH_SIM_WORLD=%d: one rank's share of the comparison at that world size, no exchange: "
                                         "NOT a measurement" % sim_world} if sim_world > 1 else {}),
                       **({"rehearsal": "BENCH_BACKEND=%s, all ranks on one device: NOT a measurement" % backend}
                          if (use_dist and (backend != "nccl" or os.environ.get("BENCH_SHARE_GPU") == "1")) else {}),
                       "step": ((("scan(batch t) || keys(scan t-2, on the device) || all-vs-all(keys of scan t-4)" if device_keys else "scan(batch t) || all-vs-all(sketches of batch t-1)") +
                                 ": the chip is partitioned by CU-masked streams -- "
                                 "dense passes back to back on %d CUs (two workgroups each), the scans' sparse stages on one stream and "
                                 "the comparisons on another that share the other %d CUs; the host queues step t+1 before collecting "
                                 "step t [schedule partition]" % (dense_cus, small_cus)) if pipelined and schedule == "partition"
                                else "scan(batch t) || all-vs-all(sketches of batch t-1): scans in order on one stream, every slot's "
                                "comparison on a stream of its own behind its dense pass; the host queues step t+1 before "
                                "collecting step t [schedule %s]" % schedule if pipelined
                                else "scan then all-vs-all on one stream, one step at a time")},
            # the rate of the comparison inside the step: its own HIP-event bracket when one was recorded, else pairs per step time
            "sketch_pairs_per_s": (pairs_per_step / (compare_avg_ms / 1e3) if compare_avg_ms > 0
                                   else pairs_per_step * args.steps / elapsed),
            "sketch_pairs_per_s_basis": "compare pipeline (HIP events)" if compare_avg_ms > 0 else "whole step (scan + all-vs-all)",
            "stage_ms": {"scan_pipeline": scan_ms / tm["scan_calls"] if tm["scan_calls"] else None, "dense_kernel": dense_avg_ms,
                         "dense_launches_timed": int(tm["dense_launches"]), "dense_launches": args.steps,
                         "compare_pipeline": compare_avg_ms if tm["compare_calls"] else None,
                         "accumulate_kernel": acc_ms / tm["accumulate_launches"] if tm["accumulate_launches"] else None},
            "superkmers_per_step": int(n_out), "inter_nonzero": inter_nonzero,
            **({"open_loop": open_loop} if open_loop is not None else {}),
            **({"long_run": long_run} if long_run is not None else {}),
            "host_ms_per_step": host_main,
            "roofline": {"kernel": "k_dense_pair (non-temporal 16-byte loads, 2-bit pack, LDS pair-table test of every m-mer position; XXH64 on survivors)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": int(d_bases.numel()),
                         "byte_model": "1 B per m-mer position (cleaned ASCII, as getLineFasta returns it)",
                         # the whole step against the same roofline: bytes of one batch / ms_per_step
                         "step_achieved": step_achieved, "step_frac": step_achieved / HBM_PEAK_GBS},
        }
        out["roofline"].update(pmc_traffic(args, "k_dense_pair"))
        # the same kernel with the GPU to itself (dense stage only, untimed extra): what the kernel reaches when nothing
        # shares its CUs -- `achieved` above is measured inside the pipelined step, beside the other streams' kernels.
        # Like the steps, consecutive launches read different batches.
        n_b = d_bases.numel()
        try:
            ctx.timing_enable(True, sp.TIME_DENSE)
            ctx.timing_read()
            for i in range(21):
                ctx.scan_hits_device(p, d_batches[i % n_batches].data_ptr(), n_b)
            ta = ctx.timing_read()
            ctx.timing_enable(False)
            alone_ms = ta["dense_ms"] / max(1, ta["dense_launches"])
            out["roofline"]["alone"] = {"dense_kernel_ms": alone_ms, "achieved": n_b / 1e9 / (alone_ms / 1e3),
                                        "frac": n_b / 1e9 / (alone_ms / 1e3) / HBM_PEAK_GBS, "launches": int(ta["dense_launches"])}
        except Exception as e:  # noqa: BLE001
            out["roofline"]["alone"] = {"error": repr(e)}
        # ... on the whole chip (context without CU mask), ASCII and 2-bit input (SURVEY.md 8d's second byte model: the
        # same positions from a quarter of the bytes -- that form is bound by the LDS table reads, not by HBM)
        try:
            out["roofline"]["whole_chip"] = {}
            ctx_full.set_cu_count(0, 2)
            torch.cuda.synchronize()
            packed = []                                   # one packed copy per batch (125 MB each: ONE would sit in the Infinity Cache)
            for t in d_batches:
                d_pk = ctx_full.pack_bases_device(t.data_ptr(), n_b)
                packed.append(device_bytes_as_tensor(d_pk, (n_b + 15) // 16 * 4 + 256, dev).clone())
            torch.cuda.synchronize()
            for name, srcs, flg, bpp in (("ascii", [t.data_ptr() for t in d_batches], 0, 1.0),
                                         ("packed_2bit", [t.data_ptr() for t in packed], sp.SPSP_SCAN_PACKED_INPUT, 0.25)):
                pw = sp.make_params(K, M, S, flags=flags | flg)
                ctx_full.scan_hits_device(pw, srcs[0], n_b)
                ctx_full.timing_enable(True, sp.TIME_DENSE)
                ctx_full.timing_read()
                for i in range(21):
                    ctx_full.scan_hits_device(pw, srcs[i % len(srcs)], n_b)
                tw = ctx_full.timing_read()
                ctx_full.timing_enable(False)
                ms_w = tw["dense_ms"] / max(1, tw["dense_launches"])
                gbs = bpp * n_b / 1e9 / (ms_w / 1e3)
                out["roofline"]["whole_chip"][name] = {"dense_kernel_ms": ms_w, "positions_per_s": n_b / (ms_w / 1e3),
                                                       "byte_model": "%g B per m-mer position" % bpp, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
            del packed
            ctx_full.set_cu_count(0, 1)
        except Exception as e:  # noqa: BLE001
            out["roofline"]["whole_chip"] = {"error": repr(e)}
        # the roofline's denominator measured on this box in this process: streaming copy and streaming read
        # of 2 x 1 GiB on the whole chip, and on the CUs the dense stream owns inside the step
        try:
            torch.cuda.synchronize()
            pm = ctx_full.measure_hbm(1 << 30, 10)
            rl = out["roofline"]
            rl["peak_measured"] = {"copy_GBps": pm["copy_GBps"], "read_GBps": pm["read_GBps"], "bytes_per_buffer": pm["bytes"], "launches": pm["reps"],
                                   "kernels": "k_measure_copy / k_measure_read (non-temporal 16-byte loads and stores, 2 x 1024 lanes per CU), "
                                              "HIP events, after two warm-up launches; copy counts bytes read + written"}
            if schedule == "partition" and pipelined:
                pd = ctx.measure_hbm(1 << 30, 10)
                rl["peak_measured"]["dense_stream"] = {"n_cu": pd["n_cu"], "copy_GBps": pd["copy_GBps"], "read_GBps": pd["read_GBps"]}
            rl["frac_of_measured_copy"] = achieved / pm["copy_GBps"]
            rl["frac_of_measured_read"] = achieved / pm["read_GBps"]
            rl["step_frac_of_measured_read"] = step_achieved / pm["read_GBps"]
        except Exception as e:  # noqa: BLE001
            out["roofline"]["peak_measured"] = {"error": repr(e)}
        out["roofline"].update(sq_counters("k_dense_pair", dense_avg_ms, dense_cus if (schedule == "partition" and pipelined) else 0))
        d_parity = d_inter
        if device_keys:
            # untimed: the step's own path once more over batch 0 -- scan -> keys on the device -> comparison -- on the whole-device
            # context; its keys must equal those parsed from the setup's sketch payloads (what the comparator reads from files)
            # and its pair matrix goes to the oracle check below
            try:
                torch.cuda.synchronize()
                d_o, n_o = ctx_full.scan_device(*scan_args[0])
                # (sorted form here: the arrays can be held against the parsed sketches element by element)
                kmn, klo, _, koff = ctx_full.sketch_keys_device(p, scan_args[0][1], scan_args[0][2], d_off.data_ptr(), d_o, n_o, first_rec)
                same = bool((koff == sk_off).all())
                if same:
                    tot = int(koff[-1])
                    same = bool(torch.equal(device_bytes_as_tensor(kmn, 4 * tot, dev), d_all_min.view(torch.uint8)) and
                                torch.equal(device_bytes_as_tensor(klo, 8 * tot, dev), d_all_lo.view(torch.uint8)))
                d_parity = torch.zeros_like(d_inter)
                torch.cuda.synchronize()
                ctx_full.compare_device(K, kmn, klo, None, koff, n_total, 0, 1, d_parity.data_ptr())
                torch.cuda.synchronize()
                out["device_keys"] = {"in_timed_step": True, "keys_per_step": last.get("keys_total"),
                                      "equal_to_keys_parsed_from_sketch_payloads": same,
                                      "what": "spsp_sketch_keys_device: scan output -> sorted distinct (minimizer, canonical k-mer) keys per genome "
                                              "(handle_superkmer's uint8 counts, the emission / reader round trip and canonize composed, "
                                              "SubSampler.cpp:243-302,458-620, Comparator.cpp:186-260); the comparison of a step works on the "
                                              "keys of the scan this slot collected two steps earlier"}
            except Exception as e:  # noqa: BLE001
                out["device_keys"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(recs, payloads, p, int(n_out), d_parity)
        if world == 1 and not args.no_extras:
            try:
                out["compare"] = compare_config3(ctx_full, dev, args.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001 -- an extra must not take the headline down
                out["compare"] = {"error": repr(e)}
            try:
                out["end_to_end"] = end_to_end(ctx_full, args.no_cpu_baseline, genomes)
            except Exception as e:  # noqa: BLE001
                out["end_to_end"] = {"error": repr(e)}
            del d_batches[1:]                              # room (and a quiet card) for the two large legs
            pk = out["roofline"].get("peak_measured", {})
            try:
                out["compare_c4"] = compare_config4(ctx_full, dev, args.no_cpu_baseline, pk)
            except Exception as e:  # noqa: BLE001
                out["compare_c4"] = {"error": repr(e)}
            try:
                out["scan_c5"] = scan_config5(ctx_full, dev, args.no_cpu_baseline, pk)
            except Exception as e:  # noqa: BLE001
                out["scan_c5"] = {"error": repr(e)}
        if c4_dist is not None:
            out["compare_c4"] = c4_dist
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # contexts and CU-masked streams are released here, not by the interpreter's teardown (which may run after the
    # HIP runtime's own: a profiler then sees a crash at exit)
    torch.cuda.synchronize()
    for c in all_ctx:
        c.close()
    ctx.close()
    ctx_full.close()
    del full_stream
    for h in (masked if schedule == "partition" and pipelined else []):
        sp.stream_destroy(local_rank, h)


def device_bytes_as_tensor(ptr, nbytes, dev):
    """a uint8 torch view of `nbytes` of device memory at `ptr` (a buffer libspsp owns): plumbing for the bench only"""
    class _Arr:
        __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=dev)


def sq_counters(kernel, kernel_ms, n_cu, default_file="r05_pmc_sq_bench.json"):
    """VALU-issue fraction of a kernel (SURVEY.md 8d asks for it next to the HBM fraction: the scan is ALU-bound in
    practice).  SQ counters cannot be read from inside this process; they come from the committed rocprofv3 --pmc passes
    of the matching command (profiles/, with provenance).  A wave64 VALU instruction holds its SIMD's issue port for 4
    cycles when one wave issues alone and for 2 when several waves interleave (MI355X_MICROARCH.md, "Wave scheduling" and
    the "vector-instruction ISSUE cost" row); with SQ_BUSY_CU_CYCLES counting the cycles of every busy CU (4 SIMDs each)
    the fraction of the issue slots taken is  SQ_INSTS_VALU * c / (4 * SQ_BUSY_CU_CYCLES), c = 2 (floor) or 4."""
    try:
        name = os.environ.get("BENCH_SQ_FILE", default_file)
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        ks = [n for n in d["kernels"] if n == kernel or n.startswith(kernel + "<")]
        c = d["kernels"][ks[0]]
        f2 = c["SQ_INSTS_VALU"] * 2.0 / (4.0 * c["SQ_BUSY_CU_CYCLES"])
        return {"valu_issue_frac": f2 * 2.0, "valu_issue_frac_floor": f2,
                "valu_issue": {"kernel": ks[0], "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_BUSY_CU_CYCLES": c["SQ_BUSY_CU_CYCLES"],
                               "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"), "SQ_LDS_BANK_CONFLICT": c.get("SQ_LDS_BANK_CONFLICT"),
                               "SQ_LDS_IDX_ACTIVE": c.get("SQ_LDS_IDX_ACTIVE"),
                               "definition": "valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (4 SIMDs x SQ_BUSY_CU_CYCLES); _floor: x 2 cycles "
                                             "(interleaved waves, SIMD-32 issue over 2 cycles)",
                               "provenance": {"file": "profiles/" + name, "commit": d.get("commit"), "command": d.get("command")}}}
    except Exception:
        return {"valu_issue_frac": None}


def pmc_traffic(args, kernel):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes of this
    same command, gfx950 correction applied).  Counters cannot be read from inside this process: the figure comes from
    the committed profile of the matching workload, with its provenance; null when there is none."""
    try:
        name = os.environ.get("BENCH_PMC_FILE", "r05_pmc_hbm_traffic.json")
        path = os.path.join(ROOT, "profiles", name)
        d = json.load(open(path))
        w = d["workload"]
        if (w["genomes"], w["genome_len"], w["k"], w["m"], w["s"], w["scan_mode"]) != (args.genomes, args.length, K, M, S, args.mode):
            return {"traffic": None}
        names = [n for n in d["kernels"] if n == kernel or n.startswith(kernel + "<")]     # (template instances: k_dense_pair<true>)
        return {"traffic": d["kernels"][names[0]]["hbm_bytes_per_launch_corrected"],
                "traffic_provenance": {"file": "profiles/" + name, "commit": d.get("commit"),
                                       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950: "
                                                 "a wide coalesced streaming read, the calibrated case)"}}
    except Exception:
        return {"traffic": None}


def cpu_baseline(recs, payloads, p, gpu_superkmers, d_inter):
    """The reference-algorithm CPU restatement (oracle/) timed on this host, rank 0, N=1 only.
    Checker code used as a reported baseline -- never on the product path.  While it is at it, it checks the
    timed GPU step against it: super-k-mers emitted over the same records, pair matrix over the same sketches."""
    from oracle import oracle_py as orc
    build_flags = orc.use_native()      # -O3 -march=native -fopenmp, for this host's CPU
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
    cores = max(1, len(os.sched_getaffinity(0)))
    items = [synth.concat_records([r]) for r in recs[:used]]
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda bo: orc.scan_timed(K, M, p.threshold, bo[0], bo[1]), items))
    wall = time.perf_counter() - t0
    all_cores = {"value": sum(r[1] for r in res) / wall if wall > 0 else None, "cores": cores,
                 "sample": "same records, one record per task over %d host threads, %.2f s wall" % (cores, wall)}
    n = len(payloads)
    want_inter, _, csec = orc.compare(payloads, timed=True)  # the reference comparator is single-threaded too
    parity = {"pair_matrix": bool((d_inter.cpu().numpy().astype(np.uint32) == want_inter).all())}
    if used == len(recs):
        parity["superkmers_per_step"] = bool(emitted == gpu_superkmers)
    return {"parity_vs_oracle": parity, "value": kmers / spent if spent > 0 else None, "unit": "k-mers hashed/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model(), "host_threads_available": cores,
            "build": "g++ " + str(build_flags),
            "sample": "oracle scan loop (SubSampler.cpp:357-455 restated), single thread, first %d of %d records "
                      "of the same workload, %.1f s" % (used, len(recs), spent),
            "all_cores": all_cores,
            "sketch_pairs_per_s": (n * (n - 1) // 2) / csec if csec > 0 else None,
            "pairs_sample": "oracle compare_sketches (Comparator.cpp:39-287 restated) over the same %d sketches, "
                            "single thread, %.2f s" % (n, csec)}


# ---------------------------------------------------------------------------------------------------------------
def compare_config4(ctx, dev, skip_oracle, peak):
    """BASELINE configs[3] on ONE GPU: 10 000 sketches of ~5 000 k-mers (500 families of 20 at mu 0.001/0.01/0.05),
    synthesised directly at the super-k-mer level, keys resident in HBM; what the reference times
    (Comparator.cpp:499-509 around compare_sketches).  The oracle cannot hold this size (one (N+1)-bit colour vector
    per distinct key: ~30 GB), so it is timed and compared on the first `sub` sketches -- pair counts do not depend on
    the other sketches -- and 300 pairs drawn from the whole matrix are held against numpy set intersections."""
    n, k, m = 10_000, 31, 11
    t0 = time.time()
    D = synth.direct_family_sketches(n, fam_size=20, k=k, m=m, seed=4, device=dev, skm_range=(120, 360))
    cnt = np.diff(D.sk_off.astype(np.int64))
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    call = lambda: ctx.compare_device(k, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, d_inter.data_ptr())  # noqa: E731
    for _ in range(2):
        call()
    reps = 5
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    for _ in range(reps):
        call()
    t = ctx.timing_read()
    ctx.timing_enable(False)
    ms = t["compare_ms"] / max(1, t["compare_calls"])
    pairs = n * (n - 1) // 2
    total = int(cnt.sum())
    no_reuse = 8.0 * total * (n - 1) + 4.0 * pairs
    compulsory = 8.0 * total + 4.0 * pairs
    out = {"workload": "BASELINE configs[3] on one GPU: %d sketches synthesised directly (500 families x 20, mu 0.001/0.01/0.05), %d keys "
                       "(mean %.0f, min %d, max %d), k=31 m=11; keys resident in HBM" % (n, total, cnt.mean(), cnt.min(), cnt.max()),
           "pairs": pairs, "pipeline_ms": ms, "sketch_pairs_per_s": pairs / (ms / 1e3) if ms > 0 else None,
           "kernel_ms": {name: t[a] / max(1, t[b]) for name, a, b in (("k_parts_scatter", "scatter_ms", "scatter_launches"),
                                                                        ("k_parts_group", "group_ms", "group_launches"),
                                                                        ("k_accumulate_sparse", "accumulate_ms", "accumulate_launches"))},
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "peak_measured_read": peak.get("read_GBps"), "unit": "GB/s",
                        "no_reuse_model": {"bytes": no_reuse, "achieved": no_reuse / ms / 1e6, "frac": no_reuse / ms / 1e6 / HBM_PEAK_GBS,
                                           "model": "sum over pairs of 8 (n_i + n_j) + 4 bytes (SURVEY.md 8d); reuse pushes it above 1"},
                        "compulsory_model": {"bytes": compulsory, "achieved": compulsory / ms / 1e6, "frac": compulsory / ms / 1e6 / HBM_PEAK_GBS,
                                             "model": "8 sum(n_i) + 4 N(N-1)/2 bytes: every key read once, every cell written once"}},
           "setup_s": setup_s}
    out["roofline"].update(pmc_compare("r05_c4_pmc_hbm_traffic.json", total))
    # sampled pairs against numpy set intersections (whole matrix)
    rng = np.random.default_rng(0)
    mn_all, lo_all = D.minimizer, D.kmer_lo

    def keys_of(i):
        a, b = int(D.sk_off[i]), int(D.sk_off[i + 1])
        return set(zip(mn_all[a:b].cpu().numpy().tolist(), lo_all[a:b].cpu().numpy().tolist()))
    bad = checked = 0
    sample = [(20 * f + int(a), 20 * f + int(b)) for f in rng.integers(0, n // 20, size=150) for a, b in [sorted(rng.choice(20, 2, replace=False))]]
    sample += [tuple(sorted(int(x) for x in rng.integers(0, n, size=2))) for _ in range(150)]
    for i, j in sample:
        if i == j:
            continue
        checked += 1
        bad += int(d_inter[i, j].item()) != len(keys_of(i) & keys_of(j))
    nz = int(torch.count_nonzero(torch.triu(d_inter, 1)).item())
    out["parity_sampled_pairs"] = {"checked": checked, "mismatches": bad, "against": "numpy set intersections of the (minimizer, k-mer) keys",
                                   "nonzero_pairs": nz, "nonzero_pairs_expected": (n // 20) * 190}
    # the same size under other collection shapes: families of 200 / 2 000 / all 10 000 genomes of one species (parts overflow ->
    # spill to HBM, bit columns + popcounts) -- sampled pairs against torch set algebra on the keys
    try:
        out["collection_shapes"] = compare_config4_shapes(ctx, dev, n, d_inter)
    except Exception as e:  # noqa: BLE001
        out["collection_shapes"] = {"error": repr(e)}
    call()                                                # (d_inter holds configs[3]'s matrix again: the oracle check below reads it)
    pls = None
    try:
        out["compare_files"] = compare_files_config4(ctx, D, n, skip_oracle)
        pls = out["compare_files"].pop("_payloads", None)
    except Exception as e:  # noqa: BLE001
        out["compare_files"] = {"error": repr(e)}
    if not skip_oracle:
        from oracle import oracle_py as orc
        sub = 1600
        pls = pls[:sub] if pls else [D.payload(i) for i in range(sub)]
        want, card, sec = orc.compare(pls, timed=True)
        got = d_inter[:sub, :sub].cpu().numpy().astype(np.uint32)
        out["cpu_baseline"] = {"sketch_pairs_per_s": (sub * (sub - 1) // 2) / sec if sec > 0 else None, "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "oracle compare_sketches (Comparator.cpp:39-287 restated) over the first %d sketches written out in the "
                                         "sketch format (%d pairs), single thread, %.2f s" % (sub, sub * (sub - 1) // 2, sec)}
        out["parity_vs_oracle"] = {"sketches": sub, "pairs": sub * (sub - 1) // 2,
                                   "equal": bool((np.triu(got, 1) == np.triu(want, 1)).all() and (card == cnt[:sub]).all()),
                                   "nonzero_pairs": int(np.count_nonzero(np.triu(want, 1)))}
    return out


def compare_config4_shapes(ctx, dev, n, d_inter):
    """configs[3]'s size, other collection structures (tools/exp/c4_shapes.py): wall ms per spsp_compare_device call in steady
    state and 40 sampled pairs per shape (inside families and across) against set algebra on the keys."""
    res = {}
    rng = np.random.default_rng(5)
    for F, shuffled in ((20, True), (200, False), (2000, False), (n, False)):
        D = synth.direct_family_sketches(n, fam_size=F, seed=4, device=dev, skm_range=(120, 360))
        fam_of = np.arange(n) // F
        if shuffled:                                       # configs[3]'s own sketches in a random order (files are not listed family by family)
            perm = rng.permutation(n)
            off0 = D.sk_off.astype(np.int64)
            cnt = np.diff(off0)[perm]
            new_off = np.zeros(n + 1, np.int64)
            new_off[1:] = np.cumsum(cnt)
            src = torch.from_numpy(np.repeat(off0[:-1][perm] - new_off[:-1], cnt)).to(dev) + torch.arange(int(new_off[-1]), device=dev)
            D.minimizer, D.kmer_lo, D.sk_off = D.minimizer[src].contiguous(), D.kmer_lo[src].contiguous(), new_off.astype(np.uint64)
            fam_of = perm // F
            del src
        torch.cuda.synchronize()
        call = lambda: ctx.compare_device(31, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, d_inter.data_ptr())  # noqa: E731
        for _ in range(2):
            call()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        ms = (time.perf_counter() - t0) * 1e3 / 3
        key = torch.stack([D.minimizer.to(torch.int64), D.kmer_lo.view(torch.int64)], 1)
        off = D.sk_off.astype(np.int64)
        pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, 20), rng.integers(0, n, 20))]
        for a in rng.integers(0, n, 20):                   # pairs inside a family
            mates = np.nonzero(fam_of == fam_of[int(a)])[0]
            pairs.append((int(a), int(mates[int(rng.integers(0, len(mates)))])))
        wrong = checked = 0
        for a, b in pairs:
            if a == b:
                continue
            i, j = min(a, b), max(a, b)
            ka, kb = key[off[i]:off[i + 1]], key[off[j]:off[j + 1]]
            want = ka.shape[0] + kb.shape[0] - torch.cat([ka, kb]).unique(dim=0).shape[0]
            wrong += int(d_inter[i, j].item()) != want
            checked += 1
        res["families_of_%d%s" % (F, "_in_random_order" if shuffled else "")] = {"keys": int(D.sk_off[-1]), "wall_ms_per_call": ms, "sketch_pairs_per_s": n * (n - 1) / 2 / (ms / 1e3),
                                     "nonzero_pairs": int(torch.count_nonzero(torch.triu(d_inter, 1)).item()), "sampled_pairs": checked, "sampled_pairs_wrong": wrong}
        del D, key
    return res


def compare_files_config4(ctx, D, n, skip_oracle):
    """What the reference's only timer brackets (Comparator.cpp:499-509: compare_sketches + both CSV dumps) at BASELINE
    configs[3]: the 10 000 sketches of the leg above written out as sketch files (tmpfs), then spsp_compare_files -- read +
    gunzip on the host threads, decode + all-vs-all on the GPU (the pair matrix comes back as its non-zero cells, not as
    400 MB), both 10^8-cell matrices formatted and gzipped (level 1, as the reference) -- with the seconds per stage; the
    CSV bytes of a run over the first 1 600 files are held against the oracle's comparator + printers."""
    import gzip
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="spsp_c4_", dir=base)
    try:
        t0 = time.perf_counter()
        pls = [D.payload(i) for i in range(n)]
        paths = []
        for i, pl in enumerate(pls):
            pth = os.path.join(tmp, "s%05d.gz" % i)
            sp.write_gz(pth, pl, 1)
            paths.append(pth)
        write_s = time.perf_counter() - t0
        ctx.compare_files(paths[:64], os.path.join(tmp, "warm"))
        # twice: the first call at this size allocates the decoder's device buffers and the block the files are read into
        # (first_call_wall_s; it also takes whatever the shared host is doing to a 40 MB pageable upload: 0.06 - 0.12 s run to
        # run), the second is what every later call of the process costs
        walls = []
        for _ in range(2):
            ctx.stage_times(reset=True)
            t0 = time.perf_counter()
            ctx.compare_files(paths, os.path.join(tmp, "all"))
            walls.append(time.perf_counter() - t0)
            st = ctx.stage_times(reset=True)
        wall = walls[1]
        sizes = {name: os.path.getsize(os.path.join(tmp, "all_%s.csv.gz" % name)) for name in ("jaccard", "containment")}
        out = {"workload": "the %d sketches as gzip sketch files on %s (%.1f MB) -> <o>_jaccard.csv.gz + <o>_containment.csv.gz (10^8 cells each)"
                           % (n, base or "the temp dir", sum(os.path.getsize(x) for x in paths) / 1e6),
               "wall_s": wall, "first_call_wall_s": walls[0], "sketch_pairs_per_s": n * (n - 1) / 2 / wall,
               "stage_s": {"read_gunzip": st["load_s"], "decode_compare_cells_to_host": st["compare_s"], "format_both_matrices": st["csv_s"],
                           "gzip_and_write": st["csv_gzip_s"]},
               "csv_gz_bytes": sizes, "files_written_s": write_s}
        # sortCSV (sort_csv.cpp:26-111) over the 10^8-cell Jaccard matrix: rows and columns back into another file-of-files order
        try:
            jac_text = sp.read_file(os.path.join(tmp, "all_jaccard.csv.gz"))
            fof = ("\n".join(paths[i] for i in np.random.default_rng(9).permutation(n)) + "\n").encode()
            t0 = time.perf_counter()
            sorted_text = sp.sort_csv(jac_text, fof)
            out["sort_csv"] = {"seconds": time.perf_counter() - t0, "csv_bytes": len(jac_text), "same_size_out": len(sorted_text) == len(jac_text)}
            del jac_text, sorted_text
        except Exception as e:  # noqa: BLE001
            out["sort_csv"] = {"error": repr(e)}
        if not skip_oracle:
            from oracle import oracle_py as orc
            sub = 1600
            ctx.compare_files(paths[:sub], os.path.join(tmp, "sub"))
            inter, card, _, _ = orc.compare(pls[:sub])
            sub_text = sp.read_file(os.path.join(tmp, "sub_jaccard.csv.gz"))
            sub_fof = ("\n".join(paths[i] for i in np.random.default_rng(10).permutation(sub)) + "\n").encode()
            out["sort_csv"]["equal_to_the_oracle_on_%d_files" % sub] = bool(sp.sort_csv(sub_text, sub_fof) == orc.sort_csv(sub_text, sub_fof))
            out["csv_bytes_equal_the_oracle_on_%d_files" % sub] = bool(all(
                gzip.open(os.path.join(tmp, "sub_%s.csv.gz" % name), "rb").read() == orc.csv(jac, paths[:sub], inter, card, None, 6, 0.0)
                for jac, name in ((True, "jaccard"), (False, "containment"))))
        out["_payloads"] = pls
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def compare_config4_dist(ctx, dev, stream, rank, world, backend):
    """BASELINE configs[3] as it is stated: 10 000 sketches all-vs-all over the ranks of this job (what
    Comparator::compare_sketches, Comparator.cpp:39-74, does on one thread).  Every rank synthesises the same 10 000 sketches
    (the single-GPU leg's, seed 4) and keeps the block it would have sketched; the comparison is split BY KEY (spsp_multi.hip,
    DESIGN.md 5): each rank deals its keys into one slot per rank (spsp_partition_keys_device), ONE RCCL all-to-all moves
    them (each key crosses the fabric once), each rank compares every sketch's keys of its hash class with all rows owned
    and emits the non-zero cells of its partial matrix straight from the row sums (spsp_compare_slots_cells_device), the
    cells are all-gathered (95 000 per rank, 0.76 MB) and added up on every rank.  Timed with a barrier + device
    synchronisation on both sides of every stage, MAX over ranks; checked against ONE device's comparison of all sketches
    on rank 0.  BENCH_C4_SKETCHES overrides the count (rehearsals)."""
    n = int(os.environ.get("BENCH_C4_SKETCHES", "10000"))
    n -= n % (20 * world)
    per = n // world
    k, m = 31, 11
    t0 = time.time()
    D = synth.direct_family_sketches(n, fam_size=20, k=k, m=m, seed=4, device=dev, skm_range=(120, 360))
    off = D.sk_off.astype(np.int64)
    a, b = int(off[rank * per]), int(off[(rank + 1) * per])
    my_mn, my_lo = D.minimizer[a:b].contiguous(), D.kmer_lo[a:b].contiguous()
    my_off = (off[rank * per:(rank + 1) * per + 1] - a).astype(np.uint64)
    ex = spd.SlotExchange(ctx, k, per, int(my_off[-1]), dev, stream=stream, reduce="cells")
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def step(times=None):
        fence(); t_a = time.perf_counter()
        h = ex.begin(my_mn.data_ptr(), my_lo.data_ptr(), None, my_off)       # partition + all-to-all
        ex.end_queue(h, d_inter)
        fence(); t_b = time.perf_counter()
        ex.end_collect(d_inter)                                                 # comparison -> cells -> all-gather -> sum
        fence(); t_c = time.perf_counter()
        if times is not None:
            times.append((t_b - t_a, t_c - t_b))
    for _ in range(2):
        step()
    reps, times = 5, []
    for _ in range(reps):
        step(times)
    if ex.overflowed(d_inter):
        return {"error": "exchange slots overflowed"}
    loc = torch.tensor([sum(t[0] for t in times) / reps, sum(t[1] for t in times) / reps], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(loc, op=dist.ReduceOp.MAX)
    ex_ms, cmp_ms = float(loc[0].item()) * 1e3, float(loc[1].item()) * 1e3
    pairs = n * (n - 1) // 2
    total = int(off[-1])
    out = None
    if rank == 0:
        # untimed: ONE device's comparison of all the sketches (what the gpu tests hold against the oracle at this shape)
        d_ref = torch.zeros((n, n), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.compare_device(k, D.minimizer.data_ptr(), D.kmer_lo.data_ptr(), None, D.sk_off, n, 0, 1, d_ref.data_ptr())
        torch.cuda.synchronize()
        got = torch.triu(d_inter, 1)
        same = bool(torch.equal(got, torch.triu(d_ref, 1)))
        ms = ex_ms + cmp_ms
        out = {"workload": "BASELINE configs[3]: %d sketches (500 families x 20, mu 0.001/0.01/0.05, the single-GPU leg's), %d keys, k=31 m=11, "
                           "all-vs-all split by key over %d ranks (%d sketches and their keys per rank to begin with)" % (n, total, world, per),
               "pairs": pairs, "ms": ms, "sketch_pairs_per_s": pairs / (ms / 1e3),
               "stage_ms": {"partition_and_all_to_all": ex_ms, "compare_cells_all_gather_sum": cmp_ms},
               "slot_bytes_per_peer": int(ex.slot_bytes), "keys_on_the_wire_per_rank_bytes": int(ex.slot_bytes) * (world - 1),
               "nonzero_pairs": int(torch.count_nonzero(got).item()),
               "exchange_check": ("sum of the ranks' partial matrices equals a one-device comparison of all %d sketches" % n) if same else "MISMATCH",
               "timing": "barrier + device synchronisation around both stages, mean of %d, max over ranks" % reps, "setup_s": setup_s}
    del d_inter
    return out


def scan_config5(ctx, dev, skip_oracle, peak, gbp=4.0):
    """BASELINE configs[4] shape: k=63 m=15 s=100 (fine sampling, long super-k-mers), records of 10^6 bp generated on
    the GPU (seed 5), `gbp` Gbp per scan call -- the 50 Gbp of the config streamed through HBM segment by segment is
    tools/c5_scan.py; this leg measures one segment: the scan loop of SubSampler.cpp:367-455 over it."""
    k, m, s = 63, 15, 100.0
    rec_len = 1_000_000
    n_rec = int(gbp * 1e9) // rec_len
    seg_n = n_rec * rec_len
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    bases = torch.empty(seg_n + 64, dtype=torch.uint8, device=dev)
    bases[seg_n:] = 65
    for a in range(0, seg_n, 1 << 27):
        b = min(seg_n, a + (1 << 27))
        bases[a:b] = lut[torch.randint(0, 4, (b - a,), device=dev, generator=g, dtype=torch.int64)]
    off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * rec_len
    torch.cuda.synchronize()
    p = sp.make_params(k, m, s)
    ctx.scan_device(p, bases.data_ptr(), seg_n, off.data_ptr(), n_rec)          # tables, buffers
    ctx.timing_enable(True, sp.TIME_DENSE | sp.TIME_SCAN)
    ctx.timing_read()
    reps = 3
    for _ in range(reps):
        d_out, n_out = ctx.scan_device(p, bases.data_ptr(), seg_n, off.data_ptr(), n_rec)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    dense_ms = t["dense_ms"] / max(1, t["dense_launches"])
    scan_ms = t["scan_ms"] / max(1, t["scan_calls"])
    kmers = seg_n - n_rec * (k - 1)
    sk = ctx.to_host(d_out, n_out, sp.SUPERKMER_DTYPE)
    rec, start, ln = sk["rec"].astype(np.int64), sk["start"].astype(np.int64), sk["len"].astype(np.int64)
    same = rec[1:] == rec[:-1]
    inv = bool((np.diff(rec) >= 0).all() and (ln >= k).all() and (start + ln <= rec_len).all() and (ln <= 2 * k - m).all()
               and (start[1:][same] >= start[:-1][same] + ln[:-1][same] - k + 1).all())
    sel = int((ln - k + 1).sum())
    gbs = seg_n / 1e9 / (dense_ms / 1e3)
    out = {"workload": "BASELINE configs[4] shape: %d records x 10^6 bp = %.1f Gbp generated on the GPU (seed 5), k=63 m=15 s=100; "
                       "one spsp_scan_device call, input resident in HBM (%.1f GB: 16 x the Infinity Cache)" % (n_rec, seg_n / 1e9, seg_n / 1e9),
           "kmers": kmers, "scan_pipeline_ms": scan_ms, "dense_kernel_ms": dense_ms, "launches_timed": int(t["dense_launches"]),
           "kmers_per_s": kmers / (scan_ms / 1e3), "kmers_per_s_dense_kernel": kmers / (dense_ms / 1e3),
           "superkmers": int(n_out), "selected_kmers": sel, "selected_over_expected": sel / (kmers / s), "stream_invariants_ok": inv,
           "roofline": {"kernel": "k_dense_bloom<15> (blocked Bloom filter over canonical 15-mers in LDS; XXH64 on survivors)", "bound": "hbm",
                        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "frac_of_measured_read": gbs / peak["read_GBps"] if peak.get("read_GBps") else None,
                        "algorithmic_bytes_per_launch": seg_n, "byte_model": "1 B per m-mer position (cleaned ASCII)"}}
    out["roofline"].update(sq_counters("k_dense_bloom", dense_ms, 0, "r05_pmc_sq_c5.json"))
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r05_c5_pmc_hbm_traffic.json")))
        kn = [x for x in d["kernels"] if x.startswith("k_dense_bloom")][0]
        out["roofline"]["traffic"] = d["kernels"][kn]["hbm_bytes_per_launch_corrected"] * (seg_n / d["workload"]["bases_per_launch"])
        out["roofline"]["traffic_provenance"] = {"file": "profiles/r05_c5_pmc_hbm_traffic.json", "commit": d.get("commit"),
                                                 "scaled_from_bases_per_launch": d["workload"]["bases_per_launch"]}
    except Exception:  # noqa: BLE001
        out["roofline"]["traffic"] = None
    # the same segment as 2-bit words (what the FASTA ingest hands the scan: k_clean_write<PACK>): k_dense_bloom<15, true> reads
    # them directly since round 5 -- no ASCII copy made on the device in front of the pass
    try:
        pp = sp.make_params(k, m, s, flags=sp.SPSP_SCAN_PACKED_INPUT)
        d_pk = ctx.pack_bases_device(bases.data_ptr(), seg_n)
        torch.cuda.synchronize()
        ctx.scan_device(pp, d_pk, seg_n, off.data_ptr(), n_rec)
        ctx.timing_enable(True, sp.TIME_DENSE | sp.TIME_SCAN)
        ctx.timing_read()
        for _ in range(reps):
            d_out_p, n_out_p = ctx.scan_device(pp, d_pk, seg_n, off.data_ptr(), n_rec)
        tp = ctx.timing_read()
        ctx.timing_enable(False)
        skp = ctx.to_host(d_out_p, n_out_p, sp.SUPERKMER_DTYPE)
        dms, sms = tp["dense_ms"] / max(1, tp["dense_launches"]), tp["scan_ms"] / max(1, tp["scan_calls"])
        out["packed_2bit"] = {"dense_kernel_ms": dms, "scan_pipeline_ms": sms, "kmers_per_s": kmers / (sms / 1e3),
                              "byte_model": "0.25 B per m-mer position", "achieved_GBps": seg_n / 4 / 1e9 / (dms / 1e3),
                              "same_stream_as_ascii": bool(n_out_p == n_out and all((skp[f] == sk[f]).all() for f in sk.dtype.names))}
        d_out, n_out = ctx.scan_device(p, bases.data_ptr(), seg_n, off.data_ptr(), n_rec)      # (the key extraction below reads the ASCII scan's output)
    except Exception as e:  # noqa: BLE001
        out["packed_2bit"] = {"error": repr(e)}
    # From the scan to the comparator's keys: the segment as ONE sketch -- a metagenome file is one
    # sketch -- through spsp_sketch_keys_device: ~4 x 10^7 selected k-mers, four orders of magnitude beyond a workgroup's LDS:
    # the table in HBM of spsp_bigkeys.hip (and, in the sorted form, its merge sort).  Host wall clock around the call.
    try:
        one = np.array([0, n_rec], dtype=np.uint32)
        res = {}
        for name, un in (("unordered", True), ("sorted", False)):
            ctx.sketch_keys_device(p, bases.data_ptr(), seg_n, off.data_ptr(), d_out, n_out, one, unordered=un)      # buffers, table
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, _, koff = ctx.sketch_keys_device(p, bases.data_ptr(), seg_n, off.data_ptr(), d_out, n_out, one, unordered=un)
            res[name + "_ms"] = (time.perf_counter() - t0) * 1e3
            res[name + "_keys"] = int(koff[1])
        out["sketch_keys"] = {"what": "spsp_sketch_keys_device over the scan's output, all %d records as one sketch (handle_superkmer's counts, the emission / "
                                      "reader round trip and canonize composed; SubSampler.cpp:243-302,458-620, Comparator.cpp:186-260)" % n_rec,
                              "sketch_keys_ms": res["unordered_ms"], "sketch_keys_sorted_ms": res["sorted_ms"], "keys": res["unordered_keys"],
                              "forms_agree_on_the_count": res["unordered_keys"] == res["sorted_keys"], "selected_kmer_occurrences": sel,
                              "genomes_through_the_table_in_hbm": ctx.sketch_keys_big_genomes()}
    except Exception as e:  # noqa: BLE001
        out["sketch_keys"] = {"error": repr(e)}
    # FILE -> SKETCH at this shape: the segment as ONE FASTA file on tmpfs through the file pipeline
    # (spsp_sketch_files: read, ingest to 2-bit words, scan, gather, the host's sketch build -- handle_superkmer + emission,
    # SubSampler.cpp:243-302, 458-620 -- gzip, write), stage seconds from the library
    try:
        out["sketch_file"] = sketch_file_config5(ctx, bases, n_rec, rec_len, k, m, s, skip_oracle)
    except Exception as e:  # noqa: BLE001
        out["sketch_file"] = {"error": repr(e)}
    if not skip_oracle:
        from oracle import oracle_py as orc
        orc.use_native()
        R = 12
        hb = bases[:R * rec_len].cpu().numpy()
        ho = np.arange(0, R + 1, dtype=np.uint64) * rec_len
        sec, km, nem = orc.scan_timed(k, m, p.threshold, hb, ho)
        want, _ = orc.scan(k, m, p.threshold, hb, ho)
        mine = sk[sk["rec"] < R]
        out["cpu_baseline"] = {"value": km / sec if sec > 0 else None, "unit": "k-mers hashed/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "oracle scan loop (SubSampler.cpp:357-455 restated), single thread, the first %d records "
                                         "(%d k-mers), %.2f s" % (R, km, sec)}
        out["parity_vs_oracle"] = {"records": R, "superkmers": int(len(want)),
                                   "equal": bool(len(mine) == len(want) and all((mine[f] == want[f]).all() for f in want.dtype.names))}
        try:
            # ... and their keys: the first R records as one sketch (1.2 x 10^5 selected k-mers: the table in HBM + the merge sort)
            # against the oracle's comparator walk (orc_sketch_keys) over the oracle's sketch of the same records
            text = b"".join(b">r%d\n" % r + hb[r * rec_len:(r + 1) * rec_len].tobytes() + b"\n" for r in range(R))
            _, _, w_mn, w_lo, w_hi = orc.sketch_keys(orc.sketch_fasta(text, k, m, s)[0])
            d_mn, d_lo, d_hi, koff = ctx.sketch_keys_device(p, bases.data_ptr(), seg_n, off.data_ptr(), d_out, n_out, np.array([0, R], dtype=np.uint32))
            tot = int(koff[1])
            out["parity_vs_oracle"]["sketch_keys"] = {"keys": tot, "equal": bool(tot == len(w_mn) and (ctx.to_host(d_mn, tot, np.uint32) == w_mn).all()
                                                                                 and (ctx.to_host(d_lo, tot, np.uint64) == w_lo).all()
                                                                                 and (ctx.to_host(d_hi, tot, np.uint64) == w_hi).all())}
        except Exception as e:  # noqa: BLE001
            out["parity_vs_oracle"]["sketch_keys"] = {"error": repr(e)}
    # configs[4] at its STATED size: 50 Gbp streamed through HBM, segment after segment generated in place on the device (seeded;
    # the generator goes on from the segment above), one spsp_scan_device call each with the kernel brackets; what is summed is
    # the scan, not the generator (tools/c5_scan.py is the same loop as a tool)
    try:
        total_bp = 50_000_000_000
        done, stream_ms, dense_total, sk_total, sel_total, km_total, inv_all, n_calls = 0, 0.0, 0.0, 0, 0, 0, True, 0
        t_wall = time.time()
        while done < total_bp:
            take_rec = min(n_rec, (total_bp - done) // rec_len)
            take = take_rec * rec_len
            for a in range(0, take, 1 << 27):
                b = min(take, a + (1 << 27))
                bases[a:b] = lut[torch.randint(0, 4, (b - a,), device=dev, generator=g, dtype=torch.int64)]
            torch.cuda.synchronize()
            ctx.timing_enable(True, sp.TIME_DENSE | sp.TIME_SCAN)
            ctx.timing_read()
            d_o, n_o = ctx.scan_device(p, bases.data_ptr(), take, off.data_ptr(), take_rec)
            t = ctx.timing_read()
            ctx.timing_enable(False)
            stream_ms += t["scan_ms"]; dense_total += t["dense_ms"]; n_calls += 1
            ln2 = ctx.to_host(d_o, n_o, sp.SUPERKMER_DTYPE)["len"].astype(np.int64)
            sk_total += int(n_o); sel_total += int((ln2 - k + 1).sum()); km_total += take - take_rec * (k - 1)
            inv_all = inv_all and bool((ln2 >= k).all() and (ln2 <= 2 * k - m).all())
            done += take
        out["stream_50gbp"] = {"workload": "BASELINE configs[4]: %.0f Gbp as %d segments of <= %d records x 10^6 bp, k=63 m=15 s=100, generated on the GPU in place" % (done / 1e9, n_calls, n_rec),
                               "kmers": km_total, "scan_pipeline_ms_total": stream_ms, "dense_kernel_ms_total": dense_total,
                               "kmers_per_s": km_total / (stream_ms / 1e3), "dense_kernel_frac_of_8TBps": done / 1e9 / (dense_total / 1e3) / HBM_PEAK_GBS,
                               "superkmers": sk_total, "selected_over_expected": sel_total / (km_total / s), "superkmer_lengths_ok": inv_all,
                               "wall_s_with_generation": time.time() - t_wall}
    except Exception as e:  # noqa: BLE001
        out["stream_50gbp"] = {"error": repr(e)}
    return out


def sketch_file_config5(ctx, bases, n_rec, rec_len, k, m, s, skip_oracle):
    """one FASTA file of n_rec records x rec_len bp (the configs[4] segment) -> one sketch file; payload bytes against the
    oracle's on a file of the first 100 records"""
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="spsp_c5_", dir=base)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("BENCH_HOST_THREADS", "16"))))
    try:
        def write(path, recs):
            with open(path, "wb") as f:
                for a in range(0, recs, 250):                # 250 Mbp of the segment at a time through the host
                    hb = bases[a * rec_len:min(recs, a + 250) * rec_len].cpu().numpy()
                    for r in range(len(hb) // rec_len):
                        f.write(b">r%d\n" % (a + r))
                        f.write(hb[r * rec_len:(r + 1) * rec_len].tobytes())
                        f.write(b"\n")
            return os.path.getsize(path)
        big, small = os.path.join(tmp, "segment.fa"), os.path.join(tmp, "prefix.fa")
        size = write(big, n_rec)
        R = min(100, n_rec)
        write(small, R)
        kmers = n_rec * (rec_len - k + 1)
        sp.sketch_files([small], [os.path.join(tmp, "prefix.gz")], k, m, s, threads=cores)          # buffers, tables, page cache of the small file
        sp.sketch_files([big], [os.path.join(tmp, "segment.gz")], k, m, s, threads=cores)            # the slot's pinned slab and device buffers at this size (0.2-0.4 s once per process)
        res = {}
        for label, T in (("threads_%d" % cores, cores), ("threads_1", 1)):
            t0 = time.perf_counter()
            r, st, _ = sp.sketch_files([big], [os.path.join(tmp, "segment.gz")], k, m, s, threads=T)
            wall = time.perf_counter() - t0
            assert r[0][0] == 0, r[0]
            res[label] = {"wall_s": wall, "kmers_per_s": kmers / wall, "fasta_GB_per_s": size / wall / 1e9,
                          "stage_s_summed_over_workers": {key: st[key] for key in ("read_s", "ingest_s", "scan_s", "gather_s", "build_s", "gzip_s")},
                          "selected_kmers": r[0][1]["selected_kmer_number"], "sketch_bytes": os.path.getsize(os.path.join(tmp, "segment.gz"))}
        out = {"workload": "the segment as ONE FASTA file on %s: %d records x %d bp (%.2f GB), k=%d m=%d s=%g -> one sketch file (spsp_sketch_files, the "
                           "library behind bin/sub_sampler)" % (base or "the temp dir", n_rec, rec_len, size / 1e9, k, m, s),
               "kmers": kmers, "host_threads_available": cores, **res}
        if not skip_oracle:
            from oracle import oracle_py as orc
            orc.use_native()
            want = orc.sketch_fasta(open(small, "rb").read(), k, m, s)[0]
            out["parity_vs_oracle"] = {"records": R, "payload_bytes": len(want), "equal": sp.read_file(os.path.join(tmp, "prefix.gz")) == want}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config3_true_shape(ctx, dev, n, with_oracle, seed=3, host_threads=None, unordered=False):
    """BASELINE configs[2] at its TRUE shape (SURVEY.md 8d "C3"): n genomes of L ~ U[2, 8] Mbp in families of 20 at mu
    0.001 / 0.01 / 0.05, generated on the device batch by batch (synth.device_family_batches), k=31 m=11 s=1000 -- sketched
    by the HIP path: ONE spsp_scan_device per batch of 100 genomes and spsp_sketch_keys_device from its super-k-mer stream
    to the comparator's keys, no file and no host code in between.  unordered: the form the pipelined step uses (an LDS
    table per genome); there a fifth of the genomes (> 6.7 Mbp: more than 6144 k-mer places) take the table in HBM
    (spsp_bigkeys.hip); the sorted form holds 8192 per genome, which this shape does not reach.  with_oracle: every
    batch is also copied to the host and sketched by the oracle (Subsampler::parse_fasta_test restated) on `host_threads`
    threads, one genome per task like the reference's OpenMP loop.  Returns the keys of all n sketches resident on the
    device, their offsets, the oracle's payloads and the timings."""
    import concurrent.futures as cf
    k, m, s = 31, 11, 1000.0
    p = sp.make_params(k, m, s)
    threads = host_threads or max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("BENCH_HOST_THREADS", "16"))))
    mins, los, cnts, payloads = [], [], [], []
    scan_ms = keys_ms = oracle_s = 0.0
    big = bases_total = kmers = 0
    if with_oracle:
        from oracle import oracle_py as orc
        orc.use_native()
    for B in synth.device_family_batches(n, seed, dev):
        nb, n_rec = B["n_bases"], len(B["rec_off"]) - 1
        d_off = torch.from_numpy(B["rec_off"].view(np.int64)).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d_sk, n_sk = ctx.scan_device(p, B["bases"].data_ptr(), nb, d_off.data_ptr(), n_rec)
        t1 = time.perf_counter()
        d_mn, d_lo, _, koff = ctx.sketch_keys_device(p, B["bases"].data_ptr(), nb, d_off.data_ptr(), d_sk, n_sk, B["first_rec"], unordered=unordered)
        t2 = time.perf_counter()
        scan_ms += (t1 - t0) * 1e3
        keys_ms += (t2 - t1) * 1e3
        big += ctx.sketch_keys_big_genomes()
        tot = int(koff[-1])
        mins.append(device_bytes_as_tensor(d_mn, 4 * tot, dev).clone())
        los.append(device_bytes_as_tensor(d_lo, 8 * tot, dev).clone())
        cnts.append(np.diff(koff.astype(np.int64)))
        bases_total += nb
        kmers += int(sum(max(0, int(B["rec_off"][r + 1] - B["rec_off"][r]) - k + 1) for r in range(n_rec)))
        if with_oracle:
            host = B["bases"][:nb].cpu().numpy()
            ro, fr = B["rec_off"], B["first_rec"]

            def one(g):
                text = b"".join(b">r%d\n" % r + host[int(ro[r]):int(ro[r + 1])].tobytes() + b"\n" for r in range(int(fr[g]), int(fr[g + 1])))
                return orc.sketch_fasta(text, k, m, s)[0]
            t0 = time.perf_counter()
            with cf.ThreadPoolExecutor(threads) as ex:
                payloads += list(ex.map(one, range(len(fr) - 1)))
            oracle_s += time.perf_counter() - t0
            del host
        del B
    cnt = np.concatenate(cnts)
    sk_off = np.zeros(n + 1, np.uint64)
    sk_off[1:] = np.cumsum(cnt)
    return {"k": k, "m": m, "s": s, "d_min": torch.cat(mins).view(torch.int32), "d_lo": torch.cat(los).view(torch.int64), "sk_off": sk_off, "cnt": cnt,
            "payloads": payloads if with_oracle else None, "big_genomes": big, "bases": bases_total, "kmers": kmers,
            "scan_ms": scan_ms, "keys_ms": keys_ms, "oracle_sketch_s": oracle_s, "oracle_threads": threads}


def compare_config3(ctx, dev, skip_oracle):
    """BASELINE configs[2] at its true shape: 1000 RefSeq-like genomes, L ~ U[2, 8] Mbp, 50 families of 20 at mu
    0.001 / 0.01 / 0.05, k=31 m=11 s=1000 (config3_true_shape: 5 Gbp generated on the device and sketched by the HIP path),
    all-vs-all on one GPU, keys resident in HBM; every pair and every cardinality against the oracle, which sketches the
    same genomes on the host threads and compares them on one."""
    n = 1000
    t0 = time.time()
    T = config3_true_shape(ctx, dev, n, not skip_oracle, unordered=True)
    ctx.compare_keys_unordered(True)
    k, m, s = T["k"], T["m"], T["s"]
    d_min, d_lo, sk_off, cnt, payloads = T["d_min"], T["d_lo"], T["sk_off"], T["cnt"], T["payloads"]
    d_inter = torch.zeros((n, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    call = lambda: ctx.compare_device(k, d_min.data_ptr(), d_lo.data_ptr(), None, sk_off, n, 0, 1, d_inter.data_ptr())  # noqa: E731
    for _ in range(3):
        call()
    reps = 20
    ctx.timing_enable(True, sp.TIME_ALL)
    ctx.timing_read()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    t = ctx.timing_read()
    ctx.timing_enable(False)
    ms = t["compare_ms"] / max(1, t["compare_calls"])
    pairs = n * (n - 1) // 2
    total = int(cnt.sum())
    no_reuse = 8.0 * total * (n - 1) + 4.0 * pairs           # sum over pairs of 8 (n_i + n_j) + 4
    compulsory = 8.0 * total + 4.0 * pairs
    out = {"workload": "BASELINE configs[2] at its true shape: %d genomes of L ~ U[2, 8] Mbp (%.2f Gbp, 50 families x 20, mu 0.001/0.01/0.05) generated "
                       "on the device, k=31 m=11 s=1000, sketched by the HIP path (scan + keys on the device, %d genomes beyond the per-genome "
                       "LDS table through the table in HBM); %d keys (mean %.0f, min %d, max %d); keys resident in HBM"
                       % (n, T["bases"] / 1e9, T["big_genomes"], total, cnt.mean(), cnt.min(), cnt.max()),
           "sketching": {"what": "per batch of 100 genomes: ONE spsp_scan_device + ONE spsp_sketch_keys_device (SPSP_KEYS_UNORDERED, the form of the timed step), host wall ms summed over the 10 batches",
                         "scan_ms": T["scan_ms"], "sketch_keys_ms": T["keys_ms"], "kmers": T["kmers"], "kmers_per_s": T["kmers"] / ((T["scan_ms"] + T["keys_ms"]) / 1e3),
                         "genomes_through_the_table_in_hbm": T["big_genomes"],
                         **({"oracle_sketch_s": T["oracle_sketch_s"], "oracle_threads": T["oracle_threads"]} if payloads is not None else {})},
           "pairs": pairs, "pipeline_ms": ms, "host_wall_ms_per_call": wall_ms,
           "sketch_pairs_per_s": pairs / (ms / 1e3) if ms > 0 else None,
           "kernel_ms": {"k_parts_scatter": t["scatter_ms"] / max(1, t["scatter_launches"]),
                         "k_parts_group": t["group_ms"] / max(1, t["group_launches"]),
                         "k_accumulate_sparse": t["accumulate_ms"] / max(1, t["accumulate_launches"])},
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "no_reuse_model": {"bytes": no_reuse, "achieved": no_reuse / ms / 1e6 if ms > 0 else None,
                                           "frac": no_reuse / ms / 1e6 / HBM_PEAK_GBS if ms > 0 else None,
                                           "model": "sum over pairs of 8 (n_i + n_j) + 4 bytes (SURVEY.md 8d); reuse pushes it above 1"},
                        "compulsory_model": {"bytes": compulsory, "achieved": compulsory / ms / 1e6 if ms > 0 else None,
                                             "frac": compulsory / ms / 1e6 / HBM_PEAK_GBS if ms > 0 else None,
                                             "model": "8 sum(n_i) + 4 N(N-1)/2 bytes: every key read once, every cell written once"}},
           "setup_s": setup_s}
    out["roofline"].update(pmc_compare("r05_compare_pmc_hbm_traffic.json", total))
    ctx.compare_keys_unordered(False)
    if not skip_oracle:
        from oracle import oracle_py as orc
        want, card, sec = orc.compare(payloads, timed=True)
        got = d_inter.cpu().numpy().astype(np.uint32)
        out["cpu_baseline"] = {"sketch_pairs_per_s": pairs / sec if sec > 0 else None, "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "oracle compare_sketches (Comparator.cpp:39-287 restated) over the same %d sketch payloads, "
                                         "single thread, %.2f s" % (n, sec)}
        out["parity_vs_oracle"] = bool((np.triu(got, 1) == np.triu(want, 1)).all() and (card == cnt).all())
        out["nonzero_pairs"] = int(np.count_nonzero(want))
    return out


def pmc_compare(name="r05_compare_pmc_hbm_traffic.json", n_keys=None):
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        tr = {kname: v["hbm_bytes_per_launch_corrected"] for kname, v in d["kernels"].items() if kname.startswith(("k_parts", "k_accumulate"))}
        out = {"traffic": tr, "traffic_total": sum(tr.values()),
               "traffic_provenance": {"file": "profiles/" + name, "commit": d.get("commit"), "command": d.get("command"),
                                      "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; per-kernel correction as "
                                                "calibrated in profiles/r03_fetch_calibration.md (tools/pmc_traffic.py)"}}
        if n_keys:
            out["traffic_bytes_per_key"] = sum(tr.values()) / n_keys
        return out
    except Exception:
        return {"traffic": None}


def end_to_end(ctx, skip_oracle, genomes=None):
    """The whole-file drivers, host I/O and PCIe included, at BASELINE configs[1] size: the 100 genomes of the step as
    FASTA files on tmpfs -> sketch files (spsp_sketch_files: the reference's OpenMP loop over the file of files,
    SubSampler.cpp:771-793, with 1, 8 and all host threads; spsp_sketch_file on one context for the per-stage times),
    spsp_compare_files over the sketches, the two CLIs as processes -- beside the oracle (Subsampler::parse_fasta_test
    SubSampler.cpp:306-510 restated, one file per thread as the reference's OpenMP loop) at the SAME thread counts."""
    import concurrent.futures as cf
    import gzip
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="spsp_e2e_", dir=base)
    # "all": the host threads this job may use -- a one-GPU box shares its host with seven other GPUs' jobs (16 of its
    # threads per GPU), so the pool is capped there even when the affinity mask shows every core
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("BENCH_HOST_THREADS", "16"))))
    try:
        gs = genomes if genomes is not None else synth.family_genomes(7, 16, GENOME_LEN, 2, MUS)
        n = len(gs)
        plain, texts = [], []
        for i, g in enumerate(gs):
            data = synth.to_fasta(g, "g%d" % i, n_records=1 + i % 3)
            texts.append(data)
            pth = os.path.join(tmp, "genome%03d.fa" % i)
            open(pth, "wb").write(data)
            plain.append(pth)
        n_gz = min(16, n)
        gz = []
        for i in range(n_gz):
            open(plain[i] + ".gz", "wb").write(gzip.compress(texts[i], 1))
            gz.append(plain[i] + ".gz")
        kmers = sum(len(g) - K + 1 for g in gs)
        kmers_gz = sum(len(g) - K + 1 for g in gs[:n_gz])
        fasta_bytes = sum(len(t) for t in texts)
        out = {"workload": "%d synthetic %d bp genomes (the step's batch) as FASTA files on %s, k=31 m=11 s=1000" % (n, len(gs[0]), base or "the temp dir"),
               "kmers": kmers, "fasta_bytes": fasta_bytes, "host_threads_available": cores}
        # ---- one context, one host thread: where the time of ONE file goes
        ctx.sketch_file(plain[0], os.path.join(tmp, "warm.gz"), K, M, S)   # buffers, tables
        for label, files, km in (("plain", plain[:n_gz], kmers_gz), ("gzip", gz, kmers_gz)):
            ctx.stage_times(reset=True)
            t0 = time.perf_counter()
            for i, f in enumerate(files):
                ctx.sketch_file(f, os.path.join(tmp, "one_%s_%03d.gz" % (label, i)), K, M, S)
            wall = time.perf_counter() - t0
            st = ctx.stage_times(reset=True)
            out["sketch_one_thread_" + label] = {"files": len(files), "wall_s": wall, "kmers_per_s": km / wall,
                                                 "stage_s": {key: st[key] for key in ("read_s", "ingest_s", "scan_s", "gather_s", "build_s", "gzip_s")}}
        # ---- the worker pool (what bin/sub_sampler -t runs): all files, 1 / 8 / all host threads
        thread_counts = sorted({1, min(8, cores), cores})
        sk_files = [os.path.join(tmp, "sk_%03d.gz" % i) for i in range(n)]
        out["sketch_files"] = {}
        sp.sketch_files(plain[:min(n, cores)], sk_files[:min(n, cores)], K, M, S, threads=cores)      # warm: page cache, HIP modules
        for T in thread_counts:
            # twice: the first call at a worker count pins the slabs of the batches it has in flight (~0.2 ms per MB, once per
            # process: reported as first_call_wall_s), the second is what every later call costs
            walls = []
            for _ in range(2):
                t0 = time.perf_counter()
                res, st, _ = sp.sketch_files(plain, sk_files, K, M, S, threads=T)
                walls.append(time.perf_counter() - t0)
                assert all(r[0] == 0 for r in res)
            wall = walls[1]
            out["sketch_files"]["threads_%d" % T] = {"wall_s": wall, "first_call_wall_s": walls[0], "kmers_per_s": kmers / wall, "fasta_GB_per_s": fasta_bytes / wall / 1e9,
                                                     "stage_s_summed_over_workers": {key: st[key] for key in ("read_s", "ingest_s", "scan_s", "gather_s", "build_s", "gzip_s")}}
        t0 = time.perf_counter()
        res, st, _ = sp.sketch_files(gz, [os.path.join(tmp, "skz_%03d.gz" % i) for i in range(n_gz)], K, M, S, threads=cores)
        wall = time.perf_counter() - t0
        out["sketch_files"]["gzip_input_threads_%d" % cores] = {"files": n_gz, "wall_s": wall, "kmers_per_s": kmers_gz / wall}
        t0 = time.perf_counter()
        ctx.compare_files(sk_files, os.path.join(tmp, "res"))
        wall = time.perf_counter() - t0
        st = ctx.stage_times(reset=True)
        out["compare_files"] = {"sketches": n, "wall_s": wall, "pairs_per_s": n * (n - 1) / 2 / wall,
                                "stage_s": {key: st[key] for key in ("load_s", "compare_s", "csv_s", "csv_gzip_s")}}
        # ---- the CLIs themselves (process start and HIP initialisation included)
        fof = os.path.join(tmp, "fof.txt")
        open(fof, "w").write("\n".join(plain) + "\n")
        cli = os.path.join(ROOT, "bin", "sub_sampler")
        if os.path.exists(cli):
            out["cli"] = {"note": "whole processes: start + HIP initialisation (~0.2 s) included"}
            for T in thread_counts:
                t0 = time.perf_counter()
                r = subprocess.run([cli, "-f", "fof.txt", "-t", str(T), "-v", "0", "-p", "cli%d_" % T], cwd=tmp, capture_output=True, text=True)
                wall = time.perf_counter() - t0
                out["cli"]["sub_sampler_-t%d" % T] = {"wall_s": wall, "kmers_per_s": kmers / wall, "rc": r.returncode}
            # a job ten times the size (the same files under ten names each): process start and HIP initialisation amortised
            big = os.path.join(tmp, "big")
            os.mkdir(big)
            links = []
            for rep in range(10):
                for i, pth in enumerate(plain):
                    ln = os.path.join(big, "r%d_genome%03d.fa" % (rep, i))
                    os.symlink(pth, ln)
                    links.append(ln)
            open(os.path.join(big, "fof.txt"), "w").write("\n".join(links) + "\n")
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-f", "fof.txt", "-t", str(cores), "-v", "0", "-p", "big_"], cwd=big, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            out["cli"]["sub_sampler_-t%d_1000_files" % cores] = {"files": len(links), "wall_s": wall, "kmers_per_s": 10 * kmers / wall, "rc": r.returncode}
            shutil.rmtree(big, ignore_errors=True)
            t0 = time.perf_counter()
            r2 = subprocess.run([os.path.join(ROOT, "bin", "comparator"), "-f", "cli%d_fof.txt" % thread_counts[-1], "-o", "cli_res"], cwd=tmp,
                                capture_output=True, text=True)
            out["cli"]["comparator"] = {"wall_s": time.perf_counter() - t0, "pairs_per_s": n * (n - 1) / 2 / (time.perf_counter() - t0), "rc": r2.returncode}
        if not skip_oracle:
            from oracle import oracle_py as orc
            out["cpu_baseline"] = {"kind": "port", "cpu_model": cpu_model(), "unit": "k-mers/s (FASTA text in memory -> gzip -9 sketch bytes)",
                                   "what": "oracle parse_fasta_test restated (clean + scan + handle_superkmer + emission) + gzip -9, one file per "
                                           "thread as the reference's OpenMP loop (SubSampler.cpp:771); bounded sample: the first min(n, 4 x threads) files"}
            pls = {}

            def one(i):
                pl = orc.sketch_fasta(texts[i], K, M, S)[0]
                gzip.compress(pl, 9)
                return i, pl
            for T in thread_counts:
                m_files = min(n, 4 * T)
                t0 = time.perf_counter()
                with cf.ThreadPoolExecutor(T) as ex:
                    for i, pl in ex.map(one, range(m_files)):
                        pls[i] = pl
                wall = time.perf_counter() - t0
                okm = sum(len(gs[i]) - K + 1 for i in range(m_files))
                out["cpu_baseline"]["threads_%d" % T] = {"cores": T, "files": m_files, "wall_s": wall, "kmers_per_s": okm / wall}
            out["cpu_baseline"]["value"] = out["cpu_baseline"]["threads_1"]["kmers_per_s"]
            out["cpu_baseline"]["cores"] = 1
            out["speedup_at_equal_threads"] = {
                "library_worker_pool": {("threads_%d" % T): out["sketch_files"]["threads_%d" % T]["kmers_per_s"] / out["cpu_baseline"]["threads_%d" % T]["kmers_per_s"]
                                        for T in thread_counts},
                **({"cli_process": {**{("threads_%d" % T): out["cli"]["sub_sampler_-t%d" % T]["kmers_per_s"] / out["cpu_baseline"]["threads_%d" % T]["kmers_per_s"]
                                       for T in thread_counts},
                                    ("threads_%d_1000_files" % cores): out["cli"]["sub_sampler_-t%d_1000_files" % cores]["kmers_per_s"] /
                                    out["cpu_baseline"]["threads_%d" % cores]["kmers_per_s"]}} if "cli" in out else {})}
            mine = {i: sp.read_file(sk_files[i]) for i in pls}
            out["parity_vs_oracle"] = {"files": len(pls), "payload_bytes_equal": bool(all(mine[i] == pls[i] for i in pls))}
            k_files = min(n, 24)
            inter, card, csec = orc.compare([pls.get(i) or orc.sketch_fasta(texts[i], K, M, S)[0] for i in range(k_files)], timed=True)
            out["cpu_baseline"]["compare_pairs_per_s"] = k_files * (k_files - 1) / 2 / csec if csec > 0 else None
        return out
    finally:
        sp.sketch_files_release()
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
