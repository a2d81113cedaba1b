"""world_size-2 tests of the multi-GPU comparison path: key
all-gather + row ownership (a rank's own block of rows, or every world-th
row).  CPU variant runs here with gloo; the GPU variant runs the real
spsp_compare_device in both ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, port, world=2, rows="block"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", SPSP_TEST_ROWS=rows)
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (gloo otherwise looks the host's name up: tens of seconds on a box whose name does not resolve)
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize("rows", ["block", "strided"])
def test_key_exchange_and_row_partition_gloo_cpu(rows):
    _launch("cpu", 29611 if rows == "block" else 29641, rows=rows)


def test_slot_exchange_and_partial_sum_gloo_cpu():
    _launch("cpu_slots", 29613)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", ["block", "strided"])
def test_two_ranks_compare_device_rows(rows):
    _launch("gpu", 29612 if rows == "block" else 29642, rows=rows)


@pytest.mark.gpu
def test_two_ranks_key_partitioned_exchange():
    _launch("gpu_slots", 29614)


@pytest.mark.parametrize("rows", ["block", "strided"])
def test_eight_ranks_c4_shape_gloo_cpu(rows):
    """world_size 8, 2 048 sketches in families (BASELINE configs[3] in small): key all-gather, row ownership (the
    rank's own 256 sketches / i % 8 == rank), strips collected on rank 0 equal an inverted-index count over all 2.1
    million pairs."""
    _launch("cpu_c4", 29621 if rows == "block" else 29643, world=8, rows=rows)


@pytest.mark.gpu
def test_two_ranks_rccl_both_exchange_forms():
    """backend "nccl" with two ranks, collectives on the contexts' own streams (needs two GPUs: skipped on a 1-GPU box)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _launch("nccl", 29623, world=2)


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py as the driver launches it for N = 2 (`python bench.py --gpus 2`: one process per rank), rehearsed on ONE
    GPU: both ranks compute on cuda:0 and the collectives go through gloo (BENCH_BACKEND / BENCH_SHARE_GPU).  The
    whole N > 1 path of the file runs -- rank-sharded genomes, key all-gather, own rows, strips to rank 0, barrier,
    max over ranks, ONE JSON line from rank 0 -- and the line's untimed check says the collected matrix equals a
    one-device comparison of all sketches; so does the multi-rank BASELINE configs[3] leg (`compare_c4`: the comparison
    split by key, slots by all-to-all, partial matrices as sparse cells), here at 400 sketches."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARE_GPU="1", BENCH_PREWARM_STEPS="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="1")
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # the PLAIN command of the driver (`python bench.py --gpus N ...`, no WORLD_SIZE): bench.py starts its two ranks itself,
    # as child processes through torch.distributed.run, before it touches the GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--genomes", "6", "--length", "300000", "--no-cpu-baseline"]
    env["BENCH_C4_SKETCHES"] = "400"                         # the multi-rank configs[3] leg at a rehearsal's size
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert d["config"]["sketches_total"] == 12
    assert d["config"]["exchange_check"].startswith("collected strips equal"), d["config"]["exchange_check"]
    assert "rehearsal" in d["config"] and d["value"] > 0 and d["valid"] is False
    c4 = d["compare_c4"]
    assert c4["exchange_check"].startswith("sum of the ranks' partial matrices equals"), c4
    assert c4["nonzero_pairs"] == 20 * 190 and c4["pairs"] == 400 * 399 // 2


def test_bench_starts_its_own_ranks_when_called_plainly(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE (the driver's command): N ranks as a CHILD process through
    torch.distributed.run on 127.0.0.1, same arguments, the child's status returned; with WORLD_SIZE set, or N = 1, nothing
    is started (CPU: the launcher only -- the GPU tier runs it for real in test_bench_two_ranks_rehearsal_on_one_gpu)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen.append((cmd, env))
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks_if_needed()
    assert e.value.code == 7 and len(seen) == 1
    cmd, env = seen[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus=2"])
    with pytest.raises(SystemExit):
        bench.launch_ranks_if_needed()
    assert seen[-1][0][seen[-1][0].index("--nproc-per-node") + 1] == "2"
    n = len(seen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3"])
    bench.launch_ranks_if_needed()                              # one GPU: this process is the bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.launch_ranks_if_needed()                              # already a rank
    assert len(seen) == n
