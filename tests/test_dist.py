"""world_size-2 tests of the multi-GPU comparison path: key
all-gather + strided row ownership.  CPU variant runs here with gloo; the GPU
variant runs the real spsp_compare_device in both ranks."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(mode, port, world=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


def test_key_exchange_and_row_partition_gloo_cpu():
    _launch("cpu", 29611)


def test_slot_exchange_and_partial_sum_gloo_cpu():
    _launch("cpu_slots", 29613)


@pytest.mark.gpu
def test_two_ranks_compare_device_rows():
    _launch("gpu", 29612)


@pytest.mark.gpu
def test_two_ranks_key_partitioned_exchange():
    _launch("gpu_slots", 29614)


def test_eight_ranks_c4_shape_gloo_cpu():
    """world_size 8, 2 048 sketches in families (BASELINE configs[3] in small): key all-gather, row ownership
    i % 8 == rank, strips collected on rank 0 equal an inverted-index count over all 2.1 million pairs."""
    _launch("cpu_c4", 29621, world=8)


@pytest.mark.gpu
def test_two_ranks_rccl_both_exchange_forms():
    """backend "nccl" with two ranks, collectives on the contexts' own streams (needs two GPUs: skipped on a 1-GPU box)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _launch("nccl", 29623, world=2)
