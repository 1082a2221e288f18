"""The N = 2 path of bench.py on the one-GPU box (round 4's verdict, item 6): two ranks of the
proof farm as CHILD processes, both on cuda:0 under gloo (RCCL refuses two ranks on one device), every rank proving its
own inputs through the drop-in prove_batch and the ranks all-gathering the 32-byte main-trace commitments.

This file sorts before every other test file on purpose: the ranks are started with fork + exec, which a process that
has initialised the GPU must not do on this pool - here nothing has touched HIP yet."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_device_gather_their_roots(built_lib):
    env = dict(os.environ, ZKSP_BENCH_SAME_DEVICE="1", ZKSP_HOST_THREADS="8", OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--batch", "16",
           "--proofs-per-step", "32", "--steps", "2", "--warmup", "1", "--device-steps", "2", "--skip-single", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-4000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines  # rank 0's JSON line is the only thing on stdout
    line = json.loads(lines[0])
    # two ranks, weak scaling: every rank proved its own 32 inputs per step; inside the timed region each rank asserted
    # that the all-gathered roots, in proof order, hold its own at its own indices (bench.py: farm.gather_roots)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["proofs_per_step"] == 64 and line["config"]["resident_chunk_per_gpu"] == 16
    assert line["value"] > 0 and abs(line["value"] - 64 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    assert line["timed_step_checked"]["public_values_checked"] == 32
    assert line["device_only"]["batch"] == 16 and line["device_only"]["value"] > 0
    assert line["roofline"]["achieved"] > 0
