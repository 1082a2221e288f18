"""Scratch timing (not a test): why bench.py's single_proof_device_ms (10.2 ms) differs from tests/gpu_single_latency.py (8.9 ms).
Steps of the bench's own order, the batch-of-one measured after each: torch initialises the device first; a client of 192; a
prove_batch call; resident passes of 192 with the profile spans on; spans off."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.set_device(0)
torch.cuda.synchronize()
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
client = zk.ProverClient(device=0, max_batch=192)
lib, h = client._lib, client._h
pk, vk = client.setup(zk.merkle_elf())


def handle(seed):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=seed).to_borsh())
    return client.machine_trace_handle(pk, s)


one = handle(1)


def single(label):
    arr = (C.c_void_p * 1)(one._h)
    assert lib.zksp_hip_machine_load(h, pk._h, arr, 1) == 0, client.last_error()
    for _ in range(5):
        assert lib.zksp_hip_machine_prove(h) == 0
    lib.zksp_hip_sync(h)
    t0 = time.perf_counter()
    for _ in range(40):
        assert lib.zksp_hip_machine_prove(h) == 0
    lib.zksp_hip_sync(h)
    print(f"{label}: {(time.perf_counter() - t0) * 1e3 / 40:.2f} ms per pass", flush=True)


single("after torch initialised the device, fresh client of 192")
stdins = []
for i in range(1024):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=100 + i).to_borsh())
    stdins.append(s)
proofs, status = client.prove_batch(pk, stdins)
assert status == [0] * 1024
single("after a prove_batch call of 1 024 (ramped)")
hs = [handle(1000 + i) for i in range(192)]
arr = (C.c_void_p * 192)(*[t._h for t in hs])
assert lib.zksp_hip_machine_load(h, pk._h, arr, 192) == 0
lib.zksp_hip_profile_reset(h)
lib.zksp_hip_profile_enable(h, 1)
for _ in range(3):
    assert lib.zksp_hip_machine_prove(h) == 0
lib.zksp_hip_sync(h)
single("after resident passes of 192, profile spans still ON")
lib.zksp_hip_profile_enable(h, 0)
single("profile spans off")
torch.cuda.synchronize()
single("after another torch synchronize")
# the bench's byte comparison: the oracle (OpenMP, the granted cores) proves one run before the batch of one is measured
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
os.environ.setdefault("OMP_NUM_THREADS", "16")
import oracle
s0 = zk.SP1Stdin()
s0.write(fx.acct_fixture(8, seed=1).to_borsh())
t = client.machine_trace(pk, s0)
t0 = time.perf_counter()
oracle.machine_prove(t)
print(f"(oracle proof: {time.perf_counter() - t0:.1f} s)", flush=True)
single("after an oracle proof on 16 OpenMP threads")
time.sleep(2.0)
single("two seconds later")
