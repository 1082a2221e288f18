"""Scratch timing (not a test): the device time of a batch of one on a client that has served a multi-chunk prove_batch call
before (copy stream and its events exist), against a fresh client - what bench.py's single_proof_device_ms measures since the
timed step is the drop-in call."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")


def single(client, pk, label):
    lib, h = client._lib, client._h
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=1).to_borsh())
    t = client.machine_trace_handle(pk, s)
    arr = (C.c_void_p * 1)(t._h)
    assert lib.zksp_hip_machine_load(h, pk._h, arr, 1) == 0, client.last_error()
    for _ in range(5):
        assert lib.zksp_hip_machine_prove(h) == 0
    lib.zksp_hip_sync(h)
    t0 = time.perf_counter()
    for _ in range(40):
        assert lib.zksp_hip_machine_prove(h) == 0
    lib.zksp_hip_sync(h)
    print(f"{label}: {(time.perf_counter() - t0) * 1e3 / 40:.2f} ms per pass", flush=True)


client = zk.ProverClient(device=0, max_batch=32)
pk, vk = client.setup(zk.merkle_elf())
single(client, pk, "fresh client")
stdins = []
for i in range(80):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=100 + i).to_borsh())
    stdins.append(s)
proofs, status = client.prove_batch(pk, stdins)
assert status == [0] * 80
single(client, pk, "after a prove_batch call of three chunks")
del proofs
single(client, pk, "after dropping its proofs")
import torch
torch.cuda.set_device(0)
x = torch.zeros(16, device="cuda")
torch.cuda.synchronize()
single(client, pk, "after torch initialised the device in this process")
