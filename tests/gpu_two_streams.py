"""Scratch experiment (not a test): two clients on one GPU, each with half the batch resident, proving
concurrently on their own streams, against one client with the whole batch."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

HALF = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NCL = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def make(n, seed0):
    client = zk.ProverClient(device=0, max_batch=n)
    pk, vk = client.setup(zk.merkle_elf())
    handles = []
    for i in range(n):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(8, seed=seed0 + i).to_borsh())
        handles.append(client.machine_trace_handle(pk, s))
    arr = (C.c_void_p * n)(*[t._h for t in handles])
    assert client._lib.zksp_hip_machine_load(client._h, pk._h, arr, n) == 0, client.last_error()
    return client, pk, handles


clients = [make(HALF, 1 + HALF * k) for k in range(NCL)]
lib = clients[0][0]._lib
for _ in range(2):
    for c, _, _ in clients:
        assert lib.zksp_hip_machine_prove(c._h) == 0
for c, _, _ in clients:
    lib.zksp_hip_sync(c._h)
steps = 6
STAGGER = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0  # seconds between the first launches of successive clients
t0 = time.perf_counter()
for st in range(steps):
    for k, (c, _, _) in enumerate(clients):
        assert lib.zksp_hip_machine_prove(c._h) == 0
        if st == 0 and STAGGER and k + 1 < len(clients):
            time.sleep(STAGGER)
for c, _, _ in clients:
    lib.zksp_hip_sync(c._h)
el = time.perf_counter() - t0
print(f"{NCL} clients x {HALF}: {el * 1e3 / steps:.1f} ms per round, {NCL * HALF * steps / el:.1f} proofs/s", flush=True)
