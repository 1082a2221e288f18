"""Resident throughput against batch size (not a test): `python tests/gpu_batch_sweep.py [batches...]` on the GPU box.
One client (max_batch = the largest size asked for); every size is a resident batch of distinct acct-d8 runs proven in
lockstep, records in HBM before the clock (bench.py's `device_only` figure at that batch); one proof of every size is
verified on the host.  `profiles/r05_batch_sweep.txt` is this script's output."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 9, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 49, 56, 64, 96, 128, 192]
top = max(sizes)
client = zk.ProverClient(device=0, max_batch=top)
lib, h = client._lib, client._h
pk, vk = client.setup(zk.merkle_elf())
host = zk.ProverClient(device=-1)
handles = []
for i in range(top):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=1 + i).to_borsh())
    handles.append(client.machine_trace_handle(pk, s))
shape = zk.machine_cover_heights(handles)
lh = (C.c_int32 * zk.MACHINE_CHIPS)(*shape)
print(f"# resident throughput against batch size (acct-d8 machine proofs, format v{zk.MACHINE_VERSION}, one MI355X, one client of max_batch {top})")
prev = 0.0
for B in sizes:
    arr = (C.c_void_p * B)(*[t._h for t in handles[:B]])
    assert lib.zksp_hip_machine_load(h, pk._h, arr, B) == 0, client.last_error()
    for _ in range(2):
        assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
    lib.zksp_hip_sync(h)
    steps = max(3, min(40, 400 // B))
    t0 = time.perf_counter()
    for _ in range(steps):
        assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
    lib.zksp_hip_sync(h)
    el = time.perf_counter() - t0
    rate = B * steps / el
    bw = lib.zksp_machine_body_words(h, (C.c_int32 * zk.MACHINE_CHIPS)(*zk.machine_cover_heights(handles[:B])))
    bodies = np.zeros((B, bw), np.uint32)
    assert lib.zksp_hip_machine_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size) == 0
    host.verify(handles[B - 1].proof_from_body(pk, bodies[B - 1], zk.machine_cover_heights(handles[:B])), vk)
    print(f"batch {B:4d}  {rate:7.1f} proofs/s  {el * 1e3 / steps:8.2f} ms/step  {el * 1e3 / steps / B:6.2f} ms/proof"
          f"{'   <-- below the batch before' if rate < prev else ''}", flush=True)
    prev = rate
