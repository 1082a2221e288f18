"""Scratch timing (not a test): device time per pass of a small resident batch of machine proofs (default one
proof), profiling spans off; the proof it leaves is checked by the host verifier."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_args[0]) if len(_args) > 0 else 1
CAP = int(_args[1]) if len(_args) > 1 else B  # the client's max_batch (bench.py measures a batch of one on a client of 192)
client = zk.ProverClient(device=int(os.environ.get("ZKSP_DEVICE", "0")), max_batch=CAP)
lib, h = client._lib, client._h
pk, vk = client.setup(zk.merkle_elf())
handles = []
for i in range(B):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=1 + i).to_borsh())
    handles.append(client.machine_trace_handle(pk, s))
arr = (C.c_void_p * B)(*[t._h for t in handles])
assert lib.zksp_hip_machine_load(h, pk._h, arr, B) == 0, client.last_error()
for _ in range(3):
    assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
lib.zksp_hip_sync(h)
steps = 40 if "--json" in sys.argv else 20
t0 = time.perf_counter()
for _ in range(steps):
    assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
lib.zksp_hip_sync(h)
el = time.perf_counter() - t0
print(f"batch {B}: {el * 1e3 / steps:.2f} ms/step", flush=True)
# host time to enqueue one pass (the call returns when everything is queued) against the synchronised pass
enq, tot = [], []
for _ in range(10):
    t0 = time.perf_counter()
    assert lib.zksp_hip_machine_prove(h) == 0
    t1 = time.perf_counter()
    lib.zksp_hip_sync(h)
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print("enqueue ms " + " ".join(f"{x:.2f}" for x in enq), flush=True)
print("one pass, synchronised, ms " + " ".join(f"{x:.2f}" for x in tot), flush=True)
shape = zk.machine_cover_heights(handles)
lh = (C.c_int32 * zk.MACHINE_CHIPS)(*shape)
bw = lib.zksp_machine_body_words(h, lh)
bodies = np.zeros((B, bw), np.uint32)
assert lib.zksp_hip_machine_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size) == 0
zk.ProverClient(device=-1).verify(handles[0].proof_from_body(pk, bodies[0], shape), vk)
print("verified")
if "--json" in sys.argv:  # (bench.py runs this file as a child process for its single-proof figure)
    import json
    print("JSON " + json.dumps({"batch": B, "ms_per_pass": el * 1e3 / steps, "synchronised_ms": sorted(tot)[len(tot) // 2],
                               "enqueue_ms": sorted(enq)[len(enq) // 2], "verified": True}), flush=True)
