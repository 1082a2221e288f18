"""Scratch timing of the device machine prover (not a test): `gpu_machine_bench.py [batch] [depth] [observe]`;
acct-d8 in the precompile shape by default, "observe" = the guest as committed (2^21-row CPU chip)."""
import ctypes as C
import importlib
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = zk.KECCAK_OBSERVE if len(sys.argv) > 3 and sys.argv[3] == "observe" else zk.KECCAK_REPLACE
client = zk.ProverClient(device=0, max_batch=B, keccak_mode=mode)
lib, h = client._lib, client._h
pk, vk = client.setup(zk.merkle_elf())
t0 = time.perf_counter()
handles = []
for i in range(B):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(depth, seed=1 + i).to_borsh())
    handles.append(client.machine_trace_handle(pk, s))
print("trace ms/proof", (time.perf_counter() - t0) * 1e3 / B, "heights", handles[0].heights(), flush=True)
arr = (C.c_void_p * B)(*[t._h for t in handles])
t0 = time.perf_counter()
assert lib.zksp_hip_machine_load(h, pk._h, arr, B) == 0, client.last_error()
print("load ms", (time.perf_counter() - t0) * 1e3, flush=True)
assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
lib.zksp_hip_sync(h)
lib.zksp_hip_profile_reset(h)
lib.zksp_hip_profile_enable(h, 1)
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    assert lib.zksp_hip_machine_prove(h) == 0, client.last_error()
lib.zksp_hip_sync(h)
el = time.perf_counter() - t0
print(f"batch {B}: {el * 1e3 / steps:.1f} ms/step, {B * steps / el:.1f} proofs/s", flush=True)
tot, cnt = C.c_double(), C.c_uint64()
for name in (b"m_trace", b"m_lde_main", b"m_commit_main", b"m_perm", b"m_lde_perm", b"m_commit_perm", b"m_quotient", b"m_lde_quot",
             b"m_commit_quot", b"m_open", b"merkle_open", b"m_reduce", b"fri_commit", b"fri_fold", b"grind", b"transcript", b"m_assemble"):
    lib.zksp_hip_profile_read(h, name, C.byref(tot), C.byref(cnt))
    print(f"  {name.decode():16s} {tot.value / steps:9.2f} ms/step")
shape = zk.machine_cover_heights(handles)
lh = (C.c_int32 * zk.MACHINE_CHIPS)(*shape)
bw = lib.zksp_machine_body_words(h, lh)
bodies = np.zeros((B, bw), np.uint32)
assert lib.zksp_hip_machine_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size) == 0
host = zk.ProverClient(device=-1, keccak_mode=mode)
t0 = time.perf_counter()
for i in range(min(B, 3)):
    host.verify(handles[i].proof_from_body(pk, bodies[i], shape), vk)
print("verified; verify ms/proof", (time.perf_counter() - t0) * 1e3 / min(B, 3), "proof bytes", bw * 4)
