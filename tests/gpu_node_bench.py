"""Scratch timing (not a test): the device stages of a resident batch of recursion-tree NODES (each one guest run of a depth-1
account proof that checks `arity` full-size leaf proofs): `gpu_node_bench.py [nodes=8] [arity=4]`."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arity = int(sys.argv[2]) if len(sys.argv) > 2 else 4
client = zk.ProverClient(device=0, max_batch=max(B, 32))
lib, h = client._lib, client._h
pk, vk = client.setup(zk.merkle_elf())
stdins = []
for i in range(B * arity):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=500 + i).to_borsh())
    stdins.append(s)
leaves, status = client.prove_batch(pk, stdins)
assert status == [0] * len(stdins)
handles = []
t0 = time.perf_counter()
for k in range(B):
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1, seed=900 + k).to_borsh())
    client.add_verified_leaves(s, leaves[arity * k:arity * (k + 1)], [vk] * arity)
    handles.append(client.machine_trace_handle(pk, s))
print("check + trace ms/node", (time.perf_counter() - t0) * 1e3 / B, "heights", handles[0].heights(), flush=True)
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
print(f"batch of {B} nodes: {el * 1e3 / steps:.1f} ms/step, {el * 1e3 / steps / B:.2f} ms per node", flush=True)
tot, cnt = C.c_double(), C.c_uint64()
for name in (b"m_trace", b"m_lde_main", b"m_commit_main", b"m_leaf_main", b"m_perm", b"m_lde_perm", b"m_commit_perm", b"m_quotient", b"m_lde_quot",
             b"m_commit_quot", b"m_open", b"merkle_open", b"m_reduce", b"fri_commit", b"fri_fold", b"grind", b"transcript", b"m_assemble"):
    lib.zksp_hip_profile_read(h, name, C.byref(tot), C.byref(cnt))
    print(f"  {name.decode():16s} {tot.value / steps:9.2f} ms/step")
