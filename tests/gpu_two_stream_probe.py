"""Probe (not a test): do two half-batches on two clients (two HIP streams) of one GPU overlap their memory-bound
and their issue-bound kernels?  Compares one client proving 2n acct-d8 runs per pass with two clients proving n each,
the passes enqueued back to back.  Usage: python tests/gpu_two_stream_probe.py [n] [steps]"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def make(count, seed0):
    client = zk.ProverClient(device=0)
    pk, vk = client.setup(zk.merkle_elf())
    handles = []
    for i in range(count):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(8, seed=seed0 + i).to_borsh())
        handles.append(client.machine_trace_handle(pk, s))
    arr = (C.c_void_p * count)(*[t._h for t in handles])
    rc = client._lib.zksp_hip_machine_load(client._h, pk._h, arr, count)
    assert rc == 0, client.last_error()
    return client, pk, handles


def run(clients):
    for c, _, _ in clients:
        assert c._lib.zksp_hip_machine_prove(c._h) == 0
    for c, _, _ in clients:
        assert c._lib.zksp_hip_sync(c._h) == 0
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, _, _ in clients:
            assert c._lib.zksp_hip_machine_prove(c._h) == 0
    for c, _, _ in clients:
        assert c._lib.zksp_hip_sync(c._h) == 0
    return (time.perf_counter() - t0) / steps


one = make(2 * n, 1)
t1 = run([one])
print(f"one client, batch {2 * n}: {t1 * 1e3:.1f} ms per pass, {2 * n / t1:.1f} proofs/s", flush=True)
del one
a, b = make(n, 1), make(n, 1 + n)
t2 = run([a, b])
print(f"two clients, batch {n} each: {t2 * 1e3:.1f} ms per pass, {2 * n / t2:.1f} proofs/s", flush=True)
