"""Probe (not a test): device time of the proving stages inside one pipelined prove_batch call against the call's wall
time - do the passes themselves slow down while uploads, tracing and wrapping run beside them, or do gaps open between
them?  Usage: python tests/gpu_batch_spans_probe.py [n_proofs] [max_batch]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 192
client = zk.ProverClient(device=0, max_batch=mb)
pk, vk = client.setup(zk.merkle_elf())
bufs = [fx.acct_fixture(8, seed=1 + i).to_borsh() for i in range(n)]
names = ["m_trace", "m_lde_main", "m_commit_main", "m_perm", "m_lde_perm", "m_commit_perm", "m_quotient", "m_lde_quot", "m_commit_quot",
         "m_open", "merkle_open", "m_reduce", "fri_commit", "fri_fold", "grind", "transcript", "m_assemble"]
lib, h = client._lib, client._h
for rep in range(2):  # the first call sizes the arena and builds the tables
    stdins = []
    for b in bufs:
        s = zk.SP1Stdin()
        s.write(b)
        stdins.append(s)
    lib.zksp_hip_profile_reset(h)
    lib.zksp_hip_profile_enable(h, 1)
    t0 = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    el = time.perf_counter() - t0
    lib.zksp_hip_profile_enable(h, 0)
    assert status == [0] * n
    tot, cnt, dev = C.c_double(), C.c_uint64(), 0.0
    for nm in names:
        if lib.zksp_hip_profile_read(h, nm.encode(), C.byref(tot), C.byref(cnt)) == 0:
            dev += tot.value
    print(f"call {rep}: {n} proofs in {el * 1e3:.0f} ms = {n / el:.1f} proofs/s; device stage spans {dev:.0f} ms = {n / dev * 1e3:.1f} proofs/s of busy time",
          flush=True)
    del proofs
