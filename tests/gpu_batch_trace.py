"""prove_batch timeline probe (not a test): ZKSP_TRACE_BATCH=1 python tests/gpu_batch_trace.py [n]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
client = zk.ProverClient(device=0, max_batch=256)
pk, vk = client.setup(zk.merkle_elf())
for rep in range(2):
    stdins = []
    for i in range(n):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(8, seed=1000 + i).to_borsh())
        stdins.append(s)
    t = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    dt = time.perf_counter() - t
    assert status == [0] * n
    print(f"rep {rep}: {n} proofs in {dt*1e3:.1f} ms = {n/dt:.0f} proofs/s", flush=True)
