"""Scratch timing of the end-to-end prove_batch path (not a test): ZKSP_TRACE_BATCH=1 prints the pipeline marks."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
client = zk.ProverClient(device=0, max_batch=int(sys.argv[2]) if len(sys.argv) > 2 else 16)
pk, vk = client.setup(zk.merkle_elf())
bufs = [fx.acct_fixture(8, seed=1000 + i).to_borsh() for i in range(n)]
for rep in range(2):
    stdins = []
    for b in bufs:
        s = zk.SP1Stdin(); s.write(b); stdins.append(s)
    t0 = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    el = time.perf_counter() - t0
    assert status == [0] * n
    print(f"rep {rep}: {n} proofs in {el*1e3:.1f} ms = {n/el:.1f} proofs/s", flush=True)
    del proofs
    print(f"        (after the C call returned: {(time.perf_counter() - t0 - el) * 1e3:.1f} ms to drop the proofs)", flush=True)
