"""Probe (not a test): the timeline of one prove_batch call (ZKSP_TRACE_BATCH=1 prints it on stderr).
Usage: ZKSP_TRACE_BATCH=1 python tests/gpu_batch_timeline_probe.py [n_proofs] [max_batch]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
client = zk.ProverClient(device=0, max_batch=mb)
pk, vk = client.setup(zk.merkle_elf())
bufs = [fx.acct_fixture(8, seed=1 + i).to_borsh() for i in range(n)]
for rep in range(2):  # the first call sizes the arena and builds the tables
    stdins = []
    for b in bufs:
        s = zk.SP1Stdin()
        s.write(b)
        stdins.append(s)
    t0 = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    el = time.perf_counter() - t0
    assert status == [0] * n
    print(f"call {rep}: {n} proofs in {el:.3f} s = {n / el:.1f} proofs/s", file=sys.stderr, flush=True)
    del proofs
