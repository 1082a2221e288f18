"""Probe (not a test): the as-committed acct-d8 machine proof (six CPU instances of 2^18 rows) on the device against the
oracle's bytes; prints the first differing body word and the section it lies in."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
nq, pw = 8, 6
client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, keccak_mode=zk.KECCAK_OBSERVE, max_batch=4)
pk, vk = client.setup(zk.merkle_elf())
seeds = [int(x) for x in os.environ.get("SEEDS", "2").split(",")]
handles, traces = [], []
for sd in seeds:
    s = zk.SP1Stdin(); s.write(fx.acct_fixture(8, seed=sd).to_borsh())
    handles.append(client.machine_trace_handle(pk, s)); traces.append(client.machine_trace(pk, s))
shape = zk.machine_cover_heights(handles)
print("shape", shape, flush=True)
bodies = client.machine_prove_resident(pk, handles)
host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw, keccak_mode=zk.KECCAK_OBSERVE)
for i in range(len(seeds)):
    proof = handles[i].proof_from_body(pk, bodies[i], shape)
    try:
        host.verify(proof, vk); print(i, "verified", flush=True)
    except Exception as e:
        print(i, "REJECTED", e, flush=True)
t0 = time.time()
exp = oracle.machine_prove(dict(traces[0], shape=shape), num_queries=nq, pow_bits=pw)
print("oracle s", time.time() - t0, flush=True)
hw = zk.MACHINE_HEADER_WORDS + (len(traces[0]["public_values"]) + 3) // 4
e = np.frombuffer(exp, dtype=np.uint32)[hw:]
bad = np.nonzero(e != bodies[0])[0]
nc = zk.MACHINE_CHIPS
print("differing words", bad.size, "first", bad[:8], "sections: main root 0..7, perm root 8..15, cums 16..%d, quot root %d.." % (15 + 4 * nc, 16 + 4 * nc))
