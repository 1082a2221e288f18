"""Probe (not a test): end-to-end time of single proofs through the reference's call (client.prove(pk, stdin).run()).
ZKSP_TRACE_BATCH=1 prints the host-side timeline of each call on stderr."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
fx = importlib.import_module("zk-state-proofs_amd.fixtures")
client = zk.ProverClient(device=0)
pk, vk = client.setup(zk.merkle_elf())
buf = fx.acct_fixture(8).to_borsh()
ts = []
for i in range(12):
    s = zk.SP1Stdin()
    s.write(buf)
    t0 = time.perf_counter()
    proof = client.prove(pk, s).run()
    ts.append((time.perf_counter() - t0) * 1e3)
client.verify(proof, vk)
print("end to end ms: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
s = zk.SP1Stdin()
s.write(buf)
t0 = time.perf_counter()
h = client.machine_trace_handle(pk, s)
print(f"guest tracing alone: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
