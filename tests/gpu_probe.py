"""Ad-hoc probe (not a test): per-stage device times and PoW witnesses for one batch."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
zk = importlib.import_module("zk-state-proofs_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
client = zk.ProverClient(device=0, max_batch=B)
lib, h = client._lib, client._h
rng = np.random.default_rng(0)
states = rng.integers(0, 2**64, (B, 62, 25), dtype=np.uint64)
npm = np.full(B, 62, np.uint32)
obs = rng.integers(0, 2**16, (B, 44), dtype=np.uint32)
obs[:, 8] = 11
assert lib.zksp_hip_load_batch(h, 11, B, 62, states.ctypes.data_as(C.c_void_p), npm.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p)) == 0
for it in range(3):
    lib.zksp_hip_profile_reset(h); lib.zksp_hip_profile_enable(h, 1)
    t = time.perf_counter()
    assert lib.zksp_hip_prove_resident(h) == 0
    lib.zksp_hip_sync(h)
    print("step ms", (time.perf_counter() - t) * 1e3)
tot, cnt = C.c_double(), C.c_uint64()
for name in (b"keccak_trace", b"lde_trace", b"leaf_hash_trace", b"merkle_upper", b"quotient", b"lde_quot", b"merkle_quot", b"open", b"merkle_open", b"reduce_openings", b"fri_commit", b"fri_fold", b"grind", b"transcript", b"assemble"):
    lib.zksp_hip_profile_read(h, name, C.byref(tot), C.byref(cnt)); print(name.decode(), round(tot.value, 3), cnt.value)
bw = lib.zksp_proof_body_words(h, 11)
bodies = np.zeros((B, bw), np.uint32)
assert lib.zksp_hip_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size) == 0
off = 16 + (2 * 2633 + 8) * 4 + 8 * 11 + 4
print("witness", sorted(int(x) for x in bodies[:, off]))
