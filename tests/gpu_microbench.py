"""Prints the instruction-rate probes (not a test)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import Gpu
zk = importlib.import_module("zk-state-proofs_amd")
g = Gpu(zk)
names = ["add_u32", "mul_lo_u32", "mul_hi_u32", "mad_u64_u32", "monty_mul", "fma_f64"]
for i, n in enumerate(names):
    print(f"{n}: {g.microbench(i):.1f} Gop/s", flush=True)
for per_cu in (1, 2, 3, 4, 6, 8):
    print(f"poseidon2 perms, {per_cu} workgroups/CU: {g.microbench(5 + per_cu):.3f} Gperm/s", flush=True)
