"""LDE throughput probe (not a test): algorithmic GB/s = 16*H*ncols bytes / time
(4H read + 4H coefficients + 8H LDE written per column)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import Gpu
zk = importlib.import_module("zk-state-proofs_amd")
g = Gpu(zk)
lib, h = g.lib, g.h
SIZES = ((10, 2633 * 16), (11, 2633 * 16), (12, 2633 * 4), (14, 2048), (16, 1024), (18, 256), (20, 128), (21, 64), (22, 32))
if os.environ.get("ZKSP_NTT_LOGH"):
    SIZES = tuple(x for x in SIZES if x[0] == int(os.environ["ZKSP_NTT_LOGH"]))
if os.environ.get("ZKSP_NTT_NCOLS"):
    SIZES = tuple((l, int(os.environ["ZKSP_NTT_NCOLS"])) for l, _ in SIZES)
for logh, ncols in SIZES:
    H = 1 << logh
    rng = np.random.default_rng(logh)
    src = g.buf(rng.integers(0, 2013265921, (min(ncols, 64), H), dtype=np.uint32))
    big_in = g.buf(nbytes=ncols * H * 4)
    lib.zksp_dev_memset(h, big_in.ptr, 1, ncols * H * 4)
    coefs = g.buf(nbytes=ncols * H * 4)
    out = g.buf(nbytes=2 * ncols * H * 4)
    g.check(lib.zksp_hip_lde(h, big_in.ptr, logh, ncols, 1, coefs.ptr, out.ptr))  # warm-up (builds tables)
    reps = 3
    t = time.perf_counter()
    for _ in range(reps):
        g.check(lib.zksp_hip_lde(h, big_in.ptr, logh, ncols, 1, coefs.ptr, out.ptr))
    dt = (time.perf_counter() - t) / reps
    gb = 16.0 * H * ncols / 1e9
    print(f"logh {logh:2d} ncols {ncols:6d}: {dt*1e3:8.3f} ms  {gb/dt:8.1f} GB/s algorithmic", flush=True)
    for b in (src, big_in, coefs, out):
        b.free()
