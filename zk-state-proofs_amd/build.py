"""In-tree build of ``libzksp.so`` (HIP kernels + C-ABI host code) for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the build container; the
resulting shared object travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

# The round-1 keccak-chip COMPONENT path (include/zksp_component.h: proof format v2, its prover, verifier and keccak-only
# kernels) is a build switch since round 5: the default libzksp.so does not contain it.  ZKSP_COMPONENT=1 in the
# environment builds libzksp_component.so - every source below with -DZKSP_COMPONENT plus COMPONENT_SOURCES - which the
# client loads under the same variable (tests/test_gpu_prover.py, the component tests of tests/test_verifier.py).
COMPONENT = os.environ.get("ZKSP_COMPONENT", "") == "1"
OBJ = os.path.join(CSRC, "_obj_component" if COMPONENT else "_obj")
LIB = os.path.join(HERE, "libzksp_component.so" if COMPONENT else "libzksp.so")

SOURCES = [
    "device/kernels_ntt.hip",
    "device/kernels_hash.hip",
    "device/kernels_stark.hip",
    "device/kernels_machine.hip",
    "device/kernels_bench.hip",
    "host/executor.cpp",
    "host/machine.cpp",
    "host/params.cpp",
    "host/p2_avx2.cpp",
    "host/p2_avx512.cpp",
    "host/cpu_features.cpp",
    "host/context.cpp",
    "host/mprover.cpp",
    "host/machine_defs.cpp",
    "host/mverifier.cpp",
    "host/zeta_program.cpp",
    "host/api.cpp",
    "host/api_prove.cpp",
    "host/api_machine.cpp",
]
COMPONENT_SOURCES = ["device/kernels_bus.hip", "host/prover.cpp", "host/verifier.cpp"]
if COMPONENT:
    SOURCES = SOURCES + COMPONENT_SOURCES
HEADERS = [
    "device/field.hpp", "device/poseidon2.hpp", "device/air_keccak.hpp", "device/air_machine.hpp", "device/kernels.h", "device/kernels_machine.h",
    "host/machine_defs.hpp", "host/mverifier.hpp", "host/zeta_program.hpp", "host/mprover.hpp", "host/host_hash.hpp",
    "host/executor.hpp", "host/machine.hpp", "host/context.hpp", "host/prover.hpp", "host/verifier.hpp", "host/api_types.hpp",
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + (["-DZKSP_COMPONENT"] if COMPONENT else [])


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newest_header() -> float:
    paths = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "zksp.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(p) for p in paths)


def _compile(src: str, force: bool, hdr_time: float) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace("/", "_") + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
        return obj
    cmd = [_hipcc(), *FLAGS, "-I", INCLUDE]
    if src in ("host/p2_avx2.cpp", "host/p2_avx512.cpp", "host/cpu_features.cpp"):
        # the host verifier's vector permutations: plain C++ (hipcc would compile a .cpp as HIP, for the GPU as well), the vector
        # extension for that file only, entered after a CPU check (cpu_features.cpp, compiled without any)
        vec = {"host/p2_avx2.cpp": ["-mavx2"], "host/p2_avx512.cpp": ["-mavx512f"]}.get(src, [])
        cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-Wall", *vec]
    elif src.endswith(".cpp") and src not in ("host/executor.cpp", "host/machine.cpp"):
        cmd += ["-x", "hip"]  # host code that shares the __host__ __device__ field/AIR headers
    cmd += ["-c", path, "-o", obj]
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_time), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
