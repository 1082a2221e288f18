"""ASan + UBSan over the host side of the boundary (executor, ELF loader, proof
parser, verifier) with mutated inputs.  CPU build only (GPU sanitizers are not
available on this pool); the harness is tests/host_fuzz.cpp."""
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "zk-state-proofs_amd", "csrc", "host")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_code_under_asan_ubsan(tmp_path, zk, fx, oracle, host_client):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "host_fuzz"
    srcs = [os.path.join(ROOT, "tests", "host_fuzz.cpp")] + [os.path.join(HOST, f) for f in
                                                             ("executor.cpp", "verifier.cpp", "params.cpp", "machine.cpp",
                                                              "machine_defs.cpp", "mverifier.cpp", "zeta_program.cpp")]
    # (-DZKSP_COMPONENT: the harness feeds the component verifier of verifier.cpp too, which the default library leaves out)
    base = [hipcc, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-DZKSP_COMPONENT", "-I", HOST,
            "-I", os.path.join(ROOT, "include")]
    both = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    # the machine verifier instantiates every chip's constraints over the extension field: UBSan's per-operation
    # checks make that one file compile for four minutes, so it gets ASan only (memory safety on untrusted bytes)
    procs = []
    for i, src in enumerate(srcs):
        flags = ["-fsanitize=address"] if src.endswith(("mverifier.cpp", "zeta_program.cpp")) else both
        procs.append(subprocess.Popen(base + flags + ["-c", src, "-o", str(tmp_path / f"o{i}.o")]))
    # the verifier's vector permutation: plain C++ with AVX2, as build.py compiles it, sanitized like the rest
    # (with the same compiler as the rest, so that one sanitizer runtime sees one kind of instrumentation: g++'s 64-byte
    # aligned stack slots of the AVX-512 file do not survive clang's fake stack)
    cxx = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(cxx):
        cxx = "/opt/rocm/lib/llvm/bin/clang++"
    for name, flag in (("p2_avx2", "-mavx2"), ("p2_avx512", "-mavx512f"), ("cpu_features", "-O1")):
        procs.append(subprocess.Popen([cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", flag, *both, "-c",
                                       os.path.join(HOST, name + ".cpp"), "-o", str(tmp_path / (name + ".o"))]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([hipcc, "-fsanitize=address,undefined", *[str(tmp_path / f"o{i}.o") for i in range(len(srcs))],
                           str(tmp_path / "p2_avx2.o"), str(tmp_path / "p2_avx512.o"), str(tmp_path / "cpu_features.o"), "-o", str(exe)])
    pk, vk = host_client.setup(zk.merkle_elf())
    stdin_bytes = fx.stdin_frame(fx.tx_fixture().to_borsh())
    (tmp_path / "stdin.bin").write_bytes(stdin_bytes)
    vk_words = [int(x) for x in np.frombuffer(vk.digest, dtype=np.uint32)]
    st = np.random.default_rng(2).integers(0, 2**64, (2, 25), dtype=np.uint64)
    pv = b"abc"
    pvd = [int(x) for x in np.frombuffer(hashlib.sha256(pv).digest(), dtype=np.uint32)]
    (tmp_path / "proof.bin").write_bytes(
        oracle.prove(st, 6, public_values=pv, pv_digest=pvd, vk_digest=vk_words, num_queries=12, pow_bits=8))
    # a machine proof of a short guest run, proven by the oracle
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1).to_borsh())
    (tmp_path / "mproof.bin").write_bytes(oracle.machine_prove(host_client.machine_trace(pk, s), num_queries=6, pow_bits=4))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe), zk.MERKLE_ELF_PATH, str(tmp_path / "stdin.bin"), str(tmp_path / "proof.bin"), "150",
                          str(tmp_path / "mproof.bin"), "6", "4"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr[-3000:]
    assert "fuzz ok" in out.stdout
