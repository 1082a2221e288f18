"""Whole-proof parity: the HIP prover must reproduce the CPU oracle's proof bytes
exactly, and the host verifier must accept honest and reject tampered proofs."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from util import Gpu, P

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("ZKSP_COMPONENT", "") != "1",
                                 reason="the keccak-chip component path is a build switch (ZKSP_COMPONENT=1; include/zksp_component.h)")]


def init_obs(vk, logh, n_perms, exit_code, pv_digest, deferred):
    o = list(vk) + [logh, n_perms, exit_code & 0xffff, exit_code >> 16]
    for w in pv_digest:
        o += [w & 0xffff, w >> 16]
    for w in deferred:
        o += [w & 0xffff, w >> 16]
    assert len(o) == 44
    return o


def device_bodies(g, logh, states_list, obs_list):
    n = len(states_list)
    max_perms = max(max(len(s) for s in states_list), 1)
    st = np.zeros((n, max_perms, 25), np.uint64)
    for i, s in enumerate(states_list):
        st[i, :len(s)] = s
    npm = np.array([len(s) for s in states_list], np.uint32)
    obs = np.array(obs_list, np.uint32)
    g.check(g.lib.zksp_hip_load_batch(g.h, logh, n, max_perms, st.ctypes.data_as(C.c_void_p),
                                      npm.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p)))
    g.check(g.lib.zksp_hip_prove_resident(g.h))
    bw = g.lib.zksp_proof_body_words(g.h, logh)
    out = np.zeros((n, bw), np.uint32)
    g.check(g.lib.zksp_hip_fetch_bodies(g.h, out.ctypes.data_as(C.c_void_p), out.size))
    roots = np.zeros((n, 8), np.uint32)
    g.check(g.lib.zksp_hip_fetch_roots(g.h, roots.ctypes.data_as(C.c_void_p), roots.size))
    assert np.array_equal(roots, out[:, :8])
    return out


@pytest.mark.parametrize("logh,nq,pow_bits", [(5, 3, 4), (7, 10, 8)])
def test_device_prover_matches_oracle_small(zk, oracle, logh, nq, pow_bits):
    g = Gpu(zk, num_queries=nq, pow_bits=pow_bits, max_batch=4)
    rng = np.random.default_rng(logh)
    vk = [int(x) for x in rng.integers(0, P, 8)]
    states, obs, exp = [], [], []
    for i in range(3):
        k = int(rng.integers(0, (1 << logh) // 24 + 1))
        st = rng.integers(0, 2**64, (k, 25), dtype=np.uint64)
        pvd = [int(x) for x in rng.integers(0, 2**32, 8)]
        states.append(st)
        obs.append(init_obs(vk, logh, k, 0, pvd, [0] * 8))
        exp.append(oracle.prove(st, logh, pv_digest=pvd, vk_digest=vk, num_queries=nq, pow_bits=pow_bits))
    bodies = device_bodies(g, logh, states, obs)
    for i in range(3):
        e = np.frombuffer(exp[i], dtype=np.uint32)[oracle.proof_header_words(0, len(states[i])):]
        assert e.shape == bodies[i].shape
        bad = np.nonzero(e != bodies[i])[0]
        assert bad.size == 0, (i, bad[:8])


@pytest.mark.parametrize("logh,n", [(13, 2), (14, 1)])
def test_device_prover_matches_oracle_tall(zk, oracle, logh, n):
    """Whole-proof parity at the largest supported trace heights (2^14 is the limit of load_batch):
    other LDE workgroup shapes, more tiles and shrinks in the opening kernels, deeper trees and
    more FRI layers than the 2^11 workload."""
    nq, pow_bits = 5, 6
    g = Gpu(zk, num_queries=nq, pow_bits=pow_bits, max_batch=n)
    rng = np.random.default_rng(200 + logh)
    vk = [int(x) for x in rng.integers(0, P, 8)]
    cap = (1 << logh) // 24
    states, obs, pvds = [], [], []
    for i in range(n):
        k = cap - 3 * i  # a full trace first, then a few padding permutations
        st = rng.integers(0, 2**64, (k, 25), dtype=np.uint64)
        pvd = [int(x) for x in rng.integers(0, 2**32, 8)]
        states.append(st)
        pvds.append(pvd)
        obs.append(init_obs(vk, logh, k, 0, pvd, [0] * 8))
    bodies = device_bodies(g, logh, states, obs)
    for i in range(n):
        exp = oracle.prove(states[i], logh, pv_digest=pvds[i], vk_digest=vk, num_queries=nq, pow_bits=pow_bits)
        e = np.frombuffer(exp, dtype=np.uint32)[oracle.proof_header_words(0, len(states[i])):]
        assert e.shape == bodies[i].shape
        bad = np.nonzero(e != bodies[i])[0]
        assert bad.size == 0, (i, bad[:8])


@pytest.mark.parametrize("logh,n", [(5, 128), (6, 96), (9, 128)])
def test_device_prover_matches_oracle_large_batch(zk, oracle, logh, n):
    """Large batches take different kernels from small ones: the LDS-transposed opening kernel needs
    ceil(2633 / 256) * batch >= 1024 workgroups; from 64 proofs on the FRI layers are committed
    one by one, and a layer with leaves * batch >= 32768 (2^9 x 128) and the tree levels that are
    wide across the batch use the lane-per-leaf / lane-per-parent kernels.  Every proof sampled
    from a wide batch must still equal the oracle's, including empty and full traces."""
    nq, pow_bits = 3, 4
    g = Gpu(zk, num_queries=nq, pow_bits=pow_bits, max_batch=n)
    rng = np.random.default_rng(100 + logh)
    vk = [int(x) for x in rng.integers(0, P, 8)]
    cap = (1 << logh) // 24
    states, obs = [], []
    for i in range(n):
        k = (0, cap)[i] if i < 2 else int(rng.integers(0, cap + 1))
        st = rng.integers(0, 2**64, (k, 25), dtype=np.uint64)
        pvd = [int(x) for x in rng.integers(0, 2**32, 8)]
        states.append(st)
        obs.append((init_obs(vk, logh, k, 0, pvd, [0] * 8), pvd))
    bodies = device_bodies(g, logh, states, [o for o, _ in obs])
    # the oracle is slow; a sample covers both ends of the batch, the edge traces and a spread
    sample = sorted(set([0, 1, 2, n // 2, n - 2, n - 1] + [int(x) for x in rng.integers(0, n, 6)]))
    for i in sample:
        exp = oracle.prove(states[i], logh, pv_digest=obs[i][1], vk_digest=vk, num_queries=nq, pow_bits=pow_bits)
        e = np.frombuffer(exp, dtype=np.uint32)[oracle.proof_header_words(0, len(states[i])):]
        bad = np.nonzero(e != bodies[i])[0]
        assert bad.size == 0, (i, bad[:8])
    # the trace commitment depends on the trace alone: as many distinct roots as distinct inputs
    roots = {bodies[i, :8].tobytes() for i in range(n)}
    assert len(roots) == len({st.tobytes() for st in states})


def test_end_to_end_acct_d8(zk, fx, oracle):
    """The keccak-chip component proof (format v2, proof_mode KECCAK_CHIP) through the reference-shaped
    flow: proof bytes identical to the oracle's, verifier accepts, tampering rejected.  The full statement
    (machine proof, the default mode) is tests/test_gpu_machine.py."""
    client = zk.ProverClient(device=0, proof_mode=zk.PROOF_KECCAK_CHIP)
    pk, vk = client.setup(zk.merkle_elf())
    inp = fx.acct_fixture(8)
    stdin = zk.SP1Stdin()
    stdin.write(inp.to_borsh())
    states = client.keccak_states(pk, stdin)
    assert states.shape == (62, 25)
    proof = client.prove(pk, stdin).run()
    assert proof.public_values == fx.ACCOUNT_VALUE
    client.verify(proof, vk)
    raw = proof.to_bytes()
    pvd = np.frombuffer(hashlib.sha256(fx.ACCOUNT_VALUE).digest(), dtype=np.uint32)
    exp = oracle.prove(states, 11, public_values=fx.ACCOUNT_VALUE, pv_digest=[int(x) for x in pvd],
                       vk_digest=[int(x) for x in np.frombuffer(vk.digest, dtype=np.uint32)])
    assert raw == exp
    # host-only client verifies too; every tampered region is rejected
    host = zk.ProverClient(device=-1, proof_mode=zk.PROOF_KECCAK_CHIP)
    host.verify(zk.SP1ProofWithPublicValues.from_bytes(raw), vk)
    for pos in (125, 30 * 4 + 72 + 3, 30 * 4 + 72 + 64 + 8, len(raw) // 2, len(raw) - 3):
        bad = bytearray(raw)
        bad[pos] ^= 1
        with pytest.raises(zk.ZkspError):
            host.verify(zk.SP1ProofWithPublicValues.from_bytes(bytes(bad)), vk)


def test_guest_panic_is_reported(zk, fx):
    client = zk.ProverClient(device=0)
    pk, vk = client.setup(zk.merkle_elf())
    inp = fx.acct_fixture(8)
    node = bytearray(inp.proof[3])
    node[-1] ^= 1
    inp.proof[3] = bytes(node)
    stdin = zk.SP1Stdin()
    stdin.write(inp.to_borsh())
    with pytest.raises(zk.GuestPanic) as ei:
        client.prove(pk, stdin).run()
    assert "Failed to verify Merkle Proof" in str(ei.value)


def test_batch_mixed_heights(zk, fx):
    client = zk.ProverClient(device=0, max_batch=3)
    pk, vk = client.setup(zk.merkle_elf())
    inputs = [fx.acct_fixture(8), fx.tx_fixture(), fx.slot_fixture(0), fx.slot_fixture(1), fx.receipt_fixture(5),
              fx.acct_fixture(1)]
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * len(inputs)
    from oracle import verify_merkle_proof
    for m, p in zip(inputs, proofs):
        client.verify(p, vk)
        assert p.public_values == verify_merkle_proof(m.root_hash, m.proof, m.key)


def test_benchmarked_configuration_matches_oracle_and_verifies(zk, fx, oracle):
    """The configuration bench.py times: trace height 2^11, a batch large enough (>= 64) to select the
    throughput kernels (lane-per-row leaf hash, lde_fixed_kernel<11, 2>, the LDS-transposed opening,
    per-layer FRI), full-size parameters (100 queries, 16 PoW bits), loaded through
    zksp_hip_load_batch / zksp_hip_prove_resident exactly as the benchmark does.  Sampled bodies are
    byte-identical to the oracle's and complete proofs built from them pass the host verifier."""
    n, logh = 64, 11
    client = zk.ProverClient(device=0, max_batch=n)
    g = Gpu.__new__(Gpu)
    g.zk, g.client, g.lib, g.h = zk, client, client._lib, client._h
    pk, vk = client.setup(zk.merkle_elf())
    vkw = [int(x) for x in np.frombuffer(vk.digest, dtype=np.uint32)]
    pv = fx.ACCOUNT_VALUE
    pvd = [int(x) for x in np.frombuffer(hashlib.sha256(pv).digest(), dtype=np.uint32)]
    states, obs = [], []
    rng = np.random.default_rng(77)
    for i in range(n):
        if i < 4:  # real guest runs: distinct depth-8 account proofs, 62 permutations each
            s = zk.SP1Stdin()
            s.write(fx.acct_fixture(8, seed=100 + i).to_borsh())
            st = client.keccak_states(pk, s)
            assert st.shape == (62, 25)
        else:      # random states, ragged permutation counts up to the capacity of the trace (85)
            st = rng.integers(0, 2**64, (int(rng.integers(1, 86)), 25), dtype=np.uint64)
        states.append(st)
        obs.append(init_obs(vkw, logh, len(st), 0, pvd, [0] * 8))
    bodies = device_bodies(g, logh, states, obs)
    for i in (0, 3, 17, n - 1):
        exp = oracle.prove(states[i], logh, public_values=pv, pv_digest=pvd, vk_digest=vkw)
        e = np.frombuffer(exp, dtype=np.uint32)[oracle.proof_header_words(len(pv), len(states[i])):]
        bad = np.nonzero(e != bodies[i])[0]
        assert bad.size == 0, (i, bad[:8])
    host = zk.ProverClient(device=-1, proof_mode=zk.PROOF_KECCAK_CHIP)
    for i in (0, 1, 2, 3, 9, 17, 31, 40, 55, n - 1):
        proof = zk.proof_from_body(bodies[i], logh, states[i], 0, pv, pvd, [0] * 8, vkw)
        assert proof.public_values == pv
        host.verify(proof, vk)
    # a body attached to another proof's inputs must not verify
    with pytest.raises(zk.ZkspError):
        host.verify(zk.proof_from_body(bodies[5], logh, states[6], 0, pv, pvd, [0] * 8, vkw), vk)


def test_batch_with_panicking_runs_in_a_mixed_call(zk, fx, oracle):
    """prove_batch over runs of several shapes, chunked (max_batch 2: every chunk is uploaded behind the one before it, of
    whatever shape), with two runs whose guest panics in the middle: their status is the guest's exit and they have no proof
    (the reference's run() fails for such an input, prover/src/bin/main.rs:71-74), every other run has a proof that verifies
    with the right public values, and one proof behind each failed run is the oracle's byte for byte."""
    from oracle import verify_merkle_proof
    client = zk.ProverClient(device=0, num_queries=6, pow_bits=5, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    inputs = [fx.acct_fixture(1, seed=3), fx.tx_fixture(), fx.acct_fixture(2, seed=5), fx.slot_fixture(0), fx.acct_fixture(1, seed=4),
              fx.receipt_fixture(5), fx.acct_fixture(2, seed=6), fx.slot_fixture(1), fx.acct_fixture(1, seed=9)]
    broken = (2, 5)
    for k in broken:
        node = bytearray(inputs[k].proof[-1])
        node[-1] ^= 1
        inputs[k].proof[-1] = bytes(node)
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    proofs, status = client.prove_batch(pk, stdins)
    for i, (m, p, st) in enumerate(zip(inputs, proofs, status)):
        if i in broken:
            assert st != 0 and p is None
            continue
        assert st == 0
        assert p.public_values == verify_merkle_proof(m.root_hash, m.proof, m.key)
        client.verify(p, vk)
    for i in (3, 6, 8):
        s = zk.SP1Stdin()
        s.write(inputs[i].to_borsh())
        raw = proofs[i].to_bytes()
        shape = [int.from_bytes(raw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
        assert raw == oracle.machine_prove(dict(client.machine_trace(pk, s), shape=shape), num_queries=6, pow_bits=5)
