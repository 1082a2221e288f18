"""(Component tests: run with ZKSP_COMPONENT=1, which builds and loads libzksp_component.so; skipped otherwise.)
Host verifier of the keccak-chip COMPONENT proof (format v2; a kernel benchmark, not a proof of execution)
against proofs made by the CPU oracle: accepts honest proofs, rejects every kind of tampering, wrong keys and
wrong parameters.  Such proofs verify only on a client created with proof_mode=PROOF_KECCAK_CHIP: the default
(MACHINE) client of the reference-shaped flow refuses them (test_default_client_rejects_component_proofs).
Runs without a GPU."""
import hashlib

import numpy as np
import pytest

P = 2013265921
NQ, POW = 12, 8


@pytest.fixture(scope="module")
def setup(zk, oracle, built_lib):
    if not zk.client.COMPONENT:
        pytest.skip("the keccak-chip component path is a build switch (ZKSP_COMPONENT=1; include/zksp_component.h)")
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW, proof_mode=zk.PROOF_KECCAK_CHIP)
    pk, vk = client.setup(zk.merkle_elf())
    vk_words = [int(x) for x in np.frombuffer(vk.digest, dtype=np.uint32)]
    rng = np.random.default_rng(11)
    st = rng.integers(0, 2**64, (4, 25), dtype=np.uint64)
    pv = b"leaf value bytes"
    pvd = [int(x) for x in np.frombuffer(hashlib.sha256(pv).digest(), dtype=np.uint32)]
    proof = oracle.prove(st, 7, public_values=pv, pv_digest=pvd, vk_digest=vk_words, num_queries=NQ, pow_bits=POW)
    return client, vk, vk_words, st, pv, pvd, proof


def test_accepts_oracle_proof(zk, setup):
    client, vk, _, _, pv, _, proof = setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    assert p.public_values == pv
    client.verify(p, vk)
    assert p.to_bytes() == proof


def test_default_client_rejects_component_proofs(zk, setup):
    """The drop-in `client.verify(&proof, &vk)` (reference prover/src/bin/main.rs:80) accepts proofs of the guest's
    execution only: a default client must refuse a keccak-chip component proof, whose public values are arbitrary
    bytes, whatever version word the proof carries."""
    _, vk, _, _, _, _, proof = setup
    default = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    with pytest.raises(zk.VerificationError) as ei:
        default.verify(zk.SP1ProofWithPublicValues.from_bytes(proof), vk)
    assert "not a machine proof" in str(ei.value)


def test_rejects_tampering_everywhere(zk, setup):
    client, vk, _, _, pv, _, proof = setup
    words = len(proof) // 4
    hdr = 30 + (len(pv) + 3) // 4
    rng = np.random.default_rng(0)
    # header fields, public values, roots, opened values, FRI roots, final poly, witness, query data
    positions = [2, 3, 4, 6, 14, 22, 30, hdr, hdr + 8, hdr + 16, hdr + 16 + 4 * 2633, hdr + 16 + 4 * (2 * 2633 + 8),
                 hdr + 16 + 4 * (2 * 2633 + 8) + 8 * 7, hdr + 16 + 4 * (2 * 2633 + 8) + 8 * 7 + 4, words - 1]
    positions += [int(x) for x in rng.integers(hdr, words, 25)]
    for w in positions:
        bad = bytearray(proof)
        bad[4 * w] ^= 1
        try:
            q = zk.SP1ProofWithPublicValues.from_bytes(bytes(bad))
        except zk.ZkspError:
            continue  # malformed header is rejected at deserialisation
        with pytest.raises(zk.ZkspError):
            client.verify(q, vk)


def test_rejects_wrong_parameters_and_keys(zk, setup, oracle):
    client, vk, vk_words, st, pv, pvd, proof = setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    other = zk.ProverClient(device=-1, num_queries=NQ + 1, pow_bits=POW, proof_mode=zk.PROOF_KECCAK_CHIP)
    with pytest.raises(zk.ZkspError):
        other.verify(p, vk)
    harder = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW + 9, proof_mode=zk.PROOF_KECCAK_CHIP)
    with pytest.raises(zk.ZkspError):
        harder.verify(p, vk)
    # a proof bound to another verifying key
    wrong_vk = [(w + 1) % P for w in vk_words]
    q = oracle.prove(st, 7, public_values=pv, pv_digest=pvd, vk_digest=wrong_vk, num_queries=NQ, pow_bits=POW)
    with pytest.raises(zk.VerificationError):
        client.verify(zk.SP1ProofWithPublicValues.from_bytes(q), vk)
    # public values that do not match the committed digest
    bad_pv = oracle.prove(st, 7, public_values=b"another value   ", pv_digest=pvd, vk_digest=vk_words, num_queries=NQ,
                          pow_bits=POW)
    with pytest.raises(zk.VerificationError):
        client.verify(zk.SP1ProofWithPublicValues.from_bytes(bad_pv), vk)


def test_rejects_truncated_and_garbage(zk, setup):
    client, vk, *_, proof = setup
    for blob in (b"", b"ZKSP", proof[:100], proof[:-4], proof + b"\0\0\0\0", bytes(len(proof))):
        try:
            q = zk.SP1ProofWithPublicValues.from_bytes(blob)
        except zk.ZkspError:
            continue
        with pytest.raises(zk.ZkspError):
            client.verify(q, vk)


def test_heights_and_empty_trace(zk, oracle, setup):
    client, vk, vk_words, *_ = setup
    pvd = [int(x) for x in np.frombuffer(hashlib.sha256(b"").digest(), dtype=np.uint32)]
    rng = np.random.default_rng(3)
    for logh, k in ((5, 0), (5, 1), (6, 2), (8, 10)):
        st = rng.integers(0, 2**64, (k, 25), dtype=np.uint64)
        pr = oracle.prove(st, logh, pv_digest=pvd, vk_digest=vk_words, num_queries=NQ, pow_bits=POW)
        client.verify(zk.SP1ProofWithPublicValues.from_bytes(pr), vk)


def test_rejects_nonzero_exit_code_and_oversized_height(zk, oracle, setup):
    """A panicked guest has no proof in the reference (run() fails, prover/src/bin/main.rs:71-74): an
    otherwise valid proof whose header carries exit code 1 is rejected.  A header claiming a height
    the prover never emits is refused at parse time, before any hashing."""
    client, vk, vk_words, st, pv, pvd, proof = setup
    bad = oracle.prove(st, 7, exit_code=1, public_values=pv, pv_digest=pvd, vk_digest=vk_words, num_queries=NQ,
                       pow_bits=POW)
    with pytest.raises(zk.VerificationError) as ei:
        client.verify(zk.SP1ProofWithPublicValues.from_bytes(bad), vk)
    assert "exit code" in str(ei.value)
    tall = bytearray(proof)
    tall[8:12] = (26).to_bytes(4, "little")   # log_h
    tall[12:16] = (0).to_bytes(4, "little")   # n_perms
    with pytest.raises(zk.ZkspError):
        zk.SP1ProofWithPublicValues.from_bytes(bytes(tall))


def test_host_poseidon2_vector_matches_scalar(zk, built_lib):
    """The host verifier's permutation in 256-bit registers (csrc/host/p2_avx2.cpp, used where the CPU has AVX2), one state at a
    time or two in lockstep, is the same function as the scalar form: the independent restatement's known answers (tests/golden/stark_kat.json), random states and
    states of extreme words, bit for bit."""
    import ctypes as C
    import json
    import os
    lib = zk.client.load_library()
    lib.zksp_host_poseidon2_permute.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    P = 2013265921
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stark_kat.json")) as f:
        kat = json.load(f)["permute"]
    rng = np.random.default_rng(77)
    words = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 0x0ffffffe, 1 << 30], dtype=np.uint32)
    states = np.concatenate([np.array([c["in"] for c in kat], dtype=np.uint32),
                             rng.integers(0, P, size=(4000, 16), dtype=np.uint32),
                             words[rng.integers(0, len(words), (1000, 16))],
                             np.repeat(words[:, None], 16, axis=1)])
    scalar, vector = states.copy(), states.copy()
    assert lib.zksp_host_poseidon2_permute(scalar.ctypes.data, len(scalar), 0) == 0
    assert scalar[:len(kat)].tolist() == [c["out"] for c in kat]
    rc = lib.zksp_host_poseidon2_permute(vector.ctypes.data, len(vector), 1)
    if rc == zk.client.ERR_UNSUPPORTED:
        pytest.skip("this CPU has no AVX2: the verifier runs the scalar permutation")
    assert rc == 0 and np.array_equal(scalar, vector)
    for n_states in (len(states), len(states) - 1, 1):  # pairs in lockstep; an odd one out runs alone
        paired = states[:n_states].copy()
        assert lib.zksp_host_poseidon2_permute(paired.ctypes.data, n_states, 2) == 0
        assert np.array_equal(paired, scalar[:n_states])
    # the AVX-512 forms (a state per register; four states in lockstep), where the CPU has them
    for impl, sizes in ((3, (len(states),)), (4, (len(states), len(states) - 1, len(states) - 2, 3))):
        for n_states in sizes:
            wide = states[:n_states].copy()
            rc = lib.zksp_host_poseidon2_permute(wide.ctypes.data, n_states, impl)
            if rc == zk.client.ERR_UNSUPPORTED:
                break
            assert rc == 0 and np.array_equal(wide, scalar[:n_states])
    bad = states[:1].copy()
    bad[0, 3] = P  # not a canonical word
    assert lib.zksp_host_poseidon2_permute(bad.ctypes.data, 1, 0) != 0
