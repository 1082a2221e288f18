"""Pins the CPU oracle: against the independent big-integer restatement
(tests/golden/stark_kat.json, made by tests/golden/gen_golden.py), against
universal known answers, and through algebraic self-checks.  PARITY UNPINNED vs
SP1/Plonky3 itself (no reference vectors exist)."""
import json
import os

import numpy as np
import pytest

P = 2013265921
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(HERE, "golden", "stark_kat.json")) as f:
        return json.load(f)


def test_poseidon2_constants(oracle, kat):
    ext, inn = oracle.poseidon2_constants()
    assert ext.tolist() == kat["ext_rc"]
    assert inn.tolist() == kat["int_rc"]
    assert all(0 <= v < P for row in kat["ext_rc"] for v in row)


def test_poseidon2_permute(oracle, kat):
    for case in kat["permute"]:
        assert oracle.poseidon2_permute(case["in"]).tolist() == case["out"]


def test_sponge_and_compress(oracle, kat):
    for case in kat["hash_elems"]:
        assert oracle.hash_elems(np.array(case["in"], np.uint32)).tolist() == case["out"]
    c = kat["compress"]
    assert oracle.compress(c["l"], c["r"]).tolist() == c["out"]


def test_challenger(oracle, kat):
    c = kat["challenger"]
    ch = oracle.OracleChallenger()
    ch.observe(c["observe1"])
    assert [ch.sample() for _ in c["sample1"]] == c["sample1"]
    ch.observe(c["observe2"])
    assert [ch.sample() for _ in c["sample2"]] == c["sample2"]


def test_ntt_against_naive_and_golden(oracle, kat):
    d = kat["dft"]
    assert oracle.ntt(np.array(d["in"], np.uint32)).tolist() == d["out"]
    rng = np.random.default_rng(0)
    for logn in range(0, 9):
        a = rng.integers(0, P, 1 << logn, dtype=np.uint32)
        f = oracle.ntt(a)
        assert (f == oracle.dft_naive(a)).all()
        assert (oracle.ntt(f, inverse=True) == a).all()


def test_coset_lde_is_horner_evaluation(oracle):
    rng = np.random.default_rng(1)
    logh, h = 5, 32
    cols = rng.integers(0, P, (3, h), dtype=np.uint32)
    for in_shift in (1, 31, 999):
        lde, coefs = oracle.coset_lde(cols, in_shift, True)
        wh = pow(31, (P - 1) >> logh, P)
        w2h = pow(31, (P - 1) >> (logh + 1), P)
        for col in range(3):
            poly = [int(c) for c in coefs[col]]
            ev = lambda x: sum(c * pow(x, k, P) for k, c in enumerate(poly)) % P
            # interpolates the input on in_shift * K_H
            for m in (0, 1, 7, 31):
                assert ev(in_shift * pow(wh, m, P) % P) == int(cols[col, m])
            for c in range(2):
                shift = 31 * (w2h if c else 1) % P
                for m in (0, 3, 31):
                    assert ev(shift * pow(wh, m, P) % P) == int(lde[col, c, m])


def test_merkle_tree_structure(oracle):
    rng = np.random.default_rng(2)
    mat = rng.integers(0, P, (5, 8), dtype=np.uint32)
    tree = oracle.merkle_commit(mat)
    assert tree.shape == (15, 8)
    for r in range(8):
        assert (tree[r] == oracle.hash_elems(mat[:, r])).all()
    off = 0
    for layer, cnt in enumerate((8, 4, 2)):
        nxt = off + cnt
        for i in range(cnt // 2):
            assert (tree[nxt + i] == oracle.compress(tree[off + 2 * i], tree[off + 2 * i + 1])).all()
        assert oracle.merkle_layer_offset(3, layer) == off
        off = nxt


def test_fri_fold_golden_and_semantics(oracle, kat):
    f = kat["fri_fold"]
    got = oracle.fri_fold(np.array(f["layer"], np.uint32), f["shift"], f["beta"])
    assert got.tolist() == f["out"]
    # folding a polynomial's evaluations gives evaluations of f_even + beta * f_odd
    rng = np.random.default_rng(3)
    loghk, hk = 4, 16
    coef = [int(x) for x in rng.integers(0, P, hk)]  # base-field poly of degree < Hk
    w, w2, shift = pow(31, (P - 1) >> loghk, P), pow(31, (P - 1) >> (loghk + 1), P), 31
    ev = lambda cs, x: sum(c * pow(x, k, P) for k, c in enumerate(cs)) % P
    layer = np.zeros((2, hk, 4), np.uint32)
    for c in range(2):
        for m in range(hk):
            layer[c, m, 0] = ev(coef, shift * (w2 if c else 1) * pow(w, m, P) % P)
    beta = [5, 0, 0, 0]
    out = oracle.fri_fold(layer, shift, beta)
    even, odd = coef[0::2], coef[1::2]
    for c in range(2):
        for m in range(hk // 2):
            x = shift * (w2 if c else 1) * pow(w, m, P) % P
            assert int(out[c, m, 0]) == (ev(even, x * x % P) + 5 * ev(odd, x * x % P)) % P
            assert out[c, m, 1:].tolist() == [0, 0, 0]


def test_keccak_f_known_answer(oracle, fx):
    # keccak-f on the zero state (first lanes of the well-known test vector)
    out = oracle.keccak_f(np.zeros(25, np.uint64))
    assert int(out[0]) == 0xF1258F7940E1DDE7 and int(out[1]) == 0x84D5CCF933C0478A
    assert [int(x) for x in out] == fx.keccak_f1600([0] * 25)


def test_keccak_air_accepts_real_traces_and_rejects_corruption(oracle):
    rng = np.random.default_rng(4)
    st = rng.integers(0, 2**64, (3, 25), dtype=np.uint64)
    logh, h = 7, 128
    tr = oracle.keccak_trace(st, logh)

    def violations(t):
        bad = 0
        for r in range(h):
            c = oracle.keccak_constraints(t[:, r], t[:, (r + 1) % h], int(r == 0), int(r == h - 1), int(r != h - 1))
            bad += int(c.any())
        return bad

    assert violations(tr) == 0
    # the final row of each real permutation carries keccak-f's output
    for p in range(3):
        out = oracle.keccak_f(st[p])
        row = tr[:, 24 * p + 23]
        lane0 = sum(int(row[2629 + l]) << (16 * l) for l in range(4))
        assert lane0 == int(out[0])
        for j in range(1, 25):
            assert sum(int(row[2465 + 4 * j + l]) << (16 * l) for l in range(4)) == int(out[j])
        assert row[24] == 1
    # single-cell corruptions in every column family are caught
    for col in (0, 24, 30, 130, 300, 600, 1000, 2470, 2570, 2630):
        t2 = tr.copy()
        t2[col, 50] = (int(t2[col, 50]) + 1) % P
        assert violations(t2) > 0, col


def test_quotient_is_a_polynomial_only_for_valid_traces(oracle):
    """The full oracle prover checks that FRI ends in a constant; a corrupted
    trace must break that (rc=3)."""
    rng = np.random.default_rng(5)
    st = rng.integers(0, 2**64, (1, 25), dtype=np.uint64)
    proof = oracle.prove(st, 5, num_queries=2, pow_bits=2)
    assert len(proof) == oracle.proof_size(5, 2, 2, 0, 1)


def test_logup_bus_running_sum(oracle):
    """phi is the exclusive running sum of export/f, its total equals the sum over the public
    I/O list, and the three bus constraints hold on the trace rows (a tampered multiplicity or
    output limb breaks the total)."""
    rng = np.random.default_rng(6)
    st = rng.integers(0, 2**64, (4, 25), dtype=np.uint64)
    logh, h = 7, 128
    tr = oracle.keccak_trace(st, logh)
    gamma, beta = rng.integers(0, P, 4, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    phi, cum = oracle.bus_perm_trace(tr, gamma, beta)
    io = oracle.bus_io_limbs(st)
    assert io.shape == (4, 200)
    # the I/O limbs are the input state and keccak-f of it
    out0 = oracle.keccak_f(st[0])
    assert [int(x) for x in io[0, 100:104]] == [(int(out0[0]) >> (16 * l)) & 0xFFFF for l in range(4)]
    assert (cum == oracle.bus_expected_sum(io, gamma, beta)).all()
    assert not phi[:, 0].any()  # phi_0 = 0
    # phi only moves after rows with export = 1 (the last round of each real permutation)
    changes = [r for r in range(h - 1) if (phi[:, r + 1] != phi[:, r]).any()]
    assert changes == [23, 47, 71, 95]  # phi steps right after each real permutation's export row
    tr2 = tr.copy()
    tr2[24, 23] = 0  # drop one export flag: the chip no longer receives that tuple
    assert (oracle.bus_perm_trace(tr2, gamma, beta)[1] != cum).any()
    io2 = io.copy()
    io2[2, 150] ^= 1  # a wrong public output limb
    assert (oracle.bus_expected_sum(io2, gamma, beta) != cum).any()
    assert oracle.bus_io_log_rows(11) == 12 and oracle.bus_io_log_rows(5) == 5


def test_verify_merkle_proof_restatement(oracle, fx):
    for m in (fx.acct_fixture(1), fx.acct_fixture(4), fx.acct_fixture(8), fx.tx_fixture(), fx.slot_fixture(3),
              fx.receipt_fixture(0), fx.receipt_fixture(7)):
        leaf = oracle.verify_merkle_proof(m.root_hash, m.proof, m.key)
        # leaf node = rlp([path, value]); the value must be what the generator embedded
        assert leaf in m.proof[-1]
    m = fx.acct_fixture(8)
    assert oracle.verify_merkle_proof(m.root_hash, m.proof, m.key) == fx.ACCOUNT_VALUE
    bad = fx.acct_fixture(8)
    node = bytearray(bad.proof[3]); node[-1] ^= 1; bad.proof[3] = bytes(node)
    with pytest.raises(ValueError):
        oracle.verify_merkle_proof(bad.root_hash, bad.proof, bad.key)
    with pytest.raises(ValueError):
        oracle.verify_merkle_proof(b"\x00" * 32, m.proof, m.key)
    # a key that leaves the proven path hits a node the proof does not carry
    with pytest.raises(ValueError):
        oracle.verify_merkle_proof(m.root_hash, m.proof, b"\xff" * 32)
    # a key that reaches the leaf but differs in its tail: "Key does not exist!"
    with pytest.raises(KeyError):
        oracle.verify_merkle_proof(m.root_hash, m.proof, m.key[:-1] + bytes([m.key[-1] ^ 1]))
