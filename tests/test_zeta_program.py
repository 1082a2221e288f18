"""Stage 2c, first piece: the constraint identity at zeta as a fixed
straight-line program of operations c = a * b + d over extension cells, recorded from the AIR templates the verifier
evaluates natively (csrc/host/zeta_program.cpp).  The program is cross-checked against the native evaluation on real proofs -
chip by chip, and the final combination over the heights' quotients - and is the same program whatever the proof's shape."""
import pytest

NQ, POW = 6, 5


@pytest.fixture(scope="module")
def proofs(zk, fx, built_lib, oracle):
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    pk, vk = client.setup(zk.merkle_elf())

    def prove(m, leaves=()):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        for lf in leaves:
            client.add_verified_leaf(s, lf, vk)
        statement = client.stdin_statement(s)
        t = client.machine_trace(pk, s)
        return zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)), statement

    leaf1, _ = prove(fx.acct_fixture(1, seed=11))
    leaf2, _ = prove(fx.slot_fixture(1))
    deep, _ = prove(fx.acct_fixture(3, seed=12))
    node, st_node = prove(fx.acct_fixture(1, seed=13), leaves=[leaf1, leaf2])  # Poseidon2, query and transcript chips with real rows
    return client, vk, [(leaf1, None), (leaf2, None), (deep, None), (node, st_node)]


def test_program_agrees_with_the_native_evaluation(zk, proofs):
    client, vk, cases = proofs
    sizes = set()
    for proof, own in cases:
        info = client.zeta_program_selftest(proof, vk, own)
        sizes.add((info["ops"], info["cells"], info["inputs"], info["constants"]))
        # the run's memory argument balanced too (every cell written once, read as often as the program says): part of the call
        assert 0 < info["inputs_read"] <= info["inputs"] and info["max_reads_of_a_cell"] > 1000  # (the constants 0 and 1)
        # ... and from the stub: the identity needs nothing of the query phase
        assert client.zeta_program_selftest(proof.stub(), vk, own) == info
    # one program for every shape: the heights enter through input cells only
    assert len(sizes) == 1
    ops, cells, inputs, consts = sizes.pop()
    # (a product that only feeds one addition is fused into it: fewer operations than cells that were numbered)
    assert 40_000 < ops < 120_000 and cells >= inputs + consts + 3 + ops and inputs > 8_000


def test_selftest_still_verifies(zk, proofs):
    """The self-test is a verification with an extra check: a proof that does not verify is refused as ever."""
    client, vk, cases = proofs
    raw = bytearray(cases[0][0].to_bytes())
    raw[-200] ^= 1
    with pytest.raises(zk.VerificationError):
        client.zeta_program_selftest(zk.SP1ProofWithPublicValues.from_bytes(bytes(raw)), vk)
    with pytest.raises(zk.ZkspError):
        client.zeta_program_selftest(cases[3][0], vk)  # a node without its statement
