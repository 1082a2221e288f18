"""Row f4, stage 1 on the CPU: a machine proof with an aggregation payload - besides the guest's run it establishes the
Poseidon2 Merkle root of n 8-word digests (the Poseidon2 chip: one 2-to-1 compression per row, children in and parent out
over the DIGEST bus, leaves and root closed by the verifier).  The reference's recursion circuit is a todo!()
(circuits/sp1-merkle-proof-recursive/src/main.rs:3-5); this is the first building block of an in-circuit verifier and
what turns config 5's all-gathered commitments into one proven root.  The oracle proves, the product's host verifier
accepts with the right leaves and rejects everything else.  Runs without a GPU."""
import os

import numpy as np
import pytest

NQ, POW = 8, 6
P = 2013265921


@pytest.fixture(scope="module")
def setup(zk, fx, oracle, built_lib):
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    pk, vk = client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1).to_borsh())
    t = client.machine_trace(pk, s)
    leaves = np.random.default_rng(7).integers(0, P, (16, 8), dtype=np.uint32)
    t["agg_leaves"] = leaves
    proof = oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
    return client, vk, t, leaves, proof


def test_root_is_the_poseidon2_merkle_root(zk, oracle, setup):
    client, vk, t, leaves, proof = setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    n, root = p.aggregation
    assert n == 16 and root == oracle.machine_agg_public(leaves)[0]
    # independently: 15 compressions, level by level, with the oracle's compression function
    lv = [np.array(x, np.uint32) for x in leaves]
    while len(lv) > 1:
        lv = [np.array(oracle.compress(lv[2 * i], lv[2 * i + 1]), np.uint32) for i in range(len(lv) // 2)]
    assert [int(x) for x in lv[0]] == root
    client.verify_aggregate(p, vk, leaves)
    assert p.public_values == t["public_values"]


def test_wrong_leaves_and_plain_verify_are_rejected(zk, setup):
    client, vk, t, leaves, proof = setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    with pytest.raises(zk.VerificationError):
        client.verify(p, vk)  # a root without its leaves proves nothing
    bad = leaves.copy()
    bad[5, 1] ^= 1
    with pytest.raises(zk.VerificationError):
        client.verify_aggregate(p, vk, bad)
    with pytest.raises(zk.VerificationError):
        client.verify_aggregate(p, vk, leaves[:8])
    swapped = leaves.copy()
    swapped[[0, 1]] = swapped[[1, 0]]  # the order of the leaves is part of the statement
    with pytest.raises(zk.VerificationError):
        client.verify_aggregate(p, vk, swapped)


def test_another_root_is_rejected(zk, oracle, setup):
    """The header's root is what the Poseidon2 chip's last row puts on the DIGEST bus: a proof that names another root
    (and is otherwise the honest proof) does not balance; neither does one whose header names another leaf count."""
    client, vk, t, leaves, proof = setup
    hw = zk.MACHINE_HEADER_WORDS
    for word in (hw - 16, hw - 9):  # a word of the root, a word of the leaf-list digest
        bad = bytearray(proof)
        bad[4 * word] ^= 1
        with pytest.raises(zk.ZkspError):
            client.verify_aggregate(zk.SP1ProofWithPublicValues.from_bytes(bytes(bad)), vk, leaves)


def test_a_wrong_compression_is_rejected(zk, oracle, setup):
    """A prover that aggregates with one wrong inner node: the leaves it was given do not hash to what the Poseidon2
    chip's rows consume, so the honest oracle cannot even build the proof, and the forced proof is rejected."""
    client, vk, t, leaves, proof = setup
    t2 = dict(t)
    l2 = leaves.copy()
    l2[9, 0] = (int(l2[9, 0]) + 1) % P
    t2["agg_leaves"] = l2
    forged = oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)  # an honest proof of ANOTHER list ...
    q = zk.SP1ProofWithPublicValues.from_bytes(forged)
    client.verify_aggregate(q, vk, l2)
    with pytest.raises(zk.VerificationError):
        client.verify_aggregate(q, vk, leaves)  # ... does not pass for this one


def test_payload_sizes(zk, oracle, setup):
    client, vk, t, leaves, _ = setup
    for n in (2, 4, 64):
        lv = np.random.default_rng(n).integers(0, P, (n, 8), dtype=np.uint32)
        t2 = dict(t, agg_leaves=lv)
        p = zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW))
        assert p.aggregation[0] == n
        client.verify_aggregate(p, vk, lv)
    s = zk.SP1Stdin()
    for bad_n in (1, 3, 12):
        with pytest.raises(zk.ZkspError):
            s.set_aggregation(np.zeros((bad_n, 8), np.uint32))
    with pytest.raises(zk.ZkspError):
        s.set_aggregation(np.full((2, 8), P, np.uint32))  # not a canonical field word


# ---- Merkle path: the same chip with digests supplied at heap keys of the caller's choice ----
def _tree(oracle, leaves):
    """heap of digests (dict key -> list of 8 ints) of the full tree over `leaves`"""
    n = len(leaves)
    heap = {n + i: [int(x) for x in leaves[i]] for i in range(n)}
    for k in range(n - 1, 0, -1):
        heap[k] = [int(x) for x in oracle.compress(np.array(heap[2 * k], np.uint32), np.array(heap[2 * k + 1], np.uint32))]
    return heap


@pytest.fixture(scope="module")
def path_setup(zk, fx, oracle, setup):
    client, vk, t, leaves, _ = setup
    heap = _tree(oracle, leaves)
    index, depth = 11, 4
    k0 = (1 << depth) + index
    siblings = [heap[(k0 >> j) ^ 1] for j in range(depth)]
    keys, digests = zk.merkle_path_nodes(index, heap[k0], siblings)
    t2 = dict(t, agg_leaves=digests, agg_keys=keys)
    proof = oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)
    return client, vk, t2, heap, index, siblings, proof


def test_merkle_path_gives_the_tree_root(zk, oracle, path_setup):
    """A leaf and the siblings along its path are enough: the proof's root is the root of the tree the path was cut
    from (four rows of the Poseidon2 chip instead of fifteen), and the host verifier accepts it with that path only."""
    client, vk, t2, heap, index, siblings, proof = path_setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    n, root = p.aggregation
    assert n == 5 and root == heap[1]
    assert oracle.machine_heights(t2)[zk.MACHINE_CHIP_NAMES.index("poseidon2")] == 5
    client.verify_merkle_path(p, vk, index, heap[16 + index], siblings)
    assert p.public_values == t2["public_values"]


def test_merkle_path_binds_leaf_position_and_siblings(zk, path_setup):
    client, vk, t2, heap, index, siblings, proof = path_setup
    p = zk.SP1ProofWithPublicValues.from_bytes(proof)
    leaf = heap[16 + index]
    with pytest.raises(zk.VerificationError):
        client.verify(p, vk)
    with pytest.raises(zk.VerificationError):
        client.verify_merkle_path(p, vk, index ^ 1, leaf, siblings)  # the neighbouring position
    with pytest.raises(zk.VerificationError):
        client.verify_merkle_path(p, vk, index, heap[16 + (index ^ 1)], siblings)  # another leaf
    bad = [list(s) for s in siblings]
    bad[2][0] ^= 1
    with pytest.raises(zk.VerificationError):
        client.verify_merkle_path(p, vk, index, leaf, bad)
    with pytest.raises(zk.VerificationError):
        client.verify_merkle_path(p, vk, index & 7, leaf, siblings[:3])  # a shorter path
    with pytest.raises(zk.VerificationError):
        client.verify_aggregate(p, vk, np.array([leaf] + siblings, np.uint32)[:4])  # the digests read as tree leaves


def test_malformed_node_sets_are_refused(zk, path_setup):
    """Every ancestor of a supplied key needs both children, and no supplied node may be another's ancestor."""
    client, vk, t2, heap, index, siblings, _ = path_setup
    s = zk.SP1Stdin()
    leaf = heap[16 + index]
    keys, digests = zk.merkle_path_nodes(index, leaf, siblings)
    lib = s._lib
    C = __import__("ctypes")

    def rc_of(k, d):
        k, d = np.ascontiguousarray(k, np.uint32), np.ascontiguousarray(d, np.uint32)
        return lib.zksp_stdin_set_aggregation_keyed(s._h, k.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), len(k))

    assert rc_of(keys, digests) == 0
    assert rc_of(keys[:-1], digests[:-1]) != 0  # the top sibling is missing: node 1 lacks a child
    k2 = keys.copy()
    k2[1] = k2[0]
    assert rc_of(k2, digests) != 0  # a repeated key
    k3 = keys.copy()
    k3[2] = keys[0] >> 1
    assert rc_of(k3, digests) != 0  # a supplied node that is the leaf's parent
