"""Row f3: offline trie construction + get_proof, checked against
the public trie test vector, the reference's native verification restated in oracle/,
and the committed guest itself (executor)."""
import importlib

import pytest


@pytest.fixture(scope="module")
def mpt():
    return importlib.import_module("zk-state-proofs_amd.mpt")


def test_public_trie_vector(mpt):
    # ethereum/tests TrieTests/trieanyorder.json "dogs"
    t = mpt.Trie()
    for k, v in ((b"do", b"verb"), (b"dog", b"puppy"), (b"doge", b"coin"), (b"horse", b"stallion")):
        t.insert(k, v)
    assert t.root_hash().hex() == "5991bb8c6514148a29db676a14ac506cd2cd5775ace63c30a4fe457715e9ac84"
    assert mpt.Trie().root_hash() == mpt.EMPTY_ROOT
    # insertion order must not matter
    t2 = mpt.Trie()
    for k, v in ((b"horse", b"stallion"), (b"doge", b"coin"), (b"do", b"verb"), (b"dog", b"puppy")):
        t2.insert(k, v)
    assert t2.root_hash() == t.root_hash()
    assert t.get(b"dog") == b"puppy" and t.get(b"cat") is None


def test_receipt_value_matches_reference_vector(mpt):
    """Same bytes as the reference's hermetic RLP test (trie-utils/tests/rlp.rs:12)."""
    log = mpt.encode_log(bytes(19) + b"\x11", [bytes(30) + b"\xde\xad", bytes(30) + b"\xbe\xef"], bytes.fromhex("0100ff"))
    got = mpt.encode_receipt(False, 1, bytes(256), [log], None)
    assert got.hex().startswith("f901668001b90100") and got.hex().endswith("beef830100ff") and len(got) == 361


def test_block_receipt_trie_proofs_verify_natively(mpt, oracle):
    receipts = mpt.synthetic_block_receipts(300, seed=7)
    trie = mpt.block_trie(receipts)
    root = trie.root_hash()
    depths = set()
    for i in (0, 1, 15, 16, 127, 128, 129, 255, 256, 299):
        inp = mpt.block_proof_input(trie, i)
        assert inp.root_hash == root
        assert oracle.verify_merkle_proof(inp.root_hash, inp.proof, inp.key) == receipts[i]
        depths.add(len(inp.proof))
    assert len(depths) >= 2  # indices 0..127, 128..255 and >= 256 sit at different depths
    # an index outside the block: a valid exclusion proof, "Key does not exist!"
    absent = mpt.block_proof_input(trie, 300)
    with pytest.raises((KeyError, ValueError)):
        oracle.verify_merkle_proof(absent.root_hash, absent.proof, absent.key)


def test_guest_accepts_block_shaped_proofs(mpt, zk, host_client, oracle):
    """The committed guest (eth_trie compiled for RV32IM) walks proofs produced by this trie:
    the strongest offline check that node encoding and proof extraction are right."""
    pk, _ = host_client.setup(zk.merkle_elf())
    receipts = mpt.synthetic_block_receipts(300, seed=3)
    trie = mpt.block_trie(receipts)
    for i in (0, 5, 128, 299):
        inp = mpt.block_proof_input(trie, i)
        s = zk.SP1Stdin()
        s.write(inp.to_borsh())
        rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_REPLACE)
        assert rc == 0 and rep.exit_code == 0, err
        assert pv == receipts[i]
    # tampering with a node deep in the path makes the guest panic like the reference
    inp = mpt.block_proof_input(trie, 200)
    node = bytearray(inp.proof[-1])
    node[-1] ^= 1
    inp.proof[-1] = bytes(node)
    s = zk.SP1Stdin()
    s.write(inp.to_borsh())
    rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_REPLACE)
    assert rep.exit_code != 0 and pv == b""


def test_small_tries_and_embedded_nodes(mpt, oracle):
    # short values: leaves shorter than 32 bytes are embedded in their parents
    t = mpt.Trie()
    items = {bytes([i]): bytes([i]) * 3 for i in range(1, 40)}
    for k, v in items.items():
        t.insert(k, v)
    for k, v in items.items():
        proof = t.get_proof(k)
        assert oracle.verify_merkle_proof(t.root_hash(), proof, k) == v
    # single-entry trie: the root is the leaf
    one = mpt.Trie()
    one.insert(b"\x80", b"x" * 50)
    assert oracle.verify_merkle_proof(one.root_hash(), one.get_proof(b"\x80"), b"\x80") == b"x" * 50
