"""Row f2 on CPU: the storage-proof composition's inputs are what the reference's circuit
(circuits/risc0-storage-proof/.../storage-circuit/src/main.rs:6-31) would accept -- checked
natively (oracle restatement) and through the committed guest."""
import importlib

import pytest


def test_synthetic_storage_input_verifies_natively(oracle, fx):
    storage = importlib.import_module("zk-state-proofs_amd.storage")
    inp, expected = storage.synthetic_storage_proof_input(n_slots=6, seed=9)
    acct_rlp = oracle.verify_merkle_proof(inp.root_hash, inp.account_proof, inp.address_keccak)
    account = storage.Account.decode_exact(acct_rlp)
    assert account.encode() == acct_rlp
    got = [oracle.verify_merkle_proof(account.storage_root, p, fx.keccak256(k))
           for p, k in zip(inp.storage_proofs, inp.storage_keys)]
    assert got == expected
    # borsh layout of StorageProofInput (reference crypto-ops/src/types.rs:11-19): the fixed
    # 32-byte address_keccak has no length prefix and comes last
    blob = inp.to_borsh()
    assert blob.endswith(inp.address_keccak) and len(inp.address_keccak) == 32


def test_guest_runs_of_the_composition(zk, host_client, fx):
    storage = importlib.import_module("zk-state-proofs_amd.storage")
    inp, expected = storage.synthetic_storage_proof_input(n_slots=3, seed=2)
    pk, _ = host_client.setup(zk.merkle_elf())

    def run(m):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        rep, pv, err, rc = host_client.execute(pk, s, zk.KECCAK_REPLACE)
        assert rc == 0 and rep.exit_code == 0, err
        return pv

    acct = storage.Account.decode_exact(run(fx.MerkleProofInput(inp.account_proof, inp.root_hash, inp.address_keccak)))
    vals = [run(fx.MerkleProofInput(p, acct.storage_root, fx.keccak256(k)))
            for p, k in zip(inp.storage_proofs, inp.storage_keys)]
    assert vals == expected


def test_account_decode_exact_rejects_malformed():
    storage = importlib.import_module("zk-state-proofs_amd.storage")
    a = storage.Account(3, 10**18, b"\x11" * 32, b"\x22" * 32)
    enc = a.encode()
    assert storage.Account.decode_exact(enc) == a
    for bad in (enc + b"\x00", enc[:-1], b"\x80", b"", enc.replace(b"\x11" * 32, b"\x11" * 31 + b"")):
        with pytest.raises(ValueError):
            storage.Account.decode_exact(bad)
