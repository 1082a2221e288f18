"""Storage-proof statement as a host-side composition of Merkle-proof guest runs
(gives BASELINE config 3 its real semantics).

The reference's only definition of a "storage proof" is the risc0 circuit
(reference circuits/risc0-storage-proof/storage-proof-circuit/storage-circuit/src/main.rs:6-31):
verify the account proof under ``address_keccak``, RLP-decode the ``Account``, verify every
storage proof against ``decoded_account.storage_root`` under ``keccak(storage_key)``, commit
the stored values.  No SP1 build of that circuit exists and no guest can be compiled here,
so the same statement is composed from 1 + N runs of the committed sp1-merkle-proof guest,
each proven on the GPU.

Limit: the committed guest commits only the leaf value, not the root or
key it was run with, and round 1 proves the keccak chip only -- so the link "slot proofs used
the account's storage_root" is checked by this host code, not by the proofs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

from .fixtures import MerkleProofInput, StorageProofInput, keccak256, rlp_bytes, rlp_list, rlp_uint
from .mpt import Trie


def _rlp_items(buf: bytes) -> List[bytes]:
    """Payloads of the items of one RLP list (flat: items themselves are not recursed)."""
    if not buf or buf[0] < 0xC0:
        raise ValueError("not an RLP list")
    if buf[0] < 0xF8:
        n, pos = buf[0] - 0xC0, 1
    else:
        ll = buf[0] - 0xF7
        n, pos = int.from_bytes(buf[1:1 + ll], "big"), 1 + ll
    if pos + n != len(buf):
        raise ValueError("RLP list length mismatch")  # decode_exact: no trailing bytes
    out = []
    while pos < len(buf):
        b0 = buf[pos]
        if b0 < 0x80:
            out.append(buf[pos:pos + 1]); pos += 1
        elif b0 < 0xB8:
            m = b0 - 0x80
            out.append(buf[pos + 1:pos + 1 + m]); pos += 1 + m
        elif b0 < 0xC0:
            ll = b0 - 0xB7
            m = int.from_bytes(buf[pos + 1:pos + 1 + ll], "big")
            out.append(buf[pos + 1 + ll:pos + 1 + ll + m]); pos += 1 + ll + m
        else:
            raise ValueError("nested list in account RLP")
    return out


@dataclass
class Account:
    """``alloy_consensus::Account`` as the reference decodes it (storage-circuit/src/main.rs:16)."""
    nonce: int
    balance: int
    storage_root: bytes
    code_hash: bytes

    @staticmethod
    def decode_exact(buf: bytes) -> "Account":
        it = _rlp_items(buf)
        if len(it) != 4 or len(it[2]) != 32 or len(it[3]) != 32:
            raise ValueError("not an account RLP")
        return Account(int.from_bytes(it[0], "big"), int.from_bytes(it[1], "big"), bytes(it[2]), bytes(it[3]))

    def encode(self) -> bytes:
        return rlp_list([rlp_uint(self.nonce), rlp_uint(self.balance), rlp_bytes(self.storage_root), rlp_bytes(self.code_hash)])


@dataclass
class StorageProofResult:
    account_proof: object        # SP1ProofWithPublicValues
    slot_proofs: List[object]
    account: Account
    values: List[bytes]          # what the reference circuit commits: one leaf value per slot


def prove_storage_proof(client, pk, inp: StorageProofInput) -> StorageProofResult:
    """storage-circuit/src/main.rs:6-31, one GPU-proven guest run per verify_merkle_proof call."""
    from .client import SP1Stdin, ZkspError
    s = SP1Stdin()
    s.write(MerkleProofInput(list(inp.account_proof), inp.root_hash, inp.address_keccak).to_borsh())
    acct_proof = client.prove(pk, s).run()
    account = Account.decode_exact(acct_proof.public_values)
    stdins = []
    for proof, key in zip(inp.storage_proofs, inp.storage_keys):
        si = SP1Stdin()
        si.write(MerkleProofInput(list(proof), account.storage_root, keccak256(key)).to_borsh())
        stdins.append(si)
    proofs, status = client.prove_batch(pk, stdins) if stdins else ([], [])
    for i, st in enumerate(status):
        if st != 0:
            raise ZkspError(st, f"storage slot {i}: {client.last_error()}")
    return StorageProofResult(acct_proof, proofs, account, [p.public_values for p in proofs])


def verify_storage_proof(client, vk, inp: StorageProofInput, res: StorageProofResult) -> None:
    """Every proof verifies, the account proof's public values decode to the account whose
    storage_root the slot runs were given, and the committed values are the slot proofs'
    public values."""
    client.verify(res.account_proof, vk)
    account = Account.decode_exact(res.account_proof.public_values)
    if account != res.account:
        raise ValueError("account does not match the account proof's public values")
    if len(res.slot_proofs) != len(inp.storage_keys):
        raise ValueError("slot proof count mismatch")
    for p, v in zip(res.slot_proofs, res.values):
        client.verify(p, vk)
        if p.public_values != v:
            raise ValueError("committed value does not match the slot proof")


def synthetic_storage_proof_input(n_slots: int, seed: int = 1, n_other_accounts: int = 20,
                                  n_other_slots: int = 40) -> Tuple[StorageProofInput, List[bytes]]:
    """A state trie with one contract whose storage trie holds the proven slots.
    Returns (input, expected committed values)."""
    import random
    rng = random.Random(seed)
    rb = lambda n: bytes(rng.getrandbits(8) for _ in range(n))
    storage = Trie()
    keys, values = [], []
    for i in range(n_slots + n_other_slots):
        key = i.to_bytes(32, "big") if i < n_slots else rb(32)
        val = rlp_uint(rng.getrandbits(rng.choice((8, 64, 160, 256))) or 1)
        storage.insert(keccak256(key), val)
        if i < n_slots:
            keys.append(key)
            values.append(val)
    address = rb(20)
    account = Account(nonce=1, balance=rng.getrandbits(70), storage_root=storage.root_hash(), code_hash=rb(32))
    state = Trie()
    state.insert(keccak256(address), account.encode())
    for _ in range(n_other_accounts):
        state.insert(keccak256(rb(20)), Account(rng.getrandbits(8), rng.getrandbits(64), rb(32), rb(32)).encode())
    addr_hash = keccak256(address)
    inp = StorageProofInput(
        account_proof=state.get_proof(addr_hash),
        storage_proofs=[storage.get_proof(keccak256(k)) for k in keys],
        root_hash=state.root_hash(),
        account_key=addr_hash,
        storage_keys=keys,
        address_keccak=addr_hash,
    )
    return inp, values
