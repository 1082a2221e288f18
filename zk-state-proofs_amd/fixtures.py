"""Synthetic Ethereum-MPT proof inputs (offline stand-in for the reference's
``trie-utils/src/proofs/*.rs`` JSON-RPC fetchers).

Everything here is host-side input preparation: keccak-256, RLP, hex-prefix
("compact") nibble encoding, the branch/leaf node builder and the borsh wire
format of ``MerkleProofInput`` (reference ``crypto-ops/src/types.rs:4-9``).
None of it is on the proving hot path; it only manufactures the byte strings
the guest ELF consumes.
"""
from __future__ import annotations

import random
import struct
from dataclasses import dataclass, field
from typing import List, Sequence

# --------------------------------------------------------------------------
# keccak-256 (original Keccak padding 0x01, as tiny-keccak's Keccak::v256,
# reference crypto-ops/src/keccak.rs:6-12)
# --------------------------------------------------------------------------
_MASK = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
# rotation offsets r[x][y]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]


def _rol(v: int, n: int) -> int:
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def keccak_f1600(lanes: List[int]) -> List[int]:
    """lanes[x + 5*y], 25 u64 words; returns the permuted state."""
    a = list(lanes)
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rol(a[x + 5 * y], _ROT[x][y])
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & _MASK & b[(i % 5 + 2) % 5 + 5 * (i // 5)])
             for i in range(25)]
        a[0] ^= _RC[rnd]
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        st = keccak_f1600(st)
    return b"".join(w.to_bytes(8, "little") for w in st[:4])


# --------------------------------------------------------------------------
# RLP
# --------------------------------------------------------------------------
def _rlp_len(n: int, base: int) -> bytes:
    if n < 56:
        return bytes([base + n])
    be = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([base + 55 + len(be)]) + be


def rlp_bytes(b: bytes) -> bytes:
    if len(b) == 1 and b[0] < 0x80:
        return bytes(b)
    return _rlp_len(len(b), 0x80) + bytes(b)


def rlp_list(items: Sequence[bytes]) -> bytes:
    """items are already-encoded RLP payloads."""
    body = b"".join(items)
    return _rlp_len(len(body), 0xC0) + body


def rlp_uint(v: int) -> bytes:
    if v == 0:
        return b"\x80"
    return rlp_bytes(v.to_bytes((v.bit_length() + 7) // 8, "big"))


def compact_nibbles(nibbles: Sequence[int], leaf: bool) -> bytes:
    """Hex-prefix encoding (yellow paper appendix C)."""
    flag = 2 if leaf else 0
    if len(nibbles) % 2:
        out = [((flag + 1) << 4) | nibbles[0]]
        rest = nibbles[1:]
    else:
        out = [flag << 4]
        rest = nibbles
    for i in range(0, len(rest), 2):
        out.append((rest[i] << 4) | rest[i + 1])
    return bytes(out)


# --------------------------------------------------------------------------
# wire types
# --------------------------------------------------------------------------
@dataclass
class MerkleProofInput:
    """Mirror of ``crypto_ops::types::MerkleProofInput``
    (reference crypto-ops/src/types.rs:4-9); field order is the borsh order."""
    proof: List[bytes] = field(default_factory=list)
    root_hash: bytes = b""
    key: bytes = b""

    def to_borsh(self) -> bytes:
        out = bytearray(struct.pack("<I", len(self.proof)))
        for node in self.proof:
            out += struct.pack("<I", len(node)) + node
        out += struct.pack("<I", len(self.root_hash)) + self.root_hash
        out += struct.pack("<I", len(self.key)) + self.key
        return bytes(out)

    @staticmethod
    def from_borsh(buf: bytes) -> "MerkleProofInput":
        off = 0

        def u32() -> int:
            nonlocal off
            (v,) = struct.unpack_from("<I", buf, off)
            off += 4
            return v

        def vec() -> bytes:
            nonlocal off
            n = u32()
            if off + n > len(buf):
                raise ValueError("borsh: truncated Vec<u8>")
            v = bytes(buf[off:off + n])
            off += n
            return v

        n = u32()
        proof = [vec() for _ in range(n)]
        root = vec()
        key = vec()
        if off != len(buf):
            raise ValueError("borsh: trailing bytes")
        return MerkleProofInput(proof, root, key)


@dataclass
class StorageProofInput:
    """Mirror of ``crypto_ops::types::StorageProofInput``
    (reference crypto-ops/src/types.rs:11-19)."""
    account_proof: List[bytes]
    storage_proofs: List[List[bytes]]
    root_hash: bytes
    account_key: bytes
    storage_keys: List[bytes]
    address_keccak: bytes  # [u8; 32], no length prefix

    def to_borsh(self) -> bytes:
        def vec(b: bytes) -> bytes:
            return struct.pack("<I", len(b)) + b

        def vecvec(v: Sequence[bytes]) -> bytes:
            return struct.pack("<I", len(v)) + b"".join(vec(x) for x in v)

        out = vecvec(self.account_proof)
        out += struct.pack("<I", len(self.storage_proofs))
        out += b"".join(vecvec(p) for p in self.storage_proofs)
        out += vec(self.root_hash) + vec(self.account_key) + vecvec(self.storage_keys)
        assert len(self.address_keccak) == 32
        return out + self.address_keccak


# --------------------------------------------------------------------------
# synthetic proof generator
# --------------------------------------------------------------------------
def _nibbles(key: bytes) -> List[int]:
    out = []
    for b in key:
        out += [b >> 4, b & 15]
    return out


def synth_proof(depth: int, key: bytes, value: bytes, seed: int) -> MerkleProofInput:
    """``depth`` nodes: depth-1 17-slot branch nodes above one leaf.

    Levels are filled bottom-up; in every branch the 15 sibling slots hold 32
    bytes each drawn from ``random.Random(seed).getrandbits(8)``, slots 0..15 in
    order, slot 16 is the empty string.  A child shorter than 32 bytes is
    embedded inline, otherwise referenced by ``rlp(keccak(child))``.
    """
    if depth < 1:
        raise ValueError("depth must be >= 1")
    nib = _nibbles(key)
    if depth - 1 > len(nib):
        raise ValueError("key too short for this depth")
    rng = random.Random(seed)
    leaf = rlp_list([rlp_bytes(compact_nibbles(nib[depth - 1:], True)), rlp_bytes(value)])
    nodes = [leaf]
    child = leaf
    for level in range(depth - 2, -1, -1):
        slots = []
        for s in range(16):
            if s == nib[level]:
                slots.append(child if len(child) < 32 else rlp_bytes(keccak256(child)))
            else:
                slots.append(rlp_bytes(bytes(rng.getrandbits(8) for _ in range(32))))
        slots.append(b"\x80")
        child = rlp_list(slots)
        nodes.append(child)
    nodes.reverse()
    return MerkleProofInput(proof=nodes, root_hash=keccak256(nodes[0]), key=bytes(key))


USDT_ADDRESS = bytes.fromhex("dAC17F958D2ee523a2206206994597C13D831ec7")
ACCOUNT_VALUE = bytes.fromhex("f8440180a0" + "11" * 32 + "a0" + "22" * 32)


def acct_fixture(depth: int = 8, seed: int = 1) -> MerkleProofInput:
    """BASELINE config 2: account-trie proof for keccak(USDT address)."""
    return synth_proof(depth, keccak256(USDT_ADDRESS), ACCOUNT_VALUE, seed)


def tx_fixture(seed: int = 1) -> MerkleProofInput:
    """BASELINE config 1 (tx-d2): key rlp(0), 111-byte typed-envelope value."""
    return synth_proof(2, b"\x80", b"\x02" + bytes(range(110)), seed)


def slot_fixture(i: int = 0, depth: int = 5) -> MerkleProofInput:
    """BASELINE config 3 (slot-d5[i]): key keccak(u256-BE(i)), seed i."""
    key = keccak256(i.to_bytes(32, "big"))
    return synth_proof(depth, key, rlp_uint(0x0DE0B6B3A7640000), i if i else 1)


def receipt_value(i: int) -> bytes:
    """0x02 || rlp([1, cumGas_i, 256-B zero bloom, 3 logs]) (749 bytes)."""
    logs = []
    for j in range(3):
        addr = bytes([0x10 + j]) * 20
        topics = [bytes([0xA0 + t]) * 32 for t in range(3)]
        data = bytes([i & 0xFF]) * 32
        logs.append(rlp_list([rlp_bytes(addr), rlp_list([rlp_bytes(t) for t in topics]), rlp_bytes(data)]))
    body = rlp_list([rlp_uint(1), rlp_uint(21000 * (i + 1)), rlp_bytes(bytes(256)), rlp_list(logs)])
    return b"\x02" + body


def receipt_fixture(i: int = 0, seed: int = 1) -> MerkleProofInput:
    """BASELINE config 4 (rcpt[i]): key rlp(i); depth 2 for i == 0 else 3."""
    key = rlp_uint(i)
    depth = 2 if i == 0 else 3
    return synth_proof(depth, key, receipt_value(i), seed)


def stdin_frame(buf: bytes) -> bytes:
    """``SP1Stdin::write(&Vec<u8>)`` framing: bincode Vec<u8> = u64-LE len || bytes
    (reference prover/src/bin/main.rs:69)."""
    return struct.pack("<Q", len(buf)) + buf
