"""In-memory Ethereum Merkle-Patricia trie: insert, root hash, proof extraction.

Offline replacement for what the reference does with ``eth_trie`` when it rebuilds a
block's transaction / receipt trie and calls ``trie.get_proof`` (reference
trie-utils/src/proofs/transaction.rs:41-73, trie-utils/src/proofs/receipt.rs:49-92,
trie-utils/src/receipt.rs:8-38).  SURVEY.md section 8f row f3: with it the prover is fed
proofs of the shape a real block produces (shared prefixes, extension nodes, embedded
short nodes) instead of random-sibling synthetic paths.  Host-side input preparation
only; nothing here is on the proving hot path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

from .fixtures import MerkleProofInput, compact_nibbles, keccak256, rlp_bytes, rlp_list, rlp_uint

EMPTY_ROOT = keccak256(b"\x80")


class _Leaf:
    __slots__ = ("path", "value")

    def __init__(self, path: List[int], value: bytes):
        self.path, self.value = path, value


class _Ext:
    __slots__ = ("path", "child")

    def __init__(self, path: List[int], child):
        self.path, self.child = path, child


class _Branch:
    __slots__ = ("children", "value")

    def __init__(self):
        self.children: List[Optional[object]] = [None] * 16
        self.value: bytes = b""


def _nibbles(key: bytes) -> List[int]:
    out: List[int] = []
    for b in key:
        out += [b >> 4, b & 15]
    return out


def _common(a: Sequence[int], b: Sequence[int]) -> int:
    n = 0
    while n < len(a) and n < len(b) and a[n] == b[n]:
        n += 1
    return n


class Trie:
    """Hexary Patricia trie over byte keys (the yellow paper's appendix D)."""

    def __init__(self):
        self.root = None
        self._enc_cache = {}  # id(node) -> RLP, valid until the next insert

    # ---- mutation ----
    def insert(self, key: bytes, value: bytes) -> None:
        if not value:
            raise ValueError("empty values are deletions; not supported")
        self._enc_cache.clear()
        self.root = self._insert(self.root, _nibbles(key), bytes(value))

    def _insert(self, node, path: List[int], value: bytes):
        if node is None:
            return _Leaf(path, value)
        if isinstance(node, _Leaf):
            n = _common(node.path, path)
            if n == len(node.path) == len(path):
                node.value = value
                return node
            br = _Branch()
            for p, v in ((node.path, node.value), (path, value)):
                if len(p) == n:
                    br.value = v
                else:
                    br.children[p[n]] = _Leaf(list(p[n + 1:]), v)
            return _Ext(list(path[:n]), br) if n else br
        if isinstance(node, _Ext):
            n = _common(node.path, path)
            if n == len(node.path):
                node.child = self._insert(node.child, path[n:], value)
                return node
            br = _Branch()
            rest = node.path[n + 1:]
            br.children[node.path[n]] = _Ext(list(rest), node.child) if rest else node.child
            if len(path) == n:
                br.value = value
            else:
                br.children[path[n]] = _Leaf(list(path[n + 1:]), value)
            return _Ext(list(path[:n]), br) if n else br
        # branch
        if not path:
            node.value = value
        else:
            node.children[path[0]] = self._insert(node.children[path[0]], path[1:], value)
        return node

    # ---- encoding ----
    def _encode(self, node) -> bytes:
        if node is None:
            return b"\x80"
        hit = self._enc_cache.get(id(node))
        if hit is not None:
            return hit
        if isinstance(node, _Leaf):
            enc = rlp_list([rlp_bytes(compact_nibbles(node.path, True)), rlp_bytes(node.value)])
        elif isinstance(node, _Ext):
            enc = rlp_list([rlp_bytes(compact_nibbles(node.path, False)), self._ref(node.child)])
        else:
            items = [self._ref(c) for c in node.children]
            items.append(rlp_bytes(node.value))
            enc = rlp_list(items)
        self._enc_cache[id(node)] = enc
        return enc

    def _ref(self, node) -> bytes:
        """How a parent refers to a child: the node itself if its RLP is shorter than
        32 bytes, else the RLP string of its keccak hash."""
        if node is None:
            return b"\x80"
        enc = self._encode(node)
        return enc if len(enc) < 32 else rlp_bytes(keccak256(enc))

    def root_hash(self) -> bytes:
        return EMPTY_ROOT if self.root is None else keccak256(self._encode(self.root))

    # ---- proofs ----
    def get_proof(self, key: bytes) -> List[bytes]:
        """RLP of every hashed node on the path of `key`, root first (embedded nodes travel
        inside their parent).  Also valid as an exclusion proof for an absent key."""
        proof: List[bytes] = []
        node, path, top = self.root, _nibbles(key), True
        while node is not None:
            enc = self._encode(node)
            if top or len(enc) >= 32:
                proof.append(enc)
            top = False
            if isinstance(node, _Leaf):
                break
            if isinstance(node, _Ext):
                if path[:len(node.path)] != node.path:
                    break
                path = path[len(node.path):]
                node = node.child
            else:
                if not path:
                    break
                node, path = node.children[path[0]], path[1:]
        return proof

    def get(self, key: bytes) -> Optional[bytes]:
        node, path = self.root, _nibbles(key)
        while node is not None:
            if isinstance(node, _Leaf):
                return node.value if node.path == path else None
            if isinstance(node, _Ext):
                if path[:len(node.path)] != node.path:
                    return None
                path, node = path[len(node.path):], node.child
            else:
                if not path:
                    return node.value or None
                node, path = node.children[path[0]], path[1:]
        return None


# ---------------------------------------------------------------------------
# block-shaped fixtures
# ---------------------------------------------------------------------------
def encode_log(address: bytes, topics: Sequence[bytes], data: bytes) -> bytes:
    """``trie_utils::types::Log`` (reference trie-utils/src/types.rs:11-35)."""
    return rlp_list([rlp_bytes(address), rlp_list([rlp_bytes(t) for t in topics]), rlp_bytes(data)])


def encode_receipt(status: bool, cumulative_gas: int, bloom: bytes, logs: Sequence[bytes], prefix: Optional[int]) -> bytes:
    """``insert_receipt``'s value: optional type byte || rlp([status, cumGas, bloom, logs])
    (reference trie-utils/src/receipt.rs:8-38)."""
    assert len(bloom) == 256
    body = rlp_list([rlp_uint(1 if status else 0), rlp_uint(cumulative_gas), rlp_bytes(bloom), rlp_list(list(logs))])
    return (bytes([prefix]) if prefix is not None else b"") + body


def synthetic_block_receipts(n: int, seed: int = 1) -> List[bytes]:
    """n receipts with the size mix of a busy block: mostly type-2, 0-4 logs each."""
    import random
    rng = random.Random(seed)
    out, gas = [], 0
    for i in range(n):
        gas += rng.randrange(21000, 400000)
        logs = []
        for _ in range(rng.choice((0, 1, 1, 2, 3, 4))):
            addr = bytes(rng.getrandbits(8) for _ in range(20))
            topics = [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(rng.randrange(1, 4))]
            data = bytes(rng.getrandbits(8) for _ in range(32 * rng.randrange(0, 4)))
            logs.append(encode_log(addr, topics, data))
        bloom = bytearray(256)
        for _ in range(3 * len(logs)):
            bloom[rng.randrange(256)] |= 1 << rng.randrange(8)
        prefix = rng.choice((None, 1, 2, 2, 2, 2, 3))
        out.append(encode_receipt(rng.random() > 0.03, gas, bytes(bloom), logs, prefix))
    return out


def block_trie(values: Sequence[bytes]) -> Trie:
    """Trie keyed by rlp(index), as both the transaction and the receipt trie are
    (reference trie-utils/src/proofs/transaction.rs:44-64, receipt.rs:55)."""
    t = Trie()
    for i, v in enumerate(values):
        t.insert(rlp_uint(i), v)
    return t


def block_proof_input(trie: Trie, index: int) -> MerkleProofInput:
    key = rlp_uint(index)
    return MerkleProofInput(proof=trie.get_proof(key), root_hash=trie.root_hash(), key=key)
