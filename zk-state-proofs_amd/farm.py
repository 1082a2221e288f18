"""Multi-GPU proof farm.

Every BASELINE throughput config is a set of independent (ELF, stdin) pairs, and
the reference proves one input per call with no shared state (reference
prover/src/bin/main.rs:59-87), so the path shards by proof: one process and one
prover client per GPU, static block-cyclic assignment, no data-path collective.
The only exchange is the one the north star names: an all-gather of every
proof's 32-byte main-trace commitment so each rank holds the ordered list that
would feed an aggregation tree (latency-bound: n * 32 bytes over xGMI).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .client import MACHINE_CHIPS, MACHINE_HEADER_WORDS, MACHINE_VERSION


def shard_indices(n_total: int, rank: int, world: int) -> List[int]:
    """Block-cyclic: proof i goes to rank i % world (costs are near-uniform within a
    config: ~171 k guest cycles per branch node)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_total, world))


def trace_root_of(proof_bytes: bytes) -> np.ndarray:
    """The 8-word main-trace commitment of a serialized proof (first body words)."""
    words = np.frombuffer(proof_bytes, dtype="<u4")
    if int(words[1]) == MACHINE_VERSION:  # machine proof: fixed header (chip heights ..), public values, then the main root
        off = MACHINE_HEADER_WORDS + (int(words[3 + MACHINE_CHIPS]) + 3) // 4
        return words[off:off + 8].copy()
    n_perms, pv_len = int(words[3]), int(words[5])
    off = 30 + (pv_len + 3) // 4 + 100 * n_perms  # fixed header, public values, public I/O list
    return words[off:off + 8].copy()


def gather_roots(local_roots: np.ndarray, n_total: int, rank: int, world: int, device=None) -> np.ndarray:
    """All-gathers per-rank roots ([n_local][8] u32, in shard order) into the global
    [n_total][8] array ordered by proof index.  Uses torch.distributed (RCCL on the
    GPUs, gloo in CPU tests); world == 1 needs no process group."""
    local_roots = np.ascontiguousarray(local_roots, dtype=np.uint32).reshape(-1, 8)
    mine = shard_indices(n_total, rank, world)
    if local_roots.shape[0] != len(mine):
        raise ValueError("local_roots does not match this rank's shard")
    if world == 1:
        return local_roots.copy()
    import torch
    import torch.distributed as dist

    per = (n_total + world - 1) // world  # ranks may differ by one proof: pad to equal length
    buf = torch.zeros((per, 8), dtype=torch.int32)
    buf[: len(mine)] = torch.from_numpy(local_roots.view(np.int32))
    if device is not None:
        buf = buf.to(device)
    out = torch.empty((world * per, 8), dtype=torch.int32, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    gathered = out.cpu().numpy().view(np.uint32).reshape(world, per, 8)
    result = np.zeros((n_total, 8), dtype=np.uint32)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        result[idx] = gathered[r, : len(idx)]
    return result


def prove_sharded(client, pk, stdins: Sequence, rank: int, world: int):
    """Proves this rank's shard of `stdins`; returns (indices, proofs, status)."""
    mine = shard_indices(len(stdins), rank, world)
    proofs, status = client.prove_batch(pk, [stdins[i] for i in mine]) if mine else ([], [])
    return mine, proofs, status


def tree_node_groups(n_leaves: int, arity: int) -> List[List[int]]:
    """The leaves of node k of one level of a recursion tree of that arity: [k * arity, (k + 1) * arity), the last node
    possibly shorter."""
    if arity < 1:
        raise ValueError("arity must be positive")
    return [list(range(a, min(a + arity, n_leaves))) for a in range(0, n_leaves, arity)]


def prove_tree_level(client, pk, leaf_vk, leaves: Sequence, node_stdins: Sequence, arity: int, rank: int, world: int):
    """One level of BASELINE config 5's recursion tree over the farm (stage 2a): node k is one
    more guest run (node_stdins[k]) whose proof also checks the query phases of its `arity` leaf proofs
    (client.add_verified_leaf, in leaf order).  Nodes are independent of one another and shard block-cyclically over the
    ranks like any other proofs; every rank holds all the leaves (they were all-gathered or are on shared storage - a node's
    host part verifies its leaves before anything is proven).  The stdins of this rank's nodes are CONSUMED: whatever leaf
    checks they carried are replaced by the node's own (so a retry of the level does not double them).  Returns (node
    indices of this rank, their proofs, status)."""
    groups = tree_node_groups(len(leaves), arity)
    if len(node_stdins) != len(groups):
        raise ValueError("one stdin per node")
    mine = shard_indices(len(groups), rank, world)
    stdins = []
    for k in mine:
        client.clear_verified_leaves(node_stdins[k])
        for i in groups[k]:
            client.add_verified_leaf(node_stdins[k], leaves[i], leaf_vk)
        stdins.append(node_stdins[k])
    proofs, status = client.prove_batch(pk, stdins) if mine else ([], [])
    return mine, proofs, status


def verify_tree_level(client, vk, leaf_vk, leaves: Sequence, nodes: Sequence, arity: int) -> None:
    """Every node proof of a level verifies, with the statement its own leaves give in leaf order (which verifies the leaves
    on the way: in stage 2a the statement is derived from them)."""
    groups = tree_node_groups(len(leaves), arity)
    if len(nodes) != len(groups):
        raise ValueError("one proof per node")
    for k, g in enumerate(groups):
        client.verify_with_leaves(nodes[k], vk, [leaves[i] for i in g], [leaf_vk] * len(g))
