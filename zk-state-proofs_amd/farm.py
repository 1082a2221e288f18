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
