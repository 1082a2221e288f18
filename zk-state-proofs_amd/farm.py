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

from typing import Callable, List, Optional, Sequence

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


def prove_tree_level(client, pk, leaf_vk, leaves: Sequence, node_stdins: Sequence, arity: int, rank: int, world: int,
                     statements: Optional[Sequence] = None, pipeline: bool = True, group: Optional[int] = None, checker=None,
                     statements_out: Optional[dict] = None):
    """One level of BASELINE config 5's recursion tree over the farm (stage 2b): node k is one
    more guest run (node_stdins[k]) whose proof also checks the query phases of its `arity` leaf proofs under the challenges
    their own transcripts yield (client.add_verified_leaf, in leaf order; a leaf that is itself a node comes with the
    statement its proof was made for: statements[i], client.add_verified_node).  Nodes are independent of one another and shard block-cyclically over the
    ranks like any other proofs; every rank holds all the leaves (they were all-gathered or are on shared storage - a node's
    host part verifies its leaves before the node is proven; with `pipeline` the checks of the next nodes run beside the proving
    of the ready ones: inside the level's one prove_batch call where the client can defer them (client.defer_verified_leaves),
    else on a helper thread - on `checker`, a client of the same parameters that needs no GPU, if one is given - feeding
    prove_batch calls of `group` nodes).  The stdins of this rank's nodes are CONSUMED: whatever leaf
    checks they carried are replaced by the node's own (so a retry of the level does not double them).  `statements_out`: a dict
    that receives, per node index of this rank, the statement its proof is made for (the public tuples its leaf checks gave -
    read from the stdin before it is proven, not derived a second time).  Returns (node indices of this rank, their proofs,
    status)."""
    groups = tree_node_groups(len(leaves), arity)
    if len(node_stdins) != len(groups):
        raise ValueError("one stdin per node")
    mine = shard_indices(len(groups), rank, world)

    chk = checker if checker is not None else client

    def check_leaves(k):
        # the host's part of node k: its leaves verified and logged (side by side, on the library's threads)
        chk.clear_verified_leaves(node_stdins[k])
        g = groups[k]
        if hasattr(chk, "add_verified_leaves"):
            chk.add_verified_leaves(node_stdins[k], [leaves[i] for i in g], [leaf_vk] * len(g),
                                    None if statements is None else [statements[i] for i in g])
        else:
            for i in g:
                if statements is not None and statements[i] is not None:
                    chk.add_verified_node(node_stdins[k], leaves[i], leaf_vk, statements[i])
                else:
                    chk.add_verified_leaf(node_stdins[k], leaves[i], leaf_vk)
        if statements_out is not None and hasattr(chk, "stdin_statement"):
            statements_out[k] = chk.stdin_statement(node_stdins[k])

    if pipeline and mine and hasattr(client, "defer_verified_leaves") and hasattr(client, "stdin_statement"):
        # the library's own pipeline: the checks are deferred to the ONE prove_batch call of the level, which makes them on its
        # tracing threads while the GPU proves the nodes that are ready (uploads and downloads overlapped as for any batch)
        for k in mine:
            client.clear_verified_leaves(node_stdins[k])
            g = groups[k]
            client.defer_verified_leaves(node_stdins[k], [leaves[i] for i in g], [leaf_vk] * len(g),
                                         None if statements is None else [statements[i] for i in g])
        proofs, status = client.prove_batch(pk, [node_stdins[k] for k in mine])
        if statements_out is not None:
            for k, st in zip(mine, status):
                if st == 0:
                    statements_out[k] = client.stdin_statement(node_stdins[k])
        return mine, proofs, status
    if len(mine) <= 2 or not pipeline:
        for k in mine:
            check_leaves(k)
        proofs, status = client.prove_batch(pk, [node_stdins[k] for k in mine]) if mine else ([], [])
        return mine, proofs, status
    # Many nodes: the host checks the leaves of the next nodes WHILE the GPU proves the ones that are ready (the library's
    # calls release the interpreter lock) - a level then takes the longer of the two parts, not their sum: a node of four
    # full-size leaves is 20 ms of host work and 32 ms of proving alone, less in a batch (bench.py leaf_check).  The GPU
    # proves `group` nodes per call (a third of the level, at most eight: a node's Poseidon2 chip is 2^19 rows, and nodes
    # proven one at a time as they become ready keep the GPU at its batch-of-one rate, which is slower than the host).
    if group is None:
        group = min(8, max(1, len(mine) // 3))
    import queue
    import threading
    ready: "queue.Queue" = queue.Queue()

    stop = threading.Event()

    def producer():
        try:
            for k in mine:
                if stop.is_set():
                    break
                check_leaves(k)
                ready.put(k)
            ready.put(None)
        except BaseException as e:  # handed to the consuming thread
            ready.put(e)

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    done, proofs_of, status_of = False, {}, {}
    try:
        while not done:
            batch = []
            item = ready.get()
            while True:
                if item is None:
                    done = True
                    break
                if isinstance(item, BaseException):
                    raise item
                batch.append(item)
                if len(batch) >= group:
                    break
                item = ready.get()
            if batch:
                pr, st = client.prove_batch(pk, [node_stdins[k] for k in batch])
                for k, p_, s_ in zip(batch, pr, st):
                    proofs_of[k], status_of[k] = p_, s_
    finally:
        stop.set()
        th.join()
    return mine, [proofs_of[k] for k in mine], [status_of[k] for k in mine]


def verify_tree_level(client, vk, leaf_vk, leaves: Sequence, nodes: Sequence, arity: int) -> None:
    """Every node proof of a level verifies, with the statement its own leaves give in leaf order (leaves may be stubs: the
    statement needs everything of a leaf but its query phase, which is what the node proves)."""
    groups = tree_node_groups(len(leaves), arity)
    if len(nodes) != len(groups):
        raise ValueError("one proof per node")
    for k, g in enumerate(groups):
        client.verify_with_leaves(nodes[k], vk, [leaves[i] for i in g], [leaf_vk] * len(g))


def _gather_objects(local, n_total: int, rank: int, world: int):
    """All-gathers a rank's block-cyclic shard of python objects into the global list ordered by index (the recursion tree's
    inputs: every rank holds every proof of a level before the next one is proven)."""
    if world == 1:
        return list(local)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, list(local))
    out = [None] * n_total
    for r in range(world):
        for j, i in enumerate(shard_indices(n_total, r, world)):
            out[i] = parts[r][j]
    return out


def prove_tree(client, host, pk, vk, leaves: Sequence, make_stdin: Callable, arity: int, rank: int = 0, world: int = 1):
    """BASELINE config 5's recursion tree, every level of it (stage 2b): the leaf proofs, then nodes
    of `arity` children level by level until one root is left.  A node is one more guest run (make_stdin(level, k)) whose proof
    checks the query phases of its children - leaves, or nodes with the statements their own proofs were made for.  The nodes
    of a level shard block-cyclically over the ranks; between levels the ranks all-gather the level's proofs (serialized) and
    statements - the farm's one exchange.  `host`: a client that derives statements (no GPU needed).  Returns
    (levels, statements): levels[0] = the leaves, levels[-1] = [root]; statements[d][k] = the public tuples proof k of level d
    was made for (None for a plain leaf)."""
    from .client import SP1ProofWithPublicValues
    levels, statements = [list(leaves)], [[None] * len(leaves)]
    depth = 0
    while len(levels[-1]) > 1:
        depth += 1
        below, st_below = levels[-1], statements[-1]
        groups = tree_node_groups(len(below), arity)
        node_stdins = [make_stdin(depth, k) for k in range(len(groups))]
        st_of = {}
        mine, proofs, status = prove_tree_level(client, pk, vk, below, node_stdins, arity, rank, world, st_below, checker=host,
                                                statements_out=st_of)
        if status != [0] * len(mine):
            raise RuntimeError(f"level {depth}: a node of rank {rank} failed: {client.last_error()}")
        local = []
        for k, p in zip(mine, proofs):
            st = st_of.get(k)
            if st is None:  # (a checking client without stdin_statement: derived from the children's stubs)
                stubs = [below[i].stub() for i in groups[k]]
                st = np.concatenate([host.leaf_public_at(stubs[j], vk, j, st_below[i]) for j, i in enumerate(groups[k])])
            local.append((p if world == 1 else p.to_bytes(), st))
        if world == 1:  # (nothing to exchange: the proof objects as they are)
            levels.append([p for p, _ in local])
            statements.append([st for _, st in local])
            continue
        gathered = _gather_objects(local, len(groups), rank, world)
        levels.append([SP1ProofWithPublicValues.from_bytes(b) for b, _ in gathered])
        statements.append([st for _, st in gathered])
    return levels, statements


def tree_of_stubs(levels: Sequence[Sequence], arity: int):
    """The `children` argument of client.verify_tree for the root of prove_tree's levels: stubs of everything below the root."""
    def node(d, k):
        if d == 0:
            return (levels[0][k].stub(), [])
        groups = tree_node_groups(len(levels[d - 1]), arity)
        return (levels[d][k].stub(), [node(d - 1, i) for i in groups[k]])
    top = len(levels) - 1
    groups = tree_node_groups(len(levels[top - 1]), arity)
    return [node(top - 1, i) for i in groups[0]]
