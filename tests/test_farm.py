"""The N > 1 path on CPU: world_size-2 gloo ranks shard a proof list block-
cyclically and all-gather the 32-byte trace commitments into proof order."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import torch.distributed as dist
    import oracle
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = farm.shard_indices(n_total, rank, world)
    roots = []
    for i in mine:  # each rank proves only its own shard (CPU oracle stands in for the GPU here)
        st = np.random.default_rng(i).integers(0, 2**64, (1, 25), dtype=np.uint64)
        roots.append(farm.trace_root_of(oracle.prove(st, 5, num_queries=1, pow_bits=1)))
    allr = farm.gather_roots(np.array(roots, np.uint32).reshape(-1, 8), n_total, rank, world)
    np.save(os.path.join(out_dir, f"roots_{rank}.npy"), allr)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices():
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    for n, w in ((0, 1), (1, 4), (7, 2), (300, 8), (1024, 8)):
        seen = sorted(i for r in range(w) for i in farm.shard_indices(n, r, w))
        assert seen == list(range(n))
        sizes = [len(farm.shard_indices(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        farm.shard_indices(4, 2, 2)


def test_gather_roots_world2_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    world, n_total = 2, 5  # uneven shards: 3 + 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    expect = []
    for i in range(n_total):
        st = np.random.default_rng(i).integers(0, 2**64, (1, 25), dtype=np.uint64)
        expect.append(farm.trace_root_of(oracle.prove(st, 5, num_queries=1, pow_bits=1)))
    expect = np.array(expect, np.uint32)
    for r in range(world):
        got = np.load(tmp_path / f"roots_{r}.npy")
        assert np.array_equal(got, expect)
    assert len({tuple(x) for x in expect}) == n_total


def test_world1_needs_no_process_group():
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    r = np.arange(24, dtype=np.uint32).reshape(3, 8)
    assert np.array_equal(farm.gather_roots(r, 3, 0, 1), r)


class _OracleBackedClient:
    """Stands in for a GPU client in CPU tests: the product's host side traces the guest (device -1), the CPU oracle proves.
    Same call shape as ProverClient.prove_batch."""

    def __init__(self, zk, oracle, nq, pw):
        self.zk, self.oracle, self.nq, self.pw = zk, oracle, nq, pw
        self.host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)

    def setup(self, elf):
        return self.host.setup(elf)

    def add_verified_leaf(self, stdin, leaf, leaf_vk):
        return self.host.add_verified_leaf(stdin, leaf, leaf_vk)

    def add_verified_node(self, stdin, node, node_vk, statement):
        return self.host.add_verified_node(stdin, node, node_vk, statement)

    def clear_verified_leaves(self, stdin):
        return self.host.clear_verified_leaves(stdin)

    def last_error(self):
        return self.host.last_error()

    def verify_with_leaves(self, proof, vk, leaves, leaf_vks):
        return self.host.verify_with_leaves(proof, vk, leaves, leaf_vks)

    def prove_batch(self, pk, stdins):
        proofs = []
        for s in stdins:
            t = self.host.machine_trace(pk, s)
            proofs.append(self.zk.SP1ProofWithPublicValues.from_bytes(self.oracle.machine_prove(t, num_queries=self.nq, pow_bits=self.pw)))
        return proofs, [0] * len(stdins)


def _machine_worker(rank, world, port, n_total, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import torch.distributed as dist
    import oracle
    zk = importlib.import_module("zk-state-proofs_amd")
    fx = importlib.import_module("zk-state-proofs_amd.fixtures")
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    client = _OracleBackedClient(zk, oracle, 2, 2)
    pk, vk = client.setup(zk.merkle_elf())
    stdins = []
    for i in range(n_total):  # every rank builds the whole workload's inputs and proves its own shard of it
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(2, seed=70 + i).to_borsh())
        stdins.append(s)
    mine, proofs, status = farm.prove_sharded(client, pk, stdins, rank, world)
    assert status == [0] * len(mine) and mine == farm.shard_indices(n_total, rank, world)
    for p in proofs:
        client.host.verify(p, vk)
    local = np.array([farm.trace_root_of(p.to_bytes()) for p in proofs], np.uint32).reshape(-1, 8)
    allr = farm.gather_roots(local, n_total, rank, world)
    np.save(os.path.join(out_dir, f"mroots_{rank}.npy"), allr)
    # what the gathered list is for: the aggregation payload of one more proof (made by rank 0 here)
    if rank == 0:
        leaves = np.vstack([allr, allr[: (1 << (n_total - 1).bit_length()) - n_total]])  # padded to a power of two
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=99).to_borsh())
        s.set_aggregation(leaves)
        t = client.host.machine_trace(pk, s)
        t["agg_leaves"] = leaves
        agg = zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(t, num_queries=2, pow_bits=2))
        client.host.verify_aggregate(agg, vk, leaves)
        np.save(os.path.join(out_dir, "agg_root.npy"), np.array(agg.aggregation[1], np.uint32))
    dist.barrier()
    dist.destroy_process_group()


def test_prove_sharded_machine_proofs_world2_gloo(tmp_path, zk, fx, oracle):
    """BASELINE configs 4 / 5 in miniature on the CPU: two gloo ranks shard a list of guest runs block-cyclically
    (farm.prove_sharded; the oracle stands in for the GPU), every rank verifies its machine proofs, the 32-byte main-trace
    commitments are all-gathered into proof order (farm.gather_roots: the path's one exchange), and rank 0 proves the
    gathered list's Poseidon2 Merkle root as the aggregation payload of one more run."""
    import torch.multiprocessing as mp
    world, n_total = 2, 3  # uneven shards: 2 + 1
    port = _free_port()
    mp.spawn(_machine_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"mroots_{r}.npy") for r in range(world)]
    assert np.array_equal(got[0], got[1]) and got[0].shape == (n_total, 8)
    assert len({tuple(x) for x in got[0]}) == n_total
    # the same commitments, computed serially
    client = _OracleBackedClient(zk, oracle, 2, 2)
    pk, _ = client.setup(zk.merkle_elf())
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    for i in (0, n_total - 1):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(2, seed=70 + i).to_borsh())
        p, _st = client.prove_batch(pk, [s])
        assert np.array_equal(farm.trace_root_of(p[0].to_bytes()), got[0][i])
    leaves = np.vstack([got[0], got[0][:1]])
    assert [int(x) for x in np.load(tmp_path / "agg_root.npy")] == oracle.machine_agg_public(leaves)[0]


def test_tree_level_of_leaf_checks(zk, fx, oracle):
    """One level of config 5's recursion tree through the farm's functions (CPU: the oracle proves): five leaf proofs, nodes of
    arity two (the last node has one leaf), sharded over two ranks in turn; every node verifies with exactly its own leaves."""
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    assert farm.tree_node_groups(5, 2) == [[0, 1], [2, 3], [4]] and farm.tree_node_groups(4, 4) == [[0, 1, 2, 3]]
    client = _OracleBackedClient(zk, oracle, 3, 3)
    pk, vk = client.setup(zk.merkle_elf())
    leaf_stdins = []
    for i in range(5):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(2, seed=300 + i).to_borsh())  # (depth 2: the seed matters)
        leaf_stdins.append(s)
    leaves, status = client.prove_batch(pk, leaf_stdins)
    assert status == [0] * 5 and len({p.to_bytes() for p in leaves}) == 5
    nodes = [None] * 3
    for rank in range(2):
        node_stdins = []
        for k in range(3):
            s = zk.SP1Stdin()
            s.write(fx.acct_fixture(1, seed=400 + k).to_borsh())
            node_stdins.append(s)
        mine, proofs, st = farm.prove_tree_level(client, pk, vk, leaves, node_stdins, 2, rank, 2)
        assert st == [0] * len(mine) and mine == farm.shard_indices(3, rank, 2)
        for k, p in zip(mine, proofs):
            nodes[k] = p
    farm.verify_tree_level(client, vk, vk, leaves, nodes, 2)
    # a node does not verify with another node's leaves, and the level does not verify with the leaves in another order
    with pytest.raises(zk.VerificationError):
        client.verify_with_leaves(nodes[0], vk, [leaves[2], leaves[3]], [vk, vk])
    with pytest.raises(zk.VerificationError):
        farm.verify_tree_level(client, vk, vk, leaves[::-1], nodes, 2)


def _tree_worker(rank, world, port, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    import torch.distributed as dist
    import oracle
    zk = importlib.import_module("zk-state-proofs_amd")
    fx = importlib.import_module("zk-state-proofs_amd.fixtures")
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "3"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    client = _OracleBackedClient(zk, oracle, 3, 3)
    pk, vk = client.setup(zk.merkle_elf())
    # the leaves of the tree: proven by the farm (block-cyclic), all-gathered like any level of the tree
    stdins = []
    for i in range(4):
        s = zk.SP1Stdin()
        s.write((fx.acct_fixture(1, seed=500) if i % 2 == 0 else fx.slot_fixture(i)).to_borsh())
        stdins.append(s)
    mine, proofs, status = farm.prove_sharded(client, pk, stdins, rank, world)
    assert status == [0] * len(mine)
    leaves = [zk.SP1ProofWithPublicValues.from_bytes(b) for b in farm._gather_objects([p.to_bytes() for p in proofs], 4, rank, world)]

    def make_stdin(depth, k):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=600 + 10 * depth + k).to_borsh())
        return s

    levels, statements = farm.prove_tree(client, client.host, pk, vk, leaves, make_stdin, 2, rank, world)
    assert [len(l) for l in levels] == [4, 2, 1] and statements[0] == [None] * 4
    root_proof = levels[-1][0]
    client.host.verify_tree(root_proof, vk, farm.tree_of_stubs(levels, 2))
    with open(os.path.join(out_dir, f"root_{rank}.bin"), "wb") as f:
        f.write(root_proof.to_bytes())
    if rank == 0:
        stubs = farm.tree_of_stubs(levels, 2)
        read = sum(len(p.to_bytes()) for p, kids in stubs) + sum(len(c.to_bytes()) for _, kids in stubs for c, _ in kids)
        full = sum(len(p.to_bytes()) for lv in levels[:-1] for p in lv)
        np.save(os.path.join(out_dir, "sizes.npy"), np.array([read, full, len(root_proof.to_bytes())]))
        # the root does not verify over another tree: two leaves swapped
        bad = [(stubs[1][0], stubs[0][1]), (stubs[0][0], stubs[1][1])]
        try:
            client.host.verify_tree(root_proof, vk, bad)
            raise AssertionError("a root verified over another tree")
        except zk.VerificationError:
            pass
    dist.barrier()
    dist.destroy_process_group()


def test_recursion_tree_world2_gloo(tmp_path, zk):
    """BASELINE config 5 in miniature on the CPU (the oracle stands in for the GPU): two gloo ranks prove four leaves, then the
    two nodes of arity two (one per rank), all-gather them with their statements, and the root over the two NODES (a node is a
    valid leaf: stage 2b); every rank holds the same root, which verifies from the stubs of the six proofs below it."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_tree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (open(tmp_path / f"root_{r}.bin", "rb").read() for r in range(2))
    assert r0 == r1
    read, full, root_len = (int(x) for x in np.load(tmp_path / "sizes.npy"))
    assert read < full  # (3 queries here: at the full 100 the stubs are a twentieth of the proofs)


def test_tree_level_pipeline_orders_groups_and_fails_cleanly():
    """farm.prove_tree_level with many nodes: the leaf checks of the next nodes run on a helper thread while the ready nodes
    are proven, in calls of at most `group` nodes; proofs come back in node order whatever the grouping was; a failing check
    surfaces in the caller and stops the helper; the unpipelined path gives the same result.  (A stand-in client: no
    cryptography here - the real thing is tests/test_gpu_machine.py::test_two_level_tree_matches_oracle and bench.py.)"""
    import threading
    import time
    farm = importlib.import_module("zk-state-proofs_amd.farm")

    class Stub:
        def __init__(self, fail_at=None, slow=0.0):
            self.checked, self.calls, self.fail_at, self.slow, self.threads = [], [], fail_at, slow, set()

        def clear_verified_leaves(self, stdin):
            stdin["leaves"] = None

        def add_verified_leaves(self, stdin, leaves, vks, statements):
            self.threads.add(threading.get_ident())
            if self.fail_at is not None and stdin["k"] == self.fail_at:
                raise RuntimeError("leaf does not verify")
            time.sleep(self.slow)
            stdin["leaves"] = list(leaves)
            self.checked.append(stdin["k"])

        def prove_batch(self, pk, stdins):
            assert all(s["leaves"] is not None for s in stdins)  # never proven before its leaves were checked
            self.calls.append([s["k"] for s in stdins])
            time.sleep(0.01)
            return [("proof", s["k"], tuple(s["leaves"])) for s in stdins], [0] * len(stdins)

    leaves = list(range(23))
    for pipeline in (True, False):
        c = Stub(slow=0.002)
        stdins = [{"k": k} for k in range(6)]
        mine, proofs, st = farm.prove_tree_level(c, None, "vk", leaves, stdins, 4, 0, 1, None, pipeline=pipeline, group=2)
        assert mine == list(range(6)) and st == [0] * 6
        assert proofs == [("proof", k, tuple(range(4 * k, min(4 * k + 4, 23)))) for k in range(6)]
        assert sorted(k for call in c.calls for k in call) == list(range(6)) and c.checked == list(range(6))
        if pipeline:
            assert all(len(call) <= 2 for call in c.calls) and threading.get_ident() not in c.threads
        else:
            assert c.calls == [list(range(6))]
    # sharded: rank 1 of 2 gets the odd nodes
    c = Stub()
    mine, proofs, st = farm.prove_tree_level(c, None, "vk", leaves, [{"k": k} for k in range(6)], 4, 1, 2, None, group=16)
    assert mine == [1, 3, 5] and [p[1] for p in proofs] == [1, 3, 5]
    # a separate checking client takes the checks, the proving client only proves
    prover, checker = Stub(), Stub()
    farm.prove_tree_level(prover, None, "vk", leaves, [{"k": k} for k in range(6)], 4, 0, 1, None, checker=checker)
    assert checker.checked == list(range(6)) and not prover.checked and prover.calls
    # a leaf that does not verify: the caller sees it, nothing hangs
    c = Stub(fail_at=3)
    with pytest.raises(RuntimeError, match="leaf does not verify"):
        farm.prove_tree_level(c, None, "vk", leaves, [{"k": k} for k in range(6)], 4, 0, 1, None, group=1)
    assert 4 not in c.checked and 5 not in c.checked
