"""Machine proof on the GPU: the HIP prover reproduces the CPU oracle's proof bytes exactly and the
host verifier accepts what it produced."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def first_difference(body, expected):
    bad = np.nonzero(body != expected)[0]
    return None if bad.size == 0 else (int(bad[0]), int(bad.size))


def shape_of(zk, raw):
    """The chip log-heights a proof was made with (its header): a batch shares one shape."""
    return [int.from_bytes(raw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]


@pytest.mark.parametrize("depth,nq,pow_bits", [(1, 8, 6)])
def test_machine_proof_matches_oracle(zk, fx, oracle, depth, nq, pow_bits):
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pow_bits, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    handles, traces = [], []
    for seed in (1, 2):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(depth, seed=seed).to_borsh())
        handles.append(client.machine_trace_handle(pk, s))
        traces.append(client.machine_trace(pk, s))
    assert handles[0].heights() == oracle.machine_heights(traces[0]) and handles[1].heights() == oracle.machine_heights(traces[1])
    # a batch is proven with ONE shape: the chip heights that cover the largest counts among its runs
    shape = zk.machine_cover_heights(handles)
    assert all(a >= b for hd in handles for a, b in zip(shape, hd.heights()))
    bodies = client.machine_prove_resident(pk, handles)
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pow_bits)
    for i in range(2):
        exp = oracle.machine_prove(dict(traces[i], shape=shape), num_queries=nq, pow_bits=pow_bits)
        hw = zk.MACHINE_HEADER_WORDS + (len(traces[i]["public_values"]) + 3) // 4
        e = np.frombuffer(exp, dtype=np.uint32)[hw:]
        assert e.shape == bodies[i].shape
        assert first_difference(bodies[i], e) is None, first_difference(bodies[i], e)
        proof = handles[i].proof_from_body(pk, bodies[i], shape)
        assert proof.to_bytes() == exp
        host.verify(proof, vk)


def test_reference_flow_acct_d8_full_size(zk, fx, oracle):
    """BASELINE config 2 through the reference's own call sequence (prover/src/bin/main.rs:59-87) with the
    default client: setup -> prove(..).run() -> public_values -> verify, at full parameters (100 queries,
    16 proof-of-work bits).  The proof is the machine proof of the 391 400-cycle precompile-shape run
    (six CPU instances of 2^16 rows x 52, ALU, bitwise and sub-word chips beside them): byte-identical to the oracle's, accepted by a host-only verifier, every
    tampered region rejected, and another public value cannot be attached."""
    client = zk.ProverClient(device=0)
    pk, vk = client.setup(zk.merkle_elf())
    inp = fx.acct_fixture(8)
    stdin = zk.SP1Stdin()
    stdin.write(inp.to_borsh())
    trace = client.machine_trace(pk, stdin)
    proof = client.prove(pk, stdin).run()
    assert proof.public_values == fx.ACCOUNT_VALUE
    client.verify(proof, vk)
    raw = proof.to_bytes()
    heights = [int.from_bytes(raw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
    # 391 400 cycles: six CPU instances of 2^16 rows, the last two all padding at the minimum height
    cpu = [heights[zk.MACHINE_CHIP_NAMES.index(n)] for n in ("cpu", "cpu2", "cpu3", "cpu4", "cpu5", "cpu6", "cpu7", "cpu8")]
    assert int.from_bytes(raw[4:8], "little") == zk.MACHINE_VERSION and cpu == [16] * 6 + [5] * 2
    assert raw == oracle.machine_prove(trace)
    host = zk.ProverClient(device=-1)
    host.verify(zk.SP1ProofWithPublicValues.from_bytes(raw), vk)
    rng = np.random.default_rng(3)
    hb = zk.MACHINE_HEADER_WORDS * 4  # exit code, public-values length, their digest, the values, the body
    for pos in [(2 + zk.MACHINE_CHIPS) * 4, (3 + zk.MACHINE_CHIPS) * 4, (4 + zk.MACHINE_CHIPS) * 4 + 1, hb + 5, hb + 72 + 3] + [int(x) for x in rng.integers(hb + 72, len(raw), 12)]:
        bad = bytearray(raw)
        bad[pos] ^= 1
        with pytest.raises(zk.ZkspError):
            host.verify(zk.SP1ProofWithPublicValues.from_bytes(bytes(bad)), vk)


def test_as_committed_instruction_stream(zk, fx):
    """The guest exactly as committed (software keccak, no precompile): 359 949 cycles for the depth-2
    transaction proof, all of them in the CPU chip (2^19 rows), keccak chips empty."""
    client = zk.ProverClient(device=0, keccak_mode=zk.KECCAK_OBSERVE, num_queries=20, pow_bits=8)
    pk, vk = client.setup(zk.merkle_elf())
    inp = fx.tx_fixture()
    stdin = zk.SP1Stdin()
    stdin.write(inp.to_borsh())
    proof = client.prove(pk, stdin).run()
    from oracle import verify_merkle_proof
    assert proof.public_values == verify_merkle_proof(inp.root_hash, inp.proof, inp.key)
    zk.ProverClient(device=-1, keccak_mode=zk.KECCAK_OBSERVE, num_queries=20, pow_bits=8).verify(proof, vk)
    # the precompile-shape key is another program: it must not accept this proof
    other = zk.ProverClient(device=-1, num_queries=20, pow_bits=8)
    _, vk2 = other.setup(zk.merkle_elf())
    with pytest.raises(zk.ZkspError):
        other.verify(proof, vk2)


def test_as_committed_acct_d8_matches_oracle(zk, fx, oracle):
    """BASELINE config 2 on the binary exactly as the reference ships it (circuits/elf/riscv32im-succinct-zkvm-elf, software
    keccak through the CPU, ALU and bitwise chips): 1 406 960 cycles in six CPU instances of 2^18 rows.  The device's proof
    body equals the oracle's byte for byte, the wrapped proof equals the oracle's proof and a host-only verifier accepts it."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, keccak_mode=zk.KECCAK_OBSERVE, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=2).to_borsh())
    handle = client.machine_trace_handle(pk, s)
    trace = client.machine_trace(pk, s)
    assert 1_400_000 < trace["cycles"].shape[0] < 1_420_000 and trace["keccak"].shape[0] == 0
    shape = zk.machine_cover_heights([handle])
    names = zk.MACHINE_CHIP_NAMES
    assert [shape[names.index(n)] for n in ("cpu", "cpu2", "cpu3", "cpu4", "cpu5", "cpu6", "cpu7", "cpu8")] == [18] * 6 + [5] * 2
    assert shape[names.index("keccak")] == 5  # no precompile call: the keccak chips are all padding
    body = client.machine_prove_resident(pk, [handle])[0]
    exp = oracle.machine_prove(dict(trace, shape=shape), num_queries=nq, pow_bits=pw)
    hw = zk.MACHINE_HEADER_WORDS + (len(trace["public_values"]) + 3) // 4
    e = np.frombuffer(exp, dtype=np.uint32)[hw:]
    assert e.shape == body.shape
    assert first_difference(body, e) is None, first_difference(body, e)
    proof = handle.proof_from_body(pk, body, shape)
    assert proof.to_bytes() == exp
    assert proof.public_values == fx.ACCOUNT_VALUE
    zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw, keccak_mode=zk.KECCAK_OBSERVE).verify(proof, vk)


def test_batch_of_machine_proofs(zk, fx, oracle):
    """Several guest runs proven in lockstep (prove_batch): runs of different size classes fall into separate groups,
    runs of one class share the shape that covers the largest of them, every proof verifies, one proof of the largest
    group equals the oracle's bytes for that shape."""
    nq, pw = 6, 5
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=4)
    pk, vk = client.setup(zk.merkle_elf())
    inputs = [fx.acct_fixture(1, seed=s) for s in range(1, 7)] + [fx.tx_fixture(), fx.slot_fixture(0)]
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    s3 = zk.SP1Stdin()
    s3.write(inputs[3].to_borsh())
    trace3 = client.machine_trace(pk, s3)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * len(inputs)
    from oracle import verify_merkle_proof
    for m, p in zip(inputs, proofs):
        assert p.public_values == verify_merkle_proof(m.root_hash, m.proof, m.key)
        client.verify(p, vk)
    raw3 = proofs[3].to_bytes()
    assert raw3 == oracle.machine_prove(dict(trace3, shape=shape_of(zk, raw3)), num_queries=nq, pow_bits=pw)
    assert len({tuple(shape_of(zk, p.to_bytes())) for p in proofs}) >= 2


def test_chunked_batch_equals_single_proofs(zk, fx, oracle):
    """prove_batch with more inputs than max_batch proves chunk k while chunk k + 1 uploads into the spare record
    set and chunk k - 1 is wrapped on the host: every proof must be byte-identical to the one a single prove()
    call makes when the shapes agree (the prover is deterministic), and to the oracle's for its shape otherwise,
    including the short last chunk."""
    nq, pw = 6, 5
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    inputs = [fx.acct_fixture(1, seed=s) for s in range(11, 16)]  # five runs of equal heights: chunks of 2, 2, 1
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * len(inputs)
    single = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=1)
    pk1, vk1 = single.setup(zk.merkle_elf())
    for m, p in zip(inputs, proofs):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        trace = single.machine_trace(pk1, s)
        q = single.prove(pk1, s).run()
        raw = p.to_bytes()
        if shape_of(zk, raw) == shape_of(zk, q.to_bytes()):
            assert q.to_bytes() == raw
        else:
            assert raw == oracle.machine_prove(dict(trace, shape=shape_of(zk, raw)), num_queries=nq, pow_bits=pw)
        client.verify(p, vk)


def test_device_proof_matches_the_frozen_pins(zk, fx):
    """The device prover against tests/golden/machine_kat.json directly (no oracle in the loop): SHA-256 of the proof
    bytes of two small fixtures at 8 queries / 6 proof-of-work bits."""
    import hashlib
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "machine_kat.json")) as f:
        kat = json.load(f)
    client = zk.ProverClient(device=0, num_queries=kat["num_queries"], pow_bits=kat["pow_bits"], max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    assert [int(x) for x in vk.machine[1]] == kat["vk_digest"]
    for name, m in (("acct-d1", fx.acct_fixture(1)), ("tx-d2", fx.tx_fixture())):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        proof = client.prove(pk, s).run().to_bytes()
        case = kat["cases"][name]
        assert len(proof) == case["proof_bytes"]
        assert hashlib.sha256(proof).hexdigest() == case["proof_sha256"], name


def test_prove_batch_of_128_acct_d8(zk, fx, oracle):
    """BASELINE config 5's substitute at test size (the reference's recursion circuit is a todo!(),
    circuits/sp1-merkle-proof-recursive/src/main.rs:3-5, so the measurable part is many acct-d8 leaf proofs whose 32-byte
    commitments a recursion tree would take in): 128 depth-8 account proofs in ONE drop-in prove_batch call (host buffers
    in, proof objects out, chunks of 64 proven while the next chunk's records upload), every proof's public values
    checked, a spread of them verified by a host-only client, one compared byte for byte with the oracle for the shape
    its chunk was proven with, and the commitments gathered in proof order as the farm would hand them on."""
    farm = __import__("importlib").import_module("zk-state-proofs_amd.farm")
    client = zk.ProverClient(device=0, max_batch=64)
    pk, vk = client.setup(zk.merkle_elf())
    n = 128
    inputs = [fx.acct_fixture(8, seed=7000 + i) for i in range(n)]
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    k = 77
    sk = zk.SP1Stdin()
    sk.write(inputs[k].to_borsh())
    trace_k = client.machine_trace(pk, sk)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * n
    assert all(p.public_values == fx.ACCOUNT_VALUE for p in proofs)
    host = zk.ProverClient(device=-1)
    for i in (0, 31, 63, 64, 77, 100, 127):
        host.verify(proofs[i], vk)
    raw = proofs[k].to_bytes()
    assert raw == oracle.machine_prove(dict(trace_k, shape=shape_of(zk, raw)))
    roots = np.array([farm.trace_root_of(p.to_bytes()) for p in proofs], np.uint32)
    assert roots.shape == (n, 8) and len({tuple(r) for r in roots}) == n
    assert np.array_equal(farm.gather_roots(roots, n, 0, 1), roots)


def test_aggregation_payload_matches_oracle(zk, fx, oracle):
    """Row f4, stage 1 on the device: proofs with an aggregation payload (the Poseidon2 chip: p2_trace_kernel,
    machine_quotient_kernel<poseidon2>, the DIGEST bus) byte-identical to the oracle's, in one batch with a proof that has
    none (one shape covers both), verified with and only with their leaves."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=3)
    pk, vk = client.setup(zk.merkle_elf())
    rng = np.random.default_rng(21)
    payloads = [rng.integers(0, 2013265921, (64, 8), dtype=np.uint32), None, rng.integers(0, 2013265921, (2, 8), dtype=np.uint32)]
    handles, traces = [], []
    for i, lv in enumerate(payloads):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=30 + i).to_borsh())
        if lv is not None:
            s.set_aggregation(lv)
        handles.append(client.machine_trace_handle(pk, s))
        t = client.machine_trace(pk, s)
        if lv is not None:
            t["agg_leaves"] = lv
        traces.append(t)
    shape = zk.machine_cover_heights(handles)
    assert shape[zk.MACHINE_CHIP_NAMES.index("poseidon2")] == 6  # 63 inner nodes of the 64-leaf tree
    bodies = client.machine_prove_resident(pk, handles)
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    for i, lv in enumerate(payloads):
        proof = handles[i].proof_from_body(pk, bodies[i], shape)
        assert proof.to_bytes() == oracle.machine_prove(dict(traces[i], shape=shape), num_queries=nq, pow_bits=pw), i
        if lv is None:
            host.verify(proof, vk)
        else:
            assert proof.aggregation == (len(lv), oracle.machine_agg_public(lv)[0])
            host.verify_aggregate(proof, vk, lv)
            with pytest.raises(zk.VerificationError):
                host.verify(proof, vk)


def test_merkle_path_payload_matches_oracle(zk, fx, oracle):
    """The same chip with digests supplied at heap keys: a Merkle path (leaf 37 of a depth-7 tree: seven rows) proven on
    the device next to a full-tree payload, both byte-identical to the oracle's and verified with their own inputs."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    rng = np.random.default_rng(5)
    leaves = rng.integers(0, 2013265921, (128, 8), dtype=np.uint32)
    heap = {128 + i: leaves[i] for i in range(128)}
    for k in range(127, 0, -1):
        heap[k] = np.array(oracle.compress(heap[2 * k], heap[2 * k + 1]), np.uint32)
    index, depth = 37, 7
    k0 = (1 << depth) + index
    siblings = [heap[(k0 >> j) ^ 1] for j in range(depth)]
    keys, digests = zk.merkle_path_nodes(index, heap[k0], siblings)
    handles, traces = [], []
    for i in range(2):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=40 + i).to_borsh())
        if i == 0:
            s.set_merkle_path(index, heap[k0], siblings)
        else:
            s.set_aggregation(leaves[:8])
        handles.append(client.machine_trace_handle(pk, s))
        t = client.machine_trace(pk, s)
        traces.append(dict(t, agg_leaves=digests, agg_keys=keys) if i == 0 else dict(t, agg_leaves=leaves[:8]))
    shape = zk.machine_cover_heights(handles)
    bodies = client.machine_prove_resident(pk, handles)
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    proofs = [handles[i].proof_from_body(pk, bodies[i], shape) for i in range(2)]
    for i in range(2):
        assert proofs[i].to_bytes() == oracle.machine_prove(dict(traces[i], shape=shape), num_queries=nq, pow_bits=pw), i
    assert proofs[0].aggregation == (depth + 1, [int(x) for x in heap[1]])
    host.verify_merkle_path(proofs[0], vk, index, heap[k0], siblings)
    with pytest.raises(zk.VerificationError):
        host.verify_merkle_path(proofs[0], vk, index + 1, heap[k0], siblings)
    host.verify_aggregate(proofs[1], vk, leaves[:8])



def test_aggregate_1024_commitments(zk, fx, oracle):
    """BASELINE config 5's shape end to end through the drop-in calls: prove leaf proofs, gather their 32-byte main-trace
    commitments in proof order (what the farm all-gathers), and prove ONE more run whose aggregation payload is the whole
    list - its root is then a proven Poseidon2 Merkle root of the 1024 commitments (here: 32 real commitments repeated,
    to keep the test short; bench.py does it with 1024 proofs)."""
    farm = __import__("importlib").import_module("zk-state-proofs_amd.farm")
    client = zk.ProverClient(device=0, max_batch=32)
    pk, vk = client.setup(zk.merkle_elf())
    stdins = []
    for i in range(32):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=900 + i).to_borsh())
        stdins.append(s)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * 32
    roots = np.array([farm.trace_root_of(p.to_bytes()) for p in proofs], np.uint32)
    leaves = np.tile(roots, (32, 1))
    assert leaves.shape == (1024, 8)
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1, seed=999).to_borsh())
    s.set_aggregation(leaves)
    agg = client.prove(pk, s).run()
    n, root = agg.aggregation
    assert n == 1024 and root == oracle.machine_agg_public(leaves)[0]
    zk.ProverClient(device=-1).verify_aggregate(agg, vk, leaves)


def test_leaf_check_matches_oracle(zk, fx, oracle):
    """Row f4, stage 2b on the device: a proof that also establishes the query phase of ANOTHER proof under the challenges that
    proof's own transcript yields - the leaf's openings as rows of the Poseidon2 chip (sponges with their Horner sums, path
    steps, injections: p2_trace_kernel, machine_quotient_kernel<poseidon2>, the DIGEST / PAIR / POS / ROOT / SEG buses), its
    transcript as rows of the transcript chip (tr_trace_kernel), the canonical bits of every query's index word, the folding
    chain and the reduced openings as rows of the query chip (qr_trace_kernel) - byte-identical to the oracle's, in one batch
    with a proof that has no payload; verified with the leaf, with its stub, with the statement derived from it, and not
    without."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1, seed=50).to_borsh())
    leaf = client.prove(pk, s).run()
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    host.verify(leaf, vk)
    handles, traces = [], []
    for i in range(2):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1, seed=51 + i).to_borsh())
        if i == 0:
            client.set_verified_leaf(s, leaf, vk)
        handles.append(client.machine_trace_handle(pk, s))
        traces.append(client.machine_trace(pk, s))
    assert len(traces[0]["leaf_p2_rows"]) > 4000 and len(traces[1]["leaf_p2_rows"]) == 0
    shape = zk.machine_cover_heights(handles)
    names = zk.MACHINE_CHIP_NAMES
    assert shape[names.index("poseidon2")] == 13 and shape[names.index("query")] == 8 and shape[names.index("transcript")] >= 6
    bodies = client.machine_prove_resident(pk, handles)
    proofs = [handles[i].proof_from_body(pk, bodies[i], shape) for i in range(2)]
    for i in range(2):
        assert proofs[i].to_bytes() == oracle.machine_prove(dict(traces[i], shape=shape), num_queries=nq, pow_bits=pw), i
    host.verify(proofs[1], vk)
    host.verify_with_leaf(proofs[0], vk, leaf, vk)
    host.verify_with_leaf(proofs[0], vk, leaf.stub(), vk)
    host.verify_public(proofs[0], vk, host.leaf_public(leaf, vk))
    assert len(host.leaf_public(leaf, vk)) < 80  # (stage 2a: 356 at these parameters, 4 500 at the full ones)
    with pytest.raises(zk.VerificationError):
        host.verify(proofs[0], vk)
    # the drop-in call gives the same proof
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1, seed=51).to_borsh())
    client.set_verified_leaf(s, leaf, vk)
    again = client.prove(pk, s).run()
    host.verify_with_leaf(again, vk, leaf, vk)
    assert again.public_tuples == proofs[0].public_tuples


def test_two_leaf_checks_match_oracle(zk, fx, oracle):
    """Several leaves beside one run (zksp_stdin_add_verified_leaf: the node of a recursion tree of that arity): the device's
    proof with the query phases of TWO leaf proofs - of different shapes - is the oracle's byte for byte, verifies with the
    leaves in the order they were added and not in the other."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    leaves = []
    for m in (fx.acct_fixture(1, seed=60), fx.slot_fixture(0)):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        leaves.append(client.prove(pk, s).run())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1, seed=61).to_borsh())
    for lf in leaves:
        client.add_verified_leaf(s, lf, vk)
    trace = client.machine_trace(pk, s)
    qr = trace["leaf_qr_rows"]
    assert len(qr) == 2 * nq * 31 and sorted(set((int(l), int(q)) for l, q in qr[:, 3:5])) == [(l, q) for l in range(2) for q in range(nq)]
    proof = client.prove(pk, s).run()
    raw = proof.to_bytes()
    assert raw == oracle.machine_prove(dict(trace, shape=shape_of(zk, raw)), num_queries=nq, pow_bits=pw)
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    host.verify_with_leaves(proof, vk, leaves, [vk, vk])
    with pytest.raises(zk.VerificationError):
        host.verify_with_leaves(proof, vk, leaves[::-1], [vk, vk])


def test_rv32m_guest_matches_oracle(zk, oracle):
    """Format v15's rows for mulh / mulhsu (multiplier chip) and div / divu / rem / remu (divider chip): the committed guest
    executes none of them, so a hand-assembled guest does, on the spec's corner cases (tests/test_rv32m.py).  The device's
    proof - trace expansion of both chips, their quotients, their LogUp columns - is the oracle's byte for byte."""
    import test_rv32m as rv
    import toy_guest as tg
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(tg.elf_of(rv.program()))
    stdin = zk.SP1Stdin()
    trace = client.machine_trace(pk, stdin)
    proof = client.prove(pk, stdin).run()
    assert proof.public_values == b""
    exp = oracle.machine_prove(trace, num_queries=nq, pow_bits=pw)
    got = np.frombuffer(proof.to_bytes(), dtype=np.uint32)
    e = np.frombuffer(exp, dtype=np.uint32)
    assert got.shape == e.shape and first_difference(got, e) is None, first_difference(got, e)
    zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw).verify(proof, vk)
    hts = shape_of(zk, proof.to_bytes())
    assert hts[zk.MACHINE_CHIP_NAMES.index("divider")] == 7


def test_two_level_tree_matches_oracle(zk, fx, oracle):
    """A node is a valid leaf (stage 2b): two leaf proofs -> a node that checks both -> a root that checks the node (whose own
    statement its header carries) and a third leaf.  Both inner proofs are the oracle's byte for byte; the root verifies from
    STUBS of everything below it (no query phase of a lower proof is read), and not with the leaves of the node swapped."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)

    def stdin_of(m):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        return s

    leaves = [client.prove(pk, stdin_of(m)).run() for m in (fx.acct_fixture(1, seed=70), fx.slot_fixture(1), fx.acct_fixture(2, seed=71))]
    s = stdin_of(fx.acct_fixture(1, seed=72))
    client.add_verified_leaf(s, leaves[0], vk)
    client.add_verified_leaf(s, leaves[1], vk)
    trace = client.machine_trace(pk, s)
    node = client.prove(pk, s).run()
    raw = node.to_bytes()
    assert raw == oracle.machine_prove(dict(trace, shape=shape_of(zk, raw)), num_queries=nq, pow_bits=pw)
    st_node = host.leaves_public(leaves[:2], [vk, vk])
    host.verify_public(node, vk, st_node)
    s = stdin_of(fx.acct_fixture(1, seed=73))
    client.add_verified_node(s, node, vk, st_node)
    client.add_verified_leaf(s, leaves[2], vk)
    trace = client.machine_trace(pk, s)
    root = client.prove(pk, s).run()
    raw = root.to_bytes()
    assert raw == oracle.machine_prove(dict(trace, shape=shape_of(zk, raw)), num_queries=nq, pow_bits=pw)
    stubs = [p.stub() for p in leaves]
    tree = [(node.stub(), [(stubs[0], []), (stubs[1], [])]), (stubs[2], [])]
    host.verify_tree(root, vk, tree)
    assert sum(len(p.to_bytes()) for p in stubs + [node.stub()]) < sum(len(p.to_bytes()) for p in leaves + [node]) // 2
    with pytest.raises(zk.VerificationError):
        host.verify_tree(root, vk, [(node.stub(), [(stubs[1], []), (stubs[0], [])]), (stubs[2], [])])
    with pytest.raises(zk.VerificationError):
        host.verify(root, vk)


def test_deferred_leaf_checks_equal_attached_ones(zk, fx):
    """zksp_stdin_defer_verified_leaves: the leaf checks made by the prove_batch call itself (on its tracing threads, beside the
    proving) give byte for byte the proofs - and the statements - that attaching the checks first gives; a run whose leaf does
    not verify fails alone, with ERR_VERIFY; a whole tree level through farm.prove_tree_level verifies from stubs."""
    nq, pw = 8, 6
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=4)
    pk, vk = client.setup(zk.merkle_elf())
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    farm = __import__("importlib").import_module("zk-state-proofs_amd.farm")

    def stdin_of(m):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        return s

    leaves, status = client.prove_batch(pk, [stdin_of(fx.acct_fixture(1 + i % 2, seed=300 + i)) for i in range(12)])
    assert status == [0] * 12
    groups = [list(range(a, a + 2)) for a in range(0, 12, 2)]
    # attached first
    attached = []
    for k, g in enumerate(groups):
        s = stdin_of(fx.acct_fixture(1, seed=400 + k))
        client.add_verified_leaves(s, [leaves[i] for i in g], [vk, vk])
        attached.append(s)
    st_attached = [client.stdin_statement(s) for s in attached]
    want, status = client.prove_batch(pk, attached)
    assert status == [0] * 6
    # deferred to the call
    deferred = []
    for k, g in enumerate(groups):
        s = stdin_of(fx.acct_fixture(1, seed=400 + k))
        client.defer_verified_leaves(s, [leaves[i] for i in g], [vk, vk])
        assert len(client.stdin_statement(s)) == 0  # nothing is checked yet
        deferred.append(s)
    got, status = client.prove_batch(pk, deferred)
    assert status == [0] * 6
    for k in range(6):
        assert got[k].to_bytes() == want[k].to_bytes()
        assert np.array_equal(client.stdin_statement(deferred[k]), st_attached[k])
        host.verify_with_leaves(got[k], vk, [leaves[i].stub() for i in groups[k]], [vk, vk])
    # a leaf that does not verify: that run fails, the others are proven
    raw = bytearray(leaves[3].to_bytes())
    raw[-40] ^= 1
    bad = zk.SP1ProofWithPublicValues.from_bytes(bytes(raw))
    mixed = []
    for k, g in enumerate(groups[:3]):
        s = stdin_of(fx.acct_fixture(1, seed=400 + k))
        client.defer_verified_leaves(s, [bad if i == 3 else leaves[i] for i in g], [vk, vk])
        mixed.append(s)
    got2, status = client.prove_batch(pk, mixed)
    assert status == [0, zk.client.ERR_VERIFY, 0] and got2[1] is None
    assert got2[0].to_bytes() == want[0].to_bytes() and got2[2].to_bytes() == want[2].to_bytes()
    assert "leaf" in client.last_error()
    # the farm's level (which defers where the client can) and the tree's verification from stubs
    node_stdins = [stdin_of(fx.acct_fixture(1, seed=400 + k)) for k in range(6)]
    st_of = {}
    mine, nodes, status = farm.prove_tree_level(client, pk, vk, leaves, node_stdins, 2, 0, 1, statements_out=st_of)
    assert status == [0] * 6 and [p.to_bytes() for p in nodes] == [p.to_bytes() for p in want]
    assert all(np.array_equal(st_of[k], st_attached[k]) for k in range(6))


def test_ramped_call_equals_plain_calls(zk, fx, oracle):
    """A prove_batch call of 384 runs or more on a client of max_batch >= 96 ramps up (api_prove.cpp): a first wave of 24 runs,
    waves growing by 7/4, the first two of them on the side lanes although chunks are copied beside them.  Every proof of such a
    call equals, byte for byte, the proof of the same run made alone when the shapes agree, and the oracle's for its shape
    otherwise; all verify and carry the right public values."""
    nq, pw = 6, 5
    client = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=96)
    pk, vk = client.setup(zk.merkle_elf())
    n = 400
    inputs = [fx.acct_fixture(1, seed=2000 + i) for i in range(n)]
    stdins = []
    for m in inputs:
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        stdins.append(s)
    proofs, status = client.prove_batch(pk, stdins)
    assert status == [0] * n
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pw)
    single = zk.ProverClient(device=0, num_queries=nq, pow_bits=pw, max_batch=1)
    pk1, vk1 = single.setup(zk.merkle_elf())
    # one run of every wave of the ramp (24, 42, 73, 127, then the rest) and the call's ends
    for i in (0, 23, 24, 65, 66, 138, 139, 265, 266, 330, n - 1):
        assert proofs[i].public_values == fx.ACCOUNT_VALUE
        host.verify(proofs[i], vk)
        s = zk.SP1Stdin()
        s.write(inputs[i].to_borsh())
        trace = single.machine_trace(pk1, s)
        q = single.prove(pk1, s).run()
        raw = proofs[i].to_bytes()
        if shape_of(zk, raw) == shape_of(zk, q.to_bytes()):
            assert q.to_bytes() == raw
        else:
            assert raw == oracle.machine_prove(dict(trace, shape=shape_of(zk, raw)), num_queries=nq, pow_bits=pw)
