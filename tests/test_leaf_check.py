"""Leaf-proof check (stage 2a; reference stub circuits/sp1-merkle-proof-recursive/src/main.rs:3-5):
a machine proof that also establishes that the QUERY PHASE of another machine proof verifies - every Merkle opening of its
four commitment rounds (sponges over the opened rows, paths with their mixed-height injections), every FRI layer opening and
the folding chain down to the final constant - as rows of the Poseidon2 chip and of the FRI fold chip, closed by public bus
tuples that carry the leaf's roots, query positions, challenges and reduced openings.

CPU tests: the records the host verifier logs are re-derived here from the leaf proof's bytes with the oracle's Poseidon2
(an independent replay of the hash work); the oracle proves, the product's verifier accepts - with the leaf, or with the
statement derived from it - and every tampering of an opened word, a path node, a pair or the statement is refused."""
import os

import numpy as np
import pytest

NQ, POW = 6, 5
P = 2013265921


@pytest.fixture(scope="module")
def setup(zk, fx, built_lib, oracle):
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    pk, vk = client.setup(zk.merkle_elf())
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(1).to_borsh())
    t_leaf = client.machine_trace(pk, s)
    leaf_bytes = oracle.machine_prove(t_leaf, num_queries=NQ, pow_bits=POW)
    leaf = zk.SP1ProofWithPublicValues.from_bytes(leaf_bytes)
    client.verify(leaf, vk)
    s2 = zk.SP1Stdin()
    s2.write(fx.acct_fixture(1, seed=2).to_borsh())
    client.set_verified_leaf(s2, leaf, vk)
    t = client.machine_trace(pk, s2)
    outer = oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
    return client, pk, vk, leaf, leaf_bytes, t, outer


def forced(oracle, t):
    os.environ["ZKSP_ORACLE_FORCE"] = "1"
    try:
        return oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
    finally:
        del os.environ["ZKSP_ORACLE_FORCE"]


def test_outer_proof_verifies_with_its_leaf(zk, setup):
    client, pk, vk, leaf, _, t, outer = setup
    proof = zk.SP1ProofWithPublicValues.from_bytes(outer)
    n_pub, digest = proof.public_tuples
    assert n_pub == len(t["leaf_pub_tuples"]) > 0 and any(digest)
    client.verify_with_leaf(proof, vk, leaf, vk)
    # the statement alone (what a verifier that never sees the leaf's openings is given) does as well
    tuples = client.leaf_public(leaf, vk)
    assert np.array_equal(tuples, t["leaf_pub_tuples"])
    client.verify_public(proof, vk, tuples)
    # a plain verify cannot vouch for a statement it was not given
    with pytest.raises(zk.VerificationError) as ei:
        client.verify(proof, vk)
    assert "public bus tuples" in str(ei.value)
    # the leaf itself carries none
    assert leaf.public_tuples == (0, [0] * 8)


def test_statement_is_bound(zk, setup):
    """Any other statement - another root, position key, challenge, reduced opening, final constant, one tuple less - is
    not what the proof closes its buses with: the header's digest (absorbed before any challenge) differs."""
    client, pk, vk, leaf, _, t, outer = setup
    proof = zk.SP1ProofWithPublicValues.from_bytes(outer)
    tuples = t["leaf_pub_tuples"]
    rng = np.random.default_rng(7)
    for k in rng.integers(0, len(tuples), 12):
        bad = tuples.copy()
        n_el = int(bad[k, 3])
        bad[k, 4 + int(rng.integers(0, n_el))] ^= 1
        with pytest.raises(zk.VerificationError):
            client.verify_public(proof, vk, bad)
    with pytest.raises(zk.VerificationError):
        client.verify_public(proof, vk, tuples[:-1])
    # ... and with the digest patched into the header to match, the buses do not balance
    bad = tuples.copy()
    k = int(np.nonzero(bad[:, 0] == 13)[0][3])  # a DIGEST tuple: the root an opening must reach
    bad[k, 4 + 5] = (int(bad[k, 4 + 5]) + 1) % P
    raw = bytearray(outer)
    hw = zk.MACHINE_HEADER_WORDS
    import importlib
    orc = importlib.import_module("oracle")
    dg = orc.hash_elems(bad.reshape(-1))
    raw[4 * (hw - 8):4 * hw] = np.asarray(dg, np.uint32).tobytes()
    with pytest.raises(zk.VerificationError):
        client.verify_public(zk.SP1ProofWithPublicValues.from_bytes(bytes(raw)), vk, bad)


def test_records_replay_from_the_leaf_proof(zk, oracle, setup):
    """The Poseidon2-chip records re-derived independently: every row's input state follows from the rows before it by the
    oracle's permutation (sponges chain their capacity, path steps and injections take the running digest), every run ends
    in the root its DIGEST tuple names, every absorbed word and every sibling is a word of the leaf proof's query section,
    and the fold records chain to the final constant."""
    client, pk, vk, leaf, leaf_bytes, t, _ = setup
    rows, folds, tuples = t["leaf_p2_rows"], t["leaf_fold_rows"], t["leaf_pub_tuples"]
    K_NODE, K_SZ, K_SC, K_PL, K_PR, K_J = 1, 2, 3, 4, 5, 6
    roots = {int(x[4]): x for x in tuples if x[0] == 13}  # tag -> DIGEST tuple
    parked = {}
    prev_out, prev = None, None
    ends = 0
    for r in rows:
        flags, tag, key, mask = (int(v) for v in r[:4])
        kind, new, snd = flags & 15, bool(flags & 16), bool(flags & 32)
        st = [int(v) for v in r[4:]]
        if kind == K_SZ:
            assert st[8:] == [0] * 8 and (not new or (key, mask) == (1, 0))
        elif kind == K_SC:
            assert prev[0] in (K_SZ, K_SC) and st[8:] == prev_out[8:] and (tag, key, mask, new) == prev[1:]
        elif kind in (K_PL, K_PR):
            side = 8 if kind == K_PR else 0
            assert st[side:side + 8] == prev_out[:8] and tag == prev[1]
            assert key == 2 * prev[2] + (kind == K_PR) and mask == 2 * prev[3]
            assert prev[0] in (K_PL, K_PR, K_J) or (prev[0] in (K_SZ, K_SC) and prev[4])
        elif kind == K_J:
            assert prev[0] in (K_PL, K_PR) and st[:8] == prev_out[:8] and (tag, key, mask) == (prev[1], prev[2], prev[3] + 1)
            assert parked.pop((tag, key, mask)) == st[8:]
        else:
            raise AssertionError(kind)
        out = [int(v) for v in oracle.poseidon2_permute(st)]
        if snd:
            if kind in (K_SZ, K_SC):
                parked[(tag, key, mask)] = out[:8]  # the hash of an injected row, for the injection with these labels
            else:
                d = roots.pop(tag)
                assert [int(v) for v in d[4:16]] == [tag, 0, key, mask] + out[:8]
                ends += 1
        prev_out, prev = out, (kind, tag, key, mask, new)
    assert not parked and not roots and ends == NQ * (4 + max(int(f[2]) for f in folds) + 1)
    # the words: everything absorbed or used as a sibling comes out of the leaf proof's bytes
    words = set(np.frombuffer(leaf_bytes, np.uint32).tolist()) | {0}
    assert set(rows[:, 4:12][(rows[:, 0] & 15 <= K_SC) & (rows[:, 0] & 15 >= K_SZ)].reshape(-1).tolist()) <= words
    sib_l = rows[(rows[:, 0] & 15) == K_PR][:, 4:12]
    sib_r = rows[(rows[:, 0] & 15) == K_PL][:, 12:20]
    assert set(sib_l.reshape(-1).tolist()) <= words and set(sib_r.reshape(-1).tolist()) <= words
    # the folds: E chains to F + RO, the last F (+ RO) is the final constant
    inv = lambda v: pow(int(v), P - 2, P)

    def e_mul(a, b):
        r = [0] * 7
        for i in range(4):
            for j in range(4):
                r[i + j] += a[i] * b[j]
        return [(r[i] + 11 * (r[i + 4] if i < 3 else 0)) % P for i in range(4)]

    fin = {int(x[4]): [int(v) for v in x[6:10]] for x in tuples if x[0] == 18}
    ro0 = {int(x[4]): [int(v) for v in x[6:10]] for x in tuples if x[0] == 17 and x[5] == 0}
    expect = None
    for f in folds:
        flags, q, k, xinv = (int(v) for v in f[:4])
        beta, lo, hi, ro = ([int(v) for v in f[4 + 4 * i:8 + 4 * i]] for i in range(4))
        e = hi if flags & 4 else lo
        if flags & 1:
            assert k == 0 and e == ro0[q]
        else:
            assert e == expect
        half = inv(2)
        d = [(lo[i] - hi[i]) * half * xinv % P for i in range(4)]
        bd = e_mul(beta, d)
        expect = [((lo[i] + hi[i]) * half + bd[i] + ro[i]) % P for i in range(4)]
        if flags & 2:
            assert expect == fin[q]


@pytest.mark.parametrize("what", ["absorbed word", "sibling", "pair", "fold bit"])
def test_tampered_openings_are_refused(zk, oracle, setup, what):
    """Flip an opened word, a path node or a FRI pair of the leaf's query phase in the records: the honest prover's buses
    do not balance (it refuses), and the proof a cheating prover would send is rejected by the verifier that holds the
    leaf's statement."""
    client, pk, vk, leaf, _, t, _ = setup
    rows, folds = t["leaf_p2_rows"].copy(), t["leaf_fold_rows"].copy()
    kind = rows[:, 0] & 15
    if what == "absorbed word":
        k = int(np.nonzero((kind == 3) & ((rows[:, 0] & 64) == 0))[0][5])  # a continuing sponge row: an opened row's words
        rows[k, 4 + 3] = (int(rows[k, 4 + 3]) + 1) % P
    elif what == "sibling":
        k = int(np.nonzero(kind == 4)[0][7])  # a path step with the running digest on the left: the sibling on the right
        rows[k, 12 + 2] = (int(rows[k, 12 + 2]) + 1) % P
    elif what == "pair":
        k = int(np.nonzero((rows[:, 0] & 64) != 0)[0][3])  # a FRI leaf: the fold chip holds the untampered pair
        rows[k, 4 + 1] = (int(rows[k, 4 + 1]) + 1) % P
    else:
        folds[2, 0] ^= 4
    t2 = dict(t, leaf_p2_rows=rows, leaf_fold_rows=folds)
    with pytest.raises(RuntimeError):
        oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)
    bad = zk.SP1ProofWithPublicValues.from_bytes(forced(oracle, t2))
    with pytest.raises(zk.VerificationError):
        client.verify_with_leaf(bad, vk, leaf, vk)


def test_a_bad_leaf_has_no_check(zk, fx, setup):
    """set_verified_leaf verifies the leaf first: flipping any opened word or path node of the leaf proof leaves the honest
    prover with nothing to prove."""
    client, pk, vk, leaf, leaf_bytes, _, _ = setup
    rng = np.random.default_rng(11)
    body0 = 4 * (zk.MACHINE_HEADER_WORDS + (len(leaf.public_values) + 3) // 4)
    for pos in rng.integers(body0 + 4 * 3000, len(leaf_bytes), 6):
        raw = bytearray(leaf_bytes)
        raw[int(pos) & ~3] ^= 1
        bad = zk.SP1ProofWithPublicValues.from_bytes(bytes(raw))
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(1).to_borsh())
        with pytest.raises(zk.VerificationError):
            client.set_verified_leaf(s, bad, vk)


def test_check_of_another_leaf_is_refused(zk, fx, oracle, setup):
    """The outer proof is about ONE leaf: verifying it with another valid leaf's statement fails."""
    client, pk, vk, leaf, _, t, outer = setup
    s = zk.SP1Stdin()
    s.write(fx.slot_fixture(0).to_borsh())
    other = zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(client.machine_trace(pk, s), num_queries=NQ, pow_bits=POW))
    client.verify(other, vk)
    assert other.to_bytes() != leaf.to_bytes()
    with pytest.raises(zk.VerificationError):
        client.verify_with_leaf(zk.SP1ProofWithPublicValues.from_bytes(outer), vk, other, vk)


def _run_rows(rows, tag):
    """Indices of the rows of the run tagged `tag`: its injected-row sponges first (not NEW), then the run proper."""
    idx = np.nonzero(rows[:, 1] == tag)[0]
    return idx


def test_structural_forgeries_are_refused(zk, oracle, setup):
    """Forgeries that keep every HASH consistent and change only what kind of step a row claims to be.
    (1) An injection passed off as a path step with the injected hash as a free sibling - so that the shorter matrices'
    opened row need not be shown: the same permutation, so every later digest is unchanged and the root is reached, but the
    position key gains a level and the injection mask loses one - the run's last tuple is not the one the verifier
    consumes (and the orphaned sponge's tuple is consumed by nobody).
    (2) The sponge rows of an injected matrix row dropped altogether: the injection row finds nothing to consume.
    (3) A run that ends one level early and claims the root there: its key is too short."""
    client, pk, vk, leaf, _, t, _ = setup
    K_PL, K_J = 4, 6
    base = t["leaf_p2_rows"]
    tag = 2  # query 0, main round: a mixed-height opening
    idx = _run_rows(base, tag)
    kinds = base[idx, 0] & 15

    def refused(rows):
        t2 = dict(t, leaf_p2_rows=rows)
        with pytest.raises(RuntimeError):
            oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)
        bad = zk.SP1ProofWithPublicValues.from_bytes(forced(oracle, t2))
        with pytest.raises(zk.VerificationError):
            client.verify_with_leaf(bad, vk, leaf, vk)

    # (1) the first injection of the run becomes a path step; keys and masks of the rest of the run follow the chip's rules
    rows = base.copy()
    run = [i for i in idx if (base[i, 0] & 16) or (base[i, 0] & 15) >= K_PL]  # the NEW sponge, then the steps
    j0 = next(i for i in run if (base[i, 0] & 15) == K_J)
    rows[j0, 0] = (rows[j0, 0] & ~np.uint32(15)) | K_PL
    key, mask = int(rows[j0 - 1, 2]), int(rows[j0 - 1, 3])
    for i in [r for r in run if r >= j0]:
        k = int(rows[i, 0] & 15)
        if k == K_J:
            mask += 1
        else:
            key, mask = 2 * key + (1 if k == 5 else 0), 2 * mask
        rows[i, 2], rows[i, 3] = key, mask % P
    refused(rows)
    # (2) the sponge that hashes the first injected row is cut out
    first = int(idx[0])
    seg = [first]
    while not (base[seg[-1], 0] & 32):
        seg.append(seg[-1] + 1)
    refused(np.delete(base, seg, axis=0))
    # (3) the run stops one step early: that row claims to be the end
    last = int(run[-1])
    rows = np.delete(base, [last], axis=0)
    rows[last - 1, 0] |= 32
    refused(rows)
    assert (kinds == K_J).sum() >= 2  # (the opening does have several injections)


# ---- several leaves beside one run (zksp_stdin_add_verified_leaf): the node of a recursion tree of that arity ----
@pytest.fixture(scope="module")
def two_leaves(zk, fx, built_lib, oracle, setup):
    client, pk, vk, leaf_a, _, _, _ = setup
    s = zk.SP1Stdin()
    s.write(fx.slot_fixture(0).to_borsh())
    leaf_b = zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(client.machine_trace(pk, s), num_queries=NQ, pow_bits=POW))
    client.verify(leaf_b, vk)
    s2 = zk.SP1Stdin()
    s2.write(fx.acct_fixture(1, seed=3).to_borsh())
    client.add_verified_leaf(s2, leaf_a, vk)
    client.add_verified_leaf(s2, leaf_b, vk)
    t = client.machine_trace(pk, s2)
    outer = oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
    return client, pk, vk, leaf_a, leaf_b, t, outer


def test_two_leaves_are_checked_by_one_proof(zk, two_leaves, setup):
    client, pk, vk, leaf_a, leaf_b, t, outer = two_leaves
    proof = zk.SP1ProofWithPublicValues.from_bytes(outer)
    one = setup[5]
    # the second leaf's rows follow the first one's; its queries are numbered from NQ on (tags 1 + 64 q + r, fold rows, tuples)
    n1 = len(one["leaf_p2_rows"])
    assert np.array_equal(t["leaf_p2_rows"][:n1], one["leaf_p2_rows"]) and len(t["leaf_p2_rows"]) > n1
    assert np.array_equal(t["leaf_fold_rows"][:len(one["leaf_fold_rows"])], one["leaf_fold_rows"])
    second = t["leaf_fold_rows"][len(one["leaf_fold_rows"]):]
    assert sorted(set(int(q) for q in second[:, 1])) == list(range(NQ, 2 * NQ))
    tags = t["leaf_p2_rows"][n1:, 1]
    assert int(tags.min()) >= 1 + 64 * NQ and len(set(int(x) for x in t["leaf_p2_rows"][:n1, 1]) & set(int(x) for x in tags)) == 0
    # the statement: the two leaves' tuples one after the other, in the order they were added
    tuples = client.leaves_public([leaf_a, leaf_b], [vk, vk])
    assert np.array_equal(tuples, t["leaf_pub_tuples"])
    assert np.array_equal(tuples[:len(one["leaf_pub_tuples"])], client.leaf_public(leaf_a, vk))
    assert proof.public_tuples[0] == len(tuples)
    client.verify_with_leaves(proof, vk, [leaf_a, leaf_b], [vk, vk])
    client.verify_public(proof, vk, tuples)
    # the chips grew with the work: twice the permutations, twice the folds
    hts = [int.from_bytes(outer[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
    h1 = [int.from_bytes(setup[6][8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
    p2 = zk.MACHINE_CHIP_NAMES.index("poseidon2")
    assert hts[p2] >= h1[p2] and (1 << hts[p2]) >= len(t["leaf_p2_rows"])


def test_other_leaves_another_order_or_one_leaf_less_are_refused(zk, two_leaves):
    client, pk, vk, leaf_a, leaf_b, t, outer = two_leaves
    proof = zk.SP1ProofWithPublicValues.from_bytes(outer)
    for leaves in ([leaf_b, leaf_a], [leaf_a], [leaf_b], [leaf_a, leaf_a], [leaf_a, leaf_b, leaf_a]):
        with pytest.raises(zk.VerificationError):
            client.verify_with_leaves(proof, vk, leaves, [vk] * len(leaves))
    with pytest.raises(zk.VerificationError):
        client.verify_with_leaf(proof, vk, leaf_a, vk)
    with pytest.raises(zk.VerificationError):
        client.verify(proof, vk)


def test_a_tampered_second_leaf_check_is_refused(zk, oracle, two_leaves, setup):
    """An opened word of the SECOND leaf changed in the records: the honest oracle refuses, a forced proof is rejected (the
    first leaf's check being intact does not help)."""
    client, pk, vk, leaf_a, leaf_b, t, outer = two_leaves
    n1 = len(setup[5]["leaf_p2_rows"])
    rows = t["leaf_p2_rows"].copy()
    k = n1 + int(np.nonzero((rows[n1:, 0] & 15) == 2)[0][5])  # a first sponge block of the second leaf
    rows[k, 4 + 2] = (int(rows[k, 4 + 2]) + 1) % P
    t2 = dict(t, leaf_p2_rows=rows)
    with pytest.raises(RuntimeError):
        oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)
    with pytest.raises(zk.VerificationError):
        client.verify_with_leaves(zk.SP1ProofWithPublicValues.from_bytes(forced(oracle, t2)), vk, [leaf_a, leaf_b], [vk, vk])
