"""Parity of every HIP kernel against the CPU oracle, through the C-ABI.
Bit-exact: all arithmetic is integer (BabyBear residues)."""
import os

import numpy as np
import pytest

from util import P

pytestmark = pytest.mark.gpu
# the keccak-only kernels belong to the round-1 component path: a build switch (ZKSP_COMPONENT=1; include/zksp_component.h)
component = pytest.mark.skipif(os.environ.get("ZKSP_COMPONENT", "") != "1", reason="component path not built (ZKSP_COMPONENT=1)")


def rnd(rng, shape):
    return rng.integers(0, P, shape, dtype=np.uint32)


def test_poseidon2_permute(gpu, oracle):
    rng = np.random.default_rng(1)
    st = rnd(rng, (1000, 16))
    st[0] = 0
    st[1] = P - 1
    got = gpu.permute(st)
    exp = np.stack([oracle.poseidon2_permute(s) for s in st])
    assert (got == exp).all()


@pytest.mark.parametrize("width,logn", [(1, 0), (3, 1), (8, 3), (9, 5), (16, 9), (37, 11), (2633, 8), (8, 12)])
def test_merkle_commit(gpu, oracle, width, logn):
    rng = np.random.default_rng(width * 100 + logn)
    mat = rnd(rng, (width, 1 << logn))
    assert (gpu.merkle_commit(mat) == oracle.merkle_commit(mat)).all()


def extreme_words(rng, shape):
    """Canonical values whose MONTGOMERY-form words (what the device holds) are extreme: 0, 1, p-1,
    p-2 and the two words around p/2, where the signed lazy arithmetic's bounds are tightest."""
    from util import from_monty
    words = np.array([0, 1, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2], dtype=np.uint32)
    return from_monty(words[rng.integers(0, len(words), shape)])


def test_poseidon2_permute_extreme_words(gpu, oracle):
    rng = np.random.default_rng(90)
    from util import from_monty
    st = extreme_words(rng, (600, 16))
    for i, w in enumerate((0, 1, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2)):
        st[i] = from_monty(np.full(16, w, np.uint32))  # every word the same extreme
    got = gpu.permute(st)
    exp = np.stack([oracle.poseidon2_permute(s) for s in st])
    assert (got == exp).all()


@pytest.mark.parametrize("width,logn,extreme", [(70, 16, False), (300, 16, True), (2633, 16, False)])
def test_merkle_commit_throughput_kernel(gpu, oracle, width, logn, extreme):
    """More than 32768 rows and at least 64 columns select leaf_hash_trace_kernel, the kernel that
    dominates a proof (one lane per row, the signed lazy permutation, state kept signed between
    absorbs); small inputs take the cooperative kernel instead.  Ragged last absorb included."""
    rng = np.random.default_rng(width + logn)
    mat = extreme_words(rng, (width, 1 << logn)) if extreme else rnd(rng, (width, 1 << logn))
    got = gpu.merkle_commit(mat)
    exp = oracle.merkle_commit(mat)
    assert (got == exp).all()


@pytest.mark.parametrize("logh", [1, 2, 3, 5, 8, 10, 11, 13, 14])
def test_lde(gpu, oracle, logh):
    rng = np.random.default_rng(logh)
    ncols = 5 if logh < 13 else 2
    cols = rnd(rng, (ncols, 1 << logh))
    for shift in (1, 31, 12345):
        lde, coefs = gpu.lde(cols, shift)
        elde, ecoefs = oracle.coset_lde(cols, shift, True)
        assert (coefs == ecoefs).all(), (logh, shift)
        assert (lde == elde).all(), (logh, shift)


@component
def test_keccak_trace(gpu, oracle):
    rng = np.random.default_rng(7)
    st = rng.integers(0, 2**64, (5, 25), dtype=np.uint64)
    for logh in (7, 8):
        assert (gpu.keccak_trace(st, logh) == oracle.keccak_trace(st, logh)).all()
    # no real permutation at all: pure padding
    assert (gpu.keccak_trace(st[:0], 5) == oracle.keccak_trace(st[:0], 5)).all()


@component
def test_keccak_quotient(gpu, oracle):
    rng = np.random.default_rng(8)
    st = rng.integers(0, 2**64, (2, 25), dtype=np.uint64)
    logh = 6
    trace = oracle.keccak_trace(st, logh)
    lde = oracle.coset_lde(trace, 1)
    alpha, gamma, beta = rnd(rng, 4), rnd(rng, 4), rnd(rng, 4)
    phi, cum = oracle.bus_perm_trace(trace, gamma, beta)
    lde_p = oracle.coset_lde(phi, 1)
    exp = oracle.keccak_quotient_bus(lde, lde_p, alpha, gamma, beta, cum)
    assert (gpu.keccak_quotient(lde, lde_p, alpha, gamma, beta, cum) == exp).all()
    # a corrupted trace / running sum gives a (non-polynomial) quotient too: values must still agree
    trace[900, 5] ^= 1
    lde = oracle.coset_lde(trace, 1)
    lde_p[2, 1, 7] = (int(lde_p[2, 1, 7]) + 5) % P
    bad_cum = rnd(rng, 4)
    exp = oracle.keccak_quotient_bus(lde, lde_p, alpha, gamma, beta, bad_cum)
    assert (gpu.keccak_quotient(lde, lde_p, alpha, gamma, beta, bad_cum) == exp).all()


@component
@pytest.mark.parametrize("logh,nperms", [(5, 1), (7, 4), (9, 21), (11, 62)])
def test_bus_perm_trace(gpu, oracle, logh, nperms):
    """LogUp running sum (row a6, lookup argument): phi columns and cumulative sum."""
    rng = np.random.default_rng(50 + logh)
    st = rng.integers(0, 2**64, (nperms, 25), dtype=np.uint64)
    trace = oracle.keccak_trace(st, logh)
    gamma, beta = rnd(rng, 4), rnd(rng, 4)
    phi, cum = gpu.bus_perm_trace(trace, gamma, beta)
    ephi, ecum = oracle.bus_perm_trace(trace, gamma, beta)
    assert (phi == ephi).all() and (cum == ecum).all()
    assert (cum == oracle.bus_expected_sum(oracle.bus_io_limbs(st), gamma, beta)).all()


@pytest.mark.parametrize("logh,ncols,shifts", [(15, 3, (1, 31)), (16, 2, (1,)), (17, 1, (12345,)), (21, 2, (1,)),
                                               (22, 1, (31,))])
def test_lde_tall_columns(gpu, oracle, logh, ncols, shifts):
    """Heights above 2^14 take the two-pass (strided + chunk) path; 2^21 is the
    as-committed guest's CPU-chip height."""
    rng = np.random.default_rng(100 + logh)
    cols = rnd(rng, (ncols, 1 << logh))
    for shift in shifts:
        lde, coefs = gpu.lde(cols, shift)
        elde, ecoefs = oracle.coset_lde(cols, shift, True)
        assert (coefs == ecoefs).all(), (logh, shift)
        assert (lde == elde).all(), (logh, shift)


def test_lde_roundtrip_property_2p20(gpu):
    """Size-independent check without the oracle: LDE of a low-degree column agrees
    with direct evaluation at a few points of both cosets."""
    logh = 20
    h = 1 << logh
    rng = np.random.default_rng(5)
    deg = 7
    coef = [int(x) for x in rng.integers(0, P, deg)]
    w = pow(31, (P - 1) >> logh, P)
    # evaluations of a degree-6 polynomial over the subgroup, built with numpy modular arithmetic
    xs = np.ones(h, dtype=np.uint64)
    acc = np.uint64(1)
    pw = np.empty(h, dtype=np.uint64)
    cur = 1
    for i in range(h):
        pw[i] = cur
        cur = cur * w % P
    col = np.zeros(h, dtype=np.uint64)
    xp = np.ones(h, dtype=np.uint64)
    for c in coef:
        col = (col + np.uint64(c) * xp) % np.uint64(P)
        xp = xp * pw % np.uint64(P)
    lde, coefs = gpu.lde(col.astype(np.uint32).reshape(1, h), 1)
    assert coefs[0, :deg].tolist() == coef and not coefs[0, deg:].any()
    w2 = pow(31, (P - 1) >> (logh + 1), P)
    ev = lambda x: sum(c * pow(x, k, P) for k, c in enumerate(coef)) % P
    for c in range(2):
        shift = 31 * (w2 if c else 1) % P
        for m in (0, 1, 12345, h // 2, h - 1):
            assert int(lde[0, c, m]) == ev(shift * pow(w, m, P) % P)


@pytest.mark.parametrize("loghk", [1, 2, 6, 11])
def test_fri_fold(gpu, oracle, loghk):
    rng = np.random.default_rng(loghk)
    layer = rnd(rng, (2, 1 << loghk, 4))
    beta = rnd(rng, 4)
    for shift in (31, 31 * 31 % P, 777):
        assert (gpu.fri_fold(layer, shift, beta) == oracle.fri_fold(layer, shift, beta)).all()
