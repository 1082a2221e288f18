"""Parity of every HIP kernel against the CPU oracle, through the C-ABI.
Bit-exact: all arithmetic is integer (BabyBear residues)."""
import numpy as np
import pytest

from util import P

pytestmark = pytest.mark.gpu


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


def test_keccak_trace(gpu, oracle):
    rng = np.random.default_rng(7)
    st = rng.integers(0, 2**64, (5, 25), dtype=np.uint64)
    for logh in (7, 8):
        assert (gpu.keccak_trace(st, logh) == oracle.keccak_trace(st, logh)).all()
    # no real permutation at all: pure padding
    assert (gpu.keccak_trace(st[:0], 5) == oracle.keccak_trace(st[:0], 5)).all()


def test_keccak_quotient(gpu, oracle):
    rng = np.random.default_rng(8)
    st = rng.integers(0, 2**64, (2, 25), dtype=np.uint64)
    logh = 6
    trace = oracle.keccak_trace(st, logh)
    lde = oracle.coset_lde(trace, 1)
    alpha = rnd(rng, 4)
    assert (gpu.keccak_quotient(lde, alpha) == oracle.keccak_quotient(lde, alpha)).all()
    # a corrupted trace gives a (non-polynomial) quotient too: values must still agree
    trace[900, 5] ^= 1
    lde = oracle.coset_lde(trace, 1)
    assert (gpu.keccak_quotient(lde, alpha) == oracle.keccak_quotient(lde, alpha)).all()


@pytest.mark.parametrize("loghk", [1, 2, 6, 11])
def test_fri_fold(gpu, oracle, loghk):
    rng = np.random.default_rng(loghk)
    layer = rnd(rng, (2, 1 << loghk, 4))
    beta = rnd(rng, 4)
    for shift in (31, 31 * 31 % P, 777):
        assert (gpu.fri_fold(layer, shift, beta) == oracle.fri_fold(layer, shift, beta)).all()
