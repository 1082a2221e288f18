"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes front-end to ``oracle/_build/libzksp_oracle.so`` (the plain-C CPU
restatement, see ``zksp_oracle.h``) plus a direct restatement of the reference's
``crypto_ops::verify_merkle_proof`` (reference crypto-ops/src/lib.rs:8-23).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module; the product path never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libzksp_oracle.so")

P = 2013265921
KA_WIDTH = 2633
KA_NUM_CONSTRAINTS = 3182


def build(force: bool = False) -> str:
    srcs = ["poseidon2.c", "ntt.c", "keccak_air.c", "bus.c", "prover.c", "machine.c", "mprover.c", "machine.h", "field.h",
            "zksp_oracle.h", "Makefile"]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib: Optional[C.CDLL] = None


class Header(C.Structure):
    _fields_ = [("log_h", C.c_uint32), ("n_perms", C.c_uint32), ("exit_code", C.c_uint32), ("pv_len", C.c_uint32),
                ("pv_digest", C.c_uint32 * 8), ("deferred_digest", C.c_uint32 * 8), ("vk_digest", C.c_uint32 * 8)]


class Config(C.Structure):
    _fields_ = [("num_queries", C.c_uint32), ("pow_bits", C.c_uint32)]


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint32 * 16), ("inbuf", C.c_uint32 * 8), ("n_in", C.c_int),
                ("outbuf", C.c_uint32 * 8), ("n_out", C.c_int)]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_proof_size.restype = C.c_size_t
        _lib.orc_proof_header_words.restype = C.c_size_t
        _lib.orc_merkle_layer_offset.restype = C.c_size_t
        _lib.orc_ch_sample.restype = C.c_uint32
        _lib.orc_ch_sample_bits.restype = C.c_uint32
        _lib.orc_ch_grind.restype = C.c_uint32
        _lib.orc_machine_proof_size.restype = C.c_size_t
        _lib.orc_machine_events.restype = C.c_size_t
        _lib.orc_machine_chip.restype = C.POINTER(MachineChip)
    return _lib


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def poseidon2_constants() -> Tuple[np.ndarray, np.ndarray]:
    ext = np.zeros(128, np.uint32)
    inn = np.zeros(13, np.uint32)
    lib().orc_poseidon2_constants(_p(ext), _p(inn))
    return ext.reshape(8, 16), inn


def poseidon2_permute(state: Sequence[int]) -> np.ndarray:
    s = _u32(state).copy()
    assert s.shape == (16,)
    lib().orc_poseidon2_permute(_p(s))
    return s


def hash_elems(v: Sequence[int]) -> np.ndarray:
    a = _u32(v)
    out = np.zeros(8, np.uint32)
    lib().orc_hash_elems(_p(a), C.c_size_t(a.size), _p(out))
    return out


def compress(l: Sequence[int], r: Sequence[int]) -> np.ndarray:
    out = np.zeros(8, np.uint32)
    la, ra = _u32(l), _u32(r)
    lib().orc_compress(_p(la), _p(ra), _p(out))
    return out


def ntt(a: np.ndarray, inverse: bool = False) -> np.ndarray:
    a = _u32(a).copy()
    logn = int(a.size).bit_length() - 1
    assert a.size == 1 << logn
    lib().orc_ntt(_p(a), logn, int(inverse))
    return a


def dft_naive(a: np.ndarray) -> np.ndarray:
    a = _u32(a)
    logn = int(a.size).bit_length() - 1
    out = np.zeros_like(a)
    lib().orc_dft_naive(_p(a), _p(out), logn)
    return out


def coset_lde(cols: np.ndarray, in_shift: int = 1, want_coefs: bool = False):
    """cols: [ncols][H] -> lde [ncols][2][H] (and coefs [ncols][H])."""
    cols = _u32(cols)
    ncols, h = cols.shape
    logh = h.bit_length() - 1
    out = np.zeros((ncols, 2, h), np.uint32)
    coefs = np.zeros((ncols, h), np.uint32) if want_coefs else None
    lib().orc_coset_lde(_p(cols), logh, ncols, C.c_uint32(in_shift), _p(out), _p(coefs) if want_coefs else None)
    return (out, coefs) if want_coefs else out


def merkle_commit(mat: np.ndarray) -> np.ndarray:
    """mat: column-major [W][N] -> tree digests [(2N-1)][8], leaves first, root last."""
    mat = _u32(mat)
    w, n = mat.shape
    logn = n.bit_length() - 1
    tree = np.zeros((2 * n - 1, 8), np.uint32)
    lib().orc_merkle_commit(_p(mat), w, logn, _p(tree))
    return tree


def merkle_layer_offset(logn: int, layer: int) -> int:
    return int(lib().orc_merkle_layer_offset(logn, layer))


def keccak_f(state: Sequence[int]) -> np.ndarray:
    s = np.ascontiguousarray(state, dtype=np.uint64).copy()
    lib().orc_keccak_f(_p(s))
    return s


def keccak_trace(states_in: np.ndarray, logh: int) -> np.ndarray:
    st = np.ascontiguousarray(states_in, dtype=np.uint64).reshape(-1, 25)
    trace = np.zeros((KA_WIDTH, 1 << logh), np.uint32)
    lib().orc_keccak_trace(_p(st), int(st.shape[0]), logh, _p(trace))
    return trace


def keccak_constraints(local: np.ndarray, nxt: np.ndarray, is_first: int, is_last: int, is_trans: int) -> np.ndarray:
    out = np.zeros(KA_NUM_CONSTRAINTS, np.uint32)
    l, n = _u32(local), _u32(nxt)
    lib().orc_keccak_constraints(_p(l), _p(n), C.c_uint32(is_first), C.c_uint32(is_last), C.c_uint32(is_trans), _p(out))
    return out


def keccak_quotient(lde: np.ndarray, alpha: Sequence[int]) -> np.ndarray:
    lde = _u32(lde)
    w, two, h = lde.shape
    assert w == KA_WIDTH and two == 2
    logh = h.bit_length() - 1
    a = _u32(alpha)
    out = np.zeros((8, h), np.uint32)
    lib().orc_keccak_quotient(_p(lde), logh, _p(a), _p(out))
    return out


def fri_fold(layer: np.ndarray, shift_k: int, beta: Sequence[int]) -> np.ndarray:
    """layer: [2][Hk][4] -> [2][Hk/2][4]."""
    layer = _u32(layer)
    _, hk, _ = layer.shape
    loghk = hk.bit_length() - 1
    b = _u32(beta)
    out = np.zeros((2, hk // 2, 4), np.uint32)
    lib().orc_fri_fold(_p(layer), loghk, C.c_uint32(shift_k), _p(b), _p(out))
    return out


class OracleChallenger:
    def __init__(self):
        self.c = Challenger()
        lib().orc_ch_init(C.byref(self.c))

    def observe(self, xs):
        for x in np.atleast_1d(_u32(xs)):
            lib().orc_ch_observe(C.byref(self.c), C.c_uint32(int(x)))

    def sample(self) -> int:
        return int(lib().orc_ch_sample(C.byref(self.c)))

    def sample_ext(self) -> List[int]:
        return [self.sample() for _ in range(4)]

    def sample_bits(self, bits: int) -> int:
        return int(lib().orc_ch_sample_bits(C.byref(self.c), bits))

    def grind(self, bits: int) -> int:
        return int(lib().orc_ch_grind(C.byref(self.c), bits))


def proof_size(logh: int, num_queries: int, pow_bits: int, pv_len: int, n_perms: int = 0) -> int:
    cfg = Config(num_queries, pow_bits)
    return int(lib().orc_proof_size(logh, C.byref(cfg), C.c_uint32(pv_len), C.c_uint32(n_perms)))


def proof_header_words(pv_len: int, n_perms: int) -> int:
    return int(lib().orc_proof_header_words(C.c_uint32(pv_len), C.c_uint32(n_perms)))


# ---- LogUp bus ----
def bus_io_limbs(states_in: np.ndarray) -> np.ndarray:
    st = np.ascontiguousarray(states_in, dtype=np.uint64).reshape(-1, 25)
    out = np.zeros((max(st.shape[0], 1), 200), np.uint32)
    lib().orc_bus_io_limbs(_p(st), int(st.shape[0]), _p(out))
    return out[: st.shape[0]]


def bus_io_log_rows(logh: int) -> int:
    return int(lib().orc_bus_io_log_rows(logh))


def bus_perm_trace(trace: np.ndarray, gamma: Sequence[int], beta: Sequence[int]):
    trace = _u32(trace)
    w, h = trace.shape
    logh = h.bit_length() - 1
    g, b = _u32(gamma), _u32(beta)
    phi = np.zeros((4, h), np.uint32)
    cum = np.zeros(4, np.uint32)
    lib().orc_bus_perm_trace(_p(trace), logh, _p(g), _p(b), _p(phi), _p(cum))
    return phi, cum


def bus_expected_sum(io_limbs: np.ndarray, gamma: Sequence[int], beta: Sequence[int]) -> np.ndarray:
    io = _u32(io_limbs).reshape(-1, 200)
    g, b = _u32(gamma), _u32(beta)
    out = np.zeros(4, np.uint32)
    lib().orc_bus_expected_sum(_p(io), int(io.shape[0]), _p(g), _p(b), _p(out))
    return out


def keccak_quotient_bus(lde: np.ndarray, lde_p: np.ndarray, alpha, gamma, beta, cum_sum) -> np.ndarray:
    lde, lde_p = _u32(lde), _u32(lde_p)
    w, two, h = lde.shape
    assert w == KA_WIDTH and two == 2 and lde_p.shape == (4, 2, h)
    logh = h.bit_length() - 1
    a, g, b, c = _u32(alpha), _u32(gamma), _u32(beta), _u32(cum_sum)
    out = np.zeros((8, h), np.uint32)
    lib().orc_keccak_quotient_bus(_p(lde), _p(lde_p), logh, _p(a), _p(g), _p(b), _p(c), _p(out))
    return out


def prove(states_in: np.ndarray, logh: int, *, exit_code: int = 0, public_values: bytes = b"",
          pv_digest: Sequence[int] = (0,) * 8, deferred_digest: Sequence[int] = (0,) * 8,
          vk_digest: Sequence[int] = (0,) * 8, num_queries: int = 100, pow_bits: int = 16) -> bytes:
    st = np.ascontiguousarray(states_in, dtype=np.uint64).reshape(-1, 25)
    hdr = Header(logh, int(st.shape[0]), exit_code, len(public_values),
                 (C.c_uint32 * 8)(*pv_digest), (C.c_uint32 * 8)(*deferred_digest), (C.c_uint32 * 8)(*vk_digest))
    cfg = Config(num_queries, pow_bits)
    cap = proof_size(logh, num_queries, pow_bits, len(public_values), int(st.shape[0]))
    buf = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    pv = (C.c_uint8 * max(1, len(public_values))).from_buffer_copy(public_values or b"\0")
    rc = lib().orc_prove(_p(st), C.byref(hdr), pv, C.byref(cfg), buf, C.c_size_t(cap), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"orc_prove failed rc={rc}")
    return bytes(buf[: n.value])


# ---------------------------------------------------------------------------
# machine proof (oracle/machine.h): inputs are the arrays ProverClient.machine_trace() returns
# ---------------------------------------------------------------------------
N_CHIPS = 27
CHIP_NAMES = ["cpu", "keccak", "keccak-mem", "mem-final", "image", "program", "mul", "table", "cpu2", "alu", "alu2",
              "subword", "subword2", "bitwise", "bitwise2", "poseidon2", "ecall"]
CPUPUB_N = 5


class MachineChip(C.Structure):
    _fields_ = [("name", C.c_char_p), ("prep_width", C.c_int), ("main_width", C.c_int), ("n_inter", C.c_int),
                ("inter", C.c_void_p), ("n_constraints", C.c_int), ("n_merged", C.c_int)]


class MachineInput(C.Structure):
    _fields_ = [("program", C.c_void_p), ("n_program", C.c_size_t), ("image", C.c_void_p), ("n_image", C.c_size_t),
                ("entry", C.c_uint32), ("text_base", C.c_uint32), ("log_prog", C.c_int), ("log_image", C.c_int),
                ("cycles", C.c_void_p), ("n_cycles", C.c_size_t), ("keccak", C.c_void_p), ("n_keccak", C.c_size_t),
                ("memfinal", C.c_void_p), ("n_memfinal", C.c_size_t), ("muls", C.c_void_p), ("n_muls", C.c_size_t),
                ("prog_mult", C.c_void_p), ("shape", C.c_void_p), ("agg_keys", C.c_void_p), ("agg_leaves", C.c_void_p), ("n_agg", C.c_size_t),
                ("leaf_p2_rows", C.c_void_p), ("n_leaf_p2", C.c_size_t), ("leaf_qr_rows", C.c_void_p), ("n_leaf_qr", C.c_size_t),
                ("leaf_tr_rows", C.c_void_p), ("n_leaf_tr", C.c_size_t),
                ("pub_tuples", C.c_void_p), ("n_pub", C.c_size_t)]


class MachinePublic(C.Structure):
    _fields_ = [("exit_code", C.c_uint32), ("pv_len", C.c_uint32), ("pv_digest", C.c_uint32 * 8),
                ("deferred_digest", C.c_uint32 * 8)]


def machine_input(t: dict):
    """MachineInput over the arrays of a machine trace; returns (struct, keep-alive list).  ``t["shape"]`` (optional):
    the chip log-heights to prove the run with (a batch shares one shape); default: the run's own minimal heights.
    ``t["agg_leaves"]`` (optional, [n][8] canonical words, n a power of two): the aggregation payload."""
    keep = {k: np.ascontiguousarray(t[k]) for k in ("program", "image", "cycles", "keccak", "memfinal", "muls",
                                                    "prog_mult")}
    info = t["info"]
    mi = MachineInput(_p(keep["program"]), len(keep["program"]), _p(keep["image"]), len(keep["image"]),
                      info.entry, int(keep["program"][0, 0]), info.log_prog, info.log_image,
                      _p(keep["cycles"]), len(keep["cycles"]), _p(keep["keccak"]), len(keep["keccak"]),
                      _p(keep["memfinal"]), len(keep["memfinal"]), _p(keep["muls"]), len(keep["muls"]),
                      _p(keep["prog_mult"]), None, None, None, 0, None, 0, None, 0, None, 0, None, 0)
    if t.get("agg_leaves") is not None and len(t["agg_leaves"]):
        keep["agg_leaves"] = np.ascontiguousarray(t["agg_leaves"], dtype=np.uint32).reshape(-1, 8)
        mi.agg_leaves = keep["agg_leaves"].ctypes.data
        mi.n_agg = len(keep["agg_leaves"])
        if t.get("agg_keys") is not None and len(t["agg_keys"]):
            keep["agg_keys"] = np.ascontiguousarray(t["agg_keys"], dtype=np.uint32)
            assert len(keep["agg_keys"]) == mi.n_agg
            mi.agg_keys = keep["agg_keys"].ctypes.data
    # leaf-proof check (row f4, stage 2b): the records of the product's host verifier (machine_trace's leaf_* sections)
    for key, field, count, words in (("leaf_p2_rows", "leaf_p2_rows", "n_leaf_p2", 32), ("leaf_qr_rows", "leaf_qr_rows", "n_leaf_qr", 132),
                                     ("leaf_tr_rows", "leaf_tr_rows", "n_leaf_tr", 32), ("leaf_pub_tuples", "pub_tuples", "n_pub", 16)):
        if t.get(key) is not None and len(t[key]):
            keep[key] = np.ascontiguousarray(t[key], dtype=np.uint32).reshape(-1, words)
            setattr(mi, field, keep[key].ctypes.data)
            setattr(mi, count, len(keep[key]))
    if t.get("shape") is not None:
        keep["shape"] = np.ascontiguousarray(t["shape"], dtype=np.int32)
        assert len(keep["shape"]) == N_CHIPS
        mi.shape = keep["shape"].ctypes.data
    return mi, keep


def machine_chip(chip: int):
    c = lib().orc_machine_chip(chip).contents
    return {"name": c.name.decode(), "prep_width": c.prep_width, "main_width": c.main_width, "n_inter": c.n_inter,
            "n_constraints": c.n_constraints, "n_merged": c.n_merged,
            "perm_width": 4 * ((c.n_inter - c.n_merged + 1) // 2 + (1 if c.n_merged else 0))}


def machine_heights(t: dict) -> List[int]:
    mi, _keep = machine_input(t)
    out = (C.c_int * N_CHIPS)()
    lib().orc_machine_heights(C.byref(mi), out)
    return list(out)


def machine_fill(t: dict, chip: int):
    """(prep [pw][H] or None, main [mw][H]) canonical traces of one chip."""
    mi, _keep = machine_input(t)
    logh = machine_heights(t)[chip]
    d = machine_chip(chip)
    h = 1 << logh
    prep = np.zeros((d["prep_width"], h), np.uint32) if d["prep_width"] else None
    main = np.zeros((d["main_width"], h), np.uint32)
    lib().orc_machine_fill(C.byref(mi), chip, logh, _p(prep) if prep is not None else None, _p(main))
    return prep, main


def machine_constraints(chip: int, prep_row, loc, nxt, is_first: int, is_last: int, is_trans: int, pub=(0, 0, 0, 0, 0)) -> np.ndarray:
    """`pub`: the five CPUPUB_* words (first pc, first time, has a successor, hand-over pc, padding pc) for the CPU
    instances."""
    d = machine_chip(chip)
    out = np.zeros(max(d["n_constraints"], 1), np.uint32)
    pr = _u32(prep_row) if prep_row is not None else np.zeros(1, np.uint32)
    l, n = _u32(loc), _u32(nxt)
    lib().orc_machine_constraints(chip, _p(pr), _p(l), _p(n), C.c_uint32(is_first), C.c_uint32(is_last),
                                  C.c_uint32(is_trans), _p(_u32(list(pub))), _p(out))
    return out[: d["n_constraints"]]


def machine_cpu_pub(t: dict, chip: int):
    mi, _keep = machine_input(t)
    out = (C.c_uint32 * CPUPUB_N)()
    lib().orc_machine_cpu_pub(C.byref(mi), chip, out)
    return [int(x) for x in out]


def machine_events(t: dict, which: int) -> np.ndarray:
    """The oracle's own list of cycle indices that occupy ALU-chip (0) / sub-word-chip (1) / bitwise-chip (2) rows."""
    mi, _keep = machine_input(t)
    n = int(lib().orc_machine_events(C.byref(mi), which, None))
    out = np.zeros(max(n, 1), np.uint32)
    lib().orc_machine_events(C.byref(mi), which, _p(out))
    return out[:n]


def machine_stage_perm(t: dict, chip: int, gamma, beta):
    """(permutation trace [perm_width][H], cumulative sum) of one chip for the given LogUp challenges."""
    mi, _keep = machine_input(t)
    d = machine_chip(chip)
    h = 1 << machine_heights(t)[chip]
    pw = d["perm_width"]
    perm, cum = np.zeros((pw, h), np.uint32), np.zeros(4, np.uint32)
    lib().orc_machine_stage_perm(C.byref(mi), chip, _p(_u32(gamma)), _p(_u32(beta)), _p(perm), _p(cum))
    return perm, [int(x) for x in cum]


def machine_stage_quotient(t: dict, chip: int, alpha, gamma, beta) -> np.ndarray:
    """Quotient values [8][H] of one chip for the given challenges."""
    mi, _keep = machine_input(t)
    h = 1 << machine_heights(t)[chip]
    quot = np.zeros((8, h), np.uint32)
    lib().orc_machine_stage_quotient(C.byref(mi), chip, _p(_u32(alpha)), _p(_u32(gamma)), _p(_u32(beta)), _p(quot))
    return quot


def machine_agg_public(leaves):
    """(Merkle root, digest of the leaf list) of an aggregation payload [n][8]."""
    lv = np.ascontiguousarray(leaves, dtype=np.uint32).reshape(-1, 8)
    root, dg = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    lib().orc_machine_agg_public(_p(lv), C.c_size_t(len(lv)), _p(root), _p(dg))
    return [int(x) for x in root], [int(x) for x in dg]


def machine_nodes_public(keys, digests):
    """(root, digest of the list) for digests supplied at heap keys, e.g. a leaf and the siblings along its Merkle path;
    raises ValueError if the set is malformed."""
    dv = np.ascontiguousarray(digests, dtype=np.uint32).reshape(-1, 8)
    kv = np.ascontiguousarray(keys, dtype=np.uint32)
    root, dg = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    if not lib().orc_machine_nodes_public(_p(kv), _p(dv), C.c_size_t(len(dv)), _p(root), _p(dg)):
        raise ValueError("malformed set of supplied nodes")
    return [int(x) for x in root], [int(x) for x in dg]


def merkle_path_nodes(index: int, leaf, siblings):
    """Heap keys and digests of a Merkle path: the leaf at key 2^d + index, sibling j at key ((2^d + index) >> j) ^ 1."""
    d = len(siblings)
    k0 = (1 << d) + index
    keys = [k0] + [(k0 >> j) ^ 1 for j in range(d)]
    return np.array(keys, np.uint32), np.array([leaf] + list(siblings), np.uint32).reshape(-1, 8)


def machine_setup(t: dict):
    """(preprocessed commitment root, verifying-key digest) as lists of 8 canonical words."""
    mi, _keep = machine_input(t)
    root, vk = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    lib().orc_machine_setup(C.byref(mi), int(t["info"].keccak_mode), _p(root), _p(vk))
    return [int(x) for x in root], [int(x) for x in vk]


def machine_prove(t: dict, *, num_queries: int = 100, pow_bits: int = 16) -> bytes:
    mi, _keep = machine_input(t)
    info = t["info"]
    pv = t["public_values"]
    pub = MachinePublic(info.exit_code, len(pv), info.pv_digest, info.deferred_digest)
    cfg = Config(num_queries, pow_bits)
    logh = (C.c_int * N_CHIPS)(*machine_heights(t))
    cap = int(lib().orc_machine_proof_size(logh, info.log_prog, info.log_image, C.byref(cfg), C.c_uint32(len(pv))))
    buf = (C.c_uint8 * cap)()
    n = C.c_size_t(0)
    pvb = (C.c_uint8 * max(1, len(pv))).from_buffer_copy(pv or b"\0")
    rc = lib().orc_machine_prove(C.byref(mi), int(info.keccak_mode), C.byref(pub), pvb, C.byref(cfg), buf, C.c_size_t(cap),
                                 C.byref(n))
    if rc != 0:
        raise RuntimeError(f"orc_machine_prove failed rc={rc}")
    return bytes(buf[: n.value])


# ---------------------------------------------------------------------------
# verify_merkle_proof restated (reference crypto-ops/src/lib.rs:8-23):
# hash every node, require the root node to hash to root_hash, walk the key.
# Raises on the same three conditions the reference panics on.
# ---------------------------------------------------------------------------
def _keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    st = np.zeros(25, np.uint64)
    for off in range(0, len(msg), rate):
        blk = np.frombuffer(bytes(msg[off:off + rate]), dtype="<u8")
        st[:17] ^= blk
        st = keccak_f(st)
    return st[:4].astype("<u8").tobytes()


def _rlp_decode(buf: bytes, off: int = 0):
    b0 = buf[off]
    if b0 < 0x80:
        return buf[off:off + 1], off + 1
    if b0 < 0xB8:
        n = b0 - 0x80
        return buf[off + 1:off + 1 + n], off + 1 + n
    if b0 < 0xC0:
        ll = b0 - 0xB7
        n = int.from_bytes(buf[off + 1:off + 1 + ll], "big")
        return buf[off + 1 + ll:off + 1 + ll + n], off + 1 + ll + n
    if b0 < 0xF8:
        n, start = b0 - 0xC0, off + 1
    else:
        ll = b0 - 0xF7
        n = int.from_bytes(buf[off + 1:off + 1 + ll], "big")
        start = off + 1 + ll
    items, p = [], start
    while p < start + n:
        it, p2 = _rlp_decode(buf, p)
        # keep raw encoding for embedded (inline) nodes
        items.append((it, buf[p:p2]))
        p = p2
    return items, start + n


def verify_merkle_proof(root_hash: bytes, proof: Sequence[bytes], key: bytes) -> bytes:
    db = {_keccak256(n): n for n in proof}
    if root_hash not in db:
        raise ValueError("Invalid merkle proof")  # EthTrie::from / root_hash assert
    nib = []
    for b in key:
        nib += [b >> 4, b & 15]
    node = db[root_hash]
    pos = 0
    while True:
        items, _ = _rlp_decode(node)
        if not isinstance(items, list):
            raise ValueError("Failed to verify Merkle Proof: InvalidProof")
        if len(items) == 17:
            if pos == len(nib):
                val = items[16][0]
                if not val:
                    raise KeyError("Key does not exist!")
                return bytes(val)
            child, raw = items[nib[pos]]
            pos += 1
        elif len(items) == 2:
            path = items[0][0]
            flag = path[0] >> 4
            pn = ([path[0] & 15] if flag & 1 else []) + [x for b in path[1:] for x in (b >> 4, b & 15)]
            if nib[pos:pos + len(pn)] != pn:
                raise KeyError("Key does not exist!")
            pos += len(pn)
            if flag & 2:
                if pos != len(nib):
                    raise KeyError("Key does not exist!")
                return bytes(items[1][0])
            child, raw = items[1]
        else:
            raise ValueError("Failed to verify Merkle Proof: InvalidProof")
        if isinstance(child, list):
            node = raw  # inline node
        elif len(child) == 32:
            if bytes(child) not in db:
                raise ValueError("Failed to verify Merkle Proof: InvalidProof")
            node = db[bytes(child)]
        elif len(child) == 0:
            raise KeyError("Key does not exist!")
        else:
            raise ValueError("Failed to verify Merkle Proof: InvalidProof")
