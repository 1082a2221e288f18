"""Host-side mirror of the reference's SP1 flow over the C-ABI (``include/zksp.h``).

The class and method names follow what the reference's tests call
(reference prover/src/bin/main.rs:59-87)::

    client = ProverClient()                 # ProverClient::new()           :61
    stdin = SP1Stdin(); stdin.write(buf)    # SP1Stdin::new() / write()     :62, :69
    pk, vk = client.setup(MERKLE_ELF)       # client.setup(ELF)             :70
    proof = client.prove(pk, stdin).run()   # client.prove(&pk, stdin).run():71-74
    proof.public_values                     # proof.public_values.to_vec()  :75
    client.verify(proof, vk)                # client.verify(&proof, &vk)    :80

Errors follow the reference's behaviour: a guest panic (the ``expect`` sites of
crypto-ops/src/lib.rs:14-22) surfaces as ``GuestPanic`` from ``run()``, a
rejected proof as ``VerificationError``.  There is no CPU proving fallback: if
the HIP library or a GPU is missing, construction/prove fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKSP_COMPONENT=1: the library built with the round-1 keccak-chip component path (include/zksp_component.h, build.py); the
# default library has no such path, and PROOF_KECCAK_CHIP clients cannot be created on it
COMPONENT = os.environ.get("ZKSP_COMPONENT", "") == "1"
_LIB_PATH = os.path.join(_HERE, "libzksp_component.so" if COMPONENT else "libzksp.so")

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_ELF, ERR_EXECUTOR, ERR_GUEST_PANIC, ERR_PROOF_FORMAT, ERR_VERIFY, \
    ERR_UNSUPPORTED = range(1, 10)
KECCAK_SOFTWARE, KECCAK_OBSERVE, KECCAK_REPLACE = 0, 1, 2
PROOF_MACHINE, PROOF_KECCAK_CHIP = 1, 2
# machine proof (format version 16): chips in proof order and the fixed header in front of the public values
MACHINE_VERSION = 16
MACHINE_CHIP_NAMES = ("cpu", "keccak", "keccak-mem", "mem-final", "image", "program", "mul", "table", "cpu2", "alu", "alu2",
                      "subword", "subword2", "bitwise", "bitwise2", "poseidon2", "ecall", "cpu3", "cpu4", "cpu5", "cpu6", "cpu7",
                      "cpu8", "query", "divider", "transcript", "hint")
MACHINE_CHIPS = len(MACHINE_CHIP_NAMES)
MACHINE_CPU_INSTANCES = 8  # cpu, cpu2 .. cpu8: one AIR, consecutive stretches of the run
# magic, version, heights, exit code, pv length, three digests, the pcs at which the later CPU instances start, the
# aggregation payload's leaf count, root and leaf-list digest, the count and digest of the public bus tuples
MACHINE_HEADER_WORDS = 2 + MACHINE_CHIPS + 2 + 24 + (MACHINE_CPU_INSTANCES - 1) + 17 + 9
PUB_TUPLE_WORDS = 16  # bus, verifier sends (1) / receives (0), multiplicity, number of elements, 12 element slots
P2_REC_WORDS, QR_REC_WORDS, TR_REC_WORDS = 32, 132, 32


def merkle_path_nodes(index: int, leaf, siblings):
    """Heap keys and digests of a Merkle path: the leaf at key 2^d + index, sibling j (from the leaf's level up) at key
    ((2^d + index) >> j) ^ 1 - what ``zksp_stdin_set_aggregation_keyed`` / ``zksp_verify_aggregate_keyed`` take."""
    import numpy as np
    sib = np.ascontiguousarray(siblings, dtype=np.uint32).reshape(-1, 8)
    d = len(sib)
    if not 0 <= index < (1 << d):
        raise ValueError("index outside the tree")
    k0 = (1 << d) + index
    keys = np.array([k0] + [(k0 >> j) ^ 1 for j in range(d)], np.uint32)
    digests = np.ascontiguousarray(np.vstack([np.asarray(leaf, np.uint32).reshape(1, 8), sib]), dtype=np.uint32)
    return keys, digests


class ZkspError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"zksp error {code}: {msg}")
        self.code = code


class GuestPanic(ZkspError):
    """The guest exited non-zero (reference: panic inside verify_merkle_proof)."""


class VerificationError(ZkspError):
    pass


class Options(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("keccak_mode", C.c_int32), ("num_queries", C.c_uint32),
                ("pow_bits", C.c_uint32), ("max_batch", C.c_uint32), ("proof_mode", C.c_int32)]


class ExecReport(C.Structure):
    _fields_ = [("cycles", C.c_uint64), ("memory_ops", C.c_uint64), ("exit_code", C.c_uint32),
                ("n_keccak", C.c_uint32), ("pv_len", C.c_uint32), ("pv_digest", C.c_uint32 * 8),
                ("syscalls", C.c_uint64 * 6), ("opcode_hist", C.c_uint64 * 64)]


class MTraceInfo(C.Structure):
    _fields_ = [("cycles", C.c_uint64), ("memory_ops", C.c_uint64), ("exit_code", C.c_uint32), ("entry", C.c_uint32),
                ("log_prog", C.c_uint32), ("log_image", C.c_uint32), ("keccak_mode", C.c_uint32),
                ("pv_digest", C.c_uint32 * 8), ("deferred_digest", C.c_uint32 * 8), ("uninit_reads", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("trace_width", C.c_uint32), ("num_constraints", C.c_uint32), ("num_queries", C.c_uint32),
                ("pow_bits", C.c_uint32), ("max_batch", C.c_uint32)]


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Loads the in-tree HIP library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the proving path)")
    lib = C.CDLL(_LIB_PATH)
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
    lib.zksp_client_new.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    lib.zksp_client_free.argtypes = [vp]
    lib.zksp_client_free.restype = None
    lib.zksp_last_error.argtypes = [vp]
    lib.zksp_last_error.restype = C.c_char_p
    lib.zksp_setup.argtypes = [vp, C.c_char_p, sz, C.POINTER(vp), C.POINTER(vp)]
    lib.zksp_pk_free.argtypes = [vp]
    lib.zksp_pk_free.restype = None
    lib.zksp_vk_free.argtypes = [vp]
    lib.zksp_vk_free.restype = None
    lib.zksp_vk_digest.argtypes = [vp, C.POINTER(u8p), C.POINTER(sz)]
    lib.zksp_stdin_new.restype = vp
    lib.zksp_stdin_write.argtypes = [vp, C.c_char_p, sz]
    lib.zksp_stdin_free.argtypes = [vp]
    lib.zksp_stdin_free.restype = None
    lib.zksp_prove.argtypes = [vp, vp, vp, C.POINTER(vp)]
    lib.zksp_prove_batch.argtypes = [vp, vp, C.POINTER(vp), sz, C.POINTER(vp), C.POINTER(C.c_int32)]
    lib.zksp_proof_public_values.argtypes = [vp, C.POINTER(u8p), C.POINTER(sz)]
    lib.zksp_proof_serialize.argtypes = [vp, C.POINTER(u8p), C.POINTER(sz)]
    lib.zksp_proof_deserialize.argtypes = [C.c_char_p, sz, C.POINTER(vp)]
    lib.zksp_proof_free.argtypes = [vp]
    lib.zksp_proof_free.restype = None
    lib.zksp_verify.argtypes = [vp, vp, vp]
    lib.zksp_execute.argtypes = [vp, vp, vp, C.c_int, C.POINTER(ExecReport), C.c_void_p, sz, C.c_void_p, sz]
    lib.zksp_execute_keccak.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz)]
    lib.zksp_opcode_name.argtypes = [C.c_int]
    lib.zksp_opcode_name.restype = C.c_char_p
    lib.zksp_machine_trace.argtypes = [vp, vp, vp, C.POINTER(vp)]
    lib.zksp_mtrace_free.argtypes = [vp]
    lib.zksp_mtrace_free.restype = None
    lib.zksp_mtrace_section.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    lib.zksp_mtrace_info.argtypes = [vp, C.POINTER(MTraceInfo)]
    lib.zksp_vk_machine.argtypes = [vp, vp, vp]
    lib.zksp_mtrace_heights.argtypes = [vp, vp]
    lib.zksp_machine_body_words.argtypes = [vp, vp]
    lib.zksp_machine_body_words.restype = sz
    lib.zksp_machine_chip_widths.argtypes = [C.c_int, vp]
    lib.zksp_machine_chip_widths.restype = C.c_char_p
    lib.zksp_hip_machine_load.argtypes = [vp, vp, C.POINTER(vp), sz]
    lib.zksp_hip_machine_prove.argtypes = [vp]
    lib.zksp_hip_release_workspace.argtypes = [vp]
    lib.zksp_hip_machine_fetch_bodies.argtypes = [vp, vp, sz]
    lib.zksp_hip_machine_fetch_roots.argtypes = [vp, vp, sz]
    lib.zksp_machine_proof_from_body.argtypes = [vp, vp, vp, vp, sz, C.POINTER(vp)]
    lib.zksp_machine_cover_heights.argtypes = [vp, sz, vp]
    lib.zksp_stdin_set_aggregation.argtypes = [vp, vp, sz]
    lib.zksp_proof_aggregation.argtypes = [vp, vp, vp]
    lib.zksp_verify_aggregate.argtypes = [vp, vp, vp, vp, sz]
    lib.zksp_stdin_set_aggregation_keyed.argtypes = [vp, vp, vp, sz]
    lib.zksp_verify_aggregate_keyed.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.zksp_stdin_set_verified_leaf.argtypes = [vp, vp, vp, vp]
    lib.zksp_stdin_add_verified_leaf.argtypes = [vp, vp, vp, vp]
    lib.zksp_leaves_public.argtypes = [vp, vp, vp, sz, vp, sz, C.POINTER(sz)]
    lib.zksp_verify_with_leaves.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.zksp_leaf_public.argtypes = [vp, vp, vp, vp, sz, C.POINTER(sz)]
    lib.zksp_leaf_public_at.argtypes = [vp, vp, vp, C.c_uint32, vp, sz, vp, sz, C.POINTER(sz)]
    lib.zksp_stdin_public_tuples.argtypes = [vp, vp, sz, C.POINTER(sz)]
    lib.zksp_stdin_defer_verified_leaves.argtypes = [vp, vp, vp, vp, vp, vp, sz]
    lib.zksp_zeta_program_selftest.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.zksp_stdin_add_verified_node.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.zksp_proof_stub.argtypes = [vp, C.POINTER(vp)]
    lib.zksp_stdin_add_verified_leaves.argtypes = [vp, vp, vp, vp, vp, vp, sz]
    lib.zksp_verify_public.argtypes = [vp, vp, vp, vp, sz]
    lib.zksp_verify_with_leaf.argtypes = [vp, vp, vp, vp, vp]
    lib.zksp_proof_public_tuples.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.zksp_hip_machine_fetch_stage.argtypes = [vp, C.c_int, C.c_int, sz, vp, sz]
    lib.zksp_hip_machine_fetch_challenges.argtypes = [vp, sz, vp]
    lib.zksp_get_params.argtypes = [vp, C.POINTER(Params)]
    if COMPONENT:
        lib.zksp_proof_body_words.argtypes = [vp, C.c_int]
        lib.zksp_proof_body_words.restype = sz
        lib.zksp_hip_load_batch.argtypes = [vp, C.c_int, sz, sz, vp, vp, vp]
        lib.zksp_hip_prove_resident.argtypes = [vp]
        lib.zksp_hip_fetch_bodies.argtypes = [vp, vp, sz]
        lib.zksp_hip_fetch_roots.argtypes = [vp, vp, sz]
        lib.zksp_proof_from_body.argtypes = [vp, sz, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_char_p, sz, vp, vp, vp,
                                             C.POINTER(vp)]
        lib.zksp_hip_keccak_trace.argtypes = [vp, vp, C.c_uint32, C.c_int, vp]
        lib.zksp_hip_keccak_quotient.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        lib.zksp_hip_bus_perm_trace.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    lib.zksp_hip_sync.argtypes = [vp]
    lib.zksp_hip_timer_start.argtypes = [vp]
    lib.zksp_hip_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.zksp_hip_profile_enable.argtypes = [vp, C.c_int]
    lib.zksp_hip_profile_reset.argtypes = [vp]
    lib.zksp_hip_profile_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.zksp_dev_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.zksp_dev_free.argtypes = [vp, vp]
    lib.zksp_dev_upload.argtypes = [vp, vp, vp, sz]
    lib.zksp_dev_download.argtypes = [vp, vp, vp, sz]
    lib.zksp_dev_memset.argtypes = [vp, vp, C.c_int, sz]
    lib.zksp_hip_lde.argtypes = [vp, vp, C.c_int, sz, C.c_uint32, vp, vp]
    lib.zksp_hip_merkle_commit.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.zksp_hip_poseidon2_permute.argtypes = [vp, vp, sz]
    lib.zksp_host_poseidon2_permute.argtypes = [vp, sz, C.c_int]
    lib.zksp_hip_fri_fold.argtypes = [vp, vp, C.c_int, C.c_uint32, vp, vp]
    lib.zksp_hip_microbench.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    _lib = lib
    return lib


# every symbol include/zksp.h declares (checked by tests/test_abi.py) ...
ABI_SYMBOLS = [
    "zksp_client_new", "zksp_client_free", "zksp_last_error", "zksp_setup", "zksp_pk_free", "zksp_vk_free",
    "zksp_vk_digest", "zksp_stdin_new", "zksp_stdin_write", "zksp_stdin_free", "zksp_prove", "zksp_prove_batch",
    "zksp_proof_public_values", "zksp_proof_serialize", "zksp_proof_deserialize", "zksp_proof_free", "zksp_verify",
    "zksp_execute", "zksp_execute_keccak", "zksp_opcode_name", "zksp_machine_trace", "zksp_mtrace_free",
    "zksp_mtrace_section", "zksp_mtrace_info", "zksp_vk_machine", "zksp_mtrace_heights", "zksp_machine_body_words",
    "zksp_machine_chip_widths", "zksp_machine_cover_heights", "zksp_stdin_set_aggregation", "zksp_proof_aggregation", "zksp_verify_aggregate",
    "zksp_stdin_set_aggregation_keyed", "zksp_verify_aggregate_keyed", "zksp_stdin_set_verified_leaf", "zksp_stdin_add_verified_leaf", "zksp_leaves_public", "zksp_verify_with_leaves", "zksp_leaf_public", "zksp_verify_public", "zksp_leaf_public_at", "zksp_stdin_public_tuples", "zksp_stdin_defer_verified_leaves", "zksp_zeta_program_selftest", "zksp_stdin_add_verified_node", "zksp_proof_stub", "zksp_stdin_add_verified_leaves",
    "zksp_verify_with_leaf", "zksp_proof_public_tuples", "zksp_hip_machine_fetch_stage", "zksp_hip_machine_fetch_challenges",
    "zksp_hip_machine_load", "zksp_hip_machine_prove", "zksp_hip_release_workspace", "zksp_hip_machine_fetch_bodies", "zksp_hip_machine_fetch_roots", "zksp_machine_proof_from_body", "zksp_get_params",
    "zksp_hip_sync", "zksp_hip_timer_start", "zksp_hip_timer_stop",
    "zksp_hip_profile_enable", "zksp_hip_profile_read", "zksp_hip_profile_reset", "zksp_dev_malloc", "zksp_dev_free",
    "zksp_dev_upload", "zksp_dev_download", "zksp_dev_memset", "zksp_hip_lde", "zksp_hip_merkle_commit",
    "zksp_hip_poseidon2_permute", "zksp_host_poseidon2_permute", "zksp_hip_fri_fold",
    "zksp_hip_microbench",
]
# ... and include/zksp_component.h (only in the library built with ZKSP_COMPONENT=1)
COMPONENT_ABI_SYMBOLS = [
    "zksp_proof_body_words", "zksp_hip_load_batch", "zksp_hip_prove_resident", "zksp_proof_from_body", "zksp_hip_fetch_bodies",
    "zksp_hip_fetch_roots", "zksp_hip_keccak_trace", "zksp_hip_keccak_quotient", "zksp_hip_bus_perm_trace",
]


class SP1Stdin:
    """``SP1Stdin`` (reference prover/src/bin/main.rs:62, :69)."""

    def __init__(self):
        self._lib = load_library()
        self._h = C.c_void_p(self._lib.zksp_stdin_new())
        if not self._h:
            raise MemoryError("zksp_stdin_new")

    def write(self, buf: bytes) -> None:
        rc = self._lib.zksp_stdin_write(self._h, bytes(buf), len(buf))
        if rc:
            raise ZkspError(rc, "stdin.write")

    def set_aggregation(self, leaves) -> None:
        """Aggregation payload (``zksp_stdin_set_aggregation``): the proof made from this stdin also establishes the
        Poseidon2 Merkle root of ``leaves`` ([n][8] canonical field words, n a power of two >= 2)."""
        import numpy as np
        lv = np.ascontiguousarray(leaves, dtype=np.uint32).reshape(-1, 8)
        rc = self._lib.zksp_stdin_set_aggregation(self._h, lv.ctypes.data_as(C.c_void_p), len(lv))
        if rc:
            raise ZkspError(rc, "stdin_set_aggregation")

    def set_merkle_path(self, index: int, leaf, siblings) -> None:
        """Merkle-path payload (``zksp_stdin_set_aggregation_keyed``): the proof made from this stdin also establishes that
        ``leaf`` (8 canonical words), hashed up along ``siblings`` ([d][8], from the leaf's level to just below the root)
        at position ``index`` of a depth-d Poseidon2 Merkle tree, gives the root ``proof.aggregation`` reports."""
        keys, digests = merkle_path_nodes(index, leaf, siblings)
        rc = self._lib.zksp_stdin_set_aggregation_keyed(self._h, keys.ctypes.data_as(C.c_void_p), digests.ctypes.data_as(C.c_void_p),
                                                        len(keys))
        if rc:
            raise ZkspError(rc, "stdin_set_aggregation_keyed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.zksp_stdin_free(self._h)
            self._h = None


class _Handle:
    def __init__(self, lib, h, free):
        self._lib, self._h, self._free = lib, h, free

    def __del__(self):
        if getattr(self, "_h", None):
            self._free(self._h)
            self._h = None


class ProvingKey(_Handle):
    pass


class VerifyingKey(_Handle):
    @property
    def digest(self) -> bytes:
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._lib.zksp_vk_digest(self._h, C.byref(p), C.byref(n))
        return bytes(p[: n.value])


    @property
    def machine(self):
        """(preprocessed-table commitment root, machine verifying-key digest): 8 canonical words each."""
        r, d = (C.c_uint32 * 8)(), (C.c_uint32 * 8)()
        self._lib.zksp_vk_machine(self._h, r, d)
        return list(r), list(d)


class SP1ProofWithPublicValues(_Handle):
    @property
    def public_values(self) -> bytes:
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._lib.zksp_proof_public_values(self._h, C.byref(p), C.byref(n))
        if rc:
            raise ZkspError(rc, "public_values")
        return bytes(p[: n.value])

    @property
    def aggregation(self):
        """(number of leaves, proven Merkle root as 8 canonical words) of the aggregation payload; (0, zeros) if none."""
        n, root = C.c_uint32(), (C.c_uint32 * 8)()
        rc = self._lib.zksp_proof_aggregation(self._h, C.byref(n), root)
        if rc:
            raise ZkspError(rc, "proof_aggregation")
        return int(n.value), [int(x) for x in root]

    @property
    def public_tuples(self):
        """(number of public bus tuples the proof's buses close with, the sponge digest of their list as 8 canonical words);
        (0, zeros) for a proof without a leaf-proof check."""
        n, dg = C.c_uint32(), (C.c_uint32 * 8)()
        rc = self._lib.zksp_proof_public_tuples(self._h, C.byref(n), dg)
        if rc:
            raise ZkspError(rc, "proof_public_tuples")
        return int(n.value), [int(x) for x in dg]

    def stub(self) -> "SP1ProofWithPublicValues":
        """The proof without its query phase (``zksp_proof_stub``): what deriving a leaf's statement needs of it."""
        h = C.c_void_p()
        rc = self._lib.zksp_proof_stub(self._h, C.byref(h))
        if rc:
            raise ZkspError(rc, "proof_stub")
        return SP1ProofWithPublicValues(self._lib, h, self._lib.zksp_proof_free)

    def to_bytes(self) -> bytes:
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        rc = self._lib.zksp_proof_serialize(self._h, C.byref(p), C.byref(n))
        if rc:
            raise ZkspError(rc, "serialize")
        return C.string_at(p, n.value)

    @staticmethod
    def from_bytes(buf: bytes) -> "SP1ProofWithPublicValues":
        lib = load_library()
        h = C.c_void_p()
        rc = lib.zksp_proof_deserialize(bytes(buf), len(buf), C.byref(h))
        if rc:
            raise ZkspError(rc, "malformed proof bytes")
        return SP1ProofWithPublicValues(lib, h, lib.zksp_proof_free)


def proof_from_body(body, log_h: int, states, exit_code: int, public_values: bytes, pv_digest, deferred_digest,
                    vk_digest) -> SP1ProofWithPublicValues:
    """Complete proof object from one body fetched off the resident path (``zksp_hip_fetch_bodies``)
    and the inputs that batch was loaded with; see ``zksp_proof_from_body``."""
    import numpy as np
    lib = load_library()
    body = np.ascontiguousarray(body, dtype=np.uint32)
    st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 25)
    pvd = np.ascontiguousarray(pv_digest, dtype=np.uint32)
    dfd = np.ascontiguousarray(deferred_digest, dtype=np.uint32)
    vkd = np.ascontiguousarray(vk_digest, dtype=np.uint32)
    h = C.c_void_p()
    rc = lib.zksp_proof_from_body(body.ctypes.data_as(C.c_void_p), body.size, log_h, st.ctypes.data_as(C.c_void_p),
                                  st.shape[0], exit_code, bytes(public_values), len(public_values),
                                  pvd.ctypes.data_as(C.c_void_p), dfd.ctypes.data_as(C.c_void_p),
                                  vkd.ctypes.data_as(C.c_void_p), C.byref(h))
    if rc:
        raise ZkspError(rc, "proof_from_body")
    return SP1ProofWithPublicValues(lib, h, lib.zksp_proof_free)


def machine_chip_widths():
    """[(name, preprocessed, main, permutation widths)] of the machine proof's chips, in proof order."""
    lib = load_library()
    out = []
    for c in range(MACHINE_CHIPS):
        w = (C.c_int32 * 3)()
        name = lib.zksp_machine_chip_widths(c, w)
        out.append((name.decode(), int(w[0]), int(w[1]), int(w[2])))
    return out


class MachineTraceHandle(_Handle):
    """A traced guest run kept on the C side (``zksp_mtrace``)."""

    def heights(self):
        lh = (C.c_int32 * MACHINE_CHIPS)()
        self._lib.zksp_mtrace_heights(self._h, lh)
        return list(lh)

    def proof_from_body(self, pk: "ProvingKey", body, heights=None) -> "SP1ProofWithPublicValues":
        """The proof object of this run from a fetched body; ``heights``: the shape the batch was proven with
        (``machine_cover_heights`` of the loaded traces), None = this run's own minimal heights."""
        import numpy as np
        body = np.ascontiguousarray(body, dtype=np.uint32)
        h = C.c_void_p()
        lh = (C.c_int32 * MACHINE_CHIPS)(*heights) if heights is not None else None
        rc = self._lib.zksp_machine_proof_from_body(pk._h, self._h, lh, body.ctypes.data_as(C.c_void_p), body.size, C.byref(h))
        if rc:
            raise ZkspError(rc, "machine_proof_from_body")
        return SP1ProofWithPublicValues(self._lib, h, self._lib.zksp_proof_free)


def machine_cover_heights(traces):
    """The shape (chip log-heights) a batch of these traced runs (MachineTraceHandle) is proven with."""
    n = len(traces)
    arr = (C.c_void_p * n)(*[t._h for t in traces])
    lh = (C.c_int32 * MACHINE_CHIPS)()
    rc = traces[0]._lib.zksp_machine_cover_heights(arr, n, lh)
    if rc:
        raise ZkspError(rc, "machine_cover_heights")
    return list(lh)


class _ProveBuilder:
    """What ``client.prove(&pk, stdin)`` returns; ``.run()`` does the work."""

    def __init__(self, client: "ProverClient", pk: ProvingKey, stdin: SP1Stdin):
        self._c, self._pk, self._stdin = client, pk, stdin

    def run(self) -> SP1ProofWithPublicValues:
        lib = self._c._lib
        h = C.c_void_p()
        rc = lib.zksp_prove(self._c._h, self._pk._h, self._stdin._h, C.byref(h))
        if rc == ERR_GUEST_PANIC:
            raise GuestPanic(rc, self._c.last_error())
        if rc:
            raise ZkspError(rc, self._c.last_error())
        return SP1ProofWithPublicValues(lib, h, lib.zksp_proof_free)


class ProverClient:
    """``ProverClient`` (reference prover/src/bin/main.rs:61).

    ``device`` is the HIP device ordinal; ``device=-1`` builds a verifier/executor-only
    client that never touches a GPU (and cannot prove).  Environment override:
    ``ZKSP_DEVICE``.
    """

    def __init__(self, device: Optional[int] = None, *, keccak_mode: int = KECCAK_REPLACE, num_queries: int = 100,
                 pow_bits: int = 16, max_batch: int = 192, proof_mode: int = PROOF_MACHINE):
        self._lib = load_library()
        if device is None:
            device = int(os.environ.get("ZKSP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        opts = Options(device, keccak_mode, num_queries, pow_bits, max_batch, proof_mode)
        self._h = C.c_void_p()
        rc = self._lib.zksp_client_new(C.byref(opts), C.byref(self._h))
        if rc:
            raise ZkspError(rc, "client_new failed" + (": no usable GPU (the proving path has no CPU fallback)"
                                                       if rc == ERR_NO_DEVICE else ""))
        self.device = device

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.zksp_client_free(self._h)
            self._h = None

    def last_error(self) -> str:
        return (self._lib.zksp_last_error(self._h) or b"").decode("utf-8", "replace")

    def params(self) -> Params:
        p = Params()
        self._lib.zksp_get_params(self._h, C.byref(p))
        return p

    def setup(self, elf: bytes) -> Tuple[ProvingKey, VerifyingKey]:
        pk, vk = C.c_void_p(), C.c_void_p()
        rc = self._lib.zksp_setup(self._h, bytes(elf), len(elf), C.byref(pk), C.byref(vk))
        if rc:
            raise ZkspError(rc, self.last_error())
        return ProvingKey(self._lib, pk, self._lib.zksp_pk_free), VerifyingKey(self._lib, vk, self._lib.zksp_vk_free)

    def prove(self, pk: ProvingKey, stdin: SP1Stdin) -> _ProveBuilder:
        return _ProveBuilder(self, pk, stdin)

    def prove_batch(self, pk: ProvingKey, stdins: Sequence[SP1Stdin]):
        """Independent proofs in lockstep; returns (proofs, status codes)."""
        n = len(stdins)
        arr = (C.c_void_p * n)(*[s._h for s in stdins])
        out = (C.c_void_p * n)()
        st = (C.c_int32 * n)()
        rc = self._lib.zksp_prove_batch(self._h, pk._h, arr, n, out, st)
        if rc:
            raise ZkspError(rc, self.last_error())
        proofs: List[Optional[SP1ProofWithPublicValues]] = []
        for i in range(n):
            proofs.append(SP1ProofWithPublicValues(self._lib, C.c_void_p(out[i]), self._lib.zksp_proof_free)
                          if out[i] else None)
        return proofs, list(st)

    def verify(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey) -> None:
        rc = self._lib.zksp_verify(self._h, proof._h, vk._h)
        if rc:
            raise VerificationError(rc, self.last_error())

    def verify_aggregate(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, leaves) -> None:
        """``verify`` for a proof with an aggregation payload: additionally, ``proof.aggregation``'s root is the Poseidon2
        Merkle root of exactly these leaves ([n][8] canonical words)."""
        import numpy as np
        lv = np.ascontiguousarray(leaves, dtype=np.uint32).reshape(-1, 8)
        rc = self._lib.zksp_verify_aggregate(self._h, proof._h, vk._h, lv.ctypes.data_as(C.c_void_p), len(lv))
        if rc:
            raise VerificationError(rc, self.last_error())

    def verify_merkle_path(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, index: int, leaf, siblings) -> None:
        """``verify`` for a proof with a Merkle-path payload: additionally, ``proof.aggregation``'s root is what this leaf
        gives when hashed up along these siblings at this position."""
        keys, digests = merkle_path_nodes(index, leaf, siblings)
        rc = self._lib.zksp_verify_aggregate_keyed(self._h, proof._h, vk._h, keys.ctypes.data_as(C.c_void_p),
                                                   digests.ctypes.data_as(C.c_void_p), len(keys))
        if rc:
            raise VerificationError(rc, self.last_error())

    def set_verified_leaf(self, stdin: SP1Stdin, leaf: SP1ProofWithPublicValues, leaf_vk: VerifyingKey) -> None:
        """Leaf-proof check (``zksp_stdin_set_verified_leaf``, SURVEY.md row f4 stage 2b): the proof made from ``stdin`` also
        establishes that the query phase of ``leaf`` verifies under the challenges its own transcript yields - every Merkle opening of its four commitment rounds and of
        its FRI layers, and the folding chain down to the final constant.  Raises VerificationError if ``leaf`` does not
        verify under this client's parameters."""
        rc = self._lib.zksp_stdin_set_verified_leaf(self._h, stdin._h, leaf._h, leaf_vk._h)
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())

    def clear_verified_leaves(self, stdin: SP1Stdin) -> None:
        """Removes every leaf-proof check from ``stdin`` (``zksp_stdin_set_verified_leaf`` with a null leaf)."""
        rc = self._lib.zksp_stdin_set_verified_leaf(self._h, stdin._h, None, None)
        if rc:
            raise ZkspError(rc, self.last_error())

    def add_verified_leaf(self, stdin: SP1Stdin, leaf: SP1ProofWithPublicValues, leaf_vk: VerifyingKey) -> None:
        """One more leaf-proof check beside the run (``zksp_stdin_add_verified_leaf``): the proof made from ``stdin`` establishes
        the query phases of ALL leaves added so far - the node of a recursion tree of that arity.  Its statement is the leaves'
        public tuples one leaf after the other, in the order they were added (``leaves_public``, ``verify_with_leaves``)."""
        rc = self._lib.zksp_stdin_add_verified_leaf(self._h, stdin._h, leaf._h, leaf_vk._h)
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())

    def add_verified_node(self, stdin: SP1Stdin, node: SP1ProofWithPublicValues, node_vk: VerifyingKey, node_statement) -> None:
        """``add_verified_leaf`` for a leaf that itself checks leaves (``zksp_stdin_add_verified_node``): ``node_statement`` is the
        list of public tuples ITS proof was made for ([n][16] canonical words; ``leaves_public`` of its own leaves)."""
        import numpy as np
        tv = np.ascontiguousarray(node_statement, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
        rc = self._lib.zksp_stdin_add_verified_node(self._h, stdin._h, node._h, node_vk._h, tv.ctypes.data_as(C.c_void_p), len(tv))
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())

    def add_verified_leaves(self, stdin: SP1Stdin, leaves, leaf_vks, statements=None) -> None:
        """``add_verified_leaf`` / ``add_verified_node`` for several leaves in one call (``zksp_stdin_add_verified_leaves``): they are
        verified and logged side by side on the host's threads.  ``statements``: None, or per leaf its own statement (None for a
        plain leaf)."""
        import numpy as np
        pa, va, k = self._handle_arrays(leaves, leaf_vks)
        own_p, own_n, keep = None, None, []
        if statements is not None and any(st is not None for st in statements):
            own_p, own_n = (C.c_void_p * k)(), (C.c_size_t * k)()
            for i, st in enumerate(statements):
                if st is not None:
                    tv = np.ascontiguousarray(st, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
                    keep.append(tv)
                    own_p[i], own_n[i] = tv.ctypes.data, len(tv)
        rc = self._lib.zksp_stdin_add_verified_leaves(self._h, stdin._h, pa, va, own_p, own_n, k)
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())

    def defer_verified_leaves(self, stdin: SP1Stdin, leaves, leaf_vks, statements=None) -> None:
        """``add_verified_leaves`` whose checks are made by the ``prove`` / ``prove_batch`` call that consumes ``stdin``
        (``zksp_stdin_defer_verified_leaves``): on its tracing threads, beside the proving of the runs that are ready.  The stdin
        keeps the leaves alive until then; a leaf that does not verify shows as that run's status (ERR_VERIFY).  The statement
        is read with ``stdin_statement`` AFTER proving."""
        import numpy as np
        pa, va, k = self._handle_arrays(leaves, leaf_vks)
        own_p, own_n, keep = None, None, []
        if statements is not None and any(st is not None for st in statements):
            own_p, own_n = (C.c_void_p * k)(), (C.c_size_t * k)()
            for i, st in enumerate(statements):
                if st is not None:
                    tv = np.ascontiguousarray(st, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
                    keep.append(tv)
                    own_p[i], own_n[i] = tv.ctypes.data, len(tv)
        rc = self._lib.zksp_stdin_defer_verified_leaves(self._h, stdin._h, pa, va, own_p, own_n, k)
        if rc:
            raise ZkspError(rc, self.last_error())
        stdin._deferred_keepalive = (list(leaves), list(leaf_vks))  # (the C side holds pointers to them until the prove call)

    def zeta_program_selftest(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, own_statement=None) -> dict:
        """Verifies ``proof`` (or its stub) and cross-checks the recorded constraint-identity program against the native evaluation
        (``zksp_zeta_program_selftest``); returns the program's sizes."""
        import numpy as np
        own = np.zeros((0, PUB_TUPLE_WORDS), np.uint32) if own_statement is None else \
            np.ascontiguousarray(own_statement, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
        info = (C.c_uint32 * 8)()
        rc = self._lib.zksp_zeta_program_selftest(self._h, proof._h, vk._h, own.ctypes.data_as(C.c_void_p) if len(own) else None, len(own), info)
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())
        return {"ops": info[0], "cells": info[1], "inputs": info[2], "constants": info[3], "max_reads_of_a_cell": info[5],
                "inputs_read": info[6]}

    def _tuples(self, call, guess: int = 128):
        """Runs ``call(out_ptr, cap_words, n_ref)`` - a C function that derives a list of public tuples - ONCE where the list
        fits `guess` tuples (a leaf's statement is some 75; deriving it is a stub check of the leaf: milliseconds), twice
        otherwise."""
        import numpy as np
        n = C.c_size_t()
        out = np.zeros((guess, PUB_TUPLE_WORDS), np.uint32)
        rc = call(out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
        if rc and n.value > guess:  # (the count is set before the buffer is found too small)
            out = np.zeros((n.value, PUB_TUPLE_WORDS), np.uint32)
            rc = call(out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
        if rc:
            raise (VerificationError if rc == ERR_VERIFY else ZkspError)(rc, self.last_error())
        return out[:n.value].copy()

    def stdin_statement(self, stdin: SP1Stdin):
        """The public tuples the proof made from ``stdin`` will carry (``zksp_stdin_public_tuples``): what the leaf checks added
        so far state, [n][16] canonical words ([0][16] without leaf checks).  Before proving - proving consumes them."""
        return self._tuples(lambda o, cap, n: self._lib.zksp_stdin_public_tuples(stdin._h, o, cap, n), guess=512)

    def leaf_public_at(self, leaf: SP1ProofWithPublicValues, leaf_vk: VerifyingKey, index: int, own_statement=None):
        """The statement tuples of the leaf at place ``index`` beside one run (``zksp_leaf_public_at``); ``leaf`` may be a stub;
        ``own_statement``: the tuples the leaf's own proof was made for, if it is a node.  Everything of the leaf but its query
        phase is checked on the way."""
        import numpy as np
        own = np.zeros((0, PUB_TUPLE_WORDS), np.uint32) if own_statement is None else \
            np.ascontiguousarray(own_statement, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
        op = own.ctypes.data_as(C.c_void_p) if len(own) else None
        return self._tuples(lambda o, cap, n: self._lib.zksp_leaf_public_at(self._h, leaf._h, leaf_vk._h, index, op, len(own), o, cap, n))

    def tree_statement(self, children, vk: VerifyingKey, workers: Optional[int] = None):
        """The statement of a node of a recursion tree from its children, recursively: ``children`` is a list of
        ``(proof_or_stub, grandchildren)`` with ``grandchildren`` a list of the same form ([] for a leaf of the tree).  All proofs
        under ``vk``.  Stubs suffice everywhere: no query phase is read.  The stub checks of one depth are independent of one
        another: they run on ``workers`` threads (default: the host's cores, at most 16), each with a verifier client of its own
        (same parameters, no GPU) - a tree over 1 024 leaves has 1 364 stubs below its root."""
        import numpy as np
        if not children:
            return np.zeros((0, PUB_TUPLE_WORDS), np.uint32)
        # the nodes by depth (the root's children are depth 0), each with its place among its siblings and its parent
        levels, frontier = [], [(k, proof, grand, None) for k, (proof, grand) in enumerate(children)]
        while frontier:
            levels.append(frontier)
            nxt = []
            for idx, (_k, _proof, grand, _parent) in enumerate(frontier):
                nxt += [(j, p2, g2, idx) for j, (p2, g2) in enumerate(grand)]
            frontier = nxt
        n_nodes = sum(len(lv) for lv in levels)
        if workers is None:
            workers = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        workers = max(1, min(workers, n_nodes))
        prm = self.params()
        clones = [self] if workers == 1 else [ProverClient(device=-1, num_queries=prm.num_queries, pow_bits=prm.pow_bits)
                                              for _ in range(workers)]
        import queue
        from concurrent.futures import ThreadPoolExecutor
        idle: "queue.Queue" = queue.Queue()
        for c in clones:
            idle.put(c)

        def one(args):
            k, proof, own = args
            c = idle.get()
            try:
                return c.leaf_public_at(proof, vk, k, own)
            finally:
                idle.put(c)

        with ThreadPoolExecutor(max_workers=workers) as ex:
            below = None  # the statements of the level below, per node: lists of its children's tuples in sibling order
            for depth in range(len(levels) - 1, -1, -1):
                lv = levels[depth]
                owns = [None] * len(lv)
                if below is not None:
                    for idx in range(len(lv)):
                        parts = below.get(idx)
                        if parts:
                            owns[idx] = np.concatenate([parts[j] for j in sorted(parts)])
                res = list(ex.map(one, [(k, proof, owns[idx]) for idx, (k, proof, _g, _p) in enumerate(lv)]))
                below = {}
                for (k, _proof, _g, parent), st in zip(lv, res):
                    below.setdefault(parent, {})[k] = st
        top = below.get(None, {})
        return np.concatenate([top[j] for j in sorted(top)])

    def verify_tree(self, root: SP1ProofWithPublicValues, vk: VerifyingKey, children) -> None:
        """``verify`` for the root of a recursion tree (``tree_statement`` for the form of ``children``): the root proof is verified
        completely, of every proof below it the header, transcript, bus balance, constraint identity at zeta and proof of work
        (from its stub) - their query phases are what the proofs above them establish."""
        self.verify_public(root, vk, self.tree_statement(children, vk))

    def _handle_arrays(self, leaves, leaf_vks):
        n = len(leaves)
        if n == 0 or len(leaf_vks) != n:
            raise ValueError("as many verifying keys as leaves, and at least one")
        return (C.c_void_p * n)(*[p._h for p in leaves]), (C.c_void_p * n)(*[v._h for v in leaf_vks]), n

    def leaves_public(self, leaves, leaf_vks):
        """The statement of a proof that checks several leaves (``zksp_leaves_public``): numpy [n][16] canonical words."""
        import numpy as np
        pa, va, k = self._handle_arrays(leaves, leaf_vks)
        return self._tuples(lambda o, cap, n: self._lib.zksp_leaves_public(self._h, pa, va, k, o, cap, n), guess=128 * k)

    def verify_with_leaves(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, leaves, leaf_vks) -> None:
        """``verify`` for a proof that checks several leaves: its statement is the one these leaves give, in this order."""
        pa, va, k = self._handle_arrays(leaves, leaf_vks)
        rc = self._lib.zksp_verify_with_leaves(self._h, proof._h, vk._h, pa, va, k)
        if rc:
            raise VerificationError(rc, self.last_error())

    def leaf_public(self, leaf: SP1ProofWithPublicValues, leaf_vk: VerifyingKey):
        """The statement of a leaf-proof check (``zksp_leaf_public``): the public bus tuples, numpy [n][16] canonical words."""
        import numpy as np
        return self._tuples(lambda o, cap, n: self._lib.zksp_leaf_public(self._h, leaf._h, leaf_vk._h, o, cap, n))

    def verify_public(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, tuples) -> None:
        """``verify`` for a proof whose buses close with these public tuples ([n][16] canonical words)."""
        import numpy as np
        tv = np.ascontiguousarray(tuples, dtype=np.uint32).reshape(-1, PUB_TUPLE_WORDS)
        rc = self._lib.zksp_verify_public(self._h, proof._h, vk._h, tv.ctypes.data_as(C.c_void_p), len(tv))
        if rc:
            raise VerificationError(rc, self.last_error())

    def verify_with_leaf(self, proof: SP1ProofWithPublicValues, vk: VerifyingKey, leaf: SP1ProofWithPublicValues,
                         leaf_vk: VerifyingKey) -> None:
        """``verify`` for a proof with a leaf-proof check: additionally, the statement it closes its buses with is the one
        ``leaf`` (which is verified on the way) gives."""
        rc = self._lib.zksp_verify_with_leaf(self._h, proof._h, vk._h, leaf._h, leaf_vk._h)
        if rc:
            raise VerificationError(rc, self.last_error())

    def execute(self, pk: ProvingKey, stdin: SP1Stdin, keccak_mode: int = KECCAK_OBSERVE):
        """Runs the guest only; returns (ExecReport, public_values, stderr_text, rc)."""
        rep = ExecReport()
        pv = (C.c_uint8 * 65536)()
        err = C.create_string_buffer(8192)
        rc = self._lib.zksp_execute(self._h, pk._h, stdin._h, keccak_mode, C.byref(rep), pv, 65536, err, 8192)
        return rep, bytes(pv[: rep.pv_len]), err.value.decode("utf-8", "replace"), rc

    def keccak_states(self, pk: ProvingKey, stdin: SP1Stdin):
        """keccak-f inputs of one guest run as a numpy [n][25] uint64 array."""
        import numpy as np
        n = C.c_size_t()
        rc = self._lib.zksp_execute_keccak(self._h, pk._h, stdin._h, None, 0, C.byref(n))
        if rc:
            raise ZkspError(rc, self.last_error())
        out = np.zeros((n.value, 25), dtype=np.uint64)
        rc = self._lib.zksp_execute_keccak(self._h, pk._h, stdin._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        if rc:
            raise ZkspError(rc, self.last_error())
        return out

    def machine_trace(self, pk: ProvingKey, stdin: SP1Stdin) -> dict:
        """Traced execution (``zksp_machine_trace``) as numpy arrays: ``cycles`` [n][12], ``keccak`` (structured:
        ts, ptr, in[25], pts[50]), ``memfinal`` [n][5], ``muls`` [n][3], ``prog_mult``, ``alu_idx``, ``sub_idx``, ``bw_idx``, ``ecall_idx``,
        ``program`` [n][9] (last row: the padding instruction), ``image`` [n][2], ``public_values`` (bytes), ``info``
        (MTraceInfo)."""
        import numpy as np
        h = C.c_void_p()
        rc = self._lib.zksp_machine_trace(self._h, pk._h, stdin._h, C.byref(h))
        if rc:
            raise ZkspError(rc, self.last_error())
        try:
            def sec(which, dtype, cols=None):
                p, n = C.c_void_p(), C.c_size_t()
                self._lib.zksp_mtrace_section(h, which, C.byref(p), C.byref(n))
                raw = C.string_at(p, n.value) if n.value else b""
                a = np.frombuffer(raw, dtype=dtype).copy()
                return a.reshape(-1, cols) if cols else a
            kdt = np.dtype([("ts", "<u4"), ("ptr", "<u4"), ("in", "<u8", (25,)), ("pts", "<u4", (50,))])
            info = MTraceInfo()
            self._lib.zksp_mtrace_info(h, C.byref(info))
            return {"cycles": sec(0, np.uint32, 12), "keccak": sec(1, kdt), "memfinal": sec(2, np.uint32, 5),
                    "muls": sec(3, np.uint32, 3), "prog_mult": sec(4, np.uint32), "alu_idx": sec(5, np.uint32),
                    "sub_idx": sec(9, np.uint32), "bw_idx": sec(10, np.uint32), "ecall_idx": sec(11, np.uint32), "div_idx": sec(15, np.uint32), "program": sec(6, np.uint32, 9), "image": sec(7, np.uint32, 2),
                    "public_values": bytes(sec(8, np.uint8)), "info": info,
                    "leaf_p2_rows": sec(12, np.uint32, P2_REC_WORDS), "leaf_qr_rows": sec(13, np.uint32, QR_REC_WORDS),
                    "leaf_tr_rows": sec(16, np.uint32, TR_REC_WORDS),
                    "leaf_pub_tuples": sec(14, np.uint32, PUB_TUPLE_WORDS)}
        finally:
            self._lib.zksp_mtrace_free(h)

    def machine_trace_handle(self, pk: ProvingKey, stdin: SP1Stdin) -> "MachineTraceHandle":
        h = C.c_void_p()
        rc = self._lib.zksp_machine_trace(self._h, pk._h, stdin._h, C.byref(h))
        if rc:
            raise ZkspError(rc, self.last_error())
        return MachineTraceHandle(self._lib, h, self._lib.zksp_mtrace_free)

    def release_workspace(self) -> None:
        """Frees the client's device arena and pinned staging buffers (``zksp_hip_release_workspace``)."""
        rc = self._lib.zksp_hip_release_workspace(self._h)
        if rc:
            raise ZkspError(rc, self.last_error())

    def machine_prove_resident(self, pk: ProvingKey, traces):
        """Loads traced runs (MachineTraceHandle), proves them in lockstep on the GPU with one shape
        (``machine_cover_heights(traces)``) and returns the proof bodies as a numpy array [n][body_words]."""
        import numpy as np
        n = len(traces)
        arr = (C.c_void_p * n)(*[t._h for t in traces])
        rc = self._lib.zksp_hip_machine_load(self._h, pk._h, arr, n)
        if rc == 0:
            rc = self._lib.zksp_hip_machine_prove(self._h)
        if rc:
            raise ZkspError(rc, self.last_error())
        lh = (C.c_int32 * MACHINE_CHIPS)(*machine_cover_heights(traces))
        bw = self._lib.zksp_machine_body_words(self._h, lh)
        out = np.zeros((n, bw), np.uint32)
        rc = self._lib.zksp_hip_machine_fetch_bodies(self._h, out.ctypes.data_as(C.c_void_p), out.size)
        if rc:
            raise ZkspError(rc, self.last_error())
        return out

    def machine_stage(self, chip: int, stage: int, proof_index: int, width: int, log_height: int):
        """One intermediate matrix of a resident proof after ``machine_prove_resident`` (kernel-level parity tests):
        stage 0 main trace, 1 LogUp permutation trace, 2 quotient values; canonical u32 [width][2^log_height]."""
        import numpy as np
        out = np.zeros((width, 1 << log_height), np.uint32)
        rc = self._lib.zksp_hip_machine_fetch_stage(self._h, chip, stage, proof_index, out.ctypes.data_as(C.c_void_p), out.size)
        if rc:
            raise ZkspError(rc, self.last_error())
        return out

    def machine_challenges(self, proof_index: int) -> dict:
        """gamma, beta, alpha, zeta (lists of 4 canonical words) and the chips' cumulative sums of a resident proof."""
        import numpy as np
        out = np.zeros(16 + 4 * MACHINE_CHIPS, np.uint32)
        rc = self._lib.zksp_hip_machine_fetch_challenges(self._h, proof_index, out.ctypes.data_as(C.c_void_p))
        if rc:
            raise ZkspError(rc, self.last_error())
        v = [int(x) for x in out]
        return {"gamma": v[0:4], "beta": v[4:8], "alpha": v[8:12], "zeta": v[12:16],
                "cum": [v[16 + 4 * c:20 + 4 * c] for c in range(MACHINE_CHIPS)]}

    def opcode_histogram(self, rep: ExecReport) -> dict:
        out = {}
        for i in range(64):
            if rep.opcode_hist[i]:
                out[self._lib.zksp_opcode_name(i).decode()] = int(rep.opcode_hist[i])
        return out
