"""MI355X-native STARK prover for the zk-state-proofs Merkle-proof guest.

The directory name carries a hyphen (it mirrors the reference's repository name),
so import it with ``importlib.import_module("zk-state-proofs_amd")``.
"""
from .client import (  # noqa: F401
    GuestPanic, ProverClient, SP1ProofWithPublicValues, SP1Stdin, VerificationError, ZkspError, load_library, proof_from_body,
    KECCAK_OBSERVE, KECCAK_REPLACE, KECCAK_SOFTWARE, PROOF_MACHINE, PROOF_KECCAK_CHIP,
    MACHINE_CHIPS, MACHINE_CHIP_NAMES, MACHINE_HEADER_WORDS, MACHINE_VERSION, PUB_TUPLE_WORDS, P2_REC_WORDS, QR_REC_WORDS, TR_REC_WORDS, machine_chip_widths,
    machine_cover_heights, merkle_path_nodes,
)
from .fixtures import MerkleProofInput, StorageProofInput  # noqa: F401

MERKLE_ELF_PATH = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(
    __import__("os").path.abspath(__file__))), "fixtures", "sp1-merkle-proof.elf")


def merkle_elf() -> bytes:
    """The committed guest, as the reference embeds it with ``include_elf!``
    (reference prover/src/bin/main.rs:1-2)."""
    with open(MERKLE_ELF_PATH, "rb") as f:
        return f.read()
