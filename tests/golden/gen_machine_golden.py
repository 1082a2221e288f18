"""Regression pins of the machine proof (format v13): for two small fixtures, the chip heights, the verifying key
and the SHA-256 of the oracle's proof bytes at 8 queries / 6 proof-of-work bits.

    python tests/golden/gen_machine_golden.py        # rewrites tests/golden/machine_kat.json

These are NOT an independent restatement (tests/golden/gen_golden.py is that, for the primitives): they are the
oracle's own output, frozen, so that an accidental change of the arithmetisation, of the transcript order or of
the proof layout shows up as a failing CPU test instead of passing silently because the oracle and the device
prover changed together.  Regenerate deliberately when the format changes, and say so in the commit."""
import hashlib
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NQ, POW = 8, 6


def cases(fx):
    return {"acct-d1": fx.acct_fixture(1), "tx-d2": fx.tx_fixture()}


def compute():
    import oracle
    zk = importlib.import_module("zk-state-proofs_amd")
    fx = importlib.import_module("zk-state-proofs_amd.fixtures")
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    pk, vk = client.setup(zk.merkle_elf())
    out = {"format_version": zk.MACHINE_VERSION, "num_queries": NQ, "pow_bits": POW,
           "vk_prep_root": [int(x) for x in vk.machine[0]], "vk_digest": [int(x) for x in vk.machine[1]], "cases": {}}
    for name, m in cases(fx).items():
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        t = client.machine_trace(pk, s)
        proof = oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
        out["cases"][name] = {"chip_log_heights": oracle.machine_heights(t), "cycles": int(t["cycles"].shape[0]),
                              "proof_bytes": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                              "public_values_sha256": hashlib.sha256(t["public_values"]).hexdigest()}
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "machine_kat.json"), "w") as f:
        json.dump(compute(), f, indent=1)
    print("wrote machine_kat.json")
