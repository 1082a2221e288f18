"""Machine proof on the GPU: the HIP prover reproduces the CPU oracle's proof bytes exactly and the
host verifier accepts what it produced."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def first_difference(body, expected):
    bad = np.nonzero(body != expected)[0]
    return None if bad.size == 0 else (int(bad[0]), int(bad.size))


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
    assert handles[0].heights() == handles[1].heights() == oracle.machine_heights(traces[0])
    bodies = client.machine_prove_resident(pk, handles)
    host = zk.ProverClient(device=-1, num_queries=nq, pow_bits=pow_bits)
    for i in range(2):
        exp = oracle.machine_prove(traces[i], num_queries=nq, pow_bits=pow_bits)
        hw = 35 + (len(traces[i]["public_values"]) + 3) // 4
        e = np.frombuffer(exp, dtype=np.uint32)[hw:]
        assert e.shape == bodies[i].shape
        assert first_difference(bodies[i], e) is None, first_difference(bodies[i], e)
        proof = handles[i].proof_from_body(pk, bodies[i])
        assert proof.to_bytes() == exp
        host.verify(proof, vk)
