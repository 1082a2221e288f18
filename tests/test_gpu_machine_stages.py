"""Kernel-level parity of the machine prover: after a proving pass the device's
intermediate matrices - every chip's main trace (the trace expansion kernels, the table chip's counted multiplicities),
LogUp permutation trace (perm_terms_cpu_kernel / perm_terms_kernel + the running-sum scans) and quotient values
(machine_quotient_kernel<chip>, the keccak task kernel) - are compared with the oracle's for the challenges the device's
own transcript sampled.  When whole-proof parity breaks, this names the kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def proven(zk, fx, oracle):
    client = zk.ProverClient(device=0, num_queries=6, pow_bits=5, max_batch=2)
    pk, vk = client.setup(zk.merkle_elf())
    handles, traces = [], []
    for m in (fx.acct_fixture(1, seed=3), fx.acct_fixture(1, seed=4)):
        s = zk.SP1Stdin()
        s.write(m.to_borsh())
        handles.append(client.machine_trace_handle(pk, s))
        traces.append(client.machine_trace(pk, s))
    shape = zk.machine_cover_heights(handles)
    client.machine_prove_resident(pk, handles)
    return client, [dict(t, shape=shape) for t in traces], shape


@pytest.mark.parametrize("index", [0, 1])
def test_main_traces(zk, oracle, proven, index):
    client, traces, shape = proven
    for chip, (name, pw, mw, ew) in enumerate(zk.machine_chip_widths()):
        dev = client.machine_stage(chip, 0, index, mw, shape[chip])
        _, exp = oracle.machine_fill(traces[index], chip)
        bad = np.argwhere(dev != exp)
        assert bad.size == 0, (name, "first differing (column, row)", bad[:4].tolist())


def test_permutation_traces_and_cumulative_sums(zk, oracle, proven):
    client, traces, shape = proven
    ch = client.machine_challenges(1)
    for chip, (name, pw, mw, ew) in enumerate(zk.machine_chip_widths()):
        dev = client.machine_stage(chip, 1, 1, ew, shape[chip])
        exp, cum = oracle.machine_stage_perm(traces[1], chip, ch["gamma"], ch["beta"])
        bad = np.argwhere(dev != exp)
        assert bad.size == 0, (name, "first differing (column, row)", bad[:4].tolist())
        assert cum == ch["cum"][chip], name


def test_quotient_values(zk, oracle, proven):
    """The chips of one height share a quotient (the first of them carries it): the device's buffer must be the sum of
    the oracle's per-chip shares, each folded with its own range of powers of alpha."""
    client, traces, shape = proven
    ch = client.machine_challenges(0)
    P = 2013265921
    names = [w[0] for w in zk.machine_chip_widths()]
    for chip in range(len(shape)):
        group = [c for c in range(len(shape)) if shape[c] == shape[chip]]
        if group[0] != chip:
            continue
        dev = client.machine_stage(chip, 2, 0, 8, shape[chip])
        exp = np.zeros_like(dev, dtype=np.uint64)
        for c in group:
            exp = (exp + oracle.machine_stage_quotient(traces[0], c, ch["alpha"], ch["gamma"], ch["beta"])) % P
        bad = np.argwhere(dev != exp.astype(np.uint32))
        assert bad.size == 0, ([names[c] for c in group], "first differing (column, row)", bad[:4].tolist())
