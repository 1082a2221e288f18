#!/usr/bin/env python3
"""Headline benchmark: Ethereum-MPT STARK proofs/sec on MI355X (BASELINE.json).

One "step" = one drop-in `prove_batch` call (the C-ABI a caller of the reference's `client.prove(&pk, stdin).run()`
binds, prover/src/bin/main.rs:71-75) over `--proofs-per-step` (1 024) synthetic acct-d8 inputs per GPU
(BASELINE configs[1]: single account-trie proof, depth 8): host buffers in, proof objects out - guest tracing on the
host threads, record upload, proving in resident chunks of `--batch` (192) proofs, download and proof wrapping, all
overlapped by the library and all inside the clock.  `value` is that end-to-end rate (round 4's verdict asked for it).
`device_only` beside it is the resident rate of rounds 1-4: one pass of the device hot path over one chunk whose
executor records (48 bytes per cycle) are in HBM before the clock starts; the `roofline` object and the per-stage spans
come from those passes (HIP events on the client's stream).

A proof is the MACHINE proof of the committed sp1-merkle-proof guest in the keccak-precompile shape: 391 400 RV32IM
cycles in six CPU chip instances of 2^16 rows (of eight), the ALU, bitwise and sub-word chips beside them, 62 keccak-f
permutations in a 2^11 x 2634 keccak chip, keccak-memory, memory-boundary, image, program, table, multiplier and
Poseidon2 chips, joined by LogUp buses -- the statement the reference's client.prove() establishes, not a component.
The device path: trace expansion -> LDE -> Poseidon2 mixed-height Merkle commitments -> LogUp -> quotients -> openings
-> FRI -> proof bytes in HBM, Fiat-Shamir on the device, no host round trip.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--proofs-per-step P]

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); `python bench.py --gpus N` starts its own ranks.
Every rank proves its own P proofs per step (weak scaling, no data-path collective) and the ranks all-gather the
32-byte main-trace commitments once at the end of the timed region.  Rank 0 prints ONE JSON line.  Sampled proofs of
the last timed step are verified on the host and one proof of the resident chunk is compared byte for byte with the
CPU oracle; a mismatch fails the benchmark.
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)

# chip widths of the machine proof (zk-state-proofs_amd/csrc/device/air_machine.hpp): (name, preprocessed, main, permutation),
# read from the library at start-up (zksp_machine_chip_widths)
CHIPS = []


def load_chip_widths():
    global CHIPS
    if not CHIPS:
        client_mod = importlib.import_module("zk-state-proofs_amd.client")
        CHIPS = client_mod.machine_chip_widths()
    return CHIPS


def usable_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota and by the GPU box's stated
    per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ZKSP_CPU_THREADS", "16"))))


def stage_bytes(heights):
    """Algorithmic HBM bytes per proof of every stage (SURVEY.md section 8d formulas, this build's widths):
    a stage reads its inputs once and writes its outputs once."""
    out = {k: 0 for k in ("m_trace", "m_lde_main", "m_leaf_main", "m_perm", "m_lde_perm", "m_quotient", "m_lde_quot",
                          "m_open", "m_reduce")}
    for c, ((name, p, w, e), lh) in enumerate(zip(load_chip_widths(), heights)):
        h = 1 << lh
        q = 8 if list(heights).index(lh) == c else 0            # the chips of a height share a quotient: its first chip's
        out["m_trace"] += 4 * h * w
        out["m_lde_main"] += 4 * h * w * 3                      # read H, write LDE 2H (no coefficient arrays)
        out["m_leaf_main"] += 4 * 2 * h * w + 32 * 2 * h        # read every LDE cell once, one digest per row
        out["m_perm"] += 4 * h * (p + w) + 4 * h * e
        out["m_lde_perm"] += 4 * h * e * 3
        out["m_quotient"] += 4 * 2 * h * (p + w + e) + 4 * q * h
        out["m_lde_quot"] += 4 * h * q * 3
        out["m_open"] += 4 * h * (p + w + e + q)
        out["m_reduce"] += 4 * 2 * h * (p + w + e + q) + 16 * 2 * h
    return out


def lib_sha256():
    path = os.path.join(ROOT, "zk-state-proofs_amd", "libzksp.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def source_sha256():
    """Content hash of the device code the library is built from: every file of csrc/device/, the device sources' names and
    the compiler flags of build.py.  The counter collection records it; a library rebuilt elsewhere from the same sources
    still matches (the .so bytes embed the build path), an edited kernel or flag does not, and a host-only change - a new
    host source in build.py's list included - does not invalidate the counters."""
    h = hashlib.sha256()
    top = os.path.join(ROOT, "zk-state-proofs_amd", "csrc", "device")
    build = importlib.import_module("zk-state-proofs_amd.build")
    h.update(repr([f for f in build.FLAGS if f != "-DZKSP_COMPONENT"]).encode())
    h.update(repr(sorted(x for x in build.SOURCES if x.startswith("device/") and x != "device/kernels_bus.hip")).encode())
    for f in sorted(os.path.join(top, n) for n in os.listdir(top) if n.endswith((".hip", ".hpp", ".h"))):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def measured_hbm_traffic(batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC collection
    (profiles/collect_r05.sh -> profiles/r05_hbm_counters.json (r04, r03: the rounds before): FETCH_SIZE and WRITE_SIZE in separate passes,
    KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The collection records the
    sha256 of the libzksp.so it ran and its batch size: any other library or batch makes the figure stale
    and this returns None."""
    for name in ("r05_hbm_counters.json", "r04_hbm_counters.json", "r03_hbm_counters.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            if int(d.get("batch", -1)) != batch or (d.get("lib_sha256") != lib_sha256() and d.get("source_sha256") != source_sha256()):
                continue
            return (2.0 * d["FETCH_SIZE"]["zksp::mmcs_leaf_kernel"][1] + d["WRITE_SIZE"]["zksp::mmcs_leaf_kernel"][1]) * 1024.0
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def valu_model(lib, h, achieved_gperm):
    """What binds the leaf hash is vector-ALU issue.  A CDNA4 SIMD retires 32 lanes per clock of the plain 32-bit VOP1/VOP2
    instructions (add, shift, move, logic: the rate behind the 157 TFLOP/s fp32 figure) and 16 lanes per clock of every
    other vector instruction (64-bit multiply-adds and additions, v_mul_lo_u32, three-operand adds ...): 78.6 and 39.3 T
    lane-ops/s on 256 CUs x 4 SIMDs at 2.4 GHz.  Ceiling = 1 / (full-rate instructions per permutation / 78.6 T +
    half-rate instructions / 39.3 T), instruction counts from the kernel's ISA (profiles/r03_leaf_opcode_mix.json,
    cross-checked there against SQ_INSTS_VALU).  Saturated single-opcode chains measured live are reported beside it
    (they reach 75-90 % of these rates: profiles/r03_opcode_rates.txt has every opcode the kernel uses)."""
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", next(n for n in ("r05_leaf_opcode_mix.json", "r04_leaf_opcode_mix.json", "r03_leaf_opcode_mix.json") if os.path.exists(os.path.join(ROOT, "profiles", n))))))["per_permutation_per_lane"]
    except (OSError, KeyError, ValueError):
        return None
    simd_clocks = 256 * 4 * 2.4e9
    peak = {"full_rate": 32 * simd_clocks, "half_rate": 16 * simd_clocks}
    measured = {}
    for cls, which in (("full_rate", 0), ("half_rate", 1)):  # chains of v_add_u32 / of v_mul_lo_u32
        g = C.c_double()
        if lib.zksp_hip_microbench(h, which, C.byref(g)) == 0:
            measured[cls] = round(g.value / 1e3, 2)
    ceiling = 1.0 / sum(mix[c] / peak[c] for c in mix) / 1e9
    return {"opcode_mix_per_permutation_per_lane": mix, "peak_lane_ops_per_s": {k: round(v / 1e12, 1) for k, v in peak.items()},
            "measured_chain_lane_ops_per_s": measured, "unit_rates": "T lane-ops/s", "ceiling_gperm_per_s": ceiling,
            "achieved_gperm_per_s": achieved_gperm, "frac_of_valu_ceiling": achieved_gperm / ceiling,
            "note": "one issue port per SIMD: the two classes' times add.  The contractual roofline fraction above is against "
                    "HBM, which this kernel does not load"}


def verify_resident_batch(zk, client, pk, vk, handles, traces, with_oracle, n_check=4):
    """Fetches the bodies the last timed step left in HBM, completes `n_check` of them (spread over the
    batch) into proofs and verifies them on the host; optionally compares one with the CPU oracle's
    bytes.  Raises on any mismatch."""
    lib, h = client._lib, client._h
    B = len(handles)
    shape = zk.machine_cover_heights(handles)  # the batch is proven with one shape: the heights of its largest counts
    lh = (C.c_int32 * zk.MACHINE_CHIPS)(*shape)
    bw = lib.zksp_machine_body_words(h, lh)
    bodies = np.zeros((B, bw), np.uint32)
    rc = lib.zksp_hip_machine_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size)
    if rc:
        raise RuntimeError(f"fetch_bodies rc={rc}: {client.last_error()}")
    host = zk.ProverClient(device=-1)
    idx = sorted({int(round(k * (B - 1) / max(1, n_check - 1))) for k in range(n_check)})
    for i in idx:
        proof = handles[i].proof_from_body(pk, bodies[i], shape)
        if proof.public_values != traces[i]:
            raise RuntimeError(f"bench: proof {i} carries wrong public values")
        host.verify(proof, vk)  # raises VerificationError
    oracle_equal, oracle_s = None, None
    if with_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        oracle.build()
        i = idx[len(idx) // 2]
        t0 = time.perf_counter()
        exp = oracle.machine_prove(dict(with_oracle(i), shape=shape))
        oracle_s = time.perf_counter() - t0
        if handles[i].proof_from_body(pk, bodies[i], shape).to_bytes() != exp:
            raise RuntimeError(f"bench: proof {i} of the timed batch differs from the CPU oracle's bytes")
        oracle_equal = i
    return {"verified_indices": idx, "oracle_byte_equal_index": oracle_equal, "proof_bytes": int(bw * 4 + zk.MACHINE_HEADER_WORDS * 4 + 72)}, oracle_s


def cpu_baseline(trace_of, first_s, seconds):
    """The oracle's whole machine-proof CPU restatement ("port") on this host, OpenMP over the granted
    cores.  One acct-d8 proof is 10-60 s of CPU work, so the sample is bounded: repetitions until
    `seconds` have passed (at least one beyond the run the verification step already timed)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    times = [first_s] if first_s else []
    t0 = time.perf_counter()
    while not times or (sum(times) < seconds and len(times) < 5):
        t1 = time.perf_counter()
        oracle.machine_prove(trace_of(0))
        times.append(time.perf_counter() - t1)
    el = sum(times)
    n = len(times)
    return {"value": n / el, "unit": "proofs/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
            "repetitions": n, "median_s_per_proof": sorted(times)[n // 2], "min_s_per_proof": min(times),
            "sample": f"{n} acct-d8 machine proofs (391 400 cycles, {oracle.N_CHIPS} chips, six CPU instances of 2^16 rows, 100 queries) in {el:.1f} s; "
                      "reference SP1 CPU prover unavailable offline, CPU baseline is this repository's oracle/ restatement"}


def keccak_chip_component(zk, fx, client_opts, pk_elf, seconds_budget=20.0):
    """Secondary figure kept for continuity with round 1: the keccak-chip-only proof (format v2), which
    is NOT a proof of execution.  Batch 256, 3 timed steps."""
    B, logh = 256, 11
    client = zk.ProverClient(device=client_opts["device"], max_batch=B, proof_mode=zk.PROOF_KECCAK_CHIP)
    lib, h = client._lib, client._h
    pk, vk = client.setup(pk_elf)
    vkw = [int(x) for x in np.frombuffer(vk.digest, dtype=np.uint32)]
    s = zk.SP1Stdin()
    s.write(fx.acct_fixture(8, seed=1).to_borsh())
    st1 = client.keccak_states(pk, s)
    rng = np.random.default_rng(9)
    states = rng.integers(0, 2**64, (B, 62, 25), dtype=np.uint64)
    states[0] = st1
    pvd = [int(x) for x in np.frombuffer(hashlib.sha256(fx.ACCOUNT_VALUE).digest(), dtype=np.uint32)]
    obs = np.zeros((B, 44), np.uint32)
    row = vkw + [logh, 62, 0, 0]
    for w in pvd:
        row += [w & 0xFFFF, w >> 16]
    row += [0] * 16
    obs[:] = row
    n_perms = np.full(B, 62, np.uint32)
    if lib.zksp_hip_load_batch(h, logh, B, 62, states.ctypes.data_as(C.c_void_p), n_perms.ctypes.data_as(C.c_void_p),
                               obs.ctypes.data_as(C.c_void_p)):
        return None
    lib.zksp_hip_prove_resident(h)
    lib.zksp_hip_sync(h)
    t0 = time.perf_counter()
    for _ in range(3):
        lib.zksp_hip_prove_resident(h)
    lib.zksp_hip_sync(h)
    return {"value": 3 * B / (time.perf_counter() - t0), "unit": "keccak-chip proofs/s", "batch": B,
            "note": "component only (round-1 format v2): binds keccak-f inputs to outputs, does not prove execution"}


def as_committed_mode(zk, fx, device, with_oracle, batch=32):
    """The same acct-d8 input with the guest exactly as committed (reference circuits/elf/riscv32im-succinct-zkvm-elf: software
    keccak, no precompile): 1 406 960 cycles in six CPU instances of 2^18 rows, keccak chips empty.  `batch` proofs resident
    (the largest the HBM left by the caller allows, at most 32), 2 timed steps; one proof verified on the host and, with
    the oracle, compared byte for byte."""
    from concurrent.futures import ThreadPoolExecutor
    client = zk.ProverClient(device=device, keccak_mode=zk.KECCAK_OBSERVE, max_batch=batch)
    lib, h = client._lib, client._h
    pk, vk = client.setup(zk.merkle_elf())
    try:
        import torch
        free_b, _tot = torch.cuda.mem_get_info(device)
    except Exception:
        free_b = 200 << 30
    NB = int(max(4, min(batch, (free_b * 4 // 5) // (4 << 30))))  # (an as-committed proof takes 3.6 GB of workspace)

    def trace(i):
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(8, seed=1 + i).to_borsh())
        return client.machine_trace_handle(pk, s)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(NB, usable_cores())) as ex:  # (the library releases the GIL while it traces)
        handles = list(ex.map(trace, range(NB)))
    trace_wall_ms = (time.perf_counter() - t0) * 1e3 / NB
    arr = (C.c_void_p * NB)(*[t._h for t in handles])
    if lib.zksp_hip_machine_load(h, pk._h, arr, NB) or lib.zksp_hip_machine_prove(h):
        return {"error": client.last_error()}
    lib.zksp_hip_sync(h)
    t0 = time.perf_counter()
    for _ in range(2):
        lib.zksp_hip_machine_prove(h)
    lib.zksp_hip_sync(h)
    el = time.perf_counter() - t0
    shape = zk.machine_cover_heights(handles)
    lh = (C.c_int32 * zk.MACHINE_CHIPS)(*shape)
    bw = lib.zksp_machine_body_words(h, lh)
    bodies = np.zeros((NB, bw), np.uint32)
    if lib.zksp_hip_machine_fetch_bodies(h, bodies.ctypes.data_as(C.c_void_p), bodies.size):
        return {"error": client.last_error()}
    client.release_workspace()
    host = zk.ProverClient(device=-1, keccak_mode=zk.KECCAK_OBSERVE)
    checked = [1, NB - 1]
    for i in checked:
        host.verify(handles[i].proof_from_body(pk, bodies[i], shape), vk)
    equal_idx = None
    if with_oracle:
        import oracle
        s = zk.SP1Stdin()
        s.write(fx.acct_fixture(8, seed=1 + checked[0]).to_borsh())
        t = client.machine_trace(pk, s)
        exp = oracle.machine_prove(dict(t, shape=shape))
        if handles[checked[0]].proof_from_body(pk, bodies[checked[0]], shape).to_bytes() != exp:
            return {"error": "as-committed proof differs from the oracle's bytes"}
        equal_idx = checked[0]
    return {"value": 2 * NB / el, "unit": "proofs/s", "batch": NB, "ms_per_proof": el * 1e3 / (2 * NB), "chip_log_heights": shape,
            "host_trace_wall_ms_per_proof": trace_wall_ms, "host_verified_indices": checked, "oracle_byte_equal_index": equal_idx,
            "note": "guest as committed: software keccak-f inside the CPU, ALU and bitwise chips (1 406 960 cycles, six CPU "
                    "instances of 2^18 rows); full parameters (100 queries, 16 PoW bits)"}


def leaf_check_mode(zk, fx, client, pk, vk, payload):
    """Row f4, stage 2b at full parameters: an acct-d8 leaf proof, then ONE more acct-d8 run whose proof also establishes the
    leaf's query phase under the challenges the leaf's own transcript yields (100 queries: the index words and their
    canonical bits, every Merkle opening of its four rounds and of its FRI layers against the roots the transcript absorbed,
    the reduced openings from the opened rows, the folding chain) - the unit of a recursion tree's inner node.  Host part
    (verifying the leaf and logging its transcript and openings as chip records) and device part timed apart; verified on the
    host with the leaf's STUB and with the statement alone.  Then a node of four leaves, and a two-level tree (16 leaves -> 4
    nodes -> 1 root, BASELINE config 5 in small) whose root verifies from the stubs of the 20 proofs below it."""
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    s = zk.SP1Stdin()
    s.write(payload)
    leaf = client.prove(pk, s).run()
    s = zk.SP1Stdin()
    s.write(payload)
    t0 = time.perf_counter()
    client.set_verified_leaf(s, leaf, vk)
    log_ms = (time.perf_counter() - t0) * 1e3
    probe = zk.SP1Stdin()
    probe.write(payload)
    client.set_verified_leaf(probe, leaf, vk)
    t = client.machine_trace(pk, probe)
    rows, qrows, trows, tuples = len(t["leaf_p2_rows"]), len(t["leaf_qr_rows"]), len(t["leaf_tr_rows"]), len(t["leaf_pub_tuples"])
    del t, probe
    times = []
    outer = None
    for _ in range(3):
        s2 = zk.SP1Stdin()
        s2.write(payload)
        client.set_verified_leaf(s2, leaf, vk)
        t1 = time.perf_counter()
        outer = client.prove(pk, s2).run()
        times.append((time.perf_counter() - t1) * 1e3)
    plain = []
    for _ in range(3):
        s3 = zk.SP1Stdin()
        s3.write(payload)
        t1 = time.perf_counter()
        client.prove(pk, s3).run()
        plain.append((time.perf_counter() - t1) * 1e3)
    host = zk.ProverClient(device=-1)
    stub = leaf.stub()
    t2 = time.perf_counter()
    host.verify_with_leaf(outer, vk, stub, vk)
    verify_ms = (time.perf_counter() - t2) * 1e3
    host.verify_public(outer, vk, host.leaf_public(stub, vk))
    raw = outer.to_bytes()
    shape = [int.from_bytes(raw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
    prove_ms, plain_ms = sorted(times)[1], sorted(plain)[1]
    names = zk.MACHINE_CHIP_NAMES
    # the node of a recursion tree of arity 4: ONE run that checks four leaf proofs (zksp_stdin_add_verified_leaf)
    arity = 4
    more = [leaf]
    for k in range(1, arity):
        sk = zk.SP1Stdin()
        sk.write(fx.acct_fixture(8, seed=4000 + k).to_borsh())
        more.append(client.prove(pk, sk).run())
    node_times, node_log = [], 0.0
    node = None
    for _ in range(2):
        s4 = zk.SP1Stdin()
        s4.write(payload)
        t1 = time.perf_counter()
        client.add_verified_leaves(s4, more, [vk] * arity)  # (verified and logged side by side on the host's threads)
        node_log = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        node = client.prove(pk, s4).run()
        node_times.append((time.perf_counter() - t1) * 1e3)
    host.verify_with_leaves(node, vk, [p.stub() for p in more], [vk] * arity)
    nraw = node.to_bytes()
    nshape = [int.from_bytes(nraw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
    node_ms = min(node_times)
    tree_node = {"leaves": arity, "poseidon2_chip_log_height": nshape[names.index("poseidon2")],
                 "host_log_ms": node_log, "prove_end_to_end_ms": node_ms,
                 "ms_per_verified_leaf": (node_ms - plain_ms + node_log) / arity, "proof_bytes": len(nraw),
                 "statement": "one run whose proof establishes the query phases of four leaf proofs under their own transcripts' "
                              "challenges; verified on the host with the four leaves' stubs"}
    # config 5 in small: 16 leaves -> 4 nodes -> 1 root; the nodes are leaves of the root (with their own statements)
    proven = {}  # (the leaves of a size are proven once: the tree of another arity over them reuses the proofs and the figure)

    def recursion_tree(n_tree, seed0, arity=arity):
        if n_tree not in proven:
            tl = []
            for i in range(n_tree):
                si = zk.SP1Stdin()
                si.write(fx.acct_fixture(8, seed=seed0 + i).to_borsh())
                tl.append(si)
            t1 = time.perf_counter()
            tleaves, status = client.prove_batch(pk, tl)
            assert status == [0] * n_tree
            proven[n_tree] = (tleaves, time.perf_counter() - t1)
            del tl
        tleaves, leaves_s = proven[n_tree]

        # (the nodes' own guest inputs are made before the clock: building an MPT fixture in Python takes 8 ms)
        node_payloads, cnt, depth = {}, n_tree, 0
        while cnt > 1:
            depth += 1
            cnt = (cnt + arity - 1) // arity
            for k in range(cnt):
                # (a node's own guest run is incidental - the reference's recursive circuit has none - so it is a small one: a
                # depth-1 account proof, 60 000 cycles instead of the leaves' 391 400)
                node_payloads[(depth, k)] = fx.acct_fixture(1, seed=seed0 + 100_000 + 4096 * depth + k).to_borsh()

        def make_stdin(depth, k):
            sdin = zk.SP1Stdin()
            sdin.write(node_payloads[(depth, k)])
            return sdin

        t1 = time.perf_counter()
        levels, _statements = farm.prove_tree(client, host, pk, vk, tleaves, make_stdin, arity)
        tree_s = time.perf_counter() - t1
        root = levels[-1][0]
        stubs = farm.tree_of_stubs(levels, arity)
        t1 = time.perf_counter()
        host.verify_tree(root, vk, stubs)
        tree_verify_s = time.perf_counter() - t1
        below = [p for lv in levels[:-1] for p in lv]
        n_nodes = sum(len(lv) for lv in levels[1:])
        rraw = root.to_bytes()
        rshape = [int.from_bytes(rraw[8 + 4 * c:12 + 4 * c], "little") for c in range(zk.MACHINE_CHIPS)]
        return {"levels": [len(lv) for lv in levels], "arity": arity, "leaves_prove_batch_s": leaves_s, "nodes_and_root_s": tree_s,
                "ms_per_node": tree_s * 1e3 / n_nodes, "leaves_per_s_through_the_whole_tree": n_tree / (leaves_s + tree_s),
                "root_verify_from_stubs_s": tree_verify_s, "root_proof_bytes": len(rraw),
                "stub_bytes_read": sum(len(p.stub().to_bytes()) for p in below),
                "full_proof_bytes_below_the_root": sum(len(p.to_bytes()) for p in below),
                "root_poseidon2_chip_log_height": rshape[names.index("poseidon2")],
                "statement": f"every node's proof checks the query phases of its {arity} children (leaves, or nodes with their own "
                             "statements); verifying the root reads the root proof and the STUBS (no query phase) of the "
                             f"{len(below)} proofs below it: their bus balance and constraint identity at zeta are still checked "
                             "natively (stage 2b).  A level is ONE prove_batch call: the host's leaf checks are deferred to it and run "
                             "on its tracing threads beside the proving of the nodes that are ready (zksp_stdin_defer_verified_leaves)"}

    # stage 2c, first piece: the constraint identity at zeta as the recorded straight-line program, cross-checked against the
    # native evaluation on this leaf (zksp_zeta_program_selftest; the first call builds the program)
    host.zeta_program_selftest(stub, vk)
    t1 = time.perf_counter()
    zinfo = host.zeta_program_selftest(stub, vk)
    zinfo = dict(zinfo, stub_check_with_selftest_ms=(time.perf_counter() - t1) * 1e3,
                 statement="the AIR templates instantiated over a recording value type: operations c = a * b + d over extension "
                           "cells, the same program for every shape; what an arithmetic chip will execute (DESIGN.md section 7.1)")
    tree = recursion_tree(16, 4100)
    # BASELINE config 5 in full: 1 024 leaf proofs -> 256 -> 64 -> 16 -> 4 -> 1 (341 node proofs)
    tree1024 = recursion_tree(1024, 8000)
    # ... and with nodes of five leaves: 258 nodes instead of 341, a node's Poseidon2 chip still 2^19 rows (448 000 of them used
    # instead of 358 400) - fewer proofs for the GPU, more checking per node for the host
    tree1024_5 = recursion_tree(1024, 60000, arity=5)
    proven.clear()
    return {"poseidon2_rows": rows, "query_rows": qrows, "transcript_rows": trows, "public_tuples": tuples, "tree_node_of_4": tree_node,
            "two_level_tree": tree, "tree_of_1024_leaves": tree1024, "tree_of_1024_leaves_arity_5": tree1024_5, "zeta_program": zinfo,
            "poseidon2_chip_log_height": shape[names.index("poseidon2")],
            "query_chip_log_height": shape[names.index("query")],
            "host_log_ms": log_ms, "prove_end_to_end_ms": prove_ms, "plain_prove_end_to_end_ms": plain_ms,
            "ms_per_verified_leaf": prove_ms - plain_ms + log_ms, "rows_per_ms": rows / max(prove_ms - plain_ms, 1e-3),
            "proof_bytes": len(raw), "stub_bytes": len(stub.to_bytes()), "host_verify_with_stub_ms": verify_ms,
            "statement": "the proof also establishes the query phase of the leaf proof under the challenges the leaf's own transcript "
                         "yields: 100 queries x (index word -> canonical bits -> positions; 4 mixed-height Merkle openings with their "
                         "sponges, Horner sums and injections + every FRI layer opening against the absorbed roots; reduced openings; "
                         "the folding chain to the final constant); public: the transcript's blocks and a few constants per leaf and "
                         "height (stage 2b; stage 2a had 4 500 public tuples per leaf); verified on the host with the leaf's stub and "
                         "with the statement alone"}


def workload_stdins(zk, fx, name):
    """The stdin buffers of one of BASELINE.json's throughput workloads, and what every proof's
    public values must be.  slot-d5x256: config 3 as 256 independent runs of the committed guest (no sp1-storage-proof
    circuit exists); rcptx300: config 4, every receipt of a 300-receipt block-shaped trie; acct-d8x1024: config 5's
    substitute (the reference's recursion circuit is a todo!()): 1024 leaf proofs whose 32-byte commitments are what a
    recursion tree would take in."""
    if name == "slot-d5x256":
        ins = [fx.slot_fixture(i) for i in range(256)]
        return [m.to_borsh() for m in ins], None
    if name == "rcptx300":
        mpt = importlib.import_module("zk-state-proofs_amd.mpt")
        receipts = mpt.synthetic_block_receipts(300, seed=12)
        trie = mpt.block_trie(receipts)
        return [mpt.block_proof_input(trie, i).to_borsh() for i in range(300)], receipts
    if name == "acct-d8x1024":
        return [fx.acct_fixture(8, seed=5000 + i).to_borsh() for i in range(1024)], [fx.ACCOUNT_VALUE] * 1024
    raise SystemExit(f"unknown workload {name}")


def pipelined_workload(zk, fx, client, pk, vk, name, n_verify=6):
    """One drop-in prove_batch call over a whole workload (host buffers in, proof objects out: guest tracing on the host
    threads, record upload, proving and download overlapped by the library), a spread of the proofs verified on the
    host afterwards.  The rate includes everything; the verification is outside the clock."""
    bufs, expect = workload_stdins(zk, fx, name)
    stdins = []
    for b in bufs:
        sdin = zk.SP1Stdin()
        sdin.write(b)
        stdins.append(sdin)
    t0 = time.perf_counter()
    proofs, status = client.prove_batch(pk, stdins)
    el = time.perf_counter() - t0
    if status != [0] * len(bufs):
        return {"error": f"{sum(1 for x in status if x)} of {len(bufs)} runs failed: {client.last_error()}"}
    host = zk.ProverClient(device=-1)
    idx = sorted({int(round(k * (len(bufs) - 1) / (n_verify - 1))) for k in range(n_verify)})
    shapes = set()
    for i, p in enumerate(proofs):
        raw = p.to_bytes()
        shapes.add(raw[8:8 + 4 * zk.MACHINE_CHIPS])
        if expect is not None and p.public_values != expect[i]:
            return {"error": f"proof {i} carries wrong public values"}
    for i in idx:
        host.verify(proofs[i], vk)
    out = {"proofs": len(bufs), "seconds": el, "proofs_per_s": len(bufs) / el, "shapes": len(shapes), "verified_indices": idx,
           "includes": "guest tracing, H2D, proving, D2H, proof objects (prove_batch end to end, one GPU)"}
    if name == "acct-d8x1024":
        # config 5's aggregation step (row f4, stage 1): the 1024 main-trace commitments, in proof order as the farm
        # all-gathers them, become the aggregation payload of one more run; its proof carries their proven Merkle root
        farm = importlib.import_module("zk-state-proofs_amd.farm")
        leaves = np.array([farm.trace_root_of(p.to_bytes()) for p in proofs], np.uint32)
        sdin = zk.SP1Stdin()
        sdin.write(bufs[0])
        sdin.set_aggregation(leaves)
        t1 = time.perf_counter()
        agg = client.prove(pk, sdin).run()
        agg_ms = (time.perf_counter() - t1) * 1e3
        host.verify_aggregate(agg, vk, leaves)
        out["aggregation"] = {"leaves": int(agg.aggregation[0]), "root": agg.aggregation[1], "prove_ms": agg_ms,
                              "statement": "one more acct-d8 run whose proof also establishes the Poseidon2 Merkle root of the 1024 "
                                           "commitments (Poseidon2 chip, 1023 rows); verified on the host with the leaves"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=192,
                    help="machine proofs proven in lockstep per GPU per step (about 1.0 GB of HBM each at acct-d8)")
    ap.add_argument("--proofs-per-step", type=int, default=1024,
                    help="acct-d8 inputs handed to one prove_batch call per GPU per step (the timed, end-to-end step)")
    ap.add_argument("--device-steps", type=int, default=5,
                    help="resident passes timed for `device_only`, the roofline and the per-stage spans")
    ap.add_argument("--cpu-seconds", type=float, default=40.0,
                    help="budget of the CPU baseline: repetitions until it is spent, at most 5 (SURVEY 8d: 5 repetitions, median and min)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-single", action="store_true", help="skip latency / end-to-end / component sections (profiling runs)")
    ap.add_argument("--workload", default="acct-d8", choices=["acct-d8", "slot-d5x256", "rcptx300", "acct-d8x1024", "all"],
                    help="acct-d8 (default): the metric's configuration, plus the pipelined workloads of BASELINE configs 3-5 as "
                         "secondary figures; a named workload: only that one's pipelined prove_batch figure is added; all = default")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the contract's line): every rank proves its own resident batch.  strong: ONE workload "
                         "(--workload slot-d5x256 / rcptx300 / acct-d8x1024) is sharded block-cyclically over the ranks "
                         "(farm.prove_sharded: prove_batch end to end per rank, then the all-gather of the roots)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + ZKSP_BENCH_SAME_DEVICE=1 rehearses the N>1 path on a one-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has
        # touched HIP or torch yet, and the ranks are CHILD processes (never an exec of this one);
        # rank 0's JSON line passes through on stdout and the launcher's exit code becomes ours.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # Rank 0's JSON line must be the ONLY thing on stdout (the driver parses it; a gloo / RCCL banner in front of it broke
    # a round-2 rehearsal): from here on everything any library prints to fd 1 goes to stderr, and the line is written to
    # the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit_json(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # the oracle's C restatement is OpenMP code; the thread count must be fixed before libgomp loads
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    # idle OpenMP workers must sleep, not spin: the library's own executor threads run on the same cores afterwards
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and the proving path has no CPU fallback")
    if os.environ.get("ZKSP_BENCH_SAME_DEVICE") == "1":
        local_rank = 0  # rehearsal: every rank on cuda:0 (RCCL refuses that, hence gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_device = torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None

    zk = importlib.import_module("zk-state-proofs_amd")
    fx = importlib.import_module("zk-state-proofs_amd.fixtures")
    farm = importlib.import_module("zk-state-proofs_amd.farm")
    B = args.batch
    client = zk.ProverClient(device=local_rank, max_batch=B)
    lib, h = client._lib, client._h
    t_setup = time.perf_counter()
    pk, vk = client.setup(zk.merkle_elf())
    setup_s = time.perf_counter() - t_setup

    if args.scaling == "strong":
        # ---- strong scaling: one workload's proofs block-cyclic over the ranks (BASELINE configs 4 and 5 on N GPUs) ----
        wname = args.workload if args.workload not in ("acct-d8", "all") else "rcptx300"
        bufs, expect = workload_stdins(zk, fx, wname)
        n_total = len(bufs)
        mine = farm.shard_indices(n_total, rank, world)
        elapsed_all = []
        for step in range(args.warmup + args.steps):
            stdins = []
            for b_ in bufs:
                sdin = zk.SP1Stdin()
                sdin.write(b_)
                stdins.append(sdin)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx, proofs, status = farm.prove_sharded(client, pk, stdins, rank, world)
            if status != [0] * len(idx):
                raise RuntimeError(f"rank {rank}: a run of the shard failed: {client.last_error()}")
            local_roots = np.array([farm.trace_root_of(p.to_bytes()) for p in proofs], np.uint32).reshape(-1, 8)
            roots = farm.gather_roots(local_roots, n_total, rank, world, device=coll_device)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                tt = torch.tensor([el], dtype=torch.float64, device=coll_device if coll_device is not None else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            if step >= args.warmup:
                elapsed_all.append(el)
            assert roots.shape == (n_total, 8)
        host = zk.ProverClient(device=-1)
        for k in (0, len(idx) // 2, len(idx) - 1):
            if expect is not None and proofs[k].public_values != expect[idx[k]]:
                raise RuntimeError("strong-scaling run: wrong public values")
            host.verify(proofs[k], vk)
        if rank == 0:
            elapsed = sum(elapsed_all)
            emit_json({"metric": "Ethereum-MPT STARK proofs/sec", "value": n_total * args.steps / elapsed, "unit": "proofs/s",
                       "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
                       "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                       "dtype": "u32 (BabyBear mod 2^31-2^27+1, Montgomery)", "data": "synthetic",
                       "config": {"workload": f"{wname}: {n_total} machine proofs sharded block-cyclically over {world} GPU(s), "
                                              "prove_batch end to end per rank (guest tracing, H2D, proving, D2H, proof objects), "
                                              "all-gather of the 32-byte roots", "proofs_per_step": n_total,
                                  "parallelism": f"proof-farm x{world}, strong scaling"},
                       "roofline": None, "cpu_baseline": None,
                       "note": "secondary mode; the contract's line (roofline, cpu_baseline, device_only) is --scaling weak"})
        if dist is not None:
            dist.destroy_process_group()
        return

    def check(rc):
        if rc:
            raise RuntimeError(f"zksp rc={rc}: {client.last_error()}")

    def sync():
        check(lib.zksp_hip_sync(h))
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- the timed region: K drop-in prove_batch calls, P host buffers in, P proof objects out per call and rank ----
    P = args.proofs_per_step
    step_bufs = [fx.acct_fixture(8, seed=5000 + rank * P + i).to_borsh() for i in range(P)]

    def fresh_stdins():
        out = []
        for buf in step_bufs:
            sdin = zk.SP1Stdin()
            sdin.write(buf)
            out.append(sdin)
        return out

    def prove_step(stdins):
        proofs, status = client.prove_batch(pk, stdins)
        if status != [0] * len(stdins):
            raise RuntimeError(f"rank {rank}: {sum(1 for x in status if x)} of {len(stdins)} runs failed: {client.last_error()}")
        return proofs

    for _ in range(args.warmup):
        del_me = prove_step(fresh_stdins())
        del del_me
    all_stdins = [fresh_stdins() for _ in range(args.steps)]  # (the caller's buffers exist before it calls: 4 KB each)
    step_s = []
    barrier()
    sync()
    t0 = time.perf_counter()
    # What the caller does with the proofs afterwards is not the prover's time: the proof objects of a step (2.7 GB per
    # 1024) stay alive while the next steps run, and are released on a helper thread once more than 48 GB are held
    # (returning 2.7 GB to the kernel takes 0.4 s - a first version of this loop dropped them inside the next step's clock).
    import threading
    kept, droppers = [], []
    keep_steps = max(1, int((48 << 30) // max(1, P * 2_800_000)))
    last_proofs = None
    for k in range(args.steps):
        t_step = time.perf_counter()
        last_proofs = prove_step(all_stdins[k])
        step_s.append(time.perf_counter() - t_step)
        kept.append(last_proofs)
        if len(kept) > keep_steps:
            old = kept.pop(0)
            th = threading.Thread(target=lambda o: o.clear(), args=(old,), daemon=True)
            del old
            th.start()
            droppers.append(th)
    if dist is not None:
        # the one exchange the path has: 32-byte main-trace commitments of every proof of the last step to every rank
        local_roots = np.array([farm.trace_root_of(p.to_bytes()) for p in last_proofs], np.uint32).reshape(-1, 8)
        n_total = world * P
        mine = farm.shard_indices(n_total, rank, world)
        roots = farm.gather_roots(local_roots, n_total, rank, world, device=coll_device)
        assert roots.shape == (n_total, 8) and np.array_equal(roots[mine], local_roots)
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device if coll_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del all_stdins
    for th in droppers:
        th.join()
    kept.clear()
    # the last step's proofs are checked, not just timed: public values of all, a spread verified on the host
    host = zk.ProverClient(device=-1)
    for i, pr in enumerate(last_proofs):
        if pr.public_values != fx.ACCOUNT_VALUE:
            raise RuntimeError(f"bench: proof {i} of the last timed step carries wrong public values")
    e2e_checked = sorted({int(round(k * (P - 1) / 5)) for k in range(6)})
    for i in e2e_checked:
        host.verify(last_proofs[i], vk)  # raises VerificationError
    step_proof_bytes = len(last_proofs[0].to_bytes())
    aggregation = None
    if rank == 0 and not args.skip_single and P >= 2 and (P & (P - 1)) == 0:
        # BASELINE config 5's aggregation step (row f4, stage 1): the P main-trace commitments of the last step, in proof
        # order as the farm all-gathers them, become the aggregation payload of one more run
        leaves = np.array([farm.trace_root_of(p.to_bytes()) for p in last_proofs], np.uint32)
        sdin = zk.SP1Stdin()
        sdin.write(step_bufs[0])
        sdin.set_aggregation(leaves)
        t1 = time.perf_counter()
        agg = client.prove(pk, sdin).run()
        agg_ms = (time.perf_counter() - t1) * 1e3
        host.verify_aggregate(agg, vk, leaves)
        aggregation = {"leaves": int(agg.aggregation[0]), "root": agg.aggregation[1], "prove_ms": agg_ms,
                       "statement": f"one more acct-d8 run whose proof also establishes the Poseidon2 Merkle root of the {P} "
                                    "commitments of the last timed step (Poseidon2 chip); verified on the host with the leaves"}
    del last_proofs

    # ---- device_only: B distinct acct-d8 runs traced on the host, records resident in HBM, `--device-steps` passes ----
    payloads = [fx.acct_fixture(8, seed=1 + rank * B + i).to_borsh() for i in range(B)]
    handles, stdins_keep = [], []
    exec_s = 0.0
    for buf in payloads:
        s = zk.SP1Stdin()
        s.write(buf)
        t_exec = time.perf_counter()
        handles.append(client.machine_trace_handle(pk, s))
        exec_s += time.perf_counter() - t_exec
        stdins_keep.append(s)
    heights = zk.machine_cover_heights(handles)  # one shape per batch: the heights that cover the largest counts
    trace_ms_per_proof = exec_s * 1e3 / B
    arr = (C.c_void_p * B)(*[t._h for t in handles])
    check(lib.zksp_hip_machine_load(h, pk._h, arr, B))  # (lays the device arena out for a resident batch of this shape)
    t_load = time.perf_counter()
    check(lib.zksp_hip_machine_load(h, pk._h, arr, B))
    load_ms = (time.perf_counter() - t_load) * 1e3
    check(lib.zksp_hip_machine_prove(h))
    sync()
    dsteps = max(1, args.device_steps)
    lib.zksp_hip_profile_reset(h)
    lib.zksp_hip_profile_enable(h, 1)
    sync()
    t0 = time.perf_counter()
    for _ in range(dsteps):
        check(lib.zksp_hip_machine_prove(h))
    sync()
    dev_elapsed = time.perf_counter() - t0
    lib.zksp_hip_profile_enable(h, 0)

    # ---- the timed batch is checked, not just timed ----
    def trace_of(i):
        return client.machine_trace(pk, stdins_keep[i])

    pvs = [fx.ACCOUNT_VALUE] * B
    # the CPU oracle (byte comparison, cpu_baseline) runs on rank 0 at N = 1 only: under torch.distributed.run every
    # rank is pinned to OMP_NUM_THREADS=1 and one acct-d8 oracle proof would take minutes
    use_oracle = rank == 0 and world == 1 and not args.no_cpu_baseline
    checked, oracle_s = verify_resident_batch(zk, client, pk, vk, handles, pvs, trace_of if use_oracle else None)

    # ---- per-stage spans (HIP events on the client's own stream) ----
    tot, cnt = C.c_double(), C.c_uint64()
    names = ["m_trace", "m_lde_main", "m_commit_main", "m_leaf_main", "m_perm", "m_lde_perm", "m_commit_perm", "m_quotient",
             "m_lde_quot", "m_commit_quot", "m_open", "merkle_open", "m_reduce", "fri_commit", "fri_fold", "grind", "transcript",
             "m_assemble"]
    spans, launches = {}, {}
    for name in names:
        check(lib.zksp_hip_profile_read(h, name.encode(), C.byref(tot), C.byref(cnt)))
        spans[name] = tot.value / dsteps
        launches[name] = cnt.value
    sb = stage_bytes(heights)
    stage_gbs = {k: round(B * v / (spans[k] * 1e-3) / 1e9, 1) for k, v in sb.items() if spans.get(k)}
    # dominant kernel: the leaf hash of the main commitment (one launch per step: mmcs_leaf_kernel over the tallest group)
    leaf_ms = spans["m_leaf_main"]
    leaf_group = [(w, lh) for (name, p, w, e), lh in zip(load_chip_widths(), heights) if lh == max(heights)]
    alg_bytes = B * sum(4 * (2 << lh) * w for w, lh in leaf_group) + B * 32 * (2 << max(heights))
    achieved = alg_bytes / (leaf_ms * 1e-3) / 1e9
    stage_gbs["m_leaf_main"] = round(achieved, 1)  # this span is the tallest group's launch only
    perms = B * (2 << max(heights)) * ((sum(w for w, _ in leaf_group) + 7) // 8)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- latency, end-to-end and component figures (rank 0, after the timed region) ----
    single_ms = single_sync_ms = e2e_ms = component = as_committed = leaf_check = in_process = None
    pipelined = {}
    if not args.skip_single:
        # device time of one resident proof: the timed batch's client takes a batch of one (the same launch sequence
        # zksp_prove enqueues for a single run).  Not a second client: the runtime maps streams onto four hardware queues,
        # and a second client's lanes would share queues with the first one's - its kernels would run one after the other.
        sc, spk = client, pk
        sh = handles[0]
        one = (C.c_void_p * 1)(sh._h)
        if lib.zksp_hip_machine_load(sc._h, spk._h, one, 1):
            raise RuntimeError(sc.last_error())
        for _ in range(5):  # (the first pass also builds the device tables of the opening stage)
            if lib.zksp_hip_machine_prove(sc._h):
                raise RuntimeError(sc.last_error())
        lib.zksp_hip_sync(sc._h)
        t1 = time.perf_counter()
        for _ in range(40):
            if lib.zksp_hip_machine_prove(sc._h):
                raise RuntimeError(sc.last_error())
        lib.zksp_hip_sync(sc._h)
        single_ms = (time.perf_counter() - t1) * 1e3 / 40
        # the same pass one at a time, the host waiting for each (what a caller of zksp_prove sees of the device)
        sync_ms = []
        for _ in range(10):
            t1 = time.perf_counter()
            if lib.zksp_hip_machine_prove(sc._h) or lib.zksp_hip_sync(sc._h):
                raise RuntimeError(sc.last_error())
            sync_ms.append((time.perf_counter() - t1) * 1e3)
        single_sync_ms = sorted(sync_ms)[len(sync_ms) // 2]
        del sh, sc, spk
        # The same measurement in a process of its own (tests/gpu_single_latency.py as a child, this process idle meanwhile):
        # what a caller that proves one run gets.  In THIS process - two minutes into the bench, after 6 000 proofs - the
        # batch of one measures 1.2 ms more, whatever the order of things before it (tests/gpu_single_in_bench_order_probe.py
        # could not reproduce it); both are reported, the child's as the figure.
        in_process = {"device_ms": single_ms, "synchronised_ms": single_sync_ms}
        try:
            import subprocess
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_single_latency.py"), "1", "--json"],
                                 capture_output=True, text=True, timeout=300, env=dict(os.environ, ZKSP_DEVICE=str(local_rank)))
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("JSON ")]
            if out.returncode == 0 and line:
                child = json.loads(line[-1][5:])
                single_ms, single_sync_ms = child["ms_per_pass"], child["synchronised_ms"]
        except Exception:  # (no child: the in-process figures stand)
            pass
        e2e = []
        for _ in range(5):
            s = zk.SP1Stdin()
            s.write(payloads[0])
            t2 = time.perf_counter()
            proof = client.prove(pk, s).run()
            e2e.append((time.perf_counter() - t2) * 1e3)
        e2e_ms = sorted(e2e)[2]  # median of 5: guest tracing, H2D, proving, D2H, proof object
        client.verify(proof, vk)
        # (round 1's keccak-chip component figure: only with the library built for that path, ZKSP_COMPONENT=1)
        component = keccak_chip_component(zk, fx, {"device": local_rank}, zk.merkle_elf()) if zk.client.COMPONENT else None
        # (acct-d8x1024, config 5's substitute, IS the timed step since round 5)
        wanted = ["slot-d5x256", "rcptx300"] if args.workload in ("acct-d8", "all") else [args.workload]
        for wname in wanted:
            pipelined[wname] = pipelined_workload(zk, fx, client, pk, vk, wname)
        leaf_check = leaf_check_mode(zk, fx, client, pk, vk, payloads[0])
        client.release_workspace()  # the as-committed batch needs the HBM the timed batch's arena holds
        as_committed = as_committed_mode(zk, fx, local_rank, use_oracle)

    total_proofs = world * P * args.steps
    out = {
        "metric": "Ethereum-MPT STARK proofs/sec",
        "value": total_proofs / elapsed,
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (BabyBear mod 2^31-2^27+1, Montgomery)",
        "data": "synthetic",
        "config": {
            "workload": f"acct-d8 x {P} per GPU per step through the drop-in prove_batch (host buffers in, proof objects out: guest "
                        "tracing on the host threads, H2D of the executor's records, proving, D2H, proof wrapping - all inside "
                        "the clock); each proof the machine proof of a depth-8 account-trie MPT proof of the committed "
                        "sp1-merkle-proof guest, whole execution proven (keccak precompile shape, 391 400 cycles; chips as name "
                        "2^log-height x (preprocessed + main + permutation columns): "
                        + ", ".join(f"{n} 2^{lh} x ({p}+{w}+{e})" for (n, p, w, e), lh in zip(load_chip_widths(), heights))
                        + "; one quotient of 8 columns per height; LogUp buses; blowup 2, 100 FRI queries, 16 PoW bits)",
            "statement": f"guest executed from its entry point to HALT(0) with these public values (machine proof, format v{zk.MACHINE_VERSION})",
            "cells_per_proof": sum((w + e + (8 if list(heights).index(lh) == c else 0)) << lh
                                   for c, ((n, p, w, e), lh) in enumerate(zip(load_chip_widths(), heights))),
            "chip_log_heights": heights,
            "resident_chunk_per_gpu": B,
            "proofs_per_step": world * P,
            "proof_bytes": step_proof_bytes,
            "host_threads_per_gpu": usable_cores(),
            "parallelism": f"proof-farm x{world} (independent proofs, all-gather of 32-byte roots only)",
        },
        "timed_steps_s": [round(x, 4) for x in step_s],
        "timed_step_checked": {"public_values_checked": P, "host_verified_indices": e2e_checked},
        # the resident rate (rounds 1-4's `value`): one pass of the device hot path over a chunk whose records are in HBM
        "device_only": {"value": B * dsteps / dev_elapsed, "unit": "proofs/s",
                        "scope": "this rank's GPU", "batch": B, "passes": dsteps, "ms_per_pass": dev_elapsed * 1e3 / dsteps,
                        "note": "executor records resident in HBM before the clock, proof bodies left in HBM (no tracing, "
                                "no PCIe); the roofline object and the stage spans below are measured over these passes"},
        "roofline": {
            "kernel": f"mmcs_leaf_kernel over the main LDE of the tallest chips ({len(leaf_group)} matrices, {sum(w for w, _ in leaf_group)} columns: the CPU instances and "
                      f"the table-sized chips; Poseidon2 sponge, {(sum(w for w, _ in leaf_group) + 7) // 8} "
                      f"permutations per row, 2^{max(heights) + 1} rows per proof)",
            # bound by vector-ALU issue (one Poseidon2 permutation per 32 bytes absorbed); achieved / peak / frac are the
            # contractual HBM figures: algorithmic bytes over the launch time against the 8 TB/s peak
            "bound": "valu",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_hbm_traffic(B),
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": leaf_ms,
            "launches_per_pass": launches["m_leaf_main"] / dsteps,
            "poseidon2_gperm_per_s": perms / (leaf_ms * 1e-3) / 1e9,
            "valu": valu_model(lib, h, perms / (leaf_ms * 1e-3) / 1e9),
        },
        "timed_batch_checked": checked,
        "device_ms_per_step_by_stage": {k: round(v, 3) for k, v in spans.items()},
        "stage_algorithmic_gbs": stage_gbs,
        "single_proof_device_ms": single_ms,
        "single_proof_in_bench_process": in_process,
        "single_proof_device_synchronised_ms": single_sync_ms,
        "single_proof_end_to_end_ms": e2e_ms,
        "host_trace_ms_per_proof": trace_ms_per_proof,  # one core: traced execution with memory-argument bookkeeping
        "records_h2d_ms_per_batch": load_ms,
        "setup_s": setup_s,
        # BASELINE configs 3 and 4 as one prove_batch call each on this GPU (host buffers in, proof objects out); config 5's
        # substitute (1 024 acct-d8 leaf proofs) is the timed step itself, its aggregation proof (stage 1) here
        "pipelined_workloads": pipelined,
        "aggregation_of_last_step": aggregation,
        "keccak_chip_component": component,
        "as_committed_2p21": as_committed,
        "leaf_check": leaf_check,
    }
    if use_oracle:
        out["cpu_baseline"] = cpu_baseline(trace_of, oracle_s, args.cpu_seconds)
    emit_json(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
