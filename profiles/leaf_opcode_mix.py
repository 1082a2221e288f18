#!/usr/bin/env python3
"""Static opcode mix of one Poseidon2 permutation inside mmcs_leaf_kernel (the dominant kernel of the
machine proof), from the gfx950 ISA hipcc emits for kernels_machine.hip.  Runs anywhere hipcc does
(no GPU needed).  Output: profiles/r05_leaf_opcode_mix.json + the raw per-block histogram.

The permutation is three loops (basic blocks that branch back to themselves) and two straight-line pieces:
  initial linear layer (x1), external rounds 0-3 (one S-box layer + linear layer per trip, x4), internal rounds 0-11
  (one round per trip, x12), internal round 12 (x1), external rounds 4-7 (x4).
The total is cross-checked against the PMC count of the same build (profiles/r05_valu_counters.json).
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "zk-state-proofs_amd", "csrc", "device", "kernels_machine.hip")
asm = "/tmp/zksp_kernels_machine.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                       "--offload-device-only", "-S", SRC, "-o", asm], stderr=subprocess.DEVNULL)
s = open(asm).read()
m = re.search(r"^_ZN4zksp16mmcs_leaf_kernelENS_8LeafArgsEPKNS_8P2ConstsE:", s, re.M)
body = s[m.end():s.find("s_endpgm", m.end())].split("\n")
# basic blocks: a label starts one, and so does the instruction after a conditional branch (the fall-through of a loop)
blocks, cur, serial = collections.OrderedDict(), "entry", 0
blocks[cur] = {"ops": collections.Counter(), "self_loop": False}
for line in body:
    lm = re.match(r"^(\.LBB\d+_\d+):", line)
    if lm:
        cur = lm.group(1)
        blocks[cur] = {"ops": collections.Counter(), "self_loop": False}
        continue
    mm = re.match(r"\s+([a-z]+_[a-z0-9_]+)\s*(\S*)", line)
    if not mm:
        continue
    blocks[cur]["ops"][mm.group(1)] += 1
    if mm.group(1).startswith("s_cbranch"):
        if mm.group(2) == cur:
            blocks[cur]["self_loop"] = True
        serial += 1
        cur = f"{cur}+{serial}"
        blocks[cur] = {"ops": collections.Counter(), "self_loop": False}
valu = lambda c: sum(v for k, v in c.items() if k.startswith("v_"))
# the permutation: the initial linear layer (straight line), external rounds 0-3 (loop, 4 trips), internal rounds 0-11
# (loop, 12 trips), internal round 12 (straight line), external rounds 4-7 (loop, 4 trips)
loops = [(b, d) for b, d in blocks.items() if d["self_loop"] and valu(d["ops"]) >= 80]
assert len(loops) == 3, [(b, valu(d["ops"])) for b, d in loops]
names = list(blocks)
first, mid = names.index(loops[0][0]), names.index(loops[1][0])
straight = [(b, blocks[b]) for b in (names[first - 1], names[mid + 1])]
assert all(valu(d["ops"]) >= 100 for _, d in straight), [(b, valu(d["ops"])) for b, d in straight]
big = [straight[0], loops[0], loops[1], straight[1], loops[2]]
weights = [1, 4, 12, 1, 4]
FULL_RATE = ("v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_ashrrev_i32_e32", "v_lshrrev_b32_e32", "v_lshlrev_b32_e32",
             "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_mov_b32_e32", "v_cndmask_b32_e32", "v_not_b32_e32")
mix = collections.Counter()
for (b, d), w in zip(big, weights):
    for op, n in d["ops"].items():
        if op.startswith("v_"):
            mix["full_rate" if op in FULL_RATE else "half_rate"] += w * n
total = sum(mix.values())
# cross-check with the counter collection of the same build, when there is one: VALU instructions of the largest launch
# (the main LDE of the tallest chips: heights from the bench line of that collection, widths from the library) per lane per
# permutation
pmc = None
try:
    import importlib
    sys.path.insert(0, ROOT)
    v = json.load(open(os.path.join(ROOT, "profiles", "r05_valu_counters.json")))
    heights = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_under_rocprof.json")))["config"]["chip_log_heights"]
    widths = importlib.import_module("zk-state-proofs_amd.client").machine_chip_widths()
    top = max(heights)
    absorptions = (sum(w for (_, _, w, _), lh in zip(widths, heights) if lh == top) + 7) // 8
    pmc = v["SQ_INSTS_VALU"]["zksp::mmcs_leaf_kernel"][1] / (v["batch"] * (2 << top) / 64) / absorptions
except (OSError, KeyError, ValueError, ImportError):
    pass
out = {"kernel": "mmcs_leaf_kernel", "per_permutation_per_lane": dict(mix), "total_valu": total,
       "block_weights": {b: w for (b, _), w in zip(big, weights)},
       "blocks": {b: dict(d["ops"]) for b, d in big},
       "pmc_valu_per_permutation_per_lane": pmc,
       "note": "full_rate = plain 32-bit VOP1/VOP2 instructions (additions, shifts, moves, logic); half_rate = every other "
               "vector instruction (v_mad_i64_i32 / v_mad_u64_u32 of the Montgomery products, v_mul_lo_u32, v_lshl_add_u64 of the "
               "linear layers' 64-bit sums, v_alignbit, v_mad_i32_i24, v_add3, v_min, lane reads)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_leaf_opcode_mix.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("per_permutation_per_lane", "total_valu", "pmc_valu_per_permutation_per_lane")}))
