#!/usr/bin/env python3
"""Static opcode mix of one Poseidon2 permutation inside mmcs_leaf_kernel (the dominant kernel of the
machine proof), from the gfx950 ISA hipcc emits for kernels_machine.hip.  Runs anywhere hipcc does
(no GPU needed).  Output: profiles/r02_leaf_opcode_mix.json + the raw per-block histogram.

The permutation is four loops (basic blocks with a back edge) plus straight-line pieces:
  initial linear layer (x1), external rounds 0-3 (one S-box layer + linear layer per iteration, x4),
  internal rounds (the compiler keeps three rounds per iteration, x4, the thirteenth round is folded
  into the surrounding code), external rounds 4-7 (x4).
Weights are therefore 1 / 4 / 4 / 4 on the four largest VALU blocks in program order; the total is
checked against the PMC count of round 1 (4 853 VALU instructions per permutation per lane).
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
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = collections.Counter()
for line in body:
    lm = re.match(r"^(\.LBB\d+_\d+):", line)
    if lm:
        cur = lm.group(1)
        blocks[cur] = collections.Counter()
        continue
    mm = re.match(r"\s+([a-z]+_[a-z0-9_]+)", line)
    if mm:
        blocks[cur][mm.group(1)] += 1
big = [(b, c) for b, c in blocks.items() if sum(v for k, v in c.items() if k.startswith("v_")) >= 150]
assert len(big) == 4, [b for b, _ in big]
weights = [1, 4, 4, 4]
CLASSES = {"mad64": ("v_mad_i64_i32", "v_mad_u64_u32"), "mul_lo": ("v_mul_lo_u32",), "add64": ("v_lshl_add_u64",)}
mix = collections.Counter()
for (b, c), w in zip(big, weights):
    for op, n in c.items():
        if not op.startswith("v_"):
            continue
        cls = next((k for k, ops in CLASSES.items() if op in ops), "simple32")
        mix[cls] += w * n
total = sum(mix.values())
out = {"kernel": "mmcs_leaf_kernel", "per_permutation_per_lane": dict(mix), "total_valu": total,
       "block_weights": {b: w for (b, _), w in zip(big, weights)},
       "blocks": {b: dict(c) for b, c in big},
       "note": "mad64 = v_mad_i64_i32 / v_mad_u64_u32 (Montgomery products and reductions), add64 = v_lshl_add_u64 "
               "(64-bit sums of the linear layers), simple32 = moves, shifts, 24-bit multiply-adds, lane reads"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_leaf_opcode_mix.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("per_permutation_per_lane", "total_valu")}))
