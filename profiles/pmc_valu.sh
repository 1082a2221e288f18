#!/bin/bash
# Vector-ALU instruction counts of the bench kernels (run on the GPU box via gpurun from the
# repository root).  Counter passes only: no trace options next to --pmc.
#   cycles_per_valu_inst = n_simd * (GRBM_GUI_ACTIVE / n_xcd) / SQ_INSTS_VALU
# GRBM_GUI_ACTIVE is summed over the 8 XCDs (for the leaf kernel it reads 8 x duration x 2.4 GHz).
# SQ_ACTIVE_INST_VALU is collected too but equals SQ_INSTS_VALU on these kernels, so it is NOT an
# independent busy measure; what the pass gives is instructions and cycles per instruction, to be
# read against per-opcode issue costs (SIMD-32: 2 cycles at best, multiplies 4-5).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_valu
mkdir -p $O
cd /tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --skip-single"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python3 $R/bench.py $ARGS > $O/a.json 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/b -- python3 $R/bench.py $ARGS > $O/b.json 2> $O/b.err
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob("$O/" + sub + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
n_simd, n_xcd = 256 * 4, 8
for k, c in acc.items():
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    if "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
        if avg.get("SQ_INSTS_VALU", 0) > 0:
            avg["cycles_per_valu_inst"] = n_simd * avg["GRBM_GUI_ACTIVE"] / n_xcd / avg["SQ_INSTS_VALU"]
    avg["dispatches"] = len(next(iter(c.values())))
    out[k] = avg
batch = json.load(open("$O/a.json"))["config"]["batch_per_gpu"]
json.dump({"batch": batch, "n_simd": n_simd, "n_xcd": n_xcd, "note": "per-dispatch averages; cycles_per_valu_inst = n_simd*(GRBM_GUI_ACTIVE/n_xcd)/SQ_INSTS_VALU",
           "kernels": out}, open("$O/valu_counters.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:12]:
    print("%-44s cycles %.4g valu_insts %.4g cycles/inst %.2f" % (k[:44], v.get("GRBM_GUI_ACTIVE", 0), v.get("SQ_INSTS_VALU", 0), v.get("cycles_per_valu_inst", 0)))
PY
