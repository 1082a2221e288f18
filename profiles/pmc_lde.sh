#!/bin/bash
# Issue-mix counters of an LDE kernel (run on the GPU box via gpurun): ZKSP_NTT_LOGH (default 11) picks the height,
# PMC_KERNEL (default lde_lds) the kernel whose dispatches are averaged.
# Counter passes only: no --kernel-trace/--sys-trace next to --pmc.
set -e
export TMPDIR=/tmp ZKSP_NTT_LOGH=${ZKSP_NTT_LOGH:-11}
K=${PMC_KERNEL:-lde_lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_lde_$ZKSP_NTT_LOGH
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/tests/gpu_ntt_bench.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tests/gpu_ntt_bench.py > $O/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/p3 -- python3 $R/tests/gpu_ntt_bench.py > $O/p3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p4 -- python3 $R/tests/gpu_ntt_bench.py > $O/p4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tests/gpu_ntt_bench.py > $O/st.log 2>&1
cat $O/st.log; head -8 $O/st/*/*kernel_stats.csv | cut -c1-160
python3 - <<PY
import csv, glob, collections
for sub in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$O/" + sub + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$K" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(sub, k, "avg %.4g n %d" % (sum(v) / len(v), len(v)))
PY
