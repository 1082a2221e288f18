#!/bin/bash
# Round-5 profile collection (run on the GPU box via gpurun from the repository root):
#   bash profiles/collect_r05.sh
# 1. kernel statistics of the bench command; 2./3. HBM traffic counters in their own passes
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE do not fit one pass; counter passes carry no
# trace options); 4. VALU instruction counts.  The program sits directly behind `--`.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd /tmp
# one small end-to-end step (the timed region of the bench line) and three resident passes of the default chunk: the kernels of a
# pass are the same either way; the counters are read per dispatch
ARGS="--steps 1 --warmup 0 --proofs-per-step 192 --device-steps 3 --no-cpu-baseline --skip-single"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/bench_stats.json 2> $O/stats.err
echo "stats done"
if [ -n "$STATS_ONLY" ]; then cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv; head -40 $O/kernel_stats.csv; exit 0; fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/bench_fetch.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/bench_write.json 2> $O/write.err
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/valu -- python3 $R/bench.py $ARGS > $O/bench_valu.json 2> $O/valu.err
echo "valu done"
python3 - <<PY
import csv, glob, json, collections, hashlib
o = "$O"
def counter(name, sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(o + "/" + sub + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), max(v), len(v)) for k, v in acc.items()}
open(o + "/kernel_stats.csv", "w").write(open(glob.glob(o + "/stats/*/*kernel_stats.csv")[0]).read())
b = json.load(open(o + "/bench_stats.json"))
lib = hashlib.sha256(open("$R/zk-state-proofs_amd/libzksp.so", "rb").read()).hexdigest()
import sys
sys.path.insert(0, "$R")
import bench
src = bench.source_sha256()
out = {"batch": b["config"]["resident_chunk_per_gpu"], "lib_sha256": lib, "source_sha256": src,
       "units": "KB per dispatch (avg, max, dispatches); the max of mmcs_leaf_kernel is the CPU chip's main-trace launch; FETCH_SIZE reads half the streamed bytes on gfx950",
       "FETCH_SIZE": counter("FETCH_SIZE", "fetch"), "WRITE_SIZE": counter("WRITE_SIZE", "write")}
json.dump(out, open(o + "/hbm_counters.json", "w"), indent=1)
valu = {"batch": out["batch"], "lib_sha256": lib, "source_sha256": src, "note": "per dispatch (avg, max, n); GRBM_GUI_ACTIVE is summed over the 8 XCDs",
        "SQ_INSTS_VALU": counter("SQ_INSTS_VALU", "valu"), "GRBM_GUI_ACTIVE": counter("GRBM_GUI_ACTIVE", "valu")}
json.dump(valu, open(o + "/valu_counters.json", "w"), indent=1)
for k in ("FETCH_SIZE", "WRITE_SIZE"):
    for kern, v in sorted(out[k].items(), key=lambda kv: -kv[1][1])[:6]:
        print(k, kern, "avg %.0f max %.0f n %d" % v)
print(json.dumps(b["roofline"]))
PY
