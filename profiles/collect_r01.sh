#!/bin/bash
# Round-1 profile collection (run on the GPU box via gpurun from the repository root).
# 1. kernel statistics of the bench command; 2./3. HBM traffic counters in their own passes
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r01
mkdir -p $O
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --skip-single"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/bench_stats.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/bench_fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/bench_write.json 2> $O/write.err
python3 - <<PY
import csv, glob, json, collections
o = "$O"
def stats():
    f = glob.glob(o + "/stats/*/*kernel_stats.csv")[0]
    return list(csv.reader(open(f)))
def counter(name, sub):
    f = glob.glob(o + "/" + sub + "/*/*counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") == name:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), max(v), len(v)) for k, v in acc.items()}
open(o + "/kernel_stats.csv", "w").write(open(glob.glob(o + "/stats/*/*kernel_stats.csv")[0]).read())
batch = json.load(open(o + "/bench_stats.json"))["config"]["batch_per_gpu"]
out = {"batch": batch, "units": "KB per dispatch (avg, max, dispatches); FETCH_SIZE reads half the streamed bytes on gfx950",
       "FETCH_SIZE": counter("FETCH_SIZE", "fetch"), "WRITE_SIZE": counter("WRITE_SIZE", "write")}
json.dump(out, open(o + "/hbm_counters.json", "w"), indent=1)
for k in ("FETCH_SIZE", "WRITE_SIZE"):
    for kern, v in sorted(out[k].items(), key=lambda kv: -kv[1][1])[:8]:
        print(k, kern, "avg %.0f max %.0f n %d" % v)
PY
