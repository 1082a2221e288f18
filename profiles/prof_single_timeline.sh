# Kernel timeline of ONE steady-state pass at batch 1 (stream / queue per kernel), for critical-path reading.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_b1t
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/stats -- python3 $R/bench.py --batch 1 --steps 12 --warmup 4 --no-cpu-baseline --skip-single > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, glob, json, os
fs = sorted(glob.glob("$O/stats/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a pass starts with the first trace-expansion kernel: find the starts of cpu_trace_kernel launches that follow a gap
names = [r["Kernel_Name"].split("(")[0].split("::")[-1] for r in rows]
first = names.index(next(n for n in names if "trace_kernel" in n))
key = names[first]
starts = [i for i, n in enumerate(names) if n == key and (i == 0 or names[i - 1] != key)]
# passes: group by the largest recurring period - use the count of launches per pass = len(rows after warm) / passes
per = {}
d = json.load(open("$O/bench.json"))
print("bench ms_per_step", d["ms_per_step"], "single_pass launches approx", len(rows) / 16.0)
n_pass = 16
L = len(rows) // n_pass
seg = rows[len(rows) - 2 * L: len(rows) - L]  # the last but one pass (approximately aligned)
# align on the first kernel named key within the segment
off = next(i for i, r in enumerate(seg) if r["Kernel_Name"].split("(")[0].split("::")[-1] == key)
base = len(rows) - 2 * L + off
seg = rows[base: base + L]
t0 = int(seg[0]["Start_Timestamp"])
qs = {}
out = open("$O/timeline.txt", "w")
for r in seg:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write("%9.1f %8.1f q%d %6s %s\n" % (s / 1e3, (e - s) / 1e3, q, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"].split("(")[0][-60:]))
out.close()
span = int(seg[-1]["End_Timestamp"]) - t0
print("pass span ms %.2f launches %d queues %d" % (span / 1e6, len(seg), len(qs)))
PY
