set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_b1
rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --skip-single > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/stats/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 40 steps' worth: find total kernel time and span
n = len(rows)
tail = rows[n // 2:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print("launches in window", len(tail), "busy ms %.2f span ms %.2f busy frac %.3f" % (busy / 1e6, span / 1e6, busy / span))
d = json.load(open("$O/bench.json")); print("ms_per_step", d["ms_per_step"])
import collections
c = collections.Counter(); t = collections.Counter()
for r in tail:
    k = r["Kernel_Name"].split("(")[0][-40:]; c[k] += 1; t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
steps = 40 * len(tail) / n * 2 / 2
for k, v in t.most_common(14): print("%-42s n %5d total ms %.2f avg us %.1f" % (k, c[k], v / 1e6, v / c[k] / 1e3))
PY
