#!/bin/bash
# Kernel trace of the default bench with the prefill and multi-sequence legs (eager decode replay).
# usage: tools/gpu/r03_trace_legs.sh OUTDIR [VAR=VALUE] [extra bench args...]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=$1; mkdir -p $O
if [ $# -ge 2 ]; then export "$2"; shift; fi
shift
export TL_NO_GRAPH=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --steps 32 --warmup 4 "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 - $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/gemm_by_grid.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    key = (n.split("(")[0][-44:], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Workgroup_Size_X"])
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("calls   avg_us   total_ms  kernel grid_x grid_y wg")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:6d} {t/c:9.2f} {t/1e3:9.2f}  {k}")
PY
cat $O/gemm_by_grid.txt | head -40
rm -rf $O/trace
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f}%  {r['Name'][:120]}")
PY
