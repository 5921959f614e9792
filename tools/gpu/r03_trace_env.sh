#!/bin/bash
# Kernel trace of the default bench's decode leg (graph replay) under one environment setting.
# usage: tools/gpu/r03_trace_env.sh OUTDIR VAR=VALUE
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=$1; mkdir -p $O
export "$2"
export TL_NO_GRAPH=1  # (eager replay: the profiler has aborted under graph replay of these kernels, profiles/README.md)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-batch-leg --no-prefill-leg --steps 128 --warmup 16 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/trace
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:110]}")
PY
tail -1 $O/bench.json | cut -c1-200
