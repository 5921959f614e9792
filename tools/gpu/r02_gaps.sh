#!/bin/bash
# kernel trace (timestamps) of the decode step replayed as a hipGraph, and eagerly; gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in eager; do
  rm -rf gpurun_out/prof_gaps
  if [ $mode = eager ]; then export TL_NO_GRAPH=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps -- python3 bench.py --no-cpu-baseline --no-prefill-leg --no-batch-leg --steps 64 --warmup 4 > gpurun_out/gaps_$mode.json 2> gpurun_out/gaps_$mode.err || { tail -5 gpurun_out/gaps_$mode.err; exit 1; }
  f=$(ls gpurun_out/prof_gaps/*/*kernel_trace.csv | head -1)
  echo "== $mode replay"; python3 tools/gap_analysis.py $f || exit 1
done
rm -rf gpurun_out/prof_gaps
