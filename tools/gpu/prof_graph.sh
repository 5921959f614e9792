#!/bin/bash
# rocprofv3 kernel trace of the bench with the decode step replayed as a hipGraph (the product path). Round 1 reported a
# profiler segfault on graph launches; this run records what happens now, stderr kept.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_graph
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_graph -- python3 bench.py --no-cpu-baseline --no-prefill-leg --no-batch-leg --steps 32 --warmup 4 > gpurun_out/r02_prof_graph.json 2> gpurun_out/r02_prof_graph.err
rc=$?
echo "exit code $rc" | tee gpurun_out/r02_prof_graph.rc
if [ $rc -eq 0 ]; then cp $(ls gpurun_out/prof_graph/*/*kernel_stats.csv | head -1) gpurun_out/r02_prof_graph_kernel_stats.csv && head -8 gpurun_out/r02_prof_graph_kernel_stats.csv; fi
rm -rf gpurun_out/prof_graph
tail -5 gpurun_out/r02_prof_graph.err | cut -c1-200
exit 0
