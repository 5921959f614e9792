cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gemv -- python3 tools/bench_gemv.py 8b 100 > gpurun_out/prof_gemv.log 2>&1 || exit 1
cat gpurun_out/prof_gemv.log | grep -v "^W2\|^E2\|rocprof" | tail -12
f=$(ls gpurun_out/prof_gemv/*/*kernel_stats.csv | head -1); head -8 $f
TL_GEMV_WAVES=8 timeout -k 10 100 python3 tools/bench_gemv.py 8b 100 | sed 's/^/W8 /' 
TL_GEMV_NT=0 timeout -k 10 100 python3 tools/bench_gemv.py 8b 100 | sed 's/^/NT0 /'
