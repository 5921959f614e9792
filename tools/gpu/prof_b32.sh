# kernel trace of the multi-sequence decode with 32 sequences on the GEMM route (direct small-token kernel)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b32
TL_NO_GRAPH=1 TL_BATCH_GEMV_MAX=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b32 -- python3 tools/bench_batch.py --batch 32 --terms 6 --steps 16 > gpurun_out/prof_b32.jsonl 2> gpurun_out/prof_b32.err || exit 1
cp $(ls gpurun_out/prof_b32/*/*kernel_stats.csv | head -1) gpurun_out/prof_b32_kernel_stats.csv
head -20 gpurun_out/prof_b32_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_b32
