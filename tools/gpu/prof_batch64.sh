# kernel trace of the multi-sequence decode with 64 sequences (64-token GEMM tile, split K)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_b64
TL_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b64 -- python3 tools/bench_batch.py --batch 64 --terms 6 --steps 16 > gpurun_out/prof_b64.jsonl 2> gpurun_out/prof_b64.err || exit 1
cp $(ls gpurun_out/prof_b64/*/*kernel_stats.csv | head -1) gpurun_out/prof_b64_kernel_stats.csv
head -16 gpurun_out/prof_b64_kernel_stats.csv | cut -c1-210
rm -rf gpurun_out/prof_b64
