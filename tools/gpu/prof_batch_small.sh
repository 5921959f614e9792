# kernel trace of the multi-sequence decode with 2 and 4 sequences (multi-vector GEMV route)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bs
TL_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bs -- python3 tools/bench_batch.py --batch 2 4 --terms 6 --steps 16 > gpurun_out/prof_bs.jsonl 2> gpurun_out/prof_bs.err || exit 1
cp $(ls gpurun_out/prof_bs/*/*kernel_stats.csv | head -1) gpurun_out/prof_bs_kernel_stats.csv
head -24 gpurun_out/prof_bs_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_bs
