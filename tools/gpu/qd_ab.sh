# A/B of the direct small-token GEMM's launch knobs on the multi-sequence decode step (GEMM route for every size)
cd $GRAFT_REPO_ROOT
: > gpurun_out/qd_ab.txt
for cfg in "2 8" "4 8" "6 8"; do
  set -- $cfg
  echo "== TL_QD_WANT=$1 TL_QD_SMAX=$2" >> gpurun_out/qd_ab.txt
  TL_QD_WANT=$1 TL_QD_SMAX=$2 TL_BATCH_GEMV_MAX=0 timeout -k 10 200 python tools/bench_batch.py --batch 32 64 --terms 6 2>> gpurun_out/qd_ab.err | cut -c60-170 >> gpurun_out/qd_ab.txt || exit 1
done
cat gpurun_out/qd_ab.txt
