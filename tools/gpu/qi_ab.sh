# A/B of the integer-plane GEMM's tile shapes: prefill tokens/s (T = 128, 512, 2048; 6 terms) and a 64-sequence step
cd $GRAFT_REPO_ROOT
for cfg in "64 128" "128 128" "128 64" "64 64"; do
  set -- $cfg
  echo "BM=$1 BN=$2"
  TL_GEMM_BM=$1 TL_GEMM_QI_BN=$2 timeout -k 10 300 python -u tools/bench_prefill.py 2>&1 | grep '"terms": 6' | cut -c1-70 || exit 1
  TL_GEMM_BM=$1 TL_GEMM_QI_BN=$2 timeout -k 10 300 python -u tools/bench_batch.py --batch 64 128 --terms 6 2>&1 | cut -c60-170 || exit 1
done
