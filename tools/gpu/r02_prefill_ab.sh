#!/bin/bash
# prefill GEMM tile A/B at 512 and 2048 tokens (8 layers of the 8B shape, 6-term grade): default 128x128 tile (one
# workgroup per CU) against the tiles that fit two workgroups per CU
cd "$(dirname "$0")/../.."
run() { echo "== $1"; env $1 timeout -k 10 200 python tools/bench_prefill.py --layers 8 --lens 512,2048 --terms 6 --reps 3 2>&1 | tail -2; }
run "TL_X=0" && run "TL_GEMM_QI_BN=64" && run "TL_GEMM_BM=64" && run "TL_GEMM_BM=64 TL_GEMM_QI_BN=64"
