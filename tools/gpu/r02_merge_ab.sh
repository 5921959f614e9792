#!/bin/bash
# q | k | v and gate | up in one launch each (default) against one launch per matrix (TL_GEMM_MERGE=0), multi-sequence decode
set -o pipefail
cd "$(dirname "$0")/../.."
for d in 0 1; do
  TL_GEMM_MERGE=$d timeout -k 10 250 python tools/bench_batch.py --batch 7 8 12 16 32 48 64 --terms 6 --steps 16 > gpurun_out/r02_merge${d}_batch.jsonl 2> gpurun_out/r02_merge${d}_batch.err || exit 1
done
for d in 0 1; do echo "== TL_GEMM_MERGE=$d"; cut -c66-190 gpurun_out/r02_merge${d}_batch.jsonl; done
