#!/bin/bash
# K-split reductions folded into the consumer kernels (default) against one reduction launch per GEMM (TL_GEMM_DEFER=0):
# multi-sequence decode at 12..64 sequences and prefill at 128 / 512 tokens, 6-term grade
set -o pipefail
cd "$(dirname "$0")/../.."
for d in 0 1; do
  TL_GEMM_DEFER=$d timeout -k 10 250 python tools/bench_batch.py --batch 12 16 32 48 64 128 --terms 6 --steps 16 > gpurun_out/r02_defer${d}_batch.jsonl 2> gpurun_out/r02_defer${d}_batch.err || exit 1
  TL_GEMM_DEFER=$d timeout -k 10 250 python tools/bench_prefill.py --lens 32,128,512,2048 --terms 6 > gpurun_out/r02_defer${d}_prefill.jsonl 2> gpurun_out/r02_defer${d}_prefill.err || exit 1
done
for d in 0 1; do echo "== TL_GEMM_DEFER=$d"; cat gpurun_out/r02_defer${d}_batch.jsonl gpurun_out/r02_defer${d}_prefill.jsonl | cut -c1-260; done
