#!/bin/bash
# multi-sequence decode and prefill sweeps after folding the K-split reductions into their consumers and merging
# q | k | v and gate | up launches (6-term grade)
set -o pipefail
cd "$(dirname "$0")/../.."
timeout -k 10 300 python tools/bench_batch.py --batch 2 4 5 8 12 16 24 32 48 64 128 256 --terms 6 --steps 16 > gpurun_out/r02_q_batch_sweep.jsonl 2> gpurun_out/r02_q_batch_sweep.err || exit 1
timeout -k 10 300 python tools/bench_prefill.py --lens 16,32,128,512,2048 --terms 6 > gpurun_out/r02_q_prefill_sweep.jsonl 2> gpurun_out/r02_q_prefill_sweep.err || exit 1
cut -c66-190 gpurun_out/r02_q_batch_sweep.jsonl; cat gpurun_out/r02_q_prefill_sweep.jsonl
