#!/bin/bash
# Round 3: every BASELINE.json configuration on one GPU (decode): TinyLlama-1.1B Q8_0 (configs[0]) and Q4_K_M (configs[1]),
# Llama-3.2-3B Q6_K (configs[2]), Llama-3-8B Q4_K_M at short / 2k / 8k context (configs[3]), and the 70B-shaped model
# (configs[4]: the N = 1 anchor of its tensor-parallel curve; 41 GB of weights, fits the 288 GB of one GPU).
set -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_cfg
mkdir -p $O
A="--no-cpu-baseline --no-batch-leg --steps 128 --warmup 16"
timeout -k 10 300 python bench.py --workload tinyllama-1.1b --mix q8_0 $A > $O/tinyllama-1.1b_q8_0.json 2> $O/tinyllama-1.1b_q8_0.err || exit 1
timeout -k 10 300 python bench.py --workload tinyllama-1.1b --mix q4_k_m $A > $O/tinyllama-1.1b_q4_k_m.json 2> $O/tinyllama-1.1b_q4_k_m.err || exit 1
timeout -k 10 300 python bench.py --workload llama-3.2-3b $A > $O/llama-3.2-3b_q6_k.json 2> $O/llama-3.2-3b_q6_k.err || exit 1
timeout -k 10 300 python bench.py --workload llama-3-8b --no-cpu-baseline --steps 128 --warmup 16 > $O/8b_short.json 2> $O/8b_short.err || exit 1
timeout -k 10 300 python bench.py --workload llama-3-8b --max-pos 8192 --prompt-len 2000 $A --no-prefill-leg > $O/8b_ctx2k.json 2> $O/8b_ctx2k.err || exit 1
timeout -k 10 300 python bench.py --workload llama-3-8b --max-pos 8192 --prompt-len 8000 $A --no-prefill-leg > $O/8b_ctx8k.json 2> $O/8b_ctx8k.err || exit 1
if [ "${SKIP_70B:-0}" != "1" ]; then
  timeout -k 10 900 python bench.py --workload llama-3-70b --max-pos 2048 $A --no-prefill-leg > $O/70b_q4_k_m.json 2> $O/70b_q4_k_m.err || exit 1
fi
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_cfg/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], "tok/s", d["ms_per_step"], "ms", d.get("hbm_gbps_whole_step"), "GB/s whole step; gate/up frac", d["roofline"]["frac"],
          "prefill", d.get("prefill", {}).get("tokens_per_s"))
PY
