#!/bin/bash
# headline + long-context decode configurations of the 8B workload (no CPU baseline / prefill / batch legs)
set -o pipefail
cd "$(dirname "$0")/../.."
A="--no-cpu-baseline --no-prefill-leg --no-batch-leg --steps 128 --warmup 16"
timeout -k 10 200 python bench.py $A > gpurun_out/r02_cfg_short.json 2> gpurun_out/r02_cfg_short.err || exit 1
for p in 500 1000 2000 4000 8000; do
  timeout -k 10 200 python bench.py $A --max-pos 8192 --prompt-len $p > gpurun_out/r02_cfg_ctx$p.json 2> gpurun_out/r02_cfg_ctx$p.err || exit 1
done
python - <<'PY'
import json
for t in ("short", "ctx500", "ctx1000", "ctx2000", "ctx4000", "ctx8000"):
    d = json.loads(open(f"gpurun_out/r02_cfg_{t}.json").read().strip().splitlines()[-1])
    e = d.get("eager_step_us", {})
    print(t, d["value"], "tok/s", d["ms_per_step"], "ms/step; attention", e.get("attention"), "attn_out", e.get("attn_out_gemv"))
PY
