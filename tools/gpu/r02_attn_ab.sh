#!/bin/bash
# A/B of the decode-attention kernel at long context: the previous library (libtlhip_old.so, built from the parent
# commit) against the current one; the 8B shape at positions 8000.. and the operator-level microbenchmark.
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
ARGS="--workload llama-3-8b --max-pos 8192 --prompt-len 8000 --steps 64 --warmup 8 --no-cpu-baseline --no-prefill-leg --no-batch-leg"
cp $L/libtlhip.so /tmp/new.so || exit 1
if [ -f $L/libtlhip_old.so ]; then
  cp $L/libtlhip_old.so $L/libtlhip.so
  timeout -k 10 200 python bench.py $ARGS > gpurun_out/r02_ctx8k_old.json 2> gpurun_out/r02_ctx8k_old.err || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
  timeout -k 10 100 python tools/bench_attn.py > gpurun_out/r02_attn_old.txt 2>&1 || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
  cp /tmp/new.so $L/libtlhip.so
fi
timeout -k 10 200 python bench.py $ARGS > gpurun_out/r02_ctx8k_new.json 2> gpurun_out/r02_ctx8k_new.err || exit 1
timeout -k 10 100 python tools/bench_attn.py > gpurun_out/r02_attn_new.txt 2>&1 || exit 1
python - <<'PY'
import json
for t in ("old", "new"):
    try:
        d = json.loads(open(f"gpurun_out/r02_ctx8k_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], "tok/s", d["ms_per_step"], "ms/step; eager per-kernel us:", d.get("eager_step_us"))
    except Exception as e:
        print(t, "n/a", e)
PY
cat gpurun_out/r02_attn_old.txt gpurun_out/r02_attn_new.txt 2>/dev/null
