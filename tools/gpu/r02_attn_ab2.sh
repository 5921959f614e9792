#!/bin/bash
# short-context A/B: headline bench (positions 4..) and the operator microbenchmark, previous library vs current
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
ARGS="--steps 128 --warmup 16 --no-cpu-baseline --no-prefill-leg --no-batch-leg"
cp $L/libtlhip.so /tmp/new.so || exit 1
cp $L/libtlhip_old.so $L/libtlhip.so
timeout -k 10 200 python bench.py $ARGS > gpurun_out/r02_short_old.json 2> gpurun_out/r02_short_old.err || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
cp /tmp/new.so $L/libtlhip.so
timeout -k 10 200 python bench.py $ARGS > gpurun_out/r02_short_new.json 2> gpurun_out/r02_short_new.err || exit 1
timeout -k 10 100 python tools/bench_attn.py > gpurun_out/r02_attn_new.txt 2>&1 || exit 1
python - <<'PY'
import json
for t in ("old", "new"):
    d = json.loads(open(f"gpurun_out/r02_short_{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], "tok/s", d["ms_per_step"], "ms/step; eager per-kernel us:", d.get("eager_step_us"))
PY
cat gpurun_out/r02_attn_new.txt
