#!/bin/bash
# A/B of one environment switch on the default bench's decode leg, alternating runs inside one GPU call.
# usage: tools/gpu/r03_env_ab.sh VAR OUTDIR [extra bench args...]
set -u
VAR=$1; OUT=$2; shift 2
mkdir -p "$OUT"
for v in 0 1 0 1; do
  env "$VAR=$v" timeout -k 10 200 python bench.py --no-prefill-leg --no-batch-leg --no-cpu-baseline --steps 256 --warmup 32 "$@" \
    > "$OUT/b_$v.json" 2> "$OUT/b_$v.err" || { tail -5 "$OUT/b_$v.err"; exit 1; }
  python - "$OUT/b_$v.json" "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step", flush=True)
PY
done
