#!/bin/bash
# A/B of one environment switch (default TL_PREFILL_PLANES = 0 / 1: the expanded prefill planes) on the default bench's
# prefill and multi-sequence legs.
# usage: [VAR=NAME] tools/gpu/r03_px_ab.sh OUTDIR [extra bench args...]
set -u
VAR=${VAR:-TL_PREFILL_PLANES}
OUT=$1; shift
mkdir -p "$OUT"
for v in 0 1 0 1; do
  env "$VAR=$v" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 64 --warmup 8 "$@" \
    > "$OUT/b_$v.json" 2> "$OUT/b_$v.err" || { tail -5 "$OUT/b_$v.err"; exit 1; }
  python - "$OUT/b_$v.json" "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pf, ms = d.get("prefill", {}), d.get("multi_sequence", {})
print(sys.argv[2], "decode", d["value"], "tok/s | prefill", {k: pf.get(k) for k in ("prompt_tokens", "tokens_per_s", "ms", "error")},
      "| multi-sequence", {k: ms.get(k) for k in ("sequences", "tokens_per_s_aggregate", "ms_per_step", "error")}, flush=True)
PY
done
