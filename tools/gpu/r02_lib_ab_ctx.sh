#!/bin/bash
# A/B of two library builds (csrc/libtlhip_base.so against libtlhip.so) on the decode bench at short context and at 2000 / 8000 positions
cd $GRAFT_REPO_ROOT
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/new.so || exit 1
A="--no-cpu-baseline --no-prefill-leg --no-batch-leg --steps 128 --warmup 16"
run() {
  for p in 4 2000 8000; do
    timeout -k 10 200 python bench.py $A --max-pos 8192 --prompt-len $p > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; return 1; }
    python - $p <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("  prompt", sys.argv[1], d["value"], "tok/s", "attention (eager)", d["eager_step_us"].get("attention"))
PY
  done
}
for r in 1 2; do
  echo "== base"; cp $L/libtlhip_base.so $L/libtlhip.so; run || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
  echo "== new"; cp /tmp/new.so $L/libtlhip.so; run || exit 1
done
