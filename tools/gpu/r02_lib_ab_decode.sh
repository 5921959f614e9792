#!/bin/bash
# A/B of two builds of the library on the default decode bench: csrc/libtlhip_base.so (kept aside by hand) against libtlhip.so
cd $GRAFT_REPO_ROOT
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/new.so || exit 1
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefill-leg --no-batch-leg > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; return 1; }
  python - <<'PY'
import json
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(d["value"], "tok/s", {k: v["us"] for k, v in d["gemv_kernels"].items()}, "attention", d["eager_step_us"].get("attention"))
PY
}
for r in 1 2; do
  echo "== base"; cp $L/libtlhip_base.so $L/libtlhip.so; run || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
  echo "== new"; cp /tmp/new.so $L/libtlhip.so; run || exit 1
done
