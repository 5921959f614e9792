#!/bin/bash
# A/B of two library builds (csrc/libtlhip_base.so against libtlhip.so) on the Llama-3.2-3B decode configuration (3 heads per workgroup)
cd $GRAFT_REPO_ROOT
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/new.so || exit 1
run() {
  timeout -k 10 200 python bench.py --workload llama-3.2-3b --no-cpu-baseline --no-prefill-leg --no-batch-leg --steps 128 --warmup 16 > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; return 1; }
  python -c "
import json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('  3b', d['value'], 'tok/s attention (eager)', d['eager_step_us'].get('attention'))"
}
for r in 1 2; do
  echo "== base"; cp $L/libtlhip_base.so $L/libtlhip.so; run || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
  echo "== new"; cp /tmp/new.so $L/libtlhip.so; run || exit 1
done
