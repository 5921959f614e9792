#!/bin/bash
# A/B (or timing-only probes: results WRONG in those builds) of alternate builds of the library on long-context decode:
# the 8B shape at positions 8000.. (eager per-kernel times) and the operator-level microbenchmark. VARIANTS="base x y"
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
ARGS="--workload llama-3-8b --max-pos 8192 --prompt-len 8000 --steps 64 --warmup 8 --no-cpu-baseline --no-prefill-leg --no-batch-leg"
cp $L/libtlhip.so /tmp/keep.so || exit 1
for v in ${VARIANTS:-base}; do
  echo "== $v"; cp $L/libtlhip_$v.so $L/libtlhip.so
  timeout -k 10 200 python bench.py $ARGS > /tmp/ctx_$v.json 2> /tmp/ctx_$v.err || { tail -5 /tmp/ctx_$v.err; cp /tmp/keep.so $L/libtlhip.so; exit 1; }
  python -c "
import json,sys
d=json.loads(open('/tmp/ctx_$v.json').read().strip().splitlines()[-1])
print(d['value'],'tok/s',d['ms_per_step'],'ms/step; eager us:',d.get('eager_step_us'))"
  if [ -z "$NO_OP" ]; then timeout -k 10 100 python tools/bench_attn.py 2>&1 | grep -E "len +(2048|8192)" || { cp /tmp/keep.so $L/libtlhip.so; exit 1; }; fi
done
cp /tmp/keep.so $L/libtlhip.so
