#!/bin/bash
# the opt-in step structures (TL_CHAIN=1 grid form, TL_FUSED_ATTN=1) on the launch-bound small models
set -o pipefail
cd "$(dirname "$0")/../.."
A="--no-cpu-baseline --no-batch-leg --no-prefill-leg --steps 128 --warmup 16"
for wl in "tinyllama-1.1b --mix q4_k_m" "llama-3.2-3b"; do
  for cfg in "TL_CHAIN=0" "TL_CHAIN=1" "TL_FUSED_ATTN=1" "TL_CHAIN=0"; do
    env $cfg timeout -k 10 200 python bench.py --workload $wl $A > /tmp/sm.json 2> /tmp/sm.err || { tail -3 /tmp/sm.err; exit 1; }
    python -c "
import json
d=json.loads(open('/tmp/sm.json').read().strip().splitlines()[-1])
print('$wl', '$cfg', d['value'], 'tok/s')"
  done
done
