#!/bin/bash
# A/B of one environment switch (VAR=NAME, values 0 / 1) on the prefill bench: tools/bench_prefill.py, 8 layers
set -o pipefail
cd "$(dirname "$0")/../.."
for v in 0 1 0 1; do
  echo "== ${VAR}=$v"
  env "${VAR}=$v" timeout -k 10 250 python tools/bench_prefill.py --layers 8 --lens ${LENS:-128,512,2048} --terms 6 2>/dev/null | cut -c1-90 || exit 1
done
