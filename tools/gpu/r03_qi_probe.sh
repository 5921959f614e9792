#!/bin/bash
# timing-only probes of the integer-plane prefill GEMM (results are WRONG in the hacked builds): which role bounds a step?
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/keep.so || exit 1
for v in ${VARIANTS:-base hack1 hack2 hack3}; do
  echo "== $v"; cp $L/libtlhip_$v.so $L/libtlhip.so
  timeout -k 10 250 python tools/bench_prefill.py --layers 8 --lens 512 --terms 6 2>/dev/null | cut -c1-90 || { cp /tmp/keep.so $L/libtlhip.so; exit 1; }
done
cp /tmp/keep.so $L/libtlhip.so
