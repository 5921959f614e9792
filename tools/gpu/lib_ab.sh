#!/bin/bash
# A/B of two builds of the library inside one call: csrc/libtlhip_base.so (kept aside by hand) against csrc/libtlhip.so,
# prefill at 128 / 512 / 2048 tokens and multi-sequence decode at 64 / 128 sequences
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/new.so || exit 1
run() {
  timeout -k 10 250 python tools/bench_prefill.py --lens 128,512,2048 --terms 6 2>/dev/null | cut -c1-60 || return 1
  timeout -k 10 250 python tools/bench_batch.py --batch 32 64 128 --terms 6 --steps 16 2>/dev/null | cut -c66-150 || return 1
}
echo "== base"; cp $L/libtlhip_base.so $L/libtlhip.so; run || { cp /tmp/new.so $L/libtlhip.so; exit 1; }
echo "== new"; cp /tmp/new.so $L/libtlhip.so; run || exit 1
echo "== base again"; cp $L/libtlhip_base.so $L/libtlhip.so; run; cp /tmp/new.so $L/libtlhip.so
