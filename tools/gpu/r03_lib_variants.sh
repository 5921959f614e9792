#!/bin/bash
# A/B of alternate builds of the library (tinyllama.cpp_amd/csrc/libtlhip_<v>.so) on a command: VARIANTS="base x" CMD="python tools/..."
set -o pipefail
cd "$(dirname "$0")/../.."
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/keep.so || exit 1
for v in ${VARIANTS:-base}; do
  echo "== $v"; cp $L/libtlhip_$v.so $L/libtlhip.so
  timeout -k 10 ${LIMIT:-250} $CMD 2>/dev/null | cut -c1-${CUT:-110} || { cp /tmp/keep.so $L/libtlhip.so; exit 1; }
done
cp /tmp/keep.so $L/libtlhip.so
