#!/bin/bash
# kernel-trace statistics of one kernel (PATTERN) for alternate builds of the library: VARIANTS="base x" PATTERN=attn_prefill
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=tinyllama.cpp_amd/csrc
cp $L/libtlhip.so /tmp/keep.so || exit 1
for v in ${VARIANTS:-base}; do
  cp $L/libtlhip_$v.so $L/libtlhip.so
  rm -rf /tmp/tv_$v
  TL_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tv_$v -- python3 tools/bench_prefill.py --layers 4 --lens ${LENS:-512} --terms 6 --reps 3 > /tmp/tv_$v.log 2> /tmp/tv_$v.err || { tail -5 /tmp/tv_$v.err; cp /tmp/keep.so $L/libtlhip.so; exit 1; }
  echo "== $v"; grep -E "${PATTERN:-attn_prefill}" $(ls /tmp/tv_$v/*/*kernel_stats.csv | head -1) | cut -d, -f1-6 | cut -c1-60,200-
  python3 - "$v" <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/tv_{sys.argv[1]}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "attn_prefill" in r["Name"] or "gemm_px" in r["Name"]:
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
cp /tmp/keep.so $L/libtlhip.so
