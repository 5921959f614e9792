#!/bin/bash
# Round profile: (1) rocprofv3 kernel trace + stats of the default bench (eager replay: TL_NO_GRAPH=1), (2) a separate
# --pmc pass for HBM traffic (FETCH_SIZE) of the same command. Writes gpurun_out/r03_prof/{decode_kernel_stats.csv,
# bench_under_profiler.json, gate_up_traffic.json}; gate_up_traffic.json carries the hash of the GEMV sources it was
# measured on and the gate/up kernel's average duration from (1): copy it to profiles/latest_gate_up_traffic.json, which is
# what bench.py reads for roofline.traffic.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_prof
rm -rf $O && mkdir -p $O
TL_NO_GRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --steps 64 --warmup 8 --batch-seqs 32 > $O/bench_under_profiler.json 2> $O/bench_under_profiler.err || exit 1
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/decode_kernel_stats.csv
head -12 $O/decode_kernel_stats.csv
rm -rf $O/trace
TL_NO_GRAPH=1 timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python3 bench.py --no-cpu-baseline --no-batch-leg --no-prefill-leg --steps 8 --warmup 2 > $O/pmc_bench.json 2> $O/pmc_bench.err || exit 1
python3 - <<'PY'
import csv, glob, hashlib, json, os
O = "gpurun_out/r03_prof"
f = glob.glob(O + "/pmc/*/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    a = agg.setdefault(r["Kernel_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    kb = v / n  # FETCH_SIZE is in KiB
    print(f"{n:6d} launches  FETCH_SIZE {kb:12.1f} KiB/launch  x2 (gfx950 correction) = {2*kb*1024/1e6:9.2f} MB  {k[:90]}")
    if "gemv_kernel<12, -1, true, 3" in k:
        out = {"kernel": k, "launches": n, "fetch_size_kib_per_launch": kb, "hbm_bytes_per_launch": int(2 * kb * 1024),
               "note": "FETCH_SIZE x 1024 x 2: gfx950 counts a 128-B request as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM)"}
for r in csv.DictReader(open(O + "/decode_kernel_stats.csv")):
    if "gemv_kernel<12, -1, true, 3" in r["Name"]:
        out["rocprof_avg_launch_us"] = round(float(r["AverageNs"]) / 1e3, 3)
        out["rocprof_calls"] = int(r["Calls"])
h = hashlib.sha256()
for s in ("tl_gemv.hip", "tl_chunk.h", "tl_internal.h"):
    h.update(open(os.path.join("tinyllama.cpp_amd", "csrc", s), "rb").read())
out["gemv_sources_sha256"] = h.hexdigest()
json.dump(out, open(O + "/gate_up_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/pmc
