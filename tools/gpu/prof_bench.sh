# rocprofv3 kernel trace + stats of the default bench; summary copied under gpurun_out/ for profiles/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
TL_NO_GRAPH=${TL_NO_GRAPH:-1} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu-baseline --steps 64 --warmup 8 > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || exit 1
f=$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1); cp $f gpurun_out/prof_bench_kernel_stats.csv; cat $f
t=$(ls gpurun_out/prof_bench/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 2000 dispatches: steady-state graph replays
tail = rows[-2000:]
gaps = {}
for a, b in zip(tail, tail[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    k = (a["Kernel_Name"][:40], b["Kernel_Name"][:40])
    gaps.setdefault(k, []).append(g)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print(f"steady state: {len(tail)} dispatches, span {span/1e3:.1f} us, kernel busy {busy/1e3:.1f} us ({100*busy/span:.1f}%), mean gap {(span-busy)/len(tail):.0f} ns")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  gap {sum(v)/len(v):7.0f} ns x {len(v):4d}  {k[0]} -> {k[1]}")
PY
rm -rf gpurun_out/prof_bench
