# Round profile: (1) rocprofv3 kernel trace + stats of the default bench (eager replay: the profiler cannot follow graph
# launches of this library), (2) a separate --pmc pass for HBM traffic (FETCH_SIZE) of the same command.
# Outputs under gpurun_out/ (copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench gpurun_out/prof_pmc
TL_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu-baseline --steps 64 --warmup 8 --batch-seqs 32 > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || exit 1
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/prof_bench_kernel_stats.csv
head -12 gpurun_out/prof_bench_kernel_stats.csv
rm -rf gpurun_out/prof_bench
TL_NO_GRAPH=1 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc -- python3 bench.py --no-cpu-baseline --no-batch-leg --steps 8 --warmup 2 > gpurun_out/prof_pmc.json 2> gpurun_out/prof_pmc.err || exit 1
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/prof_pmc/*/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    kb = v / n  # FETCH_SIZE is in KiB
    print(f"{n:6d} launches  FETCH_SIZE {kb:12.1f} KiB/launch  x2 (gfx950 correction) = {2*kb*1024/1e6:9.2f} MB  {k[:90]}")
    if "true, 3" in k or "GATE_UP" in k:
        out = {"kernel": k, "launches": n, "fetch_size_kib_per_launch": kb, "hbm_bytes_per_launch": int(2 * kb * 1024),
               "note": "FETCH_SIZE x 1024 x 2: gfx950 counts a 128-B request as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM)"}
json.dump(out, open("gpurun_out/gate_up_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/prof_pmc
