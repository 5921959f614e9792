# Prefill profile: (1) rocprofv3 kernel trace + stats of a 512-token batched prefill (Llama-3-8B shapes cut to 4 layers so
# that the run stays short; every layer launches the same kernels), (2) a separate --pmc pass for matrix-core busy cycles.
# Outputs under gpurun_out/ (copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pf gpurun_out/prof_pf_pmc
TL_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pf -- python3 tools/bench_prefill.py --layers 4 --lens 512 --terms 6 --reps 3 > gpurun_out/prof_pf.log 2> gpurun_out/prof_pf.err || exit 1
cp $(ls gpurun_out/prof_pf/*/*kernel_stats.csv | head -1) gpurun_out/prof_prefill_kernel_stats.csv
head -14 gpurun_out/prof_prefill_kernel_stats.csv
rm -rf gpurun_out/prof_pf
TL_NO_GRAPH=1 timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_pf_pmc -- python3 tools/bench_prefill.py --layers 4 --lens 512 --terms 6 --reps 1 > gpurun_out/prof_pf_pmc.log 2> gpurun_out/prof_pf_pmc.err || exit 1
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/prof_pf_pmc/*/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    a = agg.setdefault(k, {})
    c = a.setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
out = []
for k, a in agg.items():
    if "gemm_q" not in k and "gemm_px" not in k and "attn_prefill" not in k: continue
    row = {"kernel": k[:100]}
    for name, (n, v) in a.items():
        row[name] = v / n; row["launches"] = n
    # SQ_VALU_MFMA_BUSY_CYCLES: 32 per 32x32x16 bf16 MFMA (64 per 32x32x2 f32), summed over the chip's 1024 SIMDs;
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel's busy time in cycles is GRBM_GUI_ACTIVE / 8
    if row.get("GRBM_GUI_ACTIVE"):
        row["kernel_cycles"] = row["GRBM_GUI_ACTIVE"] / 8
        row["mfma_util"] = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (row["kernel_cycles"] * 1024)
    out.append(row)
    print(json.dumps(row))
json.dump(out, open("gpurun_out/prefill_mfma_counters.json", "w"), indent=1)
PY
rm -rf gpurun_out/prof_pf_pmc
