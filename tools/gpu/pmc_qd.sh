# SQ stall breakdown of the direct small-token GEMM inside a 32-sequence decode step (one --pmc pass, 8 SQ counters)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_qd
TL_NO_GRAPH=1 TL_BATCH_GEMV_MAX=0 timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_qd -- python3 tools/bench_batch.py --batch 32 --terms 6 --steps 2 > gpurun_out/pmc_qd.log 2> gpurun_out/pmc_qd.err || exit 1
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/pmc_qd/*/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm_qd_kernel" not in k: continue
    key = k[k.index("gemm_qd"):][:32] + " grid=" + r.get("Grid_Size", "?")
    a = agg.setdefault(key, {})
    c = a.setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
for k, a in sorted(agg.items()):
    row = {n: round(v / c) for n, (c, v) in a.items()}
    wc = row.get("SQ_WAVE_CYCLES", 1)
    print(k, {n: (v, round(v / wc, 3)) for n, v in row.items()})
PY
rm -rf gpurun_out/pmc_qd
