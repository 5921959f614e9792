# SQ stall breakdown of the prefill GEMM (one --pmc pass, 8 SQ counters)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm
TL_NO_GRAPH=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_gemm -- python3 tools/bench_prefill.py --layers 2 --lens 512 --terms ${TERMS:-6} --reps 1 > gpurun_out/pmc_gemm.log 2> gpurun_out/pmc_gemm.err || exit 1
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/pmc_gemm/*/*counter_collection.csv")[0]
agg = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm_q_kernel" not in k: continue
    a = agg.setdefault(k[:60], {})
    c = a.setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
for k, a in agg.items():
    row = {n: round(v / c) for n, (c, v) in a.items()}
    wc = row.get("SQ_WAVE_CYCLES", 1)
    print(k, {n: (v, round(v / wc, 3)) for n, v in row.items()})
PY
rm -rf gpurun_out/pmc_gemm
