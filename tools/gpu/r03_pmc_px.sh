#!/bin/bash
# SQ counters of the expanded-plane GEMM (gemm_px_kernel; T tokens, default 512; the two FFN shapes of Llama-3-8B) from the
# program (tools/pmc_qd_min.py 512), two passes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
summ() {
python3 - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
agg = {}
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "gemm_px" not in k: continue
    key = k[k.index("gemm_px"):][:36] + " grid=" + r.get("Grid_Size", "?")
    c = agg.setdefault(key, {}).setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
for k, a in sorted(agg.items()):
    row = {n: round(v / c) for n, (c, v) in a.items()}
    wc = row.get("SQ_WAVE_CYCLES")
    print(k, {n: (v, round(v / wc, 3)) if wc else v for n, v in row.items()})
PY
}
P="python3 tools/pmc_qd_min.py ${T:-512} px"
rm -rf gpurun_out/pmc_px1 gpurun_out/pmc_px2 gpurun_out/pmc_px3
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_px1 -- $P > gpurun_out/r03_pmc_px.log 2> gpurun_out/r03_pmc_px.err || exit 1
summ gpurun_out/pmc_px1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc_px2 -- $P >> gpurun_out/r03_pmc_px.log 2>> gpurun_out/r03_pmc_px.err || exit 1
summ gpurun_out/pmc_px2
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/pmc_px3 -- $P >> gpurun_out/r03_pmc_px.log 2>> gpurun_out/r03_pmc_px.err || exit 1
summ gpurun_out/pmc_px3
rm -rf gpurun_out/pmc_px1 gpurun_out/pmc_px2 gpurun_out/pmc_px3
