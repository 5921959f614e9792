#!/bin/bash
# SQ stall breakdown of the direct small-token GEMM (gemm_qd_kernel) of the 32-sequence decode step.
# Round 1 ran the whole batch bench under ONE --pmc pass of 8 SQ counters and rocprofv3 died with SIGSEGV. Round 2 found
# the same fault with FOUR counters: it is raised inside a kernel launch of the eager decode step (stack: memmove under
# the profiler's dispatch callback under gemv_launch_pair, profiles/r02_f_pmc_qd_abort.txt), so it is not the counter
# set. Here the profiled program is the smallest one that launches the kernel (tools/pmc_qd_min.py: six launches), in two
# passes of 4-5 counters, each a run of its own with the program directly after `--`; then, last, all 8 counters in one
# pass on the same small program.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export TL_NO_GRAPH=1 TL_BATCH_GEMV_MAX=0
summ() {
python3 - "$1" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not fs:
    print("no counter file under", sys.argv[1]); sys.exit(1)
agg = {}
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if "gemm_qd_kernel" not in k: continue
    key = k[k.index("gemm_qd"):][:32] + " grid=" + r.get("Grid_Size", "?")
    c = agg.setdefault(key, {}).setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
for k, a in sorted(agg.items()):
    row = {n: round(v / c) for n, (c, v) in a.items()}
    wc = row.get("SQ_WAVE_CYCLES")
    print(k, {n: (v, round(v / wc, 3)) if wc else v for n, v in row.items()})
PY
}
P="python3 tools/pmc_qd_min.py 32"
rm -rf gpurun_out/pmc_qd1 gpurun_out/pmc_qd2 gpurun_out/pmc_qd8
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_qd1 -- $P > gpurun_out/r02_pmc_qd_pass1.log 2> gpurun_out/r02_pmc_qd_pass1.err || exit 1
summ gpurun_out/pmc_qd1 > gpurun_out/r02_pmc_qd_pass1.txt || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_qd2 -- $P > gpurun_out/r02_pmc_qd_pass2.log 2> gpurun_out/r02_pmc_qd_pass2.err || exit 1
summ gpurun_out/pmc_qd2 > gpurun_out/r02_pmc_qd_pass2.txt || exit 1
cat gpurun_out/r02_pmc_qd_pass1.txt gpurun_out/r02_pmc_qd_pass2.txt
rm -rf gpurun_out/pmc_qd1 gpurun_out/pmc_qd2
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_qd8 -- $P > gpurun_out/r02_pmc_qd_8counters.log 2> gpurun_out/r02_pmc_qd_8counters.err
rc=$?
echo "8-counter pass exit code $rc" | tee gpurun_out/r02_pmc_qd_8counters.rc
[ $rc -eq 0 ] && summ gpurun_out/pmc_qd8 > gpurun_out/r02_pmc_qd_8counters.txt
rm -rf gpurun_out/pmc_qd8
exit 0
