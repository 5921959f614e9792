cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_attn
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_attn -- python3 tools/bench_attn.py > gpurun_out/prof_attn.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_attn/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"]]
# the tool runs 6 context lengths x (16 warm-up + 320 timed) calls, in order
for name in ("attn_decode_split", "attn_combine"):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])) if name in r["Kernel_Name"]]
    per = len(d) // 6
    for i, n in enumerate((128, 512, 1024, 2048, 4096, 8192)):
        v = sorted(d[i * per + 16:(i + 1) * per])
        kvb = 2 * n * 1024 * 4
        print(f"{name:18s} len {n:5d}: median {v[len(v)//2] / 1e3:6.2f} us  min {v[0] / 1e3:6.2f} us   {kvb / (v[len(v)//2] * 1e-9) / 1e12 if name.startswith('attn_decode') else 0:5.2f} TB/s")
PY
rm -rf gpurun_out/prof_attn
