# kernel trace of the multi-sequence decode with 128 sequences; decode-step kernels only (the trace's rows whose call count is a
# multiple of the step count are the step's; prefill / expansion kernels are listed too)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_b128
TL_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b128 -- python3 tools/bench_batch.py --batch 128 --terms 6 --steps 16 > gpurun_out/prof_b128.jsonl 2> gpurun_out/prof_b128.err || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof_b128/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 16 decode steps: find the last 16 * per-step kernels by walking back from the end over argmax kernels
idx = [i for i, r in enumerate(rows) if "batch_argmax_advance" in r["Kernel_Name"]]
lo, hi = idx[-9], idx[-1]   # 8 whole steps
agg = collections.OrderedDict()
for r in rows[lo + 1:hi + 1]:
    k = (r["Kernel_Name"][:90], r["Grid_Size_X"], r["Workgroup_Size_X"])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
span = int(rows[hi]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])
print(f"8 eager steps: kernel time {tot / 8e3:.1f} us per step, wall {span / 8e3:.1f} us per step")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0] / 8:7.1f} per step  {a[1] / a[0] / 1e3:8.2f} us  {a[1] / 8e3:8.1f} us/step  {100 * a[1] / tot:5.1f}%  {k}")
PY
