# The other BASELINE.json configurations on one GPU (decode), plus long-context decode of the headline workload.
cd $GRAFT_REPO_ROOT
for w in tinyllama-1.1b llama-3.2-3b; do
  timeout -k 10 300 python bench.py --workload $w --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/cfg_$w.json 2> gpurun_out/cfg_$w.err || exit 1
done
timeout -k 10 300 python bench.py --workload llama-3-8b --max-pos 8192 --prompt-len 2000 --steps 128 --warmup 16 --no-cpu-baseline --no-prefill-leg > gpurun_out/cfg_8b_ctx2k.json 2> gpurun_out/cfg_8b_ctx2k.err || exit 1
timeout -k 10 300 python bench.py --workload llama-3-8b --max-pos 8192 --prompt-len 8000 --steps 128 --warmup 16 --no-cpu-baseline --no-prefill-leg > gpurun_out/cfg_8b_ctx8k.json 2> gpurun_out/cfg_8b_ctx8k.err || exit 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/cfg_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], "tok/s", d["ms_per_step"], "ms", d["hbm_gbps_whole_step"], "GB/s", d["roofline"]["frac"], d.get("prefill", {}).get("tokens_per_s"))
PY
