# A/B of the flattened GEMV step scheme (TL_GEMV_FLAT=0: spans only) on the shapes it applies to
cd $GRAFT_REPO_ROOT
for flat in 0 1; do
  for w in llama-3.2-3b tinyllama-1.1b; do
    TL_GEMV_FLAT=$flat timeout -k 10 300 python bench.py --workload $w --steps 128 --warmup 16 --no-cpu-baseline --no-prefill-leg > gpurun_out/flat${flat}_$w.json 2> gpurun_out/flat${flat}_$w.err || exit 1
    python3 -c "
import json
d=json.loads(open('gpurun_out/flat${flat}_$w.json').read().strip().splitlines()[-1])
print('flat=$flat $w', d['value'], 'tok/s', d['ms_per_step'], 'ms', {k: v['us'] for k, v in d['gemv_kernels'].items()})"
  done
done
