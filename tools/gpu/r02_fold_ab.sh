#!/bin/bash
# bench.py --gpus N launched the way the driver launches it, all ranks on the box's one GPU (control-flow rehearsal), with the
# exchanges folded into the GEMV launches (default) and as launches of their own (TL_TP_PEER_FUSED=0)
cd $GRAFT_REPO_ROOT
for N in 2 4; do
  for F in 1 0; do
    o=gpurun_out/fold_n${N}_f${F}
    echo "run N=$N F=$F"; TL_BENCH_WATCHDOG=100 TL_TP_PEER_FUSED=$F TL_BENCH_SHARE_GPU=1 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 64 --warmup 8 --no-cpu-baseline --no-prefill-leg > $o.json 2> $o.err || { grep -v "amdgpu.ids\|hostname of the client" $o.err | tail -60; exit 1; }
    python - $o.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ranks", "via", "value", "ms_per_step", "launches_per_step", "exchange_launches_per_step", "all_reduce_us", "graph_captured")})
PY
  done
done
