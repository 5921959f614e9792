#!/bin/bash
# the N = 2 rehearsal with folded exchanges, several times over (a watchdog dumps the Python stacks of a run that stalls)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  o=gpurun_out/fold_rep_$i
  echo "== run $i"
  TL_BENCH_WATCHDOG=100 TL_TP_PEER_FUSED=1 TL_BENCH_SHARE_GPU=1 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 64 --warmup 8 --no-cpu-baseline --no-prefill-leg > $o.json 2> $o.err || { grep -v "amdgpu.ids\|hostname of the client" $o.err | tail -80; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$o.json').read().strip().splitlines()[-1]); print(d['value'], d['via'], d['ms_per_step'])"
done
