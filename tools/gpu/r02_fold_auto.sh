#!/bin/bash
# as rehearse.sh, exchange form not pinned: the pre-flight children time both forms and the ranks take the faster
cd $GRAFT_REPO_ROOT
N=${N:-2}
o=gpurun_out/fold_auto_n${N}
TL_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 64 --warmup 8 --no-cpu-baseline > $o.json 2> $o.err || { tail -20 $o.err; exit 1; }
cut -c1-1200 $o.json
