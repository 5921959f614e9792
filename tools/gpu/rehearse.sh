# bench.py --gpus ${N:-2} launched the way the driver launches it, all ranks on the box's one GPU (control-flow rehearsal)
cd $GRAFT_REPO_ROOT
TL_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${N:-2} --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus ${N:-2} --steps 32 --warmup 4 > gpurun_out/rehearse_n${N:-2}.json 2> gpurun_out/rehearse_n${N:-2}.err || { tail -20 gpurun_out/rehearse_n${N:-2}.err; exit 1; }
cat gpurun_out/rehearse_n${N:-2}.json | cut -c1-900
