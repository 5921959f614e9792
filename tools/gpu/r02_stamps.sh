# Round 2: per-phase stamp breakdown of the decode GEMV launches (first part)
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/stamp_gemv.py q4_k 4096 4096 q4_k 6144 4096 q4_k 28672 4096 q4_k 4096 14336 --json gpurun_out/r02_gemv_stamps${TAG}.jsonl > gpurun_out/r02_gemv_stamps${TAG}.txt 2>&1 || { tail -20 gpurun_out/r02_gemv_stamps${TAG}.txt; exit 1; }
grep -v "rep 0" gpurun_out/r02_gemv_stamps${TAG}.txt | head -80
