#!/bin/bash
# the one-layer pre-flight model between 2 processes on one GPU: step time by exchange form and GEMV grid cap
cd $GRAFT_REPO_ROOT
for C in 64 16 4; do for F in 1 0; do
  echo "== TL_GEMV_CUS=$C TL_TP_PEER_FUSED=$F"
  TL_GEMV_CUS=$C TL_TP_PEER_FUSED=$F timeout -k 10 120 python tools/peer_latency.py --tp 2 --iters 200 || exit 1
done; done
