#!/usr/bin/env python3
"""Batched-prefill throughput on one MI355X: tokens/s and matrix-core rate per (prompt length, MFMA terms).

    python tools/bench_prefill.py [--workload llama-3-8b] [--lens 128,512,2048] [--terms 6,3,1] [--reps 3]

Algorithmic GEMM flops per token = 2 * (weight elements of all layer projections) (the lm_head runs for the
last token only, as a GEMV); executed bf16 MFMA flops = (MFMAs per product: 3 / 2 / 1 for terms 6 / 3 / 1 with the integer-plane GEMM) x that."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="llama-3-8b")
    ap.add_argument("--mix", default="q4_k_m")
    ap.add_argument("--lens", default="128,512,2048")
    ap.add_argument("--terms", default="6,3,1")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--layers", type=int, default=0, help="cut the model to this many layers (profiling runs)")
    args = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine
    engine.init(0)
    cfg = dict(S.SHAPES[args.workload])
    lens = [int(v) for v in args.lens.split(",")]
    cfg["max_pos"] = min(cfg["max_pos"], max(lens) + 8)
    if args.layers:
        cfg["n_layers"] = args.layers
    m = engine.Model(cfg)
    layer_elems = 0
    for name, typ, rows, cols in S.tensor_plan(cfg, args.mix):
        m.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
        if name.startswith("blk.") and typ != S.F32:
            layer_elems += rows * cols
    m.finalize()
    rng = np.random.default_rng(0)
    out = []
    for T in lens:
        toks = rng.integers(0, cfg["vocab"], T).astype(np.int32)
        for terms in [int(v) for v in args.terms.split(",")]:
            m.set_prefill(terms=terms, min_batch=8)
            m.reset_kv()
            m.prefill(toks, 0)  # warm-up (workspace allocation, code load)
            best = 1e9
            for _ in range(args.reps):
                m.reset_kv()
                t0 = time.perf_counter()
                m.prefill(toks, 0)
                best = min(best, time.perf_counter() - t0)
            nh, hd = cfg["n_heads"], cfg["hidden"] // cfg["n_heads"]
            gemm_flops = 2.0 * T * layer_elems
            attn_flops = 4.0 * (T * (T + 1) / 2) * nh * hd * cfg["n_layers"]
            row = {"T": T, "terms": terms, "ms": round(best * 1e3, 2), "tokens_per_s": round(T / best, 1),
                   "gemm_tflops_algorithmic": round(gemm_flops / best / 1e12, 1),
                   "mfma_bf16_tflops_executed": round(gemm_flops * ({6: 3, 3: 2, 1: 1}[terms] if os.environ.get("TL_GEMM_INT", "1") != "0" else terms) / best / 1e12, 1),
                   "attn_gflop": round(attn_flops / 1e9, 1)}
            out.append(row)
            print(json.dumps(row), flush=True)
    return out


if __name__ == "__main__":
    main()
