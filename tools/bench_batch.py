#!/usr/bin/env python3
"""Aggregate decode tokens/s of the multi-sequence path (tl_model_batch_generate_greedy) on the Llama-3-8B-shaped
synthetic Q4_K_M model: B sequences, every step streams the weights once for all of them (skinny GEMMs on the matrix
cores). One JSON line per (B, terms). usage: python tools/bench_batch.py [--batch 8 32 128] [--terms 6 1] [--steps 16]"""
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
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 32, 64, 128, 256])
    ap.add_argument("--terms", type=int, nargs="+", default=[6, 3, 1])
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--max-pos", type=int, default=512)
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg = dict(S.SHAPES[a.workload])
    cfg["max_pos"] = min(cfg["max_pos"], a.max_pos)
    m = E.Model(cfg)
    for name, typ, rows, cols in S.tensor_plan(cfg, a.mix):
        m.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
    m.finalize()
    single = None
    ids = m.generate_greedy(1, 0, 8)
    t0 = time.perf_counter()
    m.generate_greedy(int(ids[-1]), 8, 64)
    single = 64 / (time.perf_counter() - t0)
    rng = np.random.default_rng(0)
    for B in a.batch:
        m.batch_init(B)
        prompts = [rng.integers(0, cfg["vocab"], a.prompt_len).astype(np.int32) for _ in range(B)]
        for terms in a.terms:
            m.set_prefill(terms=terms, min_batch=8)
            t0 = time.perf_counter()
            _, first = m.batch_prefill_packed(prompts, want_logits=False)
            t_prefill = time.perf_counter() - t0
            m.batch_generate_greedy(first, [a.prompt_len] * B, list(range(B)), 2)  # warm-up (rows are overwritten below)
            t0 = time.perf_counter()
            out = m.batch_generate_greedy(first, [a.prompt_len] * B, list(range(B)), a.steps)
            dt = time.perf_counter() - t0
            assert out.shape == (a.steps, B)
            print(json.dumps({"workload": f"{a.workload} {a.mix} multi-sequence greedy decode", "sequences": B,
                              "mfma_terms": terms, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 3),
                              "tokens_per_s_aggregate": round(B * a.steps / dt, 1),
                              "single_sequence_tokens_per_s": round(single, 1),
                              "packed_prefill_ms": round(t_prefill * 1e3, 2),
                              "positions": f"{a.prompt_len}..{a.prompt_len + a.steps - 1}"}), flush=True)
    m.free()


if __name__ == "__main__":
    main()
