#!/usr/bin/env python3
"""Per-rank GEMV kernel times of a tensor-parallel shard of the headline workload on ONE GPU (no collectives): what the
compute side of a decode step costs per rank at TP = 1, 2, 4, 8. (tl_model_bench_kernel on rank 0's shard.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg = dict(S.SHAPES["llama-3-8b"], max_pos=256)
    L = cfg["n_layers"]
    for tp in (1, 2, 4, 8):
        m = E.Model(cfg, tp_rank=0, tp_size=tp)
        for name, typ, rows, cols in S.tensor_plan(cfg, "q4_k_m"):
            m.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
        m.finalize()
        row, per_layer = {"tp": tp}, 0.0
        for k in ("qkv_gemv", "attn_out_gemv", "gate_up_gemv", "down_gemv", "lm_head_gemv"):
            us, nb = m.bench_kernel(k, iters=64 if k == "lm_head_gemv" else 4 * L, warmup=8)
            row[k] = round(us, 2)
            if k != "lm_head_gemv":
                per_layer += us
        row["gemv_us_per_layer"] = round(per_layer, 2)
        row["gemv_us_per_step"] = round(per_layer * L + row["lm_head_gemv"], 1)
        print(json.dumps(row), flush=True)
        m.free()


if __name__ == "__main__":
    main()
