#!/usr/bin/env python3
"""GEMV kernel times of the headline workload with the weights streamed from HBM (cycling over all layers) against the
same launches on ONE layer's weights, which stay in the 256 MiB Infinity Cache: what a prefetch of the next
projection's weights could buy. (tl_model_bench_kernel, TL_BENCH_LAYER_SPAN.)"""
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
    m = E.Model(cfg)
    for name, typ, rows, cols in S.tensor_plan(cfg, "q4_k_m"):
        m.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
    m.finalize()
    for span in (L, 4, 1):
        os.environ["TL_BENCH_LAYER_SPAN"] = str(span)
        row = {"layer_span": span}
        for k in ("qkv_gemv", "attn_out_gemv", "gate_up_gemv", "down_gemv"):
            us, nb = m.bench_kernel(k, iters=8 * L, warmup=L)
            row[k] = {"us": round(us, 2), "gbps": round(nb / us / 1e3, 1)}
        print(json.dumps(row), flush=True)
    m.free()


if __name__ == "__main__":
    main()
