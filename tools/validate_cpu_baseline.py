#!/usr/bin/env python3
"""How fast is the CPU-baseline port against the reference's own code? (build container only: needs oracle/_ref)
Times, on one Llama-3-8B ffn_gate-shaped Q4_K tensor (14336 x 4096, 33 MB), the pair TinyLlamaModel::forward pays per
tensor and token: the reference's dequantize_vector_q4k_to_f32 + matvec_f32_f32_vector_cpu (timed INSIDE libtlref.so, no
marshalling copies) against the oracle port's tlo_dequantize_whole_tensor + tlo_matvec_f32, same threads. Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402
from oracle.oracle import Oracle, Ref, Q4_K  # noqa: E402

S = load_package().synth


def main():
    rows, cols, reps = 14336, 4096, 3
    rng = np.random.default_rng(0)
    blocks = S.random_blocks(Q4_K, rows * cols // 256, rng, w_std=0.02)
    x = rng.standard_normal(cols).astype(np.float32)
    ref, orc = Ref(), Oracle()
    rd, rm, ry = ref.time_dequant_matvec(Q4_K, blocks, x, rows, cols, reps)
    L = orc.L
    w = np.empty(rows * cols, np.float32)
    y = np.empty(rows, np.float32)
    fp = C.POINTER(C.c_float)
    L.tlo_dequantize_whole_tensor.argtypes = [C.c_int, C.c_void_p, C.c_size_t, fp]
    L.tlo_matvec_f32.argtypes = [fp, fp, fp, C.c_int, C.c_int]
    pd = pm = 1e30
    for _ in range(reps):
        w = np.empty(rows * cols, np.float32)  # a fresh buffer per repetition, like the reference's fresh vector
        t0 = time.perf_counter()
        L.tlo_dequantize_whole_tensor(Q4_K, blocks.ctypes.data, rows * cols // 256, w.ctypes.data_as(fp))
        t1 = time.perf_counter()
        L.tlo_matvec_f32(w.ctypes.data_as(fp), x.ctypes.data_as(fp), y.ctypes.data_as(fp), rows, cols)
        t2 = time.perf_counter()
        pd, pm = min(pd, t1 - t0), min(pm, t2 - t1)
    assert np.array_equal(ry.view(np.uint32), y.view(np.uint32)), "port and reference disagree"
    out = {"tensor": "q4_k 14336x4096", "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count())),
           "reference_dequant_ms": round(rd * 1e3, 1), "reference_matvec_ms": round(rm * 1e3, 1),
           "port_dequant_ms": round(pd * 1e3, 1), "port_matvec_ms": round(pm * 1e3, 1),
           "port_vs_reference": round((rd + rm) / (pd + pm), 3),
           "note": "port_vs_reference > 1: the port is faster than the reference's own functions by that factor"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
