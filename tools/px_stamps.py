#!/usr/bin/env python3
"""Per-wave timeline of one workgroup of the expanded-plane GEMM (gemm_px_kernel), from a diagnostic build of the library
(make -C tinyllama.cpp_amd/csrc EXTRA=-DTL_PX_STAMP): shader-clock stamps per step of the 8 consumer and 4 producer waves of
workgroup (0, 0). Prints, per step (median over steps 8..40), cycles since the step's first wave left the barrier.
Usage: python tools/px_stamps.py [T] [rows] [cols]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402

pkg = load_package()
S = pkg.synth
from tinyllama_cpp_amd import engine as E  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14336
    cols = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    E.init(0)
    L = lib()
    L.tl_debug_px_stamps.argtypes = [C.c_void_p, C.c_int]
    L.tl_debug_px_stamps.restype = C.c_int
    rng = np.random.default_rng(0)
    blocks = S.random_blocks(S.Q4_K, rows * cols // 256, rng, w_std=0.02)
    w = E.QWeight(S.Q4_K, blocks, rows, cols)
    w.expand()
    x = E.DeviceBuffer.from_numpy(rng.standard_normal((T, cols)).astype(np.float32))
    y = E.DeviceBuffer(4 * T * rows)
    for _ in range(3):
        check(L.tl_matmul_q(w.h, x.ptr, y.ptr, T, None), "tl_matmul_q")
    check(L.tl_device_sync(), "sync")
    check(L.tl_debug_px_stamps(None, 1), "clear")
    check(L.tl_matmul_q(w.h, x.ptr, y.ptr, T, None), "tl_matmul_q")
    check(L.tl_device_sync(), "sync")
    st = np.zeros((12, 48, 8), np.uint64)
    check(L.tl_debug_px_stamps(st.ctypes.data, 0), "read")
    st = st.astype(np.int64)
    ks = range(8, 40)
    print(f"T={T} {rows}x{cols} Q4_K: cycles (shader clock), medians over steps 8..39 of workgroup (0,0)")
    step_len = np.median([st[:8, k + 1, 0].min() - st[:8, k, 0].min() for k in ks])
    print(f"step length (first consumer past barrier k -> past barrier k+1): {step_len:.0f}")
    names_c = ["past barrier", "half 0 MFMAs issued", "half 0 folded", "half 1 MFMAs issued", "half 1 folded", "at the barrier"]
    for wv in range(8):
        rel = [[st[wv, k, s] - st[:8, k, 0].min() for k in ks] for s in (0, 1, 2, 3, 4)] + [[st[wv, k + 1, 3 if False else 7] - st[:8, k, 0].min() for k in ks]]
        print(f"  consumer {wv}: " + " | ".join(f"{n} {np.median(r):5.0f}" for n, r in zip(names_c, rel)))
    names_p = ["past barrier", "DMAs of step k+2 issued", "step k+1 landed, at the barrier"]
    for wv in range(8, 12):
        rel = [[st[wv, k, s] - st[:8, k, 0].min() for k in ks] for s in (0, 1)] + [[st[wv, k + 1, 2] - st[:8, k, 0].min() for k in ks]]
        print(f"  producer {wv - 8}: " + " | ".join(f"{n} {np.median(r):6.0f}" for n, r in zip(names_p, rel)))


if __name__ == "__main__":
    main()
