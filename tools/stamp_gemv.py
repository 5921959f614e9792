#!/usr/bin/env python3
"""In-kernel timeline of one GEMV launch (tl_debug_gemv_stamps): per phase, min / median / max over waves of the
shader-clock offset from the earliest wave's kernel entry. Usage: python tools/stamp_gemv.py TYPE ROWS COLS"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402

pkg = load_package()
S = pkg.synth
from tinyllama_cpp_amd import engine  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402

NAMES = ["entry", "loads issued", "x staged", "first step done", "wave end"]


def main():
    typ = {"q4_k": S.Q4_K, "q6_k": S.Q6_K, "q8_0": S.Q8_0}[sys.argv[1]]
    rows, cols = int(sys.argv[2]), int(sys.argv[3])
    engine.init(0)
    rng = np.random.default_rng(1)
    blocks = S.random_blocks(typ, rows * cols // S.BLOCK_ELEMS[typ], rng, w_std=1.0 / np.sqrt(cols))
    nw = max(2, int(600e6 // blocks.nbytes))
    ws = [engine.QWeight(typ, blocks, rows, cols) for _ in range(min(nw, 40))]
    x = engine.DeviceBuffer.from_numpy(rng.standard_normal(cols).astype(np.float32))
    y = engine.DeviceBuffer(4 * rows)
    st = engine.DeviceBuffer(8 * 8 * 256 * 16)
    g, wv = C.c_int(), C.c_int()
    for rep in range(3):
        for w in ws:  # flush caches with the other copies, keep the code warm
            check(lib().tl_matvec_q(w.h, x.ptr, y.ptr, None), "matvec")
        check(lib().tl_debug_gemv_stamps(ws[0].h, x.ptr, y.ptr, st.ptr, C.byref(g), C.byref(wv), None), "stamps")
        check(lib().tl_device_sync(), "sync")
        a = st.to_numpy(np.uint64, (256 * 16, 8))[: g.value * wv.value]
        live = a[a[:, 4] > 0]
        idle = a[(a[:, 4] == 0)]
        print(f"rep {rep}: grid {g.value} x {wv.value} waves, {len(live)} working waves; per-wave shader cycles since "
              f"the wave's own entry (min / median / max):")
        for k, nm in enumerate(NAMES):
            d = (live[:, k] - live[:, 0]).astype(np.int64)
            print(f"  {nm:16s} {d.min():7d} {int(np.median(d)):7d} {d.max():7d}")
        r0, r1 = live[:, 6].astype(np.int64), live[:, 5].astype(np.int64)
        base = r0.min()
        print(f"  realtime (ns since first wave entry): entries {10 * (r0 - base).min()} / {10 * int(np.median(r0 - base))} / "
              f"{10 * (r0 - base).max()}   ends {10 * (r1 - base).min()} / {10 * int(np.median(r1 - base))} / "
              f"{10 * (r1 - base).max()}")
        clk = (live[:, 4] - live[:, 0]).astype(np.float64) / np.maximum((r1 - r0) * 10.0, 1.0)
        print(f"  shader clock ~ {np.median(clk):.2f} GHz")


if __name__ == "__main__":
    main()
