#!/usr/bin/env python3
"""GEMV micro-benchmark over the C-ABI (tl_bench_matvec): per (type, rows, cols) the average launch time with
HIP events while cycling through enough equally-shaped matrices to defeat the 256 MiB Infinity Cache, and the
algorithmic GB/s (GGUF bytes of the matrix / time). Usage: python tools/bench_gemv.py [shape-set]"""
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

SETS = {
    "8b": [(S.Q4_K, 4096, 4096), (S.Q4_K, 1024, 4096), (S.Q4_K, 14336, 4096), (S.Q4_K, 4096, 14336),
           (S.Q6_K, 4096, 14336), (S.Q6_K, 128256, 4096), (S.Q8_0, 4096, 4096), (S.Q6_K, 1024, 4096)],
    "1b": [(S.Q4_K, 2048, 2048), (S.Q4_K, 5632, 2048), (S.Q4_K, 2048, 5632), (S.Q6_K, 32000, 2048),
           (S.Q8_0, 2048, 2048), (S.Q8_0, 5632, 2048)],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "8b"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    engine.init(0)
    rng = np.random.default_rng(1)
    for typ, rows, cols in SETS[which]:
        nbytes = S.tensor_bytes(typ, rows, cols)
        nw = max(2, min(64, int(640e6 // nbytes) + 1))
        blocks = S.random_blocks(typ, rows * cols // S.BLOCK_ELEMS[typ], rng, w_std=1.0 / np.sqrt(cols))
        ws = [engine.QWeight(typ, blocks, rows, cols) for _ in range(nw)]
        arr = (C.c_void_p * nw)(*[w.h for w in ws])
        x = engine.DeviceBuffer.from_numpy(rng.standard_normal(cols).astype(np.float32))
        y = engine.DeviceBuffer(4 * rows)
        us = C.c_float()
        check(lib().tl_bench_matvec(arr, nw, x.ptr, y.ptr, iters, 20, C.byref(us), None), "tl_bench_matvec")
        gbps = nbytes / (us.value * 1e-6) / 1e9
        line = (f"{S.TYPE_NAMES[typ]:5s} {rows:6d} x {cols:5d}  {nbytes / 1e6:8.2f} MB  nw={nw:2d}  {us.value:8.2f} us  "
                f"{gbps:7.1f} GB/s  {gbps / 80:5.1f}% of 8 TB/s")
        if os.environ.get("TL_BENCH_PREFETCH"):  # graph replay without / with an Infinity-Cache prefetch branch
            for pb in [0] + [int(v) for v in os.environ["TL_BENCH_PREFETCH"].split(",")]:
                check(lib().tl_bench_matvec_prefetch(arr, nw, x.ptr, y.ptr, 5, pb, C.byref(us)), "prefetch bench")
                line += f" | pf{pb}: {us.value:6.2f} us"
        print(line, flush=True)
        for w in ws:
            w.free()


if __name__ == "__main__":
    main()
