#!/usr/bin/env python3
"""Ring-loader engine (tl_stream_matvec) against the launch-per-projection GEMV (tl_matvec_q): outputs compared, both
timed over matrix copies that do not fit the Infinity Cache. Usage: python tools/bench_stream.py [--check-only]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402

pkg = load_package()
S = pkg.synth
from tinyllama_cpp_amd import engine  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib, tools_lib  # noqa: E402

TYPES = {"q4_k": S.Q4_K, "q6_k": S.Q6_K, "q8_0": S.Q8_0}


def run(typ_name, rows, cols, timed, out):
    typ = TYPES[typ_name]
    rng = np.random.default_rng(rows + cols)
    nb = rows * cols // S.BLOCK_ELEMS[typ]
    blocks = S.random_blocks(typ, nb, rng, w_std=1.0 / np.sqrt(cols))
    nw = max(2, min(40, int(600e6 // blocks.nbytes))) if timed else 2
    ws = [engine.QWeight(typ, blocks, rows, cols) for _ in range(nw)]
    x = engine.DeviceBuffer.from_numpy(rng.standard_normal(cols).astype(np.float32))
    y = engine.DeviceBuffer(4 * rows)
    ys = engine.DeviceBuffer(4 * rows * nw)
    check(lib().tl_memset(ys.ptr, 0xff, ys.nbytes, None))
    arr = (C.c_void_p * nw)(*[w.h for w in ws])
    us_s, us_l = C.c_float(), C.c_float()
    rc = tools_lib().tl_stream_matvec(arr, nw, x.ptr, ys.ptr, 20 if timed else 1, C.byref(us_s), None)
    if rc == 2 and b"stream:" in lib().tl_last_error():  # a shape the engine's item geometry does not cover
        print(json.dumps({"type": typ_name, "rows": rows, "cols": cols, "unsupported": lib().tl_last_error().decode()}), flush=True)
        return True
    check(rc, "stream")
    check(lib().tl_matvec_q(ws[0].h, x.ptr, y.ptr, None), "matvec")
    check(lib().tl_device_sync(), "sync")
    got = ys.to_numpy(np.float32, (nw, rows))
    want = y.to_numpy(np.float32, (rows,))
    err = float(np.abs(got - want[None, :]).max())
    scale = float(np.abs(want).max())
    ok = bool(np.all(np.abs(got - want[None, :]) <= 2e-5 * max(1.0, scale)))
    rec = {"type": typ_name, "rows": rows, "cols": cols, "max_abs_diff_vs_launch_gemv": err, "max_abs_y": scale, "ok": ok}
    if timed:
        check(lib().tl_bench_matvec(arr, nw, x.ptr, y.ptr, 200, 20, C.byref(us_l), None), "bench")
        rec.update({"stream_us": round(us_s.value, 2), "launch_us": round(us_l.value, 2),
                    "stream_gbps": round(blocks.nbytes / us_s.value / 1e3, 1), "launch_gbps": round(blocks.nbytes / us_l.value / 1e3, 1)})
    print(json.dumps(rec), flush=True)
    out.append(rec)
    return ok


def main():
    engine.init(0)
    out = []
    ok = True
    for t in (() if "--timing-only" in sys.argv else ("q4_k", "q6_k", "q8_0")):
        for rows, cols in ((8, 256), (37, 2048), (50, 3072), (19, 5632), (33, 8192), (64, 4096), (9, 14336), (1030, 4096)):
            ok &= run(t, rows, cols, False, out)
    if "--timing-only" in sys.argv:
        out.clear()
    if "--check-only" not in sys.argv:
        for t, rows, cols in (("q4_k", 4096, 4096), ("q4_k", 6144, 4096), ("q4_k", 28672, 4096), ("q4_k", 4096, 14336),
                              ("q6_k", 4096, 14336), ("q6_k", 128256, 4096)):
            ok &= run(t, rows, cols, True, out)
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
