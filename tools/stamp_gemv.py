#!/usr/bin/env python3
"""In-kernel timeline of one GEMV launch (tl_debug_gemv_stamps): per phase, min / median / max over waves, in ns since
the earliest wave's kernel entry (shader-clock stamps converted with the wave's own clock / realtime ratio).
Usage: python tools/stamp_gemv.py TYPE ROWS COLS [TYPE ROWS COLS ...] [--json out.jsonl]
The stamped build's run time is not the real kernel's (the stamps serialise what the real kernel overlaps): read the
SHARES, and the wall time from tl_bench_matvec printed beside them."""
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
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402

NAMES = ["entry", "arguments in SGPRs", "x loads issued", "first loads issued", "x staged", "first step done",
         "last loads issued", "wave end"]


def one(typ_name, rows, cols, out):
    typ = {"q4_k": S.Q4_K, "q6_k": S.Q6_K}[typ_name]
    rng = np.random.default_rng(1)
    blocks = S.random_blocks(typ, rows * cols // S.BLOCK_ELEMS[typ], rng, w_std=1.0 / np.sqrt(cols))
    nw = max(2, int(600e6 // blocks.nbytes))
    ws = [engine.QWeight(typ, blocks, rows, cols) for _ in range(min(nw, 40))]
    x = engine.DeviceBuffer.from_numpy(rng.standard_normal(cols).astype(np.float32))
    y = engine.DeviceBuffer(4 * rows)
    st = engine.DeviceBuffer(8 * 10 * 256 * 16)
    g, wv = C.c_int(), C.c_int()
    # un-stamped wall time per launch, back to back over copies that do not fit the Infinity Cache
    arr = (C.c_void_p * len(ws))(*[w.h for w in ws])
    us = C.c_float()
    check(lib().tl_bench_matvec(arr, len(ws), x.ptr, y.ptr, 200, 20, C.byref(us), None), "bench")
    recs = []
    for rep in range(3):
        for w in ws:  # flush caches with the other copies, keep the code warm
            check(lib().tl_matvec_q(w.h, x.ptr, y.ptr, None), "matvec")
        check(lib().tl_debug_gemv_stamps(ws[0].h, x.ptr, y.ptr, st.ptr, C.byref(g), C.byref(wv), None), "stamps")
        check(lib().tl_device_sync(), "sync")
        a = st.to_numpy(np.uint64, (256 * 16, 10))[: g.value * wv.value]
        slot = np.arange(len(a))[a[:, 7] > 0]  # slot = wave * grid + workgroup
        live = a[a[:, 7] > 0]
        r0, r1 = live[:, 9].astype(np.int64) * 10, live[:, 8].astype(np.int64) * 10  # ns
        base = r0.min()
        clk = (live[:, 7] - live[:, 0]).astype(np.float64) / np.maximum((r1 - r0), 1.0)  # cycles per ns
        clk_med = float(np.median(clk))
        rec = {"type": typ_name, "rows": rows, "cols": cols, "bytes": int(blocks.nbytes), "grid": g.value,
               "waves_per_wg": wv.value, "working_waves": int(len(live)), "unstamped_us_per_launch": round(us.value, 2),
               "shader_clock_ghz": round(clk_med, 3), "rep": rep, "phases_ns_since_first_entry": {}}
        print(f"{typ_name} {rows}x{cols} rep {rep}: grid {g.value} x {wv.value} waves, {len(live)} working; un-stamped "
              f"{us.value:.2f} us/launch; ns since the FIRST wave's entry (min / median / max over waves):")
        for k, nm in enumerate(NAMES):
            t = (r0 - base) + (live[:, k] - live[:, 0]).astype(np.float64) / clk_med
            rec["phases_ns_since_first_entry"][nm] = [int(t.min()), int(np.median(t)), int(t.max())]
            print(f"  {nm:20s} {int(t.min()):7d} {int(np.median(t)):7d} {int(t.max()):7d}")
        loop = (live[:, 6] - live[:, 5]).astype(np.float64) / clk_med
        span = float((r1 - base).max())
        rec["steady_loop_ns_median"] = int(np.median(loop))
        rec["kernel_span_ns"] = int(span)
        # bytes requested between 'first step done' and 'last loads issued' = all but the first two steps of each wave
        print(f"  steady loop (first step done -> last loads issued): median {int(np.median(loop))} ns; kernel span "
              f"{int(span)} ns; whole-launch rate {blocks.nbytes / span:.0f} GB/s (stamped build)")
        # where does the spread of the waves' end times come from: between the waves of a workgroup (one CU) or between
        # workgroups? (decides whether handing out row groups inside a workgroup would even it out)
        end = (r0 - base) + (live[:, 7] - live[:, 0]).astype(np.float64) / clk_med
        wg = slot % g.value
        if len(np.unique(wg)) > 1 and len(live) > len(np.unique(wg)):
            means = np.array([end[wg == b].mean() for b in np.unique(wg)])
            inner = np.array([end[wg == b].max() - end[wg == b].min() for b in np.unique(wg)])
            rec["wave_end_spread_ns"] = {"within_workgroup_median": int(np.median(inner)), "within_workgroup_max": int(inner.max()),
                                         "workgroup_means_min": int(means.min()), "workgroup_means_max": int(means.max()),
                                         "all_waves_mean": int(end.mean())}
            print(f"  wave end: spread inside a workgroup median {int(np.median(inner))} / max {int(inner.max())} ns; workgroup "
                  f"means {int(means.min())} .. {int(means.max())} ns; mean over all waves {int(end.mean())} ns")
        recs.append(rec)
    out.extend(recs[1:])  # the first repetition carries the stamped kernel's code-cache misses
    for w in ws:
        w.free() if hasattr(w, "free") else None


def main():
    args = sys.argv[1:]
    jpath = None
    if "--json" in args:
        i = args.index("--json")
        jpath = args[i + 1]
        del args[i:i + 2]
    engine.init(0)
    out = []
    for i in range(0, len(args), 3):
        one(args[i], int(args[i + 1]), int(args[i + 2]), out)
    if jpath:
        with open(jpath, "w") as f:
            for r in out:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
