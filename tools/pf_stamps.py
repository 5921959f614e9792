#!/usr/bin/env python3
"""Per-wave timeline of the prefill attention kernel (attn_prefill_mfma_kernel) from a diagnostic build of the library
(tl_prefill.hip compiled with -DTL_PF_STAMP): 100 MHz wall-clock stamps of the waves of the first 256 workgroups.
Prints medians / maxima over workgroups, in us since the launch's first stamp. Usage: python tools/pf_stamps.py [T]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402

load_package()
from tinyllama_cpp_amd import engine as E  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402

NAMES = ["entry", "q loaded", "tile0: K loaded", "tile0: QK done", "tile0: softmax done", "tile0: PV done", "tiles done",
         "past barrier 0", "factors", "slots written, past barrier", "tile stored"]


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nh, nkv, hd = 32, 8, 128
    E.init(0)
    L = lib()
    L.tl_debug_pf_stamps.argtypes = [C.c_void_p]
    L.tl_debug_pf_stamps.restype = C.c_int
    rng = np.random.default_rng(0)
    q = E.DeviceBuffer.from_numpy(rng.standard_normal((T, nh * hd)).astype(np.float32))
    k = E.DeviceBuffer.from_numpy(rng.standard_normal((T, nkv * hd)).astype(np.float32))
    v = E.DeviceBuffer.from_numpy(rng.standard_normal((T, nkv * hd)).astype(np.float32))
    o = E.DeviceBuffer(4 * T * nh * hd)
    for _ in range(3):
        check(L.tl_attention_batch_prefill(q.ptr, k.ptr, v.ptr, o.ptr, T, 0, T, nh, nkv, hd, float(hd ** -0.5), None))
    check(L.tl_device_sync())
    st = np.zeros((256, 8, 32), np.uint64)
    check(L.tl_debug_pf_stamps(st.ctypes.data))
    st = st.astype(np.int64)
    t0 = st[:, :, 0][st[:, :, 0] > 0].min()
    us = (st - t0) / 100.0
    print(f"T={T}: us since the first wave's entry; per stamp: median / max over (workgroup, wave) that reached it")
    for ph in range(2):
        print(f" -- {'late (heavy)' if ph == 0 else 'early (light)'} tile of the pair")
        for i, n in enumerate(NAMES):
            kk = i if i == 0 else i + 12 * ph
            if i == 0 and ph == 1:
                continue
            m = st[:, :, kk] > 0
            if m.any():
                print(f"    {n:32s} {np.median(us[:, :, kk][m]):7.2f} / {us[:, :, kk][m].max():7.2f}   ({m.sum()} waves)")
    for wg in (0, 100, 255):
        print(f" workgroup {wg}, wave 0: " + " ".join(f"{us[wg, 0, kk]:.2f}" if st[wg, 0, kk] > 0 else "-" for kk in range(23)))
        print(f" workgroup {wg}, wave 7: " + " ".join(f"{us[wg, 7, kk]:.2f}" if st[wg, 7, kk] > 0 else "-" for kk in range(23)))


if __name__ == "__main__":
    main()
