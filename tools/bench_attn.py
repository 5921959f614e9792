#!/usr/bin/env python3
"""Decode-attention microbenchmark: tl_attention (split kernel + combine) over a device-resident f32 KV cache at several
context lengths, timed back to back on the default stream between device syncs. bytes = 2 * len * kv_dim * 4."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from _pkg import load_package
    load_package()
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd.capi import lib, check
    E.init(0)
    nh, nkv, hd = 32, 8, 128
    L = 16  # distinct caches cycled through so that K/V stream from HBM, not from the Infinity Cache
    for n in (128, 512, 1024, 2048, 4096, 8192):
        kv_dim = nkv * hd
        rng = np.random.default_rng(n)
        caches = [(E.DeviceBuffer.from_numpy(rng.standard_normal((n, kv_dim)).astype(np.float32)),
                   E.DeviceBuffer.from_numpy(rng.standard_normal((n, kv_dim)).astype(np.float32))) for _ in range(L)]
        q = E.DeviceBuffer.from_numpy(rng.standard_normal(nh * hd).astype(np.float32))
        o = E.DeviceBuffer(4 * nh * hd)
        scale = float(1.0 / np.sqrt(hd))
        def run(iters):
            for i in range(iters):
                k, v = caches[i % L]
                check(lib().tl_attention(q.ptr, k.ptr, v.ptr, o.ptr, nh, n, hd, scale, n, nkv, None))
            check(lib().tl_device_sync())
        run(L)
        t0 = time.perf_counter()
        iters = 20 * L
        run(iters)
        us = (time.perf_counter() - t0) / iters * 1e6
        nbytes = 2 * n * kv_dim * 4
        print(f"len {n:5d}: {us:7.2f} us per call (split + combine launches), {nbytes / us / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
