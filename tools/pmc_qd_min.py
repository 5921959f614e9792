#!/usr/bin/env python3
"""The smallest program that launches the direct small-token GEMM (gemm_qd_kernel) of the 32-sequence decode step:
tl_matmul_q with 32 tokens on the two Llama-3-8B FFN shapes, three launches each and nothing else (no model, no decode
step). For rocprofv3 --pmc passes: with a whole model in the process the profiler's counter collection faulted inside a
launch of the eager decode step (profiles/r02_f_pmc_qd_abort.txt), whatever the counter set."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd.capi import lib, check
    E.init(0)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    expand = len(sys.argv) > 2 and sys.argv[2] == "px"  # the expanded-plane kernel (tl_qweight_expand) instead
    rng = np.random.default_rng(0)
    for rows, cols in ((14336, 4096), (4096, 14336)):
        blocks = S.random_blocks(S.Q4_K, rows * cols // 256, rng, w_std=0.02)
        w = E.QWeight(S.Q4_K, blocks, rows, cols)
        if expand:
            w.expand()
        x = E.DeviceBuffer.from_numpy(rng.standard_normal((T, cols)).astype(np.float32))
        y = E.DeviceBuffer(4 * T * rows)
        for _ in range(3):
            check(lib().tl_matmul_q(w.h, x.ptr, y.ptr, T, None), "tl_matmul_q")
        check(lib().tl_device_sync(), "sync")
        print(rows, cols, "ok", float(np.abs(y.to_numpy(np.float32, (T, rows))).max()), flush=True)


if __name__ == "__main__":
    main()
