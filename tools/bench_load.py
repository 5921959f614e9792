#!/usr/bin/env python3
"""Load path: a synthetic GGUF file of the Llama-3-8B shapes (cut to --layers layers) is written to a scratch
directory, then opened twice through the C++ host layer (mmap -> tl_model_set_tensor -> plane layout on the device):
seconds and GB/s of quantized bytes from the mapping to HBM. The second open reads from a warm page cache."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--mix", default="q4_k_m")
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import hostapi as H
    cfg = dict(S.SHAPES["llama-3-8b"], n_layers=a.layers, max_pos=256)
    cfg, tensors = S.build_model(cfg, a.mix, seed=3)
    with tempfile.TemporaryDirectory() as d:
        path = S.write_gguf(os.path.join(d, "m.gguf"), cfg, tensors)
        del tensors
        size = os.path.getsize(path)
        for run in ("first open", "second open (warm page cache)"):
            t0 = time.perf_counter()
            s = H.Session(path)
            wall = time.perf_counter() - t0
            n, qbytes, secs = s.load_stats()
            print(json.dumps({"run": run, "file_bytes": size, "tensors": n, "quantized_bytes": qbytes,
                              "open_seconds": round(secs, 3), "wall_seconds": round(wall, 3),
                              "gb_per_s": round(qbytes / secs / 1e9, 2)}), flush=True)
            s.close()


if __name__ == "__main__":
    main()
