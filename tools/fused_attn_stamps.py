#!/usr/bin/env python3
"""Timeline of an attn_output launch that carries its layer's attention (TL_FUSED_ATTN=1; tl_chain_set_stamps): when the
workgroups entered, issued their first weight loads, saw all attention records published, had x staged, issued their last
loads and finished; for the workgroups with an attention task also when their records were acknowledged and their arrival
added. min / median / max over workgroups, in us since the launch's first workgroup entered.
Usage: python tools/fused_attn_stamps.py [--workload llama-3-8b] [--layer 5] [--pos 24] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TL_NO_GRAPH"] = "1"
os.environ["TL_FUSED_ATTN"] = "1"
from _pkg import load_package  # noqa: E402

pkg = load_package()
S = pkg.synth
from tinyllama_cpp_amd import engine  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402

NAMES = ["entry", "first loads issued", "all records published", "x staged", "last loads issued",
         "(attention) records acknowledged", "(attention) arrival added", "thread 0 done"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="llama-3-8b")
    ap.add_argument("--mix", default="q4_k_m")
    ap.add_argument("--layer", type=int, default=5)
    ap.add_argument("--pos", type=int, default=24)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    engine.init(0)
    cfg = dict(S.SHAPES[args.workload])
    cfg["max_pos"] = min(cfg["max_pos"], 2048)
    model = engine.Model(cfg)
    for name, typ, rows, cols in S.tensor_plan(cfg, args.mix):
        model.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
    model.finalize()
    nl = cfg["n_layers"]
    words = nl * 256 * 8
    buf = engine.DeviceBuffer(8 * words)
    pos = args.pos
    ids = model.generate_greedy(1, pos - 8, 8)  # warm
    out = []
    for rep in range(3):
        check(lib().tl_memset(buf.ptr, 0, 8 * words, None), "memset")
        check(lib().tl_chain_set_stamps(buf.ptr, words), "stamps on")
        ids = model.generate_greedy(int(ids[-1]), pos, 1)
        pos += 1
        check(lib().tl_chain_set_stamps(None, 0), "stamps off")
        a = buf.to_numpy(np.uint64, (nl, 256, 8)).astype(np.int64)
        la = a[args.layer]
        live = la[:, 0] > 0
        base = la[live, 0].min()
        rec = {"rep": rep, "layer": args.layer, "pos": pos - 1, "rows": {}}
        print(f"rep {rep}, layer {args.layer}, position {pos - 1}: us since the launch's first workgroup entry (min / median / max)")
        for k, nm in enumerate(NAMES):
            sel = live & (la[:, k] > 0)
            if not sel.any():
                continue
            v = (la[sel, k] - base) / 100.0
            rec["rows"][nm] = [int(sel.sum()), round(float(v.min()), 2), round(float(np.median(v)), 2), round(float(v.max()), 2)]
            print(f"    {nm:34s} n={int(sel.sum()):3d} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
        ends = a[:, :, 7].max(axis=1)
        starts = np.where(a[:, :, 0] > 0, a[:, :, 0], np.iinfo(np.int64).max).min(axis=1)
        rec["launch_us_all_layers_median"] = round(float(np.median(ends - starts) / 100.0), 2)
        rec["layer_period_us_median"] = round(float(np.median(np.diff(starts)) / 100.0), 2)
        print(f"  all layers: launch entry -> last thread-0 exit, median {rec['launch_us_all_layers_median']} us; "
              f"entry to next layer's entry, median {rec['layer_period_us_median']} us (eager launches)")
        out.append(rec)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
