#!/usr/bin/env python3
"""Timeline of the decode step's chain launches (tl_chain_set_stamps): for every task of one chain launch, when its
workgroups entered, issued their first weight loads, saw the predecessor task complete, had x staged, issued their last
loads, had their stores acknowledged and added their arrival: min / median / max over the task's workgroups, in us since
the launch's first workgroup entered. Run with TL_NO_GRAPH=1.
Usage: TL_NO_GRAPH=1 python tools/chain_stamps.py [--workload llama-3-8b] [--launch 5] [--json out.json]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TL_NO_GRAPH", "1")
from _pkg import load_package  # noqa: E402

pkg = load_package()
S = pkg.synth
from tinyllama_cpp_amd import engine  # noqa: E402
from tinyllama_cpp_amd.capi import check, lib  # noqa: E402

NAMES = ["entry", "first loads issued", "predecessor seen", "x staged", "last loads issued", "stores acknowledged",
         "arrival added"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="llama-3-8b")
    ap.add_argument("--mix", default="q4_k_m")
    ap.add_argument("--launch", type=int, default=5, help="which chain launch of the step (= layer) to print")
    ap.add_argument("--tasks", type=int, default=4)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    engine.init(0)
    cfg = dict(S.SHAPES[args.workload])
    cfg["max_pos"] = min(cfg["max_pos"], 512)
    model = engine.Model(cfg)
    for name, typ, rows, cols in S.tensor_plan(cfg, args.mix):
        model.set_tensor(name, typ, S.make_tensor(name, typ, rows, cols, seed=42), rows, cols)
    model.finalize()
    nl = cfg["n_layers"]
    words = nl * args.tasks * 256 * 8
    buf = engine.DeviceBuffer(8 * words)
    ids = model.generate_greedy(1, 0, 24)  # warm
    pos = 24
    out = []
    for rep in range(3):
        check(lib().tl_memset(buf.ptr, 0, 8 * words, None), "memset")
        check(lib().tl_chain_set_stamps(buf.ptr, words), "stamps on")
        ids = model.generate_greedy(int(ids[-1]), pos, 1)
        pos += 1
        check(lib().tl_chain_set_stamps(None, 0), "stamps off")
        a = buf.to_numpy(np.uint64, (nl, args.tasks, 256, 8)).astype(np.int64)
        la = a[args.launch]
        live = la[:, :, 0] > 0
        base = la[:, :, 0][live].min()
        rec = {"rep": rep, "launch": args.launch, "tasks": []}
        print(f"rep {rep}, chain launch {args.launch}: us since the launch's first workgroup entry (min / median / max over workgroups)")
        for t in range(args.tasks):
            w = la[t][live[t]]
            if len(w) == 0:
                continue
            row = {"task": t, "workgroups": int(len(w))}
            print(f"  task {t} ({len(w)} workgroups)")
            for k, nm in enumerate(NAMES):
                v = (w[:, k] - base) / 100.0
                row[nm] = [round(float(v.min()), 2), round(float(np.median(v)), 2), round(float(v.max()), 2)]
                print(f"    {nm:22s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
            rec["tasks"].append(row)
        # whole launch and whole step
        allv = a[:, :, :, :][a[:, :, :, 0] > 0]
        rec["launch_us"] = round(float((la[live][:, 6].max() - base) / 100.0), 2)
        rec["step_first_to_last_stamp_us"] = round(float((allv[:, 6].max() - allv[:, 0].min()) / 100.0), 2)
        print(f"  launch: {rec['launch_us']} us from first entry to last arrival; all chain launches of the step span "
              f"{rec['step_first_to_last_stamp_us']} us")
        out.append(rec)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
