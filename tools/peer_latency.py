"""Latency of the decode step's all-reduce of f32[hidden] through the peer-to-peer exchange (csrc/tl_peer.hip), measured
with tp PROCESSES that share GPU 0 (the only multi-rank set-up a one-GPU box offers: the words cross between processes
and between the chip's L2s, not over xGMI links -- the protocol's floor, to which a link hop adds its latency).
usage: python tools/peer_latency.py [--tp 2 4 8] [--hidden 4096] [--iters 2000]    (parent)
Prints one JSON line per tp with the average time of back-to-back exchanges as every rank saw it."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, tp_size, hidden, iters, workdir, device=0, rv_timeout=120.0):
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd import tp
    E.init(device)
    heads = hidden // 128
    if hidden == 4096:  # ONE real layer of Llama-3-8B and its lm_head: the exchanges carry what the bench's will carry
        cfg = dict(S.SHAPES["llama-3-8b"], n_layers=1, max_pos=64)
    else:
        cfg = dict(S.SHAPES["llama-3-8b"], hidden=hidden, n_heads=heads, n_kv_heads=8, inter=512 * tp_size, n_layers=1,
                   vocab=1024 * tp_size, max_pos=64)
    cfg, tensors = S.build_model(cfg, "q4_k_m", seed=1)
    m = E.Model(cfg, tp_rank=rank, tp_size=tp_size)
    rv = tp.FileRendezvous(workdir, rank, tp_size, timeout=rv_timeout)
    if not tp.peer_link(m, rv.gather, rv.agree):
        print(f"rank {rank}: not linked: {E.lib().tl_last_error().decode(errors='replace')}")
        return 3
    for name, (typ, arr, rows, cols) in tensors.items():
        m.set_tensor(name, typ, arr, rows, cols)
    rv.agree(True)  # the warm-up step inside finalize is a collective of the group: enter it together
    m.finalize()
    rv.agree(True)
    us, nbytes = m.bench_kernel("tp_all_reduce", iters=iters, warmup=200)
    ids = m.generate_greedy(1, 0, 32)
    rv.agree(True)
    import time
    # A step of this one-layer model is mostly its three exchanges: bench.py's pre-flight compares the two forms of the
    # exchange (folded into the GEMVs / launches of their own) by this time. The short run in front is the barrier: its
    # last exchange leaves the ranks within microseconds of each other (the file rendezvous above does not).
    m.generate_greedy(1, 0, 8)
    t0 = time.perf_counter()
    m.generate_greedy(1, 0, 48)
    step_us = (time.perf_counter() - t0) / 48 * 1e6
    rv.agree(True)
    with open(os.path.join(workdir, f"lat_{rank}.json"), "w") as f:
        json.dump({"rank": rank, "us": us, "bytes": nbytes, "ids": [int(i) for i in ids], "step_us": step_us,
                   "fused": bool(m.tp_peer_fused())}, f)
    rv.agree(True)
    m.free()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tp", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--worker", type=int, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dir", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)  # bench.py's pre-flight: this rank's own GPU
    ap.add_argument("--rv-timeout", type=float, default=120.0, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.worker:
        return worker(a.worker[0], a.worker[1], a.hidden, a.iters, a.dir, a.device, a.rv_timeout)
    rc = 0
    for tp_size in a.tp:
        with tempfile.TemporaryDirectory() as d:
            procs = [subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "--worker", str(r), str(tp_size),
                                       "--hidden", str(a.hidden), "--iters", str(a.iters), "--dir", d]) for r in range(tp_size)]
            try:
                codes = [p.wait(timeout=300) for p in procs]
            finally:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            if any(codes):
                print(json.dumps({"tp": tp_size, "error": f"worker exit codes {codes}"}))
                rc = 1
                continue
            res = [json.load(open(os.path.join(d, f"lat_{r}.json"))) for r in range(tp_size)]
            same = all(r["ids"] == res[0]["ids"] for r in res)
            print(json.dumps({"tp": tp_size, "ranks_share_one_gpu": True, "hidden": a.hidden, "bytes": res[0]["bytes"],
                              "iters": a.iters, "us_per_all_reduce": [round(r["us"], 2) for r in res],
                              "exchanges_folded_into_gemvs": res[0]["fused"],
                              "us_per_step_of_the_one_layer_model": [round(r["step_us"], 1) for r in res],
                              "ranks_agree_on_tokens": same}))
    return rc


if __name__ == "__main__":
    sys.exit(main())
