#!/usr/bin/env python3
"""bench.py — greedy decode throughput of the quantized forward pass on MI355X (BASELINE.json metric).

A "step" is one decoded token: one pass of the hot path (embedding lookup, L transformer layers over the
device-resident KV cache, lm_head, on-device greedy sampling) for one sequence. The default workload is the one
the north star's target is quoted on and that fits one GPU: Llama-3-8B-shaped synthetic random-init Q4_K_M
(SURVEY §8d tensor mix; 4.62 GB of quantized weights per token), weights resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 128 --warmup 16 [--workload llama-3-8b|tinyllama-1.1b|llama-3.2-3b|tiny]

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME sequence is decoded by a tensor-parallel
group: every rank holds 1/N of every projection (tp.py), two f32[hidden] RCCL all-reduces per layer and one logits
all-gather per token run inside the captured step ("scaling": "strong": total work is fixed, per-GPU bytes = B_tok / N).

Prints ONE JSON line (rank 0). Timing: barrier + device sync on both sides, max over ranks. `roofline` is for the
dominant kernel (the fused RMSNorm + gate/up GEMV + SwiGLU launch): algorithmic bytes per launch / its average
launch duration, measured with HIP events on the engine's stream over back-to-back launches that cycle through the
layers (tl_model_bench_kernel). `cpu_baseline` times the CPU oracle (the reference's
OpenMP CPU path restated, oracle/) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIX = {"llama-3-8b": "q4_k_m", "tinyllama-1.1b": "q4_k_m", "llama-3.2-3b": "q6_k", "tiny": "q4_k_m",
       "tiny-gqa": "q4_k_m"}


METRIC = "decode tokens/sec + achieved HBM GB/s (% of peak) at Q4_K_M, 1/2/4/8 GPU"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="llama-3-8b")
    ap.add_argument("--mix", default=None)
    ap.add_argument("--max-pos", type=int, default=2048, help="KV-cache length allocated for the run")
    ap.add_argument("--no-prefill-leg", action="store_true")
    ap.add_argument("--no-batch-leg", action="store_true")
    ap.add_argument("--batch-seqs", type=int, default=128, help="sequences of the multi-sequence decode leg")
    ap.add_argument("--prompt-len", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-layers", type=int, default=8, help="layers in the CPU-baseline sample")
    return ap.parse_args()


# TL_BENCH_SHARE_GPU=1: all ranks of a `--gpus N` launch use GPU 0 (no RCCL communicator: it refuses two ranks on one
# device; the shards are linked by the peer exchange only). A REHEARSAL of the control flow, not a measurement: the line
# carries "rehearsal" and its numbers mean nothing.
SHARE_GPU = os.environ.get("TL_BENCH_SHARE_GPU", "0") == "1"
# where the tensors of the host-side collectives (handles, flags, times) live: on the GPU for RCCL; in host memory for the
# rehearsal's gloo group (gloo collectives on CUDA tensors stalled in all_gather once in about ten rehearsal runs, both
# ranks inside the call: profiles/r02_x_rehearsal_stall_traceback.txt)
HOST_DEV = "cpu" if SHARE_GPU else "cuda"
if SHARE_GPU and int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # exchanges folded into GEMV launches wait for the same launch of the other ranks: on ONE device all ranks' launches
    # have to fit on the chip together (csrc/tl_gemv.hip TL_GEMV_CUS); with a device per rank nothing is capped
    os.environ.setdefault("TL_GEMV_CUS", str(max(1, 128 // int(os.environ["WORLD_SIZE"]))))


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        if SHARE_GPU:  # rehearsal of the multi-rank path on a one-GPU box: every rank on device 0, gloo as the host channel
            local = 0
            torch.cuda.set_device(0)
            dist_mod.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local))
        dist = dist_mod
    return rank, world, local, dist


def main():
    args = parse()
    # Watchdog (seconds; TL_BENCH_WATCHDOG, 900 by default for multi-rank runs): dump every thread's Python stack to stderr
    # and exit instead of sitting in a collective for ever. Multi-rank runs also shorten the peer exchange's poll deadline
    # (5 s by default in the library): the ranks enter every collective call together here (barriers in front of finalize),
    # and a rank that is really gone should cost a second per exchange, not five, before the error return.
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    wd = int(os.environ.get("TL_BENCH_WATCHDOG", "900" if multi else "0"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    if multi:
        os.environ.setdefault("TL_TP_PEER_TIMEOUT_MS", "1000")
    rank, world, local, dist = dist_setup(args.gpus)
    from _pkg import load_package
    pkg = load_package()
    S = pkg.synth
    from tinyllama_cpp_amd import engine
    engine.init(local)

    mix = args.mix or MIX.get(args.workload, "q4_k_m")
    cfg = dict(S.SHAPES[args.workload])
    cfg["max_pos"] = min(cfg["max_pos"], args.max_pos)
    plan = S.tensor_plan(cfg, mix)

    # ---- build + upload the synthetic model tensor by tensor (host copy dropped right away) ---------------
    t_load = time.time()
    tp_err = None
    try:
        pkg_tp = __import__("tinyllama_cpp_amd.tp", fromlist=["check"])
        pkg_tp.check(cfg, world)
    except ValueError as e:  # this shape does not split `world` ways: say so and stop, on every rank
        tp_err = str(e)
    if tp_err:
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "tokens/s", "n_gpus": world,
                              "error": f"workload {args.workload} cannot be tensor-parallel over {world} GPUs: {tp_err}"}))
        if dist is not None:
            dist.destroy_process_group()
        return 1
    keep = {}  # host copies kept for the CPU-baseline sample: first layers + globals
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    preflight = {}  # TL_TP_PEER_FUSED value -> us per step of the pre-flight model (max over ranks), or "failed"

    made = []  # every model build() created, so that a build that raised half-way can still be freed

    def build(tp_rank, tp_size):
        model = engine.Model(cfg, tp_rank=tp_rank, tp_size=tp_size)
        made.append(model)
        if tp_size > 1:  # rank 0 makes the communicator id, torch.distributed carries it to the other ranks
            import torch
            uid = torch.zeros(128, dtype=torch.uint8, device=HOST_DEV)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(engine.tp_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            if not SHARE_GPU:
                model.tp_init(uid.cpu().numpy().tobytes())
            # the decode step's 16 KB exchanges: one-shot peer-to-peer kernels over hipIpc-mapped areas (csrc/tl_peer.hip)
            # when every rank passes their self-check, else the RCCL calls (TL_TP_PEER=0 forces those)
            if os.environ.get("TL_TP_PEER", "1") != "0":
                def gather(b):
                    mine = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(HOST_DEV)
                    parts = [torch.empty_like(mine) for _ in range(tp_size)]
                    dist.all_gather(parts, mine)
                    return [p_.cpu().numpy().tobytes() for p_ in parts]

                def agree(flag):
                    t = torch.tensor([1 if flag else 0], device=HOST_DEV)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return bool(int(t.item()))

                # Pre-flight: link + self-check + a short run of the exchange in THROW-AWAY child processes first, one per
                # rank on this rank's GPU (tools/peer_latency.py worker). The exchange has only ever run between
                # processes on one GPU; should the mapped areas fault over real links, that costs a child, not the bench:
                # every rank then stays on RCCL.
                import subprocess
                import tempfile
                # The children run the step in both forms of the exchange (folded into the GEMV launches / launches of
                # their own; TL_TP_PEER_FUSED pins one) and time 48 steps of their one-layer model, which is mostly its
                # three exchanges: the ranks agree on the faster form that every child survived, as measured over the
                # links of THIS node (single-lane pushes from a GEMV epilogue and coalesced pushes from an exchange
                # kernel need not cost the same over xGMI; a one-GPU box cannot tell).
                pinned = os.environ.get("TL_TP_PEER_FUSED")
                linked, best = False, (None, float("inf"))
                for mode in ([pinned] if pinned is not None else ["1", "0"]):
                    box = [tempfile.mkdtemp(prefix="tl_peer_") if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    cmd = [sys.executable, "-u", os.path.join(ROOT, "tools", "peer_latency.py"), "--worker", str(tp_rank),
                           str(tp_size), "--iters", "50", "--dir", box[0], "--device", str(local), "--rv-timeout", "30"]
                    step_us = float("inf")
                    try:
                        pre_ok = subprocess.run(cmd, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                                env=dict(os.environ, TL_TP_PEER_FUSED=mode)).returncode == 0
                        if pre_ok:
                            step_us = float(json.load(open(os.path.join(box[0], f"lat_{tp_rank}.json")))["step_us"])
                    except Exception:
                        pre_ok = False
                    if agree(pre_ok):
                        t = torch.tensor([step_us], device=HOST_DEV, dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # (the same number on every rank: the same choice)
                        preflight[mode] = round(float(t.item()), 1)
                        if float(t.item()) < best[1]:
                            best = (mode, float(t.item()))
                    else:
                        preflight[mode] = "failed"
                if best[0] is not None:
                    os.environ["TL_TP_PEER_FUSED"] = best[0]  # (read by tl_model_tp_peer_enable)
                    linked = pkg_tp.peer_link(model, gather, agree)
                if SHARE_GPU and not linked:
                    raise RuntimeError("ranks sharing one GPU can only be linked by the peer exchange, which was refused: "
                                       + engine.lib().tl_last_error().decode(errors="replace"))
        elif os.environ.get("TL_TP_SELFTEST"):  # one-rank communicator: the RCCL calls of the step on a single GPU
            model.tp_init(engine.tp_unique_id())
        for name, typ, rows, cols in plan:
            arr = S.make_tensor(name, typ, rows, cols, seed=42)
            model.set_tensor(name, typ, arr, rows, cols)
            layer = int(name.split(".")[1]) if name.startswith("blk.") else -1
            if want_cpu and (layer < 0 or layer < args.cpu_layers):
                keep[name] = (typ, arr, rows, cols)
            del arr
        if tp_size > 1 and dist is not None:
            dist.barrier()  # finalize runs one step, exchanges included: the ranks must enter it together (uploads skew)
        model.finalize()
        return model

    tp_fallback = None
    if world == 1:
        model = build(0, 1)
    else:
        # tensor parallel over the ranks. If the group cannot be set up (an exception on any rank, agreed on through
        # torch.distributed) the ranks try once more on the library collectives alone (TL_TP_PEER=0: a peer exchange that
        # passed its pre-flight and then failed in the full model's warm-up step leaves its group out of step, a fresh
        # group on RCCL is not affected); if that fails too, every rank falls back to a whole-model replica and the line
        # says so: weak scaling, no exchange, labelled as a fallback. (A hang inside a collective cannot be caught this way.)
        import torch
        model, tp_errors = None, []
        attempts = ["as configured"] + (["rccl only"] if not SHARE_GPU and os.environ.get("TL_TP_PEER", "1") != "0" else [])
        for attempt in attempts:
            if attempt == "rccl only":
                os.environ["TL_TP_PEER"] = "0"
            err = ""
            try:
                model = build(rank, world)
            except Exception as e:
                err = str(e)[:200]
            ok = torch.tensor([0 if err else 1], device=HOST_DEV)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                break
            errs = [None] * world
            dist.all_gather_object(errs, err)
            tp_errors.append(f"{attempt}: " + next((e for e in errs if e), "unknown"))
            for m_ in made:  # (also the one a build that raised never returned)
                m_.free()
            made.clear()
            model = None
        if model is None:
            tp_fallback = "; ".join(tp_errors)
            model = build(0, 1)
    t_load = time.time() - t_load

    # ---- prompt (token by token: P < 32), then warm-up and the timed region --------------------------------
    prompt = np.random.default_rng(0).integers(0, cfg["vocab"], args.prompt_len)
    nxt = 0
    tp_world = 1 if (world == 1 or tp_fallback) else world  # ranks that share ONE sequence
    if len(prompt) >= 32 and tp_world == 1:  # the generate loop's rule (api.cpp:601): batched prefill from 32 tokens on
        nxt = int(np.argmax(model.prefill(prompt.astype(np.int32), 0)))
    else:
        for pos, tok in enumerate(prompt):
            _, nxt = model.decode(int(tok), pos, want_logits=False)
    pos = len(prompt)
    if args.warmup > 0:
        ids = model.generate_greedy(nxt, pos, args.warmup)
        nxt, pos = int(ids[-1]), pos + args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        engine.check(engine.lib().tl_device_sync(), "sync")

    barrier()
    t0 = time.perf_counter()
    ids = model.generate_greedy(nxt, pos, args.steps)  # K graph replays back to back + one sync
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], device=HOST_DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert len(ids) == args.steps and all(0 <= int(i) < cfg["vocab"] for i in ids)
    mid_pos = pos + args.steps // 2
    w_bytes, kv_bytes = model.bytes_per_token(mid_pos)
    tok_s = args.steps / dt * (world if tp_fallback else 1)  # replicas (fallback): every rank decodes its own sequence

    # ---- the dominant kernel (fused RMSNorm + ffn_gate/ffn_up GEMV + SwiGLU): average launch duration with HIP events
    #      on the engine's stream, back-to-back launches cycling through the layers (every launch streams from HBM);
    #      algorithmic bytes = GGUF bytes of the two matrices one launch reads --------------------------
    gu_us, gu_bytes = model.bench_kernel("gate_up_gemv", iters=8 * cfg["n_layers"], warmup=8 * cfg["n_layers"])  # (the clocks settle during the warm-up)
    achieved = gu_bytes / (gu_us * 1e-6) / 1e9 if gu_us > 0 else 0.0
    kern = {}
    for k in ("qkv_gemv", "attn_out_gemv", "gate_up_gemv", "down_gemv", "lm_head_gemv"):
        us, nb = model.bench_kernel(k, iters=64 if k == "lm_head_gemv" else 4 * cfg["n_layers"], warmup=8)
        kern[k] = {"us": round(us, 2), "gbps": round(nb / (us * 1e-6) / 1e9, 1)}
    if tp_world > 1:  # the step's all-reduce of f32[hidden], back to back on every rank (2 per layer in a step)
        us, nb = model.bench_kernel("tp_all_reduce", iters=500, warmup=50)
        kern["tp_all_reduce"] = {"us": round(us, 2), "bytes": int(nb), "as": "a launch of its own, back to back",
                                 "via": ("peer-folded" if model.tp_peer_fused() else "peer") if model.tp_peer_enabled() else "rccl"}
    # whole eager steps with an event after every launch (includes ~5 us of event overhead per launch; the GEMV rows
    # above are the clean numbers): kept for the non-GEMV kernels
    prof = model.profile_decode(int(ids[-1]), pos + args.steps, min(8, max(1, cfg["max_pos"] - pos - args.steps)))
    # roofline.traffic is a KEPT figure: the --pmc FETCH_SIZE pass of this kernel on this workload (tools/gpu/r03_prof_round.sh
    # -> profiles/latest_gate_up_traffic.json), not measured in this process. The file carries a hash of the GEMV sources
    # it was measured on; a figure of other sources is not reported (traffic = null, the reason in traffic_source).
    traffic, traffic_source, rocprof_us = None, "no kept PMC figure for this workload", None
    tpath = os.path.join(ROOT, "profiles", "latest_gate_up_traffic.json")
    if args.workload == "llama-3-8b" and mix == "q4_k_m" and world == 1 and os.path.exists(tpath):
        kept = json.load(open(tpath))
        if kept.get("gemv_sources_sha256") == gemv_sources_hash():
            traffic = kept.get("hbm_bytes_per_launch")
            rocprof_us = kept.get("rocprof_avg_launch_us")
            traffic_source = ("kept PMC figure, not measured in this process: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) on these "
                              "GEMV sources, profiles/latest_gate_up_traffic.json")
        else:
            traffic_source = "stale: profiles/latest_gate_up_traffic.json was measured on other GEMV sources (hash differs)"

    out = {
        "metric": METRIC,
        "value": round(tok_s, 2),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak" if tp_fallback else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic random-init weights (GGUF block formats), fixed token-id prompt",
        "config": {"workload": f"{args.workload} {mix} greedy decode, batch 1, prompt {args.prompt_len}, "
                               f"positions {pos}..{pos + args.steps - 1}, f32 KV cache",
                   "parallelism": "single GPU" if world == 1 else
                                  f"FALLBACK: {world} whole-model replicas, no exchange (tensor-parallel set-up failed: {tp_fallback})"
                                  if tp_fallback else
                                  f"tp{world}: row/column-split projections, 2 all-reduces of f32[{cfg['hidden']}] "
                                  f"per layer + logits all-gather "
                                  f"({('one-shot peer-to-peer exchange over xGMI (hipIpc-mapped areas) ' + ('inside the attn_output / ffn_down / lm_head launches' if model.tp_peer_fused() else 'as one kernel each')) if model.tp_peer_enabled() else 'RCCL'}), "
                                  f"{'inside the captured step graph' if model.graph_captured() else 'eager launches (capture refused)'}",
                   "weight_bytes_per_token_per_gpu": int(w_bytes), "kv_bytes_per_token_mid_per_gpu": int(kv_bytes)},
        # per GPU: every rank streams its own 1/N of the bytes in the same step time
        "hbm_gbps_whole_step": round((w_bytes + kv_bytes) / (dt / args.steps) / 1e9, 1),
        "hbm_frac_whole_step": round((w_bytes + kv_bytes) / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
        "roofline": {"bound": "hbm", "kernel": "gemv_kernel<.., GEMV_GATE_UP> (RMSNorm + ffn_gate/ffn_up GEMV + SwiGLU)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "bytes_per_launch": int(gu_bytes), "avg_launch_us": round(gu_us, 2),
                     "timed_with": "HIP events on the engine's stream around back-to-back launches (achieved and frac are from these)",
                     # the same kernel's average duration in the kept rocprofv3 --kernel-trace --stats summary, and the
                     # fraction it gives (the profiler's per-dispatch time excludes the boundary the event pair includes
                     # and adds its own overhead: the two differ by a few percent)
                     "rocprof_avg_launch_us": rocprof_us,
                     "frac_rocprof": round(gu_bytes / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if rocprof_us else None,
                     "launches": 8 * cfg["n_layers"]},
        "gemv_kernels": kern,
        "eager_step_us": {k: round(v[0] * 1e3, 2) for k, v in prof.items()},
        "load_s": round(t_load, 1),
        # the expanded copy of the matrices kept for the batched legs below (tl_qweight_expand; 0: none): never read by
        # the timed decode step, whose bytes are the quantized planes counted above
        "prefill_planes_bytes_per_gpu": int(model.prefill_planes_bytes()),
    }
    if world > 1:  # the multi-rank line describes itself at the top level (the driver keeps only this line)
        ar = kern.get("tp_all_reduce", {})
        out["ranks"] = world
        out["via"] = "replicas (tensor-parallel set-up failed)" if tp_fallback else ar.get("via", "rccl")
        out["all_reduce_us"] = ar.get("us")
        out["all_reduces_per_step"] = 0 if tp_fallback else 2 * cfg["n_layers"]
        out["exchange_us_per_step"] = None if tp_fallback or ar.get("us") is None else round(2 * cfg["n_layers"] * ar["us"], 1)
        out["graph_captured"] = bool(model.graph_captured())
        fz = (not tp_fallback) and model.tp_peer_enabled() and model.tp_peer_fused()
        out["preflight_step_us_by_fused"] = preflight
        if not tp_fallback and tp_errors:
            out["tp_retries"] = tp_errors  # (the group that ran is the second attempt, on RCCL alone)
        out["launches_per_step"] = model.graph_nodes()  # (folded exchanges: the unsharded step's count)
        out["exchange_launches_per_step"] = 0 if tp_fallback or fz else 2 * cfg["n_layers"] + 1
        out["weight_bytes_per_token_per_rank"] = int(w_bytes)
        out["kv_bytes_per_token_per_rank"] = int(kv_bytes)
    if SHARE_GPU and world > 1:
        out["rehearsal"] = f"{world} ranks share ONE GPU (TL_BENCH_SHARE_GPU=1): control-flow check only, the numbers are not a measurement"

    # ---- batched prefill on the matrix cores (outside the timed decode region): 512 prompt tokens, FP32-grade 6-term form
    # (under tensor parallelism every rank runs it: the prefill's all-reduces are collective calls)
    if cfg["max_pos"] >= 512 + 8 and not args.no_prefill_leg:
        try:
            out["prefill"] = prefill_leg(model, S, cfg, mix, 512, world=tp_world)
        except Exception as e:  # the decode numbers above stay valid
            out["prefill"] = {"error": str(e)[:300]}

    # ---- multi-sequence decode (generate_batch's engine; outside the timed single-sequence region): B
    # sequences share every step's weight stream (skinny GEMMs on the matrix cores), FP32-grade 6-term products
    if tp_world == 1 and world == 1 and not args.no_batch_leg and cfg["max_pos"] >= 64:
        try:
            out["multi_sequence"] = batch_leg(model, cfg, args.batch_seqs)
        except Exception as e:
            out["multi_sequence"] = {"error": str(e)[:300]}

    # ---- CPU baseline: the oracle in the reference's structure on a bounded sample ------------------------------
    if want_cpu:
        out["cpu_baseline"] = cpu_baseline(cfg, keep, args.cpu_layers, w_bytes)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def gemv_sources_hash():
    """sha256 over the sources of the decode GEMV kernels (what a kept PMC figure was measured on)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("tl_gemv.hip", "tl_chunk.h", "tl_internal.h"):
        h.update(open(os.path.join(ROOT, "tinyllama.cpp_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def batch_leg(model, cfg, B, plen=16, steps=16, terms=6):
    """aggregate tokens/s of tl_model_batch_generate_greedy: B sequences (prompts of plen tokens), `steps` greedy steps"""
    rng = np.random.default_rng(2)
    model.batch_init(B)
    model.set_prefill(terms=terms, min_batch=8)
    prompts = [rng.integers(0, cfg["vocab"], plen).astype(np.int32) for _ in range(B)]
    t0 = time.perf_counter()
    _, first = model.batch_prefill_packed(prompts, want_logits=False)  # all prompts as one packed batch
    t_prefill = time.perf_counter() - t0
    seqs, pos = list(range(B)), [plen] * B
    model.batch_generate_greedy(first, pos, seqs, 2)  # warm-up (its cache rows are overwritten by the timed run)
    t0 = time.perf_counter()
    ids = model.batch_generate_greedy(first, pos, seqs, steps)
    dt = time.perf_counter() - t0
    assert ids.shape == (steps, B)
    return {"sequences": B, "steps": steps, "mfma_terms": terms, "ms_per_step": round(dt / steps * 1e3, 3),
            "tokens_per_s_aggregate": round(B * steps / dt, 1), "positions": f"{plen}..{plen + steps - 1}",
            "packed_prefill_ms": round(t_prefill * 1e3, 2), "prompt_tokens_total": B * plen,
            "note": "every step streams the weights once for all sequences; each sequence decodes as it would alone"}


def prefill_leg(model, S, cfg, mix, T, terms=6, reps=3, world=1):
    """tokens/s of tl_model_prefill (MFMA GEMMs over the quantized weights, FP32-MFMA attention) for a T-token prompt;
    executed bf16 MFMA flops = (MFMAs per product) x the algorithmic GEMM flops 2*T*(layer weight elements)."""
    toks = np.random.default_rng(1).integers(0, cfg["vocab"], T).astype(np.int32)
    model.set_prefill(terms=terms, min_batch=8)
    model.reset_kv()
    model.prefill(toks, 0)
    best = 1e9
    for _ in range(reps):
        model.reset_kv()
        t0 = time.perf_counter()
        model.prefill(toks, 0)
        best = min(best, time.perf_counter() - t0)
    elems = sum(r * c for n, t, r, c in S.tensor_plan(cfg, mix) if n.startswith("blk.") and t != S.F32)
    flops = 2.0 * T * elems / world  # per GPU
    # MFMAs per product: the integer-plane GEMM multiplies EXACT integer weights (one bf16 plane) with the activation in
    # 3 / 2 / 1 bf16 planes (FP32-grade / 2^-17 / plain bf16); TL_GEMM_INT=0 (exact-dequantization kernel) spends 6 / 3 / 1
    per_product = {6: 3, 3: 2, 1: 1}[terms] if os.environ.get("TL_GEMM_INT", "1") != "0" else terms
    return {"prompt_tokens": T, "mfma_terms": terms, "mfma_per_product": per_product,
            "precision": {6: "fp32-grade", 3: "2^-17 per product", 1: "bf16 activations"}[terms],
            "ms": round(best * 1e3, 2), "tokens_per_s": round(T / best, 1),
            "gemm_tflops_algorithmic": round(flops / best / 1e12, 1),
            "bf16_mfma_tflops_executed": round(flops * per_product / best / 1e12, 1),
            "frac_of_bf16_dense_peak": round(flops * per_product / best / 2.5e15, 4),
            "note": "whole prefill wall time incl. attention and glue (rank 0's clock; TFLOP/s per GPU); per-kernel MFMA "
                    "busy counters: profiles/"}


def cpu_baseline(cfg, keep, n_layers, w_bytes):
    """Times tlo_forward in ref_structure mode (whole-tensor FP32 dequantization on first use, Kahan-double
    mat-vec over OpenMP rows, eviction of layer l-2: what TinyLlamaModel::forward pays) for `n_layers` layers, once
    cold and once warm, plus the lm_head mat-vec, and scales to the full depth: (L-2) cold + 2 warm layers + tail."""
    from oracle.oracle import Oracle, OracleModel
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    o = Oracle()
    sub = dict(cfg, n_layers=n_layers, max_pos=16)
    om = OracleModel(o, sub, keep)
    om.set_ref_structure(True)
    x = om.lookup_embedding(1)
    t0 = time.perf_counter()
    om.forward(x, 0, n_layers_limit=n_layers, want_logits=False)  # cold: every tensor dequantized, as for the
    t_cold = (time.perf_counter() - t0) / n_layers                # L-2 layers evicted since the previous token
    t0 = time.perf_counter()
    om.forward(x, 1, n_layers_limit=n_layers, want_logits=False)  # warm: the last two layers keep their FP32 copy
    t_warm = (time.perf_counter() - t0) / n_layers
    om.forward(x, 2)  # first logits call dequantizes lm_head once (kept, weight_management.cpp:10-42)
    t0 = time.perf_counter()
    om.forward(x, 3)
    t_tail = max(time.perf_counter() - t0 - t_warm * n_layers, 0.0)  # final norm + lm_head mat-vec
    L = cfg["n_layers"]
    t_tok = t_cold * max(L - 2, 0) + t_warm * min(L, 2) + t_tail
    om.close()
    port = {"value": round(1.0 / t_tok, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{n_layers} of {cfg['n_layers']} layers x 2 tokens + lm_head, scaled to full depth; "
                      "reference structure (per-token whole-tensor dequant, Kahan-double mat-vec, l-2 eviction)",
            "s_per_token": round(t_tok, 3), "gbps": round(w_bytes / t_tok / 1e9, 3)}
    ref = reference_cpu_baseline(cfg, keep, w_bytes, threads)
    if ref is None:  # oracle/_ref did not travel: the port alone, with the ratio measured in the build container
        port["port_vs_reference"] = PORT_VS_REFERENCE_MEASURED
        port["port_vs_reference_source"] = "profiles/r02_c_cpu_baseline_validation.json (build container, 8 threads)"
        return port
    ref["port_value"] = port["value"]
    ref["port_vs_reference"] = round(port["value"] / ref["value"], 3)
    return ref


PORT_VS_REFERENCE_MEASURED = 1.6  # tools/validate_cpu_baseline.py, q4_k 14336x4096, best of several runs: 1.46-1.72


def reference_cpu_baseline(cfg, keep, w_bytes, threads):
    """The same per-token cost from the REFERENCE'S OWN compiled functions (oracle/_ref/libtlref.so, built from its
    quantization.cpp + utils.cpp): dequantize_vector_*_to_f32 then matvec_f32_f32_vector_cpu on each of layer 0's seven
    projection tensors (what ensure_*_dequantized + the projections of TinyLlamaModel::forward cost per layer and
    token) and on lm_head, composed as forward composes them: (L-2) layers pay dequantize + mat-vec, the two layers that
    keep their FP32 copy and lm_head pay the mat-vec only. Attention, norms, RoPE and SwiGLU (< 1 % of the port's time at
    this position) are left out, which flatters the reference slightly. None when the library is not there."""
    try:
        from oracle.oracle import Ref
        ref = Ref()
        if not ref.can_time():
            return None
    except Exception:
        return None
    rng = np.random.default_rng(7)
    deq = mv = 0.0
    used = []
    for name, (typ, arr, rows, cols) in keep.items():
        if not (name.startswith("blk.0.") or name == "output.weight") or arr.dtype != np.uint8:
            continue
        x = rng.standard_normal(cols).astype(np.float32)
        try:
            d, m, _ = ref.time_dequant_matvec(typ, arr, x, rows, cols, reps=1)
        except AssertionError:
            return None  # a format the reference has no whole-tensor dequantizer for
        if name == "output.weight":
            t_head = m
        else:
            deq, mv = deq + d, mv + m
        used.append(name)
    if "output.weight" not in used or len(used) < 2:
        return None
    L = cfg["n_layers"]
    t_tok = (deq + mv) * max(L - 2, 0) + mv * min(L, 2) + t_head
    return {"value": round(1.0 / t_tok, 4), "unit": "tokens/s", "cores": threads, "kind": "reference",
            "sample": "the reference's own dequantize_vector_*_to_f32 + matvec_f32_f32_vector_cpu (oracle/_ref, compiled "
                      f"from its sources) on layer 0's {len(used) - 1} projection tensors and lm_head, composed per token "
                      "as its forward does: (L-2) x (dequantize + mat-vec) + 2 x mat-vec + lm_head mat-vec; "
                      "attention/norm/rope not included",
            "s_per_token": round(t_tok, 3), "gbps": round(w_bytes / t_tok / 1e9, 3),
            "layer_dequant_s": round(deq, 3), "layer_matvec_s": round(mv, 3), "lm_head_matvec_s": round(t_head, 3)}


if __name__ == "__main__":
    sys.exit(main() or 0)
