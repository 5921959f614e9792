"""The N > 1 path on the CPU: world_size-2 (and 4) process groups over gloo run the tensor-parallel decode step with
tp.py's shards — per-rank mat-vecs on the shards through the oracle's primitives, sum all-reduce after attn_output and
ffn_down (rank 0 folds the residual in), all-gather of the vocab-sharded logits — and must reproduce the unsharded
oracle forward. This is the same sharding plan and the same collective placement the GPU engine uses (tl_model.hip);
there the collectives are RCCL calls, here gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGIT_TOL = 1e-4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, mix, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from _pkg import load_package
    from oracle.oracle import Oracle, OracleModel
    pkg = load_package()
    from tinyllama_cpp_amd import tp
    S = pkg.synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    cfg, tensors = S.build_model(shape, mix, seed=5)
    tp.check(cfg, world)
    H, L, V = cfg["hidden"], cfg["n_layers"], cfg["vocab"]
    nh, nkv = cfg["n_heads"], cfg["n_kv_heads"]
    hd = H // nh
    d = tp.local_dims(cfg, world)
    nh_l, nkv_l, I_l, V_l = d["n_heads"], d["n_kv_heads"], d["inter"], d["vocab"]
    sh = {n: tp.shard_tensor(n, t, a, r, c, rank, world) + (t,) for n, (t, a, r, c) in tensors.items()}

    def mv(name, x):
        a, r, c, t = sh[name]
        return o.matvec_q(t, a, x, r, c)

    def all_reduce(partial):
        t = torch.from_numpy(np.ascontiguousarray(partial))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    table = o.rope_table(cfg["max_pos"], hd, cfg["rope_theta"])
    kc = [np.zeros((cfg["max_pos"], nkv_l * hd), np.float32) for _ in range(L)]
    vc = [np.zeros((cfg["max_pos"], nkv_l * hd), np.float32) for _ in range(L)]
    om = OracleModel(o, cfg, tensors) if rank == 0 else None  # the unsharded reference, on rank 0 only
    toks = np.random.default_rng(2).integers(0, V, 6).astype(np.int32)
    worst = 0.0
    for pos, tok in enumerate(toks):
        emb_t, emb, er, ec = tensors["token_embd.weight"]
        x = o.dequantize(emb_t, emb.reshape(er, -1, emb.shape[-1])[int(tok)]).astype(np.float32)
        for l in range(L):
            xn = o.rmsnorm_vector(x, tensors[f"blk.{l}.attn_norm.weight"][1], cfg["eps"])
            q = o.rope_apply(mv(f"blk.{l}.attn_q.weight", xn), nh_l, hd, pos, table)
            k = o.rope_apply(mv(f"blk.{l}.attn_k.weight", xn), nkv_l, hd, pos, table)
            kc[l][pos], vc[l][pos] = k, mv(f"blk.{l}.attn_v.weight", xn)
            att = o.attention_decode(q, kc[l], vc[l], nh_l, nkv_l, hd, pos)
            part = mv(f"blk.{l}.attn_output.weight", att)
            x = all_reduce(part + x if rank == 0 else part)
            xn = o.rmsnorm_vector(x, tensors[f"blk.{l}.ffn_norm.weight"][1], cfg["eps"])
            h = o.swiglu(mv(f"blk.{l}.ffn_gate.weight", xn), mv(f"blk.{l}.ffn_up.weight", xn))
            part = mv(f"blk.{l}.ffn_down.weight", h)
            x = all_reduce(part + x if rank == 0 else part)
        xn = o.rmsnorm_vector(x, tensors["output_norm.weight"][1], cfg["eps"])
        mine = torch.from_numpy(np.ascontiguousarray(mv("output.weight", xn)))
        parts = [torch.empty(V_l, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, mine)
        logits = torch.cat(parts).numpy()
        if rank == 0:
            _, want = om.forward(om.lookup_embedding(int(tok)), pos)
            err = np.abs(logits - want) / np.maximum(1.0, np.abs(want))
            worst = max(worst, float(err.max()))
            assert int(np.argmax(logits)) == int(np.argmax(want))
    if rank == 0:
        with open(os.path.join(out_dir, "worst.txt"), "w") as f:
            f.write(repr(worst))
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,mix,world", [("tiny-gqa", "q4_k_m", 2), ("tiny-tp", "q6_k", 4)])
def test_tp_forward_over_gloo(tmp_path, shape, mix, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, shape, mix, str(tmp_path)), nprocs=world, join=True)
    worst = float(open(tmp_path / "worst.txt").read())
    assert worst <= LOGIT_TOL, worst


def test_shard_shapes_and_bytes():
    sys.path.insert(0, ROOT)
    from _pkg import load_package
    pkg = load_package()
    from tinyllama_cpp_amd import tp
    S = pkg.synth
    cfg = S.SHAPES["llama-3-8b"]
    for size in (1, 2, 4, 8):
        tp.check(cfg, size)
    with pytest.raises(ValueError):
        tp.check(S.SHAPES["llama-3.2-3b"], 8)  # 3072 / 8 = 384 is not a multiple of 256
    with pytest.raises(ValueError):
        tp.check(S.SHAPES["tinyllama-1.1b"], 8)  # 4 kv heads
    cfg, tensors = S.build_model("tiny-tp", "q4_k_m", seed=1)
    for name, (t, a, r, c) in tensors.items():
        total = 0
        for rank in range(4):
            al, rl, cl = tp.shard_tensor(name, t, a, r, c, rank, 4)
            kind = tp.kind_of(name)
            assert (rl, cl) == {"row": (r // 4, c), "col": (r, c // 4), "rep": (r, c)}[kind], name
            total += al.nbytes
        assert total == (a.nbytes if tp.kind_of(name) != "rep" else 4 * a.nbytes)
    # column shards are the right super-blocks: dequantize a shard == the column slice of the dequantized whole
    from oracle.oracle import Oracle
    o = Oracle()
    t, a, r, c = tensors["blk.0.ffn_down.weight"]
    whole = o.dequantize(t, a).reshape(r, c)
    al, rl, cl = tp.shard_tensor("blk.0.ffn_down.weight", t, a, r, c, 2, 4)
    assert np.array_equal(o.dequantize(t, al).reshape(rl, cl), whole[:, 2 * cl:3 * cl])
