"""Tensor-parallel shards on ONE GPU (MI355X): all shards of a tp_size-way model live in this process and step in
lockstep through tl_tp_local_group_decode (collectives as plain kernels). Exercises tp.py's slicing, the local shapes
in tl_model, the partial + residual-on-rank-0 scheme, the logits all-gather placement and per-shard sampling, against
the unsharded CPU oracle. RCCL itself only enters when the ranks are separate processes (bench.py --gpus N)."""
import numpy as np
import pytest

from oracle.oracle import OracleModel

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


@pytest.mark.parametrize("shape,mix,tp", [("tiny-gqa", "q4_k_m", 2), ("tiny-tp", "q4_k_m", 2), ("tiny-tp", "q4_k_m", 4),
                                          ("tiny-tp", "q6_k", 4), ("tiny-tp", "q8_0", 2)])
def test_tp_local_group_matches_oracle(pkg, oracle, shape, mix, tp):
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=5)
    shards = [E.Model(cfg, tp_rank=r, tp_size=tp).load(tensors) for r in range(tp)]
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(2).integers(0, cfg["vocab"], 9).astype(np.int32)
    om.reset_kv()
    for pos, tok in enumerate(toks):
        _, want = om.forward(om.lookup_embedding(int(tok)), pos)
        logits, nxt = E.tp_local_group_decode(shards, int(tok), pos)
        assert close(logits, want, LOGIT_TOL).all(), (pos, np.abs(logits - want).max())
        assert nxt == int(np.argmax(want))
    # each shard's KV cache holds its own kv heads of the oracle's cache
    hd = cfg["hidden"] // cfg["n_heads"]
    kvl = cfg["n_kv_heads"] // tp * hd
    for r, m in enumerate(shards):
        kvd = np.empty((len(toks), kvl), np.float32)
        from tinyllama_cpp_amd.capi import check, lib
        check(lib().tl_memcpy_d2h(kvd.ctypes.data, m.device_ptr(0, 0), kvd.nbytes, None), "d2h")
        want_k = om.kcache(0, len(toks))[:, r * kvl:(r + 1) * kvl]
        assert close(kvd, want_k, LOGIT_TOL).all()
        w, _ = m.bytes_per_token(3)
        assert w < pkg.synth.weight_bytes_per_token(cfg, mix)  # a shard reads a fraction of the bytes
    for m in shards:
        m.free()
    om.close()


def test_tp_rejects_bad_splits(pkg):
    from tinyllama_cpp_amd import engine as E
    cfg = dict(pkg.synth.SHAPES["tiny-gqa"])
    with pytest.raises(ValueError):
        E.Model(cfg, tp_rank=0, tp_size=4)  # 2 kv heads do not split 4 ways
    with pytest.raises(ValueError):
        E.Model(dict(pkg.synth.SHAPES["tiny"]), tp_rank=0, tp_size=2)  # 128-column shards of attn_output


def test_rccl_calls_capture_on_one_rank(pkg, monkeypatch):
    """The RCCL all-reduce / all-gather calls of the step, and their capture into the decode graph, on a one-rank
    communicator (TL_TP_SELFTEST): the token stream must be the plain single-GPU one."""
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg, tensors = pkg.synth.build_model("tiny-gqa", "q4_k_m", seed=5)
    plain = E.Model(cfg).load(tensors)
    want = plain.generate_greedy(3, 0, 24)
    monkeypatch.setenv("TL_TP_SELFTEST", "1")
    m = E.Model(cfg)
    for name, (typ, arr, rows, cols) in tensors.items():
        m.set_tensor(name, typ, arr, rows, cols)
    m.tp_init(E.tp_unique_id())
    m.finalize()
    got = m.generate_greedy(3, 0, 24)
    assert np.array_equal(got, want)
    assert m.graph_captured(), "RCCL collectives were not captured into the decode graph (eager fallback)"
    m.free()
    plain.free()


@pytest.mark.parametrize("shape,mix,tp,T", [("tiny-gqa", "q4_k_m", 2, 70), ("tiny-tp", "q4_k_m", 4, 33), ("tiny-tp", "q6_k", 2, 100)])
def test_tp_local_group_prefill_matches_oracle(pkg, oracle, shape, mix, tp, T):
    """Batched (MFMA) prefill of a sharded model: per-rank GEMMs on the shards, one T x hidden all-reduce per projection
    pair, vocabulary-sharded last-token logits; then sharded decode continues from the sharded caches."""
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=9)
    shards = [E.Model(cfg, tp_rank=r, tp_size=tp).load(tensors) for r in range(tp)]
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(T).integers(0, cfg["vocab"], T).astype(np.int32)
    X = np.stack([om.lookup_embedding(t) for t in toks])
    _, lb = om.forward_batch(X, 0)
    last = E.tp_local_group_prefill(shards, toks, 0)
    assert close(last, lb[-1], LOGIT_TOL).all(), np.abs(last - lb[-1]).max()
    hd = cfg["hidden"] // cfg["n_heads"]
    kvl = cfg["n_kv_heads"] // tp * hd
    from tinyllama_cpp_amd.capi import check, lib
    for r, m in enumerate(shards):
        kvd = np.empty((T, kvl), np.float32)
        check(lib().tl_memcpy_d2h(kvd.ctypes.data, m.device_ptr(0, cfg["n_layers"] - 1), kvd.nbytes, None), "d2h")
        assert close(kvd, om.kcache(cfg["n_layers"] - 1, T)[:, r * kvl:(r + 1) * kvl], LOGIT_TOL).all()
    tok = int(np.argmax(lb[-1]))
    for i in range(4):  # decode goes on from the prefilled, sharded caches
        _, want = om.forward(om.lookup_embedding(tok), T + i)
        logits, nxt = E.tp_local_group_decode(shards, tok, T + i)
        assert close(logits, want, LOGIT_TOL).all()
        assert nxt == int(np.argmax(want))
        tok = nxt
    for m in shards:
        m.free()
    om.close()


def test_tp8_at_the_70b_shard_shapes(pkg, oracle):
    """BASELINE config 5 (Llama-3-70B-shaped, TP = 8) at its real per-rank shapes — hidden 8192, 8 query heads + 1 kv head
    and 3584 intermediate columns per rank — cut to 2 layers and a 4096-token vocabulary so that the CPU oracle can
    follow: decode in lockstep over all 8 shards on one GPU, then a batched prefill."""
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg = dict(pkg.synth.SHAPES["llama-3-70b"], n_layers=2, vocab=4096, max_pos=64)
    cfg, tensors = pkg.synth.build_model(cfg, "q4_k_m", seed=70)
    shards = [E.Model(cfg, tp_rank=r, tp_size=8).load(tensors) for r in range(8)]
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(70).integers(0, cfg["vocab"], 4).astype(np.int32)
    for pos, tok in enumerate(toks):
        _, want = om.forward(om.lookup_embedding(int(tok)), pos)
        logits, nxt = E.tp_local_group_decode(shards, int(tok), pos)
        assert close(logits, want, LOGIT_TOL).all(), (pos, np.abs(logits - want).max())
        assert nxt == int(np.argmax(want))
    ptoks = np.random.default_rng(71).integers(0, cfg["vocab"], 24).astype(np.int32)
    for m in shards:
        m.reset_kv()
    om.reset_kv()
    _, lb = om.forward_batch(np.stack([om.lookup_embedding(t) for t in ptoks]), 0)
    last = E.tp_local_group_prefill(shards, ptoks, 0)
    assert close(last, lb[-1], LOGIT_TOL).all(), np.abs(last - lb[-1]).max()
    for m in shards:
        m.free()
    om.close()
