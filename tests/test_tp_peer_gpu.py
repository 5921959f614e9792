"""The peer-to-peer exchange of the tensor-parallel decode step (csrc/tl_peer.hip) between SEPARATE PROCESSES: tp worker
processes share GPU 0, map each other's receive areas through hipIpc handles, pass the collective self-check and decode
one sequence, every process replaying its own captured step graph. On one device the words cross between processes (and
between the 8 L2s of the chip) instead of xGMI links; protocol, epochs, slot reuse, bounded polls and the wiring are the
same code that bench.py --gpus N runs. Results vs the unsharded CPU oracle; all ranks must agree bit for bit.
Both forms of the exchange run: folded into the attn_output / ffn_down / lm_head launches (the default; the step then has
exactly the launches of the unsharded step) and as launches of their own (TL_TP_PEER_FUSED=0). A folded exchange waits
inside a GEMV launch for the same launch of the other ranks, so ranks that share ONE device cap their GEMV grids
(TL_GEMV_CUS) to fit on the chip together; with a device per rank that is not needed."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.oracle import OracleModel

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4
HERE = os.path.dirname(os.path.abspath(__file__))


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


def _unsharded_nodes(pkg, shape, mix):
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=5)
    m = E.Model(cfg)
    m.load(tensors)
    n = m.graph_nodes()
    m.free()
    return n


@pytest.mark.parametrize("shape,mix,tp,fused", [("tiny-gqa", "q4_k_m", 2, 1), ("tiny-tp", "q4_k_m", 4, 1),
                                                ("tiny-gqa", "q6_k", 2, 1), ("tiny-gqa", "q4_k_m", 2, 0),
                                                ("tiny-tp", "q4_k_m", 4, 0)])
def test_peer_linked_processes_match_oracle(pkg, oracle, tmp_path, shape, mix, tp, fused):
    env = dict(os.environ, TL_TP_PEER_TIMEOUT_MS="4000", TL_TP_PEER_FUSED=str(fused), TL_GEMV_CUS=str(128 // tp))
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_tp_peer_worker.py"), str(r), str(tp), shape, mix,
                               str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(tp)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out.decode(errors="replace")[-2000:])
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(f"rank {i} rc {p.returncode}\n{o}" for i, (p, o) in enumerate(zip(procs, outs)))
    res = [dict(np.load(os.path.join(tmp_path, f"result_{r}.npz"))) for r in range(tp)]
    for r in res:
        assert bool(r["linked"]) and bool(r["peer"]) and bool(r["captured"]) and bool(r["fused"]) == bool(fused)
    # folded: no launch is an exchange; as launches: 2 per layer + the logits gather on top of the unsharded step
    extra = 0 if fused else 2 * pkg.synth.SHAPES[shape]["n_layers"] + 1
    assert all(int(r["nodes"]) == _unsharded_nodes(pkg, shape, mix) + extra for r in res), [int(r["nodes"]) for r in res]
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=5)
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(2).integers(0, cfg["vocab"], 9).astype(np.int32)
    om.reset_kv()
    want = np.stack([om.forward(om.lookup_embedding(int(t)), pos)[1] for pos, t in enumerate(toks)])
    assert close(res[0]["logits"], want, LOGIT_TOL).all(), np.abs(res[0]["logits"] - want).max()
    tok, ids = int(np.argmax(want[-1])), []
    for i in range(24):
        _, lg = om.forward(om.lookup_embedding(tok), len(toks) + i)
        tok = int(np.argmax(lg))
        ids.append(tok)
    assert np.array_equal(res[0]["ids"], np.array(ids, np.int32))
    om.reset_kv()
    ptoks = np.random.default_rng(3).integers(0, cfg["vocab"], 40).astype(np.int32)
    _, lb = om.forward_batch(np.stack([om.lookup_embedding(t) for t in ptoks]), 0)
    assert close(res[0]["last"], lb[-1], LOGIT_TOL).all()
    for r in res[1:]:  # the rank-ordered sums make every rank's results identical
        assert np.array_equal(r["logits"], res[0]["logits"]) and np.array_equal(r["ids"], res[0]["ids"])
        assert np.array_equal(r["last"], res[0]["last"])
    om.close()


def _run_workers(tmp_path, tp, shape, mix, extra=(), env_extra=None):
    env = dict(os.environ, TL_TP_PEER_TIMEOUT_MS="4000", TL_GEMV_CUS=str(128 // tp))
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_tp_peer_worker.py"), str(r), str(tp), shape, mix,
                               str(tmp_path), *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(tp)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out.decode(errors="replace")[-2000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return procs, outs


def test_missing_rank_is_an_error_not_a_hang(tmp_path):
    """Rank 1 links but never steps: rank 0's decode gives up at the poll deadline and returns an error."""
    procs, outs = _run_workers(tmp_path, 2, "tiny-gqa", "q4_k_m", extra=("rank1-absent",),
                               env_extra={"TL_TP_PEER_TIMEOUT_MS": "300"})
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    r0 = dict(np.load(os.path.join(tmp_path, "result_0.npz")))
    assert "timed out" in str(r0["outcome"]), r0["outcome"]


def test_late_rank_at_finalize_is_reported(tmp_path):
    """Rank 1 reaches tl_model_finalize 1.5 s after rank 0 with a 300 ms poll deadline: the warm-up step's exchange gives up
    and finalize returns the error instead of leaving the flag for the first decode to trip over."""
    procs, outs = _run_workers(tmp_path, 2, "tiny-gqa", "q4_k_m", extra=("rank1-late",),
                               env_extra={"TL_TP_PEER_TIMEOUT_MS": "300"})
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    r0 = dict(np.load(os.path.join(tmp_path, "result_0.npz")))
    assert "tl_model_finalize" in str(r0["outcome"]) and "timed out" in str(r0["outcome"]), r0["outcome"]


def test_late_rank_behind_a_host_barrier_is_fine(tmp_path):
    procs, outs = _run_workers(tmp_path, 2, "tiny-gqa", "q4_k_m", extra=("rank1-late-barrier",),
                               env_extra={"TL_TP_PEER_TIMEOUT_MS": "300"})
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(outs)
    res = [dict(np.load(os.path.join(tmp_path, f"result_{r}.npz"))) for r in range(2)]
    assert all(str(r["outcome"]) == "ok" for r in res), [str(r["outcome"]) for r in res]
    assert res[0]["logits"].size > 1 and np.array_equal(res[0]["logits"], res[1]["logits"])


def test_import_rejects_garbage_handles(pkg):
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd.capi import TlError
    E.init(0)
    cfg = dict(pkg.synth.SHAPES["tiny-gqa"])
    m = E.Model(cfg, tp_rank=0, tp_size=2)
    with pytest.raises(TlError):
        m.tp_peer_import([b"\0" * 64, b"\0" * 64])  # nothing exported yet
    handle = m.tp_peer_export()
    assert len(handle) == 64
    with pytest.raises(TlError):
        m.tp_peer_import([handle, bytes(range(64))])  # rank 1's handle is not a handle
    with pytest.raises(TlError):
        m.tp_peer_enable(True)  # exported but never linked: enabling would skip exchanges that cannot run
    assert not m.tp_peer_enabled()
    with pytest.raises(TlError):
        E.Model(dict(pkg.synth.SHAPES["tiny"]), tp_rank=0, tp_size=1).tp_peer_export()  # tp_size 1 has nothing to link
    m.free()
