"""Batched prefill on the matrix cores (MI355X), through the C-ABI, against the CPU oracle.

Bars: the prefill GEMM with 6 bf16 MFMAs per FP32 product (the default) meets the decode tolerance
(|delta| <= 1e-4 * max(1, |ref|) vs the Kahan-double oracle); the reduced forms are opt-in and carry their own stated
tolerances (3 terms: 1e-3, 1 term = plain bf16: 5e-2 of the output scale). Prefill attention (FP32 MFMA): 2e-5 absolute.
Whole-model prefill (all logits, KV contents) at the decode tolerance, incl. prompts longer than one chunk, ragged
lengths and a prefill that continues an existing cache."""
import numpy as np
import pytest

from oracle.oracle import OracleModel, Q4_K, Q6_K, Q8_0, Q8_K

pytestmark = pytest.mark.gpu
TYPES = {"q4_k": Q4_K, "q6_k": Q6_K, "q8_0": Q8_0, "q8_k": Q8_K}
TOL = {6: 1e-4, 3: 1e-3, 1: 5e-2}


@pytest.fixture(scope="module")
def E(pkg):
    from tinyllama_cpp_amd import engine
    engine.init(0)
    yield engine
    from tinyllama_cpp_amd.capi import lib
    lib().tl_gemm_set_terms(6)


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


def _matmul(E, w, x):
    from tinyllama_cpp_amd.capi import lib, check
    xb = E.DeviceBuffer.from_numpy(np.ascontiguousarray(x, np.float32))
    yb = E.DeviceBuffer(4 * x.shape[0] * w.rows)
    check(lib().tl_matmul_q(w.h, xb.ptr, yb.ptr, x.shape[0], None))
    return yb.to_numpy(np.float32, (x.shape[0], w.rows))


@pytest.mark.parametrize("name,rows,cols,T", [
    ("q4_k", 512, 1024, 8), ("q4_k", 333, 512, 33), ("q4_k", 1024, 2048, 130), ("q6_k", 256, 1024, 200),
    ("q6_k", 100, 256, 64), ("q8_0", 192, 512, 129), ("q8_0", 64, 32, 16), ("q8_k", 128, 768, 40)])
def test_gemm_vs_oracle(E, pkg, oracle, name, rows, cols, T):
    from tinyllama_cpp_amd.capi import lib, check
    t, S = TYPES[name], pkg.synth
    rng = np.random.default_rng(rows + 3 * cols + T)
    blocks = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
    x = rng.standard_normal((T, cols)).astype(np.float32)
    sc = np.ones((T, 1), np.float32)  # rows of very different scale share a tile; the bars scale with the row
    sc[0], sc[-1] = 2.0 ** -10, 64.0
    x *= sc
    want = oracle.matmul_q_batch(t, blocks, x, T, rows, cols)
    w = E.QWeight(t, blocks, rows, cols)
    for terms in (6, 3, 1):
        check(lib().tl_gemm_set_terms(terms))
        y = _matmul(E, w, x)
        if terms == 6:
            assert close(y / sc, want / sc, TOL[6]).all(), np.abs(y / sc - want / sc).max()
        else:  # reduced forms: relative to the scale of the token's output row
            rowmax = np.maximum(sc, np.abs(want).max(axis=1, keepdims=True))
            assert (np.abs(y - want) <= TOL[terms] * rowmax).all(), (terms, (np.abs(y - want) / rowmax).max())
    check(lib().tl_gemm_set_terms(6))
    # each token row only depends on its own input row: a batch of the same rows in another order gives the same rows
    perm = rng.permutation(T)
    assert np.array_equal(_matmul(E, w, x[perm]), _matmul(E, w, x)[perm])
    # fewer than 8 tokens: GEMV launches, bit-identical to the mat-vec
    y3 = _matmul(E, w, x[:3])
    assert np.array_equal(y3[1], w.matvec(x[1]))


@pytest.mark.parametrize("name,rows,cols,T", [
    ("q4_k", 333, 4096, 9), ("q4_k", 1024, 14336, 12), ("q6_k", 257, 1024, 32), ("q6_k", 4100, 4096, 40),
    ("q8_0", 100, 192, 17), ("q8_k", 130, 768, 20), ("q4_k", 96, 256, 64), ("q8_0", 2048, 2048, 50)])
def test_direct_small_token_gemm(E, pkg, oracle, monkeypatch, name, rows, cols, T):
    """gemm_qd_kernel (weights from global memory straight into the MFMA operand; launches of at most 32 tokens by
    default, 64 with TL_GEMM_DIRECT=64): every format, row counts off the 32-row fragments, K splits with many and with
    a single 64-column step, 32- and 64-token tiles -- against the oracle at the GEMM bars, and against the
    integer-plane kernel it stands in for (TL_GEMM_DIRECT=0): same factorisation, so FP32 rounding apart."""
    from tinyllama_cpp_amd.capi import lib, check
    t, S = TYPES[name], pkg.synth
    rng = np.random.default_rng(7 * rows + cols + T)
    blocks = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
    x = rng.standard_normal((T, cols)).astype(np.float32)
    sc = np.ones((T, 1), np.float32)
    sc[0], sc[-1] = 2.0 ** -10, 64.0
    x *= sc
    want = oracle.matmul_q_batch(t, blocks, x, T, rows, cols)
    w = E.QWeight(t, blocks, rows, cols)
    rowmax = np.maximum(sc, np.abs(want).max(axis=1, keepdims=True))
    out = {}
    for direct in ("64", "0"):
        monkeypatch.setenv("TL_GEMM_DIRECT", direct)
        for terms in (6, 3):
            check(lib().tl_gemm_set_terms(terms))
            y = _matmul(E, w, x)
            out[direct, terms] = y
            if terms == 6:
                assert close(y / sc, want / sc, TOL[6]).all(), (direct, np.abs(y / sc - want / sc).max())
            else:
                assert (np.abs(y - want) <= TOL[3] * rowmax).all(), (direct, (np.abs(y - want) / rowmax).max())
    check(lib().tl_gemm_set_terms(6))
    assert (np.abs(out["64", 6] - out["0", 6]) <= 2e-5 * rowmax).all()
    monkeypatch.setenv("TL_GEMM_DIRECT", "64")
    perm = rng.permutation(T)
    assert np.array_equal(_matmul(E, w, x[perm]), out["64", 6][perm])
    if T <= 32:  # the default limit takes the same kernel as the explicit one
        monkeypatch.delenv("TL_GEMM_DIRECT")
        assert np.array_equal(_matmul(E, w, x), out["64", 6])


@pytest.mark.parametrize("name,rows,cols,T", [
    ("q4_k", 4096, 4096, 2), ("q4_k", 333, 1024, 3), ("q4_k", 1024, 14336, 5), ("q6_k", 257, 4096, 4),
    ("q6_k", 64, 14336, 7), ("q8_0", 1000, 2048, 6), ("q8_0", 33, 32, 2), ("q8_k", 128, 768, 3), ("q4_k", 1, 256, 7)])
def test_few_tokens_take_the_multi_vector_gemv(E, pkg, oracle, name, rows, cols, T):
    """2..7 tokens: one pass over the weights for up to four tokens (gemv_nv_kernel; 14336 columns only stage two
    vectors at a time; Q8_K falls back to one mat-vec per token). Every row is bit-identical to the single mat-vec
    (same chunk dots, same lane order) and within the GEMV tolerance of the oracle."""
    t, S = TYPES[name], pkg.synth
    rng = np.random.default_rng(rows + cols + T)
    blocks = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
    x = rng.standard_normal((T, cols)).astype(np.float32)
    x[0] *= 1e-3
    x[-1] *= 50.0
    w = E.QWeight(t, blocks, rows, cols)
    y = _matmul(E, w, x)
    want = oracle.matmul_q_batch(t, blocks, x, T, rows, cols)
    sc = np.array([1e-3] + [1.0] * (T - 2) + [50.0], np.float32)[:, None]
    assert close(y / sc, want / sc, 1e-4).all(), np.abs(y / sc - want / sc).max()
    for i in range(T):
        assert np.array_equal(y[i], w.matvec(x[i])), i


@pytest.mark.parametrize("name,rows,cols,T", [
    ("q4_k", 512, 1024, 65), ("q4_k", 333, 512, 130), ("q4_k", 4096, 4096, 512), ("q4_k", 1024, 14336, 128),
    ("q6_k", 256, 1024, 200), ("q6_k", 4100, 4096, 257), ("q8_0", 192, 512, 129), ("q8_0", 2048, 2048, 384),
    ("q8_k", 128, 768, 140), ("q6_k", 1000, 256, 100), ("q4_k", 4096, 4096, 64), ("q4_k", 700, 14336, 40),
    ("q6_k", 1024, 4096, 50), ("q8_0", 300, 512, 33)])
def test_expanded_plane_gemm_is_bit_identical(E, pkg, oracle, monkeypatch, name, rows, cols, T):
    """tl_qweight_expand: the batched GEMM fed by LDS-DMA from the resident bf16 integer plane + scale words
    (gemm_px_kernel, 256-row tiles, producers that only issue DMAs) forms exactly the sums of the kernel that unpacks the
    quantized planes in every workgroup (gemm_qi_ws_kernel): every format, row counts off the 256-row tile, K splits (few
    tiles) and none (many), token counts off the 128-token tile, 3- and 6-term products -- bit for bit, and within the GEMM
    bar of the oracle."""
    from tinyllama_cpp_amd.capi import lib, check
    t, S = TYPES[name], pkg.synth
    rng = np.random.default_rng(11 * rows + cols + T)
    blocks = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
    x = rng.standard_normal((T, cols)).astype(np.float32)
    sc = np.ones((T, 1), np.float32)
    sc[0], sc[-1] = 2.0 ** -10, 64.0
    x *= sc
    w = E.QWeight(t, blocks, rows, cols)
    before = {}
    for terms in (6, 3):
        check(lib().tl_gemm_set_terms(terms))
        before[terms] = _matmul(E, w, x)
    def px_launches():
        import ctypes as C
        n = C.c_longlong()
        check(lib().tl_debug_gemm_px_launches(C.byref(n)))
        return n.value

    monkeypatch.setenv("TL_GEMM_PX_SLICES", "0")  # the unpacking kernel's K slices: then the sums agree bit for bit
    n0 = px_launches()
    nbytes = w.expand()
    mt, nk = (rows + 255) // 256, cols // 32
    assert nbytes == mt * nk * (256 * 32 * 2 + 3 * 256 * 4)
    for terms in (6, 3):
        check(lib().tl_gemm_set_terms(terms))
        y = _matmul(E, w, x)
        assert np.array_equal(y, before[terms]), (terms, np.abs(y - before[terms]).max())
        n1 = px_launches()
        monkeypatch.setenv("TL_GEMM_PX", "0")  # the planes are there but ignored: the unpacking kernel again
        assert np.array_equal(_matmul(E, w, x), before[terms])
        monkeypatch.delenv("TL_GEMM_PX")
        assert px_launches() == n1
    assert n1 == n0 + 2  # (one launch per product: the expanded-plane kernel formed y)
    monkeypatch.setenv("TL_GEMM_PX_BM", "128")  # half a record per workgroup (the form taken for few row tiles): same sums
    assert np.array_equal(_matmul(E, w, x), before[3])
    monkeypatch.delenv("TL_GEMM_PX_BM")
    assert px_launches() == n1 + 1
    check(lib().tl_gemm_set_terms(6))
    # the kernel's own K slices (the default): other partial sums, FP32 rounding apart
    monkeypatch.delenv("TL_GEMM_PX_SLICES")
    y6 = _matmul(E, w, x)
    rowmax = np.maximum(sc, np.abs(before[6]).max(axis=1, keepdims=True))
    assert (np.abs(y6 - before[6]) <= 2e-5 * rowmax).all()
    if rows * cols * T <= 4096 * 4096 * 64:  # (the oracle's Kahan-double dots: seconds)
        want = oracle.matmul_q_batch(t, blocks, x, T, rows, cols)
        assert close(y / sc, want / sc, TOL[3]).all()
        assert close(y6 / sc, want / sc, TOL[6]).all()
    w.free()


def test_gemm_edge_scales(E, pkg, oracle):
    """fp16 sub-normal / zero / negative super-scales inside an MFMA tile."""
    from tinyllama_cpp_amd.capi import lib, check
    S = pkg.synth
    rng = np.random.default_rng(4)
    rows, cols, T = 64, 512, 24
    x = rng.standard_normal((T, cols)).astype(np.float32)
    check(lib().tl_gemm_set_terms(6))
    for t in (Q4_K, Q6_K, Q8_0, Q8_K):
        blocks = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=0.02)
        edge = S.edge_blocks(t, rng)
        keep = [i for i, h in enumerate(S.EDGE_F16) if (h & 0x7c00) != 0x7c00 and (h & 0x7fff) < 0x7000]
        blocks[: len(keep)] = edge[keep]
        want = oracle.matmul_q_batch(t, blocks, x, T, rows, cols)
        y = _matmul(E, E.QWeight(t, blocks, rows, cols), x)
        assert np.isfinite(want).all()
        assert close(y, want, 1e-4 * max(1.0, np.abs(want).max())).all()


@pytest.mark.parametrize("nh,nkv,hd,T,start", [(4, 2, 64, 33, 0), (8, 2, 64, 100, 57), (32, 8, 128, 70, 0),
                                               (16, 4, 128, 45, 200), (8, 8, 128, 1, 31),
                                               # pairs of query tiles: an odd tile count (the middle tile alone), heads not
                                               # a multiple of 8, several key tiles per wave (the K prefetch), the bench shape
                                               (8, 2, 128, 160, 0), (12, 4, 128, 300, 100), (8, 2, 64, 257, 300),
                                               (32, 8, 128, 512, 0)])
def test_prefill_attention_vs_oracle(E, oracle, nh, nkv, hd, T, start):
    from tinyllama_cpp_amd.capi import lib, check
    rng = np.random.default_rng(nh * 100 + T)
    npos = start + T
    max_seq = npos + 3
    kc = np.zeros((max_seq, nkv * hd), np.float32)
    vc = np.zeros((max_seq, nkv * hd), np.float32)
    kc[:npos] = rng.standard_normal((npos, nkv * hd))
    vc[:npos] = rng.standard_normal((npos, nkv * hd))
    Q = rng.standard_normal((T, nh * hd)).astype(np.float32)
    dev = lambda a: E.DeviceBuffer.from_numpy(np.ascontiguousarray(a, np.float32))
    kb, vb, Qb, Ob = dev(kc), dev(vc), dev(Q), E.DeviceBuffer(Q.nbytes)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(hd)))
    check(lib().tl_attention_batch_prefill(Qb.ptr, kb.ptr, vb.ptr, Ob.ptr, T, start, max_seq, nh, nkv, hd, scale, None))
    got = Ob.to_numpy(np.float32, Q.shape)
    want = oracle.attention_batch(Q, kc, vc, T, start, nh, nkv, hd)
    assert np.allclose(got, want, rtol=0, atol=2e-5), np.abs(got - want).max()


@pytest.mark.parametrize("shape,mix,T", [("tiny", "q4_k_m", 600), ("tiny-gqa", "q6_k", 129), ("tiny-gqa", "q8_0", 8),
                                         ("tiny-tp", "q4_k_m", 77)])
def test_model_prefill_batched_vs_oracle(E, pkg, oracle, shape, mix, T):
    cfg = dict(pkg.synth.SHAPES[shape], max_pos=max(640, T + 8) if T > 256 else pkg.synth.SHAPES[shape]["max_pos"])
    cfg, tensors = pkg.synth.build_model(cfg, mix, seed=8)
    m = E.Model(cfg).load(tensors)
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(T).integers(0, cfg["vocab"], T).astype(np.int32)
    last, allb = m.prefill(toks, 0, all_logits=True)
    X = np.stack([om.lookup_embedding(t) for t in toks])
    _, lb = om.forward_batch(X, 0)
    assert close(allb, lb, TOL[6]).all(), np.abs(allb - lb).max()
    assert close(last, lb[-1], TOL[6]).all()
    for layer in (0, cfg["n_layers"] - 1):
        k, v = m.kv_cache(layer, T)
        assert close(k, om.kcache(layer, T), TOL[6]).all()
        assert close(v, om.vcache(layer, T), TOL[6]).all()
    # greedy continuation from the prefilled cache == the oracle's continuation from its own cache
    if T + 6 <= cfg["max_pos"]:
        first = int(np.argmax(lb[-1]))
        got = m.generate_greedy(first, T, 6, 1.0, history=[first])
        tok, want = first, []
        for i in range(6):
            _, lg = om.forward(om.lookup_embedding(tok), T + i)
            tok = int(np.argmax(lg))
            want.append(tok)
        assert got.tolist() == want
    m.free()
    om.close()


def test_model_prefill_continues_cache_and_reduced_terms(E, pkg, oracle):
    """A second prefill call appends to the cache (start_pos > 0); the token-by-token route (min_batch above the
    prompt length) and the batched route agree; the reduced-precision forms stay within their stated tolerances."""
    cfg, tensors = pkg.synth.build_model("tiny-gqa", "q4_k_m", seed=3)
    m = E.Model(cfg).load(tensors)
    om = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(6).integers(0, cfg["vocab"], 90).astype(np.int32)
    X = np.stack([om.lookup_embedding(t) for t in toks])
    _, lb = om.forward_batch(X, 0)
    m.prefill(toks[:50], 0)
    last = m.prefill(toks[50:], 50)
    assert close(last, lb[-1], TOL[6]).all(), np.abs(last - lb[-1]).max()
    m.reset_kv()
    m.set_prefill(terms=0, min_batch=1000)
    serial = m.prefill(toks, 0)
    assert close(serial, lb[-1], TOL[6]).all()
    for terms in (3, 1):
        m.reset_kv()
        m.set_prefill(terms=terms, min_batch=8)
        got = m.prefill(toks, 0)
        scale = max(1.0, np.abs(lb[-1]).max())
        assert (np.abs(got - lb[-1]) <= TOL[terms] * scale).all(), (terms, np.abs(got - lb[-1]).max())
    from tinyllama_cpp_amd.capi import TlError
    with pytest.raises(TlError):
        m.set_prefill(terms=2)
    with pytest.raises(TlError):
        m.prefill(toks, cfg["max_pos"] - 10)  # runs past the cache
    m.free()
    om.close()


def test_prefill_attention_writes_the_gemm_planes_bit_identically(E, pkg, monkeypatch):
    """The prefill attention kernel writes the attn_output GEMM's activation planes itself (no split launch); every logit of
    an all-logits prefill -- one call, and a second call appending to the cache, with 6 and 3 MFMA terms -- is bit for bit
    what TL_ATTN_PLANES=0 (attention -> float rows -> split launch) gives."""
    rng = np.random.default_rng(12)
    for shape, mix, n in (("tiny-gqa", "q4_k_m", 150), ("tiny-tp", "q6_k", 77)):
        cfg, tensors = pkg.synth.build_model(shape, mix, seed=4)
        toks = rng.integers(0, cfg["vocab"], n).astype(np.int32)
        got = {}
        for ap in ("0", "1"):
            monkeypatch.setenv("TL_ATTN_PLANES", ap)
            m = E.Model(cfg).load(tensors)
            out = []
            for terms in (6, 3):
                m.reset_kv()
                m.set_prefill(terms=terms, min_batch=8)
                _, a = m.prefill(toks[:n - 40], 0, all_logits=True)
                _, b = m.prefill(toks[n - 40:], n - 40, all_logits=True)
                out += [np.array(a), np.array(b)]
            got[ap] = out
            m.free()
        for a, b in zip(got["0"], got["1"]):
            assert a.shape == b.shape and np.array_equal(a, b), (shape, np.abs(a - b).max())
