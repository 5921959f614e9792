"""The CPU oracle (oracle/tl_oracle.c) against (a) the committed golden vectors, which are outputs of
the reference's own code (oracle/make_golden.py), and (b) the live reference build oracle/_ref when
present. Bit-exact everywhere: same compiler flags, same libm."""
import numpy as np
import pytest

from oracle.oracle import OracleModel, Ref, Q4_K, Q6_K, Q8_0, Q8_K

F32 = 0

TYPES = {"q4_k": Q4_K, "q6_k": Q6_K, "q8_0": Q8_0, "q8_k": Q8_K}


def bits_equal_or_both_nan(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_fp16_all_values(oracle, golden):
    g = golden["g0_fp16"]
    for h in range(0, 65536):
        assert oracle.fp16_to_fp32_bits(h, False) == int(g["plain"][h])
        assert oracle.fp16_to_fp32_bits(h, True) == int(g["scale"][h])


def test_fp32_to_fp16(oracle, golden):
    g = golden["g0_fp16"]
    for f, h in zip(g["f32_in"], g["f16_out"]):
        assert oracle.fp32_to_fp16(f) == int(h)


@pytest.mark.parametrize("name", list(TYPES))
def test_dequant_bit_exact(oracle, golden, name):
    g = golden["g1_dequant"]
    got = oracle.dequantize(TYPES[name], g[name + "_blocks"])
    assert bits_equal_or_both_nan(got, g[name + "_f32bits"].view(np.float32))


@pytest.mark.parametrize("name", list(TYPES))
def test_gemv_and_gemm(oracle, golden, name):
    g = golden["g2_ops"]
    for rows, cols in ((256, 256), (512, 256), (96, 512)):
        k = f"gemv_{name}_{rows}x{cols}"
        y = oracle.matvec_q(TYPES[name], g[k + "_w"], g[k + "_x"], rows, cols)
        assert np.array_equal(y.view(np.uint32), g[k + "_y"].view(np.uint32))
    if name != "q8_k":
        k = f"gemm_{name}"
        y = oracle.matmul_q_batch(TYPES[name], g[k + "_w"], g[k + "_x"], 5, 64, 256)
        assert np.array_equal(y.view(np.uint32), g[k + "_y"].view(np.uint32))


def test_glue_ops(oracle, golden):
    g = golden["g2_ops"]
    for i, eps in enumerate(g["rms_eps"]):
        assert np.array_equal(oracle.rmsnorm_vector(g["rms_x"], g["rms_w"], float(eps)), g[f"rms_vec_{i}"])
        assert np.array_equal(oracle.rmsnorm_batch(g["rms_x"].reshape(8, 256), g["rms_w"][:256], float(eps)),
                              g[f"rms_batch_{i}"])
    for hd, nh in ((64, 4), (128, 3)):
        for pos in (0, 1, 37, 95, 96):
            x, tab = g[f"rope{hd}_x"], g[f"rope{hd}_table"]
            assert np.array_equal(oracle.rope_apply(x, nh, hd, pos, tab, True), g[f"rope{hd}_adj_{pos}"])
            assert np.array_equal(oracle.rope_apply(x, nh, hd, pos, tab, False), g[f"rope{hd}_half_{pos}"])
        assert np.array_equal(g[f"rope{hd}_adj_96"], g[f"rope{hd}_x"])  # pos >= max_pos: untouched
    q, k = g["ropeb_q"].copy(), g["ropeb_k"].copy()
    for t in range(6):
        q[t] = oracle.rope_apply(q[t], 4, 64, 30 + t, g["rope64_table"], True)
        k[t] = oracle.rope_apply(k[t], 2, 64, 30 + t, g["rope64_table"], True)
    assert np.array_equal(q, g["ropeb_q_out"]) and np.array_equal(k, g["ropeb_k_out"])
    assert np.array_equal(oracle.softmax(g["softmax_x"]), g["softmax_y"])
    assert np.array_equal(oracle.silu(g["silu_x"]), g["silu_y"])
    assert np.array_equal(oracle.quantize_q8_k(g["q8kq_x"]), g["q8kq_blocks"])


def test_rope_table_formula(oracle):
    # gpu_initialization.cpp:48-58 restated; spot-check against float32 numpy of the same expression
    t = oracle.rope_table(16, 64, 10000.0)
    for pos in (0, 3, 15):
        for i in (0, 2, 62):
            freq = np.float32(np.float32(10000.0) ** np.float32(-np.float32(i) / np.float32(64)))
            ang = np.float32(np.float32(pos) * freq)
            assert abs(t[pos, i // 2, 0] - np.cos(ang, dtype=np.float32)) <= 1e-6
            assert abs(t[pos, i // 2, 1] - np.sin(ang, dtype=np.float32)) <= 1e-6


def test_attention_decode_matches_batch_and_numpy(oracle):
    rng = np.random.default_rng(3)
    nh, nkv, hd, npos = 8, 2, 64, 19
    kc = rng.standard_normal((npos, nkv * hd)).astype(np.float32)
    vc = rng.standard_normal((npos, nkv * hd)).astype(np.float32)
    q = rng.standard_normal((npos, nh * hd)).astype(np.float32)
    dec = np.stack([oracle.attention_decode(q[t], kc, vc, nh, nkv, hd, t) for t in range(npos)])
    bat = oracle.attention_batch(q, kc, vc, npos, 0, nh, nkv, hd)
    assert np.allclose(dec, bat, rtol=0, atol=1e-6)
    # float64 numpy statement of model.cpp:632-663
    t = npos - 1
    for h in range(nh):
        kvh = h // (nh // nkv)
        K = kc[: t + 1, kvh * hd:(kvh + 1) * hd].astype(np.float64)
        V = vc[: t + 1, kvh * hd:(kvh + 1) * hd].astype(np.float64)
        s = K @ q[t, h * hd:(h + 1) * hd].astype(np.float64) / np.sqrt(hd)
        p = np.exp(s - s.max())
        p /= p.sum()
        assert np.allclose(dec[t, h * hd:(h + 1) * hd], p @ V, atol=2e-6)


def test_sampling_glue(oracle):
    v = np.array([0.5, 2.0, 2.0, -1.0], np.float32)
    assert oracle.argmax(v) == 1  # first maximal element (api.cpp:79)
    l = np.array([2.0, -2.0, 1.0, 4.0], np.float32)
    out = oracle.apply_repetition_penalty(l, [0, 1, 7, -1], 1.3)
    assert np.array_equal(out, np.array([np.float32(2.0) / np.float32(1.3), np.float32(-2.0) * np.float32(1.3), 1.0, 4.0], np.float32))
    assert np.array_equal(oracle.apply_repetition_penalty(l, [0, 1], 1.0), l)  # penalty <= 1: no-op
    gen = list(range(70)) + [3]
    big = np.ones(128, np.float32)
    out = oracle.apply_repetition_penalty(big, gen, 2.0)
    assert out[3] == 0.5 and out[0] == 1.0 and out[6] == 1.0 and out[7] == 0.5  # window = last 64


@pytest.mark.parametrize("tag", ["tiny_q4_k_m", "tiny_q6_k", "tiny_q8_0", "tiny-gqa_q4_k_m"])
def test_end_to_end_golden(oracle, pkg, golden, tag):
    g = golden["g3_e2e"]
    shape, mix = tag.split("_", 1)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=42)
    m = OracleModel(oracle, cfg, tensors)
    for plen in (4, 40):
        prompt = g[f"{tag}_p{plen}_prompt"]
        ids, logits = m.generate(prompt, 12, penalty=1.0, dump_logits=True)
        assert np.array_equal(ids, g[f"{tag}_p{plen}_pen1.0_ids"])
        assert np.array_equal(logits, g[f"{tag}_p{plen}_logits"])
        assert np.array_equal(m.generate(prompt, 12, penalty=1.3), g[f"{tag}_p{plen}_pen1.3_ids"])
    # ref_structure (whole-tensor dequant + f32 mat-vec + l-2 eviction) is bit-identical
    m.set_ref_structure(True)
    ids2, logits2 = m.generate(g[f"{tag}_p4_prompt"], 12, penalty=1.0, dump_logits=True)
    assert np.array_equal(ids2, g[f"{tag}_p4_pen1.0_ids"]) and np.array_equal(logits2, g[f"{tag}_p4_logits"])
    m.close()


def test_prefill_matches_stepwise(oracle, pkg):
    """Batched prefill reproduces token-by-token logits to <= 1e-6 (SURVEY §8c probe)."""
    cfg, tensors = pkg.synth.build_model("tiny", "q4_k_m", seed=7)
    m = OracleModel(oracle, cfg, tensors)
    toks = np.random.default_rng(0).integers(0, cfg["vocab"], 9)
    m.reset_kv()
    X = np.stack([m.lookup_embedding(t) for t in toks])
    _, lb = m.forward_batch(X, 0)
    m.reset_kv()
    for p, t in enumerate(toks):
        _, l = m.forward(m.lookup_embedding(t), p)
        assert np.abs(l - lb[p]).max() <= 2e-6
    m.close()


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference(oracle, pkg):
    ref = Ref()
    S = pkg.synth
    rng = np.random.default_rng(11)
    for t in (Q4_K, Q6_K, Q8_0, Q8_K):
        for blocks in (S.random_blocks(t, 40, rng), S.edge_blocks(t, rng),
                       rng.integers(0, 256, size=(40, S.BLOCK_BYTES[t]), dtype=np.uint8)):
            assert bits_equal_or_both_nan(oracle.dequantize(t, blocks), ref.dequantize(t, blocks))
        rows, cols = 40, 768
        w = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=0.03)
        x = rng.standard_normal(cols).astype(np.float32)
        assert np.array_equal(oracle.matvec_q(t, w, x, rows, cols), ref.matvec_q(t, w, x, rows, cols))
    # F32 and BF16 matrices: the reference's BF16 mat-vec widens each weight and is then its F32 mat-vec, which is what
    # the backend relies on when it widens BF16 tensors to F32 at upload (tl_model_set_tensor)
    rows, cols = 24, 512
    wf = (rng.standard_normal((rows, cols)) * 0.05).astype(np.float32)
    x = rng.standard_normal(cols).astype(np.float32)
    assert np.array_equal(oracle.matvec_q(F32, wf, x, rows, cols), ref.matvec_q(F32, wf, x, rows, cols))
    wb = (wf.view(np.uint32) >> 16).astype(np.uint16)
    wide = (wb.astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(oracle.matvec_q(F32, wide, x, rows, cols), ref.matvec_q(30, wb, x, rows, cols))
    x = (rng.standard_normal(4096) * 7).astype(np.float32)
    w = rng.standard_normal(4096).astype(np.float32)
    assert np.array_equal(oracle.rmsnorm_vector(x, w, 1e-5), ref.rmsnorm_vector(x, w, 1e-5))
    assert np.array_equal(oracle.softmax(x), ref.softmax(x))
    assert np.array_equal(oracle.silu(x), ref.silu(x))
