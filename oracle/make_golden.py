#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the build container only).

Primitive fixtures (G1/G2) are OUTPUTS OF THE REFERENCE ITSELF: oracle/_ref/libtlref.so is the
reference's quantization.cpp / utils.cpp compiled from /root/reference (oracle/Makefile `ref`).
The end-to-end fixture (G3) is produced by our oracle restatement (tl_oracle.c) — the reference's
TinyLlamaModel cannot be built here (model.h needs nlohmann/json.hpp) — and is a regression pin of the
composition, whose every primitive is pinned by G1/G2.

Fixtures are data only: seeded inputs + expected outputs, as .npz.

    python oracle/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT  # the script dir (oracle/) would shadow the `oracle` package
from _pkg import load_package  # noqa: E402
from oracle.oracle import Oracle, OracleModel, Ref, Q4_K, Q6_K, Q8_0, Q8_K  # noqa: E402

S = load_package().synth
OUT = os.path.join(ROOT, "tests", "golden")
NAMES = {Q4_K: "q4_k", Q6_K: "q6_k", Q8_0: "q8_0", Q8_K: "q8_k"}


def extended_gguf_bytes(cfg, tensors, alignment, extra_tensor=None):
    """A synthetic GGUF v3 file with every scalar metadata type, the tokenizer arrays the reference retains
    (gguf_parser.cpp:245-348) and a non-default alignment: bytes of the file."""
    import struct
    gs = S._gs
    V = cfg["vocab"]
    toks = [f"<tok{i}>" for i in range(V)]
    kvs = [
        S._kv_str("general.architecture", "llama"), S._kv_str("general.name", "golden g4"),
        S._kv_u32("general.alignment", alignment),
        S._kv_u32("llama.embedding_length", cfg["hidden"]), S._kv_u32("llama.feed_forward_length", cfg["inter"]),
        S._kv_u32("llama.attention.head_count", cfg["n_heads"]),
        gs("llama.attention.head_count_kv") + struct.pack("<IQ", 10, cfg["n_kv_heads"]),      # u64
        gs("llama.block_count") + struct.pack("<Ii", 5, cfg["n_layers"]),                       # i32
        gs("llama.context_length") + struct.pack("<IH", 2, cfg["max_pos"]),                     # u16
        S._kv_u32("llama.vocab_size", V),
        S._kv_f32("llama.attention.layer_norm_rms_epsilon", cfg["eps"]),
        gs("llama.rope.freq_base") + struct.pack("<Id", 12, cfg["rope_theta"]),                 # f64
        gs("tokenizer.ggml.bos_token_id") + struct.pack("<IB", 0, 1),                           # u8
        gs("tokenizer.ggml.eos_token_id") + struct.pack("<Ib", 1, 2),                           # i8
        gs("tokenizer.ggml.unknown_token_id") + struct.pack("<Ih", 3, -1),                      # i16
        gs("golden.some_i64") + struct.pack("<Iq", 11, -(1 << 40)),                             # i64
        gs("golden.some_bool") + struct.pack("<IB", 7, 1),                                      # bool
        S._kv_str("tokenizer.ggml.model", "llama"),
        gs("tokenizer.ggml.tokens") + struct.pack("<IIQ", 9, 8, V) + b"".join(gs(t) for t in toks),
        gs("tokenizer.ggml.scores") + struct.pack("<IIQ", 9, 6, V) + struct.pack(f"<{V}f", *[-float(i) for i in range(V)]),
        gs("tokenizer.ggml.token_type") + struct.pack("<IIQ", 9, 5, V) + struct.pack(f"<{V}i", *([1] * V)),
        gs("golden.skipped_array") + struct.pack("<IIQ", 9, 4, 3) + struct.pack("<3I", 7, 8, 9),  # u32 array: skipped
    ]
    items = list(tensors.items()) + ([extra_tensor] if extra_tensor else [])
    infos, offset, order = [], 0, []
    for name, (typ, arr, rows, cols) in items:
        dims = [cols] if typ == S.F32 and rows == 1 else [cols, rows]
        infos.append(gs(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) +
                     struct.pack("<IQ", typ, offset))
        order.append((np.ascontiguousarray(arr).tobytes(), offset))
        offset = (offset + len(order[-1][0]) + alignment - 1) // alignment * alignment
    head = struct.pack("<IIQQ", 0x46554747, 3, len(infos), len(kvs)) + b"".join(kvs) + b"".join(infos)
    head += b"\0" * ((-len(head)) % alignment)
    body = bytearray(offset)
    for raw, off in order:
        body[off:off + len(raw)] = raw
    return head + bytes(body)


def make_g4(ref):
    """G4: a synthetic GGUF file and what the reference's OWN parser (load_gguf_meta, gguf_parser.cpp:123; size rules
    quantization.cpp:646-717) makes of it — tensor names, types, dims, offsets, element and byte counts, a CRC of the bytes
    where the reference's consumers read them, alignment, metadata; plus the parser's verdict on a file that holds a
    Q8_K tensor (it throws: ggml_type_block_size returns 0 for Q8_K)."""
    import json
    import tempfile
    cfg = dict(hidden=256, inter=256, n_heads=4, n_kv_heads=2, n_layers=1, vocab=32, max_pos=4096, eps=1e-5, rope_theta=500000.0)
    cfg, tensors = S.build_model(cfg, "q4_k_m", seed=11)
    raw = extended_gguf_bytes(cfg, tensors, 64)
    rng = np.random.default_rng(4)
    q8k = ("blk.0.extra_q8k.weight", (Q8_K, S.random_blocks(Q8_K, 2, rng), 2, 256))
    raw_q8k = extended_gguf_bytes(cfg, dict(list(tensors.items())[:1]), 32, extra_tensor=q8k)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, blob in (("good", raw), ("q8k", raw_q8k)):
            path = os.path.join(d, tag + ".gguf")
            with open(path, "wb") as f:
                f.write(blob)
            parsed = ref.gguf_parse(path, use_mmap=True)
            assert parsed == ref.gguf_parse(path, use_mmap=False), "mmap and bulk-read parses differ"
            out[tag] = np.frombuffer(blob, np.uint8)
            out[tag + "_ref_parse_json"] = np.frombuffer(json.dumps(parsed, sort_keys=True).encode(), np.uint8)
    assert "error" not in json.loads(bytes(out["good_ref_parse_json"]))
    assert "error" in json.loads(bytes(out["q8k_ref_parse_json"]))
    np.savez_compressed(os.path.join(OUT, "g4_gguf.npz"), **out)
    print("g4_gguf.npz", os.path.getsize(os.path.join(OUT, "g4_gguf.npz")), json.loads(bytes(out["q8k_ref_parse_json"])))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref, orc = Ref(), Oracle()
    if sys.argv[1:] == ["g4"]:  # only the GGUF fixture (the others are unchanged since round 1)
        return make_g4(ref)
    rng = np.random.default_rng(20261110)

    # G0: fp16 -> fp32 for ALL 65536 halves, both flag values (bits), + fp32->fp16 samples
    h = np.arange(65536, dtype=np.uint32)
    plain = np.array([ref.fp16_to_fp32_bits(int(v), False) for v in h], np.uint32)
    scale = np.array([ref.fp16_to_fp32_bits(int(v), True) for v in h], np.uint32)
    fs = np.concatenate([
        (rng.standard_normal(4096) * 10.0 ** rng.uniform(-9, 6, 4096)).astype(np.float32),
        np.array([0, -0.0, np.inf, -np.inf, 65504, 65520, 1e-8, 6e-8, 6.1e-5, -3.3e-5], np.float32)])
    f2h = np.array([ref.fp32_to_fp16(float(v)) for v in fs], np.uint16)
    np.savez_compressed(os.path.join(OUT, "g0_fp16.npz"), plain=plain, scale=scale, f32_in=fs, f16_out=f2h)

    # G1: per format, random + scale-edge-case + all-bytes-random blocks -> dequantized f32 (bit patterns)
    g1 = {}
    for t, nm in NAMES.items():
        blocks = np.concatenate([
            S.random_blocks(t, 24, rng, w_std=0.05),
            S.edge_blocks(t, rng),
            rng.integers(0, 256, size=(10, S.BLOCK_BYTES[t]), dtype=np.uint8)])
        g1[nm + "_blocks"] = blocks
        g1[nm + "_f32bits"] = ref.dequantize(t, blocks).view(np.uint32)
    np.savez_compressed(os.path.join(OUT, "g1_dequant.npz"), **g1)

    # G2: per-op vectors
    g2 = {}
    for t, nm in NAMES.items():
        for rows, cols in ((256, 256), (512, 256), (96, 512)):
            w = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
            x = rng.standard_normal(cols).astype(np.float32)
            key = f"gemv_{nm}_{rows}x{cols}"
            g2[key + "_w"], g2[key + "_x"] = w, x
            g2[key + "_y"] = ref.matvec_q(t, w, x, rows, cols)
        if t != Q8_K:
            rows, cols, T = 64, 256, 5
            w = S.random_blocks(t, rows * cols // S.BLOCK_ELEMS[t], rng, w_std=1.0 / np.sqrt(cols))
            X = rng.standard_normal((T, cols)).astype(np.float32)
            key = f"gemm_{nm}"
            g2[key + "_w"], g2[key + "_x"] = w, X
            g2[key + "_y"] = ref.matmul_q_batch(t, w, X, T, rows, cols)
    x = (rng.standard_normal(2048) * 2.5).astype(np.float32)
    w = (1.0 + 0.1 * rng.standard_normal(2048)).astype(np.float32)
    g2["rms_x"], g2["rms_w"] = x, w
    for i, eps in enumerate((1e-5, 1e-6, 1e-3)):
        g2[f"rms_vec_{i}"] = ref.rmsnorm_vector(x, w, eps)
        g2[f"rms_batch_{i}"] = ref.rmsnorm_batch(x.reshape(8, 256), w[:256], eps)
    g2["rms_eps"] = np.array([1e-5, 1e-6, 1e-3], np.float32)
    for hd, nh, theta in ((64, 4, 10000.0), (128, 3, 500000.0)):
        table = orc.rope_table(96, hd, theta)  # table formula itself is restated (gpu_initialization.cpp:48-58)
        q = rng.standard_normal(nh * hd).astype(np.float32)
        g2[f"rope{hd}_table"], g2[f"rope{hd}_x"] = table, q
        for pos in (0, 1, 37, 95, 96):
            g2[f"rope{hd}_adj_{pos}"] = ref.rope_apply(q, nh, hd, pos, table, True)
            g2[f"rope{hd}_half_{pos}"] = ref.rope_apply(q, nh, hd, pos, table, False)
    qb = rng.standard_normal((6, 4 * 64)).astype(np.float32)
    kb = rng.standard_normal((6, 2 * 64)).astype(np.float32)
    g2["ropeb_q"], g2["ropeb_k"] = qb, kb
    g2["ropeb_q_out"], g2["ropeb_k_out"] = ref.rope_batch(qb, kb, 6, 4, 2, 64, 30, g2["rope64_table"], True)
    s = (rng.standard_normal(777) * 5).astype(np.float32)
    g2["softmax_x"], g2["softmax_y"] = s, ref.softmax(s)
    g2["silu_x"], g2["silu_y"] = s, ref.silu(s)
    xq = (rng.standard_normal(1024) * 3).astype(np.float32)
    g2["q8kq_x"], g2["q8kq_blocks"] = xq, ref.quantize_q8_k(xq)
    np.savez_compressed(os.path.join(OUT, "g2_ops.npz"), **g2)

    # G3: end-to-end on tiny synthetic models (oracle restatement; see module docstring)
    g3 = {}
    for shape, mix in (("tiny", "q4_k_m"), ("tiny", "q6_k"), ("tiny", "q8_0"), ("tiny-gqa", "q4_k_m")):
        cfg, tensors = S.build_model(shape, mix, seed=42)
        m = OracleModel(orc, cfg, tensors)
        tag = f"{shape}_{mix}"
        for plen in (4, 40):
            prompt = np.random.default_rng(plen).integers(0, cfg["vocab"], plen).astype(np.int32)
            for pen in (1.0, 1.3):
                ids, logits = m.generate(prompt, 12, penalty=pen, dump_logits=True)
                g3[f"{tag}_p{plen}_pen{pen}_ids"] = ids
                if pen == 1.0:
                    g3[f"{tag}_p{plen}_logits"] = logits
            g3[f"{tag}_p{plen}_prompt"] = prompt
        m.close()
    np.savez_compressed(os.path.join(OUT, "g3_e2e.npz"), **g3)
    make_g4(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
