"""Synthetic random-init models in the reference's on-disk formats.

Everything here is host-side numpy: random quantized super-blocks in the exact GGUF byte layouts
(quantization.h:57-124), a model builder for the BASELINE.json shapes with the "Q4_K_M" tensor-type
mix, and a GGUF-v3 writer (gguf_parser.cpp:134-528 is the reader it targets).

Scales are chosen so that dequantized weights have std ~= gain/sqrt(fan_in): activations stay O(1)
through any depth, logits are finite and the greedy top-2 gap is far above the fp32 tolerance.
"""
import struct

import numpy as np

F32, Q8_0, Q4_K, Q6_K, Q8_K = 0, 8, 12, 14, 15
BF16 = 30  # accepted by tl_model_set_tensor (widened to F32 at upload); a uint16 array
BLOCK_BYTES = {F32: 4, Q8_0: 34, Q4_K: 144, Q6_K: 210, Q8_K: 290, BF16: 2}
BLOCK_ELEMS = {F32: 1, Q8_0: 32, Q4_K: 256, Q6_K: 256, Q8_K: 256, BF16: 1}
TYPE_NAMES = {F32: "F32", Q8_0: "Q8_0", Q4_K: "Q4_K", Q6_K: "Q6_K", Q8_K: "Q8_K"}


def tensor_bytes(typ, rows, cols):
    return rows * cols // BLOCK_ELEMS[typ] * BLOCK_BYTES[typ]


def _put_f16(blocks, off, vals):
    blocks[:, off:off + 2] = np.asarray(vals, np.float16).reshape(-1, 1).view(np.uint8)


def random_blocks(typ, nblocks, rng, w_std=0.02):
    """`nblocks` random blocks of `typ` as a uint8 array [nblocks, block_bytes]; payload bytes are
    uniform random, fp16 super-scales are drawn so the dequantized weights have std ~= w_std."""
    bb = BLOCK_BYTES[typ]
    b = rng.integers(0, 256, size=(nblocks, bb), dtype=np.uint8)
    u = rng.uniform(0.6, 1.4, size=nblocks)
    if typ == Q4_K:
        # w = d*sc*q - dmin*m, sc,m in [0,63], q in [0,15]: std(w) ~= d * 36.4 * 4.6 when centred
        d = w_std / 167.0 * u
        _put_f16(b, 0, d)
        _put_f16(b, 2, d * rng.uniform(6.5, 8.5, size=nblocks))  # ~zero-mean weights
    elif typ == Q6_K:
        # w = d*sc*q, sc int8 (kept in [-32,31]), q in [-32,31]: std ~= d * 18.5 * 18.5
        sc = rng.integers(-32, 32, size=(nblocks, 16), dtype=np.int8)
        b[:, 192:208] = sc.view(np.uint8)
        _put_f16(b, 208, w_std / 342.0 * u)
    elif typ in (Q8_0, Q8_K):
        # w = d*q, q int8 uniform: std ~= d * 73.9
        _put_f16(b, 0, w_std / 73.9 * u)
    else:
        raise ValueError(typ)
    return b


EDGE_F16 = [0x0000, 0x8000, 0x0001, 0x8001, 0x03ff, 0x83ff, 0x0400, 0x3c00, 0xbc00, 0x7bff, 0xfbff,
            0x1a3c, 0x9a3c, 0x2e66]
"""fp16 scale edge cases: +-0, smallest/largest subnormal, smallest normal, +-1, +-max finite, a few
ordinary values incl. negatives (Q8 scales take |d|, quantization.cpp:85-87)."""


def edge_blocks(typ, rng):
    """Blocks whose fp16 scale fields walk EDGE_F16 (both d and dmin for Q4_K), random payload."""
    n = len(EDGE_F16)
    b = rng.integers(0, 256, size=(n, BLOCK_BYTES[typ]), dtype=np.uint8)
    e = np.array(EDGE_F16, np.uint16)
    off = 208 if typ == Q6_K else 0
    b[:, off:off + 2] = e.reshape(-1, 1).view(np.uint8)
    if typ == Q4_K:
        b[:, 2:4] = np.roll(e, 3).reshape(-1, 1).view(np.uint8)
    return b


# ---- model shapes -----------------------------------------------------------------
SHAPES = {
    "tiny": dict(hidden=256, inter=512, n_heads=4, n_kv_heads=2, n_layers=3, vocab=64, max_pos=128,
                 eps=1e-5, rope_theta=10000.0),
    "tiny-gqa": dict(hidden=512, inter=1024, n_heads=8, n_kv_heads=2, n_layers=2, vocab=512, max_pos=256,
                     eps=1e-5, rope_theta=10000.0),
    "tiny-tp": dict(hidden=1024, inter=2048, n_heads=16, n_kv_heads=4, n_layers=2, vocab=512, max_pos=128,
                    eps=1e-5, rope_theta=10000.0),
    "tinyllama-1.1b": dict(hidden=2048, inter=5632, n_heads=32, n_kv_heads=4, n_layers=22, vocab=32000,
                           max_pos=2048, eps=1e-5, rope_theta=10000.0),
    "llama-3.2-3b": dict(hidden=3072, inter=8192, n_heads=24, n_kv_heads=8, n_layers=28, vocab=128256,
                         max_pos=2048, eps=1e-5, rope_theta=500000.0),
    "llama-3-8b": dict(hidden=4096, inter=14336, n_heads=32, n_kv_heads=8, n_layers=32, vocab=128256,
                       max_pos=8192, eps=1e-5, rope_theta=500000.0),
    "llama-3-70b": dict(hidden=8192, inter=28672, n_heads=64, n_kv_heads=8, n_layers=80, vocab=128256,
                        max_pos=8192, eps=1e-5, rope_theta=500000.0),
}


def tensor_plan(cfg, mix="q4_k_m"):
    """[(name, type, rows, cols)] in file order. mix: 'q4_k_m' (Q4_K except attn_v/ffn_down
    Q6_K on layers i<L/8 || i>=7L/8 || (i-L/8)%3==2, output Q6_K), 'q6_k', 'q8_0', 'q4_k'."""
    H, I, L, V = cfg["hidden"], cfg["inter"], cfg["n_layers"], cfg["vocab"]
    kvd = cfg["n_kv_heads"] * (H // cfg["n_heads"])
    base = {"q4_k_m": Q4_K, "q4_k": Q4_K, "q6_k": Q6_K, "q8_0": Q8_0}[mix]

    def more_bits(i):
        return mix == "q4_k_m" and (i < L // 8 or i >= 7 * L // 8 or (i - L // 8) % 3 == 2)

    plan = [("token_embd.weight", base, V, H)]
    for i in range(L):
        hi = Q6_K if more_bits(i) else base
        plan += [
            (f"blk.{i}.attn_norm.weight", F32, 1, H),
            (f"blk.{i}.attn_q.weight", base, H, H),
            (f"blk.{i}.attn_k.weight", base, kvd, H),
            (f"blk.{i}.attn_v.weight", hi, kvd, H),
            (f"blk.{i}.attn_output.weight", base, H, H),
            (f"blk.{i}.ffn_norm.weight", F32, 1, H),
            (f"blk.{i}.ffn_gate.weight", base, I, H),
            (f"blk.{i}.ffn_up.weight", base, I, H),
            (f"blk.{i}.ffn_down.weight", hi, H, I),
        ]
    plan += [("output_norm.weight", F32, 1, H),
             ("output.weight", Q6_K if mix == "q4_k_m" else base, V, H)]
    return plan


def weight_bytes_per_token(cfg, mix="q4_k_m"):
    """Algorithmic weight bytes one decoded token must read (SURVEY §8d B_tok without the KV terms):
    every projection + output.weight + ONE row of token_embd."""
    total = 0
    for name, typ, rows, cols in tensor_plan(cfg, mix):
        if typ == F32:
            continue
        total += tensor_bytes(typ, 1 if name == "token_embd.weight" else rows, cols)
    return total


def make_tensor(name, typ, rows, cols, seed):
    """Deterministic content of one tensor: (type, ndarray) — uint8 [nblocks, bb] or float32 [cols]."""
    rng = np.random.default_rng([seed, abs(hash_name(name))])
    if typ == F32:
        return (1.0 + 0.1 * rng.standard_normal(cols)).astype(np.float32)
    if name == "token_embd.weight":
        std = 1.0
    else:
        std = 1.0 / np.sqrt(cols)
    return random_blocks(typ, rows * cols // BLOCK_ELEMS[typ], rng, w_std=std)


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def build_model(shape, mix="q4_k_m", seed=42):
    """(cfg, {name: (type, array, rows, cols)}) fully materialised in host memory."""
    cfg = dict(SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    tensors = {}
    for name, typ, rows, cols in tensor_plan(cfg, mix):
        tensors[name] = (typ, make_tensor(name, typ, rows, cols, seed), rows, cols)
    return cfg, tensors


# ---- GGUF v3 writer ----------------------------------------------------------------------------
def _gs(s):
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def _kv_u32(k, v):
    return _gs(k) + struct.pack("<II", 4, v)


def _kv_f32(k, v):
    return _gs(k) + struct.pack("<If", 6, v)


def _kv_str(k, v):
    return _gs(k) + struct.pack("<I", 8) + _gs(v)


def _kv_i32(k, v):
    return _gs(k) + struct.pack("<Ii", 5, v)


def _gb(b):
    return struct.pack("<Q", len(b)) + bytes(b)


def tokenizer_kvs(tokens, model="llama", merges=None, scores=None, token_types=None, bos=None, eos=None, unk=None,
                  pad=None, chat_template=None, vocab_size=None):
    """The tokenizer.* metadata the reference reads (gguf_parser.cpp:240-300, model_config.cpp:199-253). `tokens`,
    `merges`, `chat_template` are bytes (token texts need not be UTF-8)."""
    kvs = [_kv_str("tokenizer.ggml.model", model),
           _gs("tokenizer.ggml.tokens") + struct.pack("<IIQ", 9, 8, len(tokens)) + b"".join(_gb(t) for t in tokens)]
    if merges is not None:
        kvs.append(_gs("tokenizer.ggml.merges") + struct.pack("<IIQ", 9, 8, len(merges)) + b"".join(_gb(m) for m in merges))
    if scores is not None:
        kvs.append(_gs("tokenizer.ggml.scores") + struct.pack("<IIQ", 9, 6, len(scores)) +
                   np.asarray(scores, np.float32).tobytes())
    if token_types is not None:
        kvs.append(_gs("tokenizer.ggml.token_type") + struct.pack("<IIQ", 9, 5, len(token_types)) +
                   np.asarray(token_types, np.int32).tobytes())
    for key, v in (("bos_token_id", bos), ("eos_token_id", eos), ("unk_token_id", unk), ("padding_token_id", pad)):
        if v is not None:
            kvs.append(_kv_i32("tokenizer.ggml." + key, v))
    if chat_template is not None:
        kvs.append(_gs("tokenizer.chat_template") + struct.pack("<I", 8) + _gb(chat_template))
    if vocab_size is not None:
        kvs.append(_kv_u32("tokenizer.ggml.vocab_size", vocab_size))
    return kvs


def write_metadata_gguf(path, kvs, architecture="llama"):
    """A GGUF v3 file with metadata only (no tensors): what the tokenizer tests open."""
    kvs = [_kv_str("general.architecture", architecture)] + list(kvs)
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQQ", 0x46554747, 3, 0, len(kvs)) + b"".join(kvs))
    return path


def write_gguf(path, cfg, tensors, alignment=32, extra_kvs=()):
    """Write `tensors` ({name: (type, array, rows, cols)}) with the metadata keys the reference reads
    (model_config.cpp:149-221). ggml dim order: dims[0] = cols (in_features), dims[1] = rows."""
    kvs = [
        _kv_str("general.architecture", "llama"),
        _kv_str("general.name", "synthetic"),
        _kv_u32("general.alignment", alignment),
        _kv_u32("llama.embedding_length", cfg["hidden"]),
        _kv_u32("llama.feed_forward_length", cfg["inter"]),
        _kv_u32("llama.attention.head_count", cfg["n_heads"]),
        _kv_u32("llama.attention.head_count_kv", cfg["n_kv_heads"]),
        _kv_u32("llama.block_count", cfg["n_layers"]),
        _kv_u32("llama.context_length", cfg["max_pos"]),
        _kv_u32("llama.vocab_size", cfg["vocab"]),
        _kv_f32("llama.attention.layer_norm_rms_epsilon", cfg["eps"]),
        _kv_f32("llama.rope.freq_base", cfg["rope_theta"]),
    ]
    if not extra_kvs:
        kvs += [_kv_u32("tokenizer.ggml.bos_token_id", 1), _kv_u32("tokenizer.ggml.eos_token_id", 2)]
    kvs += list(extra_kvs)
    infos, offset, order = [], 0, []
    for name, (typ, arr, rows, cols) in tensors.items():
        nbytes = arr.nbytes
        dims = [cols] if typ == F32 and rows == 1 else [cols, rows]
        info = _gs(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims)
        info += struct.pack("<IQ", typ, offset)
        infos.append(info)
        order.append((arr, offset))
        offset = (offset + nbytes + alignment - 1) // alignment * alignment
    head = struct.pack("<IIQQ", 0x46554747, 3, len(infos), len(kvs)) + b"".join(kvs) + b"".join(infos)
    pad = (-len(head)) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        start = f.tell()
        for arr, off in order:
            f.seek(start + off)
            f.write(np.ascontiguousarray(arr).tobytes())
        end = start + offset
        f.seek(0, 2)
        if f.tell() < end:
            f.write(b"\0" * (end - f.tell()))
    return path
