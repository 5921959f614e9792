"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE ONLY).

`Oracle`  -> oracle/libtloracle.so  (our C restatement, tl_oracle.c)
`Ref`     -> oracle/_ref/libtlref.so (the reference's own Tier-1 sources behind ref_shim.cpp), present
             only where oracle/Makefile's `ref` target has been run (/root/reference needed to build).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

F32, Q8_0, Q4_K, Q6_K, Q8_K = 0, 8, 12, 14, 15
BLOCK_BYTES = {F32: 4, Q8_0: 34, Q4_K: 144, Q6_K: 210, Q8_K: 290}
BLOCK_ELEMS = {F32: 1, Q8_0: 32, Q4_K: 256, Q6_K: 256, Q8_K: 256}

# tensor kinds (tl_oracle.h)
T_ATTN_Q, T_ATTN_K, T_ATTN_V, T_ATTN_O, T_FFN_GATE, T_FFN_UP, T_FFN_DOWN, T_ATTN_NORM, T_FFN_NORM, \
    T_TOKEN_EMBD, T_OUTPUT_NORM, T_OUTPUT = range(12)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def _v(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def build(ref=None):
    """Compile the oracle (and, when /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref is None:
        ref = os.path.isdir("/root/reference")
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "libtloracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        L.tlo_fp16_to_fp32.restype = C.c_float
        L.tlo_fp16_to_fp32.argtypes = [C.c_uint16, C.c_int]
        L.tlo_fp32_to_fp16.restype = C.c_uint16
        L.tlo_fp32_to_fp16.argtypes = [C.c_float]
        L.tlo_dequantize.argtypes = [C.c_int, C.c_void_p, C.c_size_t, _fp]
        L.tlo_quantize_q8_k.argtypes = [_fp, C.c_size_t, C.c_void_p]
        L.tlo_matvec_q.argtypes = [C.c_int, C.c_void_p, _fp, _fp, C.c_int, C.c_int]
        L.tlo_matvec_f32.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int]
        L.tlo_matmul_q_batch.argtypes = [C.c_int, C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int]
        L.tlo_rmsnorm_vector.argtypes = [_fp, _fp, _fp, C.c_int, C.c_float]
        L.tlo_rmsnorm_batch.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_float]
        L.tlo_rope_table.argtypes = [_fp, C.c_int, C.c_int, C.c_float]
        L.tlo_rope_apply.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int]
        L.tlo_softmax.argtypes = [_fp, _fp, C.c_int]
        L.tlo_silu.argtypes = [_fp, _fp, C.c_int]
        L.tlo_swiglu.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int]
        L.tlo_attention_decode.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tlo_attention_batch.argtypes = [_fp, _fp, _fp, _fp] + [C.c_int] * 5
        L.tlo_argmax.restype = C.c_int
        L.tlo_argmax.argtypes = [_fp, C.c_int]
        L.tlo_apply_repetition_penalty.argtypes = [_fp, C.c_int, _ip, C.c_int, C.c_float, C.c_int]
        L.tlo_model_create.restype = C.c_void_p
        L.tlo_model_create.argtypes = [C.c_int] * 7 + [C.c_float, C.c_float]
        L.tlo_model_free.argtypes = [C.c_void_p]
        L.tlo_model_set_tensor.restype = C.c_int
        L.tlo_model_set_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.tlo_model_set_ref_structure.argtypes = [C.c_void_p, C.c_int]
        L.tlo_model_reset_kv.argtypes = [C.c_void_p]
        L.tlo_lookup_embedding.argtypes = [C.c_void_p, C.c_int, _fp]
        L.tlo_forward.argtypes = [C.c_void_p, _fp, C.c_int, _fp, C.c_int]
        L.tlo_forward_batch.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _fp]
        L.tlo_generate.restype = C.c_int
        L.tlo_generate.argtypes = [C.c_void_p, _ip, C.c_int, C.c_int, C.c_float, C.c_int, _ip, _fp]
        L.tlo_model_kcache.restype = _fp
        L.tlo_model_kcache.argtypes = [C.c_void_p, C.c_int]
        L.tlo_model_vcache.restype = _fp
        L.tlo_model_vcache.argtypes = [C.c_void_p, C.c_int]

    # ---- primitives -------------------------------------------------------------------
    def fp16_to_fp32(self, h, is_scale=False):
        return float(self.L.tlo_fp16_to_fp32(int(h), int(is_scale)))

    def fp16_to_fp32_bits(self, h, is_scale=False):
        return int(np.float32(self.L.tlo_fp16_to_fp32(int(h), int(is_scale))).view(np.uint32))

    def fp32_to_fp16(self, f):
        return int(self.L.tlo_fp32_to_fp16(float(f)))

    def dequantize(self, typ, blocks):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        nb = blocks.size // BLOCK_BYTES[typ]
        out = np.empty(nb * BLOCK_ELEMS[typ], np.float32)
        self.L.tlo_dequantize(typ, _v(blocks), nb, _f(out))
        return out

    def quantize_q8_k(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.size // 256) * 290, np.uint8)
        self.L.tlo_quantize_q8_k(_f(x), x.size, _v(out))
        return out

    def matvec_q(self, typ, blocks, x, rows, cols):
        blocks = np.ascontiguousarray(blocks)
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(rows, np.float32)
        self.L.tlo_matvec_q(typ, _v(blocks), _f(x), _f(y), rows, cols)
        return y

    def matmul_q_batch(self, typ, blocks, x, T, rows, cols):
        blocks = np.ascontiguousarray(blocks)
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty((T, rows), np.float32)
        self.L.tlo_matmul_q_batch(typ, _v(blocks), _f(x), _f(y), T, rows, cols)
        return y

    def rmsnorm_vector(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        out = np.empty_like(x)
        self.L.tlo_rmsnorm_vector(_f(x), _f(w), _f(out), x.size, eps)
        return out

    def rmsnorm_batch(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        out = np.empty_like(x)
        self.L.tlo_rmsnorm_batch(_f(x), _f(w), _f(out), x.shape[0], x.shape[1], eps)
        return out

    def rope_table(self, max_pos, head_dim, theta):
        t = np.empty((max_pos, head_dim // 2, 2), np.float32)
        self.L.tlo_rope_table(_f(t), max_pos, head_dim, theta)
        return t

    def rope_apply(self, x, num_heads, head_dim, pos, table, adjacent=True):
        x = np.array(x, np.float32, copy=True)
        table = np.ascontiguousarray(table, np.float32)
        self.L.tlo_rope_apply(_f(x), num_heads, head_dim, pos, _f(table), table.shape[0], int(adjacent))
        return x

    def softmax(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.tlo_softmax(_f(x), _f(out), x.size)
        return out

    def silu(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.tlo_silu(_f(x), _f(out), x.size)
        return out

    def swiglu(self, gate, up, batch_form=False):
        gate = np.ascontiguousarray(gate, np.float32)
        up = np.ascontiguousarray(up, np.float32)
        out = np.empty_like(gate)
        self.L.tlo_swiglu(_f(gate), _f(up), _f(out), gate.size, int(batch_form))
        return out

    def attention_decode(self, q, kc, vc, n_heads, n_kv_heads, head_dim, pos):
        q = np.ascontiguousarray(q, np.float32)
        kc = np.ascontiguousarray(kc, np.float32)
        vc = np.ascontiguousarray(vc, np.float32)
        out = np.empty(n_heads * head_dim, np.float32)
        self.L.tlo_attention_decode(_f(q), _f(kc), _f(vc), _f(out), n_heads, n_kv_heads, head_dim, pos)
        return out

    def attention_batch(self, q, kc, vc, T, start_pos, n_heads, n_kv_heads, head_dim):
        q = np.ascontiguousarray(q, np.float32)
        kc = np.ascontiguousarray(kc, np.float32)
        vc = np.ascontiguousarray(vc, np.float32)
        out = np.empty((T, n_heads * head_dim), np.float32)
        self.L.tlo_attention_batch(_f(q), _f(kc), _f(vc), _f(out), T, start_pos, n_heads, n_kv_heads, head_dim)
        return out

    def argmax(self, v):
        v = np.ascontiguousarray(v, np.float32)
        return int(self.L.tlo_argmax(_f(v), v.size))

    def apply_repetition_penalty(self, logits, generated, penalty, window=64):
        logits = np.array(logits, np.float32, copy=True)
        g = np.ascontiguousarray(generated, np.int32)
        self.L.tlo_apply_repetition_penalty(_f(logits), logits.size, g.ctypes.data_as(_ip), g.size, penalty, window)
        return logits


    def sample_candidates(self, logits, temperature, top_k, top_p):
        """(ids, probs) of sample_top_k_top_p_temperature (api.cpp:126-216) before the draw."""
        logits = np.ascontiguousarray(logits, np.float32)
        ids = np.empty(logits.size, np.int32)
        probs = np.empty(logits.size, np.float32)
        self.L.tlo_sample_candidates.argtypes = [_fp, C.c_int, C.c_float, C.c_int, C.c_float, _ip, _fp]
        n = self.L.tlo_sample_candidates(_f(logits), logits.size, temperature, top_k, top_p, ids.ctypes.data_as(_ip), _f(probs))
        return ids[:n].copy(), probs[:n].copy()

    def discrete_draw(self, probs, u):
        probs = np.ascontiguousarray(probs, np.float32)
        self.L.tlo_discrete_draw.argtypes = [_fp, C.c_int, C.c_double]
        return int(self.L.tlo_discrete_draw(_f(probs), probs.size, float(u)))


class OracleModel:
    """A model held by the oracle. `tensors`: dict name -> (type, uint8/float32 ndarray, rows, cols),
    names as in GGUF. Arrays are kept alive here (the C side borrows them)."""

    KINDS = {"attn_q": T_ATTN_Q, "attn_k": T_ATTN_K, "attn_v": T_ATTN_V, "attn_output": T_ATTN_O,
             "ffn_gate": T_FFN_GATE, "ffn_up": T_FFN_UP, "ffn_down": T_FFN_DOWN,
             "attn_norm": T_ATTN_NORM, "ffn_norm": T_FFN_NORM}

    def __init__(self, oracle, cfg, tensors):
        self.o = oracle
        self.cfg = cfg
        L = oracle.L
        self.h = L.tlo_model_create(cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv_heads"],
                                    cfg["n_layers"], cfg["vocab"], cfg["max_pos"], cfg["eps"], cfg["rope_theta"])
        self._keep = []
        for name, (typ, arr, rows, cols) in tensors.items():
            arr = np.ascontiguousarray(arr)
            self._keep.append(arr)
            if name == "token_embd.weight":
                kind, layer = T_TOKEN_EMBD, -1
            elif name == "output_norm.weight":
                kind, layer = T_OUTPUT_NORM, -1
            elif name == "output.weight":
                kind, layer = T_OUTPUT, -1
            else:
                _, li, kn, _ = name.split(".")
                kind, layer = self.KINDS[kn], int(li)
            rc = L.tlo_model_set_tensor(self.h, kind, layer, typ, _v(arr), rows, cols)
            assert rc == 0, name

    def close(self):
        if self.h:
            self.o.L.tlo_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ref_structure(self, on):
        self.o.L.tlo_model_set_ref_structure(self.h, int(on))

    def reset_kv(self):
        self.o.L.tlo_model_reset_kv(self.h)

    def lookup_embedding(self, token):
        out = np.empty(self.cfg["hidden"], np.float32)
        self.o.L.tlo_lookup_embedding(self.h, int(token), _f(out))
        return out

    def forward(self, x, pos, n_layers_limit=-1, want_logits=True):
        x = np.array(x, np.float32, copy=True)
        logits = np.empty(self.cfg["vocab"], np.float32) if want_logits else None
        self.o.L.tlo_forward(self.h, _f(x), pos, _f(logits) if want_logits else None, n_layers_limit)
        return x, logits

    def forward_batch(self, x, start_pos=0, want_logits=True):
        x = np.array(x, np.float32, copy=True)
        T = x.shape[0]
        logits = np.empty((T, self.cfg["vocab"]), np.float32) if want_logits else None
        self.o.L.tlo_forward_batch(self.h, _f(x), T, start_pos, _f(logits) if want_logits else None)
        return x, logits

    def generate(self, prompt, max_new, penalty=1.0, eos_id=-1, dump_logits=False):
        p = np.ascontiguousarray(prompt, np.int32)
        out = np.zeros(max_new + 1, np.int32)
        dump = np.zeros((max_new + 1, self.cfg["vocab"]), np.float32) if dump_logits else None
        n = self.o.L.tlo_generate(self.h, p.ctypes.data_as(_ip), p.size, max_new, penalty, eos_id,
                                  out.ctypes.data_as(_ip), _f(dump) if dump_logits else None)
        return (out[:n].copy(), dump[:n].copy()) if dump_logits else out[:n].copy()

    def kcache(self, layer, n_pos):
        kvd = self.cfg["n_kv_heads"] * (self.cfg["hidden"] // self.cfg["n_heads"])
        p = self.o.L.tlo_model_kcache(self.h, layer)
        return np.ctypeslib.as_array(p, shape=(n_pos, kvd)).copy()

    def set_kv(self, layer, k, v):
        """Overwrite the first len(k) positions of a layer's K/V cache (long-context tests seed both sides this way
        instead of paying thousands of CPU forward passes)."""
        kvd = self.cfg["n_kv_heads"] * (self.cfg["hidden"] // self.cfg["n_heads"])
        for p, a in ((self.o.L.tlo_model_kcache(self.h, layer), k), (self.o.L.tlo_model_vcache(self.h, layer), v)):
            np.ctypeslib.as_array(p, shape=(a.shape[0], kvd))[...] = a

    def vcache(self, layer, n_pos):
        kvd = self.cfg["n_kv_heads"] * (self.cfg["hidden"] // self.cfg["n_heads"])
        p = self.o.L.tlo_model_vcache(self.h, layer)
        return np.ctypeslib.as_array(p, shape=(n_pos, kvd)).copy()


class Ref:
    """The reference's own code (Tier-1 subset). Raises FileNotFoundError where it was never built."""

    PATH = os.path.join(HERE, "_ref", "libtlref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        if not self.available():
            raise FileNotFoundError(self.PATH)
        L = self.L = C.CDLL(self.PATH)
        L.ref_init()
        L.ref_fp16_to_fp32.restype = C.c_float
        L.ref_fp16_to_fp32.argtypes = [C.c_uint16, C.c_int]
        L.ref_fp32_to_fp16.restype = C.c_uint16
        L.ref_fp32_to_fp16.argtypes = [C.c_float]
        L.ref_dequantize.argtypes = [C.c_int, C.c_void_p, C.c_size_t, _fp]
        L.ref_quantize_q8_k.argtypes = [_fp, C.c_size_t, C.c_void_p]
        L.ref_matvec_q.argtypes = [C.c_int, C.c_void_p, _fp, _fp, C.c_int, C.c_int]
        L.ref_matmul_batch.argtypes = [C.c_int, C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int]
        L.ref_rmsnorm_vector.argtypes = [_fp, _fp, _fp, C.c_int, C.c_float]
        L.ref_rmsnorm_batch.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_float]
        L.ref_rope_apply.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int]
        L.ref_rope_batch.argtypes = [_fp, _fp] + [C.c_int] * 5 + [_fp, C.c_int, C.c_int]
        L.ref_softmax.argtypes = [_fp, _fp, C.c_int]
        L.ref_silu.argtypes = [_fp, _fp, C.c_int]
        if hasattr(L, "ref_gguf_open"):  # (a libtlref.so built before round 2 lacks the parser / timing doorways)
            L.ref_last_error.restype = C.c_char_p
            L.ref_gguf_open.restype = C.c_long
            L.ref_gguf_open.argtypes = [C.c_char_p, C.c_int]
            L.ref_gguf_alignment.restype = C.c_ulonglong
            L.ref_gguf_meta_count.restype = C.c_long
            L.ref_gguf_tensor.argtypes = [C.c_long, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                          C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong),
                                          C.POINTER(C.c_ulonglong)]
            L.ref_gguf_tensor_bytes.restype = C.c_long
            L.ref_gguf_tensor_bytes.argtypes = [C.c_long, C.c_void_p, C.c_size_t]
            L.ref_gguf_meta.argtypes = [C.c_long, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong),
                                        C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.c_char_p, C.c_size_t]
            L.ref_time_dequant_matvec.argtypes = [C.c_int, C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def gguf_parse(self, path, use_mmap=True):
        """The file as the reference's load_gguf_meta (gguf_parser.cpp:123) sees it: {"alignment", "tensors": [dict],
        "metadata": {key: value}} or {"error": text} when the parser throws."""
        import zlib
        L = self.L
        n = L.ref_gguf_open(os.fsencode(path), int(use_mmap))
        if n < 0:
            return {"error": L.ref_last_error().decode(errors="replace")}
        out = {"alignment": int(L.ref_gguf_alignment()), "tensors": [], "metadata": {}}
        name = C.create_string_buffer(512)
        for i in range(n):
            typ, nd = C.c_uint(), C.c_uint()
            dims = (C.c_ulonglong * 4)()
            off, ne, nb = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
            assert L.ref_gguf_tensor(i, name, 512, C.byref(typ), C.byref(nd), dims, C.byref(off), C.byref(ne), C.byref(nb)) == 0
            raw = np.empty(nb.value, np.uint8)
            got = L.ref_gguf_tensor_bytes(i, _v(raw), raw.size)
            out["tensors"].append({"name": name.value.decode(), "type": typ.value, "dims": list(dims)[: nd.value],
                                   "offset": off.value, "num_elements": ne.value, "size_in_bytes": nb.value,
                                   "crc32": zlib.crc32(raw.tobytes()) if got == nb.value else None})
        key, sval = C.create_string_buffer(512), C.create_string_buffer(4096)
        for i in range(L.ref_gguf_meta_count()):
            kind, u, sg, f = C.c_int(), C.c_ulonglong(), C.c_longlong(), C.c_double()
            assert L.ref_gguf_meta(i, key, 512, C.byref(kind), C.byref(u), C.byref(sg), C.byref(f), sval, 4096) == 0
            v = {0: u.value, 1: sg.value, 2: f.value, 3: bool(u.value), 4: sval.value.decode(errors="replace"),
                 5: ["array", u.value]}[kind.value]
            out["metadata"][key.value.decode()] = v
        L.ref_gguf_close()
        return out

    def can_time(self):
        return hasattr(self.L, "ref_time_dequant_matvec")

    def time_dequant_matvec(self, typ, blocks, x, rows, cols, reps=3):
        """(dequantize seconds, mat-vec seconds, y): the reference's dequantize_vector_*_to_f32 +
        matvec_f32_f32_vector_cpu on one tensor, best of `reps`, timed inside the library (no marshalling copies)."""
        blocks = np.ascontiguousarray(blocks)
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(rows, np.float32)
        d, m = C.c_double(), C.c_double()
        assert self.L.ref_time_dequant_matvec(typ, _v(blocks), _f(x), _f(y), rows, cols, reps, C.byref(d), C.byref(m)) == 0
        return d.value, m.value, y

    def fp16_to_fp32_bits(self, h, is_scale=False):
        return int(np.float32(self.L.ref_fp16_to_fp32(int(h), int(is_scale))).view(np.uint32))

    def fp32_to_fp16(self, f):
        return int(self.L.ref_fp32_to_fp16(float(f)))

    def dequantize(self, typ, blocks):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        nb = blocks.size // BLOCK_BYTES[typ]
        out = np.empty(nb * BLOCK_ELEMS[typ], np.float32)
        assert self.L.ref_dequantize(typ, _v(blocks), nb, _f(out)) == 0
        return out

    def quantize_q8_k(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.size // 256) * 290, np.uint8)
        assert self.L.ref_quantize_q8_k(_f(x), x.size, _v(out)) == 290
        return out

    def matvec_q(self, typ, blocks, x, rows, cols):
        blocks = np.ascontiguousarray(blocks)
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(rows, np.float32)
        assert self.L.ref_matvec_q(typ, _v(blocks), _f(x), _f(y), rows, cols) == 0
        return y

    def matmul_q_batch(self, typ, blocks, x, T, rows, cols):
        blocks = np.ascontiguousarray(blocks)
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty((T, rows), np.float32)
        assert self.L.ref_matmul_batch(typ, _v(blocks), _f(x), _f(y), T, rows, cols) == 0
        return y

    def rmsnorm_vector(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        out = np.empty_like(x)
        self.L.ref_rmsnorm_vector(_f(x), _f(w), _f(out), x.size, eps)
        return out

    def rmsnorm_batch(self, x, w, eps):
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        out = np.empty_like(x)
        self.L.ref_rmsnorm_batch(_f(x), _f(w), _f(out), x.shape[0], x.shape[1], eps)
        return out

    def rope_apply(self, x, num_heads, head_dim, pos, table, adjacent=True):
        x = np.array(x, np.float32, copy=True)
        table = np.ascontiguousarray(table, np.float32)
        self.L.ref_rope_apply(_f(x), num_heads, head_dim, pos, _f(table), table.shape[0], int(adjacent))
        return x

    def rope_batch(self, q, k, T, nq, nkv, head_dim, start_pos, table, adjacent=True):
        q = np.array(q, np.float32, copy=True)
        k = np.array(k, np.float32, copy=True)
        table = np.ascontiguousarray(table, np.float32)
        self.L.ref_rope_batch(_f(q), _f(k), T, nq, nkv, head_dim, start_pos, _f(table), table.shape[0], int(adjacent))
        return q, k

    def softmax(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.ref_softmax(_f(x), _f(out), x.size)
        return out

    def silu(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.L.ref_silu(_f(x), _f(out), x.size)
        return out
