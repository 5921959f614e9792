"""ctypes view of the C++ host layer (libtlhost.so, include/tlhost.h): GGUF reader, tokenizer, Session, host-side sampler.

The host layer is C++ (tinyllama.cpp_amd/host); this module only exists so that the tests can drive it. It has no
fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "host", "libtlhost.so")


class TlHostError(RuntimeError):
    pass


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "tlh_gguf_open": [C.c_char_p, _vp],
    "tlh_gguf_close": [_vp],
    "tlh_gguf_counts": [_vp, _vp, _vp, _vp],
    "tlh_gguf_tensor": [_vp, C.c_uint64, C.c_char_p, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp],
    "tlh_gguf_meta_int": [_vp, C.c_char_p, _vp, _vp],
    "tlh_gguf_meta_float": [_vp, C.c_char_p, _vp, _vp],
    "tlh_gguf_meta_string": [_vp, C.c_char_p, C.c_char_p, C.c_size_t, _vp],
    "tlh_gguf_model_config": [_vp, _vp, _vp],
    "tlh_session_open": [C.c_char_p, _i, _i, _vp],
    "tlh_session_close": [_vp],
    "tlh_session_set_repetition_penalty": [_vp, _f],
    "tlh_session_seed": [_vp, C.c_uint32],
    "tlh_session_generate": [_vp, _vp, _i, _i, _f, _i, _f, _vp, _i, _vp],
    "tlh_session_set_max_batch": [_vp, _i],
    "tlh_session_generate_batch": [_vp, _vp, _vp, _i, _i, _f, _i, _f, _vp, _vp],
    "tlh_session_load_stats": [_vp, _vp, _vp, _vp],
    "tlh_tokenizer_from_gguf": [_vp, _i, _vp],
    "tlh_tokenizer_close": [_vp],
    "tlh_tokenizer_info": [_vp, _vp],
    "tlh_tokenizer_encode": [_vp, C.c_char_p, C.c_size_t, _i, _i, _vp, _i, _vp],
    "tlh_tokenizer_decode": [_vp, _vp, _i, _i, _vp, C.c_size_t, _vp],
    "tlh_tokenizer_apply_chat_template": [_vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, _vp, C.c_size_t, _vp],
    "tlh_tokenizer_format_prompt": [_vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, _i, _vp, C.c_size_t, _vp],
    "tlh_session_generate_text": [_vp, C.c_char_p, C.c_size_t, _i, _f, _i, _f, C.c_char_p, C.c_size_t, _i, _vp,
                                  C.c_size_t, _vp],
    "tlh_session_generate_batch_text": [_vp, C.c_char_p, _vp, _i, _i, _f, _i, _f, C.c_char_p, C.c_size_t, _i, _vp, C.c_size_t,
                                        _vp, _vp],
    "tlh_apply_repetition_penalty": [_vp, _i, _vp, _i, _f],
    "tlh_sampling_distribution": [_vp, _i, _f, _i, _f, _vp, _vp, _i, _vp],
    "tlh_sample": [_vp, _i, _f, _i, _f, C.c_uint32, _vp],
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TlHostError(f"{LIB_PATH} is missing: build it (make -C tinyllama.cpp_amd/host)")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.tlh_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise TlHostError(f"{what}: {lib().tlh_last_error().decode(errors='replace')}")


class GGUF:
    def __init__(self, path):
        h = C.c_void_p()
        check(lib().tlh_gguf_open(os.fsencode(path), C.byref(h)), "tlh_gguf_open")
        self.h = h.value

    def close(self):
        if self.h:
            lib().tlh_gguf_close(self.h)
            self.h = None

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().tlh_gguf_counts(self.h, C.byref(a), C.byref(b), C.byref(c)), "tlh_gguf_counts")
        return a.value, b.value, c.value

    def tensor(self, i):
        """(name, type, dims, offset, bytes as a uint8 array copied out of the mapping)"""
        name = C.create_string_buffer(256)
        typ, nd = C.c_uint32(), C.c_uint32()
        dims = (C.c_uint64 * 4)()
        off, nb = C.c_uint64(), C.c_uint64()
        data = C.c_void_p()
        check(lib().tlh_gguf_tensor(self.h, i, name, 256, C.byref(typ), dims, C.byref(nd), C.byref(off), C.byref(nb),
                                    C.byref(data)), "tlh_gguf_tensor")
        raw = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint8)), shape=(nb.value,)).copy()
        return name.value.decode(), typ.value, list(dims)[: nd.value], off.value, raw

    def meta_int(self, key):
        v, f = C.c_int64(), C.c_int()
        check(lib().tlh_gguf_meta_int(self.h, key.encode(), C.byref(v), C.byref(f)), "tlh_gguf_meta_int")
        return v.value if f.value else None

    def meta_float(self, key):
        v, f = C.c_double(), C.c_int()
        check(lib().tlh_gguf_meta_float(self.h, key.encode(), C.byref(v), C.byref(f)), "tlh_gguf_meta_float")
        return v.value if f.value else None

    def meta_string(self, key):
        buf, f = C.create_string_buffer(512), C.c_int()
        check(lib().tlh_gguf_meta_string(self.h, key.encode(), buf, 512, C.byref(f)), "tlh_gguf_meta_string")
        return buf.value.decode() if f.value else None

    def model_config(self):
        out, ft = (C.c_int * 9)(), (C.c_float * 2)()
        check(lib().tlh_gguf_model_config(self.h, out, ft), "tlh_gguf_model_config")
        keys = ("hidden", "inter", "n_heads", "n_kv_heads", "n_layers", "vocab", "max_pos", "bos", "eos")
        d = dict(zip(keys, list(out)))
        d["eps"], d["rope_theta"] = ft[0], ft[1]
        return d


def _bytes_out(call, what, cap=4096):
    """Run call(buf, cap, byref(n)) growing the buffer until the whole result fits."""
    while True:
        buf, n = C.create_string_buffer(cap), C.c_size_t(0)
        check(call(buf, cap, C.byref(n)), what)
        if n.value <= cap:
            return buf.raw[: n.value]
        cap = n.value


def _b(x):
    return x.encode() if isinstance(x, str) else bytes(x)


class Tokenizer:
    """tlhost::Tokenizer (host/tl_tokenizer.hpp) over the C view. Text goes in as str or bytes and comes out as bytes."""
    FAMILIES = ("unknown", "llama_sentencepiece", "llama3_tiktoken", "word_level")

    def __init__(self, gguf, family=0):
        self._own = None
        if not isinstance(gguf, GGUF):
            gguf = self._own = GGUF(gguf)
        h = C.c_void_p()
        check(lib().tlh_tokenizer_from_gguf(gguf.h, int(family), C.byref(h)), "tlh_tokenizer_from_gguf")
        self.h = h.value
        if self._own:  # the tokenizer copies what it needs out of the file
            self._own.close()

    def close(self):
        if self.h:
            lib().tlh_tokenizer_close(self.h)
            self.h = None

    def info(self):
        out = (C.c_int * 6)()
        check(lib().tlh_tokenizer_info(self.h, out), "tlh_tokenizer_info")
        return dict(zip(("family", "vocab", "bos", "eos", "unk", "pad"), list(out)))

    def encode(self, text, add_bos=True, add_eos=False):
        t = _b(text)
        cap = 2 * len(t) + 16
        while True:
            ids, n = np.zeros(cap, np.int32), C.c_int(0)
            check(lib().tlh_tokenizer_encode(self.h, t, len(t), int(add_bos), int(add_eos), ids.ctypes.data, cap, C.byref(n)),
                  "tlh_tokenizer_encode")
            if n.value <= cap:
                return ids[: n.value].tolist()
            cap = n.value

    def decode(self, ids, skip_special_tokens=True):
        a = np.ascontiguousarray(ids, np.int32)
        return _bytes_out(lambda buf, cap, n: lib().tlh_tokenizer_decode(self.h, a.ctypes.data if a.size else None, a.size,
                                                                         int(skip_special_tokens), buf, cap, n),
                          "tlh_tokenizer_decode")

    def apply_chat_template(self, user, system=b""):
        u, s = _b(user), _b(system)
        return _bytes_out(lambda buf, cap, n: lib().tlh_tokenizer_apply_chat_template(self.h, u, len(u), s, len(s), buf, cap, n),
                          "tlh_tokenizer_apply_chat_template")

    def format_prompt(self, user, system=b"", apply_q_a_format=False):
        u, s = _b(user), _b(system)
        return _bytes_out(lambda buf, cap, n: lib().tlh_tokenizer_format_prompt(self.h, u, len(u), s, len(s),
                                                                                int(apply_q_a_format), buf, cap, n),
                          "tlh_tokenizer_format_prompt")


class Session:
    def __init__(self, path, device=0, max_context=0):
        h = C.c_void_p()
        check(lib().tlh_session_open(os.fsencode(path), device, max_context, C.byref(h)), "tlh_session_open")
        self.h = h.value

    def close(self):
        if self.h:
            lib().tlh_session_close(self.h)
            self.h = None

    def set_repetition_penalty(self, p):
        check(lib().tlh_session_set_repetition_penalty(self.h, float(p)))

    def seed(self, s):
        check(lib().tlh_session_seed(self.h, int(s)))

    def generate(self, prompt, steps=128, temperature=0.1, top_k=40, top_p=0.9):
        p = np.ascontiguousarray(prompt, np.int32)
        out = np.zeros(max(steps, 1), np.int32)
        n = C.c_int(0)
        check(lib().tlh_session_generate(self.h, p.ctypes.data, p.size, int(steps), float(temperature), int(top_k),
                                         float(top_p), out.ctypes.data, out.size, C.byref(n)), "tlh_session_generate")
        return out[: n.value]

    def generate_text(self, prompt, steps=128, temperature=0.1, top_k=40, top_p=0.9, system_prompt=b"", apply_q_a_format=False):
        """TinyLlamaSession::generate as the reference declares it: text in, text (bytes) out."""
        p, s = _b(prompt), _b(system_prompt)
        return _bytes_out(lambda buf, cap, n: lib().tlh_session_generate_text(
            self.h, p, len(p), int(steps), float(temperature), int(top_k), float(top_p), s, len(s), int(apply_q_a_format),
            buf, cap, n), "tlh_session_generate_text", cap=4096 + 256 * max(int(steps), 0))

    def generate_batch_text(self, prompts, steps=128, temperature=0.1, top_k=40, top_p=0.9, system_prompt=b"",
                            apply_q_a_format=False):
        """TinyLlamaSession::generate_batch as the reference declares it: texts in, texts (bytes) out."""
        ps = [_b(p) for p in prompts]
        flat, sysb = b"".join(ps), _b(system_prompt)
        offs = (C.c_size_t * (len(ps) + 1))(*np.concatenate([[0], np.cumsum([len(p) for p in ps])]).astype(int).tolist())
        outs = (C.c_size_t * (len(ps) + 1))()
        raw = _bytes_out(lambda buf, cap, n: lib().tlh_session_generate_batch_text(
            self.h, flat, offs, len(ps), int(steps), float(temperature), int(top_k), float(top_p), sysb, len(sysb),
            int(apply_q_a_format), buf, cap, outs, n), "tlh_session_generate_batch_text",
            cap=4096 + 256 * max(int(steps), 0) * max(len(ps), 1))
        return [raw[outs[i]: outs[i + 1]] for i in range(len(ps))]

    def set_max_batch(self, n):
        check(lib().tlh_session_set_max_batch(self.h, int(n)), "tlh_session_set_max_batch")

    def generate_batch(self, prompts, steps=128, temperature=0.1, top_k=40, top_p=0.9):
        """TinyLlamaSession::generate_batch on ids: list of generated-id arrays, one per prompt."""
        flat = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32) for p in prompts]) if len(prompts) else
                                    np.zeros(0, np.int32), np.int32)
        offs = np.zeros(len(prompts) + 1, np.int32)
        offs[1:] = np.cumsum([len(p) for p in prompts])
        out = np.zeros((max(len(prompts), 1), max(steps, 1)), np.int32)
        lens = np.zeros(max(len(prompts), 1), np.int32)
        check(lib().tlh_session_generate_batch(self.h, flat.ctypes.data if flat.size else None, offs.ctypes.data,
                                               len(prompts), int(steps), float(temperature), int(top_k), float(top_p),
                                               out.ctypes.data, lens.ctypes.data), "tlh_session_generate_batch")
        return [out[i, : lens[i]].copy() for i in range(len(prompts))]

    def load_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_double()
        check(lib().tlh_session_load_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


def apply_repetition_penalty(logits, generated, penalty):
    l = np.array(logits, np.float32, copy=True)
    g = np.ascontiguousarray(generated, np.int32)
    check(lib().tlh_apply_repetition_penalty(l.ctypes.data, l.size, g.ctypes.data if g.size else None, g.size, float(penalty)))
    return l


def sampling_distribution(logits, temperature, top_k, top_p):
    l = np.ascontiguousarray(logits, np.float32)
    ids, probs, n = np.zeros(l.size, np.int32), np.zeros(l.size, np.float32), C.c_int(0)
    check(lib().tlh_sampling_distribution(l.ctypes.data, l.size, float(temperature), int(top_k), float(top_p),
                                          ids.ctypes.data, probs.ctypes.data, l.size, C.byref(n)))
    return ids[: n.value], probs[: n.value]


def sample(logits, temperature, top_k, top_p, seed):
    l = np.ascontiguousarray(logits, np.float32)
    out = C.c_int(-1)
    check(lib().tlh_sample(l.ctypes.data, l.size, float(temperature), int(top_k), float(top_p), int(seed), C.byref(out)))
    return out.value
