"""ctypes view of the C++ host layer (libtlhost.so, include/tlhost.h): GGUF reader, Session, host-side sampler.

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
