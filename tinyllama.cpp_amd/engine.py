"""Thin host-side objects over the C-ABI (capi.py): device buffers, quantized weights, the model engine.

These mirror what the reference's host keeps as raw device-pointer members (model.h:498-554 weights and
workspaces, model.h:135-143 KV) — here each is an owning Python object around a C-ABI handle. All compute
happens in libtlhip.so; nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import capi, tp
from .capi import check, lib

_initialized = False


def init(device=0):
    global _initialized
    check(lib().tl_init(device), "tl_init")
    _initialized = True


def _ensure_init():
    if not _initialized:
        init(0)


class DeviceBuffer:
    """A device allocation with numpy round trips (tl_malloc / tl_memcpy_*)."""

    def __init__(self, nbytes):
        _ensure_init()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().tl_malloc(C.byref(p), self.nbytes), "tl_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 4))
        if a.nbytes:
            check(lib().tl_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, None), "tl_memcpy_h2d")
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(lib().tl_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "tl_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().tl_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class QWeight:
    """One quantized matrix resident in HBM (tl_qweight). `blocks`: raw GGUF blocks as a uint8 array."""

    def __init__(self, typ, blocks, rows, cols):
        _ensure_init()
        blocks = np.ascontiguousarray(blocks)
        h = C.c_void_p()
        check(lib().tl_qweight_upload(typ, blocks.ctypes.data, rows, cols, C.byref(h), None), "tl_qweight_upload")
        self.h, self.type, self.rows, self.cols = h.value, typ, rows, cols

    def device_bytes(self):
        n = C.c_size_t()
        check(lib().tl_qweight_info(self.h, None, None, None, C.byref(n)), "tl_qweight_info")
        return n.value

    def expand(self):
        """Keep the expanded prefill planes of this matrix (tl_qweight_expand); returns their device bytes."""
        n = C.c_size_t()
        check(lib().tl_qweight_expand(self.h, C.byref(n), None), "tl_qweight_expand")
        check(lib().tl_device_sync(), "sync")
        return n.value

    def dequantize(self, row0=0, nrows=None):
        nrows = self.rows - row0 if nrows is None else nrows
        out = DeviceBuffer(max(4, 4 * nrows * self.cols))
        check(lib().tl_dequantize_rows(self.h, row0, nrows, out.ptr, None), "tl_dequantize_rows")
        check(lib().tl_device_sync(), "sync")
        return out.to_numpy(np.float32, (nrows, self.cols))

    def matvec(self, x):
        xb = DeviceBuffer.from_numpy(np.asarray(x, np.float32))
        yb = DeviceBuffer(4 * self.rows)
        check(lib().tl_matvec_q(self.h, xb.ptr, yb.ptr, None), "tl_matvec_q")
        check(lib().tl_device_sync(), "sync")
        return yb.to_numpy(np.float32, (self.rows,))

    def free(self):
        if self.h:
            lib().tl_qweight_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sample_top_k_top_p(logits, temperature, top_k, top_p, u):
    """Device sampler as an operator: (drawn id, candidate ids, renormalised candidate probabilities)."""
    l = np.ascontiguousarray(logits, np.float32)
    cap = l.size if top_k <= 0 or top_k > l.size else top_k
    lb = DeviceBuffer.from_numpy(l)
    idb, nb = DeviceBuffer(4), DeviceBuffer(4)
    cib, cpb = DeviceBuffer(4 * cap), DeviceBuffer(4 * cap)
    check(lib().tl_sample_top_k_top_p(lb.ptr, l.size, float(temperature), int(top_k), float(top_p), float(u), idb.ptr,
                                      cib.ptr, cpb.ptr, nb.ptr, None), "tl_sample_top_k_top_p")
    check(lib().tl_device_sync(), "sync")
    n = int(nb.to_numpy(np.int32, (1,))[0])
    return (int(idb.to_numpy(np.int32, (1,))[0]), cib.to_numpy(np.int32, (cap,))[:n].copy(),
            cpb.to_numpy(np.float32, (cap,))[:n].copy())


def tp_unique_id():
    """128 bytes identifying a new RCCL communicator (rank 0 creates it, the host broadcasts it)."""
    buf = (C.c_ubyte * 128)()
    check(lib().tl_tp_unique_id(buf), "tl_tp_unique_id")
    return bytes(buf)


def tp_local_group_decode(models, token, pos, want_logits=True):
    """One decode step of a tensor-parallel group whose shards all live in this process (test hook)."""
    n = len(models)
    arr = (C.c_void_p * n)(*[m.h for m in models])
    logits = np.empty(models[0].cfg["vocab"], np.float32) if want_logits else None
    nxt = C.c_int(-1)
    check(lib().tl_tp_local_group_decode(arr, n, int(token), int(pos), logits.ctypes.data if want_logits else None,
                                         C.byref(nxt)), "tl_tp_local_group_decode")
    return logits, nxt.value


def tp_local_group_prefill(models, tokens, start_pos=0):
    """Batched prefill of a tensor-parallel group whose shards all live in this process (test hook): last logits."""
    n = len(models)
    arr = (C.c_void_p * n)(*[m.h for m in models])
    toks = np.ascontiguousarray(tokens, np.int32)
    last = np.empty(models[0].cfg["vocab"], np.float32)
    check(lib().tl_tp_local_group_prefill(arr, n, toks.ctypes.data, toks.size, int(start_pos), last.ctypes.data),
          "tl_tp_local_group_prefill")
    return last


KINDS = {"attn_q": capi.T_ATTN_Q, "attn_k": capi.T_ATTN_K, "attn_v": capi.T_ATTN_V,
         "attn_output": capi.T_ATTN_OUTPUT, "ffn_gate": capi.T_FFN_GATE, "ffn_up": capi.T_FFN_UP,
         "ffn_down": capi.T_FFN_DOWN, "attn_norm": capi.T_ATTN_NORM, "ffn_norm": capi.T_FFN_NORM}


def tensor_kind(name):
    """GGUF tensor name (weight_management.cpp:732-960) -> (kind, layer)."""
    if name == "token_embd.weight":
        return capi.T_TOKEN_EMBD, -1
    if name == "output_norm.weight":
        return capi.T_OUTPUT_NORM, -1
    if name == "output.weight":
        return capi.T_OUTPUT, -1
    _, li, kn, _ = name.split(".")
    return KINDS[kn], int(li)


PROFILE_KIND_INDEX = {"qkv_gemv": 1, "attn_out_gemv": 4, "gate_up_gemv": 5, "down_gemv": 6, "lm_head_gemv": 7,
                      "tp_all_reduce": 9}


class Model:
    """The device-resident model (tl_model): cfg is a dict with the synth.SHAPES keys."""

    def __init__(self, cfg, tp_rank=0, tp_size=1):
        """tp_size > 1: this handle is shard `tp_rank` of a tensor-parallel group (tp.py). set_tensor() still takes the
        FULL tensors and slices them; call tp_init(unique_id) before finalize() when the ranks are separate processes."""
        _ensure_init()
        tp.check(cfg, tp_size)
        self.tp_rank, self.tp_size = int(tp_rank), int(tp_size)
        self.cfg = dict(cfg)
        c = capi.ModelConfig(cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv_heads"], cfg["n_layers"],
                             cfg["vocab"], cfg["max_pos"], cfg["eps"], cfg["rope_theta"], tp_rank, tp_size)
        h = C.c_void_p()
        check(lib().tl_model_create(C.byref(c), C.byref(h)), "tl_model_create")
        self.h = h.value

    def set_tensor(self, name, typ, arr, rows, cols):
        kind, layer = tensor_kind(name)
        arr, rows, cols = tp.shard_tensor(name, typ, arr, rows, cols, self.tp_rank, self.tp_size)
        arr = np.ascontiguousarray(arr)
        check(lib().tl_model_set_tensor(self.h, kind, layer, typ, arr.ctypes.data, rows, cols),
              f"tl_model_set_tensor({name})")

    def load(self, tensors):
        for name, (typ, arr, rows, cols) in tensors.items():
            self.set_tensor(name, typ, arr, rows, cols)
        self.finalize()
        return self

    def tp_init(self, unique_id):
        """Join the RCCL communicator of the group (unique_id: 128 bytes from tp_unique_id() on rank 0)."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        check(lib().tl_model_tp_init(self.h, buf), "tl_model_tp_init")

    # ---- multi-sequence decode -------------------------------------------------------------------------
    def batch_init(self, max_batch):
        check(lib().tl_model_batch_init(self.h, int(max_batch)), "tl_model_batch_init")

    def batch_prefill(self, seq_idx, tokens, start_pos=0):
        """Prompt of one sequence into its region of the sequence-major cache; returns its last logits."""
        toks = np.ascontiguousarray(tokens, np.int32)
        last = np.empty(self.cfg["vocab"], np.float32)
        check(lib().tl_model_batch_prefill(self.h, int(seq_idx), toks.ctypes.data, toks.size, int(start_pos),
                                           last.ctypes.data), "tl_model_batch_prefill")
        return last

    def batch_prefill_packed(self, prompts, slots=None, want_logits=True):
        """All prompts as ONE packed batch (batch_prefill_parallel): prompt i into cache slot slots[i] (default i) from
        position 0. Returns (last-token logits [n, vocab] or None, their first-maximum ids [n])."""
        n = len(prompts)
        slots = list(range(n)) if slots is None else [int(x) for x in slots]
        toks = np.concatenate([np.asarray(p, np.int32) for p in prompts]).astype(np.int32)
        pos = np.concatenate([np.arange(len(p), dtype=np.int32) for p in prompts])
        seq = np.concatenate([np.full(len(p), s, np.int32) for p, s in zip(prompts, slots)])
        last = (np.cumsum([len(p) for p in prompts]) - 1).astype(np.int32)
        logits = np.empty((n, self.cfg["vocab"]), np.float32) if want_logits else None
        ids = np.empty(n, np.int32)
        check(lib().tl_model_batch_prefill_packed(self.h, toks.size, toks.ctypes.data, pos.ctypes.data, seq.ctypes.data, n,
                                                  last.ctypes.data, logits.ctypes.data if want_logits else None,
                                                  ids.ctypes.data), "tl_model_batch_prefill_packed")
        return logits, ids

    def batch_decode(self, tokens, positions, seq_idx, want_logits=True):
        """One step for the active sequences: (logits [n, vocab] or None, first-maximum ids [n])."""
        t, p, q = (np.ascontiguousarray(a, np.int32) for a in (tokens, positions, seq_idx))
        assert t.size == p.size == q.size
        logits = np.empty((t.size, self.cfg["vocab"]), np.float32) if want_logits else None
        nxt = np.empty(t.size, np.int32)
        check(lib().tl_model_batch_decode(self.h, t.size, t.ctypes.data, p.ctypes.data, q.ctypes.data,
                                          logits.ctypes.data if want_logits else None, nxt.ctypes.data),
              "tl_model_batch_decode")
        return logits, nxt

    def batch_generate_greedy(self, first_tokens, start_pos, seq_idx, n_steps):
        """n_steps greedy steps for all given sequences on the device: ids [n_steps, n]."""
        t, p, q = (np.ascontiguousarray(a, np.int32) for a in (first_tokens, start_pos, seq_idx))
        assert t.size == p.size == q.size
        out = np.zeros((max(n_steps, 1), t.size), np.int32)
        check(lib().tl_model_batch_generate_greedy(self.h, t.size, t.ctypes.data, p.ctypes.data, q.ctypes.data,
                                                   int(n_steps), out.ctypes.data), "tl_model_batch_generate_greedy")
        return out[:n_steps]

    def tp_peer_export(self):
        """This rank's 64-byte IPC handle of its peer-exchange receive area (tl_model_tp_peer_export)."""
        buf = (C.c_ubyte * 64)()
        check(lib().tl_model_tp_peer_export(self.h, buf), "tl_model_tp_peer_export")
        return bytes(buf)

    def tp_peer_import(self, handles):
        """Map the areas of all ranks (handles: tp_size x 64 bytes in rank order) and run the collective self-check:
        True when every value arrived here. The ranks must still agree (min over ranks) before tp_peer_enable()."""
        raw = b"".join(bytes(h) for h in handles)
        assert len(raw) == 64 * self.tp_size, "one 64-byte handle per rank"
        buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
        ok = C.c_int(0)
        check(lib().tl_model_tp_peer_import(self.h, buf, C.byref(ok)), "tl_model_tp_peer_import")
        return bool(ok.value)

    def tp_peer_enable(self, on=True):
        check(lib().tl_model_tp_peer_enable(self.h, int(bool(on))), "tl_model_tp_peer_enable")

    def tp_peer_enabled(self):
        out = C.c_int(0)
        check(lib().tl_model_tp_peer_enabled(self.h, C.byref(out)), "tl_model_tp_peer_enabled")
        return bool(out.value)

    def graph_nodes(self):
        """Launches in the captured decode step (0: not captured), tl_model_graph_nodes."""
        out = C.c_int(0)
        check(lib().tl_model_graph_nodes(self.h, C.byref(out)), "tl_model_graph_nodes")
        return int(out.value)

    def tp_peer_fused(self):
        """True when the exchanges ride inside the attn_output / ffn_down / lm_head launches (tl_model_tp_peer_fused)."""
        out = C.c_int(0)
        check(lib().tl_model_tp_peer_fused(self.h, C.byref(out)), "tl_model_tp_peer_fused")
        return bool(out.value)

    def finalize(self):
        check(lib().tl_model_finalize(self.h), "tl_model_finalize")

    def graph_captured(self):
        """True when the decode step replays as one hipGraph (False: eager launches, see tl_model_graph_captured)."""
        out = C.c_int(0)
        check(lib().tl_model_graph_captured(self.h, C.byref(out)), "tl_model_graph_captured")
        return bool(out.value)

    def reset_kv(self):
        check(lib().tl_model_reset_kv(self.h), "tl_model_reset_kv")

    def decode(self, token, pos, want_logits=True):
        logits = np.empty(self.cfg["vocab"], np.float32) if want_logits else None
        nxt = C.c_int(-1)
        check(lib().tl_model_decode(self.h, int(token), int(pos), logits.ctypes.data if want_logits else None,
                                    C.byref(nxt)), "tl_model_decode")
        return logits, nxt.value

    def generate_greedy(self, first_token, start_pos, n_steps, penalty=1.0, history=()):
        hist = np.ascontiguousarray(history, np.int32)
        out = np.zeros(max(n_steps, 1), np.int32)
        check(lib().tl_model_generate_greedy(self.h, int(first_token), int(start_pos), int(n_steps), float(penalty),
                                             hist.ctypes.data if hist.size else None, hist.size, out.ctypes.data),
              "tl_model_generate_greedy")
        return out[:n_steps]

    def generate_sampled(self, first_token, start_pos, n_steps, penalty=1.0, history=(), temperature=0.8, top_k=40,
                         top_p=0.9, seed=0):
        hist = np.ascontiguousarray(history, np.int32)
        out = np.zeros(max(n_steps, 1), np.int32)
        check(lib().tl_model_generate_sampled(self.h, int(first_token), int(start_pos), int(n_steps), float(penalty),
                                              hist.ctypes.data if hist.size else None, hist.size, float(temperature),
                                              int(top_k), float(top_p), int(seed), out.ctypes.data),
              "tl_model_generate_sampled")
        return out[:n_steps]

    def set_prefill(self, terms=0, min_batch=8):
        """bf16 MFMAs per FP32 product in the batched-prefill GEMMs (6 default / 3 / 1; 0 = process-wide value) and the
        prompt length from which tl_model_prefill runs batched."""
        check(lib().tl_model_set_prefill(self.h, int(terms), int(min_batch)), "tl_model_set_prefill")

    def prefill(self, tokens, start_pos=0, all_logits=False):
        toks = np.ascontiguousarray(tokens, np.int32)
        last = np.empty(self.cfg["vocab"], np.float32)
        allb = np.empty((toks.size, self.cfg["vocab"]), np.float32) if all_logits else None
        check(lib().tl_model_prefill(self.h, toks.ctypes.data, toks.size, int(start_pos), last.ctypes.data,
                                     allb.ctypes.data if all_logits else None), "tl_model_prefill")
        return (last, allb) if all_logits else last

    def generate(self, prompt, max_new, penalty=1.0, eos_id=-1):
        """TinyLlamaSession::generate on token ids, greedy (api.cpp:585-813): prompts of >= 32
        tokens go through tl_model_prefill, shorter ones are fed token by token; then device-side greedy steps."""
        prompt = [int(t) for t in prompt]
        P = len(prompt)
        max_pos = self.cfg["max_pos"]
        self.reset_kv()  # kv_cache_.clear_data(), api.cpp:592
        if P >= 32:
            logits = self.prefill(prompt, 0)
        else:
            for pos in range(P - 1):
                if pos >= max_pos:
                    return np.zeros(0, np.int32)
                self.decode(prompt[pos], pos, want_logits=False)
            if P - 1 >= max_pos:
                return np.zeros(0, np.int32)
            logits, _ = self.decode(prompt[P - 1], P - 1)
        first = int(np.argmax(logits))  # empty history: the penalty is a no-op for the first sampled id
        ids = [first]
        # remaining steps on the device; the generate loop's bounds: pos < P + max_new - 1 and pos < max_pos
        n_more = min(max_new - 1, max_pos - P)
        if n_more > 0:  # an EOS sampled at pos == P-1 does not stop the loop (api.cpp:794)
            more = self.generate_greedy(first, P, n_more, penalty, history=[first])
            for t in more:
                ids.append(int(t))
                if int(t) == eos_id:
                    break
        return np.array(ids[:max_new], np.int32)

    PROFILE_KINDS = ("embed", "qkv_gemv", "rope_kv", "attention", "attn_out_gemv", "gate_up_gemv", "down_gemv",
                     "lm_head_gemv", "sample")

    def profile_decode(self, first_token, start_pos, n_steps):
        """{kind: (avg_ms_per_launch, launches)} from HIP events around every kernel of eager decode steps."""
        ms = np.zeros(9, np.float32)
        cnt = np.zeros(9, np.int32)
        check(lib().tl_model_profile_decode(self.h, int(first_token), int(start_pos), int(n_steps), ms.ctypes.data,
                                            cnt.ctypes.data), "tl_model_profile_decode")
        return {k: (float(ms[i]) / max(int(cnt[i]), 1), int(cnt[i])) for i, k in enumerate(self.PROFILE_KINDS)}

    def bench_kernel(self, kind, iters=256, warmup=32):
        """(avg_us_per_launch, gguf_bytes_per_launch) of one kind of the step's GEMV kernels, back to back."""
        us, nb = C.c_float(), C.c_double()
        check(lib().tl_model_bench_kernel(self.h, PROFILE_KIND_INDEX[kind], int(iters), int(warmup), C.byref(us),
                                          C.byref(nb)), "tl_model_bench_kernel")
        return us.value, nb.value

    def device_ptr(self, which, layer=0):
        p = C.c_void_p()
        check(lib().tl_model_device_ptr(self.h, which, layer, C.byref(p)), "tl_model_device_ptr")
        return p.value

    def kv_cache(self, layer, n_pos):
        kvd = self.cfg["n_kv_heads"] * (self.cfg["hidden"] // self.cfg["n_heads"])
        out = []
        for which in (0, 1):
            a = np.empty((n_pos, kvd), np.float32)
            check(lib().tl_memcpy_d2h(a.ctypes.data, self.device_ptr(which, layer), a.nbytes, None), "d2h")
            out.append(a)
        return out

    def set_kv_cache(self, layer, k, v):
        """Overwrite the first len(k) positions of a layer's K/V cache from host arrays [n_pos][kv_dim] (tests)."""
        for which, a in ((0, k), (1, v)):
            a = np.ascontiguousarray(a, np.float32)
            check(lib().tl_memcpy_h2d(self.device_ptr(which, layer), a.ctypes.data, a.nbytes, None), "h2d")
        check(lib().tl_device_sync(), "sync")

    def bytes_per_token(self, pos):
        w, kv = C.c_double(), C.c_double()
        check(lib().tl_model_bytes_per_token(self.h, int(pos), C.byref(w), C.byref(kv)), "tl_model_bytes_per_token")
        return w.value, kv.value

    def prefill_planes_bytes(self):
        """Device bytes of the expanded prefill planes kept for this model (0: none), tl_model_prefill_planes_bytes."""
        n = C.c_size_t()
        check(lib().tl_model_prefill_planes_bytes(self.h, C.byref(n)), "tl_model_prefill_planes_bytes")
        return n.value

    def free(self):
        if self.h:
            lib().tl_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
