"""Multi-sequence decode (generate_batch api.cpp:836, forward_cpu_batch_generation model.cpp:1158) on MI355X:
B sequences with prompts of different lengths share every step's weight stream (skinny GEMMs on the matrix cores) over a
sequence-major KV cache. The bar is the reference's sequential fallback: every sequence must decode exactly as it does
alone -- logits within the decode tolerance of the single-sequence CPU oracle, identical greedy ids -- whatever else is
in the batch, in whatever slot order, and while other sequences join or leave."""
import numpy as np
import pytest

from oracle.oracle import OracleModel

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))


def oracle_run(om, prompt, n_steps):
    """greedy ids + the logits that produced each of them, single sequence, no penalty"""
    om.reset_kv()
    logits = None
    for pos, t in enumerate(prompt):
        _, logits = om.forward(om.lookup_embedding(int(t)), pos)
    ids, rows = [], []
    for i in range(n_steps):
        rows.append(logits)
        tok = int(np.argmax(logits))
        ids.append(tok)
        if i + 1 < n_steps:
            _, logits = om.forward(om.lookup_embedding(tok), len(prompt) + i)
    return np.array(ids, np.int32), np.stack(rows)


@pytest.mark.parametrize("gemv_max", ["8", "0", "3"])
@pytest.mark.parametrize("shape,mix", [("tiny-gqa", "q4_k_m"), ("tiny", "q6_k"), ("tiny-tp", "q8_0")])
def test_batch_matches_single_sequence_oracle(pkg, oracle, monkeypatch, shape, mix, gemv_max):
    """gemv_max: steps with at most that many sequences use multi-vector GEMVs, larger ones matrix-core GEMMs
    (12 = default: every step of this test on GEMVs; 0: all on GEMMs; 3: the test crosses between the two)."""
    from tinyllama_cpp_amd import engine as E
    monkeypatch.setenv("TL_BATCH_GEMV_MAX", gemv_max)
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=21)
    m = E.Model(cfg).load(tensors)
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(4)
    lens = [5, 1, 17, 9, 3]
    prompts = [rng.integers(0, cfg["vocab"], n).astype(np.int32) for n in lens]
    slots = [3, 0, 6, 2, 5]  # sequence i lives in cache slot slots[i] (not the batch order)
    n_steps = 7
    want = [oracle_run(om, p, n_steps) for p in prompts]
    m.batch_init(8)
    # prompts: batched prefill per sequence into its slot; its last logits pick the first generated id
    first = []
    for p, s, (ids, rows) in zip(prompts, slots, want):
        last = m.batch_prefill(s, p)
        assert close(last, rows[0], LOGIT_TOL).all(), np.abs(last - rows[0]).max()
        first.append(int(np.argmax(last)))
        assert first[-1] == ids[0]
    # step by step with logits: all five active, then sequences 1 and 3 leave (as finished ones do in the reference)
    toks, pos = list(first), list(lens)
    active = list(range(5))
    for step in range(1, n_steps):
        if step == 3:
            active = [0, 2, 4]
        logits, nxt = m.batch_decode([toks[i] for i in active], [pos[i] for i in active], [slots[i] for i in active])
        for j, i in enumerate(active):
            assert close(logits[j], want[i][1][step], LOGIT_TOL).all(), (step, i, np.abs(logits[j] - want[i][1][step]).max())
            assert nxt[j] == want[i][0][step]
            toks[i], pos[i] = int(nxt[j]), pos[i] + 1
    # the device-resident loop from the same prompts (fresh prefill: the cache rows are simply overwritten)
    for p, s in zip(prompts, slots):
        m.batch_prefill(s, p)
    ids = m.batch_generate_greedy(first, lens, slots, n_steps - 1)
    for i in range(5):
        assert np.array_equal(ids[:, i], want[i][0][1:]), i
    # the single-sequence engine state is untouched by batch mode
    single = m.generate(prompts[2], n_steps)
    assert np.array_equal(single, want[2][0])
    m.free()
    om.close()


@pytest.mark.parametrize("shape,mix,lens", [("tiny-gqa", "q4_k_m", [5, 1, 17, 9, 3]), ("tiny", "q6_k", [40, 2, 33]),
                                            ("tiny-tp", "q4_k_m", [12] * 12 + [7] * 13)])
def test_packed_prefill_of_all_prompts(pkg, oracle, shape, mix, lens):
    """batch_prefill_parallel (api.cpp:1144): every prompt's tokens in ONE packed batch with sequence-aware attention; the
    last-token logits equal the single-sequence oracle's, and decode continues from the caches it left. (The layers of
    a 35..250-token batch take the GEMM route; the head runs on the n_last rows only: multi-vector GEMVs for <= 12 rows,
    a GEMM for the 25 of the third case.)"""
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg, tensors = pkg.synth.build_model(shape, mix, seed=31)
    m = E.Model(cfg).load(tensors)
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(sum(lens))
    prompts = [rng.integers(0, cfg["vocab"], n).astype(np.int32) for n in lens]
    B = len(prompts)
    slots = list(rng.permutation(B + 2)[:B])
    m.batch_init(B + 2)
    logits, ids = m.batch_prefill_packed(prompts, slots)
    n_steps = 4
    want = [oracle_run(om, p, n_steps) for p in prompts]
    for i in range(B):
        assert close(logits[i], want[i][1][0], LOGIT_TOL).all(), (i, np.abs(logits[i] - want[i][1][0]).max())
        assert ids[i] == want[i][0][0]
    out = m.batch_generate_greedy(ids, lens, slots, n_steps - 1)
    for i in range(B):
        assert np.array_equal(out[:, i], want[i][0][1:]), i
    m.free()
    om.close()


def test_batch_long_contexts_and_eight_head_groups(pkg, oracle):
    """TinyLlama-1.1B's attention geometry (head_dim 64, 8 query heads per kv head: two workgroups per (kv head, token))
    at 2 layers, with contexts of 70..150 rows (several 64-row passes of the batched decode attention) next to a short one."""
    from tinyllama_cpp_amd import engine as E
    E.init(0)
    cfg = dict(pkg.synth.SHAPES["tinyllama-1.1b"], n_layers=2, vocab=2048, max_pos=256)
    cfg, tensors = pkg.synth.build_model(cfg, "q4_k_m", seed=23)
    m = E.Model(cfg).load(tensors)
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(6)
    lens = [150, 70, 4]
    prompts = [rng.integers(0, cfg["vocab"], n).astype(np.int32) for n in lens]
    n_steps = 5
    want = [oracle_run(om, p, n_steps) for p in prompts]
    m.batch_init(4)
    logits, ids = m.batch_prefill_packed(prompts, slots=[2, 0, 3])
    for i in range(3):
        assert close(logits[i], want[i][1][0], LOGIT_TOL).all(), (i, np.abs(logits[i] - want[i][1][0]).max())
    toks, pos = [int(x) for x in ids], list(lens)
    for step in range(1, n_steps):
        lg, nxt = m.batch_decode(toks, pos, [2, 0, 3])
        for i in range(3):
            assert close(lg[i], want[i][1][step], LOGIT_TOL).all(), (step, i, np.abs(lg[i] - want[i][1][step]).max())
            assert nxt[i] == want[i][0][step]
        toks, pos = [int(x) for x in nxt], [p + 1 for p in pos]
    m.free()
    om.close()


def test_packed_prefill_rejects_bad_batches(pkg):
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd.capi import TlError, check, lib
    E.init(0)
    cfg, tensors = pkg.synth.build_model("tiny", "q4_k_m", seed=1)
    m = E.Model(cfg).load(tensors)
    m.batch_init(2)

    def call(toks, pos, seq, last):
        a = [np.ascontiguousarray(x, np.int32) for x in (toks, pos, seq, last)]
        ids = np.zeros(max(len(last), 1), np.int32)
        check(lib().tl_model_batch_prefill_packed(m.h, a[0].size, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                                  a[3].size, a[3].ctypes.data, None, ids.ctypes.data), "packed")
        return ids
    call([1, 2, 3], [0, 1, 0], [0, 0, 1], [1, 2])  # fine
    with pytest.raises(TlError):
        call([1, 2], [1, 0], [0, 0], [1])  # positions of a sequence out of order
    with pytest.raises(TlError):
        call([1, 2], [0, 0], [0, 2], [1])  # sequence index out of range
    with pytest.raises(TlError):
        call([1, 2], [0, 1], [0, 0], [2])  # output row outside the batch
    with pytest.raises(TlError):
        call([1, 2, 3], [0, 0, 1], [0, 1, 1], [0, 1, 2])  # more output rows than max_batch
    with pytest.raises(TlError):
        call([1, cfg["vocab"]], [0, 1], [0, 0], [1])  # token id
    m.free()


def test_batch_argument_errors(pkg):
    from tinyllama_cpp_amd import engine as E
    from tinyllama_cpp_amd.capi import TlError
    E.init(0)
    cfg, tensors = pkg.synth.build_model("tiny", "q4_k_m", seed=1)
    m = E.Model(cfg).load(tensors)
    with pytest.raises(TlError):
        m.batch_decode([1], [0], [0])  # before batch_init
    with pytest.raises(TlError):
        m.batch_init(0)
    with pytest.raises(TlError):
        m.batch_init(100000)
    m.batch_init(4)
    with pytest.raises(TlError):
        m.batch_prefill(4, [1, 2, 3])  # slot out of range
    with pytest.raises(TlError):
        m.batch_decode([1, 2], [0, 0], [1, 1])  # repeated sequence index
    with pytest.raises(TlError):
        m.batch_decode([1], [cfg["max_pos"]], [0])  # position out of range
    with pytest.raises(TlError):
        m.batch_decode([cfg["vocab"]], [0], [0])  # token id out of range
    with pytest.raises(TlError):
        m.batch_decode([1, 2, 3, 4, 5], [0] * 5, [0, 1, 2, 3, 4])  # more than max_batch (and slot 4)
    with pytest.raises(TlError):
        m.batch_generate_greedy([1], [cfg["max_pos"] - 2], [0], 5)  # would run past max_pos
    m.free()
    with pytest.raises(TlError):
        E.Model(dict(pkg.synth.SHAPES["tiny-gqa"]), tp_rank=0, tp_size=2).batch_init(2)  # not finalized / sharded


def test_deferred_ksplit_reductions_are_bit_identical(pkg, monkeypatch):
    """Skinny GEMMs split K; their partial tiles are summed by the NEXT kernel (RMSNorm split, SwiGLU split, RoPE + cache
    write) instead of a reduction launch of their own. Same sums in the same order: logits of a batched step, a packed
    prefill and an all-logits prefill are bit-for-bit those of TL_GEMM_DEFER=0 (one ksplit_reduce launch per GEMM), with
    and without q | k | v and gate | up merged into one launch each."""
    from tinyllama_cpp_amd import engine as E
    monkeypatch.setenv("TL_BATCH_GEMV_MAX", "0")
    E.init(0)
    rng = np.random.default_rng(8)
    for shape, mix in (("tiny-tp", "q4_k_m"), ("tiny-gqa", "q6_k")):
        cfg, tensors = pkg.synth.build_model(shape, mix, seed=5)
        prompts = [rng.integers(0, cfg["vocab"], n).astype(np.int32) for n in [9, 3, 14, 6, 11, 2, 8, 5, 7, 10, 4, 12]]
        long_prompt = rng.integers(0, cfg["vocab"], 100).astype(np.int32)
        got = {}
        for defer, merge in (("0", "1"), ("1", "0"), ("1", "1")):
            monkeypatch.setenv("TL_GEMM_DEFER", defer)  # read when the model's prefill workspaces are made
            monkeypatch.setenv("TL_GEMM_MERGE", merge)  # q | k | v and gate | up in one launch each (needs DEFER)
            monkeypatch.setenv("TL_ATTN_PLANES", defer)  # batched attention writes the attn_output GEMM's planes itself
            m = E.Model(cfg).load(tensors)
            m.batch_init(12)
            logits, ids = m.batch_prefill_packed(prompts)
            step, nxt = m.batch_decode(list(ids), [len(p) for p in prompts], list(range(12)))
            more = m.batch_generate_greedy(list(nxt), [len(p) + 1 for p in prompts], list(range(12)), 3)
            _, all_logits = m.prefill(long_prompt, all_logits=True)       # 100 tokens: integer-plane kernel, 128-token tiles
            _, all_logits40 = m.prefill(long_prompt[:40], all_logits=True)  # 40 tokens: its 64-token tiles
            got[defer + merge] = [np.array(x) for x in (logits, ids, step, nxt, more, all_logits, all_logits40)]
            m.free()
        for other in ("10", "11"):
            for a, b in zip(got["01"], got[other]):
                assert a.shape == b.shape and np.array_equal(a, b), (shape, other, np.abs(a.astype(np.float64) - b).max())
