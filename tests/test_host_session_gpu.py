"""The C++ host layer end to end on MI355X: GGUF file -> tlhost::Session (mmap -> tl_model_set_tensor -> decode graph)
-> generate() on token ids, against the CPU oracle's restatement of the same loop; and the tl_run binary."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import OracleModel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H(pkg):
    from tinyllama_cpp_amd import hostapi
    hostapi.lib()
    return hostapi


@pytest.mark.parametrize("shape,mix", [("tiny", "q4_k_m"), ("tiny-gqa", "q6_k"), ("tiny-gqa", "q8_0")])
def test_session_generate_matches_oracle(H, pkg, oracle, tmp_path, shape, mix):
    S = pkg.synth
    cfg, tensors = S.build_model(shape, mix, seed=13)
    path = S.write_gguf(str(tmp_path / "m.gguf"), cfg, tensors)
    s = H.Session(path)
    ntens, qbytes, secs = s.load_stats()
    assert ntens == len(tensors)
    assert qbytes == sum(a.nbytes for t, a, r, c in tensors.values() if t != S.F32)
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(3)
    for plen in (1, 5, 40):  # 40 >= 32 takes the batched-prefill route (api.cpp:601)
        prompt = rng.integers(0, cfg["vocab"], plen).astype(np.int32)
        for pen in (1.0, 1.3):
            s.set_repetition_penalty(pen)
            got = s.generate(prompt, 10, temperature=0.0)
            want = om.generate(prompt, 10, penalty=pen, eos_id=2)
            want = want[0] if isinstance(want, tuple) else want
            assert np.array_equal(got, want), (plen, pen, got, want)
    # context bound: generation stops at max_position_embeddings
    prompt = rng.integers(0, cfg["vocab"], cfg["max_pos"] - 3).astype(np.int32)
    assert len(s.generate(prompt, 50, temperature=0.0)) == 4
    # sampled generation: reproducible with a seed, ids in range, differs from greedy somewhere
    s.seed(7)
    a = s.generate(prompt[:6], 12, temperature=1.0, top_k=20, top_p=0.95)
    s.seed(7)
    b = s.generate(prompt[:6], 12, temperature=1.0, top_k=20, top_p=0.95)
    assert np.array_equal(a, b) and ((a >= 0) & (a < cfg["vocab"])).all()
    assert len(a) == 12 or (len(a) < 12 and a[-1] == 2)  # an EOS (id 2 in the synthetic GGUF) ends the stream early
    s.close()
    om.close()


def test_session_generate_batch_matches_single_sequence_runs(H, pkg, oracle, tmp_path):
    """TinyLlamaSession::generate_batch on ids (api.cpp:836): every prompt's ids equal the oracle's single-sequence
    greedy run without penalty, cut at EOS; the reference's argument errors; sampled batches stay in range."""
    from tinyllama_cpp_amd.hostapi import TlHostError as HostError
    S = pkg.synth
    cfg, tensors = S.build_model("tiny-gqa", "q4_k_m", seed=17)
    path = S.write_gguf(str(tmp_path / "b.gguf"), cfg, tensors)
    s = H.Session(path)
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(8)
    prompts = [rng.integers(0, cfg["vocab"], n).astype(np.int32) for n in (3, 11, 1, 35)]
    with pytest.raises(HostError):
        s.generate_batch(prompts, 8, temperature=0.0)  # max_batch_size is 1 by default (api.h:46)
    s.set_max_batch(6)
    with pytest.raises(HostError):
        s.generate_batch([], 8)  # "Cannot process empty prompts vector" (api.cpp:844)
    got = s.generate_batch(prompts, 9, temperature=0.0)
    assert len(got) == len(prompts)
    for p, g in zip(prompts, got):
        want = om.generate(p, 9, penalty=1.0, eos_id=2)
        want = want[0] if isinstance(want, tuple) else want
        assert np.array_equal(g, want), (len(p), g, want)
    # a second call reuses the batch cache; single-sequence generate still works beside it
    again = s.generate_batch(prompts[:2], 5, temperature=0.0)
    assert np.array_equal(again[0], got[0][:5]) and np.array_equal(again[1], got[1][:5])
    s.set_repetition_penalty(1.0)
    one = s.generate(prompts[1], 9, temperature=0.0)
    assert np.array_equal(one, got[1])
    s.seed(3)
    sampled = s.generate_batch(prompts, 6, temperature=0.9, top_k=30, top_p=0.9)
    for g in sampled:
        assert 1 <= len(g) <= 6 and ((g >= 0) & (g < cfg["vocab"])).all()
        assert len(g) == 6 or g[-1] == 2
    s.close()
    om.close()


def test_session_generate_batch_context_bound_is_per_sequence(H, pkg, oracle, tmp_path):
    """A sequence that reaches the end of the context drops out of the step's batch (api.cpp:1031-1046); the others go on
    to the requested number of steps. (The greedy route used to cut every sequence where the longest one ended.)"""
    S = pkg.synth
    cfg = dict(S.SHAPES["tiny-gqa"], max_pos=48)
    cfg, tensors = S.build_model(cfg, "q4_k_m", seed=17)
    s = H.Session(S.write_gguf(str(tmp_path / "c.gguf"), cfg, tensors))
    om = OracleModel(oracle, cfg, tensors)
    rng = np.random.default_rng(5)
    prompts = [rng.integers(3, cfg["vocab"], n).astype(np.int32) for n in (40, 5, 44)]
    s.set_max_batch(4)
    got = s.generate_batch(prompts, 20, temperature=0.0)
    for p, g in zip(prompts, got):
        n = min(20, cfg["max_pos"] - len(p) + 1)  # the first id comes from the prompt, then one per free position
        want = om.generate(p, n, penalty=1.0, eos_id=2)
        want = want[0] if isinstance(want, tuple) else want
        assert np.array_equal(g, want), (len(p), g, want)
    s.close()
    om.close()


def test_session_rejects_unusable_files(H, pkg, tmp_path):
    S = pkg.synth
    cfg, tensors = S.build_model("tiny", "q4_k_m", seed=13)
    missing = {k: v for k, v in tensors.items() if k != "blk.1.ffn_up.weight"}
    with pytest.raises(H.TlHostError):
        H.Session(S.write_gguf(str(tmp_path / "missing.gguf"), cfg, missing))
    with pytest.raises(H.TlHostError):
        H.Session(str(tmp_path / "nope.gguf"))


def test_tl_run_binary(pkg, oracle, tmp_path):
    S = pkg.synth
    exe = os.path.join(ROOT, "tinyllama.cpp_amd", "host", "tl_run")
    assert os.path.exists(exe), "build the host layer first (__graft_entry__.build())"
    cfg, tensors = S.build_model("tiny", "q4_k_m", seed=42)
    path = S.write_gguf(str(tmp_path / "m.gguf"), cfg, tensors)
    out = subprocess.run([exe, path, "--ids", "3,14,15,9", "--steps", "8", "--penalty", "1.3"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])["generated"]
    om = OracleModel(oracle, cfg, tensors)
    want = om.generate(np.array([3, 14, 15, 9], np.int32), 8, penalty=1.3, eos_id=2)
    want = want[0] if isinstance(want, tuple) else want
    assert got == list(map(int, want))
    om.close()


def test_session_generates_from_text(H, pkg, oracle, tmp_path):
    """TinyLlamaSession::generate as the reference declares it (text in, text out; api.cpp:488-835): the file's tokenizer
    formats and encodes the prompt, the ids run through the same loop as above, the generated ids are decoded without
    special tokens. Word-level, SentencePiece-style and Llama-3-style (recognised from the file's metadata, 128256 vocabulary
    entries) vocabularies (tokenizer parity itself: tests/test_tokenizer_cpu.py, unpinned); and tl_run --prompt."""
    import test_tokenizer_cpu as TT
    from oracle import tokenizer_oracle as TO
    S = pkg.synth
    sp = TT.spm_parts()
    tk = TT.tiktoken_parts()
    for family in ("word", "spm", "tiktoken"):
        cfg = dict(S.SHAPES["tiny"])
        if family == "tiktoken":
            # Llama-3 as the reference recognises it (model_config.cpp:232): gpt2 + merges + 128256 vocabulary entries, which
            # is also the model's vocabulary size; no template in the file -> the hard-coded Llama-3 instruct layout
            V = 128256
            fill = V - len(tk["tokens"])
            tokens = tk["tokens"] + [b"[unused%d]" % i for i in range(fill)]
            types = tk["token_types"] + [1] * fill
            cfg["vocab"] = V
            kvs = S.tokenizer_kvs(tokens, model="gpt2", merges=tk["merges"], token_types=types, bos=tk["bos"], eos=tk["eos"],
                                  vocab_size=V)
            o = TO.TokenizerOracle(TO.LLAMA3_TIKTOKEN, tokens, merges=tk["merges"], token_types=types, bos=tk["bos"],
                                   eos=tk["eos"])
            prompt, system, eos = b"what is the answer ?", b"be brief", tk["eos"]
        elif family == "word":
            tokens = [b"<pad>", b"<unk>", b"<s>", b"</s>"] + [b"w%d" % i for i in range(4, cfg["vocab"])]
            kvs = S.tokenizer_kvs(tokens, model="word", bos=2, eos=3, unk=1, pad=0)
            o = TO.TokenizerOracle(TO.WORD_LEVEL, tokens, bos=2, eos=3, unk=1, pad=0)
            prompt, system, eos = b"W10 w20  w33 zzz", b"", 3
        else:
            pad = -len(sp["tokens"]) % 64  # unused entries up to a multiple of 64 rows for the embedding / output matrices
            tokens = sp["tokens"] + [b"[unused%d]" % i for i in range(pad)]
            scores, types = sp["scores"] + [-1e4 - i for i in range(pad)], sp["token_types"] + [1] * pad
            cfg["vocab"] = len(tokens)
            kvs = S.tokenizer_kvs(tokens, scores=scores, token_types=types, bos=1, eos=2, unk=0)
            o = TO.TokenizerOracle(TO.LLAMA_SENTENCEPIECE, tokens, scores=scores, token_types=types, bos=1, eos=2, unk=0)
            prompt, system, eos = b"what is the answer ?", b"be brief", 2
        _, tensors = S.build_model(cfg, "q4_k_m", seed=21)
        path = S.write_gguf(str(tmp_path / f"{family}.gguf"), cfg, tensors, extra_kvs=kvs)
        s = H.Session(path)
        om = OracleModel(oracle, cfg, tensors)
        want_by_qa = {}
        for qa in (False, True):
            ids = o.encode(o.format_prompt(prompt, system, qa), True, False)
            assert len(ids) >= 3
            want = om.generate(np.array(ids, np.int32), 12, penalty=1.3, eos_id=eos)
            want_by_qa[qa] = [int(i) for i in (want[0] if isinstance(want, tuple) else want)]
            assert s.generate_text(prompt, 12, temperature=0.0, system_prompt=system, apply_q_a_format=qa) == \
                o.decode(want_by_qa[qa], True), (family, qa)
        # generate_batch on text: every prompt's text equals its single run's (greedy, no penalty on this path, api.cpp:993)
        s.set_max_batch(3)
        s.set_repetition_penalty(1.0)
        batch_prompts = [prompt, prompt[:7], prompt + b" " + prompt]
        texts = s.generate_batch_text(batch_prompts, 6, temperature=0.0, system_prompt=system)
        for bp, text in zip(batch_prompts, texts):
            assert text == s.generate_text(bp, 6, temperature=0.0, system_prompt=system), (family, bp)
        with pytest.raises(H.TlHostError, match="exceeds maximum batch size"):
            s.generate_batch_text([prompt] * 4, 2)
        s.set_repetition_penalty(1.3)
        s.close()
        om.close()
        if family == "spm":
            exe = os.path.join(ROOT, "tinyllama.cpp_amd", "host", "tl_run")
            out = subprocess.run([exe, path, "--prompt", prompt.decode(), "--system", system.decode(), "--steps", "12"],
                                 capture_output=True, timeout=120)
            assert out.returncode == 0, out.stderr
            first, rest = out.stdout.split(b"\n", 1)
            assert json.loads(first)["generated"] == want_by_qa[False]
            assert rest[:-1] == o.decode(want_by_qa[False], True)
    # a file without tokenizer metadata: the text entry point refuses as the reference does (api.cpp:490-492)
    cfg, tensors = S.build_model("tiny", "q4_k_m", seed=13)
    s = H.Session(S.write_gguf(str(tmp_path / "plain.gguf"), cfg, tensors))
    with pytest.raises(H.TlHostError, match="Model or tokenizer not loaded"):
        s.generate_text(b"hello", 4)
    s.close()

