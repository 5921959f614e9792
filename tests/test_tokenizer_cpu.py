"""the C++ tokenizer (host/tl_tokenizer.cpp over include/tlhost.h) against oracle/tokenizer_oracle.py.

PARITY UNPINNED: the reference's tokenizer.cpp cannot be built here (Boost, nlohmann/json) and the reference holds no
tokenizer vectors, so nothing in this file has been compared with the reference running. Two kinds of test:
  * hand-derived cases: expected ids / text worked out by reading tokenizer.cpp for one small vocabulary, written as
    literals (they pin the quirks: where "Ġ" does and does not appear, the capitalised fallback, byte tokens, UNK);
  * agreement of the C++ code with the independent Python restatement on a few thousand seeded strings, for every
    family, through files written by synth.write_metadata_gguf (so the GGUF arrays, the family detection and the
    special-id metadata are on the path too).
No GPU calls."""
import collections

import numpy as np
import pytest

from oracle import tokenizer_oracle as TO

GP, SP = TO.GPT2_SPACE, TO.SP_SPACE

CORPUS = (b"the quick brown fox jumps over the lazy dog . the dog barks , the fox runs away ! "
          b"hello world , hello there . what is the answer to the question ? the answer is 42 . "
          b"tokenizers split text into pieces ; pieces become ids . ids become text again . "
          b"a model reads ids and writes ids , nothing else . The Capital of France is Paris .")


@pytest.fixture(scope="module")
def H(pkg):
    from tinyllama_cpp_amd import hostapi
    hostapi.lib()
    return hostapi


def train_bpe(words, n_merges):
    """A toy BPE trainer: (merges as (left, right) in order, the set of symbols). Words are lists of byte-string symbols."""
    words = [list(w) for w in words]
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w in words:
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += 1
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        if pairs[best] < 2:
            break
        merges.append(best)
        for w in words:
            i = 0
            while i + 1 < len(w):
                if (w[i], w[i + 1]) == best:
                    w[i:i + 2] = [w[i] + w[i + 1]]
                else:
                    i += 1
    return merges


def utf8_chars(s):
    return [c.encode() for c in s]


# ---- vocabularies ----------------------------------------------------------------------------------------------------------
def spm_parts(with_merges=False, chat_template=None, byte_hex_upper=True):
    """A TinyLlama-shaped vocabulary: <unk> <s> </s>, 256 byte tokens (type 6), then pieces with distinct scores."""
    words = [utf8_chars(SP.decode() + w.decode()) for w in CORPUS.split()]
    merges = train_bpe(words, 120)
    tokens = [b"<unk>", b"<s>", b"</s>"] + [(b"<0x%02X>" if byte_hex_upper else b"<0x%02x>") % i for i in range(256)]
    types = [2, 3, 3] + [6] * 256
    pieces = [SP] + sorted({c for w in words for c in w} - {SP}) + [a + b for a, b in merges]
    pieces += ["é".encode(), "日".encode(), SP + b"World", b"\n"]  # "▁World" with no "▁world": the capitalised fallback
    seen = set(tokens)
    for p in pieces:
        if p not in seen:
            seen.add(p)
            tokens.append(p)
            types.append(1)
    scores = [0.0] * 259 + [-(i + 1) * 0.5 for i in range(len(tokens) - 259)]  # earlier merge = higher score, all distinct
    parts = dict(family=TO.LLAMA_SENTENCEPIECE, tokens=tokens, scores=scores, token_types=types, bos=1, eos=2, unk=0)
    if with_merges:
        parts["merges"] = [a + b" " + b for a, b in merges]
    if chat_template is not None:
        parts["chat_template"] = chat_template
    return parts


SPECIALS = [b"<|begin_of_text|>", b"<|end_of_text|>", b"<|start_header_id|>", b"<|end_header_id|>", b"<|eot_id|>", b"<|x|>",
            b"<|eot_id|>extra"]


def tiktoken_parts(odd_merges=False, literal_space=False, byte_tokens=False, chat_template=None):
    """A Llama-3-shaped vocabulary: printable ASCII as single-byte tokens, "Ġ", trained merges, specials of type 3."""
    words = []
    for i, w in enumerate(CORPUS.split()):
        words.append(utf8_chars((GP.decode() if i else "") + w.decode()))
    merges = train_bpe(words, 150)
    tokens = [bytes([c]) for c in range(0x21, 0x7F)] + [GP, "é".encode(), b"\n", b"\n\n"]
    if literal_space:
        tokens.append(b" ")
    if byte_tokens:
        tokens += [b"<0x%02x>" % i for i in (0x09, 0x0a, 0x20, 0xc3, 0xe6)]
    for t in [a + b for a, b in merges] + [b"hello", b"world", GP + b"hello"]:  # (whole words match without merges, :2234)
        if t not in tokens:
            tokens.append(t)
    lines = [a + b" " + b for a, b in merges]
    if odd_merges:
        # keys with a blank inside (the second to fourth lookups of add_bigram_to_queue_refactored), malformed lines that
        # take no rank, and a repeated concatenation whose later rank wins
        lines = [b"nospace", b" leading", b"trailing ", GP + b"  q", GP + b"t  h", b"a  b", b"t h"] + lines + [b"th e", b"t he"]
        tokens += [GP + b"q", GP + b"th", b"ab"]
    types = [1] * len(tokens)
    for s in SPECIALS:
        tokens.append(s)
        types.append(3)
    bos, eos = tokens.index(b"<|begin_of_text|>"), tokens.index(b"<|end_of_text|>")
    parts = dict(family=TO.LLAMA3_TIKTOKEN, tokens=tokens, merges=lines, token_types=types, bos=bos, eos=eos)
    if chat_template is not None:
        parts["chat_template"] = chat_template
    return parts


def word_parts():
    tokens = [b"<pad>", b"<unk>", b"<s>", b"</s>"] + sorted(set(CORPUS.lower().split()))
    return dict(family=TO.WORD_LEVEL, tokens=tokens, bos=2, eos=3, unk=1, pad=0)


def both(H, pkg, tmp_path, parts, override=True, name="t.gguf"):
    """(C++ tokenizer opened from a GGUF file holding `parts`, oracle built from the same parts)."""
    S = pkg.synth
    p = dict(parts)
    fam = p.pop("family")
    model = {TO.WORD_LEVEL: "word", TO.LLAMA3_TIKTOKEN: "gpt2"}.get(fam, "llama")
    kvs = S.tokenizer_kvs(p["tokens"], model=model, merges=p.get("merges"), scores=p.get("scores"),
                          token_types=p.get("token_types"), bos=p.get("bos"), eos=p.get("eos"), unk=p.get("unk"),
                          pad=p.get("pad"), chat_template=p.get("chat_template"),
                          vocab_size=None if override else 128256)
    path = S.write_metadata_gguf(str(tmp_path / name), kvs)
    t = H.Tokenizer(path, family=fam if override else 0)
    o = TO.TokenizerOracle(fam, p["tokens"], merges=p.get("merges", ()), scores=p.get("scores", ()),
                           token_types=p.get("token_types", ()), bos=p.get("bos", -1), eos=p.get("eos", -1),
                           unk=p.get("unk", -1), pad=p.get("pad", -1), chat_template=p.get("chat_template", b""))
    return t, o


ALPHABET = ([bytes([c]) for c in b"abcdefghilnoqrstuwxyTHCFP"] * 3 + [b" "] * 14 + [b"the", b"hello", b"world", b"answer"] * 2 +
            [b"\n", b"\t", b"\r\n", b"  ", b"\n\n", b".", b",", b"!", b"?", b";", b"4", b"2", b"<", b"|", b">", b"<|", b"|>",
             b"<|eot_id|>", b"<|x|>", b"<|nope|>", b"<|eot_id|>extra", b".<", b".|", b"<s>", b"</s>", b"<unk>", "é".encode(),
             "日".encode(), "Ġ".encode(), SP, b"\xc3", b"\xe6\x97", b"\xff", b"\x00", b"\x0b", b"_", b"-", b"'s", b"{", b"0x"])


def random_texts(seed, n, max_len=24):
    rng = np.random.default_rng(seed)
    out = [b"", b" ", b"  ", b"\n", b"<", b"|", b"<|", b".", b"hello world", b" hello  world ", b"Hello World", b"a<b", b"x|y",
           b"the answer is 42.", b"what is <|eot_id|> ? <|nope|> .<|x|>", b"tab\there\r\nnew", "café 日本".encode(),
           b"trunc\xe6\x97", CORPUS[:200], GP + b"q", GP + b"the", b"anb", GP + b" q", b"thq " + GP + b"th" + GP,
           SP + b"the" + SP, b"q" * 40]
    for _ in range(n):
        k = int(rng.integers(1, max_len))
        out.append(b"".join(ALPHABET[i] for i in rng.integers(0, len(ALPHABET), k)))
    return out


def check_agreement(t, o, texts, seed):
    rng = np.random.default_rng(seed)
    info = t.info()
    assert (info["family"], info["vocab"]) == (o.family, len(o.id_to_token))
    assert (info["bos"], info["eos"], info["unk"], info["pad"]) == (o.bos_id, o.eos_id, o.unk_id, o.pad_id)
    for text in texts:
        want = o.encode(text, True, False)
        got = t.encode(text, True, False)
        assert got == want, (text, got, want)
        assert t.encode(text, False, True) == o.encode(text, False, True), text
        for skip in (True, False):
            assert t.decode(want, skip) == o.decode(want, skip), (text, want, skip)
    V = len(o.id_to_token)
    for _ in range(200):
        ids = [int(i) for i in rng.integers(-2, V + 2, int(rng.integers(0, 12)))]
        for skip in (True, False):
            assert t.decode(ids, skip) == o.decode(ids, skip), (ids, skip)
    for user, system, qa in ((b"hi there", b"", False), (b"hi", b"be brief", False), (b"hi", b"", True), (b"hi", b"sys", True)):
        assert t.apply_chat_template(user, system) == o.apply_chat_template(user, system)
        assert t.format_prompt(user, system, qa) == o.format_prompt(user, system, qa)


# ---- hand-derived cases ------------------------------------------------------------------------------------------------------
def test_sentencepiece_hand_derived(H, pkg, tmp_path):
    parts = spm_parts()
    t, o = both(H, pkg, tmp_path, parts)
    tok = {x: i for i, x in enumerate(parts["tokens"])}
    hello, world_cap, the = tok[SP + b"hello"], tok[SP + b"World"], tok[SP + b"the"]
    # every whitespace-separated word is a segment of its own and gets the "▁" prefix (tokenizer.cpp:1337-1353, :85-111)
    assert t.encode(b"hello the") == [1, hello, the]
    assert t.encode(b"  hello \n the ", add_bos=False, add_eos=True) == [hello, the, 2]
    # "▁world" is no vocabulary entry; its pieces merge as far as the generated merges go. The capitalised fallback
    # (:219-231) needs the WHOLE piece to be missing: build it from a piece that cannot be reached by merges
    ids = t.encode(b"hello world", add_bos=False)
    assert ids[0] == hello and t.decode(ids) == b"hello world"
    assert t.encode(b"") == [1] and t.encode(b"", add_bos=False) == []
    # a byte outside the vocabulary's pieces: its own unit of one byte -> the type-6 byte token (:233-244)
    assert t.encode(b"~", add_bos=False) == [tok[SP], 3 + ord("~")] or t.encode(b"~", add_bos=False) == o.encode(b"~", False)
    # a multi-byte character without a piece is UNK (no byte fallback for pieces longer than one byte, :246-251)
    assert t.encode("ж".encode(), add_bos=False)[-1] == 0
    # decode: <0xNN> becomes the byte, "▁" a blank, the leading blank goes, specials go when asked (:1511-1652)
    assert t.decode([1, hello, world_cap, 3 + 0x21, 2]) == b"hello World!"
    assert t.decode([1, hello, 2], skip_special_tokens=False) == b"<s> hello</s>"
    assert t.decode([tok[SP], tok[SP], hello, tok[SP], tok[SP], the]) == b"hello the"
    assert t.decode([hello, 10 ** 6], skip_special_tokens=False) == b"hello[INVALID_ID:1000000]"
    assert t.decode([hello, 10 ** 6]) == b"hello"


def test_sentencepiece_capitalised_fallback(H, pkg, tmp_path):
    # three-entry vocabulary beyond the specials: "▁Zed" only. "zed" splits into ▁ z e d (no merges), none of which
    # exists -> UNK each; the piece "▁Zed" is reachable only if a merge chain exists, so give explicit merges
    tokens = [b"<unk>", b"<s>", b"</s>", SP, b"z", b"e", b"d", SP + b"Zed", b"Q"]
    merges = [SP + b" z", SP + b"z e", SP + b"ze d"]
    parts = dict(family=TO.LLAMA_SENTENCEPIECE, tokens=tokens, merges=merges, scores=[0.0] * len(tokens), bos=1, eos=2, unk=0)
    t, o = both(H, pkg, tmp_path, parts)
    assert t.encode(b"zed", add_bos=False) == [7] == o.encode(b"zed", False)  # "▁zed" missing, "▁Zed" found (:219-231)
    assert t.encode(b"q", add_bos=False) == o.encode(b"q", False) == [3, 8]      # "q" -> "Q"
    assert t.encode(b"Zed", add_bos=False) == [3, 0, 5, 6]                        # "Z" is neither a piece nor a byte token


def test_tiktoken_hand_derived(H, pkg, tmp_path):
    parts = tiktoken_parts()
    t, o = both(H, pkg, tmp_path, parts)
    tok = {x: i for i, x in enumerate(parts["tokens"])}
    bos, eos = tok[b"<|begin_of_text|>"], tok[b"<|end_of_text|>"]
    assert b"hello" in tok and b"world" in tok and GP + b"hello" in tok
    # the blank between two words is a word of its own (\s+), looked up as "Ġ": "world" never gets the prefix (:2152-2254)
    assert t.encode(b"hello world") == [bos, tok[b"hello"], tok[GP], tok[b"world"]]
    assert t.encode(b"hello world", add_bos=False, add_eos=True) == [tok[b"hello"], tok[GP], tok[b"world"], eos]
    # the text already starts with the BOS string: no second BOS (:1183-1192); specials match longest first (:2164-2176)
    assert t.encode(b"<|begin_of_text|>hello") == [bos, tok[b"hello"]]
    assert t.encode(b"<|eot_id|>extra<|eot_id|>", add_bos=False) == [tok[b"<|eot_id|>extra"], tok[b"<|eot_id|>"]]
    # a '<' that opens no <|...|>: everything from there on goes byte by byte (:2187-2210)
    assert t.encode(b"hello<the", add_bos=False) == [tok[b"hello"], tok[b"<"], tok[b"t"], tok[b"h"], tok[b"e"]]
    # "<|nope|>" matches the first alternative, is no token, and goes through BPE as a word
    assert t.encode(b"<|nope|>", add_bos=False) == o.encode(b"<|nope|>", False)
    # decode: "Ġ" at the front of a token is a blank unless it is the first thing written (:1490-1504)
    assert t.decode([bos, tok[GP + b"hello"], tok[GP], tok[b"world"], eos]) == b"hello world"
    assert t.decode([bos, tok[b"hello"]], skip_special_tokens=False) == b"<|begin_of_text|>hello"
    assert t.decode([tok[b"hello"], -5], skip_special_tokens=False) == b"hello[INVALID_ID:-5]"
    # the hard-coded Llama-3 layout when the file has no template (:1707-1739)
    assert t.apply_chat_template(b"hi", b"") == (b"<|begin_of_text|><|start_header_id|>user<|end_header_id|>\n\nhi<|eot_id|>"
                                                 b"<|start_header_id|>assistant<|end_header_id|>\n\n")
    assert t.format_prompt(b"hi", b"", False) == t.apply_chat_template(b"hi", b"")
    assert t.format_prompt(b"hi", b"sys", True) == b"sys\\n\\nQ: hi\\nA:"  # backslash-n, not line breaks (api.cpp:520-527)


def test_word_level_hand_derived(H, pkg, tmp_path):
    parts = word_parts()
    t, o = both(H, pkg, tmp_path, parts)
    tok = {x: i for i, x in enumerate(parts["tokens"])}
    assert t.encode(b"The  QUICK\tfox zzz") == [tok[b"the"], tok[b"quick"], tok[b"fox"], 1]  # lower case, no BOS, UNK = 1
    assert t.decode([2, tok[b"the"], 1, tok[b"fox"], 3, 0]) == b"the fox"
    assert t.decode([2, tok[b"the"], 99999], skip_special_tokens=False) == b"<s> the"
    assert t.format_prompt(b"hi", b"sys", True) == b"hi"  # api.cpp:513-517


# ---- agreement with the Python restatement --------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["scores", "merges", "lower_hex", "jinja", "plain_template"])
def test_sentencepiece_agrees_with_restatement(H, pkg, tmp_path, variant):
    parts = spm_parts(with_merges=variant == "merges", byte_hex_upper=variant != "lower_hex",
                      chat_template={"jinja": b"{% for m in messages %}{{ m.content }}{% endfor %}",
                                     "plain_template": b"[INST] user_prompt [/INST]"}.get(variant))
    t, o = both(H, pkg, tmp_path, parts)
    assert o.merges, "the vocabulary must yield merges"
    check_agreement(t, o, random_texts(11, 600), seed=1)


@pytest.mark.parametrize("variant", ["plain", "odd_merges", "literal_space", "byte_tokens", "detected", "template"])
def test_tiktoken_agrees_with_restatement(H, pkg, tmp_path, variant):
    parts = tiktoken_parts(odd_merges=variant == "odd_merges", literal_space=variant == "literal_space",
                           byte_tokens=variant == "byte_tokens",
                           chat_template=b"{{bos_token}}{{user_prompt}}" if variant == "template" else None)
    # "detected": no family override; the file says gpt2 + merges + vocab_size 128256 (model_config.cpp:232-236)
    t, o = both(H, pkg, tmp_path, parts, override=variant != "detected")
    check_agreement(t, o, random_texts(12, 600), seed=2)


def test_word_level_and_unknown_family_agree(H, pkg, tmp_path):
    t, o = both(H, pkg, tmp_path, word_parts())
    check_agreement(t, o, random_texts(13, 200), seed=3)
    parts = dict(spm_parts(), family=TO.UNKNOWN)
    S = pkg.synth
    kvs = S.tokenizer_kvs(parts["tokens"], model="other", scores=parts["scores"], token_types=parts["token_types"], bos=1, eos=2)
    path = S.write_metadata_gguf(str(tmp_path / "u.gguf"), kvs, architecture="other")
    t = H.Tokenizer(path)  # arch "other", model "other": UNKNOWN (model_config.cpp:246-252) -> BOS / EOS only (:1392-1402)
    assert t.info()["family"] == TO.UNKNOWN
    assert t.encode(b"hello") == [1] and t.encode(b"hello", add_eos=True) == [1, 2] and t.encode(b"x", add_bos=False) == []


def test_family_detection_and_special_ids(H, pkg, tmp_path):
    S = pkg.synth
    sp = spm_parts()
    # TinyLlama-style file: arch llama, model llama -> SentencePiece family; ids missing from the file are found by their
    # default strings (:1006-1030), UNK defaults to 0 (:1000-1003), no padding id -> -1
    path = S.write_metadata_gguf(str(tmp_path / "a.gguf"), S.tokenizer_kvs(sp["tokens"], scores=sp["scores"],
                                                                           token_types=sp["token_types"]))
    t = H.Tokenizer(path)
    assert t.info() == dict(family=TO.LLAMA_SENTENCEPIECE, vocab=len(sp["tokens"]), bos=1, eos=2, unk=0, pad=-1)
    # gpt2 + merges but an ordinary vocabulary size: NOT Llama-3 (model_config.cpp:232), falls to the architecture test
    tk = tiktoken_parts()
    path = S.write_metadata_gguf(str(tmp_path / "b.gguf"), S.tokenizer_kvs(tk["tokens"], model="gpt2", merges=tk["merges"]))
    assert H.Tokenizer(path).info()["family"] == TO.LLAMA_SENTENCEPIECE
    # an out-of-range BOS id falls back to "<s>" by name; a padding id out of range with no "<pad>" entry stays as it is
    path = S.write_metadata_gguf(str(tmp_path / "c.gguf"), S.tokenizer_kvs(sp["tokens"], scores=sp["scores"], bos=10 ** 6, pad=10 ** 6))
    info = H.Tokenizer(path).info()
    assert info["bos"] == 1 and info["pad"] == 10 ** 6
    # a file without tokens is refused with the reference's message (:560-563)
    path = S.write_metadata_gguf(str(tmp_path / "d.gguf"), [S._kv_str("tokenizer.ggml.model", "llama")])
    with pytest.raises(H.TlHostError, match="does not contain 'tokenizer.ggml.tokens'"):
        H.Tokenizer(path)


def test_round_trips_on_plain_text(H, pkg, tmp_path):
    """decode(encode(x)) gives x back for single-blank-separated lower-case text of vocabulary words (the property the
    reference's generate() relies on when it echoes text)."""
    for parts in (spm_parts(), word_parts()):
        t, _ = both(H, pkg, tmp_path, parts)
        for text in (b"the quick brown fox", b"hello there", b"ids become text again ."):
            assert t.decode(t.encode(text)) == text
    t, _ = both(H, pkg, tmp_path, tiktoken_parts())
    for text in (b"the quick brown fox", b"hello there"):
        assert t.decode(t.encode(text)) == text
