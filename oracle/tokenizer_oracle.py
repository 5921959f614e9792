"""oracle/tokenizer_oracle.py — TEST INFRASTRUCTURE, not product code (only tests/ may import it).

A Python restatement of the reference's tokenizer for GGUF models (tokenizer.cpp), written from the reference's source
independently of tinyllama.cpp_amd/host/tl_tokenizer.cpp: the two pre-tokenisation patterns are the reference's own
pattern strings handed to Python's regex engine (POSIX classes spelled out as their classic-locale ASCII sets, which
Python's `re` does not parse), where the C++ product code scans by hand.

PARITY UNPINNED. tokenizer.cpp includes Boost.Regex, Boost.Xpressive and nlohmann/json, none of which this image has, so
it cannot be compiled into oracle/_ref; the reference holds no tokenizer test vectors (its tests need model files that
are not in the repository). What the tests prove is that two independent readings of the same source agree; neither
reading has been run against the reference itself.

Text and tokens are `bytes` throughout (the reference works on std::string bytes).
"""
import heapq
import re

import numpy as np

UNKNOWN, LLAMA_SENTENCEPIECE, LLAMA3_TIKTOKEN, WORD_LEVEL = 0, 1, 2, 3
GPT2_SPACE = "Ġ".encode()  # "Ġ", BPE_SPACE_CHAR (tokenizer.cpp:26)
SP_SPACE = "▁".encode()  # "▁"

# tokenizer.cpp:67   R"([\r\n]+|[[:space:]]+|[^\r\n[:space:][:alnum:]]+|[[:alnum:]]+)"   (boost::regex, leftmost search)
SCORES_RE = re.compile(rb"[\r\n]+|[ \t\n\v\f\r]+|[^\r\n \t\n\v\f\r0-9A-Za-z]+|[0-9A-Za-z]+")
# tokenizer.cpp:2153 R"(<\|[^|]+\||[[:alnum:]]+|\.(?![<|])|[^\s<|]+|\s+)"                (xpressive, icase, match_continuous)
TIKTOKEN_RE = re.compile(rb"<\|[^|]+\||[0-9A-Za-z]+|\.(?![<|])|[^\s<|]+|\s+", re.IGNORECASE)

_UTF8_LEN = (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4)  # unicode_char_len, tokenizer.cpp:1133-1138


def _isspace(b):  # std::isspace in the classic locale, on one byte value
    return b == 0x20 or 0x09 <= b <= 0x0D


def _isprint(b):
    return 0x20 <= b <= 0x7E


def _hex_token(i):
    return b"<0x%02x>" % i


def _stoi16(two):
    """std::stoi(s, nullptr, 16) on a two-byte string; None where it throws."""
    m = re.match(rb"[ \t\n\v\f\r]*([+-]?)(?:0[xX](?=[0-9a-fA-F]))?([0-9a-fA-F]+)", two)
    if not m:
        return None
    v = int(m.group(2), 16)
    return -v if m.group(1) == b"-" else v


def family_from_metadata(model, arch, has_merges, vocab_size):
    """model_config.cpp:221-253."""
    if model == "word":
        return WORD_LEVEL
    if has_merges and vocab_size == 128256 and model == "gpt2":
        return LLAMA3_TIKTOKEN
    if arch == "llama" or "Llama-2" in arch or "TinyLlama" in arch:
        return LLAMA_SENTENCEPIECE
    return UNKNOWN


def _load_merges(lines):
    """tokenizer.cpp:624-636 / 689-700."""
    out, rank = {}, 0
    for m in lines:
        sp = m.find(b" ")
        if sp != -1 and 0 < sp < len(m) - 1:
            out[m[:sp] + m[sp + 1:]] = rank
            rank += 1
    return out


def _merges_from_scores(tokens, scores):
    """generate_bpe_merges_from_vocab_scores, tokenizer.cpp:422-514 (ties of equal scores: unspecified there, stable here)."""
    merges = {}
    if len(scores) == 0 or not tokens:
        return merges
    first = {}
    for i, t in enumerate(tokens):
        first.setdefault(t, i)  # std::find returns the first entry of that text
    scored = [(np.float32(scores[i]), t) for i, t in enumerate(tokens)
              if i < len(scores) and len(t) > 1 and b"<" not in t and b">" not in t and t != SP_SPACE]
    scored.sort(key=lambda st: -float(st[0]))
    rank, done = 0, set()
    for _, token in scored:
        if token in done:
            continue
        best, found = np.float32(-np.inf), False
        for split in range(1, len(token)):
            l, r = first.get(token[:split]), first.get(token[split:])
            if l is None or r is None:
                continue
            ls = np.float32(scores[l]) if l < len(scores) else np.float32(0)
            rs = np.float32(scores[r]) if r < len(scores) else np.float32(0)
            if np.float32(ls + rs) > best:
                best, found = np.float32(ls + rs), True
        if found and token not in merges:
            merges[token] = rank
            rank += 1
        done.add(token)
        if rank >= 50000:
            break
    return merges


class TokenizerOracle:
    def __init__(self, family, tokens, merges=(), scores=(), token_types=(), bos=-1, eos=-1, unk=-1, pad=-1,
                 chat_template=b""):
        """Tokenizer(const GGUFData&, const ModelConfig&), tokenizer.cpp:516-1048."""
        if not tokens:
            raise RuntimeError("GGUF data does not contain 'tokenizer.ggml.tokens'")
        self.family = family
        self.chat_template = chat_template
        self.id_to_token = list(tokens)
        self.token_to_id = {t: i for i, t in enumerate(tokens)}
        self.merges = {}
        if family == LLAMA3_TIKTOKEN:
            self.merges = _load_merges(merges)
        elif family == LLAMA_SENTENCEPIECE:
            self.merges = _load_merges(merges) if merges else _merges_from_scores(self.id_to_token, scores)
        self.byte_to_id, self.added, self.id_to_added = {}, {}, {}
        have_types = len(token_types) > 0 and len(token_types) == len(tokens)
        bytes_from_types = 0
        if have_types:
            for i, (tt, t) in enumerate(zip(token_types, tokens)):
                as_byte = False
                if tt == 6:
                    if len(t) == 1:
                        self.byte_to_id[t[0]] = i
                        as_byte = True
                    elif t.startswith(b"<0x") and t.endswith(b">") and len(t) == 6:
                        v = _stoi16(t[3:5])
                        if v is not None:
                            self.byte_to_id[v & 0xFF] = i
                            as_byte = True
                    bytes_from_types += as_byte
                if not as_byte and tt in (2, 3, 4, 5) and t not in self.added:
                    self.added[t] = i
                    self.id_to_added[i] = t
        if family == LLAMA3_TIKTOKEN and (not have_types or bytes_from_types == 0):  # :784-860, :865-938
            self.byte_to_id = {}
            for b in range(256):
                if _hex_token(b) in self.token_to_id:
                    self.byte_to_id[b] = self.token_to_id[_hex_token(b)]
                elif _isprint(b) and bytes([b]) in self.token_to_id:
                    tid = self.token_to_id[bytes([b])]
                    if tid not in self.byte_to_id.values():
                        self.byte_to_id[b] = tid
        if 0x20 not in self.byte_to_id:  # :942-990
            for b in range(256):
                if b in self.byte_to_id:
                    continue
                if _hex_token(b) in self.token_to_id:
                    self.byte_to_id[b] = self.token_to_id[_hex_token(b)]
                elif bytes([b]) in self.token_to_id:
                    self.byte_to_id[b] = self.token_to_id[bytes([b])]
            if 0x20 not in self.byte_to_id and SP_SPACE in self.token_to_id:
                self.byte_to_id[0x20] = self.token_to_id[SP_SPACE]
        self.bos_id, self.eos_id, self.pad_id = bos, eos, pad
        self.unk_id = 0 if unk < 0 else unk  # :1000-1003

        def setup(idv, default):  # setup_special_token, :1006-1026
            n = len(self.id_to_token)
            if 0 <= idv < n:
                return idv, self.id_to_token[idv]
            if default in self.token_to_id and (idv == -1 or idv >= n):
                idv = self.token_to_id[default]
            return idv, default

        self.bos_id, self.bos = setup(self.bos_id, b"<s>")
        self.eos_id, self.eos = setup(self.eos_id, b"</s>")
        self.unk_id, self.unk = setup(self.unk_id, b"<unk>")
        if pad == -1:
            self.pad = b""
        else:
            self.pad_id, self.pad = setup(self.pad_id, b"<pad>")

    # ---- encode, tokenizer.cpp:1141-1410 -------------------------------------------------------------------------------
    def encode(self, text, add_bos=True, add_eos=False):
        ids = []
        if self.family == WORD_LEVEL:
            for word in re.split(rb"[ \t\n\v\f\r]+", text):
                if word:
                    w = bytes(c + 32 if 0x41 <= c <= 0x5A else c for c in word)
                    ids.append(self.token_to_id.get(w, self.unk_id))
            return ids
        if self.family == LLAMA3_TIKTOKEN:
            if add_bos and self.bos_id != -1 and (not self.bos or not text.startswith(self.bos)):
                ids.append(self.bos_id)
            ids += self.bpe_tokenize_to_ids(text)
            if add_eos and self.eos_id != -1:
                ids.append(self.eos_id)
            return ids
        if self.family == LLAMA_SENTENCEPIECE:
            if add_bos and self.bos_id != -1:
                ids.append(self.bos_id)
            segments, cur = [], b""  # pre_tok_type_ stays "unknown" for GGUF files: the whitespace split of :1337-1353
            for c in text:
                if _isspace(c):
                    if cur:
                        segments.append(cur)
                        cur = b""
                    segments.append(bytes([c]))
                else:
                    cur += bytes([c])
            if cur:
                segments.append(cur)
            for seg in segments:
                ids += self.tokens_to_ids(self.bpe_tokenize_from_scores(seg))
            if add_eos and self.eos_id != -1:
                ids.append(self.eos_id)
            return ids
        if add_bos and self.bos_id != -1:
            ids.append(self.bos_id)
        if add_eos and self.eos_id != -1:
            ids.append(self.eos_id)
        return ids

    def bpe_tokenize_from_scores(self, text):
        """tokenizer.cpp:59-169."""
        units, rest = [], text
        while True:
            m = SCORES_RE.search(rest)
            if not m:
                break
            if m.group(0):
                units.append(m.group(0))
            rest = rest[m.end():]
        if rest:
            units.append(rest)
        out, prefix = [], True
        for unit in units:
            if all(_isspace(c) for c in unit):
                prefix = True
                continue
            word = SP_SPACE + unit if prefix else unit
            prefix = False
            chars, i = [], 0
            while i < len(word):
                n = _UTF8_LEN[word[i] >> 4]
                if i + n <= len(word):
                    chars.append(word[i:i + n])
                else:
                    chars.append(word[i:])
                    break
                i += n
            while len(chars) > 1:
                best_rank, best_i = None, -1
                for i in range(len(chars) - 1):
                    r = self.merges.get(chars[i] + chars[i + 1])
                    if r is not None and (best_rank is None or r < best_rank):
                        best_rank, best_i = r, i
                if best_i < 0:
                    break
                chars[best_i:best_i + 2] = [chars[best_i] + chars[best_i + 1]]
            out += chars
        return out

    def tokens_to_ids(self, pieces):
        """tokenizer.cpp:171-256."""
        ids = []
        for t in pieces:
            if t in self.added:
                ids.append(self.added[t])
            elif t in self.token_to_id:
                ids.append(self.token_to_id[t])
            else:
                cap = self._capitalize(t)
                if cap != t and cap in self.token_to_id:
                    ids.append(self.token_to_id[cap])
                elif len(t) == 1 and t[0] in self.byte_to_id:
                    ids.append(self.byte_to_id[t[0]])
                else:
                    ids.append(self.unk_id)
        return ids

    @staticmethod
    def _capitalize(s):  # capitalize_first_letter, tokenizer.cpp:1980-2008
        if not s:
            return s
        pos = 0
        if s.startswith(SP_SPACE):
            if len(s) > len(SP_SPACE):
                pos = len(SP_SPACE)
            else:
                return s
        c = s[pos]
        return s[:pos] + bytes([c - 32 if 0x61 <= c <= 0x7A else c]) + s[pos + 1:]

    def _bigram_rank(self, left, right):
        """add_bigram_to_queue_refactored, tokenizer.cpp:2398-2469: rank to queue the pair under, None if no attempt is a merge."""
        attempts = []
        if left == GPT2_SPACE:
            attempts.append(GPT2_SPACE + b" " + right)
        attempts.append(left + right)
        if left.startswith(GPT2_SPACE) and left != GPT2_SPACE:
            attempts.append(left + b" " + right)
        if len(left) == 2 and len(right) == 1:
            attempts.append(left[:1] + b" " + right)
        ranks = [self.merges[a] for a in attempts if a in self.merges]
        return min(ranks) if ranks else None

    def _byte_fallback(self, c):
        return self.byte_to_id.get(c, self.unk_id)

    def bpe_tokenize_to_ids(self, text):
        """tokenizer.cpp:2129-2394 with add_bos = add_eos = ignore_merges = false (what encode passes)."""
        out = []
        specials = sorted(self.added.items(), key=lambda kv: -len(kv[0]))
        cur = 0
        while cur < len(text):
            hit = next(((t, i) for t, i in specials if t and text.startswith(t, cur)), None)
            if hit:
                out.append(hit[1])
                cur += len(hit[0])
                continue
            m = TIKTOKEN_RE.match(text, cur)
            if not m:  # :2187-2210: the rest of the text, byte by byte
                for c in text[cur:]:
                    b = bytes([c])
                    out.append(self.token_to_id[b] if b in self.token_to_id else self._byte_fallback(c))
                break
            original = m.group(0)
            cur += len(original)
            if original in self.token_to_id:
                out.append(self.token_to_id[original])
                continue
            word = original
            if word[:1] == b" ":
                word = GPT2_SPACE + word[1:]
            # symbols: [prev, next, offset, n]
            sym, off, ok = [], 0, True
            while off < len(word):
                n = _UTF8_LEN[word[off] >> 4]
                if off + n > len(word):
                    ok = False
                    break
                sym.append([-1, -1, off, n])
                off += n
            if not ok or not sym:
                out.append(self.unk_id)
                continue
            for i in range(len(sym)):
                sym[i][0] = i - 1 if i > 0 else -1
                sym[i][1] = i + 1 if i < len(sym) - 1 else -1
            text_of = lambda s: word[s[2]:s[2] + s[3]]
            queue = []

            def add_bigram(first):
                if first < 0 or first >= len(sym):
                    return
                s1 = sym[first]
                second = s1[1]
                if second < 0 or second >= len(sym) or second <= first:
                    return
                s2 = sym[second]
                if s1[3] == 0 or s2[3] == 0:
                    return
                r = self._bigram_rank(text_of(s1), text_of(s2))
                if r is not None:
                    heapq.heappush(queue, (r, first))

            for i in range(len(sym) - 1):
                add_bigram(i)
            while queue:
                _, p1 = heapq.heappop(queue)
                if sym[p1][3] == 0:
                    continue
                p2 = sym[p1][1]
                if p2 == -1 or sym[p2][3] == 0:
                    continue
                sym[p1][3] += sym[p2][3]
                sym[p2][3] = 0
                sym[p1][1] = sym[p2][1]
                if sym[p1][1] != -1:
                    sym[sym[p1][1]][0] = p1
                if sym[p1][0] != -1:
                    add_bigram(sym[p1][0])
                if sym[p1][1] != -1:
                    add_bigram(p1)
            word_ids, i = [], 0
            while i != -1:
                s = sym[i]
                if s[3] != 0:
                    piece = text_of(s)
                    if piece in self.token_to_id:
                        word_ids.append(self.token_to_id[piece])
                    else:
                        word_ids += [self._byte_fallback(c) for c in piece]
                i = s[1]
            out += word_ids if word_ids else [self.unk_id]
        return out

    # ---- decode, tokenizer.cpp:1412-1652 --------------------------------------------------------------------------------
    def decode(self, ids, skip_special=True):
        n = len(self.id_to_token)
        core = (self.bos_id, self.eos_id, self.pad_id, self.unk_id)
        if self.family == WORD_LEVEL:
            return b" ".join(self.id_to_token[i] for i in ids if 0 <= i < n and not (skip_special and i in core))
        if self.family == LLAMA_SENTENCEPIECE:
            return self._decode_sentencepiece(ids, skip_special)
        out, first = b"", True
        for i in ids:
            if i < 0 or i >= n:
                if not skip_special:
                    out += b"[INVALID_ID:%d]" % i
                    first = False
                continue
            if skip_special and (i in core or i in self.id_to_added):
                continue
            t = self.id_to_token[i]
            if not t:
                if not skip_special and self.unk_id != -1:
                    t = self.unk
                else:
                    continue
            if t.startswith(GPT2_SPACE):
                if not first:
                    out += b" "
                out += t[len(GPT2_SPACE):]
            else:
                out += t
            first = False
        return out

    def _decode_sentencepiece(self, ids, skip_special):
        n = len(self.id_to_token)
        core = (self.bos_id, self.eos_id, self.pad_id, self.unk_id)
        out, first = b"", True
        for i in ids:
            special = False
            if skip_special and (i in core or i in self.id_to_added):
                continue
            if 0 <= i < n:
                t = self.id_to_token[i]
            elif i in self.id_to_added:
                t = self.id_to_added[i]
            else:
                t = b"" if skip_special else b"[INVALID_ID:%d]" % i
                special = True
            if len(t) == 6 and t.startswith(b"<0x") and t[5:6] == b">":
                v = _stoi16(t[3:5])
                if v is not None:
                    t = bytes([v & 0xFF])
            if not t and not special:
                if self.unk_id != -1:
                    if not skip_special or self.unk_id != i:
                        t = self.unk
                    else:
                        continue
                elif not skip_special:
                    t = b"[EMPTY_TOKEN_FOR_ID:%d]" % i
                else:
                    continue
            if t in (SP_SPACE, GPT2_SPACE):
                if first:
                    continue
                if not out or out[-1:] != b" ":
                    out += b" "
                first = False
                continue
            if t:
                prefix = SP_SPACE if t.startswith(SP_SPACE) else GPT2_SPACE if t.startswith(GPT2_SPACE) else None
                if prefix:
                    if not first and (not out or out[-1:] != b" "):
                        out += b" "
                    content = t[len(prefix):]
                    k = len(content) - len(content.lstrip(b" "))
                    if k < len(content):  # find_first_not_of(' ') != npos
                        content = content[k:]
                    out += content
                else:
                    out += t
                first = False
        return out

    # ---- templates ----------------------------------------------------------------------------------------------------------
    def apply_chat_template(self, user, system=b""):
        """tokenizer.cpp:1663-1741."""
        ct = self.chat_template
        jinja = bool(ct) and (b"{%" in ct or b"{{" in ct)
        if ct and not jinja:
            # (the reference's replace_all scans on after each replacement; str.replace does the same left to right)
            return (ct.replace(b"{{bos_token}}", self.bos if self.bos_id != -1 else b"")
                      .replace(b"{{eos_token}}", self.eos if self.eos_id != -1 else b"")
                      .replace(b"{{user_prompt}}", user).replace(b"{{system_message}}", system))
        bos = self.bos if self.bos_id != -1 else b"<s>"
        sh, eh, eot = b"<|start_header_id|>", b"<|end_header_id|>", b"<|eot_id|>"
        out = bos
        if system:
            out += sh + b"system" + eh + b"\n\n" + system + eot
        out += sh + b"user" + eh + b"\n\n" + user + eot
        out += sh + b"assistant" + eh + b"\n\n"
        return out

    def format_prompt(self, user, system=b"", apply_q_a_format=False):
        """api.cpp:505-572 (the "\\n" there are a backslash and an n, not a line break)."""
        bn = b"\\n"
        if self.family == WORD_LEVEL:
            return user
        if apply_q_a_format:
            return (system + bn + bn if system else b"") + b"Q: " + user + bn + b"A:"
        if self.chat_template:
            if self.family == LLAMA_SENTENCEPIECE and b"{%" in self.chat_template:
                bbn = b"\\\\n"
                return (system + bbn + bbn if system else b"") + b"Q: " + user + bbn + b"A:"
            return self.apply_chat_template(user, system)
        if self.family == LLAMA3_TIKTOKEN:
            return self.apply_chat_template(user, system)
        return system + bn + bn + user if system else user
