// tl_tokenizer.cpp — see tl_tokenizer.hpp. Every function names the reference lines it restates.
#include "tl_tokenizer.hpp"

#include <algorithm>
#include <cstdio>
#include <limits>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <unordered_set>

namespace tlhost {
namespace {

const std::string kGpt2Space = "\xC4\xA0";     // "Ġ" (tokenizer.cpp:26)
const std::string kSpSpace = "\xE2\x96\x81";   // "▁" U+2581

// classic-locale character classes (what [[:space:]], \s, [[:alnum:]], std::isspace / isalnum / isprint give there)
inline bool c_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
inline bool c_alnum(unsigned char c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
inline bool c_print(unsigned char c) { return c >= 0x20 && c <= 0x7E; }

// length of a UTF-8 character from its first byte (tokenizer.cpp:1133-1138)
inline size_t utf8_len(char c) {
  static const size_t lookup[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
  return lookup[static_cast<unsigned char>(c) >> 4];
}

std::string hex_byte_token(int i) {
  char buf[8];
  snprintf(buf, sizeof(buf), "<0x%02x>", i);
  return buf;
}

bool parse_hex_byte_token(const std::string& t, int* value) {  // "<0xNN>" (std::stoi(.., 16) of the two characters)
  if (t.size() != 6 || t.compare(0, 3, "<0x") != 0 || t[5] != '>') return false;
  int v = 0, digits = 0;
  size_t i = 3;
  // std::stoi skips leading whitespace and accepts a sign; two characters leave little room, keep to what can parse
  bool neg = false;
  while (i < 5 && c_space((unsigned char)t[i])) ++i;
  if (i < 5 && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
  for (; i < 5; ++i) {
    const char c = t[i];
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
    else break;
    v = v * 16 + d;
    ++digits;
  }
  if (digits == 0) return false;  // std::stoi throws, the reference keeps the token as it is
  *value = neg ? -v : v;
  return true;
}

std::string replace_all(std::string s, const std::string& from, const std::string& to) {  // tokenizer.cpp:1654-1661
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.size(), to);
    pos += to.size();
  }
  return s;
}

// merges "left right" -> rank by position, keyed by the concatenation; a malformed line is skipped without using up a
// rank, a repeated concatenation keeps the LATER rank (tokenizer.cpp:624-636, 689-700)
void load_merges(const std::vector<std::string>& lines, std::unordered_map<std::string, int>& out) {
  out.clear();
  int rank = 0;
  for (const std::string& m : lines) {
    const size_t sp = m.find(' ');
    if (sp != std::string::npos && sp > 0 && sp < m.size() - 1) out[m.substr(0, sp) + m.substr(sp + 1)] = rank++;
  }
}

// SentencePiece files carry no merges: ranks are made up from the scores (tokenizer.cpp:422-514): every token of more than
// one byte without '<' / '>' that splits into two vocabulary entries becomes a merge, best score first. (The reference
// looks the halves up with a linear std::find, i.e. the FIRST entry of that text; a map of first entries does the same.)
std::unordered_map<std::string, int> merges_from_scores(const std::vector<std::string>& tokens, const std::vector<float>& scores) {
  std::unordered_map<std::string, int> merges;
  if (scores.empty() || tokens.empty()) return merges;
  std::unordered_map<std::string, size_t> first;
  for (size_t i = 0; i < tokens.size(); ++i) first.emplace(tokens[i], i);
  std::vector<std::pair<float, std::string>> scored;
  for (size_t id = 0; id < tokens.size() && id < scores.size(); ++id) {
    const std::string& t = tokens[id];
    if (t.size() > 1 && t.find('<') == std::string::npos && t.find('>') == std::string::npos && t != kSpSpace)
      scored.emplace_back(scores[id], t);
  }
  // (same call as the reference: among equal scores the order is whatever std::sort leaves)
  std::sort(scored.begin(), scored.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  int rank = 0;
  std::unordered_set<std::string> seen;
  for (const auto& st : scored) {
    const std::string& token = st.second;
    if (seen.count(token)) continue;
    bool found = false;
    float best = -std::numeric_limits<float>::infinity();
    for (size_t split = 1; split < token.size(); ++split) {
      const auto l = first.find(token.substr(0, split)), r = first.find(token.substr(split));
      if (l == first.end() || r == first.end()) continue;
      const float ls = l->second < scores.size() ? scores[l->second] : 0.0f;
      const float rs = r->second < scores.size() ? scores[r->second] : 0.0f;
      if (ls + rs > best) {
        best = ls + rs;
        found = true;
      }
    }
    if (found && !merges.count(token)) merges[token] = rank++;
    seen.insert(token);
    if (rank >= 50000) break;
  }
  return merges;
}

}  // namespace

namespace {
std::string meta_string(const GGUFFile& f, const char* key, const std::string& dflt) {
  const GGUFValue* v = f.meta(key);
  return v && v->kind == GGUFValue::STRING ? v->s : dflt;
}
int meta_integer(const GGUFFile& f, const char* key, int dflt) {
  const GGUFValue* v = f.meta(key);
  int64_t x = 0;
  return v && v->as_integer(&x) ? (int)x : dflt;
}
}  // namespace

TokenizerFamily tokenizer_family_from_gguf(const GGUFFile& f) {
  const std::string model = meta_string(f, "tokenizer.ggml.model", "");
  const std::string arch = meta_string(f, "general.architecture", "unknown");
  const GGUFValue* merges = f.meta("tokenizer.ggml.merges");
  const bool has_merges = merges && !merges->strings.empty();
  const int vocab = meta_integer(f, "tokenizer.ggml.vocab_size", meta_integer(f, "llama.vocab_size", 32000));
  if (model == "word") return TokenizerFamily::WORD_LEVEL;
  if (has_merges && vocab == 128256 && model == "gpt2") return TokenizerFamily::LLAMA3_TIKTOKEN;
  if (arch == "llama" || arch.find("Llama-2") != std::string::npos || arch.find("TinyLlama") != std::string::npos)
    return TokenizerFamily::LLAMA_SENTENCEPIECE;
  return TokenizerFamily::UNKNOWN;
}

Tokenizer::Tokenizer(const GGUFFile& f, TokenizerFamily family_override) {
  Parts p;
  p.family = family_override != TokenizerFamily::UNKNOWN ? family_override : tokenizer_family_from_gguf(f);
  if (const GGUFValue* v = f.meta("tokenizer.ggml.tokens")) p.tokens = v->strings;
  if (const GGUFValue* v = f.meta("tokenizer.ggml.merges")) p.merges = v->strings;
  if (const GGUFValue* v = f.meta("tokenizer.ggml.scores")) p.scores = v->floats;
  if (const GGUFValue* v = f.meta("tokenizer.ggml.token_type")) p.token_types = v->ints;
  p.bos = meta_integer(f, "tokenizer.ggml.bos_token_id", -1);  // model_config.cpp:201-204
  p.eos = meta_integer(f, "tokenizer.ggml.eos_token_id", -1);
  p.unk = meta_integer(f, "tokenizer.ggml.unk_token_id", -1);
  p.pad = meta_integer(f, "tokenizer.ggml.padding_token_id", -1);
  p.chat_template = meta_string(f, "tokenizer.chat_template", "");
  init(p);
}

Tokenizer::Tokenizer(const Parts& p) { init(p); }

// tokenizer.cpp:516-1048
void Tokenizer::init(const Parts& p) {
  family_ = p.family;
  chat_template_ = p.chat_template;
  if (p.tokens.empty()) throw std::runtime_error("GGUF data does not contain 'tokenizer.ggml.tokens'");  // :560-563
  id_to_token_ = p.tokens;
  token_to_id_.reserve(id_to_token_.size());
  for (size_t i = 0; i < id_to_token_.size(); ++i) token_to_id_[id_to_token_[i]] = (int)i;  // a repeated text keeps the LAST id

  if (family_ == TokenizerFamily::LLAMA3_TIKTOKEN) {
    load_merges(p.merges, merges_);
  } else if (family_ == TokenizerFamily::LLAMA_SENTENCEPIECE) {
    if (!p.merges.empty()) load_merges(p.merges, merges_);
    else merges_ = merges_from_scores(id_to_token_, p.scores);
  }

  // byte tokens and added / special tokens from the token types (:721-773): 6 = byte, 2..5 = control / user-defined / ...
  const bool have_types = !p.token_types.empty() && p.token_types.size() == id_to_token_.size();
  int bytes_from_types = 0;
  if (have_types) {
    for (size_t i = 0; i < p.token_types.size(); ++i) {
      const int tt = p.token_types[i];
      const std::string& t = id_to_token_[i];
      bool as_byte = false;
      if (tt == 6) {
        int v = 0;
        if (t.size() == 1) {
          byte_to_id_[t[0]] = (int)i;
          as_byte = true;
        } else if (t.size() == 6 && t.compare(0, 3, "<0x") == 0 && t.back() == '>' && parse_hex_byte_token(t, &v)) {
          byte_to_id_[static_cast<char>(v)] = (int)i;
          as_byte = true;
        }
        if (as_byte) ++bytes_from_types;
      }
      if (!as_byte && tt >= 2 && tt <= 5 && !added_tokens_.count(t)) {
        added_tokens_[t] = (int)i;
        id_to_added_[(int)i] = t;
      }
    }
  }
  // Llama-3 files mark no token as a byte: scan the vocabulary for <0xNN>, else a printable literal whose id is not
  // taken yet (:778-825 with types, :827-874 without)
  if (family_ == TokenizerFamily::LLAMA3_TIKTOKEN && (!have_types || bytes_from_types == 0)) {
    byte_to_id_.clear();
    for (int i = 0; i < 256; ++i) {
      const auto hex = token_to_id_.find(hex_byte_token(i));
      if (hex != token_to_id_.end()) {
        byte_to_id_[static_cast<char>(i)] = hex->second;
      } else if (c_print((unsigned char)i)) {
        const auto lit = token_to_id_.find(std::string(1, static_cast<char>(i)));
        if (lit == token_to_id_.end()) continue;
        bool taken = false;
        for (const auto& kv : byte_to_id_) taken |= kv.second == lit->second;
        if (!taken) byte_to_id_[static_cast<char>(i)] = lit->second;
      }
    }
  }
  // no byte token for the space yet: fill every byte still missing from <0xNN> or the literal, and as a last resort map
  // ' ' to the id of "▁" (:876-920)
  if (!byte_to_id_.count(' ')) {
    for (int i = 0; i < 256; ++i) {
      const char c = static_cast<char>(i);
      if (byte_to_id_.count(c)) continue;
      const auto hex = token_to_id_.find(hex_byte_token(i));
      if (hex != token_to_id_.end()) {
        byte_to_id_[c] = hex->second;
      } else {
        const auto lit = token_to_id_.find(std::string(1, c));
        if (lit != token_to_id_.end()) byte_to_id_[c] = lit->second;
      }
    }
    if (!byte_to_id_.count(' ')) {
      const auto sp = token_to_id_.find(kSpSpace);
      if (sp != token_to_id_.end()) byte_to_id_[' '] = sp->second;
    }
  }

  // special ids and their strings (:922-972)
  bos_id_ = p.bos;
  eos_id_ = p.eos;
  unk_id_ = p.unk < 0 ? 0 : p.unk;
  pad_id_ = p.pad;
  auto setup = [&](int& id, std::string& text, const std::string& dflt) {
    if (id >= 0 && (size_t)id < id_to_token_.size()) {
      text = id_to_token_[id];
      return;
    }
    text = dflt;
    const auto it = token_to_id_.find(dflt);
    if (it != token_to_id_.end() && (id == -1 || id >= 0)) id = it->second;  // (-1 or beyond the vocabulary; an id below -1 stays)
  };
  setup(bos_id_, bos_, "<s>");
  setup(eos_id_, eos_, "</s>");
  setup(unk_id_, unk_, "<unk>");
  if (p.pad == -1) pad_.clear();
  else setup(pad_id_, pad_, "<pad>");

  added_by_length_.assign(added_tokens_.begin(), added_tokens_.end());
  std::sort(added_by_length_.begin(), added_by_length_.end(),
            [](const auto& a, const auto& b) { return a.first.size() > b.first.size(); });
}

// ---- encode (tokenizer.cpp:1141-1410) ----------------------------------------------------------------------------------
std::vector<int> Tokenizer::encode(const std::string& text, bool add_bos, bool add_eos) const {
  std::vector<int> ids;
  if (family_ == TokenizerFamily::WORD_LEVEL) {  // :1161-1176: whitespace split, lower case, whole-word lookup; no BOS / EOS
    std::istringstream in(text);
    std::string word;
    while (in >> word) {
      for (char& c : word)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
      const auto it = token_to_id_.find(word);
      ids.push_back(it != token_to_id_.end() ? it->second : unk_id_);
    }
    return ids;
  }
  if (family_ == TokenizerFamily::LLAMA3_TIKTOKEN) {  // :1178-1195
    if (add_bos && bos_id_ != -1 && (bos_.empty() || text.compare(0, bos_.size(), bos_) != 0)) ids.push_back(bos_id_);
    const std::vector<int> body = tiktoken_ids(text);
    ids.insert(ids.end(), body.begin(), body.end());
    if (add_eos && eos_id_ != -1) ids.push_back(eos_id_);
    return ids;
  }
  if (family_ == TokenizerFamily::LLAMA_SENTENCEPIECE) {
    // GGUF path with pre_tok_type_ "unknown" (nothing sets it for GGUF files): the WHITESPACE fallback (:1337-1353): every
    // run of non-whitespace is a segment, every whitespace character a segment of its own (which then yields nothing)
    if (add_bos && bos_id_ != -1) ids.push_back(bos_id_);
    std::string run;
    auto flush = [&]() {
      if (run.empty()) return;
      const std::vector<int> piece_ids = pieces_to_ids(pieces_from_scores(run));
      ids.insert(ids.end(), piece_ids.begin(), piece_ids.end());
      run.clear();
    };
    for (char c : text) {
      if (c_space((unsigned char)c)) {
        flush();
        const std::vector<int> ws = pieces_to_ids(pieces_from_scores(std::string(1, c)));  // (always empty)
        ids.insert(ids.end(), ws.begin(), ws.end());
      } else {
        run += c;
      }
    }
    flush();
    if (add_eos && eos_id_ != -1) ids.push_back(eos_id_);
    return ids;
  }
  if (add_bos && bos_id_ != -1) ids.push_back(bos_id_);  // unknown family (:1399-1406)
  if (add_eos && eos_id_ != -1) ids.push_back(eos_id_);
  return ids;
}

// bpe_tokenize_from_scores (:59-169)
std::vector<std::string> Tokenizer::pieces_from_scores(const std::string& text) const {
  // leftmost search for  [\r\n]+ | [[:space:]]+ | [^\r\n[:space:][:alnum:]]+ | [[:alnum:]]+ : every byte starts one of the
  // four, so the units tile the text; the first alternative that matches at a position wins (a run of blanks swallows
  // the line breaks after it, a run of line breaks stops at the first blank)
  std::vector<std::string> units;
  for (size_t i = 0; i < text.size();) {
    const unsigned char c = (unsigned char)text[i];
    size_t j = i + 1;
    if (c == '\r' || c == '\n') {
      while (j < text.size() && (text[j] == '\r' || text[j] == '\n')) ++j;
    } else if (c_space(c)) {
      while (j < text.size() && c_space((unsigned char)text[j])) ++j;
    } else if (c_alnum(c)) {
      while (j < text.size() && c_alnum((unsigned char)text[j])) ++j;
    } else {
      while (j < text.size() && !c_space((unsigned char)text[j]) && !c_alnum((unsigned char)text[j])) ++j;
    }
    units.push_back(text.substr(i, j - i));
    i = j;
  }
  std::vector<std::string> out;
  bool needs_prefix = true;  // the first unit of the text and every unit after whitespace get the "▁" prefix
  for (const std::string& unit : units) {
    bool blank = true;
    for (char c : unit) blank &= c_space((unsigned char)c);
    if (blank) {
      needs_prefix = true;
      continue;
    }
    const std::string word = needs_prefix ? kSpSpace + unit : unit;
    needs_prefix = false;
    std::vector<std::string> parts;  // UTF-8 characters (a truncated last one stays as it is)
    for (size_t i = 0; i < word.size();) {
      const size_t n = utf8_len(word[i]);
      if (i + n <= word.size()) {
        parts.push_back(word.substr(i, n));
      } else {
        parts.push_back(word.substr(i));
        break;
      }
      i += n;
    }
    // merge the adjacent pair of lowest rank (first such pair) until none is a merge
    while (parts.size() > 1) {
      int best_rank = std::numeric_limits<int>::max(), best = -1;
      for (size_t i = 0; i + 1 < parts.size(); ++i) {
        const auto it = merges_.find(parts[i] + parts[i + 1]);
        if (it != merges_.end() && it->second < best_rank) {
          best_rank = it->second;
          best = (int)i;
        }
      }
      if (best < 0) break;
      parts[best] += parts[best + 1];
      parts.erase(parts.begin() + best + 1);
    }
    out.insert(out.end(), parts.begin(), parts.end());
  }
  return out;
}

// tokens_to_ids (:171-256): added tokens, vocabulary, vocabulary with the first letter capitalised, byte token, UNK
std::vector<int> Tokenizer::pieces_to_ids(const std::vector<std::string>& pieces) const {
  std::vector<int> ids;
  ids.reserve(pieces.size());
  for (const std::string& t : pieces) {
    const auto added = added_tokens_.find(t);
    if (added != added_tokens_.end()) {
      ids.push_back(added->second);
      continue;
    }
    const auto base = token_to_id_.find(t);
    if (base != token_to_id_.end()) {
      ids.push_back(base->second);
      continue;
    }
    std::string cap = t;  // capitalize_first_letter (:1980-2008): the letter after a leading "▁"
    if (!cap.empty()) {
      size_t pos = 0;
      bool skip = false;
      if (cap.compare(0, kSpSpace.size(), kSpSpace) == 0) {
        if (cap.size() > kSpSpace.size()) pos = kSpSpace.size();
        else skip = true;
      }
      if (!skip && cap[pos] >= 'a' && cap[pos] <= 'z') cap[pos] = (char)(cap[pos] - 'a' + 'A');
    }
    if (cap != t) {
      const auto it = token_to_id_.find(cap);
      if (it != token_to_id_.end()) {
        ids.push_back(it->second);
        continue;
      }
    }
    if (t.size() == 1) {
      const auto b = byte_to_id_.find(t[0]);
      if (b != byte_to_id_.end()) {
        ids.push_back(b->second);
        continue;
      }
    }
    ids.push_back(unk_id_);
  }
  return ids;
}

// add_bigram_to_queue_refactored (:2398-2470): the rank under which the pair (left, right) is queued, -1 if none. Tried:
// "Ġ" + " " + right when left is "Ġ"; left + right; left + " " + right when left starts with "Ġ"; left[0] + " " + right
// when left has two bytes and right one. The lowest rank wins.
int Tokenizer::merge_rank(const std::string& left, const std::string& right) const {
  int best = std::numeric_limits<int>::max();
  bool found = false;
  auto attempt = [&](const std::string& key) {
    const auto it = merges_.find(key);
    if (it != merges_.end() && it->second < best) {
      best = it->second;
      found = true;
    }
  };
  if (left == kGpt2Space) attempt(kGpt2Space + " " + right);
  attempt(left + right);
  if (left.compare(0, kGpt2Space.size(), kGpt2Space) == 0 && left != kGpt2Space) attempt(left + " " + right);
  if (left.size() == 2 && right.size() == 1) attempt(left.substr(0, 1) + " " + right);
  return found ? best : -1;
}

// bpe_tokenize_to_ids (:2129-2394) without its BOS / EOS arguments (encode passes false for both)
std::vector<int> Tokenizer::tiktoken_ids(const std::string& text) const {
  std::vector<int> out;
  auto byte_ids = [&](const std::string& bytes, bool whole_token_first) {
    for (char c : bytes) {
      if (whole_token_first) {
        const auto it = token_to_id_.find(std::string(1, c));
        if (it != token_to_id_.end()) {
          out.push_back(it->second);
          continue;
        }
      }
      const auto b = byte_to_id_.find(c);
      out.push_back(b != byte_to_id_.end() ? b->second : unk_id_);  // (unk_id_ is never -1 after init)
    }
  };
  size_t cur = 0;
  while (cur < text.size()) {
    bool special = false;
    for (const auto& st : added_by_length_) {  // longest special token first (:2164-2176)
      // (an EMPTY added token would match everywhere and never advance in the reference, :2168-2171: skipped here)
      if (!st.first.empty() && text.compare(cur, st.first.size(), st.first) == 0) {
        out.push_back(st.second);
        cur += st.first.size();
        special = true;
        break;
      }
    }
    if (special) continue;
    // one word: the pattern  <\|[^|]+\| | [[:alnum:]]+ | \.(?![<|]) | [^\s<|]+ | \s+  anchored at cur (:2152-2187)
    const size_t n = text.size();
    const unsigned char c = (unsigned char)text[cur];
    size_t end = cur;
    if (c == '<' && cur + 1 < n && text[cur + 1] == '|') {
      const size_t bar = text.find('|', cur + 2);  // [^|]+ needs at least one character before the closing bar
      if (bar != std::string::npos && bar > cur + 2) end = bar + 1;
    }
    if (end == cur) {
      if (c_alnum(c)) {
        end = cur + 1;
        while (end < n && c_alnum((unsigned char)text[end])) ++end;
      } else if (c == '.' && !(cur + 1 < n && (text[cur + 1] == '<' || text[cur + 1] == '|'))) {
        end = cur + 1;
      } else if (!c_space(c) && c != '<' && c != '|') {
        end = cur + 1;
        while (end < n && !c_space((unsigned char)text[end]) && text[end] != '<' && text[end] != '|') ++end;
      } else if (c_space(c)) {
        end = cur + 1;
        while (end < n && c_space((unsigned char)text[end])) ++end;
      }
    }
    if (end == cur) {  // a '<' or '|' no alternative takes: the REST of the text goes byte by byte (:2187-2210)
      byte_ids(text.substr(cur), true);
      break;
    }
    const std::string word = text.substr(cur, end - cur);
    cur = end;
    const auto direct = token_to_id_.find(word);
    if (direct != token_to_id_.end()) {
      out.push_back(direct->second);
      continue;
    }
    std::string w = word;  // a leading blank becomes "Ġ" (:2246-2254)
    if (w[0] == ' ') w = w.size() > 1 ? kGpt2Space + w.substr(1) : kGpt2Space;

    struct Sym { int prev, next; size_t off, n; };
    std::vector<Sym> sym;
    bool broken = false;
    for (size_t off = 0; off < w.size();) {
      const size_t len = utf8_len(w[off]);
      if (off + len > w.size()) {  // a truncated character: the word yields UNK (:2273-2291)
        broken = true;
        break;
      }
      sym.push_back({-1, -1, off, len});
      off += len;
    }
    if (broken || sym.empty()) {
      out.push_back(unk_id_);
      continue;
    }
    for (size_t i = 0; i < sym.size(); ++i) {
      sym[i].prev = i > 0 ? (int)i - 1 : -1;
      sym[i].next = i + 1 < sym.size() ? (int)i + 1 : -1;
    }
    typedef std::pair<int, int> Entry;  // (rank, index of the left symbol), lowest first
    std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> queue;
    auto push_pair = [&](int left) {
      if (left < 0 || (size_t)left >= sym.size()) return;
      const int right = sym[left].next;
      if (right < 0 || (size_t)right >= sym.size() || right <= left) return;
      if (sym[left].n == 0 || sym[right].n == 0) return;
      const int rank = merge_rank(w.substr(sym[left].off, sym[left].n), w.substr(sym[right].off, sym[right].n));
      if (rank >= 0) queue.push({rank, left});
    };
    for (size_t i = 0; i + 1 < sym.size(); ++i) push_pair((int)i);
    while (!queue.empty()) {
      const int left = queue.top().second;
      queue.pop();
      if (sym[left].n == 0) continue;
      const int right = sym[left].next;
      if (right == -1 || sym[right].n == 0) continue;
      // (no check that the pair still has the rank it was queued with: :2312-2339 merges whatever sits there now)
      sym[left].n += sym[right].n;
      sym[right].n = 0;
      sym[left].next = sym[right].next;
      if (sym[left].next != -1) sym[sym[left].next].prev = left;
      if (sym[left].prev != -1) push_pair(sym[left].prev);
      if (sym[left].next != -1) push_pair(left);
    }
    const size_t before = out.size();
    for (int i = 0; i != -1; i = sym[i].next) {
      if (sym[i].n == 0) continue;
      const std::string piece = w.substr(sym[i].off, sym[i].n);
      const auto it = token_to_id_.find(piece);
      if (it != token_to_id_.end()) out.push_back(it->second);
      else byte_ids(piece, false);
    }
    if (out.size() == before) out.push_back(unk_id_);
  }
  return out;
}

// ---- decode (tokenizer.cpp:1412-1652) ----------------------------------------------------------------------------------
std::string Tokenizer::decode(const std::vector<int>& ids, bool skip_special) const {
  if (family_ == TokenizerFamily::WORD_LEVEL) {  // :1415-1431
    std::string out;
    bool first = true;
    for (int id : ids) {
      if (id < 0 || (size_t)id >= id_to_token_.size()) continue;
      if (skip_special && (id == bos_id_ || id == eos_id_ || id == pad_id_ || id == unk_id_)) continue;
      if (!first) out += ' ';
      out += id_to_token_[id];
      first = false;
    }
    return out;
  }
  if (family_ == TokenizerFamily::LLAMA_SENTENCEPIECE) return decode_sentencepiece(ids, skip_special);
  std::string out;  // Llama-3 and unknown families (:1438-1508)
  bool first = true;
  for (int id : ids) {
    if (id < 0 || (size_t)id >= id_to_token_.size()) {
      if (!skip_special) {
        out += "[INVALID_ID:" + std::to_string(id) + "]";
        first = false;
      }
      continue;
    }
    if (skip_special && (id == bos_id_ || id == eos_id_ || id == pad_id_ || id == unk_id_ || id_to_added_.count(id))) continue;
    std::string t = id_to_token_[id];
    if (t.empty()) {
      if (skip_special || unk_id_ == -1) continue;
      t = unk_;
    }
    if (t.compare(0, kGpt2Space.size(), kGpt2Space) == 0) {
      if (!first) out += ' ';
      out += t.substr(kGpt2Space.size());
    } else {
      out += t;
    }
    first = false;
  }
  return out;
}

std::string Tokenizer::decode_sentencepiece(const std::vector<int>& ids, bool skip_special) const {
  std::string out;
  bool first = true;
  for (int id : ids) {
    if (skip_special && (id == bos_id_ || id == eos_id_ || id == pad_id_ || id == unk_id_ || id_to_added_.count(id))) continue;
    std::string t;
    bool invalid = false;
    if (id >= 0 && (size_t)id < id_to_token_.size()) {
      t = id_to_token_[id];
    } else {
      const auto a = id_to_added_.find(id);
      if (a != id_to_added_.end()) {
        t = a->second;
      } else {
        t = skip_special ? std::string() : "[INVALID_ID:" + std::to_string(id) + "]";
        invalid = true;
      }
    }
    int byte = 0;
    if (parse_hex_byte_token(t, &byte)) t = std::string(1, static_cast<char>(byte));  // <0xNN> -> the byte (:1555-1564)
    if (t.empty() && !invalid) {  // an empty vocabulary entry (:1566-1591)
      if (unk_id_ != -1) {
        if (skip_special && unk_id_ == id) continue;
        t = unk_;
      } else {
        if (skip_special) continue;
        t = "[EMPTY_TOKEN_FOR_ID:" + std::to_string(id) + "]";
      }
    }
    if (t == kSpSpace || t == kGpt2Space) {  // the bare space marker (:1593-1607)
      if (first) continue;
      if (out.empty() || out.back() != ' ') out += ' ';
      first = false;
      continue;
    }
    if (t.empty()) continue;
    const bool sp = t.compare(0, kSpSpace.size(), kSpSpace) == 0;
    const bool gp = !sp && t.compare(0, kGpt2Space.size(), kGpt2Space) == 0;
    if (sp || gp) {
      if (!first && (out.empty() || out.back() != ' ')) out += ' ';
      std::string content = t.substr(sp ? kSpSpace.size() : kGpt2Space.size());
      const size_t k = content.find_first_not_of(' ');  // leading blanks of the content go (an all-blank one stays)
      if (k != std::string::npos) content = content.substr(k);
      out += content;
    } else {
      out += t;
    }
    first = false;
  }
  return out;
}

// ---- chat template and prompt formatting --------------------------------------------------------------------------------
// apply_chat_template (:1663-1741): a GGUF template without Jinja marks gets its {{...}} placeholders replaced; a Jinja one
// (or none) is replaced by the hard-coded Llama-3 instruct layout
std::string Tokenizer::apply_chat_template(const std::string& user_prompt, const std::string& system_message) const {
  const bool jinja = !chat_template_.empty() &&
                     (chat_template_.find("{%") != std::string::npos || chat_template_.find("{{") != std::string::npos);
  if (!chat_template_.empty() && !jinja) {
    // (unreachable for a template that has placeholders: "{{" already counts as Jinja; kept as the reference has it)
    std::string t = chat_template_;
    t = replace_all(t, "{{bos_token}}", bos_id_ != -1 ? bos_ : "");
    t = replace_all(t, "{{eos_token}}", eos_id_ != -1 ? eos_ : "");
    t = replace_all(t, "{{user_prompt}}", user_prompt);
    t = replace_all(t, "{{system_message}}", system_message);
    return t;
  }
  const std::string bos = bos_id_ != -1 ? bos_ : "<s>";
  const std::string sh = "<|start_header_id|>", eh = "<|end_header_id|>", eot = "<|eot_id|>";
  std::string out = bos;
  if (!system_message.empty()) out += sh + "system" + eh + "\n\n" + system_message + eot;
  out += sh + "user" + eh + "\n\n" + user_prompt + eot;
  out += sh + "assistant" + eh + "\n\n";
  return out;
}

// api.cpp:505-572. The Q/A wrappers carry LITERAL backslash-n sequences (the source writes "\\n"), two backslashes in the
// SentencePiece-with-Jinja branch: restated as they are.
std::string Tokenizer::format_prompt(const std::string& user_prompt, const std::string& system_prompt, bool apply_q_a_format,
                                     bool batch_variant) const {
  if (family_ == TokenizerFamily::WORD_LEVEL && !batch_variant) return user_prompt;
  if (apply_q_a_format)
    return system_prompt.empty() ? "Q: " + user_prompt + "\\nA:" : system_prompt + "\\n\\nQ: " + user_prompt + "\\nA:";
  if (!chat_template_.empty()) {
    if (family_ == TokenizerFamily::LLAMA_SENTENCEPIECE && chat_template_.find("{%") != std::string::npos)
      return system_prompt.empty() ? "Q: " + user_prompt + "\\\\nA:" : system_prompt + "\\\\n\\\\nQ: " + user_prompt + "\\\\nA:";
    return apply_chat_template(user_prompt, system_prompt);
  }
  if (family_ == TokenizerFamily::LLAMA3_TIKTOKEN) return apply_chat_template(user_prompt, system_prompt);
  return system_prompt.empty() ? user_prompt : system_prompt + "\\n\\n" + user_prompt;
}

}  // namespace tlhost
