// tl_tokenizer.hpp — the reference's tokenizer (tokenizer.cpp) without Boost: SURVEY §8 row f3.
//
// What TinyLlamaSession needs from Tokenizer when the model comes from a GGUF file: the constructor from the file's
// metadata (tokenizer.cpp:516-1048), encode (:1141-1410), decode (:1412-1652), apply_chat_template (:1663-1741), for the
// three families model_config.cpp:221-253 tells apart (Llama-3 "tiktoken" BPE with merge ranks, Llama-2 / TinyLlama
// SentencePiece-style BPE from scores, word-level). The two Boost regexes on that path are ordered alternations of
// ASCII character classes in the classic locale; they are matched here by hand-written scanners:
//   tiktoken (:2152, anchored at the cursor):  <\|[^|]+\|  |  [[:alnum:]]+  |  \.(?![<|])  |  [^\s<|]+  |  \s+
//   scores   (:66, leftmost search):           [\r\n]+  |  [[:space:]]+  |  [^\r\n[:space:][:alnum:]]+  |  [[:alnum:]]+
// This is a restatement of the reference's behaviour, quirks included (a word never receives the "Ġ" of the space in
// front of it; a '<' that opens no <|...|> sends the rest of the text down a byte-by-byte path; stale entries of the merge
// queue merge whatever pair sits there; a multi-byte piece outside the vocabulary becomes UNK, not bytes; ...), not a
// tokenizer that agrees with llama.cpp or tiktoken. PARITY UNPINNED: tokenizer.cpp needs Boost and nlohmann/json, which
// this image lacks, and the reference ships no tokenizer vectors; tests compare with an independent Python restatement
// (oracle/tokenizer_oracle.py) that runs the reference's pattern strings through Python's regex engine.
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "tl_gguf.hpp"

namespace tlhost {

enum class TokenizerFamily { UNKNOWN = 0, LLAMA_SENTENCEPIECE = 1, LLAMA3_TIKTOKEN = 2, WORD_LEVEL = 3 };

// model_config.cpp:221-253 on the file's metadata
TokenizerFamily tokenizer_family_from_gguf(const GGUFFile& f);

class Tokenizer {
 public:
  // family_override: UNKNOWN = decide from the file as the reference does; anything else forces that family (tests with
  // small vocabularies: the reference recognises Llama-3 by its vocabulary size of 128256)
  explicit Tokenizer(const GGUFFile& f, TokenizerFamily family_override = TokenizerFamily::UNKNOWN);
  // the same from explicit parts (what the constructor reads out of the file)
  struct Parts {
    TokenizerFamily family = TokenizerFamily::UNKNOWN;
    std::vector<std::string> tokens, merges;
    std::vector<float> scores;
    std::vector<int32_t> token_types;
    int bos = -1, eos = -1, unk = -1, pad = -1;
    std::string chat_template;
  };
  explicit Tokenizer(const Parts& p);

  std::vector<int> encode(const std::string& text, bool add_bos = true, bool add_eos = false) const;
  std::string decode(const std::vector<int>& ids, bool skip_special_tokens = true) const;
  std::string apply_chat_template(const std::string& user_prompt, const std::string& system_message) const;
  // the prompt TinyLlamaSession::generate encodes (api.cpp:505-572): Q/A wrapper, GGUF template, Llama-3 template or raw.
  // batch_variant: generate_batch's copy of that logic (api.cpp:872-905), which lacks the word-level case
  std::string format_prompt(const std::string& user_prompt, const std::string& system_prompt, bool apply_q_a_format,
                            bool batch_variant = false) const;

  TokenizerFamily family() const { return family_; }
  int vocab_size() const { return (int)id_to_token_.size(); }
  int bos_token_id() const { return bos_id_; }
  int eos_token_id() const { return eos_id_; }
  int unk_token_id() const { return unk_id_; }
  int pad_token_id() const { return pad_id_; }
  const std::string& chat_template() const { return chat_template_; }

 private:
  void init(const Parts& p);
  std::vector<int> tiktoken_ids(const std::string& text) const;                    // bpe_tokenize_to_ids :2129-2394
  std::vector<std::string> pieces_from_scores(const std::string& segment) const;   // bpe_tokenize_from_scores :59-169
  std::vector<int> pieces_to_ids(const std::vector<std::string>& pieces) const;    // tokens_to_ids :171-256
  std::string decode_sentencepiece(const std::vector<int>& ids, bool skip_special) const;  // :1511-1652
  int merge_rank(const std::string& left, const std::string& right) const;         // add_bigram_to_queue_refactored :2398

  TokenizerFamily family_ = TokenizerFamily::UNKNOWN;
  std::vector<std::string> id_to_token_;
  std::unordered_map<std::string, int> token_to_id_, merges_, added_tokens_;
  std::unordered_map<int, std::string> id_to_added_;
  std::vector<std::pair<std::string, int>> added_by_length_;  // longest first (sort_tokens_by_length_desc :2115)
  std::map<char, int> byte_to_id_;
  int bos_id_ = -1, eos_id_ = -1, unk_id_ = -1, pad_id_ = -1;
  std::string bos_, eos_, unk_, pad_, chat_template_;
};

}  // namespace tlhost
