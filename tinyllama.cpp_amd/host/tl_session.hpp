// tl_session.hpp — the C++ host above the C-ABI (include/tlhip.h): what TinyLlamaSession / TinyLlamaModel keep doing on
// the host once the CUDA backend is replaced by libtlhip.so.
//
// Mirrors tinyllama::TinyLlamaSession (api.h:26-143). generate() / generate_batch() exist twice: on TOKEN IDS (the part
// whose parity is pinned), and on text as in the reference, through tl_tokenizer.hpp (a restatement of tokenizer.cpp
// without Boost; parity unpinned, see there). Same argument meaning (steps, temperature, top_k, top_p, system prompt,
// Q/A hint), same loop (api.cpp:585-813), same error convention (std::runtime_error).
#pragma once
#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "tl_gguf.hpp"
#include "tl_tokenizer.hpp"
#include "tlhip.h"

namespace tlhost {

// what parse_model_config_from_gguf reads (model_config.cpp:149-221)
struct ModelConfig {
  int hidden_size = 0, intermediate_size = 0, num_attention_heads = 0, num_key_value_heads = 0, num_hidden_layers = 0;
  int vocab_size = 32000, max_position_embeddings = 2048;
  float rms_norm_eps = 1e-5f, rope_theta = 10000.0f;
  int bos_token_id = 1, eos_token_id = 2;
  std::string architecture, name;
};
ModelConfig config_from_gguf(const GGUFFile& f);

struct LoadStats {
  size_t tensors = 0, quantized_bytes = 0;
  double seconds = 0.0;
};

class Session {
 public:
  // model_path: a .gguf file. device: HIP device ordinal. max_context > 0 caps the KV cache length below the file's
  // context length. tp_rank / tp_size: tensor-parallel shard (the caller wires the communicator through model()).
  // max_batch_size: most prompts one generate_batch() call may carry (api.h:39-46; 1 = single-sequence use only).
  explicit Session(const std::string& model_path, int device = 0, int max_context = 0, int max_batch_size = 1);
  ~Session();
  Session(const Session&) = delete;
  Session& operator=(const Session&) = delete;

  // TinyLlamaSession::generate on ids: returns the generated ids (prompt excluded). temperature < 0.05 is greedy
  // (api.cpp:134) and runs entirely on the device; otherwise top-k / top-p sampling (api.cpp:126-216), on the device as
  // well when the effective top_k is <= 1024 (tl_model_generate_sampled), else on the host from the returned logits.
  std::vector<int> generate(const std::vector<int>& prompt_ids, int steps = 128, float temperature = 0.1f, int top_k = 40,
                            float top_p = 0.9f);

  // TinyLlamaSession::generate_batch on ids (api.cpp:836-1142): all prompts advance together, one step streams the
  // weights once for all of them (tl_model_batch_*). Returns the generated ids per prompt (an EOS ends a sequence and
  // is kept, api.cpp:1004,1092). No repetition penalty on this path (api.cpp:993,1084). Throws std::runtime_error on an
  // empty prompt list or more prompts than max_batch_size (api.cpp:844-851). temperature < 0.05: the whole loop runs
  // on the device; otherwise the logits of the active sequences come back every step and the host sampler draws.
  std::vector<std::vector<int>> generate_batch(const std::vector<std::vector<int>>& prompts, int steps = 128,
                                               float temperature = 0.1f, int top_k = 40, float top_p = 0.9f);

  // the reference's own signatures (api.h:59-78): the prompt is formatted (Tokenizer::format_prompt, api.cpp:505-572),
  // encoded with BOS, generated from as above, and the generated ids are decoded without special tokens. Throw
  // std::runtime_error when the file carries no tokenizer ("Model or tokenizer not loaded.", api.cpp:491).
  std::string generate(const std::string& prompt, int steps = 128, float temperature = 0.1f, int top_k = 40,
                       float top_p = 0.9f, const std::string& system_prompt = "", bool apply_q_a_format = false);
  std::vector<std::string> generate_batch(const std::vector<std::string>& prompts, int steps = 128, float temperature = 0.1f,
                                          int top_k = 40, float top_p = 0.9f, const std::string& system_prompt = "",
                                          bool apply_q_a_format = false);
  // null when the file has no tokenizer.ggml.tokens (the synthetic test models)
  const Tokenizer* get_tokenizer() const { return tokenizer_.get(); }

  void set_max_batch_size(int n);
  int get_max_batch_size() const { return max_batch_size_; }
  void set_repetition_penalty(float p) { repetition_penalty_ = p; }
  float get_repetition_penalty() const { return repetition_penalty_; }
  void seed(uint32_t s) { rng_.seed(s); }
  const ModelConfig& get_config() const { return config_; }
  const LoadStats& load_stats() const { return stats_; }
  tl_model* model() { return model_; }

 private:
  ModelConfig config_;
  tl_model* model_ = nullptr;
  std::unique_ptr<Tokenizer> tokenizer_;
  std::mt19937 rng_{std::random_device{}()};
  float repetition_penalty_ = 1.3f;  // api.h:135
  int max_batch_size_ = 1;           // api.h:142
  int batch_ready_ = 0;              // sequences the device-side batch cache was sized for
  LoadStats stats_;
};

// api.cpp:86-97 and :126-216 on the host (used on the non-greedy path, where the logits come back from the device)
void apply_repetition_penalty(std::vector<float>& logits, const std::vector<int>& generated, float penalty,
                              size_t window = 64);
int sample_top_k_top_p_temperature(const std::vector<float>& logits, float temperature, int top_k, float top_p,
                                   std::mt19937& rng);
// the candidate set and renormalised probabilities the sampler draws from (exposed for tests)
void sampling_distribution(const std::vector<float>& logits, float temperature, int top_k, float top_p,
                           std::vector<int>* ids, std::vector<float>* probs);

}  // namespace tlhost
