// tl_session.cpp — see tl_session.hpp.
#include "tl_session.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <stdexcept>

namespace tlhost {

namespace {

void tl_check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string("tlhip: ") + what + ": " + tl_last_error());
}

int meta_int(const GGUFFile& f, const char* key, int dflt) {
  const GGUFValue* v = f.meta(key);
  int64_t x = 0;
  return (v && v->as_integer(&x)) ? (int)x : dflt;
}
float meta_float(const GGUFFile& f, const char* key, float dflt) {
  const GGUFValue* v = f.meta(key);
  double x = 0;
  return (v && v->kind == GGUFValue::FLOAT && v->as_double(&x)) ? (float)x : dflt;
}
std::string meta_str(const GGUFFile& f, const char* key) {
  const GGUFValue* v = f.meta(key);
  return (v && v->kind == GGUFValue::STRING) ? v->s : std::string();
}

// GGUF tensor name -> (kind, layer) of tl_model_set_tensor (names as consumed by weight_management.cpp:732-960)
bool tensor_slot(const std::string& name, int* kind, int* layer) {
  *layer = -1;
  if (name == "token_embd.weight") { *kind = TL_T_TOKEN_EMBD; return true; }
  if (name == "output_norm.weight") { *kind = TL_T_OUTPUT_NORM; return true; }
  if (name == "output.weight") { *kind = TL_T_OUTPUT; return true; }
  if (name.compare(0, 4, "blk.") != 0) return false;
  const size_t dot = name.find('.', 4);
  if (dot == std::string::npos) return false;
  try {
    *layer = std::stoi(name.substr(4, dot - 4));
  } catch (...) {
    return false;
  }
  static const struct { const char* suffix; int kind; } table[] = {
      {"attn_q.weight", TL_T_ATTN_Q},           {"attn_k.weight", TL_T_ATTN_K},       {"attn_v.weight", TL_T_ATTN_V},
      {"attn_output.weight", TL_T_ATTN_OUTPUT}, {"ffn_gate.weight", TL_T_FFN_GATE},   {"ffn_up.weight", TL_T_FFN_UP},
      {"ffn_down.weight", TL_T_FFN_DOWN},       {"attn_norm.weight", TL_T_ATTN_NORM}, {"ffn_norm.weight", TL_T_FFN_NORM}};
  const std::string suffix = name.substr(dot + 1);
  for (const auto& e : table)
    if (suffix == e.suffix) {
      *kind = e.kind;
      return true;
    }
  return false;  // rope_freqs.weight and friends are ignored, as in the reference
}

}  // namespace

ModelConfig config_from_gguf(const GGUFFile& f) {
  ModelConfig c;
  c.vocab_size = meta_int(f, "tokenizer.ggml.vocab_size", meta_int(f, "llama.vocab_size", 32000));
  c.hidden_size = meta_int(f, "llama.embedding_length", 0);
  c.intermediate_size = meta_int(f, "llama.feed_forward_length", 0);
  c.num_attention_heads = meta_int(f, "llama.attention.head_count", 0);
  c.num_key_value_heads = meta_int(f, "llama.attention.head_count_kv", c.num_attention_heads);
  c.num_hidden_layers = meta_int(f, "llama.block_count", 0);
  // model_config.cpp:168-177: llama.* with phi.* / mistral.* fallbacks, 4096 when the key is missing, and anything outside
  // (0, 8192] becomes 2048
  c.max_position_embeddings =
      meta_int(f, "llama.context_length", meta_int(f, "phi.context_length", meta_int(f, "mistral.context_length", 4096)));
  if (c.max_position_embeddings <= 0 || c.max_position_embeddings > 8192) c.max_position_embeddings = 2048;
  c.rms_norm_eps = meta_float(f, "llama.attention.layer_norm_rms_epsilon", 1e-5f);
  c.rope_theta = meta_float(f, "llama.rope.freq_base", 10000.0f);
  // model_config.cpp:242-248: a file classified LLAMA3_TIKTOKEN whose rope_theta came out as 10000 re-reads the key with a
  // default of 500000 and takes it when it is above 10000 -- i.e. a Llama-3 file WITHOUT llama.rope.freq_base runs at 500000
  if (c.rope_theta == 10000.0f && tokenizer_family_from_gguf(f) == TokenizerFamily::LLAMA3_TIKTOKEN) {
    const float cand = meta_float(f, "llama.rope.freq_base", 500000.0f);
    if (cand > 10000.0f) c.rope_theta = cand;
  }
  c.bos_token_id = meta_int(f, "tokenizer.ggml.bos_token_id", -1);  // (-1: no such token, model_config.cpp:199-200)
  c.eos_token_id = meta_int(f, "tokenizer.ggml.eos_token_id", -1);
  c.architecture = meta_str(f, "general.architecture");
  c.name = meta_str(f, "general.name");
  return c;
}

Session::Session(const std::string& model_path, int device, int max_context, int max_batch_size) {
  const auto t0 = std::chrono::steady_clock::now();
  GGUFFile file;
  file.open(model_path);
  config_ = config_from_gguf(file);
  if (config_.hidden_size <= 0 || config_.num_hidden_layers <= 0 || config_.num_attention_heads <= 0 ||
      config_.intermediate_size <= 0)
    throw std::runtime_error("Session: " + model_path + " lacks the llama.* shape metadata");
  if (max_context > 0 && max_context < config_.max_position_embeddings) config_.max_position_embeddings = max_context;
  if (max_batch_size < 1) throw std::runtime_error("Session: max_batch_size must be >= 1");
  max_batch_size_ = max_batch_size;
  if (const GGUFValue* toks = file.meta("tokenizer.ggml.tokens"))
    if (!toks->strings.empty()) tokenizer_.reset(new Tokenizer(file));
  tl_check(tl_init(device), "tl_init");
  tl_model_config mc = {};
  mc.hidden_size = config_.hidden_size;
  mc.intermediate_size = config_.intermediate_size;
  mc.num_attention_heads = config_.num_attention_heads;
  mc.num_key_value_heads = config_.num_key_value_heads;
  mc.num_hidden_layers = config_.num_hidden_layers;
  mc.vocab_size = config_.vocab_size;
  mc.max_position_embeddings = config_.max_position_embeddings;
  mc.rms_norm_eps = config_.rms_norm_eps;
  mc.rope_theta = config_.rope_theta;
  mc.tp_rank = 0;
  mc.tp_size = 1;
  tl_check(tl_model_create(&mc, &model_), "tl_model_create");
  try {
    // raw blocks straight from the mapping: no typed host copy, no FP32 expansion (vs weight_management.cpp:677-973)
    for (const GGUFTensor& t : file.tensors()) {
      int kind = 0, layer = -1;
      if (!tensor_slot(t.name, &kind, &layer)) continue;
      if (layer >= config_.num_hidden_layers) continue;
      int type = (int)t.type;
      if (type != TL_F32 && type != TL_BF16 && type != TL_Q8_0 && type != TL_Q4_K && type != TL_Q6_K && type != TL_Q8_K)
        throw std::runtime_error("Session: tensor " + t.name + " has ggml type " + std::to_string(type) +
                                 ", which this backend does not hold (F32, BF16, Q8_0, Q4_K, Q6_K, Q8_K)");
      const int cols = (int)t.cols(), rows = (int)t.rows();
      tl_check(tl_model_set_tensor(model_, kind, layer, type, t.data, rows, cols), ("tl_model_set_tensor " + t.name).c_str());
      ++stats_.tensors;
      if (type != TL_F32 && type != TL_BF16) stats_.quantized_bytes += t.nbytes;
    }
    tl_check(tl_model_finalize(model_), "tl_model_finalize");
  } catch (...) {
    tl_model_free(model_);
    model_ = nullptr;
    throw;
  }
  stats_.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

Session::~Session() {
  if (model_) tl_model_free(model_);
}

std::vector<int> Session::generate(const std::vector<int>& prompt, int steps, float temperature, int top_k, float top_p) {
  std::vector<int> generated;
  const int P = (int)prompt.size();
  if (P == 0 || steps <= 0) return generated;  // empty token list -> empty result (api.cpp:576-580)
  const int max_pos = config_.max_position_embeddings, V = config_.vocab_size;
  const bool greedy = temperature < 0.05f;
  tl_check(tl_model_reset_kv(model_), "tl_model_reset_kv");  // kv_cache_.clear_data(), api.cpp:592
  std::vector<float> logits((size_t)V);
  int pos;
  // ---- the prompt: batched prefill from 32 tokens on (api.cpp:601), token by token below that ---------------------
  if (P >= 32) {
    if (P > max_pos) throw std::runtime_error("Session::generate: prompt longer than the context");
    tl_check(tl_model_prefill(model_, prompt.data(), P, 0, logits.data(), nullptr), "tl_model_prefill");
  } else {
    for (pos = 0; pos < P; ++pos) {
      if (pos >= max_pos) return generated;  // api.cpp:697-702
      tl_check(tl_model_decode(model_, prompt[pos], pos, pos == P - 1 ? logits.data() : nullptr, nullptr), "tl_model_decode");
    }
  }
  // ---- first sampled id: from the last prompt position's logits; the history is empty, the penalty a no-op -----------
  int next = greedy ? (int)(std::max_element(logits.begin(), logits.end()) - logits.begin())
                    : sample_top_k_top_p_temperature(logits, temperature, top_k, top_p, rng_);
  generated.push_back(next);
  // remaining positions P .. P + steps - 2, bounded by the context; an EOS sampled at pos == P-1 does not stop (api.cpp:794)
  const int n_more = std::min(steps - 1, max_pos - P);
  if (n_more <= 0) return generated;
  if (greedy) {  // the whole loop on the device; EOS is applied on the returned ids
    std::vector<int> more((size_t)n_more);
    tl_check(tl_model_generate_greedy(model_, next, P, n_more, repetition_penalty_, generated.data(), 1, more.data()),
             "tl_model_generate_greedy");
    for (int id : more) {
      generated.push_back(id);
      if (id == config_.eos_token_id) break;
    }
    return generated;
  }
  const int k_eff = (top_k <= 0 || top_k > V) ? V : top_k;
  if (k_eff <= 1024) {  // the sampler runs on the device too (tl_sample.hip): no logits cross the bus per token
    std::vector<int> more((size_t)n_more);
    const unsigned long long seed = ((unsigned long long)rng_() << 32) | rng_();
    tl_check(tl_model_generate_sampled(model_, next, P, n_more, repetition_penalty_, generated.data(), 1, temperature,
                                       top_k, top_p, seed, more.data()),
             "tl_model_generate_sampled");
    for (int id : more) {
      generated.push_back(id);
      if (id == config_.eos_token_id) break;
    }
    return generated;
  }
  // "no top-k" over a large vocabulary: logits back per token, the reference's host sampler
  for (pos = P; pos < P + n_more; ++pos) {
    tl_check(tl_model_decode(model_, next, pos, logits.data(), nullptr), "tl_model_decode");
    apply_repetition_penalty(logits, generated, repetition_penalty_);
    next = sample_top_k_top_p_temperature(logits, temperature, top_k, top_p, rng_);
    generated.push_back(next);
    if (next == config_.eos_token_id) break;
  }
  return generated;
}

std::string Session::generate(const std::string& prompt, int steps, float temperature, int top_k, float top_p,
                              const std::string& system_prompt, bool apply_q_a_format) {
  if (!model_ || !tokenizer_) throw std::runtime_error("Model or tokenizer not loaded.");
  const std::vector<int> ids = tokenizer_->encode(tokenizer_->format_prompt(prompt, system_prompt, apply_q_a_format), true, false);
  if (ids.empty()) return "";  // api.cpp:576-580
  return tokenizer_->decode(generate(ids, steps, temperature, top_k, top_p), true);
}

std::vector<std::string> Session::generate_batch(const std::vector<std::string>& prompts, int steps, float temperature,
                                                 int top_k, float top_p, const std::string& system_prompt,
                                                 bool apply_q_a_format) {
  if (prompts.empty()) throw std::runtime_error("Cannot process empty prompts vector for batch generation.");
  if ((int)prompts.size() > max_batch_size_)
    throw std::runtime_error("Batch size " + std::to_string(prompts.size()) + " exceeds maximum batch size " +
                             std::to_string(max_batch_size_));
  if (!model_ || !tokenizer_) throw std::runtime_error("Model or tokenizer not loaded for batch generation.");
  std::vector<std::vector<int>> ids(prompts.size());
  for (size_t i = 0; i < prompts.size(); ++i) {
    ids[i] = tokenizer_->encode(tokenizer_->format_prompt(prompts[i], system_prompt, apply_q_a_format, true), true, false);
    if (ids[i].empty()) ids[i].push_back(tokenizer_->bos_token_id());  // api.cpp:911-914
  }
  const std::vector<std::vector<int>> out = generate_batch(ids, steps, temperature, top_k, top_p);
  std::vector<std::string> text(out.size());
  for (size_t i = 0; i < out.size(); ++i) text[i] = tokenizer_->decode(out[i], true);  // api.cpp:1107
  return text;
}

void Session::set_max_batch_size(int n) {
  if (n < 1) throw std::runtime_error("Session: max_batch_size must be >= 1");
  max_batch_size_ = n;
}

std::vector<std::vector<int>> Session::generate_batch(const std::vector<std::vector<int>>& prompts, int steps,
                                                      float temperature, int top_k, float top_p) {
  if (prompts.empty()) throw std::runtime_error("Cannot process empty prompts vector for batch generation.");  // api.cpp:844
  if ((int)prompts.size() > max_batch_size_)
    throw std::runtime_error("Batch size " + std::to_string(prompts.size()) + " exceeds maximum batch size " +
                             std::to_string(max_batch_size_));  // api.cpp:848-851
  const int B = (int)prompts.size(), max_pos = config_.max_position_embeddings, V = config_.vocab_size;
  std::vector<std::vector<int>> out((size_t)B);
  if (steps <= 0) return out;
  for (const auto& p : prompts) {
    if (p.empty()) throw std::runtime_error("Session::generate_batch: empty prompt (the reference substitutes BOS after tokenizing, api.cpp:912-915)");
    if ((int)p.size() > max_pos) throw std::runtime_error("Session::generate_batch: prompt longer than the context");
  }
  if (batch_ready_ < max_batch_size_) {  // kv_cache_.initialize_batch (api.cpp:925): once, for the session's maximum
    tl_check(tl_model_batch_init(model_, max_batch_size_), "tl_model_batch_init");
    batch_ready_ = max_batch_size_;
  }
  const bool greedy = temperature < 0.05f;
  // ---- phase 1: the prompts (batch_prefill_parallel, api.cpp:1144): sequence i lives in cache slot i -------------------
  std::vector<float> logits((size_t)V);
  std::vector<int> cur((size_t)B), pos((size_t)B), seq((size_t)B);
  std::vector<char> done((size_t)B, 0);
  {  // all prompts as one packed batch, token by token in sequence order (api.cpp:1175-1222)
    std::vector<int> ptok, ppos, pseq, last((size_t)B);
    for (int i = 0; i < B; ++i) {
      for (size_t k = 0; k < prompts[i].size(); ++k) {
        ptok.push_back(prompts[i][k]);
        ppos.push_back((int)k);
        pseq.push_back(i);
      }
      last[i] = (int)ptok.size() - 1;
    }
    std::vector<float> last_logits(greedy ? 0 : (size_t)B * V);
    tl_check(tl_model_batch_prefill_packed(model_, (int)ptok.size(), ptok.data(), ppos.data(), pseq.data(), B, last.data(),
                                           greedy ? nullptr : last_logits.data(), cur.data()),
             "tl_model_batch_prefill_packed");
    if (!greedy)
      for (int i = 0; i < B; ++i) {
        std::copy(last_logits.begin() + (size_t)i * V, last_logits.begin() + (size_t)(i + 1) * V, logits.begin());
        cur[i] = sample_top_k_top_p_temperature(logits, temperature, top_k, top_p, rng_);
      }
  }
  for (int i = 0; i < B; ++i) {
    out[i].push_back(cur[i]);
    pos[i] = (int)prompts[i].size();
    seq[i] = i;
    done[i] = cur[i] == config_.eos_token_id;
  }
  // ---- phase 2: steps 1 .. steps-1 (api.cpp:1013-1103), bounded by the context ------------------------------------------
  if (greedy) {  // ids and positions advance on the device; a column is cut at its EOS afterwards (sequences never interact)
    // In rounds: a round runs as many steps as the active sequence closest to the end of the context still has room for;
    // that sequence then drops out (api.cpp:1031-1046 drops it from the step's batch) and the others carry on.
    int produced = 1;
    while (produced < steps) {
      std::vector<int> at, ap, as;
      int room = max_pos;
      for (int i = 0; i < B; ++i)
        if (!done[i] && pos[i] < max_pos) {
          at.push_back(cur[i]);
          ap.push_back(pos[i]);
          as.push_back(i);
          room = std::min(room, max_pos - pos[i]);
        }
      if (at.empty()) break;
      const int n = std::min(steps - produced, room), A = (int)at.size();
      std::vector<int> ids((size_t)n * A);
      tl_check(tl_model_batch_generate_greedy(model_, A, at.data(), ap.data(), as.data(), n, ids.data()),
               "tl_model_batch_generate_greedy");
      for (int a = 0; a < A; ++a) {
        const int i = as[a];
        for (int k = 0; k < n && !done[i]; ++k) {
          const int id = ids[(size_t)k * A + a];
          out[i].push_back(id);
          if (id == config_.eos_token_id) done[i] = 1;
        }
        cur[i] = ids[(size_t)(n - 1) * A + a];
        pos[i] += n;
      }
      produced += n;
    }
    return out;
  }
  std::vector<float> step_logits;
  for (int step = 1; step < steps; ++step) {
    std::vector<int> at, ap, as;  // the active sequences of this step (api.cpp:1031-1046)
    for (int i = 0; i < B; ++i)
      if (!done[i] && pos[i] < max_pos) {
        at.push_back(cur[i]);
        ap.push_back(pos[i]);
        as.push_back(i);
      }
    if (at.empty()) break;
    step_logits.resize(at.size() * (size_t)V);
    tl_check(tl_model_batch_decode(model_, (int)at.size(), at.data(), ap.data(), as.data(), step_logits.data(), nullptr),
             "tl_model_batch_decode");
    for (size_t a = 0; a < at.size(); ++a) {
      const int i = as[a];
      std::copy(step_logits.begin() + a * V, step_logits.begin() + (a + 1) * V, logits.begin());
      const int next = sample_top_k_top_p_temperature(logits, temperature, top_k, top_p, rng_);
      cur[i] = next;
      out[i].push_back(next);
      ++pos[i];
      if (next == config_.eos_token_id) done[i] = 1;
    }
  }
  return out;
}

// ---- sampling on the host (non-greedy path) ---------------------------------------------------------------------------
void apply_repetition_penalty(std::vector<float>& logits, const std::vector<int>& generated, float penalty, size_t window) {
  if (penalty <= 1.0f || generated.empty()) return;
  const size_t first = generated.size() > window ? generated.size() - window : 0;
  for (size_t i = first; i < generated.size(); ++i) {  // an id that occurs twice in the window is penalised twice
    const int id = generated[i];
    if (id < 0 || (size_t)id >= logits.size()) continue;
    logits[id] = logits[id] > 0.0f ? logits[id] / penalty : logits[id] * penalty;
  }
}

void sampling_distribution(const std::vector<float>& logits, float temperature, int top_k, float top_p,
                           std::vector<int>* ids, std::vector<float>* probs) {
  const int V = (int)logits.size();
  if (top_k > V) top_k = V;
  if (top_k <= 0) top_k = V;
  const float mx = *std::max_element(logits.begin(), logits.end());
  const float inv_t = 1.0f / temperature;
  // softmax in double over (logit - max) / T, each probability then rounded to float (api.cpp:143-175)
  std::vector<double> e((size_t)V);
  double total = 0.0;
  for (int i = 0; i < V; ++i) {
    e[i] = std::exp((double)((logits[i] - mx) * inv_t));
    total += e[i];
  }
  std::vector<std::pair<float, int>> cand((size_t)V);
  for (int i = 0; i < V; ++i) cand[i] = {(float)(total > 0.0 ? e[i] / total : 1.0 / V), i};
  // descending by (probability, id): among equal probabilities the HIGHER id comes first, as std::greater on pairs does
  std::sort(cand.begin(), cand.end(), std::greater<std::pair<float, int>>());
  cand.resize((size_t)top_k);
  float cum = 0.0f;
  size_t keep = cand.size();
  for (size_t i = 0; i < cand.size(); ++i) {
    cum += cand[i].first;
    if (cum >= top_p) {
      keep = i + 1;
      break;
    }
  }
  cand.resize(keep);
  float kept = 0.0f;
  for (const auto& c : cand) kept += c.first;
  ids->clear();
  probs->clear();
  for (const auto& c : cand) {
    ids->push_back(c.second);
    probs->push_back(kept > 0.0f ? c.first / kept : 1.0f / (float)cand.size());
  }
}

int sample_top_k_top_p_temperature(const std::vector<float>& logits, float temperature, int top_k, float top_p,
                                   std::mt19937& rng) {
  if (logits.empty()) throw std::runtime_error("Cannot sample from empty logits.");
  if (temperature < 0.05f) return (int)(std::max_element(logits.begin(), logits.end()) - logits.begin());
  std::vector<int> ids;
  std::vector<float> probs;
  sampling_distribution(logits, temperature, top_k, top_p, &ids, &probs);
  std::discrete_distribution<int> dist(probs.begin(), probs.end());
  return ids[(size_t)dist(rng)];
}

}  // namespace tlhost
