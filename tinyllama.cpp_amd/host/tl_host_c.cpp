// tl_host_c.cpp — extern "C" view of the C++ host layer (include/tlhost.h).
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "tl_session.hpp"
#include "tlhost.h"

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
  } catch (...) {
    g_err = "unknown C++ exception";
  }
  return 1;
}
int fail(const char* msg) {
  g_err = msg;
  return 2;
}
void copy_str(const std::string& s, char* buf, size_t n) {
  if (!buf || n == 0) return;
  const size_t k = s.size() < n - 1 ? s.size() : n - 1;
  memcpy(buf, s.data(), k);
  buf[k] = 0;
}
}  // namespace

struct tlh_gguf {
  tlhost::GGUFFile f;
};
struct tlh_session {
  tlhost::Session* s = nullptr;
};
struct tlh_tokenizer {
  tlhost::Tokenizer* t = nullptr;
};
namespace {
std::string str_arg(const char* p, size_t n) { return p && n ? std::string(p, n) : std::string(); }
void copy_bytes(const std::string& s, char* buf, size_t cap, size_t* n_out) {
  if (buf && cap) memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
  *n_out = s.size();
}
}  // namespace

extern "C" {

const char* tlh_last_error(void) { return g_err.c_str(); }

int tlh_gguf_open(const char* path, tlh_gguf** out) {
  if (!path || !out) return fail("tlh_gguf_open: null argument");
  tlh_gguf* g = new tlh_gguf();
  const int rc = guarded([&] { g->f.open(path); });
  if (rc != 0) {
    delete g;
    return rc;
  }
  *out = g;
  return 0;
}
int tlh_gguf_close(tlh_gguf* g) {
  delete g;
  return 0;
}
int tlh_gguf_counts(const tlh_gguf* g, uint64_t* nt, uint64_t* nm, uint64_t* al) {
  if (!g) return fail("tlh_gguf_counts: null handle");
  if (nt) *nt = g->f.tensors().size();
  if (nm) *nm = g->f.metadata().size();
  if (al) *al = g->f.alignment();
  return 0;
}
int tlh_gguf_tensor(const tlh_gguf* g, uint64_t i, char* name_buf, size_t buf_len, uint32_t* type, uint64_t dims[4],
                    uint32_t* n_dims, uint64_t* offset, uint64_t* nbytes, const void** data) {
  if (!g) return fail("tlh_gguf_tensor: null handle");
  if (i >= g->f.tensors().size()) return fail("tlh_gguf_tensor: index out of range");
  const tlhost::GGUFTensor& t = g->f.tensors()[(size_t)i];
  copy_str(t.name, name_buf, buf_len);
  if (type) *type = t.type;
  if (dims)
    for (size_t d = 0; d < 4; ++d) dims[d] = d < t.dims.size() ? t.dims[d] : 1;
  if (n_dims) *n_dims = (uint32_t)t.dims.size();
  if (offset) *offset = t.offset;
  if (nbytes) *nbytes = t.nbytes;
  if (data) *data = t.data;
  return 0;
}
int tlh_gguf_meta_int(const tlh_gguf* g, const char* key, int64_t* value, int* found) {
  if (!g || !key || !value || !found) return fail("tlh_gguf_meta_int: null argument");
  const tlhost::GGUFValue* v = g->f.meta(key);
  *found = v && v->as_integer(value);
  return 0;
}
int tlh_gguf_meta_float(const tlh_gguf* g, const char* key, double* value, int* found) {
  if (!g || !key || !value || !found) return fail("tlh_gguf_meta_float: null argument");
  const tlhost::GGUFValue* v = g->f.meta(key);
  *found = v && v->kind == tlhost::GGUFValue::FLOAT && v->as_double(value);
  return 0;
}
int tlh_gguf_meta_string(const tlh_gguf* g, const char* key, char* buf, size_t buf_len, int* found) {
  if (!g || !key || !found) return fail("tlh_gguf_meta_string: null argument");
  const tlhost::GGUFValue* v = g->f.meta(key);
  *found = v && v->kind == tlhost::GGUFValue::STRING;
  if (*found) copy_str(v->s, buf, buf_len);
  return 0;
}
int tlh_gguf_model_config(const tlh_gguf* g, int out[9], float eps_theta[2]) {
  if (!g || !out || !eps_theta) return fail("tlh_gguf_model_config: null argument");
  return guarded([&] {
    const tlhost::ModelConfig c = tlhost::config_from_gguf(g->f);
    const int v[9] = {c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.num_hidden_layers,
                      c.vocab_size, c.max_position_embeddings, c.bos_token_id, c.eos_token_id};
    memcpy(out, v, sizeof(v));
    eps_theta[0] = c.rms_norm_eps;
    eps_theta[1] = c.rope_theta;
  });
}

int tlh_session_open(const char* path, int device, int max_context, tlh_session** out) {
  if (!path || !out) return fail("tlh_session_open: null argument");
  tlh_session* h = new tlh_session();
  const int rc = guarded([&] { h->s = new tlhost::Session(path, device, max_context); });
  if (rc != 0) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}
int tlh_session_close(tlh_session* s) {
  if (s) delete s->s;
  delete s;
  return 0;
}
int tlh_session_set_repetition_penalty(tlh_session* s, float p) {
  if (!s) return fail("tlh_session_set_repetition_penalty: null handle");
  s->s->set_repetition_penalty(p);
  return 0;
}
int tlh_session_seed(tlh_session* s, uint32_t seed) {
  if (!s) return fail("tlh_session_seed: null handle");
  s->s->seed(seed);
  return 0;
}
int tlh_session_generate(tlh_session* s, const int* prompt, int n_prompt, int steps, float temperature, int top_k,
                         float top_p, int* out_ids, int out_cap, int* n_out) {
  if (!s || !n_out || (n_prompt > 0 && !prompt) || n_prompt < 0) return fail("tlh_session_generate: bad argument");
  return guarded([&] {
    const std::vector<int> ids = s->s->generate(std::vector<int>(prompt, prompt + n_prompt), steps, temperature, top_k, top_p);
    const int n = (int)ids.size() < out_cap ? (int)ids.size() : out_cap;
    if (out_ids) memcpy(out_ids, ids.data(), sizeof(int) * (size_t)n);
    *n_out = n;
  });
}
int tlh_session_set_max_batch(tlh_session* s, int max_batch_size) {
  if (!s) return fail("tlh_session_set_max_batch: null handle");
  return guarded([&] { s->s->set_max_batch_size(max_batch_size); });
}
int tlh_session_generate_batch(tlh_session* s, const int* prompt_ids, const int* offsets, int n_prompts, int steps,
                               float temperature, int top_k, float top_p, int* out_ids, int* out_lens) {
  if (!s || n_prompts < 0 || (n_prompts > 0 && (!prompt_ids || !offsets)) || !out_lens || (steps > 0 && !out_ids))
    return fail("tlh_session_generate_batch: bad argument");
  return guarded([&] {
    std::vector<std::vector<int>> prompts((size_t)n_prompts);
    for (int i = 0; i < n_prompts; ++i) {
      if (offsets[i + 1] < offsets[i]) throw std::runtime_error("tlh_session_generate_batch: offsets must not decrease");
      prompts[i].assign(prompt_ids + offsets[i], prompt_ids + offsets[i + 1]);
    }
    const std::vector<std::vector<int>> out = s->s->generate_batch(prompts, steps, temperature, top_k, top_p);
    for (int i = 0; i < n_prompts; ++i) {
      const int n = (int)out[i].size() < steps ? (int)out[i].size() : steps;
      out_lens[i] = n;
      if (n > 0) memcpy(out_ids + (size_t)i * steps, out[i].data(), sizeof(int) * (size_t)n);
    }
  });
}
int tlh_session_load_stats(const tlh_session* s, uint64_t* tensors, uint64_t* qbytes, double* seconds) {
  if (!s) return fail("tlh_session_load_stats: null handle");
  const tlhost::LoadStats& st = s->s->load_stats();
  if (tensors) *tensors = st.tensors;
  if (qbytes) *qbytes = st.quantized_bytes;
  if (seconds) *seconds = st.seconds;
  return 0;
}

int tlh_tokenizer_from_gguf(const tlh_gguf* g, int family_override, tlh_tokenizer** out) {
  if (!g || !out || family_override < 0 || family_override > 3) return fail("tlh_tokenizer_from_gguf: bad argument");
  tlh_tokenizer* h = new tlh_tokenizer();
  const int rc = guarded([&] { h->t = new tlhost::Tokenizer(g->f, (tlhost::TokenizerFamily)family_override); });
  if (rc != 0) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}
int tlh_tokenizer_close(tlh_tokenizer* t) {
  if (t) delete t->t;
  delete t;
  return 0;
}
int tlh_tokenizer_info(const tlh_tokenizer* t, int out[6]) {
  if (!t || !out) return fail("tlh_tokenizer_info: null argument");
  const int v[6] = {(int)t->t->family(), t->t->vocab_size(),   t->t->bos_token_id(),
                    t->t->eos_token_id(), t->t->unk_token_id(), t->t->pad_token_id()};
  memcpy(out, v, sizeof(v));
  return 0;
}
int tlh_tokenizer_encode(const tlh_tokenizer* t, const char* text, size_t text_len, int add_bos, int add_eos, int* ids,
                         int cap, int* n_out) {
  if (!t || !n_out || (text_len > 0 && !text) || cap < 0) return fail("tlh_tokenizer_encode: bad argument");
  return guarded([&] {
    const std::vector<int> v = t->t->encode(str_arg(text, text_len), add_bos != 0, add_eos != 0);
    const size_t k = v.size() < (size_t)cap ? v.size() : (size_t)cap;
    if (ids && k) memcpy(ids, v.data(), sizeof(int) * k);
    *n_out = (int)v.size();
  });
}
int tlh_tokenizer_decode(const tlh_tokenizer* t, const int* ids, int n_ids, int skip_special, char* buf, size_t cap,
                         size_t* n_out) {
  if (!t || !n_out || n_ids < 0 || (n_ids > 0 && !ids)) return fail("tlh_tokenizer_decode: bad argument");
  return guarded([&] { copy_bytes(t->t->decode(std::vector<int>(ids, ids + n_ids), skip_special != 0), buf, cap, n_out); });
}
int tlh_tokenizer_apply_chat_template(const tlh_tokenizer* t, const char* user, size_t user_len, const char* system,
                                      size_t system_len, char* buf, size_t cap, size_t* n_out) {
  if (!t || !n_out) return fail("tlh_tokenizer_apply_chat_template: null argument");
  return guarded([&] { copy_bytes(t->t->apply_chat_template(str_arg(user, user_len), str_arg(system, system_len)), buf, cap, n_out); });
}
int tlh_tokenizer_format_prompt(const tlh_tokenizer* t, const char* user, size_t user_len, const char* system,
                                size_t system_len, int qa, char* buf, size_t cap, size_t* n_out) {
  if (!t || !n_out) return fail("tlh_tokenizer_format_prompt: null argument");
  return guarded([&] { copy_bytes(t->t->format_prompt(str_arg(user, user_len), str_arg(system, system_len), qa != 0), buf, cap, n_out); });
}
int tlh_session_generate_text(tlh_session* s, const char* prompt, size_t prompt_len, int steps, float temperature, int top_k,
                              float top_p, const char* system_prompt, size_t system_len, int qa, char* buf, size_t cap,
                              size_t* n_out) {
  if (!s || !n_out) return fail("tlh_session_generate_text: null argument");
  return guarded([&] {
    copy_bytes(s->s->generate(str_arg(prompt, prompt_len), steps, temperature, top_k, top_p, str_arg(system_prompt, system_len), qa != 0),
               buf, cap, n_out);
  });
}

int tlh_session_generate_batch_text(tlh_session* s, const char* prompts, const size_t* offsets, int n_prompts, int steps,
                                    float temperature, int top_k, float top_p, const char* system_prompt, size_t system_len,
                                    int qa, char* buf, size_t cap, size_t* out_offsets, size_t* n_out) {
  if (!s || !n_out || !out_offsets || n_prompts < 0 || (n_prompts > 0 && !offsets))
    return fail("tlh_session_generate_batch_text: bad argument");
  return guarded([&] {
    std::vector<std::string> in((size_t)n_prompts);
    for (int i = 0; i < n_prompts; ++i) {
      if (offsets[i + 1] < offsets[i]) throw std::runtime_error("tlh_session_generate_batch_text: offsets must not decrease");
      in[i] = str_arg(prompts ? prompts + offsets[i] : nullptr, offsets[i + 1] - offsets[i]);
    }
    const std::vector<std::string> out =
        s->s->generate_batch(in, steps, temperature, top_k, top_p, str_arg(system_prompt, system_len), qa != 0);
    std::string all;
    out_offsets[0] = 0;
    for (int i = 0; i < n_prompts; ++i) {
      all += out[i];
      out_offsets[i + 1] = all.size();
    }
    copy_bytes(all, buf, cap, n_out);
  });
}

int tlh_apply_repetition_penalty(float* logits, int n, const int* generated, int n_generated, float penalty) {
  if (!logits || n < 0 || n_generated < 0 || (n_generated > 0 && !generated)) return fail("tlh_apply_repetition_penalty: bad argument");
  return guarded([&] {
    std::vector<float> l(logits, logits + n);
    tlhost::apply_repetition_penalty(l, std::vector<int>(generated, generated + n_generated), penalty);
    memcpy(logits, l.data(), sizeof(float) * (size_t)n);
  });
}
int tlh_sampling_distribution(const float* logits, int n, float temperature, int top_k, float top_p, int* ids, float* probs,
                              int cap, int* n_out) {
  if (!logits || n <= 0 || !ids || !probs || !n_out) return fail("tlh_sampling_distribution: bad argument");
  return guarded([&] {
    std::vector<int> i;
    std::vector<float> p;
    tlhost::sampling_distribution(std::vector<float>(logits, logits + n), temperature, top_k, top_p, &i, &p);
    const int k = (int)i.size() < cap ? (int)i.size() : cap;
    memcpy(ids, i.data(), sizeof(int) * (size_t)k);
    memcpy(probs, p.data(), sizeof(float) * (size_t)k);
    *n_out = k;
  });
}
int tlh_sample(const float* logits, int n, float temperature, int top_k, float top_p, uint32_t seed, int* id_out) {
  if (!logits || n < 0 || !id_out) return fail("tlh_sample: bad argument");
  return guarded([&] {
    std::mt19937 rng(seed);
    *id_out = tlhost::sample_top_k_top_p_temperature(std::vector<float>(logits, logits + n), temperature, top_k, top_p, rng);
  });
}

}  // extern "C"
