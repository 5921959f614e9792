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
