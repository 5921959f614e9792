/* tlhost.h — C view of the C++ host layer (tinyllama.cpp_amd/host, libtlhost.so) that sits above tlhip.h.
 *
 * The host layer itself is C++ (tl_gguf.hpp: GGUF reader replacing gguf_parser.cpp:123-806 for this backend;
 * tl_session.hpp: tlhost::Session mirroring tinyllama::TinyLlamaSession, api.h:26-143, on token ids). These entry points
 * exist so that the repository's Python tests (and any other FFI) can drive it; a C++ host links the classes directly.
 * Every function returns 0 on success; tlh_last_error() has the text (the C++ layer throws std::runtime_error). */
#ifndef TLHOST_H
#define TLHOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* tlh_last_error(void);

/* ---- GGUF reader (no GPU involved) : load_gguf_meta, gguf_parser.cpp:123 ---- */
typedef struct tlh_gguf tlh_gguf;
int tlh_gguf_open(const char* path, tlh_gguf** out);
int tlh_gguf_close(tlh_gguf* g);
int tlh_gguf_counts(const tlh_gguf* g, uint64_t* n_tensors, uint64_t* n_metadata, uint64_t* alignment);
/* tensor i: name copied into name_buf (NUL-terminated, truncated to buf_len), ggml type, dims[4] (unused = 1), offset
 * relative to the data section, byte size, and a pointer to the bytes inside the mapping (valid until close) */
int tlh_gguf_tensor(const tlh_gguf* g, uint64_t i, char* name_buf, size_t buf_len, uint32_t* type, uint64_t dims[4],
                    uint32_t* n_dims, uint64_t* offset, uint64_t* nbytes, const void** data);
/* metadata lookups: *found = 0 when the key is absent or of another kind */
int tlh_gguf_meta_int(const tlh_gguf* g, const char* key, int64_t* value, int* found);
int tlh_gguf_meta_float(const tlh_gguf* g, const char* key, double* value, int* found);
int tlh_gguf_meta_string(const tlh_gguf* g, const char* key, char* buf, size_t buf_len, int* found);
/* the model configuration as parse_model_config_from_gguf derives it (model_config.cpp:149-221):
 * out[0..8] = hidden, intermediate, heads, kv heads, layers, vocab, context, bos id, eos id; eps_theta[0..1] */
int tlh_gguf_model_config(const tlh_gguf* g, int out[9], float eps_theta[2]);

/* ---- session : TinyLlamaSession (api.h:42-78) on token ids ---- */
typedef struct tlh_session tlh_session;
int tlh_session_open(const char* gguf_path, int device, int max_context, tlh_session** out);
int tlh_session_close(tlh_session* s);
int tlh_session_set_repetition_penalty(tlh_session* s, float penalty);
int tlh_session_seed(tlh_session* s, uint32_t seed);
/* generate(): writes up to out_cap generated ids (prompt excluded), *n_out = how many */
int tlh_session_generate(tlh_session* s, const int* prompt_ids, int n_prompt, int steps, float temperature, int top_k,
                         float top_p, int* out_ids, int out_cap, int* n_out);
/* TinyLlamaSession's max_batch_size constructor argument (api.h:39-46), settable after open; default 1 */
int tlh_session_set_max_batch(tlh_session* s, int max_batch_size);
/* generate_batch() (api.cpp:836-1142) on ids: prompt i = prompt_ids[offsets[i] .. offsets[i+1]) (offsets has n_prompts + 1
 * entries); out_ids is [n_prompts][steps] (row i holds out_lens[i] generated ids). Errors as the reference throws them:
 * empty prompt list, more prompts than max_batch_size. */
int tlh_session_generate_batch(tlh_session* s, const int* prompt_ids, const int* offsets, int n_prompts, int steps,
                               float temperature, int top_k, float top_p, int* out_ids, int* out_lens);
/* load statistics: tensors uploaded, quantized bytes taken from the mapping, seconds for open() */
int tlh_session_load_stats(const tlh_session* s, uint64_t* tensors, uint64_t* quantized_bytes, double* seconds);

/* ---- tokenizer : Tokenizer (tokenizer.h:27-235) for models that come as GGUF files; tl_tokenizer.hpp, no GPU involved.
 * PARITY UNPINNED (the reference's tokenizer.cpp needs Boost and cannot be built in this image, and the reference holds no
 * tokenizer vectors): tests compare with oracle/tokenizer_oracle.py. Text is bytes + length both ways (a decoded byte
 * token may be 0); on output *n_out is the full length, of which at most `cap` bytes / ids were written. ---- */
typedef struct tlh_tokenizer tlh_tokenizer;
/* Tokenizer(gguf_data, config) (tokenizer.cpp:516). family_override: 0 = as model_config.cpp:221-253 decides from the
 * file; 1 = Llama-2 / TinyLlama SentencePiece, 2 = Llama-3 tiktoken, 3 = word level */
int tlh_tokenizer_from_gguf(const tlh_gguf* g, int family_override, tlh_tokenizer** out);
int tlh_tokenizer_close(tlh_tokenizer* t);
/* out[0..5] = family, vocabulary size, bos, eos, unk, pad id */
int tlh_tokenizer_info(const tlh_tokenizer* t, int out[6]);
/* Tokenizer::encode (tokenizer.cpp:1141) */
int tlh_tokenizer_encode(const tlh_tokenizer* t, const char* text, size_t text_len, int add_bos, int add_eos, int* ids,
                         int cap, int* n_out);
/* Tokenizer::decode (tokenizer.cpp:1412) */
int tlh_tokenizer_decode(const tlh_tokenizer* t, const int* ids, int n_ids, int skip_special_tokens, char* buf, size_t cap,
                         size_t* n_out);
/* Tokenizer::apply_chat_template (tokenizer.cpp:1663) */
int tlh_tokenizer_apply_chat_template(const tlh_tokenizer* t, const char* user, size_t user_len, const char* system,
                                      size_t system_len, char* buf, size_t cap, size_t* n_out);
/* the prompt TinyLlamaSession::generate encodes (api.cpp:505-572) */
int tlh_tokenizer_format_prompt(const tlh_tokenizer* t, const char* user, size_t user_len, const char* system,
                                size_t system_len, int apply_q_a_format, char* buf, size_t cap, size_t* n_out);
/* TinyLlamaSession::generate on text (api.h:59-66): format, encode, generate, decode. Fails when the file has no tokenizer. */
int tlh_session_generate_text(tlh_session* s, const char* prompt, size_t prompt_len, int steps, float temperature, int top_k,
                              float top_p, const char* system_prompt, size_t system_len, int apply_q_a_format, char* buf,
                              size_t cap, size_t* n_out);

/* TinyLlamaSession::generate_batch on text (api.h:68-78, api.cpp:836-1142): prompt i = prompts[offsets[i] .. offsets[i+1]);
 * the results come back concatenated in buf with their end offsets in out_offsets[n_prompts + 1] (out_offsets[0] = 0);
 * *n_out = total bytes, of which at most cap were written. */
int tlh_session_generate_batch_text(tlh_session* s, const char* prompts, const size_t* offsets, int n_prompts, int steps,
                                    float temperature, int top_k, float top_p, const char* system_prompt, size_t system_len,
                                    int apply_q_a_format, char* buf, size_t cap, size_t* out_offsets, size_t* n_out);

/* ---- host-side sampler pieces (api.cpp:86-97, 126-216), exposed for parity tests; no GPU involved ---- */
int tlh_apply_repetition_penalty(float* logits, int n, const int* generated, int n_generated, float penalty);
/* candidate ids and renormalised probabilities the sampler draws from; *n_out <= cap entries written */
int tlh_sampling_distribution(const float* logits, int n, float temperature, int top_k, float top_p, int* ids,
                              float* probs, int cap, int* n_out);
int tlh_sample(const float* logits, int n, float temperature, int top_k, float top_p, uint32_t seed, int* id_out);

#ifdef __cplusplus
}
#endif
#endif
