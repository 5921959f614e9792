/*
 * tl_oracle.h — CPU oracle for the quantized transformer forward pass.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C restatement of the reference's CPU
 * algorithm (JohnnyTeutonic/tinyllama.cpp); it exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the reference arithmetic on a box where
 * /root/reference does not exist. Nothing under tinyllama.cpp_amd/ may call into it.
 *
 * Parity pin: every primitive here is checked bit-for-bit (or to 0 ulp where libm is shared)
 * against the reference's own code compiled from /root/reference (oracle/_ref, "Tier-1":
 * quantization.cpp + utils.cpp + logger.cpp + gguf_parser.cpp) by tests/test_oracle_vs_ref.py
 * and against committed fixtures generated from that build (tests/golden/, made by
 * oracle/make_golden.py). The layer composition (model.cpp:536-807, cpu_batch_processor.cpp,
 * api.cpp:585-813) is restated from source; TinyLlamaModel itself is unbuildable here
 * (model.h:6 needs <nlohmann/json.hpp>, absent from the image's include paths), so the
 * end-to-end composition is pinned only through its primitives.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (NO -march=native: FMA contraction changes the
 * dequantized bit patterns).
 */
#ifndef TL_ORACLE_H
#define TL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids as the reference uses them (ggml_types.h:16-46) */
enum {
  TLO_F32 = 0,
  TLO_Q8_0 = 8,
  TLO_Q4_K = 12,
  TLO_Q6_K = 14,
  TLO_Q8_K = 15
};

#define TLO_QK_K 256
#define TLO_QK8_0 32

/* ---- scalar / block level ------------------------------------------------------------ */
float tlo_fp16_to_fp32(uint16_t h, int is_gguf_scale_field); /* quantization.cpp:47-90 */
uint16_t tlo_fp32_to_fp16(float f);                          /* quantization.cpp:92-136 */

size_t tlo_type_block_bytes(int type); /* quantization.cpp:646-686 */
size_t tlo_type_block_elems(int type); /* quantization.cpp:688-717 (Q8_K: 256 here) */

/* nblocks blocks -> nblocks*block_elems floats */
void tlo_dequantize(int type, const void* blocks, size_t nblocks, float* out);
/* the same values in the reference's whole-tensor structure (zero-filled fresh vector, then the serial block loop) */
void tlo_dequantize_whole_tensor(int type, const void* blocks, size_t nblocks, float* out);
/* quantize_fp32_to_q8_K, quantization.cpp:719-770 (290-byte blocks, fp16 d) */
void tlo_quantize_q8_k(const float* x, size_t n, void* blocks_out);

/* ---- mat-vec / mat-mul --------------------------------------------------------------- */
/* fused dequant + Kahan-double dot: utils.cpp:293-347, 784-888, 399-426 */
void tlo_matvec_q(int type, const void* blocks, const float* x, float* y, int rows, int cols);
/* utils.cpp:349-397 */
void tlo_matvec_f32(const float* w, const float* x, float* y, int rows, int cols);
/* utils.cpp:730-782: Y[T][rows] = X[T][cols] . W^T, per-token Kahan-double dots */
void tlo_matmul_q_batch(int type, const void* blocks, const float* x, float* y, int T, int rows,
                        int cols);

/* ---- glue ops ------------------------------------------------------------------------ */
void tlo_rmsnorm_vector(const float* x, const float* w, float* out, int n, float eps); /* utils.cpp:648-673 */
void tlo_rmsnorm_batch(const float* x, const float* w, float* out, int T, int n, float eps); /* utils.cpp:613-646 */
/* gpu_initialization.cpp:48-58: table[pos][i] = (cos, sin), i < head_dim/2 */
void tlo_rope_table(float* table, int max_pos, int head_dim, float theta);
/* utils.cpp:428-489 */
void tlo_rope_apply(float* x, int num_heads, int head_dim, int pos, const float* table,
                    int max_pos, int adjacent);
void tlo_softmax(const float* x, float* out, int n);           /* utils.cpp:675-698 */
void tlo_silu(const float* x, float* out, int n);              /* utils.cpp:700-707 */
/* decode form model.cpp:734-735 (batch_form=0) / batch form cpu_batch_processor.cpp:252-256 */
void tlo_swiglu(const float* gate, const float* up, float* out, int n, int batch_form);
/* model.cpp:632-663; caches are [pos][n_kv_heads][head_dim] */
void tlo_attention_decode(const float* q, const float* kcache, const float* vcache, float* out,
                          int n_heads, int n_kv_heads, int head_dim, int pos);
/* cpu_attention.cpp:91-201 */
void tlo_attention_batch(const float* q, const float* kcache, const float* vcache, float* out,
                         int T, int start_pos, int n_heads, int n_kv_heads, int head_dim);

/* ---- sampling glue (api.cpp:73-97, 126-136) ------------------------------------------ */
int tlo_argmax(const float* v, int n);
void tlo_apply_repetition_penalty(float* logits, int vocab, const int* generated, int n_generated,
                                  float penalty, int window);

/* sample_top_k_top_p_temperature (api.cpp:126-216) without the draw: candidate ids in draw order + renormalised
 * probabilities (room for `vocab` entries each); returns their count. temperature < 0.05: the greedy branch. */
int tlo_sample_candidates(const float* logits, int vocab, float temperature, int top_k, float top_p, int* ids,
                          float* probs);
/* std::discrete_distribution's draw for a uniform u in [0,1) (libstdc++ bits/random.tcc:2656-2713) */
int tlo_discrete_draw(const float* probs, int n, double u);

/* ---- model-level --------------------------------------------------------------------- */
typedef struct tlo_model tlo_model;

/* per-layer tensor kinds */
enum {
  TLO_T_ATTN_Q = 0,
  TLO_T_ATTN_K,
  TLO_T_ATTN_V,
  TLO_T_ATTN_O,
  TLO_T_FFN_GATE,
  TLO_T_FFN_UP,
  TLO_T_FFN_DOWN,
  TLO_T_ATTN_NORM, /* F32 */
  TLO_T_FFN_NORM,  /* F32 */
  TLO_T_TOKEN_EMBD, /* layer = -1 */
  TLO_T_OUTPUT_NORM,
  TLO_T_OUTPUT
};

tlo_model* tlo_model_create(int hidden, int inter, int n_heads, int n_kv_heads, int n_layers,
                            int vocab, int max_pos, float eps, float rope_theta);
void tlo_model_free(tlo_model* m);
/* data is BORROWED (must outlive the model); rows = output features, cols = input features */
int tlo_model_set_tensor(tlo_model* m, int kind, int layer, int type, const void* data, int rows,
                         int cols);
/* ref_structure != 0: dequantize each whole tensor to f32 first, then the f32 mat-vec, and evict
 * layer l-2's f32 copies after layer l (weight_management.cpp:10-196, model.cpp:771-777) — bit-identical
 * output, used as the timed "reference OpenMP CPU path" */
void tlo_model_set_ref_structure(tlo_model* m, int ref_structure);
void tlo_model_reset_kv(tlo_model* m); /* KVCache::clear_data, model.h:182-197 */
void tlo_lookup_embedding(tlo_model* m, int token, float* out); /* model_utils.cpp:11-182 */
/* TinyLlamaModel::forward, model.cpp:536-807: x[hidden] in/out, logits[vocab] out (may be NULL).
 * n_layers_limit < 0: all layers */
void tlo_forward(tlo_model* m, float* x, int pos, float* logits, int n_layers_limit);
/* forward_cpu_batch + forward_cpu_logits_batch, cpu_batch_processor.cpp:11-281, model.cpp:1094-1156.
 * x[T][hidden] in/out; logits[T][vocab] (may be NULL) */
void tlo_forward_batch(tlo_model* m, float* x, int T, int start_pos, float* logits);
/* generate() loop on token ids (api.cpp:585-813); greedy. Returns #generated.
 * logits_dump (may be NULL): [max_new][vocab], the (pre-penalty) logits at each sampled position */
int tlo_generate(tlo_model* m, const int* prompt, int n_prompt, int max_new, float penalty,
                 int eos_id, int* out_ids, float* logits_dump);
const float* tlo_model_kcache(tlo_model* m, int layer);
const float* tlo_model_vcache(tlo_model* m, int layer);

#ifdef __cplusplus
}
#endif
#endif
