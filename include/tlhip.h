/*
 * tlhip.h — C-ABI of libtlhip.so, the MI355X (gfx950) backend that replaces tinyllama.cpp's CUDA/cuBLAS
 * backend (reference: cuda_kernels.h:104-643 + the raw cudaMalloc/cudaMemcpy/cublas* calls inlined in
 * model.cpp, gpu_initialization.cpp, weight_management.cpp, kv_cache.cpp, gpu_memory_optimization.cpp).
 *
 * Conventions
 *   - plain C: pointers and sizes only. `void* stream` is a hipStream_t (NULL = the default stream),
 *     the counterpart of the reference's `cudaStream_t stream = 0` arguments.
 *   - every function returns int: 0 = ok, non-zero = error; tl_last_error() gives the text. The C++ host
 *     layer (tinyllama.cpp_amd/host/tl_session.cpp:14-16, `check`) turns non-zero into std::runtime_error, which is the
 *     reference's convention (gpuErrchk, cuda_kernels.h:66-82). Argument-validation failures return an
 *     error instead of the reference's log-and-return-silently (cuda_kernels.cu:1335-1351).
 *   - `*_dev` pointers are device pointers BORROWED from the caller; `*_host` are host pointers borrowed
 *     for the duration of the call. tl_qweight / tl_model handles own their device memory.
 *   - one host thread per handle, no re-entrancy (the reference's threading model).
 *
 * The big difference from the reference backend: weights stay QUANTIZED on the device. The reference
 * dequantizes every GPU layer to FP32 on the host and calls cublasSgemm with N=1
 * (weight_management.cpp:199-435, cuda_kernels.cu:291-314); here raw GGUF super-blocks are uploaded once,
 * re-laid-out into 16-byte-aligned planes (tl_qweight) and consumed by block-dequant GEMV kernels.
 */
#ifndef TLHIP_H
#define TLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml tensor type ids (reference ggml_types.h:21-42) */
enum tl_type { TL_F32 = 0, TL_Q8_0 = 8, TL_Q4_K = 12, TL_Q6_K = 14, TL_Q8_K = 15,
               TL_BF16 = 30 /* tl_model_set_tensor only: widened to F32 at upload (exact) */ };

typedef struct tl_qweight tl_qweight; /* one quantized [rows x cols] matrix resident in HBM */
typedef struct tl_model tl_model;     /* a whole model + KV cache + workspaces + decode graph */

/* ---- runtime: replaces the inline cudaSetDevice/cudaMalloc/cudaMemcpy/cudaFree/cudaStream*/
/*      cublasCreate sites (215 of them) ---------------------------------------- */
const char* tl_last_error(void);
/* hipSetDevice + sanity (gfx950 expected) + this device's address of the static no-allocation workspace
 * (tl_attention_noalloc); call once per device the process uses, with that device's ordinal */
int tl_init(int device_ordinal);
/* frees the workspaces that the operator-level entry points (tl_attention, tl_matmul_q, tl_sample_top_k_top_p) grow
 * lazily per device; optional (the process exit frees them too). Those entry points share one workspace per device:
 * calls that overlap on one device must be serialised by the caller (one host thread, as in the reference). */
int tl_shutdown(void);
int tl_device_count(int* count_out);
int tl_malloc(void** dev_out, size_t bytes);  /* cudaMalloc sites e.g. gpu_initialization.cpp:91-531 */
int tl_free(void* dev);
int tl_memset(void* dev, int value, size_t bytes, void* stream);
int tl_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int tl_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int tl_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int tl_stream_create(void** stream_out);
int tl_stream_destroy(void* stream);
int tl_stream_sync(void* stream);             /* cudaStreamSynchronize, model.cpp:1085 */
int tl_device_sync(void);

/* ---- quantized weights: replaces ensure_f32_concatenated_weights_loaded (weight_management.cpp:199-435)
 *      and the BF16 embedding/lm_head uploads (gpu_initialization.cpp:182-342) --------------------- */
/* host_blocks: rows*cols/block_elems raw GGUF blocks (quantization.h:57-124), row-major by output row.
 * cols must be a multiple of 256 for the K-quants, of 32 for Q8_0 and F32. */
int tl_qweight_upload(int type, const void* host_blocks, int rows, int cols, tl_qweight** out, void* stream);
int tl_qweight_free(tl_qweight* w);
int tl_qweight_info(const tl_qweight* w, int* type, int* rows, int* cols, size_t* device_bytes);
/* Optional: keeps an expanded copy of the matrix for the batched paths (tl_matmul_q with num_tokens > 64, prefill, the
 * multi-sequence decode step): the integers as one bf16 plane + FP32 scale words, 2.375 bytes per weight, from which the
 * matrix cores are fed by LDS-DMA instead of unpacking the quantized planes in every workgroup. Results do not change by
 * a bit. The counterpart of the reference's FP32 expansion, ensure_f32_concatenated_weights_loaded
 * (weight_management.cpp:199-435; 4 bytes per weight there). tl_model_finalize does this for a model's matrices when the
 * copy is at most a quarter of the device's memory (TL_PREFILL_PLANES=0 / 1: never / always). tl_qweight_free drops it.
 * *expanded_bytes (may be NULL): device bytes of the copy (0: this format has none -- F32 / BF16 matrices). */
int tl_qweight_expand(tl_qweight* w, size_t* expanded_bytes, void* stream);

/* Bit-exact dequantization on the device (parity hook for dequantize_q4_k_m / dequantize_q6_k /
 * dequantize_q8_0_block / dequantize_q8_k, quantization.cpp:171-266, 1009-1052): rows [row0, row0+nrows)
 * -> out_dev[nrows*cols] f32. Also what lookup_embedding (model_utils.cpp:11-182) becomes for one row. */
int tl_dequantize_rows(const tl_qweight* w, int row0, int nrows, float* out_dev, void* stream);
/* quantize_fp32_to_q8_K (quantization.cpp:719-770) on the device, bit-exact: n (multiple of 256) FP32 values ->
 * n/256 GGUF Q8_K blocks of 290 B {d:f16, qs[256]:i8, bsums[16]:i16} at blocks_dev */
int tl_quantize_q8_k(const float* x_dev, size_t n, void* blocks_dev, void* stream);

/* y[rows] = W x : replaces matvec_f32_f32_cuda (cuda_kernels.h:171) / matvec_bf16_f32_cuda (:424), on the
 * quantized weights directly. FP32 accumulate; tolerance vs the reference's Kahan-double CPU dot is stated
 * in tests/test_gpu_parity.py (test_gemv_golden, test_gemv_vs_oracle_model_shapes). */
int tl_matvec_q(const tl_qweight* w, const float* x_dev, float* y_dev, void* stream);
/* Y[T][rows] = X[T][cols] W^T : replaces gemm_f32_f32_cuda / gemm_f32_to_bf16_f32_cuda via
 * smart_gemm_batch_cuda (model.cpp:2141-2249) for batched prefill. */
int tl_matmul_q(const tl_qweight* w, const float* x_dev, float* y_dev, int num_tokens, void* stream);
/* bf16 MFMAs per FP32 product in tl_matmul_q (num_tokens >= 8) and, by default, in tl_model_prefill: both operands
 * are split into bf16 planes and a product is assembled from 1 (plain bf16), 3 or 6 (default; FP32-grade) MFMAs. */
int tl_gemm_set_terms(int terms);

/* ---- glue operators (same argument meaning as the cuda_kernels.h wrappers they replace) ------- */
/* rmsnorm_vector_cuda(_optimized) cuda_kernels.h:104,569: out = x*rsqrt(mean(x^2)+max(eps,1e-5))*w
 * (the eps floor is the CPU path's, utils.cpp:666-667, which is the parity target) */
int tl_rmsnorm_vector(const float* x_dev, const float* weight_dev, float* out_dev, int n, float eps, void* stream);
/* rmsnorm_batch_cuda cuda_kernels.h:124 (plain eps, utils.cpp:640) */
int tl_rmsnorm_batch(float* out_dev, const float* in_dev, const float* weight_dev, int num_tokens, int hidden_size,
                     float eps, void* stream);
/* RoPE table of gpu_initialization.cpp:48-58, computed on the host (glibc cosf/sinf/powf, exactly as the
 * reference) and uploaded: table_dev[max_pos][head_dim/2][2] */
int tl_rope_table_upload(float* table_dev, int max_pos, int head_dim, float theta, void* stream);
/* rope_cuda cuda_kernels.h:239 */
int tl_rope(float* vec_dev, int num_heads, int head_dim, const float* freqs_cis_dev, int pos, int max_pos,
            int use_adjacent_pairing, void* stream);
/* rope_batch_cuda cuda_kernels.h:490 */
int tl_rope_batch(float* q_batch_dev, float* k_batch_dev, const float* freqs_cis_dev, int num_tokens,
                  int num_q_heads, int num_kv_heads, int head_dim, int start_pos, int max_pos,
                  int use_adjacent_pairing, void* stream);
/* update_kv_cache_cuda cuda_kernels.h:344 — cache layout [max_seq_len][num_kv_heads][head_dim] */
int tl_update_kv_cache(float* cache_base_dev, const float* kv_vector_dev, int pos, int kv_head_idx,
                       int max_seq_len, int num_kv_heads, int head_dim, void* stream);
/* update_kv_cache_batch_cuda cuda_kernels.h:537 */
int tl_update_kv_cache_batch(float* cache_base_dev, const float* kv_batch_dev, int start_pos, int num_tokens,
                             int num_kv_heads, int head_dim, int cache_max_seq_len, void* stream);
/* attention_cuda / attention_cuda_optimized cuda_kernels.h:259,607 (math of model.cpp:632-663) */
int tl_attention(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev, float* out_dev,
                 int num_heads, int current_seq_len, int head_dim, float scale, int cache_max_seq_len,
                 int cache_num_kv_heads, void* stream);
/* tl_attention for callers that cannot allocate. No counterpart in the reference: it is for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc, where tl_attention's lazily grown workspace is not
 * an option. Same arguments, same kernels and bit-identical results; the split partials live in a static device array
 * of 64 heads x 32 splits x (128 + 4) floats (1.03 MiB, held by every GPU the library runs on until the process ends).
 * Calls hipMalloc / hipFree never; needs tl_init(device) to have run for the current device (it looks up that
 * device's copy of the array); rejects num_heads * (head_dim + 4) beyond 64 * 132 with an error. All calls on one GPU
 * share the array: calls that overlap on one GPU (two streams) race. */
int tl_attention_noalloc(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev, float* out_dev,
                         int num_heads, int current_seq_len, int head_dim, float scale, int cache_max_seq_len,
                         int cache_num_kv_heads, void* stream);
/* sample_top_k_top_p_temperature (api.cpp:126-216) as a device operator (the reference only has it on the host):
 * logits_dev[n] is not modified; u in [0,1) is the caller's uniform variate; *id_dev = the drawn id; optional outputs
 * (NULL to skip): the candidate ids and renormalised probabilities in draw order (capacity >= min(top_k, n)) and their
 * count. Effective top_k <= 1024. */
int tl_sample_top_k_top_p(const float* logits_dev, int n, float temperature, int top_k, float top_p, float u,
                          int* id_dev, int* cand_ids_dev, float* cand_probs_dev, int* n_cand_dev, void* stream);
/* attention_batch_prefill_cuda cuda_kernels.h:498 (math of cpu_attention.cpp:91-201), causal from the cache */
int tl_attention_batch_prefill(const float* q_batch_dev, const float* k_cache_dev, const float* v_cache_dev,
                               float* out_batch_dev, int num_tokens, int start_pos, int cache_max_seq_len,
                               int num_q_heads, int num_kv_heads, int head_dim, float scale, void* stream);
/* add_residual_cuda / add_vectors_cuda / add_residual_batch_cuda cuda_kernels.h:317,329,517 */
int tl_add_residual(const float* a_dev, const float* b_dev, float* out_dev, size_t n, void* stream);
/* swiglu_cuda cuda_kernels.h:379 (decode form, model.cpp:734-735) / swiglu_batch_cuda :482
 * (batch form, cpu_batch_processor.cpp:252-256) */
int tl_swiglu(const float* gate_dev, const float* up_dev, float* out_dev, size_t n, int batch_form, void* stream);
/* lookup_embedding_cuda cuda_kernels.h:408, from the quantized table */
int tl_lookup_embedding(const tl_qweight* table, int token_id, float* out_dev, void* stream);
/* argmax + repetition penalty on the device (api.cpp:73-97,134-136): first maximal index */
int tl_argmax(const float* logits_dev, int n, int* result_dev, void* stream);

/* ---- whole-model engine: what TinyLlamaModel::forward_device (model.cpp:810-1090) + KVCache
 *      (model.h:132-230, kv_cache.cpp:10-169) + the persistent workspaces (gpu_memory_optimization.cpp:8-52)
 *      become ------------------------------------------------------------------------------------- */
typedef struct tl_model_config {
  int hidden_size, intermediate_size, num_attention_heads, num_key_value_heads, num_hidden_layers, vocab_size;
  int max_position_embeddings; /* RoPE table rows and KV cache length */
  float rms_norm_eps, rope_theta;
  int tp_rank, tp_size; /* tensor parallel shard this handle holds (1 = unsharded) */
} tl_model_config;

/* tensor kinds for tl_model_set_tensor; names are the GGUF tensor names the reference consumes
 * (weight_management.cpp:732-960) */
enum tl_tensor_kind {
  TL_T_ATTN_Q = 0, TL_T_ATTN_K, TL_T_ATTN_V, TL_T_ATTN_OUTPUT, TL_T_FFN_GATE, TL_T_FFN_UP, TL_T_FFN_DOWN,
  TL_T_ATTN_NORM, TL_T_FFN_NORM, TL_T_TOKEN_EMBD, TL_T_OUTPUT_NORM, TL_T_OUTPUT
};

int tl_model_create(const tl_model_config* cfg, tl_model** out);
int tl_model_free(tl_model* m);
/* host_data: raw GGUF tensor bytes (blocks row-major by output row). With tp_size > 1 the caller passes THIS RANK'S SHARD
 * with its local shape: attn_q/k/v, ffn_gate/up, output by rows (rank-th block of heads / intermediate columns / vocab
 * rows), attn_output/ffn_down by columns (rank-th block of input columns, a multiple of 256); token_embd and the norms
 * whole (tinyllama.cpp_amd/tp.py has the slicing). layer = -1 for globals. */
int tl_model_set_tensor(tl_model* m, int kind, int layer, int type, const void* host_data, int rows, int cols);
/* allocate KV cache + workspaces, upload the RoPE table, check every tensor is present */
int tl_model_finalize(tl_model* m);
int tl_model_reset_kv(tl_model* m); /* KVCache::clear_data (model.h:182-197) */
/* One decode step = lookup_embedding + forward_device for one token at `pos`; logits_host may be NULL.
 * next_token_host (may be NULL) receives argmax(logits) computed on the device (greedy, no penalty). */
int tl_model_decode(tl_model* m, int token_id, int pos, float* logits_host, int* next_token_host);
/* Greedy generation entirely on the device: feeds argmax back as the next input for n_steps steps starting
 * at (first_token, start_pos); writes the n_steps sampled ids. Repetition penalty as api.cpp:86-97 over the
 * last 64 ids of `history` + what is generated (penalty <= 1: off). */
int tl_model_generate_greedy(tl_model* m, int first_token, int start_pos, int n_steps, float penalty,
                             const int* history_host, int n_history, int* out_ids_host);
/* The same loop with sample_top_k_top_p_temperature (api.cpp:126-216) on the device instead of arg-max: penalty, FP64
 * softmax of (logit - max) / temperature rounded to FP32, top_k by (probability, higher id first), FP32 cumulative
 * top_p cut, renormalisation, inverse-CDF draw as std::discrete_distribution does it. temperature < 0.05 = greedy.
 * Limit: effective top_k <= 1024 ("no top-k", i.e. top_k <= 0 or >= vocab, is rejected: sample on the host). The
 * uniform variates come from a counter-based generator keyed by `seed`. */
int tl_model_generate_sampled(tl_model* m, int first_token, int start_pos, int n_steps, float penalty,
                              const int* history_host, int n_history, float temperature, int top_k, float top_p,
                              unsigned long long seed, int* out_ids_host);
/* batched prefill of T prompt tokens starting at start_pos (forward_device_batch_prefill, model.cpp:1526-1862);
 * last_logits_host[vocab] (may be NULL) = logits of the last token; all_logits_host[T][vocab] may be NULL */
int tl_model_prefill(tl_model* m, const int* tokens_host, int num_tokens, int start_pos, float* last_logits_host,
                     float* all_logits_host);
/* Batched-prefill numerics / routing of this model (no counterpart in the reference, whose GPU prefill is FP32
 * cublasSgemm or BF16 cublasGemmEx by a T >= 4 rule, model.cpp:2152-2153): terms = bf16 MFMAs per FP32 product in the
 * prefill GEMMs: 6 (default, FP32-grade, logits within the decode tolerance), 3 (~2^-17 per product), 1 (plain bf16),
 * 0 = the process-wide value of tl_gemm_set_terms; prompts shorter than min_batch (default 8) and tensor-parallel
 * shards replay the decode graph token by token. */
int tl_model_set_prefill(tl_model* m, int terms, int min_batch);
/* device pointers into the model's state, for parity tests and for hosts that keep using the operator-level
 * calls: which = 0 K cache, 1 V cache of `layer` ([max_pos][n_kv_heads_local][head_dim] f32), 2 = x (hidden) */
int tl_model_device_ptr(tl_model* m, int which, int layer, void** dev_out);
/* algorithmic bytes one decode step at position pos must move on this rank (SURVEY §8d B_tok) */
int tl_model_bytes_per_token(const tl_model* m, int pos, double* weight_bytes, double* kv_bytes);
/* device bytes of the expanded prefill planes that tl_model_finalize keeps for this model's matrices (tl_qweight_expand;
 * 0: none -- switched off, or not affordable on this device). They serve the batched paths only: a decode step never
 * reads them. */
int tl_model_prefill_planes_bytes(const tl_model* m, size_t* bytes_out);
/* Multi-sequence decode (generate_batch api.cpp:836-1142; batch_prefill_parallel :1144, batch_generation_parallel :1423;
 * forward_cpu_batch_generation model.cpp:1158-1523 / forward_device_batch_generation :1863-2114): up to max_batch
 * sequences share the weights of every step -- the projections of the B tokens are skinny GEMMs on the matrix cores, the
 * KV cache is sequence-major ([seq][max_pos][kv heads][head_dim], model.cpp:1310-1312), attention runs per token over its
 * own sequence. tl_model_batch_init (after tl_model_finalize, tp_size 1) allocates that cache; tl_model_batch_prefill
 * fills one sequence's prompt (positions start_pos..) and returns its last logits; tl_model_batch_decode is ONE step with
 * batch_generation_parallel's arguments (token, position and sequence index per active sequence; indices must be
 * distinct), returning logits [n][vocab] and/or the first-maximum ids; tl_model_batch_generate_greedy runs n_steps steps
 * with ids and positions advancing on the device: out_ids_host [n_steps][n]. No repetition penalty on this path
 * (api.cpp:993,1084). Each sequence decodes exactly as it would alone (the reference's sequential fallback,
 * api.cpp:1116-1130); its parallel path leaves the current token out of its own attention (model.cpp:1327). */
int tl_model_batch_init(tl_model* m, int max_batch);
int tl_model_batch_prefill(tl_model* m, int seq_idx, const int* tokens_host, int num_tokens, int start_pos,
                           float* last_logits_host);
/* batch_prefill_parallel (api.cpp:1144): the tokens of ALL prompts as one packed batch (token i of sequence seq_idx[i] at
 * position positions[i]; the tokens of a sequence in position order), sequence-aware attention
 * (cpu_attention.cpp:254-325), chunks of 512 tokens. last_rows[j] = index of a token whose logits [n_last][vocab] and
 * first-maximum id are returned (the last token of each prompt); n_last <= max_batch; either output may be NULL. */
int tl_model_batch_prefill_packed(tl_model* m, int n_tokens, const int* tokens_host, const int* positions_host,
                                  const int* seq_idx_host, int n_last, const int* last_rows_host, float* last_logits_host,
                                  int* last_ids_host);
int tl_model_batch_decode(tl_model* m, int n, const int* tokens_host, const int* positions_host, const int* seq_idx_host,
                          float* logits_host, int* next_tokens_host);
int tl_model_batch_generate_greedy(tl_model* m, int n, const int* first_tokens_host, const int* start_pos_host,
                                   const int* seq_idx_host, int n_steps, int* out_ids_host);
/* tensor-parallel wiring (one process per GPU): RCCL communicator over xGMI; unique_id is the ncclUniqueId (128 bytes)
 * produced by tl_tp_unique_id on rank 0 and broadcast by the host (torch.distributed). Call tl_model_tp_init on every
 * rank before tl_model_finalize: the decode step (incl. its 2 all-reduces per layer and the logits all-gather) is
 * captured into the graph there. Replaces distributed_coordinator.cpp's MPI request routing. */
int tl_tp_unique_id(void* id_out_128);
int tl_model_tp_init(tl_model* m, const void* unique_id_128);
/* Peer-to-peer exchange for the decode step (no counterpart in the reference): the two f32[hidden] all-reduces per layer
 * and the logits all-gather of a tp_size 2..8 group as ONE small kernel each, which pushes 8-byte {value, epoch} words
 * straight into receive areas that every rank maps from every other rank (hipIpc; xGMI links between the GPUs of a
 * node) and sums what arrives in rank order. Wiring, on every rank, before tl_model_finalize:
 *   tl_model_tp_peer_export  -> this rank's 64-byte IPC handle; the host gathers the handles of all ranks (rank order)
 *   tl_model_tp_peer_import  -> maps them and runs a COLLECTIVE self-check (known sums); *ok_out = 1 if it passed here
 *   the host agrees on min(ok) over the ranks and calls tl_model_tp_peer_enable(m, agreed) on every rank.
 * Enabled: the decode step uses the peer kernels (the batched prefill's T x hidden all-reduces stay with RCCL and need
 * tl_model_tp_init; without a communicator a sharded prefill goes token by token). Every poll is bounded (5 s, or
 * TL_TP_PEER_TIMEOUT_MS): a missing rank turns into an error return of the decode / generate call, not a hang.
 * By default the exchanges are not even launches: the attn_output / ffn_down / lm_head GEMVs push their rows as they
 * finish them and collect the other ranks' rows before they end, so the sharded step has exactly the launches of the
 * unsharded one (tl_model_tp_peer_fused reports 1; TL_TP_PEER_FUSED=0 keeps one exchange kernel per all-reduce /
 * all-gather). Folded exchanges wait inside full-chip launches: ranks that SHARE a device (rehearsals) must cap the
 * launches so that all ranks fit together (TL_GEMV_CUS = CUs / ranks); with a device per rank nothing is needed. */
int tl_model_tp_peer_export(tl_model* m, void* handle_out_64);
int tl_model_tp_peer_import(tl_model* m, const void* handles_tp_x_64, int* ok_out);
int tl_model_tp_peer_enable(tl_model* m, int on);
int tl_model_tp_peer_enabled(const tl_model* m, int* on_out);
int tl_model_tp_peer_fused(const tl_model* m, int* on_out);
/* Diagnostic (no counterpart in the reference): *captured_out = 1 when the decode step replays as one hipGraph, 0 when
 * it is enqueued launch by launch (TL_NO_GRAPH=1, a shard without a communicator, or a collective that refused
 * capture). With TL_TP_SELFTEST=1 in the environment tl_model_tp_init gives a tp_size-1 model a one-rank communicator,
 * so the RCCL calls of the step and their capture can be exercised on one GPU. */
int tl_model_graph_captured(const tl_model* m, int* captured_out);
/* Diagnostic (no counterpart in the reference): the number of nodes (launches) of the captured decode step, 0 when the
 * step is not captured. A tensor-parallel shard whose exchanges ride inside its GEMVs has as many as the whole model. */
int tl_model_graph_nodes(const tl_model* m, int* nodes_out);
/* Test hook: ALL n shards of a tp_size = n model live in this process on this device (created with tp_rank = 0..n-1,
 * finalized without tl_model_tp_init); one decode step in lockstep, collectives as plain kernels. */
int tl_tp_local_group_decode(tl_model* const* shards, int n, int token_id, int pos, float* logits_host,
                             int* next_token_host);

/* the same for the batched prefill of a prompt (collectives as plain kernels); last_logits_host[vocab] may be NULL */
int tl_tp_local_group_prefill(tl_model* const* shards, int n, const int* tokens_host, int num_tokens, int start_pos,
                              float* last_logits_host);

#ifdef __cplusplus
}
#endif
#endif
