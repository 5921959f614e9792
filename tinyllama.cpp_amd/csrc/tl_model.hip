// tl_model.hip — the whole-model engine behind the C-ABI: what TinyLlamaModel::forward_device
// (model.cpp:810-1090), KVCache (model.h:132-230, kv_cache.cpp:10-169), the persistent workspaces
// (gpu_memory_optimization.cpp:8-52) and the greedy part of the generate loop (api.cpp:696-813) become on
// MI355X.
//
// Design (MI355X-first, not the reference's call pattern):
//   - weights stay quantized in HBM in the plane layout (tl_internal.h), uploaded once;
//   - one decode step is 2 + 6*L kernels: embed | per layer {norm+QKV GEMV, RoPE+KV write, attention,
//     O GEMV+residual, norm+gate/up GEMV+SwiGLU, down GEMV+residual} | norm+lm_head GEMV | penalty+argmax+advance;
//   - the step is captured ONCE into a hipGraph. Token id, position and the generated-id history live in a small
//     device-side state block that the last kernel of a step updates, so the graph is replayed back-to-back with
//     no host round trip, no per-token D2H of logits and no graph update (the reference syncs the stream and copies
//     vocab*4 bytes to the host every token, model.cpp:1085-1088).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <rccl/rccl.h>

#include "tl_internal.h"

namespace {

struct DevState {  // mirrors the device-side step state
  int token;
  int pos;
  int n_hist;
  float penalty;
};
constexpr int kStateHeaderInts = 4;

struct LayerW {
  tl_qweight* q[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* attn_norm = nullptr;
  float* ffn_norm = nullptr;
};

}  // namespace

struct tl_model {
  tl_model_config cfg;
  int head_dim = 0, kv_dim = 0;  // kv_dim: LOCAL kv heads x head_dim
  // tensor parallel shard (tp == 1: the whole model): local query heads / kv heads / intermediate columns / vocab rows
  int tp = 1, rank = 0;
  int nh_l = 0, nkv_l = 0, I_l = 0, V_l = 0, qd_l = 0;
  float* xp = nullptr;     // tp > 1: this rank's partial of attn_output / ffn_down, all-reduced into x
  float* zeros = nullptr;  // tp > 1: the 'residual' of ranks != 0
  bool allocated = false;  // buffers exist (set by tl_model_finalize, also when the graph is skipped)
  ncclComm_t comm = nullptr;
  // tp > 1: one-shot peer-to-peer exchange (tl_peer.hip) for the decode step's 16 KB all-reduces and the logits
  // all-gather, used instead of the RCCL calls once every rank's self-check passed (tl_model_tp_peer_enable)
  TlPeer* peer = nullptr;
  bool peer_on = false;
  // peer_on: the exchanges ride in the epilogues of the launches that produce their inputs (attn_output, ffn_down, lm_head;
  // tl_gemv.hip PEER) instead of launches of their own; TL_TP_PEER_FUSED=0 keeps the exchange launches (A/B, fallback)
  bool peer_fused = true;
  bool fused(const tl_qweight* w) const { return peer_on && peer_fused && w->type != TL_F32; }
  bool linked() const { return comm != nullptr || peer_on; }  // this shard can exchange with the rest of its group
  std::vector<LayerW> layers;
  tl_qweight* tok = nullptr;
  tl_qweight* out = nullptr;
  float* out_norm = nullptr;
  // KV cache: per layer [max_pos][n_kv_heads][head_dim] f32 (the reference's layout, model.cpp:622-625)
  float* kv_base = nullptr;
  size_t kv_layer_elems = 0;
  // workspaces
  float *x = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *att = nullptr, *h = nullptr, *logits = nullptr;
  float* h_up = nullptr;  // up projection of a gate/up pair stored in two different formats (two launches + SwiGLU)
  float* rope = nullptr;
  float* cand_v = nullptr;     // per-slice arg-max candidates (sampling)
  int* cand_i = nullptr;
  float* attn_ws = nullptr;    // split-attention partials
  float* xn = nullptr;         // normalised input of the unfused (F32-matrix) projections
  bool has_f32 = false;        // some projection matrix is F32 (set by finalize): no MFMA prefill, no multi-sequence step
  bool attn_split = false;
  int* state = nullptr;  // DevState header + history ring
  int hist_cap = 0;
  hipStream_t stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  // the same step with the top-k / top-p sampler as its tail (captured at the first sampled generation)
  hipGraph_t graph_sampled = nullptr;
  hipGraphExec_t graph_exec_sampled = nullptr;
  TlSampleParams* sparams = nullptr;
  float* sample_probs = nullptr;
  bool finalized = false;
  bool no_graph = false;  // TL_NO_GRAPH=1: replay steps eagerly (profilers that cannot follow graph launches)
  double weight_bytes = 0.0;  // algorithmic (GGUF) bytes of all per-token matrices incl. one embedding row
  // batched prefill (MFMA GEMMs over a chunk of tokens): workspaces, allocated at the first batched call
  struct Prefill {
    int cap = 0;               // tokens per chunk the buffers hold
    float *xb = nullptr, *qb = nullptr, *kb = nullptr, *vb = nullptr, *ab = nullptr, *gb = nullptr, *ub = nullptr;
    float* xpart = nullptr;    // tp > 1: this rank's partial of attn_output / ffn_down for the chunk, all-reduced into xb
    // K-split GEMMs leave their partial tiles for the next kernel to sum (no reduction launch, tl_gemm.hip GemmDefer):
    // part[0] q / attn_output / gate / ffn_down, part[1] k / up, part[2] v; `pend` = the projection not yet added to xb
    // (folded into the next RMSNorm split, which writes the new residual stream to xb2; the two then swap)
    float* xb2 = nullptr;
    float* part[3] = {nullptr, nullptr, nullptr};
    size_t part_cap = 0;
    GemmParts pend = {nullptr, 0, 0};
    GemmDefer dq = {}, dk = {}, dv = {};  // q / k / v of the layer in hand (qkv_deferred)
    void* planes = nullptr;    // bf16 [3][plane_stride]
    size_t plane_stride = 0;
    float* logits_all = nullptr;  // [cap][vocab], only when all logits were asked for
    int* toks = nullptr;
    int toks_cap = 0;
    // independent GEMMs of a layer (q | k, v and gate | up) run on two streams so that their workgroups share the chip
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
  } pf;
  int prefill_terms = 0;     // MFMAs per product in the prefill GEMMs (0: tl_gemm_set_terms' process-wide value)
  int prefill_min_batch = 8; // prompts shorter than this replay the decode graph token by token
  // multi-sequence decode: a second, SEQUENCE-MAJOR cache [layer][K|V][seq][max_pos][kv_dim] and the
  // device-resident per-sequence state of the batched generation loop (tl_model_batch_init)
  struct Batch {
    int max_batch = 0;
    float* kv = nullptr;
    int *toks = nullptr, *pos = nullptr, *seq = nullptr;  // [max_batch]
    float* last_h = nullptr;                              // [max_batch][hidden]: hidden rows whose logits a packed prefill returns
    int* ids = nullptr;                                   // generated ids [step][n]
    size_t ids_cap = 0;
    int gemv_max = 0;  // steps with at most this many sequences take the multi-vector GEMV route (0: never)
  } bt;
  // Chain launches (tl_gemv.hip): the step's GEMVs between two attention launches -- attn_output, gate/up, ffn_down and the
  // next layer's QKV (the lm_head after the last layer) -- run as tasks of ONE launch, handed over through arrival counters
  // instead of kernel boundaries. Planned by tl_model_finalize, when TL_CHAIN=1 is set, for unsharded models whose launches
  // all have a chain body (a measured negative result: off by default).
  struct Chain {
    bool on = false;
    void* tasks = nullptr;     // device: the tasks of all chain launches, TL_CHAIN_TASK_BYTES each
    unsigned* sync = nullptr;  // device: [generation word | error word | pad to 64 words | arrival counters of every task]
    struct Launch {
      int first_task, ntasks, max_nwg;
      size_t lds;
    };
    std::vector<Launch> launches;  // launches[l]: behind the attention of layer l
    unsigned* gen() const { return sync; }
    unsigned* err() const { return sync + 1; }
    unsigned* counters(int first_task) const { return sync + 64 + tl_chain_sync_words(first_task); }
  } chain;
  // The attention of a layer folded into the layer's attn_output launch (tl_gemv.hip, gemv_attn_kernel): four dependent
  // launches per layer instead of five. sync: tl_fused_attn_sync_words() words + a line for the error word.
  size_t expanded_bytes = 0;  // device bytes of the expanded prefill planes this model's finalize added (tl_gemm_expand)
  struct FusedAttention {
    bool on = false;
    unsigned* sync = nullptr;
    unsigned* gen() const { return sync + TL_SYNC_LINE_WORDS; }
    unsigned* err() const { return sync + tl_fused_attn_sync_words(); }
  } fattn;
  int alt_seq = -1;  // >= 0: kc() / vc() address this sequence's region of the batch cache (prompt prefill of a sequence)
  float* kc(int l) const {
    return alt_seq >= 0 ? bt.kv + (((size_t)l * 2) * bt.max_batch + alt_seq) * kv_layer_elems
                        : kv_base + (size_t)l * 2 * kv_layer_elems;
  }
  float* vc(int l) const {
    return alt_seq >= 0 ? bt.kv + (((size_t)l * 2 + 1) * bt.max_batch + alt_seq) * kv_layer_elems
                        : kv_base + (size_t)l * 2 * kv_layer_elems + kv_layer_elems;
  }
  float* bkc(int l) const { return bt.kv + ((size_t)l * 2) * bt.max_batch * kv_layer_elems; }      // sequence 0 of layer l
  float* bvc(int l) const { return bt.kv + ((size_t)l * 2 + 1) * bt.max_batch * kv_layer_elems; }
};

static int upload_f32(float** dst, const void* host, size_t n, hipStream_t s) {
  if (*dst) {
    TL_HIP(hipFree(*dst));
    *dst = nullptr;
  }
  TL_HIP(hipMalloc(dst, n * sizeof(float)));
  TL_HIP(hipMemcpyAsync(*dst, host, n * sizeof(float), hipMemcpyHostToDevice, s));
  TL_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int tl_model_create(const tl_model_config* cfg, tl_model** out) {
  TL_REQUIRE(cfg && out, "tl_model_create: null argument");
  TL_REQUIRE(cfg->hidden_size > 0 && cfg->intermediate_size > 0 && cfg->num_attention_heads > 0 &&
                 cfg->num_key_value_heads > 0 && cfg->num_hidden_layers > 0 && cfg->vocab_size > 0 &&
                 cfg->max_position_embeddings > 0,
             "tl_model_create: non-positive dimension in config");
  TL_REQUIRE(cfg->hidden_size % cfg->num_attention_heads == 0, "tl_model_create: hidden_size %d not divisible by %d heads",
             cfg->hidden_size, cfg->num_attention_heads);
  TL_REQUIRE(cfg->num_attention_heads % cfg->num_key_value_heads == 0, "tl_model_create: %d heads not divisible by %d kv heads",
             cfg->num_attention_heads, cfg->num_key_value_heads);
  TL_REQUIRE(cfg->hidden_size % 32 == 0 && cfg->intermediate_size % 32 == 0, "tl_model_create: sizes must be multiples of 32");
  const int tp = cfg->tp_size <= 0 ? 1 : cfg->tp_size;
  const int hd = cfg->hidden_size / cfg->num_attention_heads;
  TL_REQUIRE(cfg->tp_rank >= 0 && cfg->tp_rank < tp, "tl_model_create: tp_rank %d outside [0,%d)", cfg->tp_rank, tp);
  // Megatron-style shards: q/k/v/gate/up/output by output rows (heads, intermediate columns, vocab rows),
  // attn_output/ffn_down by input columns, which must fall on 256-element super-block boundaries
  TL_REQUIRE(cfg->num_attention_heads % tp == 0 && cfg->num_key_value_heads % tp == 0 &&
                 cfg->intermediate_size % tp == 0 && cfg->vocab_size % tp == 0,
             "tl_model_create: tp_size %d does not divide heads %d / kv heads %d / intermediate %d / vocab %d", tp,
             cfg->num_attention_heads, cfg->num_key_value_heads, cfg->intermediate_size, cfg->vocab_size);
  TL_REQUIRE(tp == 1 || ((cfg->num_attention_heads / tp * hd) % 256 == 0 && (cfg->intermediate_size / tp) % 256 == 0),
             "tl_model_create: tp_size %d: column shards of attn_output (%d) / ffn_down (%d) are not multiples of 256", tp,
             cfg->num_attention_heads / tp * hd, cfg->intermediate_size / tp);
  tl_model* m = new tl_model();
  m->cfg = *cfg;
  m->cfg.tp_size = tp;
  m->tp = tp;
  m->rank = cfg->tp_rank;
  m->head_dim = hd;
  m->nh_l = cfg->num_attention_heads / tp;
  m->nkv_l = cfg->num_key_value_heads / tp;
  m->I_l = cfg->intermediate_size / tp;
  m->V_l = cfg->vocab_size / tp;
  m->qd_l = m->nh_l * hd;
  m->kv_dim = hd * m->nkv_l;
  m->layers.resize(cfg->num_hidden_layers);
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete m;
    tl_set_error("tl_model_create: hipStreamCreate failed: %s", hipGetErrorString(e));
    return 1;
  }
  *out = m;
  return 0;
}

extern "C" int tl_model_free(tl_model* m) {
  if (!m) return 0;
  if (m->stream) hipStreamSynchronize(m->stream);
  if (m->graph_exec) hipGraphExecDestroy(m->graph_exec);
  if (m->graph) hipGraphDestroy(m->graph);
  if (m->graph_exec_sampled) hipGraphExecDestroy(m->graph_exec_sampled);
  if (m->graph_sampled) hipGraphDestroy(m->graph_sampled);
  hipFree(m->sparams);
  hipFree(m->sample_probs);
  hipFree(m->fattn.sync);
  for (auto& l : m->layers) {
    for (auto*& w : l.q) tl_qweight_free(w);
    hipFree(l.attn_norm);
    hipFree(l.ffn_norm);
  }
  tl_qweight_free(m->tok);
  tl_qweight_free(m->out);
  if (m->comm) ncclCommDestroy(m->comm);
  tl_peer_destroy(m->peer);
  float* bufs[] = {m->out_norm, m->kv_base, m->x, m->xp, m->zeros, m->q, m->k, m->v, m->att, m->h, m->h_up, m->logits, m->rope};
  for (float* b : bufs) hipFree(b);
  hipFree(m->state);
  hipFree(m->cand_v);
  hipFree(m->cand_i);
  hipFree(m->attn_ws);
  hipFree(m->xn);
  hipFree(m->chain.tasks);
  hipFree(m->chain.sync);
  float* pbufs[] = {m->pf.xb, m->pf.qb, m->pf.kb, m->pf.vb, m->pf.ab, m->pf.gb, m->pf.ub, m->pf.logits_all, m->pf.xpart,
                    m->pf.xb2, m->pf.part[0], m->pf.part[1], m->pf.part[2]};
  for (float* b : pbufs) hipFree(b);
  hipFree(m->pf.planes);
  hipFree(m->pf.toks);
  hipFree(m->bt.kv);
  hipFree(m->bt.toks);
  hipFree(m->bt.pos);
  hipFree(m->bt.seq);
  hipFree(m->bt.ids);
  hipFree(m->bt.last_h);
  if (m->pf.fork) hipEventDestroy(m->pf.fork);
  if (m->pf.join) hipEventDestroy(m->pf.join);
  if (m->pf.side) hipStreamDestroy(m->pf.side);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  return 0;
}

// BF16 -> F32 (exact: the bits become the upper half of the word; bfloat16_to_float32, utils.cpp:144-166)
static std::vector<float> widen_bf16(const void* host_data, size_t n) {
  const uint16_t* b = static_cast<const uint16_t*>(host_data);
  std::vector<float> out(n);
  for (size_t i = 0; i < n; ++i) {
    const uint32_t bits = (uint32_t)b[i] << 16;
    memcpy(&out[i], &bits, 4);
  }
  return out;
}

extern "C" int tl_model_set_tensor(tl_model* m, int kind, int layer, int type, const void* host_data, int rows,
                                   int cols) {
  TL_REQUIRE(m && host_data, "tl_model_set_tensor: null argument");
  TL_REQUIRE(!m->finalized, "tl_model_set_tensor: model already finalized");
  TL_REQUIRE(rows > 0 && cols > 0, "tl_model_set_tensor: shape %d x %d", rows, cols);
  if (type == TL_BF16) {
    // BF16 tensors (SafeTensors-derived files; weight_management.cpp:888-954 keeps them as uint16 and the CPU forward
    // widens every weight on the fly, utils.cpp:1046-1079) are widened once, here, and held as F32 matrices
    const std::vector<float> wide = widen_bf16(host_data, (size_t)rows * cols);
    return tl_model_set_tensor(m, kind, layer, TL_F32, wide.data(), rows, cols);
  }
  const tl_model_config& c = m->cfg;
  const int H = c.hidden_size, I = m->I_l, V = c.vocab_size;  // I: this rank's share
  const int qd = m->qd_l, kvd = m->kv_dim;
  if (kind == TL_T_ATTN_NORM || kind == TL_T_FFN_NORM || kind == TL_T_OUTPUT_NORM) {
    // norms must be F32 tensors (weight_management.cpp:838,926,957) or BF16 (widened above, model.cpp:84-107)
    TL_REQUIRE(type == TL_F32, "tl_model_set_tensor: norm weights must be F32 or BF16 (got type %d)", type);
    TL_REQUIRE((size_t)rows * cols == (size_t)H, "tl_model_set_tensor: norm weight has %d x %d elements, expected %d", rows, cols, H);
    if (kind == TL_T_OUTPUT_NORM) return upload_f32(&m->out_norm, host_data, H, m->stream);
    TL_REQUIRE(layer >= 0 && layer < c.num_hidden_layers, "tl_model_set_tensor: layer %d out of range", layer);
    return upload_f32(kind == TL_T_ATTN_NORM ? &m->layers[layer].attn_norm : &m->layers[layer].ffn_norm, host_data, H,
                      m->stream);
  }
  int er = 0, ec = 0;
  tl_qweight** slot = nullptr;
  switch (kind) {
    case TL_T_ATTN_Q: er = qd; ec = H; break;
    case TL_T_ATTN_K: er = kvd; ec = H; break;
    case TL_T_ATTN_V: er = kvd; ec = H; break;
    case TL_T_ATTN_OUTPUT: er = H; ec = qd; break;  // column shard
    case TL_T_FFN_GATE: er = I; ec = H; break;
    case TL_T_FFN_UP: er = I; ec = H; break;
    case TL_T_FFN_DOWN: er = H; ec = I; break;
    case TL_T_TOKEN_EMBD: er = V; ec = H; slot = &m->tok; break;
    case TL_T_OUTPUT: er = m->V_l; ec = H; slot = &m->out; break;  // vocab-row shard
    default: TL_REQUIRE(false, "tl_model_set_tensor: unknown tensor kind %d", kind);
  }
  if (!slot) {
    TL_REQUIRE(layer >= 0 && layer < c.num_hidden_layers, "tl_model_set_tensor: layer %d out of range", layer);
    slot = &m->layers[layer].q[kind];
  }
  // Block formats run the fused GEMV / MFMA launches. An F32 matrix (the reference accepts them too: "transformer_cpp
  // fp32 exports", weight_management.cpp:889-955; BF16 arrives here as F32) runs the unfused F32 route of the decode step
  // (launch_proj below); such a model prefills token by token and has no multi-sequence step.
  TL_REQUIRE(type == TL_F32 || type == TL_Q4_K || type == TL_Q6_K || type == TL_Q8_0 || type == TL_Q8_K,
             "tl_model_set_tensor: kind %d has ggml type %d; the engine holds F32, BF16 (widened), Q4_K, Q6_K, Q8_0 and "
             "Q8_K matrices", kind, type);
  TL_REQUIRE(rows == er && cols == ec,
             "tl_model_set_tensor: kind %d has shape %d x %d, config expects %d x %d (tp rank %d of %d: pass this rank's shard)",
             kind, rows, cols, er, ec, m->rank, m->tp);
  if (*slot) {
    tl_qweight_free(*slot);
    *slot = nullptr;
  }
  return tl_qweight_upload(type, host_data, rows, cols, slot, m->stream);
}

static double gguf_bytes(const tl_qweight* w, int rows) {
  return (double)rows * w->cols / tl_type_block_elems(w->type) * (double)tl_type_block_bytes(w->type);
}

// ---- the per-step state + final kernel ------------------------------------------------------------------
// Sampling is two launches: (1) argmax_slices_kernel: ARGMAX_SLICES workgroups each find the first maximum of one slice
// of the raw logits; (2) sample_advance_kernel (one workgroup): applies the repetition penalty (api.cpp:86-97, in
// order, once per occurrence) to the <= 64 penalised logits in memory, then takes the first maximum (api.cpp:79,135)
// over {slice winners} U {penalised ids}. The penalty only ever lowers a logit, so the only way a slice winner can be
// wrong is when it is itself a penalised id: such a slice is rescanned in full (rare, <= 64 slices).
constexpr int ARGMAX_SLICES = 256;
constexpr int ARGMAX_THREADS = 256;

__device__ __forceinline__ void better(float v, int i, float& best, int& idx) {
  if (v > best || (v == best && i < idx)) {
    best = v;
    idx = i;
  }
}
// workgroup-wide first-maximum; result valid in thread 0. bv/bi: >= 16 entries of LDS each
__device__ __forceinline__ void block_argmax(float& best, int& idx, float* bv, int* bi) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(idx, o);
    better(ov, oi, best, idx);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    bv[threadIdx.x >> 6] = best;
    bi[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) better(bv[w], bi[w], best, idx);
}

__global__ __launch_bounds__(ARGMAX_THREADS) void argmax_slices_kernel(const float* __restrict__ logits, int n,
                                                                       float* __restrict__ cand_v,
                                                                       int* __restrict__ cand_i) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int per = (n + ARGMAX_SLICES - 1) / ARGMAX_SLICES;
  const int lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += ARGMAX_THREADS) {
    const float v = logits[i];
    if (v > best) {  // strictly greater keeps the first maximum within a thread's stride
      best = v;
      idx = i;
    }
  }
  block_argmax(best, idx, bv, bi);
  if (threadIdx.x == 0) {
    cand_v[blockIdx.x] = best;
    cand_i[blockIdx.x] = idx;
  }
}

__global__ __launch_bounds__(1024) void sample_advance_kernel(float* __restrict__ logits, int n,
                                                              const float* __restrict__ cand_v,
                                                              const int* __restrict__ cand_i, int* __restrict__ state,
                                                              int hist_cap) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int win[64];
  __shared__ int nwin;
  int* hist = state + kStateHeaderInts;
  const int n_hist = state[2];
  const float penalty = __int_as_float(state[3]);
  const int per = (n + ARGMAX_SLICES - 1) / ARGMAX_SLICES;
  if (threadIdx.x == 0) {
    int k = 0;
    if (penalty > 1.0f) {  // sequential on purpose: a repeated id is penalised once per occurrence, in order
      const int start = n_hist > 64 ? n_hist - 64 : 0;
      for (int i = start; i < n_hist; ++i) {
        const int id = hist[i];
        if (id < 0 || id >= n) continue;
        const float l = logits[id];
        logits[id] = (l > 0.0f) ? l / penalty : l * penalty;
        win[k++] = id;
      }
    }
    nwin = k;
  }
  __syncthreads();
  const int nw = nwin;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int s = threadIdx.x; s < ARGMAX_SLICES; s += blockDim.x) {
    const int ci = cand_i[s];
    bool penalised = false;
    for (int k = 0; k < nw; ++k) penalised |= win[k] == ci;
    if (!penalised) {
      if (ci != 0x7fffffff) better(cand_v[s], ci, best, idx);
    } else {  // the slice winner was lowered: rescan the slice (now holding the penalised values)
      const int lo = s * per, hi = lo + per < n ? lo + per : n;
      for (int i = lo; i < hi; ++i) {
        const float v = logits[i];
        if (v > best || (v == best && i < idx)) {
          best = v;
          idx = i;
        }
      }
    }
  }
  block_argmax(best, idx, bv, bi);
  if (threadIdx.x == 0) {
    if (idx == 0x7fffffff) idx = 0;             // all -inf/NaN: std::max_element returns the first element
    state[0] = idx;                             // next input token
    state[1] = state[1] + 1;                    // next position
    if (n_hist < hist_cap) hist[n_hist] = idx;  // generated ids, in order
    state[2] = n_hist + 1;
  }
}

// Optional per-kernel timing: when `prof` is set, an event is recorded before the first launch and after every
// launch of the step (kinds in tl_model_profile_decode's order).
struct StepProf {
  std::vector<hipEvent_t> ev;
  std::vector<int> kind;
  int mark(hipStream_t s, int k) {
    hipEvent_t e;
    TL_HIP(hipEventCreate(&e));
    TL_HIP(hipEventRecord(e, s));
    ev.push_back(e);
    kind.push_back(k);
    return 0;
  }
};
#define PROF_MARK(k) do { if (prof) TL_TRY(prof->mark(s, (k))); } while (0)

// ---- the GEMV launches of a step (also used one kind at a time by tl_model_bench_kernel) ---------------------------
static GemvArgs args_qkv(tl_model* m, int l) {
  LayerW& L = m->layers[l];
  GemvArgs a = {};
  a.seg[0] = {qw_view(L.q[TL_T_ATTN_Q]), m->q};
  a.seg[1] = {qw_view(L.q[TL_T_ATTN_K]), m->k};
  a.seg[2] = {qw_view(L.q[TL_T_ATTN_V]), m->v};
  a.nseg = 3;
  a.cols = m->cfg.hidden_size;
  a.x = m->x;
  a.norm_w = L.attn_norm;
  a.eps = m->cfg.rms_norm_eps;
  a.role = GEMV_QKV;
  return a;
}
// tp > 1: the row-split projections (attn_output, ffn_down) write this rank's PARTIAL sum into xp; rank 0 folds the
// residual into its partial, so that the all-reduce (sum) of the partials is the new residual stream on every rank.
// With the exchange folded into the launch (tl_model::fused) the sum over ranks lands in x directly.
static GemvArgs args_attn_out(tl_model* m, int l) {
  GemvArgs o = {};
  const bool fz = m->fused(m->layers[l].q[TL_T_ATTN_OUTPUT]);
  o.seg[0] = {qw_view(m->layers[l].q[TL_T_ATTN_OUTPUT]), m->tp > 1 && !fz ? m->xp : m->x};
  o.peer = fz ? tl_peer_fused_reduce(m->peer) : nullptr;
  o.nseg = 1;
  o.cols = m->qd_l;
  o.x = m->att;
  if (m->attn_split) {  // the attention output is still in split partials: combined while staging x
    o.attn_ws = m->attn_ws;
    o.attn_pos = m->state + 1;
    o.attn_hd = m->head_dim;
    o.attn_merged = 1;
  }
  o.resid = m->rank == 0 ? m->x : m->zeros;
  o.role = GEMV_ATTN_OUT;
  return o;
}
static GemvArgs args_gate_up(tl_model* m, int l) {
  LayerW& L = m->layers[l];
  GemvArgs g = {};
  g.seg[0] = {qw_view(L.q[TL_T_FFN_GATE]), m->h};
  g.seg[1] = {qw_view(L.q[TL_T_FFN_UP]), m->h_up};  // (only written when gate and up differ in format)
  g.nseg = 2;
  g.pair = 1;
  g.cols = m->cfg.hidden_size;
  g.x = m->x;
  g.norm_w = L.ffn_norm;
  g.eps = m->cfg.rms_norm_eps;
  g.role = GEMV_GATE_UP;
  return g;
}
static GemvArgs args_down(tl_model* m, int l) {
  GemvArgs d = {};
  const bool fz = m->fused(m->layers[l].q[TL_T_FFN_DOWN]);
  d.seg[0] = {qw_view(m->layers[l].q[TL_T_FFN_DOWN]), m->tp > 1 && !fz ? m->xp : m->x};
  d.peer = fz ? tl_peer_fused_reduce(m->peer) : nullptr;
  d.nseg = 1;
  d.cols = m->I_l;
  d.x = m->h;
  d.resid = m->rank == 0 ? m->x : m->zeros;
  d.role = GEMV_DOWN;
  return d;
}
static GemvArgs args_lm_head(tl_model* m) {
  GemvArgs f = {};
  f.seg[0] = {qw_view(m->out), m->logits + (size_t)m->rank * m->V_l};  // this rank's vocab rows
  f.nseg = 1;
  f.cols = m->cfg.hidden_size;
  f.x = m->x;
  f.norm_w = m->out_norm;
  f.eps = m->cfg.rms_norm_eps;
  f.role = GEMV_LM_HEAD;
  f.peer = m->fused(m->out) ? tl_peer_fused_gather(m->peer) : nullptr;
  return f;
}

// One projection launch of the step. Block formats: the fused GEMV as described by `a`. If a matrix of the launch is F32
// (tl_model_set_tensor), the same computation unfused: combine of the attention partials / RMSNorm into a buffer, one
// plain GEMV per matrix, SwiGLU: matvec_f32_f32_vector_cpu's place in model.cpp:577-749.
static int launch_proj(tl_model* m, const GemvArgs& a, hipStream_t s) {
  bool f32 = false;
  for (int i = 0; i < a.nseg; ++i) f32 |= a.seg[i].w.type == TL_F32;
  if (!f32) return tl_launch_gemv(a, s);
  const float* x = a.x;
  if (a.attn_ws) {
    AttnArgs c = {};
    c.ws = const_cast<float*>(a.attn_ws);
    c.out = m->att;
    c.nh = m->nh_l;
    c.hd = a.attn_hd;
    c.pos_dev = a.attn_pos;
    c.done_cnt = a.attn_merged ? reinterpret_cast<unsigned*>(c.ws) : nullptr;  // (only "merged or not" is read)
    TL_TRY(tl_launch_attn_combine(c, s));
    x = m->att;
  }
  if (a.norm_w) {
    TL_TRY(tl_launch_rmsnorm(a.x, a.norm_w, m->xn, 1, a.cols, a.eps, 1, s));
    x = m->xn;
  }
  for (int i = 0; i < a.nseg; ++i) {
    GemvArgs one = {};
    one.seg[0] = a.seg[i];
    one.nseg = 1;
    one.cols = a.cols;
    one.x = x;
    one.resid = a.resid;
    one.role = GEMV_GENERIC;
    TL_TRY(tl_launch_gemv(one, s));
  }
  if (a.pair) TL_TRY(tl_swiglu(a.seg[0].out, a.seg[1].out, a.seg[0].out, (size_t)a.seg[0].w.rows, 0, s));
  return 0;
}

#define TL_NCCL(expr)                                                                            \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) {                                                                     \
      tl_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_)); \
      return 4;                                                                                  \
    }                                                                                            \
  } while (0)

// x = sum over ranks of xp (RCCL over xGMI; one f32[hidden_size] all-reduce per row-split projection)
// (w: the row-split matrix whose launch produced xp; when that launch carried the exchange itself there is nothing to do)
static int tp_all_reduce(tl_model* m, hipStream_t s, const tl_qweight* w = nullptr) {
  if (m->tp == 1 && !m->comm) return 0;
  if (w && m->fused(w)) return 0;
  if (m->peer_on) return tl_peer_all_reduce(m->peer, m->xp, m->x, s);
  TL_REQUIRE(m->comm, "tl_model: tp_size %d but tl_model_tp_init was not called", m->tp);
  // (tp == 1 with a communicator is the one-rank self-test of tl_model_tp_init: in place on x)
  TL_NCCL(ncclAllReduce(m->tp > 1 ? m->xp : m->x, m->x, (size_t)m->cfg.hidden_size, ncclFloat, ncclSum, m->comm, s));
  return 0;
}

// every rank gets the full logits (in place: its own vocabulary rows already sit at rank * V_l) and samples
static int tp_gather_logits(tl_model* m, hipStream_t s) {
  if (m->tp == 1 && !m->comm) return 0;
  if (m->fused(m->out)) return 0;  // the lm_head launch gathered
  if (m->peer_on) return tl_peer_all_gather(m->peer, m->logits, s);
  TL_REQUIRE(m->comm, "tl_model: tp_size %d but tl_model_tp_init was not called", m->tp);
  TL_NCCL(ncclAllGather(m->logits + (size_t)m->rank * m->V_l, m->logits, (size_t)m->V_l, ncclFloat, m->comm, s));
  return 0;
}
// after a synchronised run: did a peer exchange give up waiting (a rank of the group is gone or out of step)?
static int tp_check_peer(tl_model* m) {
  if (!m->peer_on) return 0;
  unsigned flag = 0;
  TL_TRY(tl_peer_error_async(m->peer, &flag, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  if (flag != 0) {
    // the flag is sticky on the device: clear it, so that the NEXT call reports its own exchanges, not this one's
    TL_TRY(tl_peer_clear_error(m->peer, m->stream));
    TL_HIP(hipStreamSynchronize(m->stream));
    tl_set_error("tl_model: a peer-to-peer exchange timed out waiting for another rank of the tensor-parallel group");
    return 5;
  }
  return 0;
}

// ---- the step in pieces, so that a group of shards living in ONE process can be driven in lockstep (tests) ------------
static int enqueue_embed(tl_model* m, hipStream_t s) {
  // gather form: a token id outside the table embeds as a zero vector, as lookup_embedding does (model_utils.cpp:15-19)
  // and as the prefill route (prefill_embed) does, instead of silently reading row 0
  return tl_launch_dequant_gather(m->tok, m->state, 1, m->x, s);
}
// RoPE + KV write + attention of layer l (split partials in attn_ws, or the output in att)
static AttnArgs args_attn_split(tl_model* m, int l) {
  AttnArgs t = {};
  t.q = m->q;
  t.k = m->k;
  t.v = m->v;
  t.kc = m->kc(l);
  t.vc = m->vc(l);
  t.table = m->rope;
  t.out = m->att;
  t.ws = m->attn_ws;
  t.nh = m->nh_l;
  t.nkv = m->nkv_l;
  t.hd = m->head_dim;
  t.pos_dev = m->state + 1;
  t.max_seq = m->cfg.max_position_embeddings;
  t.scale = 1.0f / sqrtf((float)m->head_dim);
  t.done_cnt = reinterpret_cast<unsigned*>(m->attn_ws + tl_attn_ws_floats(m->nh_l, m->head_dim));  // nh_l counters
  return t;
}
static int enqueue_attention(tl_model* m, int l, hipStream_t s, StepProf* prof) {
  const tl_model_config& c = m->cfg;
  const int* pos_dev = m->state + 1;
  const float scale = 1.0f / sqrtf((float)m->head_dim);
  if (m->attn_split) {  // RoPE + KV write + split attention in one launch
    PROF_MARK(2);
    TL_TRY(tl_launch_attn_split(args_attn_split(m, l), true, s));
  } else {
    TL_TRY(tl_launch_rope_kv(m->q, m->k, m->v, m->kc(l), m->vc(l), m->rope, m->nh_l, m->nkv_l, m->head_dim, pos_dev,
                             c.max_position_embeddings, s));
    PROF_MARK(2);
    TL_TRY(tl_launch_attn_decode(m->q, m->kc(l), m->vc(l), m->att, m->nh_l, m->nkv_l, m->head_dim, 0, pos_dev,
                                 c.max_position_embeddings, scale, s));
  }
  return 0;
}
// norm + QKV GEMV, RoPE + KV write + attention, attn_output GEMV (-> x, or the partial xp when tp > 1)
static int enqueue_attn_half(tl_model* m, int l, hipStream_t s, StepProf* prof) {
  TL_TRY(launch_proj(m, args_qkv(m, l), s));
  PROF_MARK(1);
  if (m->fattn.on && !prof) {  // attention + attn_output in one launch
    GemvArgs o = args_attn_out(m, l);
    o.attn_merged = 0;
    return tl_launch_gemv_attn(o, args_attn_split(m, l), m->fattn.sync, (unsigned)l + 1u,
                               (unsigned)m->cfg.num_hidden_layers + 1u, m->fattn.err(), s);
  }
  TL_TRY(enqueue_attention(m, l, s, prof));
  PROF_MARK(3);
  return launch_proj(m, args_attn_out(m, l), s);
}
// norm + gate/up GEMV + SwiGLU, ffn_down GEMV (-> x, or the partial xp when tp > 1)
static int enqueue_mlp_half(tl_model* m, int l, hipStream_t s, StepProf* prof) {
  TL_TRY(launch_proj(m, args_gate_up(m, l), s));
  PROF_MARK(5);
  return launch_proj(m, args_down(m, l), s);
}
static int enqueue_head(tl_model* m, hipStream_t s) { return launch_proj(m, args_lm_head(m), s); }
static int enqueue_sample(tl_model* m, hipStream_t s) {
  const int V = m->cfg.vocab_size;
  argmax_slices_kernel<<<ARGMAX_SLICES, ARGMAX_THREADS, 0, s>>>(m->logits, V, m->cand_v, m->cand_i);
  sample_advance_kernel<<<1, 256, 0, s>>>(m->logits, V, m->cand_v, m->cand_i, m->state, m->hist_cap);
  TL_HIP(hipGetLastError());
  return 0;
}

static int enqueue_sample_topk(tl_model* m, hipStream_t s) {
  return tl_launch_sample_topk(m->logits, m->cfg.vocab_size, m->sample_probs, m->state, kStateHeaderInts, m->hist_cap,
                               m->sparams, nullptr, nullptr, nullptr, nullptr, s);
}

// The chain form of the step (tl_model::Chain): embed | QKV of layer 0 | per layer: attention, then ONE launch with the tasks
// attn_output -> gate/up -> ffn_down -> next QKV (lm_head after the last layer) | sampling | generation bump.
static int plan_chain(tl_model* m) {
  m->chain.on = false;
  // opt-in: measured SLOWER than one launch per projection in both forms built (profiles/r03_a_chain_*)
  const char* e = getenv("TL_CHAIN");
  if (!e || atoi(e) == 0) return 0;
  if (m->tp != 1 || m->comm || m->has_f32 || !m->attn_split) return 0;
  const int L = m->cfg.num_hidden_layers;
  std::vector<unsigned char> host((size_t)L * 4 * TL_CHAIN_TASK_BYTES, 0);
  m->chain.launches.clear();
  for (int l = 0; l < L; ++l) {
    const GemvArgs args[4] = {args_attn_out(m, l), args_gate_up(m, l), args_down(m, l),
                              l + 1 < L ? args_qkv(m, l + 1) : args_lm_head(m)};
    tl_model::Chain::Launch cl = {l * 4, 4, 0, 0};
    int pred = 0;
    for (int j = 0; j < 4; ++j) {
      size_t lds = 0;
      int nwg = 0;
      bool ok = false;
      TL_TRY(tl_chain_plan(args[j], host.data() + (size_t)(l * 4 + j) * TL_CHAIN_TASK_BYTES, &lds, pred, &nwg, &ok));
      if (!ok) return 0;  // a launch without a chain body: the whole model keeps one launch per projection
      pred = nwg;
      if (nwg > cl.max_nwg) cl.max_nwg = nwg;
      if (lds > cl.lds) cl.lds = lds;
    }
    m->chain.launches.push_back(cl);
  }
  const size_t sync_bytes = sizeof(unsigned) * (64 + tl_chain_sync_words(L * 4));
  TL_HIP(hipMalloc(&m->chain.tasks, host.size()));
  TL_HIP(hipMalloc(&m->chain.sync, sync_bytes));
  TL_HIP(hipMemcpyAsync(m->chain.tasks, host.data(), host.size(), hipMemcpyHostToDevice, m->stream));
  TL_HIP(hipMemsetAsync(m->chain.sync, 0, sync_bytes, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));  // host is a local buffer
  m->chain.on = true;
  return 0;
}
// Expanded prefill planes for every quantized matrix of the model (tl_gemm.hip, gemm_px_kernel), when they are affordable:
// at most a quarter of the device's memory and what is free now less 4 GiB (TL_PREFILL_PLANES=0 / 1: never / always).
static int expand_prefill_planes(tl_model* m) {
  const char* e = getenv("TL_PREFILL_PLANES");
  const int mode = e ? (atoi(e) != 0 ? 1 : 0) : -1;
  if (mode == 0) return 0;
  std::vector<const tl_qweight*> ws;
  for (auto& l : m->layers)
    for (auto* w : l.q)
      if (tl_gemm_expandable(w)) ws.push_back(w);
  if (tl_gemm_expandable(m->out)) ws.push_back(m->out);
  size_t need = 0;
  for (auto* w : ws)
    if (!tl_gemm_expanded(w)) need += tl_gemm_expanded_bytes(w->rows, w->cols);
  if (mode < 0) {
    size_t free_b = 0, total_b = 0;
    TL_HIP(hipMemGetInfo(&free_b, &total_b));
    if (need > total_b / 4 || need + ((size_t)4 << 30) > free_b) return 0;
  }
  for (auto* w : ws) TL_TRY(tl_gemm_expand(w, m->stream));
  m->expanded_bytes = need;
  return 0;
}

// TL_FUSED_ATTN=1: fold every layer's attention into its attn_output launch, where that launch has the folded form
static int plan_fused_attention(tl_model* m) {
  m->fattn.on = false;
  const char* e = getenv("TL_FUSED_ATTN");
  if (!e || atoi(e) == 0) return 0;
  if (m->tp != 1 || m->comm || m->has_f32 || !m->attn_split || m->chain.on) return 0;
  for (int l = 0; l < m->cfg.num_hidden_layers; ++l) {
    GemvArgs o = args_attn_out(m, l);
    o.attn_merged = 0;
    if (!tl_gemv_attn_supported(o, args_attn_split(m, l))) return 0;
  }
  const size_t bytes = sizeof(unsigned) * (tl_fused_attn_sync_words() + TL_SYNC_LINE_WORDS);
  if (!m->fattn.sync) TL_HIP(hipMalloc(&m->fattn.sync, bytes));
  TL_HIP(hipMemsetAsync(m->fattn.sync, 0, bytes, m->stream));
  m->fattn.on = true;
  return 0;
}
static int enqueue_sample(tl_model* m, hipStream_t s);
static int enqueue_sample_topk(tl_model* m, hipStream_t s);
static int enqueue_step_chain(tl_model* m, bool sampled) {
  hipStream_t s = m->stream;
  const int L = m->cfg.num_hidden_layers;
  TL_TRY(enqueue_embed(m, s));
  TL_TRY(launch_proj(m, args_qkv(m, 0), s));
  for (int l = 0; l < L; ++l) {
    TL_TRY(enqueue_attention(m, l, s, nullptr));
    const tl_model::Chain::Launch& cl = m->chain.launches[l];
    TL_TRY(tl_launch_chain(static_cast<const unsigned char*>(m->chain.tasks) + (size_t)cl.first_task * TL_CHAIN_TASK_BYTES,
                           cl.ntasks, cl.max_nwg, cl.lds, m->chain.counters(cl.first_task), m->chain.gen(), m->chain.err(), s));
  }
  if (sampled) TL_TRY(enqueue_sample_topk(m, s));
  else TL_TRY(enqueue_sample(m, s));
  return tl_launch_chain_gen_bump(m->chain.gen(), s);
}
// after a synchronised run: did a task of a chain launch give up waiting for its predecessor?
static int chain_check(tl_model* m) {
  if (m->fattn.on) {
    unsigned flag = 0;
    TL_HIP(hipMemcpy(&flag, m->fattn.err(), sizeof(flag), hipMemcpyDeviceToHost));
    if (flag != 0) {
      TL_HIP(hipMemset(m->fattn.err(), 0, sizeof(flag)));
      tl_set_error("tl_model: an attn_output launch timed out waiting for the attention it carries");
      return 6;
    }
  }
  if (!m->chain.on) return 0;
  unsigned flag = 0;
  TL_HIP(hipMemcpy(&flag, m->chain.err(), sizeof(flag), hipMemcpyDeviceToHost));
  if (flag != 0) {
    TL_HIP(hipMemset(m->chain.err(), 0, sizeof(flag)));
    tl_set_error("tl_model: a task of a chain launch timed out waiting for its predecessor");
    return 6;
  }
  return 0;
}

static int enqueue_step(tl_model* m, StepProf* prof = nullptr, bool sampled = false) {
  hipStream_t s = m->stream;
  if (m->chain.on && !prof) return enqueue_step_chain(m, sampled);
  PROF_MARK(-1);
  TL_TRY(enqueue_embed(m, s));
  PROF_MARK(0);
  for (int l = 0; l < m->cfg.num_hidden_layers; ++l) {
    TL_TRY(enqueue_attn_half(m, l, s, prof));
    TL_TRY(tp_all_reduce(m, s, m->layers[l].q[TL_T_ATTN_OUTPUT]));
    PROF_MARK(4);
    TL_TRY(enqueue_mlp_half(m, l, s, prof));
    TL_TRY(tp_all_reduce(m, s, m->layers[l].q[TL_T_FFN_DOWN]));
    PROF_MARK(6);
  }
  TL_TRY(enqueue_head(m, s));
  TL_TRY(tp_gather_logits(m, s));
  PROF_MARK(7);
  if (sampled) TL_TRY(enqueue_sample_topk(m, s));
  else TL_TRY(enqueue_sample(m, s));
  PROF_MARK(8);
  if (m->fattn.on && !prof) TL_TRY(tl_launch_chain_gen_bump(m->fattn.gen(), s));
  return 0;
}

// ---- a whole tensor-parallel group inside ONE process on ONE device (parity tests of the sharded path on a single
//      GPU; the collectives are plain kernels here, RCCL is only needed across processes) ---------------------------
__global__ void group_sum_kernel(const float* const* __restrict__ parts, float* const* __restrict__ outs, int n, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  float acc = 0.f;
  for (int r = 0; r < n; ++r) acc += parts[r][i];  // rank order, like a ring reduce started at rank 0
  for (int r = 0; r < n; ++r) outs[r][i] = acc;
}

extern "C" int tl_tp_local_group_decode(tl_model* const* ms, int n, int token_id, int pos, float* logits_host,
                                        int* next_token_host) {
  TL_REQUIRE(ms && n >= 1 && n <= 16, "tl_tp_local_group_decode: bad group");
  tl_model* m0 = ms[0];
  for (int r = 0; r < n; ++r) {
    TL_REQUIRE(ms[r] && ms[r]->allocated && ms[r]->tp == n && ms[r]->rank == r,
               "tl_tp_local_group_decode: member %d is not rank %d of a finalized %d-way group", r, r, n);
    TL_REQUIRE(ms[r]->cfg.hidden_size == m0->cfg.hidden_size && ms[r]->cfg.vocab_size == m0->cfg.vocab_size,
               "tl_tp_local_group_decode: members disagree on the model shape");
  }
  TL_REQUIRE(pos >= 0 && pos < m0->cfg.max_position_embeddings, "tl_tp_local_group_decode: pos %d out of range", pos);
  hipStream_t s = m0->stream;
  const int H = m0->cfg.hidden_size, V = m0->cfg.vocab_size;
  std::vector<const float*> parts(n);
  std::vector<float*> outs(n);
  for (int r = 0; r < n; ++r) {
    parts[r] = ms[r]->xp;
    outs[r] = ms[r]->x;
  }
  const float** parts_dev = nullptr;
  float** outs_dev = nullptr;
  TL_HIP(hipMalloc(&parts_dev, sizeof(float*) * n));
  TL_HIP(hipMalloc(&outs_dev, sizeof(float*) * n));
  TL_HIP(hipMemcpy(parts_dev, parts.data(), sizeof(float*) * n, hipMemcpyHostToDevice));
  TL_HIP(hipMemcpy(outs_dev, outs.data(), sizeof(float*) * n, hipMemcpyHostToDevice));
  int rc = 0;
  auto all = [&](auto&& f) {
    for (int r = 0; r < n && rc == 0; ++r) rc = f(ms[r]);
  };
  auto reduce = [&]() {
    if (rc == 0 && n > 1) group_sum_kernel<<<(H + 255) / 256, 256, 0, s>>>(parts_dev, outs_dev, n, H);
  };
  for (int r = 0; r < n && rc == 0; ++r) {
    const int st[4] = {token_id, pos, 0, 0x3f800000};
    if (hipMemcpyAsync(ms[r]->state, st, sizeof(st), hipMemcpyHostToDevice, s) != hipSuccess) rc = 1;
  }
  if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = 1;  // `st` is a stack buffer
  all([&](tl_model* m) { return enqueue_embed(m, s); });
  for (int l = 0; l < m0->cfg.num_hidden_layers && rc == 0; ++l) {
    all([&](tl_model* m) { return enqueue_attn_half(m, l, s, nullptr); });
    reduce();
    all([&](tl_model* m) { return enqueue_mlp_half(m, l, s, nullptr); });
    reduce();
  }
  all([&](tl_model* m) { return enqueue_head(m, s); });
  if (rc == 0 && n > 1) {  // all-gather: rank r's vocab rows into everybody's logits
    for (int r = 0; r < n && rc == 0; ++r)
      for (int q = 0; q < n && rc == 0; ++q)
        if (q != r && hipMemcpyAsync(ms[q]->logits + (size_t)r * ms[r]->V_l, ms[r]->logits + (size_t)r * ms[r]->V_l,
                                     sizeof(float) * (size_t)ms[r]->V_l, hipMemcpyDeviceToDevice, s) != hipSuccess)
          rc = 1;
  }
  all([&](tl_model* m) { return enqueue_sample(m, s); });
  if (rc == 0 && logits_host &&
      hipMemcpyAsync(logits_host, m0->logits, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, s) != hipSuccess)
    rc = 1;
  if (rc == 0 && next_token_host &&
      hipMemcpyAsync(next_token_host, m0->state, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess)
    rc = 1;
  const hipError_t e = hipStreamSynchronize(s);
  hipFree(parts_dev);
  hipFree(outs_dev);
  if (rc > 1) return rc;
  TL_REQUIRE(rc == 0, "tl_tp_local_group_decode: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
  TL_HIP(e);
  return 0;
}

// Measurement hook: `iters` back-to-back launches of ONE kind of the step's GEMVs (1 QKV, 4 attn_output, 5 gate/up,
// 6 down, 7 lm_head), cycling through the layers so that every launch streams its weights from HBM, between two HIP
// events on the engine's stream. bytes_per_launch_out = the average GGUF bytes of the matrices one launch reads.
extern "C" int tl_model_bench_kernel(tl_model* m, int kind, int iters, int warmup, float* avg_us_out,
                                     double* bytes_per_launch_out) {
  TL_REQUIRE(m && m->finalized && iters > 0 && warmup >= 0 && avg_us_out, "tl_model_bench_kernel: bad argument");
  TL_REQUIRE(kind == 1 || kind == 4 || kind == 5 || kind == 6 || kind == 7 || kind == 9, "tl_model_bench_kernel: kind %d", kind);
  if (kind == 9) {  // the step's all-reduce of f32[hidden] (COLLECTIVE: every rank of the group makes the same call)
    TL_REQUIRE(m->tp > 1 && m->linked(), "tl_model_bench_kernel: kind 9 needs a linked tensor-parallel shard");
    hipStream_t s = m->stream;
    hipEvent_t e0, e1;
    TL_HIP(hipEventCreate(&e0));
    TL_HIP(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) TL_TRY(tp_all_reduce(m, s));
    TL_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) TL_TRY(tp_all_reduce(m, s));
    TL_HIP(hipEventRecord(e1, s));
    TL_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    TL_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *avg_us_out = ms * 1e3f / iters;
    if (bytes_per_launch_out) *bytes_per_launch_out = sizeof(float) * (double)m->cfg.hidden_size;
    return tp_check_peer(m);
  }
  // TL_BENCH_LAYER_SPAN=n cycles over the first n layers only (1: the weights stay in the 256 MiB Infinity Cache)
  int L = m->cfg.num_hidden_layers;
  if (const char* e = getenv("TL_BENCH_LAYER_SPAN")) {
    const int n = atoi(e);
    if (n >= 1 && n < L) L = n;
  }
  auto args_of = [&](int i) {
    const int l = i % L;
    GemvArgs a;
    switch (kind) {
      case 1: a = args_qkv(m, l); break;
      case 4: a = args_attn_out(m, l); break;
      case 5: a = args_gate_up(m, l); break;
      case 6: a = args_down(m, l); break;
      default: a = args_lm_head(m); break;
    }
    a.peer = nullptr;  // the timed launch alone: not a collective of the group
    return a;
  };
  double bytes = 0.0;
  for (int i = 0; i < iters; ++i) {
    const GemvArgs a = args_of(i);
    for (int k = 0; k < a.nseg; ++k)
      bytes += (double)a.seg[k].w.rows * a.seg[k].w.cols / tl_type_block_elems(a.seg[k].w.type) *
               (double)tl_type_block_bytes(a.seg[k].w.type);
  }
  hipStream_t s = m->stream;
  hipEvent_t e0, e1;
  TL_HIP(hipEventCreate(&e0));
  TL_HIP(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) TL_TRY(tl_launch_gemv(args_of(i), s));
  TL_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) TL_TRY(tl_launch_gemv(args_of(i), s));
  TL_HIP(hipEventRecord(e1, s));
  TL_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TL_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_us_out = ms * 1e3f / iters;
  if (bytes_per_launch_out) *bytes_per_launch_out = bytes / iters;
  return 0;
}

static int set_state(tl_model* m, int token, int pos, float penalty, const int* hist, int n_hist) {
  std::vector<int> hbuf(kStateHeaderInts + (n_hist > 0 ? n_hist : 0));
  hbuf[0] = token;
  hbuf[1] = pos;
  hbuf[2] = n_hist;
  memcpy(&hbuf[3], &penalty, sizeof(float));
  if (n_hist > 0) memcpy(&hbuf[kStateHeaderInts], hist, sizeof(int) * (size_t)n_hist);
  TL_HIP(hipMemcpyAsync(m->state, hbuf.data(), hbuf.size() * sizeof(int), hipMemcpyHostToDevice, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));  // hbuf is a stack-lifetime buffer
  return 0;
}

extern "C" int tl_model_finalize(tl_model* m) {
  TL_REQUIRE(m, "tl_model_finalize: null handle");
  if (m->finalized) return 0;
  const tl_model_config& c = m->cfg;
  TL_REQUIRE(m->tok && m->out && m->out_norm, "tl_model_finalize: token_embd / output / output_norm missing "
             "(the reference has no tied-embedding fallback either, model.cpp:797)");
  m->weight_bytes = gguf_bytes(m->tok, 1) + gguf_bytes(m->out, m->out->rows);
  for (int l = 0; l < c.num_hidden_layers; ++l) {
    const LayerW& L = m->layers[l];
    for (int k = 0; k < 7; ++k) {
      TL_REQUIRE(L.q[k], "tl_model_finalize: layer %d is missing tensor kind %d", l, k);
      m->weight_bytes += gguf_bytes(L.q[k], L.q[k]->rows);
    }
    TL_REQUIRE(L.attn_norm && L.ffn_norm, "tl_model_finalize: layer %d is missing a norm weight", l);
  }
  const size_t H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, P = c.max_position_embeddings;
  m->kv_layer_elems = P * (size_t)m->kv_dim;
  TL_HIP(hipMalloc(&m->kv_base, sizeof(float) * 2 * m->kv_layer_elems * c.num_hidden_layers));
  TL_HIP(hipMemsetAsync(m->kv_base, 0, sizeof(float) * 2 * m->kv_layer_elems * c.num_hidden_layers, m->stream));
  TL_HIP(hipMalloc(&m->x, sizeof(float) * H));
  m->has_f32 = m->out->type == TL_F32;
  for (int l = 0; l < c.num_hidden_layers; ++l)
    for (int k = 0; k < 7; ++k) m->has_f32 |= m->layers[l].q[k]->type == TL_F32;
  if (m->has_f32) TL_HIP(hipMalloc(&m->xn, sizeof(float) * (H > (size_t)m->qd_l ? H : (size_t)m->qd_l)));
  TL_HIP(hipMalloc(&m->q, sizeof(float) * (size_t)m->qd_l));
  if (m->tp > 1) {
    TL_HIP(hipMalloc(&m->xp, sizeof(float) * H));
    TL_HIP(hipMalloc(&m->zeros, sizeof(float) * H));
    TL_HIP(hipMemsetAsync(m->zeros, 0, sizeof(float) * H, m->stream));
  }
  TL_HIP(hipMalloc(&m->k, sizeof(float) * m->kv_dim));
  TL_HIP(hipMalloc(&m->v, sizeof(float) * m->kv_dim));
  TL_HIP(hipMalloc(&m->att, sizeof(float) * (size_t)m->qd_l));
  TL_HIP(hipMalloc(&m->h, sizeof(float) * (size_t)m->I_l));
  TL_HIP(hipMalloc(&m->h_up, sizeof(float) * (size_t)m->I_l));
  (void)I;
  TL_HIP(hipMalloc(&m->logits, sizeof(float) * V));
  TL_HIP(hipMalloc(&m->rope, sizeof(float) * P * m->head_dim));
  TL_TRY(tl_rope_table_upload(m->rope, (int)P, m->head_dim, c.rope_theta, m->stream));
  TL_HIP(hipMalloc(&m->cand_v, sizeof(float) * ARGMAX_SLICES));
  TL_HIP(hipMalloc(&m->cand_i, sizeof(int) * ARGMAX_SLICES));
  m->attn_split = tl_attn_split_supported(m->nh_l, m->nkv_l, m->head_dim);
  if (m->attn_split) {  // partial records + one arrival counter per head group (zero between launches)
    const size_t ws_bytes = sizeof(float) * (tl_attn_ws_floats(m->nh_l, m->head_dim) + (size_t)m->nh_l);
    TL_HIP(hipMalloc(&m->attn_ws, ws_bytes));
    TL_HIP(hipMemsetAsync(m->attn_ws, 0, ws_bytes, m->stream));
  }
  m->hist_cap = (int)P + 64;
  TL_HIP(hipMalloc(&m->state, sizeof(int) * (kStateHeaderInts + m->hist_cap)));
  TL_HIP(hipMemsetAsync(m->state, 0, sizeof(int) * (kStateHeaderInts + m->hist_cap), m->stream));
  m->allocated = true;
  if (m->tp > 1 && !m->linked()) {
    // a shard without a communicator: only usable through tl_tp_local_group_decode (all shards in one process)
    TL_HIP(hipStreamSynchronize(m->stream));
    m->no_graph = true;
    m->finalized = true;
    return 0;
  }
  // one eager step (position 0) outside capture: sets kernel attributes and validates every launch. With the peer
  // exchange on, this step is a COLLECTIVE: every rank of the group must reach tl_model_finalize within the poll deadline
  // (TL_TP_PEER_TIMEOUT_MS, 5 s by default) of the others; put a host barrier before it (bench.py does). A rank that
  // waited in vain gets error 5 here and the group is out of step for good: free the models of all ranks.
  if (m->peer_on) TL_TRY(tl_peer_clear_error(m->peer, m->stream));
  TL_TRY(expand_prefill_planes(m));
  TL_TRY(plan_chain(m));
  TL_TRY(plan_fused_attention(m));
  TL_TRY(set_state(m, 0, 0, 1.0f, nullptr, 0));
  TL_TRY(enqueue_step(m));
  TL_HIP(hipStreamSynchronize(m->stream));
  TL_TRY(chain_check(m));
  if (tp_check_peer(m) != 0) {
    tl_set_error("tl_model_finalize: the warm-up step's peer-to-peer exchange timed out: the ranks of a tensor-parallel "
                 "group must call tl_model_finalize together (host barrier first); free this group's models");
    return 5;
  }
  if (const char* e = getenv("TL_NO_GRAPH")) m->no_graph = atoi(e) != 0;
  if (!m->no_graph) {  // capture the step once
    TL_HIP(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_step(m);
    hipError_t e = hipStreamEndCapture(m->stream, &m->graph);
    if (rc == 0 && e == hipSuccess) e = hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0);
    if (rc != 0 || e != hipSuccess) {
      if (!m->comm || m->peer_on) {
        if (rc != 0) return rc;
        TL_HIP(e);
      }
      // tensor parallel: a collective library that cannot be captured -> replay the step eagerly instead
      (void)hipGetLastError();
      m->graph_exec = nullptr;
      m->no_graph = true;
    }
  }
  // the warm-up step wrote position 0 of the cache: clear it again
  TL_HIP(hipMemsetAsync(m->kv_base, 0, sizeof(float) * 2 * m->kv_layer_elems * c.num_hidden_layers, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  m->finalized = true;
  return 0;
}

extern "C" int tl_model_reset_kv(tl_model* m) {
  TL_REQUIRE(m && m->finalized, "tl_model_reset_kv: model not finalized");
  TL_HIP(hipMemsetAsync(m->kv_base, 0, sizeof(float) * 2 * m->kv_layer_elems * m->cfg.num_hidden_layers, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

extern "C" int tl_model_decode(tl_model* m, int token_id, int pos, float* logits_host, int* next_token_host) {
  TL_REQUIRE(m && m->finalized, "tl_model_decode: model not finalized");
  TL_REQUIRE(m->tp == 1 || m->linked(), "tl_model_decode: a shard without a communicator runs only in tl_tp_local_group_decode");
  TL_REQUIRE(pos >= 0 && pos < m->cfg.max_position_embeddings, "tl_model_decode: pos %d outside [0,%d)", pos,
             m->cfg.max_position_embeddings);
  TL_TRY(set_state(m, token_id, pos, 1.0f, nullptr, 0));
  if (m->no_graph) TL_TRY(enqueue_step(m));
  else TL_HIP(hipGraphLaunch(m->graph_exec, m->stream));
  if (logits_host)
    TL_HIP(hipMemcpyAsync(logits_host, m->logits, sizeof(float) * (size_t)m->cfg.vocab_size, hipMemcpyDeviceToHost, m->stream));
  if (next_token_host) TL_HIP(hipMemcpyAsync(next_token_host, m->state, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  TL_TRY(chain_check(m));
  return tp_check_peer(m);
}

extern "C" int tl_model_generate_greedy(tl_model* m, int first_token, int start_pos, int n_steps, float penalty,
                                        const int* history_host, int n_history, int* out_ids_host) {
  TL_REQUIRE(m && m->finalized, "tl_model_generate_greedy: model not finalized");
  TL_REQUIRE(m->tp == 1 || m->linked(), "tl_model_generate_greedy: a shard without a communicator runs only in tl_tp_local_group_decode");
  TL_REQUIRE(n_steps >= 0 && start_pos >= 0 && start_pos + n_steps <= m->cfg.max_position_embeddings,
             "tl_model_generate_greedy: positions [%d,%d) exceed max_position_embeddings %d", start_pos, start_pos + n_steps,
             m->cfg.max_position_embeddings);
  TL_REQUIRE(n_history >= 0 && (n_history == 0 || history_host), "tl_model_generate_greedy: bad history");
  // only the last 64 ids matter for the penalty window (api.cpp:88-90)
  const int keep = n_history > 64 ? 64 : n_history;
  TL_REQUIRE(keep + n_steps <= m->hist_cap, "tl_model_generate_greedy: %d steps exceed the id history capacity", n_steps);
  TL_TRY(set_state(m, first_token, start_pos, penalty, history_host ? history_host + (n_history - keep) : nullptr, keep));
  for (int i = 0; i < n_steps; ++i) {
    if (m->no_graph) TL_TRY(enqueue_step(m));
    else TL_HIP(hipGraphLaunch(m->graph_exec, m->stream));
  }
  if (n_steps > 0 && out_ids_host)
    TL_HIP(hipMemcpyAsync(out_ids_host, m->state + kStateHeaderInts + keep, sizeof(int) * (size_t)n_steps,
                          hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  TL_TRY(chain_check(m));
  return tp_check_peer(m);
}

// Sampled generation entirely on the device: the decode step with tl_sample.hip's top-k / top-p sampler as
// its tail. temperature < 0.05 samples greedily (api.cpp:134). The uniform variates come from a counter-based
// generator keyed by `seed` (the reference seeds std::mt19937 from std::random_device: not reproducible either).
extern "C" int tl_model_generate_sampled(tl_model* m, int first_token, int start_pos, int n_steps, float penalty,
                                         const int* history_host, int n_history, float temperature, int top_k,
                                         float top_p, unsigned long long seed, int* out_ids_host) {
  TL_REQUIRE(m && m->finalized, "tl_model_generate_sampled: model not finalized");
  TL_REQUIRE(m->tp == 1 || m->linked(), "tl_model_generate_sampled: a shard without a communicator cannot step alone");
  TL_REQUIRE(n_steps >= 0 && start_pos >= 0 && start_pos + n_steps <= m->cfg.max_position_embeddings,
             "tl_model_generate_sampled: positions [%d,%d) exceed max_position_embeddings %d", start_pos,
             start_pos + n_steps, m->cfg.max_position_embeddings);
  TL_REQUIRE(n_history >= 0 && (n_history == 0 || history_host), "tl_model_generate_sampled: bad history");
  const int V = m->cfg.vocab_size;
  const int k = (top_k <= 0 || top_k > V) ? V : top_k;
  TL_REQUIRE(temperature < 0.05f || k <= TL_SAMPLE_MAX_TOP_K,
             "tl_model_generate_sampled: effective top_k %d exceeds the device sampler's limit %d (sample on the host)", k,
             TL_SAMPLE_MAX_TOP_K);
  const int keep = n_history > 64 ? 64 : n_history;
  TL_REQUIRE(keep + n_steps <= m->hist_cap, "tl_model_generate_sampled: %d steps exceed the id history capacity", n_steps);
  if (!m->sparams) {
    TL_HIP(hipMalloc(&m->sparams, sizeof(TlSampleParams)));
    TL_HIP(hipMalloc(&m->sample_probs, sizeof(float) * (size_t)V));
  }
  TlSampleParams h = {};
  h.temperature = temperature;
  h.top_k = top_k;
  h.top_p = top_p;
  h.seed = seed;
  TL_HIP(hipMemcpyAsync(m->sparams, &h, sizeof(h), hipMemcpyHostToDevice, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));  // h lives on this stack frame
  TL_TRY(set_state(m, first_token, start_pos, penalty, history_host ? history_host + (n_history - keep) : nullptr, keep));
  if (!m->no_graph && !m->graph_exec_sampled) {  // capture once; a refused capture falls back to eager launches
    TL_HIP(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step(m, nullptr, true);
    hipError_t e = hipStreamEndCapture(m->stream, &m->graph_sampled);
    if (rc == 0 && e == hipSuccess) e = hipGraphInstantiate(&m->graph_exec_sampled, m->graph_sampled, nullptr, nullptr, 0);
    if (rc != 0 || e != hipSuccess) {
      (void)hipGetLastError();
      m->graph_exec_sampled = nullptr;
    }
  }
  for (int i = 0; i < n_steps; ++i) {
    if (m->graph_exec_sampled) TL_HIP(hipGraphLaunch(m->graph_exec_sampled, m->stream));
    else TL_TRY(enqueue_step(m, nullptr, true));
  }
  if (n_steps > 0 && out_ids_host)
    TL_HIP(hipMemcpyAsync(out_ids_host, m->state + kStateHeaderInts + keep, sizeof(int) * (size_t)n_steps,
                          hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  TL_TRY(chain_check(m));
  return tp_check_peer(m);
}

// ---- batched prefill (forward_device_batch_prefill model.cpp:1526-1862; math of CPUBatchProcessor::forward_cpu_batch,
//      cpu_batch_processor.cpp:11-281) -----------------------------------------------------------------------------------
// A chunk of up to kPrefillChunk tokens goes through every layer as a batch: RMSNorm -> bf16 planes, q/k/v GEMMs on
// the matrix cores (tl_gemm.hip), batched RoPE + KV write, causal attention from the cache (tl_prefill.hip), O GEMM +
// residual, RMSNorm -> planes, gate/up GEMMs, SwiGLU -> planes, down GEMM + residual. Longer prompts are walked chunk by
// chunk over the growing cache. Short prompts (< prefill_min_batch) replay the decode graph token by token instead (same positions, KV contents and logits to ~1e-6).
constexpr int kPrefillChunk = 512;

__global__ void set_token_kernel(int* state, const int* tokens, int i) { state[0] = tokens[i]; }

static int ensure_prefill_ws(tl_model* m, bool all_logits) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  if (p.cap == 0) {
    const int cap = c.max_position_embeddings < kPrefillChunk ? c.max_position_embeddings : kPrefillChunk;
    const size_t H = c.hidden_size, maxk = H > (size_t)m->I_l ? H : (size_t)m->I_l;
    const size_t kmax = maxk > (size_t)m->qd_l ? maxk : (size_t)m->qd_l;
    p.plane_stride = tl_gemm_padded_tokens(cap) * kmax;
    TL_HIP(hipMalloc(&p.xb, sizeof(float) * cap * H));
    if (m->tp > 1) TL_HIP(hipMalloc(&p.xpart, sizeof(float) * cap * H));
    const char* dz = getenv("TL_GEMM_DEFER");  // 0: every K-split GEMM runs its own reduction launch (A/B runs)
    if (m->tp == 1 && !(dz && atoi(dz) == 0)) {
      // a split launch has fewer than 4 * CUs (slice, tile) pairs of at most 128 x 128 outputs
      const size_t rows_max = kmax > (size_t)c.vocab_size ? kmax : (size_t)c.vocab_size;
      size_t floats = (size_t)4 * tl_num_cus() * 128 * 128;
      if (floats > (size_t)8 * cap * rows_max) floats = (size_t)8 * cap * rows_max;
      TL_HIP(hipMalloc(&p.xb2, sizeof(float) * cap * H));
      for (int i = 0; i < 3; ++i) TL_HIP(hipMalloc(&p.part[i], sizeof(float) * floats));
      p.part_cap = floats;
    }
    TL_HIP(hipMalloc(&p.qb, sizeof(float) * cap * (size_t)m->qd_l));
    TL_HIP(hipMalloc(&p.kb, sizeof(float) * cap * (size_t)m->kv_dim));
    TL_HIP(hipMalloc(&p.vb, sizeof(float) * cap * (size_t)m->kv_dim));
    TL_HIP(hipMalloc(&p.ab, sizeof(float) * cap * (size_t)m->qd_l));
    TL_HIP(hipMalloc(&p.gb, sizeof(float) * cap * (size_t)m->I_l));
    TL_HIP(hipMalloc(&p.ub, sizeof(float) * cap * (size_t)m->I_l));
    TL_HIP(hipMalloc(&p.planes, tl_gemm_planes_bytes(p.plane_stride)));
    TL_HIP(hipMemsetAsync(p.planes, 0, tl_gemm_planes_bytes(p.plane_stride), m->stream));
    TL_HIP(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking));
    TL_HIP(hipEventCreateWithFlags(&p.fork, hipEventDisableTiming));
    TL_HIP(hipEventCreateWithFlags(&p.join, hipEventDisableTiming));
    p.cap = cap;
  }
  if (all_logits && !p.logits_all) TL_HIP(hipMalloc(&p.logits_all, sizeof(float) * p.cap * (size_t)c.vocab_size));
  return 0;
}

// One chunk (tokens toks_dev[0..T) at positions start_pos.., T <= pf.cap) in pieces, so that a tensor-parallel group can
// be driven in lockstep; the residual stream lives in pf.xb. tp > 1: the row-split GEMMs (attn_output, ffn_down) write
// this rank's partial for the whole chunk into pf.xpart (rank 0 folds the residual in) and ONE all-reduce of T x hidden
// floats per projection pair turns the partials into the new residual stream.
static int prefill_embed(tl_model* m, const int* toks_dev, int T, hipStream_t s) {
  m->pf.pend = {nullptr, 0, 0};  // (a chunk that failed half-way must not leave its pending projection to the next one)
  return tl_launch_dequant_gather(m->tok, toks_dev, T, m->pf.xb, s);
}
// RMSNorm of the residual stream -> activation planes; a projection still pending as K-split partials is added on the way
static int split_residual(tl_model* m, const float* norm_w, int T, hipStream_t s) {
  tl_model::Prefill& p = m->pf;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  if (!p.pend.S) return tl_launch_split(1, p.xb, norm_w, p.planes, p.plane_stride, T, m->cfg.hidden_size, terms, m->cfg.rms_norm_eps, s);
  TL_TRY(tl_launch_split(1, p.xb, norm_w, p.planes, p.plane_stride, T, m->cfg.hidden_size, terms, m->cfg.rms_norm_eps, s,
                         &p.pend, nullptr, p.xb2));
  std::swap(p.xb, p.xb2);
  p.pend = {nullptr, 0, 0};
  return 0;
}
// the residual stream complete in pf.xb (after the last layer, or wherever something other than a split reads it)
static int flush_residual(tl_model* m, hipStream_t s) {
  tl_model::Prefill& p = m->pf;
  if (!p.pend.S) return 0;
  TL_TRY(tl_launch_ksplit_reduce(p.pend, p.xb, p.xb, s));
  p.pend = {nullptr, 0, 0};
  return 0;
}
// a row-split projection (attn_output, ffn_down) added to the residual stream: now, or pending
static int project_residual(tl_model* m, const tl_qweight* w, int T, hipStream_t s) {
  tl_model::Prefill& p = m->pf;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  if (m->tp > 1 || !p.part_cap)
    return tl_launch_gemm_q(qw_view(w), p.planes, p.plane_stride, m->tp > 1 ? p.xpart : p.xb, m->rank == 0 ? p.xb : nullptr, T,
                            terms, s);
  GemmDefer d = {p.part[0], p.part_cap, {nullptr, 0, 0}};
  TL_TRY(tl_launch_gemm_q(qw_view(w), p.planes, p.plane_stride, p.xb, p.xb, T, terms, s, &d));
  p.pend = d.got;
  return 0;
}
// q, k, v of the T tokens whose planes are staged: into pf.qb / kb / vb, or (K split and the workspaces there) as partial
// tiles described by pf.dq / dk / dv .got for the RoPE + cache-write kernel to sum
static int qkv_deferred(tl_model* m, LayerW& L, int T, hipStream_t s) {
  tl_model::Prefill& p = m->pf;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  const size_t ps = p.plane_stride;
  p.dq = {p.part[0], p.part_cap, {nullptr, 0, 0}};
  p.dk = {p.part[1], p.part_cap, {nullptr, 0, 0}};
  p.dv = {p.part[2], p.part_cap, {nullptr, 0, 0}};
  const bool defer = p.part_cap != 0;
  // one launch where that changes no bit (runs of equal formats: Q4_K_M files hold v as Q6_K in some layers)
  bool merged = false;
  const QW w3[3] = {qw_view(L.q[TL_T_ATTN_Q]), qw_view(L.q[TL_T_ATTN_K]), qw_view(L.q[TL_T_ATTN_V])};
  float* const y3[3] = {p.qb, p.kb, p.vb};
  GemmDefer* const d3[3] = {&p.dq, &p.dk, &p.dv};
  const int same = w3[2].type == w3[0].type ? 3 : 2;
  GemmDefer* const* d3p = defer ? d3 : nullptr;
  if (w3[1].type == w3[0].type) TL_TRY(tl_launch_gemm_q_multi(w3, same, p.planes, ps, y3, T, terms, s, d3p, &merged));
  if (merged) {
    if (same == 2) TL_TRY(tl_launch_gemm_q(w3[2], p.planes, ps, p.vb, nullptr, T, terms, s, defer ? &p.dv : nullptr));
    return 0;
  }
  TL_HIP(hipEventRecord(p.fork, s));
  TL_HIP(hipStreamWaitEvent(p.side, p.fork, 0));
  TL_TRY(tl_launch_gemm_q(w3[0], p.planes, ps, p.qb, nullptr, T, terms, s, defer ? &p.dq : nullptr));
  TL_TRY(tl_launch_gemm_q(w3[1], p.planes, ps, p.kb, nullptr, T, terms, p.side, defer ? &p.dk : nullptr));
  TL_TRY(tl_launch_gemm_q(w3[2], p.planes, ps, p.vb, nullptr, T, terms, p.side, defer ? &p.dv : nullptr));
  TL_HIP(hipEventRecord(p.join, p.side));
  TL_HIP(hipStreamWaitEvent(s, p.join, 0));
  return 0;
}
static int prefill_attn_half(tl_model* m, int l, int T, int start_pos, hipStream_t s) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  LayerW& L = m->layers[l];
  const int H = c.hidden_size, hd = m->head_dim;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  const size_t ps = p.plane_stride;
  const float scale = 1.0f / sqrtf((float)hd);
  (void)H;
  TL_TRY(split_residual(m, L.attn_norm, T, s));
  if (p.part_cap && m->tp == 1) {
    // q | k | v with their K-split sums left to ONE RoPE + cache-write kernel (instead of three reductions, the rotation
    // and two cache copies), merged into one launch where that changes no bit
    TL_TRY(qkv_deferred(m, L, T, s));
    TL_TRY(tl_launch_batch_rope_kv(p.qb, p.kb, p.vb, m->kc(l), m->vc(l), 0, m->rope, nullptr, nullptr, T, m->nh_l, m->nkv_l, hd,
                                   c.max_position_embeddings, 1, s, &p.dq.got, &p.dk.got, &p.dv.got, start_pos));
  } else {
    TL_HIP(hipEventRecord(p.fork, s));
    TL_HIP(hipStreamWaitEvent(p.side, p.fork, 0));
    TL_TRY(tl_launch_gemm_q(qw_view(L.q[TL_T_ATTN_Q]), p.planes, ps, p.qb, nullptr, T, terms, s));
    TL_TRY(tl_launch_gemm_q(qw_view(L.q[TL_T_ATTN_K]), p.planes, ps, p.kb, nullptr, T, terms, p.side));
    TL_TRY(tl_launch_gemm_q(qw_view(L.q[TL_T_ATTN_V]), p.planes, ps, p.vb, nullptr, T, terms, p.side));
    TL_HIP(hipEventRecord(p.join, p.side));
    TL_HIP(hipStreamWaitEvent(s, p.join, 0));
    TL_TRY(tl_rope_batch(p.qb, p.kb, m->rope, T, m->nh_l, m->nkv_l, hd, start_pos, c.max_position_embeddings, 1, s));
    TL_TRY(tl_update_kv_cache_batch(m->kc(l), p.kb, start_pos, T, m->nkv_l, hd, c.max_position_embeddings, s));
    TL_TRY(tl_update_kv_cache_batch(m->vc(l), p.vb, start_pos, T, m->nkv_l, hd, c.max_position_embeddings, s));
  }
  // the attention output goes straight into the planes of the attn_output GEMM where the MFMA kernel serves the head shape
  // (no split launch; TL_ATTN_PLANES=0 keeps the launch, for A/B runs and the bit-identity test)
  const char* ap = getenv("TL_ATTN_PLANES");
  if (tl_attn_prefill_mfma_supported(m->nh_l, m->nkv_l, hd) && !(ap && atoi(ap) == 0)) {
    TL_TRY(tl_launch_attn_prefill_mfma(p.qb, m->kc(l), m->vc(l), p.ab, T, start_pos, m->nh_l, m->nkv_l, hd, scale, s, p.planes,
                                       ps, terms));
  } else {
    TL_TRY(tl_attention_batch_prefill(p.qb, m->kc(l), m->vc(l), p.ab, T, start_pos, c.max_position_embeddings, m->nh_l,
                                      m->nkv_l, hd, scale, s));
    TL_TRY(tl_launch_split(0, p.ab, nullptr, p.planes, ps, T, m->qd_l, terms, 0.f, s));
  }
  return project_residual(m, L.q[TL_T_ATTN_OUTPUT], T, s);
}
static int prefill_mlp_half(tl_model* m, int l, int T, hipStream_t s) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  LayerW& L = m->layers[l];
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  const size_t ps = p.plane_stride;
  (void)c;
  TL_TRY(split_residual(m, L.ffn_norm, T, s));
  GemmDefer dg = {p.part[0], p.part_cap, {nullptr, 0, 0}}, du = {p.part[1], p.part_cap, {nullptr, 0, 0}};
  const bool defer = p.part_cap != 0;
  bool merged = false;  // gate | up in one launch where that changes no bit
  const QW w2[2] = {qw_view(L.q[TL_T_FFN_GATE]), qw_view(L.q[TL_T_FFN_UP])};
  float* const y2[2] = {p.gb, p.ub};
  GemmDefer* const d2[2] = {&dg, &du};
  GemmDefer* const* d2p = defer ? d2 : nullptr;
  if (w2[0].type == w2[1].type) TL_TRY(tl_launch_gemm_q_multi(w2, 2, p.planes, ps, y2, T, terms, s, d2p, &merged));
  if (!merged) {
    TL_HIP(hipEventRecord(p.fork, s));
    TL_HIP(hipStreamWaitEvent(p.side, p.fork, 0));
    TL_TRY(tl_launch_gemm_q(w2[0], p.planes, ps, p.gb, nullptr, T, terms, s, defer ? &dg : nullptr));
    TL_TRY(tl_launch_gemm_q(w2[1], p.planes, ps, p.ub, nullptr, T, terms, p.side, defer ? &du : nullptr));
    TL_HIP(hipEventRecord(p.join, p.side));
    TL_HIP(hipStreamWaitEvent(s, p.join, 0));
  }
  TL_TRY(tl_launch_split(2, p.gb, p.ub, p.planes, ps, T, m->I_l, terms, 0.f, s, &dg.got, &du.got));
  return project_residual(m, L.q[TL_T_FFN_DOWN], T, s);
}
static int prefill_all_reduce(tl_model* m, int T, hipStream_t s) {
  if (m->tp == 1) return 0;
  TL_REQUIRE(m->comm, "tl_model: tp_size %d but tl_model_tp_init was not called", m->tp);
  TL_NCCL(ncclAllReduce(m->pf.xpart, m->pf.xb, (size_t)T * m->cfg.hidden_size, ncclFloat, ncclSum, m->comm, s));
  return 0;
}
// logits of the chunk's last token through the decode step's fused norm + lm_head GEMV (this rank's vocabulary rows)
static int prefill_head(tl_model* m, int T, hipStream_t s) {
  const size_t H = m->cfg.hidden_size;
  TL_HIP(hipMemcpyAsync(m->x, m->pf.xb + (size_t)(T - 1) * H, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
  return enqueue_head(m, s);
}

// the prompt's token ids on the device (buffer kept and grown across calls)
static int upload_tokens(tl_model* m, const int* tokens_host, int num_tokens, hipStream_t s) {
  tl_model::Prefill& p = m->pf;
  if (p.toks_cap < num_tokens) {
    TL_HIP(hipStreamSynchronize(s));
    if (p.toks) TL_HIP(hipFree(p.toks));
    p.toks = nullptr;
    p.toks_cap = 0;
    TL_HIP(hipMalloc(&p.toks, sizeof(int) * (size_t)num_tokens));
    p.toks_cap = num_tokens;
  }
  TL_HIP(hipMemcpyAsync(p.toks, tokens_host, sizeof(int) * (size_t)num_tokens, hipMemcpyHostToDevice, s));
  return 0;
}
static int prepare_prefill(tl_model* m, const int* tokens_host, int num_tokens, bool all_logits, hipStream_t s) {
  TL_TRY(ensure_prefill_ws(m, all_logits));
  return upload_tokens(m, tokens_host, num_tokens, s);
}

static int prefill_batched(tl_model* m, const int* tokens_host, int num_tokens, int start_pos, float* last_logits_host,
                           float* all_logits_host) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  hipStream_t s = m->stream;
  TL_TRY(prepare_prefill(m, tokens_host, num_tokens, all_logits_host != nullptr, s));
  const size_t H = c.hidden_size, V = c.vocab_size;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  for (int c0 = 0; c0 < num_tokens; c0 += p.cap) {
    const int T = num_tokens - c0 < p.cap ? num_tokens - c0 : p.cap;
    TL_TRY(prefill_embed(m, p.toks + c0, T, s));
    for (int l = 0; l < c.num_hidden_layers; ++l) {
      TL_TRY(prefill_attn_half(m, l, T, start_pos + c0, s));
      TL_TRY(prefill_all_reduce(m, T, s));
      TL_TRY(prefill_mlp_half(m, l, T, s));
      TL_TRY(prefill_all_reduce(m, T, s));
    }
    TL_TRY(flush_residual(m, s));
    if (all_logits_host) {  // final norm + lm_head for every token of the chunk (forward_cpu_logits_batch, model.cpp:1094-1156)
      TL_TRY(tl_launch_split(1, p.xb, m->out_norm, p.planes, p.plane_stride, T, (int)H, terms, c.rms_norm_eps, s));
      TL_TRY(tl_launch_gemm_q(qw_view(m->out), p.planes, p.plane_stride, p.logits_all, nullptr, T, terms, s));
      TL_HIP(hipMemcpyAsync(all_logits_host + (size_t)c0 * V, p.logits_all, sizeof(float) * (size_t)T * V,
                            hipMemcpyDeviceToHost, s));
    }
    if (c0 + T == num_tokens) {
      TL_TRY(prefill_head(m, T, s));
      TL_TRY(tp_gather_logits(m, s));
      if (last_logits_host)
        TL_HIP(hipMemcpyAsync(last_logits_host, m->logits, sizeof(float) * V, hipMemcpyDeviceToHost, s));
    }
  }
  TL_HIP(hipStreamSynchronize(s));
  return tp_check_peer(m);
}

// The batched prefill of a whole tensor-parallel group inside ONE process on ONE device (test hook, as above).
__global__ void group_sum_rows_kernel(const float* const* __restrict__ parts, float* const* __restrict__ outs, int n,
                                      size_t len) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  float acc = 0.f;
  for (int r = 0; r < n; ++r) acc += parts[r][i];
  for (int r = 0; r < n; ++r) outs[r][i] = acc;
}

extern "C" int tl_tp_local_group_prefill(tl_model* const* ms, int n, const int* tokens_host, int num_tokens,
                                         int start_pos, float* last_logits_host) {
  TL_REQUIRE(ms && n >= 2 && n <= 16 && tokens_host && num_tokens > 0, "tl_tp_local_group_prefill: bad argument");
  tl_model* m0 = ms[0];
  for (int r = 0; r < n; ++r)
    TL_REQUIRE(ms[r] && ms[r]->allocated && ms[r]->tp == n && ms[r]->rank == r,
               "tl_tp_local_group_prefill: member %d is not rank %d of a finalized %d-way group", r, r, n);
  TL_REQUIRE(start_pos >= 0 && start_pos + num_tokens <= m0->cfg.max_position_embeddings,
             "tl_tp_local_group_prefill: positions out of range");
  for (int r = 0; r < n; ++r) TL_REQUIRE(!ms[r]->has_f32, "tl_tp_local_group_prefill: F32 / BF16 matrices have no MFMA route");
  hipStream_t s = m0->stream;
  for (int r = 0; r < n; ++r) TL_TRY(prepare_prefill(ms[r], tokens_host, num_tokens, false, s));
  std::vector<const float*> parts(n);
  std::vector<float*> outs(n);
  for (int r = 0; r < n; ++r) {
    parts[r] = ms[r]->pf.xpart;
    outs[r] = ms[r]->pf.xb;
  }
  const float** parts_dev = nullptr;
  float** outs_dev = nullptr;
  TL_HIP(hipMalloc(&parts_dev, sizeof(float*) * n));
  TL_HIP(hipMalloc(&outs_dev, sizeof(float*) * n));
  TL_HIP(hipMemcpy(parts_dev, parts.data(), sizeof(float*) * n, hipMemcpyHostToDevice));
  TL_HIP(hipMemcpy(outs_dev, outs.data(), sizeof(float*) * n, hipMemcpyHostToDevice));
  int rc = 0;
  auto all = [&](auto&& f) {
    for (int r = 0; r < n && rc == 0; ++r) rc = f(ms[r]);
  };
  const int cap = m0->pf.cap, H = m0->cfg.hidden_size, V = m0->cfg.vocab_size;
  for (int c0 = 0; c0 < num_tokens && rc == 0; c0 += cap) {
    const int T = num_tokens - c0 < cap ? num_tokens - c0 : cap;
    const size_t len = (size_t)T * H;
    auto reduce = [&]() {
      if (rc == 0) group_sum_rows_kernel<<<(unsigned)((len + 255) / 256), 256, 0, s>>>(parts_dev, outs_dev, n, len);
    };
    all([&](tl_model* m) { return prefill_embed(m, m->pf.toks + c0, T, s); });
    for (int l = 0; l < m0->cfg.num_hidden_layers && rc == 0; ++l) {
      all([&](tl_model* m) { return prefill_attn_half(m, l, T, start_pos + c0, s); });
      reduce();
      all([&](tl_model* m) { return prefill_mlp_half(m, l, T, s); });
      reduce();
    }
    if (c0 + T == num_tokens) {
      all([&](tl_model* m) { return prefill_head(m, T, s); });
      for (int r = 1; r < n && rc == 0; ++r)  // rank r's vocabulary rows into rank 0's logits
        if (hipMemcpyAsync(m0->logits + (size_t)r * ms[r]->V_l, ms[r]->logits + (size_t)r * ms[r]->V_l,
                           sizeof(float) * (size_t)ms[r]->V_l, hipMemcpyDeviceToDevice, s) != hipSuccess)
          rc = 1;
      if (rc == 0 && last_logits_host &&
          hipMemcpyAsync(last_logits_host, m0->logits, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, s) != hipSuccess)
        rc = 1;
    }
  }
  const hipError_t e = hipStreamSynchronize(s);
  hipFree(parts_dev);
  hipFree(outs_dev);
  if (rc > 1) return rc;
  TL_REQUIRE(rc == 0, "tl_tp_local_group_prefill: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
  TL_HIP(e);
  return 0;
}


// ---- multi-sequence decode (generate_batch api.cpp:836-1142, batch_generation_parallel :1423,
//      forward_cpu_batch_generation model.cpp:1158-1523 / forward_device_batch_generation :1863-2114) -------------------
// B sequences advance one token each per step. The projections of the B tokens run as skinny GEMMs on the matrix cores
// (the batched-prefill kernels: the weights are streamed once per step for all sequences), RoPE / KV write / attention
// work per token on its own sequence's region of a sequence-major cache (tl_batch.hip), all-token logits come from one
// more GEMM, and a per-row arg-max writes the next token ids and advances the positions on the device, so that a greedy
// generation loop needs no host round trip per step. Every sequence decodes exactly as it would alone (see tl_batch.hip
// on the reference's history_len quirk). tp_size 1 only.
static int batch_attn_half(tl_model* m, int l, int T, hipStream_t s) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  LayerW& L = m->layers[l];
  const int H = c.hidden_size, hd = m->head_dim;
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  const size_t ps = p.plane_stride;
  const float scale = 1.0f / sqrtf((float)hd);
  (void)H;
  TL_TRY(split_residual(m, L.attn_norm, T, s));
  TL_TRY(qkv_deferred(m, L, T, s));
  TL_TRY(tl_launch_batch_rope_kv(p.qb, p.kb, p.vb, m->bkc(l), m->bvc(l), m->kv_layer_elems, m->rope, m->bt.pos, m->bt.seq, T,
                                 m->nh_l, m->nkv_l, hd, c.max_position_embeddings, m->bt.max_batch, s, &p.dq.got, &p.dk.got,
                                 &p.dv.got));
  // the attention output goes straight into the planes of the attn_output GEMM (no split launch; TL_ATTN_PLANES=0 keeps
  // the launch, for A/B runs and the bit-identity test)
  const char* ap = getenv("TL_ATTN_PLANES");
  if (ap && atoi(ap) == 0) {
    TL_TRY(tl_launch_batch_attn(p.qb, m->bkc(l), m->bvc(l), m->kv_layer_elems, p.ab, m->bt.pos, m->bt.seq, T, m->nh_l,
                                m->nkv_l, hd, c.max_position_embeddings, m->bt.max_batch, scale, s));
    TL_TRY(tl_launch_split(0, p.ab, nullptr, p.planes, ps, T, m->qd_l, terms, 0.f, s));
  } else {
    TL_TRY(tl_launch_batch_attn(p.qb, m->bkc(l), m->bvc(l), m->kv_layer_elems, p.ab, m->bt.pos, m->bt.seq, T, m->nh_l,
                                m->nkv_l, hd, c.max_position_embeddings, m->bt.max_batch, scale, s, p.planes, ps, terms));
  }
  return project_residual(m, L.q[TL_T_ATTN_OUTPUT], T, s);
}

// The same step for a FEW sequences (T <= batch_gemv_max, default 4): the projections as multi-vector GEMVs
// (gemv_nv_kernel: one pass over the weights for up to four tokens, RMSNorm / residual / SwiGLU fused as in the
// single-sequence step) instead of matrix-core GEMMs, whose cost per step does not fall below that of 128 tokens.
static bool batch_gemv_ok(const tl_model* m) {
  auto ok = [](const tl_qweight* w) { return w->type == TL_Q4_K || w->type == TL_Q6_K || w->type == TL_Q8_0; };
  if (!ok(m->out) || tl_gemv_nv_capacity(m->cfg.hidden_size) < 2 || tl_gemv_nv_capacity(m->I_l) < 2 ||
      tl_gemv_nv_capacity(m->qd_l) < 2)
    return false;
  for (const LayerW& L : m->layers)
    for (int k = 0; k < 7; ++k)
      if (!ok(L.q[k])) return false;
  return true;
}
static int batch_layers_gemv(tl_model* m, int T, hipStream_t s) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  const int hd = m->head_dim;
  const float scale = 1.0f / sqrtf((float)hd);
  TL_TRY(prefill_embed(m, m->bt.toks, T, s));
  for (int l = 0; l < c.num_hidden_layers; ++l) {
    LayerW& L = m->layers[l];
    {  // RMSNorm + q / k / v (one launch per run of equal formats)
      const GemvSeg segs[3] = {{qw_view(L.q[TL_T_ATTN_Q]), p.qb}, {qw_view(L.q[TL_T_ATTN_K]), p.kb}, {qw_view(L.q[TL_T_ATTN_V]), p.vb}};
      for (int i = 0; i < 3;) {
        int j = i + 1;
        while (j < 3 && segs[j].w.type == segs[i].w.type) ++j;
        GemvArgs a = {};
        for (int k = i; k < j; ++k) a.seg[k - i] = segs[k];
        a.nseg = j - i;
        a.cols = c.hidden_size;
        a.x = p.xb;
        a.norm_w = L.attn_norm;
        a.eps = c.rms_norm_eps;
        TL_TRY(tl_launch_gemv_nv(a, T, s));
        i = j;
      }
    }
    TL_TRY(tl_launch_batch_rope_kv(p.qb, p.kb, p.vb, m->bkc(l), m->bvc(l), m->kv_layer_elems, m->rope, m->bt.pos, m->bt.seq, T,
                                   m->nh_l, m->nkv_l, hd, c.max_position_embeddings, m->bt.max_batch, s));
    TL_TRY(tl_launch_batch_attn(p.qb, m->bkc(l), m->bvc(l), m->kv_layer_elems, p.ab, m->bt.pos, m->bt.seq, T, m->nh_l,
                                m->nkv_l, hd, c.max_position_embeddings, m->bt.max_batch, scale, s));
    {  // attn_output + residual
      GemvArgs a = {};
      a.seg[0] = {qw_view(L.q[TL_T_ATTN_OUTPUT]), p.xb};
      a.nseg = 1;
      a.cols = m->qd_l;
      a.x = p.ab;
      a.resid = p.xb;
      TL_TRY(tl_launch_gemv_nv(a, T, s));
    }
    {  // RMSNorm + gate / up + SwiGLU
      GemvArgs a = {};
      a.cols = c.hidden_size;
      a.x = p.xb;
      a.norm_w = L.ffn_norm;
      a.eps = c.rms_norm_eps;
      if (L.q[TL_T_FFN_GATE]->type == L.q[TL_T_FFN_UP]->type) {
        a.seg[0] = {qw_view(L.q[TL_T_FFN_GATE]), p.gb};
        a.seg[1] = {qw_view(L.q[TL_T_FFN_UP]), p.ub};
        a.nseg = 2;
        a.pair = 1;
        TL_TRY(tl_launch_gemv_nv(a, T, s));
      } else {
        a.nseg = 1;
        a.seg[0] = {qw_view(L.q[TL_T_FFN_GATE]), p.gb};
        TL_TRY(tl_launch_gemv_nv(a, T, s));
        a.seg[0] = {qw_view(L.q[TL_T_FFN_UP]), p.ub};
        TL_TRY(tl_launch_gemv_nv(a, T, s));
        TL_TRY(tl_swiglu(p.gb, p.ub, p.gb, (size_t)T * m->I_l, 0, s));  // decode form, as the fused pair epilogue
      }
    }
    {  // ffn_down + residual
      GemvArgs a = {};
      a.seg[0] = {qw_view(L.q[TL_T_FFN_DOWN]), p.xb};
      a.nseg = 1;
      a.cols = m->I_l;
      a.x = p.gb;
      a.resid = p.xb;
      TL_TRY(tl_launch_gemv_nv(a, T, s));
    }
  }
  return 0;
}

// one step for the T tokens in bt.toks / bt.pos / bt.seq; ids_dev: [T] next ids (first maximum of each logits row);
// advance: the ids become the next tokens and the positions move on (device-resident loop)
// all layers for the T tokens in bt.toks / bt.pos / bt.seq: the residual stream ends up in pf.xb [T][hidden]
static int batch_layers(tl_model* m, int T, hipStream_t s) {
  TL_REQUIRE(T >= 1 && T <= m->pf.cap, "tl_model: a batched step of %d tokens exceeds the chunk workspaces (%d)", T, m->pf.cap);
  if (T <= m->bt.gemv_max) return batch_layers_gemv(m, T, s);
  TL_TRY(prefill_embed(m, m->bt.toks, T, s));
  for (int l = 0; l < m->cfg.num_hidden_layers; ++l) {
    TL_TRY(batch_attn_half(m, l, T, s));
    TL_TRY(prefill_mlp_half(m, l, T, s));
  }
  return flush_residual(m, s);
}
// final RMSNorm + lm_head of R hidden rows -> pf.logits_all [R][vocab]
static int batch_head(tl_model* m, const float* hidden, int R, hipStream_t s) {
  TL_REQUIRE(R >= 1 && R <= m->pf.cap, "tl_model: %d logits rows exceed the chunk workspaces (%d)", R, m->pf.cap);
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  if (R <= m->bt.gemv_max) {
    GemvArgs f = {};
    f.seg[0] = {qw_view(m->out), p.logits_all};
    f.nseg = 1;
    f.cols = c.hidden_size;
    f.x = hidden;
    f.norm_w = m->out_norm;
    f.eps = c.rms_norm_eps;
    return tl_launch_gemv_nv(f, R, s);
  }
  const int terms = m->prefill_terms ? m->prefill_terms : tl_gemm_default_terms();
  TL_TRY(tl_launch_split(1, hidden, m->out_norm, p.planes, p.plane_stride, R, c.hidden_size, terms, c.rms_norm_eps, s));
  return tl_launch_gemm_q(qw_view(m->out), p.planes, p.plane_stride, p.logits_all, nullptr, R, terms, s);
}

static int batch_step(tl_model* m, int T, int* ids_dev, bool advance, float* logits_host) {
  const tl_model_config& c = m->cfg;
  tl_model::Prefill& p = m->pf;
  hipStream_t s = m->stream;
  TL_TRY(batch_layers(m, T, s));
  TL_TRY(batch_head(m, p.xb, T, s));
  if (logits_host)
    TL_HIP(hipMemcpyAsync(logits_host, p.logits_all, sizeof(float) * (size_t)T * c.vocab_size, hipMemcpyDeviceToHost, s));
  return tl_launch_batch_argmax_advance(p.logits_all, T, c.vocab_size, advance ? m->bt.toks : nullptr,
                                        advance ? m->bt.pos : nullptr, ids_dev, s);
}

extern "C" int tl_model_batch_init(tl_model* m, int max_batch) {
  TL_REQUIRE(m && m->finalized, "tl_model_batch_init: model not finalized");
  TL_REQUIRE(m->tp == 1, "tl_model_batch_init: multi-sequence decode is not available on tensor-parallel shards");
  TL_REQUIRE(!m->has_f32, "tl_model_batch_init: a model with F32 / BF16 projection matrices has no multi-sequence step "
                          "(its matrices run the unfused F32 route; decode the sequences one by one)");
  TL_REQUIRE(max_batch >= 1 && max_batch <= kPrefillChunk, "tl_model_batch_init: max_batch %d outside 1..%d", max_batch,
             kPrefillChunk);
  TL_REQUIRE(tl_batch_attn_supported(m->nh_l, m->nkv_l, m->head_dim),
             "tl_model_batch_init: head configuration %d/%d/%d is not supported by the batched decode attention", m->nh_l,
             m->nkv_l, m->head_dim);
  TL_REQUIRE(max_batch <= (m->cfg.max_position_embeddings < kPrefillChunk ? m->cfg.max_position_embeddings : kPrefillChunk),
             "tl_model_batch_init: max_batch %d exceeds the chunk capacity of this model", max_batch);
  TL_HIP(hipStreamSynchronize(m->stream));
  tl_model::Batch& b = m->bt;
  if (b.kv) {
    TL_HIP(hipFree(b.kv));
    TL_HIP(hipFree(b.toks));
    TL_HIP(hipFree(b.pos));
    TL_HIP(hipFree(b.seq));
    TL_HIP(hipFree(b.last_h));
    if (b.ids) TL_HIP(hipFree(b.ids));
    b = tl_model::Batch();
  }
  const size_t elems = 2 * (size_t)m->cfg.num_hidden_layers * max_batch * m->kv_layer_elems;
  TL_HIP(hipMalloc(&b.kv, sizeof(float) * elems));
  TL_HIP(hipMemsetAsync(b.kv, 0, sizeof(float) * elems, m->stream));
  TL_HIP(hipMalloc(&b.toks, sizeof(int) * kPrefillChunk));  // a packed prompt chunk carries up to kPrefillChunk tokens
  TL_HIP(hipMalloc(&b.pos, sizeof(int) * kPrefillChunk));
  TL_HIP(hipMalloc(&b.seq, sizeof(int) * kPrefillChunk));
  TL_HIP(hipMalloc(&b.last_h, sizeof(float) * (size_t)max_batch * m->cfg.hidden_size));
  b.max_batch = max_batch;
  b.gemv_max = batch_gemv_ok(m) ? 4 : 0;  // one pass of up to four vectors: 2.5 / 4.0 / 4.1 ms per step for 2 / 3 / 4 sequences; the
                                          // direct small-token GEMM route costs 5.3 ms whatever the count (two GEMV passes: 6.1 - 7.8)
  if (const char* e = getenv("TL_BATCH_GEMV_MAX")) {  // A/B runs: 0 = always the matrix-core GEMMs
    const int v = atoi(e);
    if (v >= 0 && v <= 64 && b.gemv_max) b.gemv_max = v;
  }
  TL_TRY(ensure_prefill_ws(m, true));
  TL_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

// the prompt of ONE sequence into its region of the batch cache (positions start_pos ..), batched on the matrix cores
extern "C" int tl_model_batch_prefill(tl_model* m, int seq_idx, const int* tokens_host, int num_tokens, int start_pos,
                                      float* last_logits_host) {
  TL_REQUIRE(m && m->bt.max_batch > 0, "tl_model_batch_prefill: call tl_model_batch_init first");
  TL_REQUIRE(seq_idx >= 0 && seq_idx < m->bt.max_batch, "tl_model_batch_prefill: sequence %d outside [0,%d)", seq_idx,
             m->bt.max_batch);
  TL_REQUIRE(tokens_host && num_tokens > 0 && start_pos >= 0 && start_pos + num_tokens <= m->cfg.max_position_embeddings,
             "tl_model_batch_prefill: positions [%d,%d) exceed max_position_embeddings %d", start_pos, start_pos + num_tokens,
             m->cfg.max_position_embeddings);
  m->alt_seq = seq_idx;
  const int rc = prefill_batched(m, tokens_host, num_tokens, start_pos, last_logits_host, nullptr);
  m->alt_seq = -1;
  return rc;
}

static int batch_upload(tl_model* m, const char* who, int n, const int* tokens, const int* positions, const int* seqs,
                        int extra_steps) {
  TL_REQUIRE(m && m->bt.max_batch > 0, "%s: call tl_model_batch_init first", who);
  TL_REQUIRE(n >= 1 && n <= m->bt.max_batch && tokens && positions && seqs, "%s: %d active sequences (1..%d)", who, n,
             m->bt.max_batch);
  std::vector<char> seen(m->bt.max_batch, 0);
  for (int i = 0; i < n; ++i) {
    TL_REQUIRE(seqs[i] >= 0 && seqs[i] < m->bt.max_batch && !seen[seqs[i]], "%s: sequence index %d out of range or repeated",
               who, seqs[i]);
    seen[seqs[i]] = 1;
    TL_REQUIRE(positions[i] >= 0 && positions[i] + extra_steps <= m->cfg.max_position_embeddings,
               "%s: position %d (+%d steps) exceeds max_position_embeddings %d", who, positions[i], extra_steps,
               m->cfg.max_position_embeddings);
    TL_REQUIRE(tokens[i] >= 0 && tokens[i] < m->cfg.vocab_size, "%s: token id %d outside the vocabulary", who, tokens[i]);
  }
  hipStream_t s = m->stream;
  TL_HIP(hipMemcpyAsync(m->bt.toks, tokens, sizeof(int) * n, hipMemcpyHostToDevice, s));
  TL_HIP(hipMemcpyAsync(m->bt.pos, positions, sizeof(int) * n, hipMemcpyHostToDevice, s));
  TL_HIP(hipMemcpyAsync(m->bt.seq, seqs, sizeof(int) * n, hipMemcpyHostToDevice, s));
  TL_HIP(hipStreamSynchronize(s));  // the host arrays belong to the caller
  return 0;
}
static int batch_ids(tl_model* m, size_t n);
// batch_prefill_parallel (api.cpp:1144-1421): the tokens of ALL prompts in one packed batch, token i of sequence
// seq_idx[i] at position positions[i]; attention is sequence-aware (attention_batch_cpu_sequence_aware,
// cpu_attention.cpp:254-325): every K/V row of a chunk is written before its attention runs, and a token reads rows
// 0..pos of its own sequence only. Tokens of one sequence must come in position order (as the reference packs them);
// chunks of up to pf.cap (<= kPrefillChunk) tokens. last_rows[j]: index of a token whose logits / first-maximum id are wanted
// (normally the last token of each prompt), at most max_batch of them.
extern "C" int tl_model_batch_prefill_packed(tl_model* m, int n_tokens, const int* tokens_host, const int* positions_host,
                                             const int* seq_idx_host, int n_last, const int* last_rows_host,
                                             float* last_logits_host, int* last_ids_host) {
  TL_REQUIRE(m && m->bt.max_batch > 0, "tl_model_batch_prefill_packed: call tl_model_batch_init first");
  TL_REQUIRE(n_tokens >= 1 && tokens_host && positions_host && seq_idx_host, "tl_model_batch_prefill_packed: empty batch");
  TL_REQUIRE(n_last >= 0 && n_last <= m->bt.max_batch && (n_last == 0 || last_rows_host),
             "tl_model_batch_prefill_packed: %d output rows (at most max_batch = %d)", n_last, m->bt.max_batch);
  const tl_model_config& c = m->cfg;
  std::vector<int> next_pos(m->bt.max_batch, -1);  // position order within a sequence (first occurrence sets the start)
  for (int i = 0; i < n_tokens; ++i) {
    const int sq = seq_idx_host[i], ps = positions_host[i];
    TL_REQUIRE(sq >= 0 && sq < m->bt.max_batch, "tl_model_batch_prefill_packed: sequence index %d outside [0,%d)", sq, m->bt.max_batch);
    TL_REQUIRE(ps >= 0 && ps < c.max_position_embeddings, "tl_model_batch_prefill_packed: position %d outside [0,%d)", ps,
               c.max_position_embeddings);
    TL_REQUIRE(next_pos[sq] < 0 || ps == next_pos[sq],
               "tl_model_batch_prefill_packed: the tokens of sequence %d are not in consecutive position order", sq);
    next_pos[sq] = ps + 1;
    TL_REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < c.vocab_size, "tl_model_batch_prefill_packed: token id %d outside the vocabulary",
               tokens_host[i]);
  }
  for (int j = 0; j < n_last; ++j)
    TL_REQUIRE(last_rows_host[j] >= 0 && last_rows_host[j] < n_tokens, "tl_model_batch_prefill_packed: output row %d outside the batch",
               last_rows_host[j]);
  hipStream_t s = m->stream;
  const size_t H = c.hidden_size;
  const int chunk = m->pf.cap;  // tokens the chunk workspaces hold: min(kPrefillChunk, max_position_embeddings)
  TL_REQUIRE(chunk >= 1 && chunk <= kPrefillChunk, "tl_model_batch_prefill_packed: workspaces are missing");
  for (int c0 = 0; c0 < n_tokens; c0 += chunk) {
    const int T = n_tokens - c0 < chunk ? n_tokens - c0 : chunk;
    TL_HIP(hipMemcpyAsync(m->bt.toks, tokens_host + c0, sizeof(int) * T, hipMemcpyHostToDevice, s));
    TL_HIP(hipMemcpyAsync(m->bt.pos, positions_host + c0, sizeof(int) * T, hipMemcpyHostToDevice, s));
    TL_HIP(hipMemcpyAsync(m->bt.seq, seq_idx_host + c0, sizeof(int) * T, hipMemcpyHostToDevice, s));
    TL_TRY(batch_layers(m, T, s));
    for (int j = 0; j < n_last; ++j) {
      const int r = last_rows_host[j] - c0;
      if (r >= 0 && r < T)
        TL_HIP(hipMemcpyAsync(m->bt.last_h + (size_t)j * H, m->pf.xb + (size_t)r * H, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
    }
  }
  if (n_last > 0) {
    TL_TRY(batch_ids(m, (size_t)n_last));
    TL_TRY(batch_head(m, m->bt.last_h, n_last, s));
    if (last_logits_host)
      TL_HIP(hipMemcpyAsync(last_logits_host, m->pf.logits_all, sizeof(float) * (size_t)n_last * c.vocab_size,
                            hipMemcpyDeviceToHost, s));
    TL_TRY(tl_launch_batch_argmax_advance(m->pf.logits_all, n_last, c.vocab_size, nullptr, nullptr, m->bt.ids, s));
    if (last_ids_host)
      TL_HIP(hipMemcpyAsync(last_ids_host, m->bt.ids, sizeof(int) * n_last, hipMemcpyDeviceToHost, s));
  }
  TL_HIP(hipStreamSynchronize(s));  // also: the host arrays belong to the caller
  return 0;
}

static int batch_ids(tl_model* m, size_t n) {
  if (m->bt.ids_cap < n) {
    if (m->bt.ids) TL_HIP(hipFree(m->bt.ids));
    m->bt.ids = nullptr;
    m->bt.ids_cap = 0;
    TL_HIP(hipMalloc(&m->bt.ids, sizeof(int) * n));
    m->bt.ids_cap = n;
  }
  return 0;
}

// one step: token i of sequence seq_idx[i] at position positions[i] (batch_generation_parallel's arguments);
// logits_host [n][vocab] and next_tokens_host [n] (first maximum of each row) may be NULL
extern "C" int tl_model_batch_decode(tl_model* m, int n, const int* tokens_host, const int* positions_host,
                                     const int* seq_idx_host, float* logits_host, int* next_tokens_host) {
  TL_TRY(batch_upload(m, "tl_model_batch_decode", n, tokens_host, positions_host, seq_idx_host, 1));
  TL_TRY(batch_ids(m, (size_t)n));
  TL_TRY(batch_step(m, n, m->bt.ids, false, logits_host));
  if (next_tokens_host)
    TL_HIP(hipMemcpyAsync(next_tokens_host, m->bt.ids, sizeof(int) * n, hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

// n_steps greedy steps for n sequences without a host round trip per step: out_ids_host [n_steps][n], sequence i's
// ids in column i (the caller cuts a column at its EOS; the reference drops finished sequences from the batch, which
// changes nothing for the others: sequences never interact)
extern "C" int tl_model_batch_generate_greedy(tl_model* m, int n, const int* first_tokens_host, const int* start_pos_host,
                                              const int* seq_idx_host, int n_steps, int* out_ids_host) {
  TL_REQUIRE(n_steps >= 0 && out_ids_host, "tl_model_batch_generate_greedy: bad argument");
  TL_TRY(batch_upload(m, "tl_model_batch_generate_greedy", n, first_tokens_host, start_pos_host, seq_idx_host, n_steps));
  if (n_steps == 0) return 0;
  TL_TRY(batch_ids(m, (size_t)n * n_steps));
  for (int i = 0; i < n_steps; ++i) TL_TRY(batch_step(m, n, m->bt.ids + (size_t)i * n, true, nullptr));
  TL_HIP(hipMemcpyAsync(out_ids_host, m->bt.ids, sizeof(int) * (size_t)n * n_steps, hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

extern "C" int tl_model_set_prefill(tl_model* m, int terms, int min_batch) {
  TL_REQUIRE(m, "tl_model_set_prefill: null model");
  TL_REQUIRE(terms == 0 || terms == 1 || terms == 3 || terms == 6, "tl_model_set_prefill: terms must be 0, 1, 3 or 6 (got %d)", terms);
  TL_REQUIRE(min_batch >= 1, "tl_model_set_prefill: min_batch must be >= 1");
  m->prefill_terms = terms;
  m->prefill_min_batch = min_batch;
  return 0;
}

extern "C" int tl_model_prefill(tl_model* m, const int* tokens_host, int num_tokens, int start_pos,
                                float* last_logits_host, float* all_logits_host) {
  TL_REQUIRE(m && m->finalized, "tl_model_prefill: model not finalized");
  TL_REQUIRE(m->tp == 1 || m->linked(), "tl_model_prefill: a shard without a communicator runs only in tl_tp_local_group_decode");
  TL_REQUIRE(tokens_host && num_tokens > 0, "tl_model_prefill: empty prompt");
  TL_REQUIRE(start_pos >= 0 && start_pos + num_tokens <= m->cfg.max_position_embeddings,
             "tl_model_prefill: positions [%d,%d) exceed max_position_embeddings %d", start_pos, start_pos + num_tokens,
             m->cfg.max_position_embeddings);
  // (all-token logits of a vocabulary-sharded lm_head are not gathered: that request takes the token-by-token route)
  // (the T x hidden all-reduces of a sharded batched prefill are bandwidth-sized: they stay with RCCL; a group that is
  //  linked by the peer exchange only takes the token-by-token route too)
  // (a model with F32 matrices has no MFMA route: token by token through the decode step)
  if (num_tokens >= m->prefill_min_batch && !m->has_f32 && (m->tp == 1 || (!all_logits_host && m->comm)))
    return prefill_batched(m, tokens_host, num_tokens, start_pos, last_logits_host, all_logits_host);
  TL_TRY(upload_tokens(m, tokens_host, num_tokens, m->stream));
  TL_TRY(set_state(m, tokens_host[0], start_pos, 1.0f, nullptr, 0));
  const size_t V = (size_t)m->cfg.vocab_size;
  for (int i = 0; i < num_tokens; ++i) {
    set_token_kernel<<<1, 1, 0, m->stream>>>(m->state, m->pf.toks, i);
    if (m->no_graph) TL_TRY(enqueue_step(m));
    else TL_HIP(hipGraphLaunch(m->graph_exec, m->stream));
    if (all_logits_host)
      TL_HIP(hipMemcpyAsync(all_logits_host + (size_t)i * V, m->logits, sizeof(float) * V, hipMemcpyDeviceToHost, m->stream));
  }
  if (last_logits_host)
    TL_HIP(hipMemcpyAsync(last_logits_host, m->logits, sizeof(float) * V, hipMemcpyDeviceToHost, m->stream));
  TL_HIP(hipStreamSynchronize(m->stream));
  return tp_check_peer(m);
}

extern "C" int tl_model_profile_decode(tl_model* m, int first_token, int start_pos, int n_steps, float* kind_ms_out,
                                       int* kind_launches_out) {
  TL_REQUIRE(m && m->finalized && kind_ms_out && kind_launches_out, "tl_model_profile_decode: bad argument");
  TL_REQUIRE(m->tp == 1 || m->linked(), "tl_model_profile_decode: a shard without a communicator cannot step alone");
  TL_REQUIRE(n_steps > 0 && start_pos >= 0 && start_pos + n_steps <= m->cfg.max_position_embeddings,
             "tl_model_profile_decode: positions out of range");
  for (int k = 0; k < 9; ++k) {
    kind_ms_out[k] = 0.f;
    kind_launches_out[k] = 0;
  }
  TL_TRY(set_state(m, first_token, start_pos, 1.0f, nullptr, 0));
  for (int i = 0; i < n_steps; ++i) {
    StepProf prof;
    int rc = enqueue_step(m, &prof);
    hipError_t e = hipStreamSynchronize(m->stream);
    if (rc == 0 && e == hipSuccess) {
      for (size_t j = 1; j < prof.ev.size(); ++j) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, prof.ev[j - 1], prof.ev[j]) == hipSuccess) {
          kind_ms_out[prof.kind[j]] += ms;
          kind_launches_out[prof.kind[j]] += 1;
        }
      }
    }
    for (hipEvent_t ev : prof.ev) hipEventDestroy(ev);
    if (rc != 0) return rc;
    TL_HIP(e);
  }
  return 0;
}

extern "C" int tl_model_device_ptr(tl_model* m, int which, int layer, void** dev_out) {
  TL_REQUIRE(m && m->finalized && dev_out, "tl_model_device_ptr: bad argument");
  if (which == 2) {
    *dev_out = m->x;
    return 0;
  }
  TL_REQUIRE(layer >= 0 && layer < m->cfg.num_hidden_layers, "tl_model_device_ptr: layer %d out of range", layer);
  TL_REQUIRE(which == 0 || which == 1, "tl_model_device_ptr: unknown buffer %d", which);
  *dev_out = which == 0 ? m->kc(layer) : m->vc(layer);
  return 0;
}

extern "C" int tl_model_prefill_planes_bytes(const tl_model* m, size_t* bytes_out) {
  TL_REQUIRE(m && m->finalized && bytes_out, "tl_model_prefill_planes_bytes: bad argument");
  *bytes_out = m->expanded_bytes;
  return 0;
}

extern "C" int tl_model_bytes_per_token(const tl_model* m, int pos, double* weight_bytes, double* kv_bytes) {
  TL_REQUIRE(m && m->finalized, "tl_model_bytes_per_token: model not finalized");
  if (weight_bytes) *weight_bytes = m->weight_bytes;
  if (kv_bytes)  // read K and V of positions 0..pos, write one position
    *kv_bytes = (double)m->cfg.num_hidden_layers * 2.0 * ((double)pos + 2.0) * m->kv_dim * sizeof(float);
  return 0;
}

extern "C" int tl_model_graph_captured(const tl_model* m, int* captured_out) {
  TL_REQUIRE(m && m->finalized && captured_out, "tl_model_graph_captured: bad argument");
  *captured_out = m->graph_exec != nullptr && !m->no_graph;
  return 0;
}

extern "C" int tl_model_graph_nodes(const tl_model* m, int* nodes_out) {
  TL_REQUIRE(m && m->finalized && nodes_out, "tl_model_graph_nodes: bad argument");
  *nodes_out = 0;
  if (!m->graph || m->no_graph) return 0;
  size_t n = 0;
  TL_HIP(hipGraphGetNodes(m->graph, nullptr, &n));
  *nodes_out = (int)n;
  return 0;
}

extern "C" int tl_tp_unique_id(void* id_out_128) {
  TL_REQUIRE(id_out_128, "tl_tp_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  TL_NCCL(ncclGetUniqueId(&id));
  memcpy(id_out_128, &id, sizeof(id));
  return 0;
}
extern "C" int tl_model_tp_init(tl_model* m, const void* unique_id_128) {
  TL_REQUIRE(m && unique_id_128, "tl_model_tp_init: null argument");
  TL_REQUIRE(!m->finalized, "tl_model_tp_init: call before tl_model_finalize (the decode graph captures the collectives)");
  // tp_size 1 needs no communicator; TL_TP_SELFTEST=1 makes a one-rank communicator anyway, so that the RCCL calls and
  // their graph capture can be exercised on a single GPU (tests/test_tp_gpu.py)
  const char* st = getenv("TL_TP_SELFTEST");
  if (m->tp == 1 && !(st && atoi(st) != 0)) return 0;
  ncclUniqueId id;
  memcpy(&id, unique_id_128, sizeof(id));
  TL_NCCL(ncclCommInitRank(&m->comm, m->tp, id, m->rank));
  return 0;
}


// ---- peer-to-peer exchange wiring (tl_peer.hip) ------------------------------------------------------------------------
extern "C" int tl_model_tp_peer_export(tl_model* m, void* handle_out_64) {
  TL_REQUIRE(m && handle_out_64, "tl_model_tp_peer_export: null argument");
  TL_REQUIRE(!m->finalized, "tl_model_tp_peer_export: call before tl_model_finalize (the decode graph captures the exchanges)");
  TL_REQUIRE(m->tp >= 2 && m->tp <= TL_PEER_MAX_RANKS, "tl_model_tp_peer_export: tp_size %d outside 2..%d", m->tp,
             TL_PEER_MAX_RANKS);
  TL_REQUIRE(!m->peer, "tl_model_tp_peer_export: already exported");
  return tl_peer_create(m->tp, m->rank, m->cfg.hidden_size, m->V_l, &m->peer, handle_out_64);
}
extern "C" int tl_model_tp_peer_import(tl_model* m, const void* handles, int* ok_out) {
  TL_REQUIRE(m && handles && ok_out, "tl_model_tp_peer_import: null argument");
  TL_REQUIRE(m->peer && !m->finalized, "tl_model_tp_peer_import: call after tl_model_tp_peer_export and before tl_model_finalize");
  *ok_out = 0;
  TL_TRY(tl_peer_connect(m->peer, handles));
  return tl_peer_selftest(m->peer, m->stream, ok_out);
}
extern "C" int tl_model_tp_peer_enable(tl_model* m, int on) {
  TL_REQUIRE(m && !m->finalized, "tl_model_tp_peer_enable: call before tl_model_finalize");
  TL_REQUIRE(!on || (m->peer && tl_peer_fused_reduce(m->peer)), "tl_model_tp_peer_enable: nothing imported (tl_model_tp_peer_import first)");
  m->peer_on = on != 0;
  if (const char* e = getenv("TL_TP_PEER_FUSED")) m->peer_fused = atoi(e) != 0;
  return 0;
}
extern "C" int tl_model_tp_peer_fused(const tl_model* m, int* on_out) {
  TL_REQUIRE(m && on_out, "tl_model_tp_peer_fused: null argument");
  *on_out = m->peer_on && m->peer_fused ? 1 : 0;
  return 0;
}
extern "C" int tl_model_tp_peer_enabled(const tl_model* m, int* on_out) {
  TL_REQUIRE(m && on_out, "tl_model_tp_peer_enabled: null argument");
  *on_out = m->peer_on ? 1 : 0;
  return 0;
}
