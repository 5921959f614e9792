// tl_ops.hip — the glue operators around the GEMVs: RMSNorm, RoPE, KV-cache writes, decode / prefill
// attention over the device-resident cache, residual add, SwiGLU, arg-max sampling.
//
// Each replaces a wrapper of the reference's CUDA backend (cuda_kernels.h:104-643) but computes what the
// reference's CPU path computes (utils.cpp, model.cpp:609-663, cpu_attention.cpp), which is the parity target.
#include <math.h>

#include <vector>

#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// block-wide sum / max over up to 1024 threads through `red` (>= 32 floats of LDS); all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---- RMSNorm: one block per token (utils.cpp:613-673) -------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ out, int n, float eps) {
  __shared__ float red[32];
  const float* xt = x + (size_t)blockIdx.x * n;
  float* ot = out + (size_t)blockIdx.x * n;
  float ssq = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) ssq += xt[i] * xt[i];
  const float tot = block_sum(ssq, red);
  const float nf = 1.0f / sqrtf(tot / (float)n + eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) ot[i] = xt[i] * nf * w[i];
}

int tl_launch_rmsnorm(const float* x, const float* w, float* out, int T, int n, float eps, int floor_eps,
                      hipStream_t s) {
  if (floor_eps && eps < 1e-5f) eps = 1e-5f;  // SAFE_MAX(eps, numeric::MIN_NORM_EPS), utils.cpp:666-667
  rmsnorm_kernel<<<T, 256, 0, s>>>(x, w, out, n, eps);
  TL_HIP(hipGetLastError());
  return 0;
}

extern "C" int tl_rmsnorm_vector(const float* x_dev, const float* weight_dev, float* out_dev, int n, float eps,
                                 void* stream) {
  TL_REQUIRE(x_dev && weight_dev && out_dev && n > 0, "tl_rmsnorm_vector: bad argument");
  return tl_launch_rmsnorm(x_dev, weight_dev, out_dev, 1, n, eps, 1, static_cast<hipStream_t>(stream));
}

extern "C" int tl_rmsnorm_batch(float* out_dev, const float* in_dev, const float* weight_dev, int num_tokens,
                                int hidden_size, float eps, void* stream) {
  TL_REQUIRE(out_dev && in_dev && weight_dev && hidden_size > 0 && num_tokens >= 0, "tl_rmsnorm_batch: bad argument");
  if (num_tokens == 0) return 0;
  return tl_launch_rmsnorm(in_dev, weight_dev, out_dev, num_tokens, hidden_size, eps, 0,
                           static_cast<hipStream_t>(stream));
}

// ---- RoPE ------------------------------------------------------------------------------------------
extern "C" int tl_rope_table_upload(float* table_dev, int max_pos, int head_dim, float theta, void* stream) {
  TL_REQUIRE(table_dev && max_pos > 0 && head_dim > 0 && head_dim % 2 == 0, "tl_rope_table_upload: bad argument");
  // gpu_initialization.cpp:48-58: host libm, FP32, exactly the reference's expression
  std::vector<float> t((size_t)max_pos * head_dim);
  for (int pos = 0; pos < max_pos; ++pos)
    for (int i = 0; i < head_dim; i += 2) {
      const float freq = powf(theta, -((float)i) / head_dim);
      const float angle = pos * freq;
      t[2 * ((size_t)pos * (head_dim / 2) + i / 2)] = cosf(angle);
      t[2 * ((size_t)pos * (head_dim / 2) + i / 2) + 1] = sinf(angle);
    }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TL_HIP(hipMemcpyAsync(table_dev, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, s));
  TL_HIP(hipStreamSynchronize(s));
  return 0;
}

// rotate pair i of head h of token t (utils.cpp:462-486); one thread per pair
// (HIP's __fmul_rn is a plain multiply that hipcc may still contract into an FMA: contraction is switched off
// for this function so that each product is rounded separately, as in the reference's x86-64 build.)
__device__ __forceinline__ void rope_pair(float* xh, int i, int half, const float* tp, int adjacent) {
#pragma clang fp contract(off)
  const float c = tp[2 * i], s = tp[2 * i + 1];
  const int i0 = adjacent ? 2 * i : i, i1 = adjacent ? 2 * i + 1 : i + half;
  const float a = xh[i0], b = xh[i1];
  xh[i0] = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
  xh[i1] = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
}

// q [T][nq][hd] and (optionally) k [T][nkv][hd] in place, token t at position start_pos+t
__global__ void rope_kernel(float* __restrict__ q, float* __restrict__ k, const float* __restrict__ table, int T,
                            int nq, int nkv, int hd, int start_pos, const int* __restrict__ pos_dev, int max_pos,
                            int adjacent) {
  const int half = hd >> 1;
  const int per_tok = (nq + nkv) * half;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * per_tok) return;
  const int t = gid / per_tok, rem = gid % per_tok;
  const int h = rem / half, i = rem % half;
  const int pos = (pos_dev ? *pos_dev : start_pos) + t;
  if (pos < 0 || pos >= max_pos) return;  // utils.cpp:437-439
  const float* tp = table + 2 * (size_t)pos * half;
  if (h < nq) rope_pair(q + ((size_t)t * nq + h) * hd, i, half, tp, adjacent);
  else rope_pair(k + ((size_t)t * nkv + (h - nq)) * hd, i, half, tp, adjacent);
}

static int launch_rope(float* q, float* k, const float* table, int T, int nq, int nkv, int hd, int start_pos,
                       const int* pos_dev, int max_pos, int adjacent, hipStream_t s) {
  const int n = T * (nq + nkv) * (hd / 2);
  if (n == 0) return 0;
  rope_kernel<<<(n + 255) / 256, 256, 0, s>>>(q, k, table, T, nq, nkv, hd, start_pos, pos_dev, max_pos, adjacent);
  TL_HIP(hipGetLastError());
  return 0;
}

extern "C" int tl_rope(float* vec_dev, int num_heads, int head_dim, const float* freqs_cis_dev, int pos, int max_pos,
                       int use_adjacent_pairing, void* stream) {
  TL_REQUIRE(vec_dev && freqs_cis_dev && num_heads > 0 && head_dim > 0 && head_dim % 2 == 0, "tl_rope: bad argument");
  return launch_rope(vec_dev, nullptr, freqs_cis_dev, 1, num_heads, 0, head_dim, pos, nullptr, max_pos,
                     use_adjacent_pairing, static_cast<hipStream_t>(stream));
}

extern "C" int tl_rope_batch(float* q_batch_dev, float* k_batch_dev, const float* freqs_cis_dev, int num_tokens,
                             int num_q_heads, int num_kv_heads, int head_dim, int start_pos, int max_pos,
                             int use_adjacent_pairing, void* stream) {
  TL_REQUIRE(q_batch_dev && k_batch_dev && freqs_cis_dev && head_dim > 0 && head_dim % 2 == 0 && num_tokens >= 0,
             "tl_rope_batch: bad argument");
  return launch_rope(q_batch_dev, k_batch_dev, freqs_cis_dev, num_tokens, num_q_heads, num_kv_heads, head_dim,
                     start_pos, nullptr, max_pos, use_adjacent_pairing, static_cast<hipStream_t>(stream));
}

// ---- KV cache writes: cache [max_seq][nkv][hd] (model.cpp:622-625, cpu_attention.cpp:54-86) -----------
__global__ void kv_write_kernel(float* __restrict__ cache, const float* __restrict__ src, int T, int kv_dim,
                                int start_pos, const int* __restrict__ pos_dev, int max_seq) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * kv_dim) return;
  const int t = gid / kv_dim, d = gid % kv_dim;
  const int pos = (pos_dev ? *pos_dev : start_pos) + t;
  if (pos < 0 || pos >= max_seq) return;
  cache[(size_t)pos * kv_dim + d] = src[(size_t)t * kv_dim + d];
}

extern "C" int tl_update_kv_cache(float* cache_base_dev, const float* kv_vector_dev, int pos, int kv_head_idx,
                                  int max_seq_len, int num_kv_heads, int head_dim, void* stream) {
  TL_REQUIRE(cache_base_dev && kv_vector_dev, "tl_update_kv_cache: null argument");
  TL_REQUIRE(pos >= 0 && pos < max_seq_len && kv_head_idx >= 0 && kv_head_idx < num_kv_heads && head_dim > 0,
             "tl_update_kv_cache: pos %d / head %d out of range", pos, kv_head_idx);
  // one head vector -> cache[pos][kv_head_idx][:]
  float* dst = cache_base_dev + ((size_t)pos * num_kv_heads + kv_head_idx) * head_dim;
  TL_HIP(hipMemcpyAsync(dst, kv_vector_dev, sizeof(float) * (size_t)head_dim, hipMemcpyDeviceToDevice,
                        static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int tl_update_kv_cache_batch(float* cache_base_dev, const float* kv_batch_dev, int start_pos,
                                        int num_tokens, int num_kv_heads, int head_dim, int cache_max_seq_len,
                                        void* stream) {
  TL_REQUIRE(cache_base_dev && kv_batch_dev && num_tokens >= 0 && start_pos >= 0, "tl_update_kv_cache_batch: bad argument");
  const int kv_dim = num_kv_heads * head_dim, n = num_tokens * kv_dim;
  if (n == 0) return 0;
  kv_write_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(
      cache_base_dev, kv_batch_dev, num_tokens, kv_dim, start_pos, nullptr, cache_max_seq_len);
  TL_HIP(hipGetLastError());
  return 0;
}

// fused decode-step helper for the engine: RoPE(q), RoPE(k), K/V -> cache[pos]
__global__ void rope_kv_kernel(float* __restrict__ q, float* __restrict__ k, const float* __restrict__ v,
                               float* __restrict__ kcache, float* __restrict__ vcache,
                               const float* __restrict__ table, int nq, int nkv, int hd,
                               const int* __restrict__ pos_dev, int max_pos) {
  const int half = hd >> 1;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pos = *pos_dev;
  const int nrot = (nq + nkv) * half;
  const int kv_dim = nkv * hd;
  if (gid < nrot) {
    const int h = gid / half, i = gid % half;
    const bool in_range = pos >= 0 && pos < max_pos;
    const float* tp = table + 2 * (size_t)(in_range ? pos : 0) * half;
    if (h < nq) {
      if (in_range) rope_pair(q + (size_t)h * hd, i, half, tp, 1);
    } else {
      float* kh = k + (size_t)(h - nq) * hd;
      if (in_range) {
        rope_pair(kh, i, half, tp, 1);
        float* dst = kcache + (size_t)pos * kv_dim + (size_t)(h - nq) * hd;
        dst[2 * i] = kh[2 * i];
        dst[2 * i + 1] = kh[2 * i + 1];
      }
    }
  } else if (gid < nrot + kv_dim) {
    const int d = gid - nrot;
    if (pos >= 0 && pos < max_pos) vcache[(size_t)pos * kv_dim + d] = v[d];
  }
}

int tl_launch_rope_kv(float* q, float* k, const float* v, float* kcache, float* vcache, const float* table, int nq,
                      int nkv, int hd, const int* pos_dev, int max_pos, hipStream_t s) {
  const int n = (nq + nkv) * (hd / 2) + nkv * hd;
  rope_kv_kernel<<<(n + 255) / 256, 256, 0, s>>>(q, k, v, kcache, vcache, table, nq, nkv, hd, pos_dev, max_pos);
  TL_HIP(hipGetLastError());
  return 0;
}

// ---- decode attention (model.cpp:632-663): one workgroup per query head --------------------------------
// scores[t] = (q . K[t]) * scale ; softmax with max-subtract and 1/(sum + 1e-9) ; out = sum_t p[t] V[t].
// Scores live in LDS (4 B per cached position).
__global__ __launch_bounds__(256) void attn_decode_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                          const float* __restrict__ vc, float* __restrict__ out,
                                                          int nh, int nkv, int hd, int seq_len,
                                                          const int* __restrict__ pos_dev, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;          // 32 floats
  float* part = sm + 32;    // 4 * hd floats (partial outputs per wave)
  float* sc = sm + 32 + 4 * hd;
  const int h = blockIdx.x;
  const int len = pos_dev ? (*pos_dev + 1) : seq_len;
  const int kvh = h / (nh / nkv);
  const int kv_dim = nkv * hd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qh = q + (size_t)h * hd;
  // hd/64 query elements per lane (hd = 64 or 128; generic loop for other sizes)
  const int per = (hd + 63) >> 6;
  float qr[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) qr[e] = (e < per && lane + 64 * e < hd) ? qh[lane + 64 * e] : 0.f;
  float lmax = -INFINITY;
  for (int t = wave; t < len; t += 4) {
    const float* kt = kc + (size_t)t * kv_dim + (size_t)kvh * hd;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < per && lane + 64 * e < hd) s = fmaf(qr[e], kt[lane + 64 * e], s);
    s = wsum(s) * scale;
    if (lane == 0) sc[t] = s;
    lmax = fmaxf(lmax, s);
  }
  const float mx = block_max(lmax, red);
  float lsum = 0.f;
  for (int t = threadIdx.x; t < len; t += blockDim.x) {
    const float e = expf(sc[t] - mx);
    sc[t] = e;
    lsum += e;
  }
  const float tot = block_sum(lsum, red);
  const float inv = 1.0f / (tot + 1e-9f);  // utils.cpp:692
  // out[d] = sum_t p[t] V[t][d]: wave w takes t = w, w+4, ...; lanes take d
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = wave; t < len; t += 4) {
    const float p = sc[t];
    const float* vt = vc + (size_t)t * kv_dim + (size_t)kvh * hd;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < per && lane + 64 * e < hd) o[e] = fmaf(p, vt[lane + 64 * e], o[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < per && lane + 64 * e < hd) part[wave * hd + lane + 64 * e] = o[e];
  __syncthreads();
  for (int d = threadIdx.x; d < hd; d += blockDim.x)
    out[(size_t)h * hd + d] = (part[d] + part[hd + d] + part[2 * hd + d] + part[3 * hd + d]) * inv;
}

int tl_launch_attn_decode(const float* q, const float* kc, const float* vc, float* out, int nh, int nkv, int hd,
                          int seq_len, const int* pos_dev, int max_seq, float scale, hipStream_t s) {
  TL_REQUIRE(hd > 0 && hd <= 256, "attention: head_dim %d unsupported (max 256)", hd);
  TL_REQUIRE(nkv > 0 && nh % nkv == 0, "attention: %d heads not divisible by %d kv heads", nh, nkv);
  const int cap = pos_dev ? max_seq : seq_len;  // LDS sized for the longest possible sequence under a graph
  const size_t lds = sizeof(float) * (32 + 4 * (size_t)hd + (size_t)cap);
  TL_REQUIRE(lds <= 160 * 1024, "attention: sequence length %d needs %zu B of LDS (> 160 KiB)", cap, lds);
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(attn_decode_kernel, 160 * 1024, lds_raised));
  attn_decode_kernel<<<nh, 256, lds, s>>>(q, kc, vc, out, nh, nkv, hd, seq_len, pos_dev, scale);
  TL_HIP(hipGetLastError());
  return 0;
}

extern "C" int tl_attention(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev, float* out_dev,
                            int num_heads, int current_seq_len, int head_dim, float scale, int cache_max_seq_len,
                            int cache_num_kv_heads, void* stream) {
  TL_REQUIRE(q_dev && k_cache_dev && v_cache_dev && out_dev, "tl_attention: null argument");
  TL_REQUIRE(current_seq_len > 0 && current_seq_len <= cache_max_seq_len, "tl_attention: seq_len %d outside (0,%d]",
             current_seq_len, cache_max_seq_len);
  if (tl_attn_split_supported(num_heads, cache_num_kv_heads, head_dim)) {
    // split partials live in the per-device operator workspace (tl_scratch; not for use under stream capture)
    void* ws = nullptr;
    TL_TRY(tl_scratch(TL_SCRATCH_ATTN, tl_attn_ws_floats(num_heads, head_dim) * sizeof(float), &ws));
    AttnArgs a = {};
    a.q = q_dev;
    a.kc = const_cast<float*>(k_cache_dev);
    a.vc = const_cast<float*>(v_cache_dev);
    a.out = out_dev;
    a.ws = static_cast<float*>(ws);
    a.nh = num_heads;
    a.nkv = cache_num_kv_heads;
    a.hd = head_dim;
    a.seq_len = current_seq_len;
    a.max_seq = cache_max_seq_len;
    a.scale = scale;
    TL_TRY(tl_launch_attn_split(a, false, static_cast<hipStream_t>(stream)));
    return tl_launch_attn_combine(a, static_cast<hipStream_t>(stream));
  }
  return tl_launch_attn_decode(q_dev, k_cache_dev, v_cache_dev, out_dev, num_heads, cache_num_kv_heads, head_dim,
                               current_seq_len, nullptr, cache_max_seq_len, scale, static_cast<hipStream_t>(stream));
}

// ---- tl_attention for callers that cannot allocate ------------------------------------------------------------------
// tl_attention keeps its split partials in a lazily grown hipMalloc workspace (tl_scratch), which rules it out under
// HIP graph capture, in serving loops and in harnesses that forbid allocation. tl_attention_noalloc runs the same two
// kernels over a file-scope device array instead. The array's name is only a host-side stand-in: its device address is
// looked up once per device (tl_ops_init_device, called by tl_init with that device current) and only that pointer is
// ever handed to a kernel.
constexpr int TL_NOALLOC_MAX_HEADS = 64, TL_NOALLOC_MAX_HEAD_DIM = 128, TL_NOALLOC_MAX_DEVICES = 64;
constexpr size_t TL_NOALLOC_FLOATS =
    (size_t)TL_NOALLOC_MAX_HEADS * TL_ATTN_MAX_SPLITS * (TL_NOALLOC_MAX_HEAD_DIM + 4);  // 270,336 floats = 1.03 MiB per GPU
static __device__ __align__(256) float scratch[TL_NOALLOC_FLOATS];
static float* g_noalloc_ptr[TL_NOALLOC_MAX_DEVICES] = {};  // per device: each GPU holds its own copy at its own address

int tl_ops_init_device(int dev) {
  TL_REQUIRE(dev >= 0 && dev < TL_NOALLOC_MAX_DEVICES, "tl_init: device ordinal %d beyond the %d this library tracks", dev,
             TL_NOALLOC_MAX_DEVICES);
  if (g_noalloc_ptr[dev]) return 0;
  float* p = nullptr;
  TL_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
  TL_REQUIRE(p, "tl_init: the static workspace has no address on device %d", dev);
  g_noalloc_ptr[dev] = p;
  return 0;
}

extern "C" int tl_attention_noalloc(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev, float* out_dev,
                                    int num_heads, int current_seq_len, int head_dim, float scale, int cache_max_seq_len,
                                    int cache_num_kv_heads, void* stream) {
  TL_REQUIRE(q_dev && k_cache_dev && v_cache_dev && out_dev, "tl_attention_noalloc: null argument");
  TL_REQUIRE(current_seq_len > 0 && current_seq_len <= cache_max_seq_len, "tl_attention_noalloc: seq_len %d outside (0,%d]",
             current_seq_len, cache_max_seq_len);
  if (!tl_attn_split_supported(num_heads, cache_num_kv_heads, head_dim))  // the single-pass kernel needs no workspace
    return tl_launch_attn_decode(q_dev, k_cache_dev, v_cache_dev, out_dev, num_heads, cache_num_kv_heads, head_dim,
                                 current_seq_len, nullptr, cache_max_seq_len, scale, static_cast<hipStream_t>(stream));
  TL_REQUIRE(tl_attn_ws_floats(num_heads, head_dim) <= TL_NOALLOC_FLOATS,
             "tl_attention_noalloc: %d heads x head_dim %d need %zu floats of partials, the static workspace holds %zu "
             "(at most %d heads of head_dim %d); use tl_attention",
             num_heads, head_dim, tl_attn_ws_floats(num_heads, head_dim), TL_NOALLOC_FLOATS, TL_NOALLOC_MAX_HEADS,
             TL_NOALLOC_MAX_HEAD_DIM);
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  TL_REQUIRE(dev >= 0 && dev < TL_NOALLOC_MAX_DEVICES && g_noalloc_ptr[dev],
             "tl_attention_noalloc: call tl_init(%d) first (it looks up this device's static workspace)", dev);
  // every split below the active count writes its whole record before the combine reads it, and the combine reads no
  // other record: nothing left by an earlier call is ever read, so nothing needs clearing
  AttnArgs a = {};
  a.q = q_dev;
  a.kc = const_cast<float*>(k_cache_dev);
  a.vc = const_cast<float*>(v_cache_dev);
  a.out = out_dev;
  a.ws = g_noalloc_ptr[dev];
  a.nh = num_heads;
  a.nkv = cache_num_kv_heads;
  a.hd = head_dim;
  a.seq_len = current_seq_len;
  a.max_seq = cache_max_seq_len;
  a.scale = scale;
  TL_TRY(tl_launch_attn_split(a, false, static_cast<hipStream_t>(stream)));
  return tl_launch_attn_combine(a, static_cast<hipStream_t>(stream));
}

// ---- prefill attention (cpu_attention.cpp:91-201): one workgroup per (token, head), causal via the cache ---
__global__ __launch_bounds__(256) void attn_prefill_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                           const float* __restrict__ vc, float* __restrict__ out,
                                                           int T, int start_pos, int nh, int nkv, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;
  float* part = sm + 32;
  float* sc = sm + 32 + 4 * hd;
  const int t_tok = blockIdx.x / nh, h = blockIdx.x % nh;
  const int len = start_pos + t_tok + 1;
  const int kvh = h / (nh / nkv);
  const int kv_dim = nkv * hd;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qh = q + ((size_t)t_tok * nh + h) * hd;
  const int per = (hd + 63) >> 6;
  float qr[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) qr[e] = (e < per && lane + 64 * e < hd) ? qh[lane + 64 * e] : 0.f;
  float lmax = -INFINITY;
  for (int t = wave; t < len; t += 4) {
    const float* kt = kc + (size_t)t * kv_dim + (size_t)kvh * hd;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < per && lane + 64 * e < hd) s = fmaf(qr[e], kt[lane + 64 * e], s);
    s = wsum(s) * scale;
    if (lane == 0) sc[t] = s;
    lmax = fmaxf(lmax, s);
  }
  const float mx = block_max(lmax, red);
  float lsum = 0.f;
  for (int t = threadIdx.x; t < len; t += blockDim.x) {
    const float e = expf(sc[t] - mx);
    sc[t] = e;
    lsum += e;
  }
  const float tot = block_sum(lsum, red);
  const float inv = 1.0f / (tot + 1e-9f);
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = wave; t < len; t += 4) {
    const float p = sc[t];
    const float* vt = vc + (size_t)t * kv_dim + (size_t)kvh * hd;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < per && lane + 64 * e < hd) o[e] = fmaf(p, vt[lane + 64 * e], o[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < per && lane + 64 * e < hd) part[wave * hd + lane + 64 * e] = o[e];
  __syncthreads();
  for (int d = threadIdx.x; d < hd; d += blockDim.x)
    out[((size_t)t_tok * nh + h) * hd + d] = (part[d] + part[hd + d] + part[2 * hd + d] + part[3 * hd + d]) * inv;
}

extern "C" int tl_attention_batch_prefill(const float* q_batch_dev, const float* k_cache_dev,
                                          const float* v_cache_dev, float* out_batch_dev, int num_tokens,
                                          int start_pos, int cache_max_seq_len, int num_q_heads, int num_kv_heads,
                                          int head_dim, float scale, void* stream) {
  TL_REQUIRE(q_batch_dev && k_cache_dev && v_cache_dev && out_batch_dev, "tl_attention_batch_prefill: null argument");
  TL_REQUIRE(num_tokens >= 0 && start_pos >= 0 && start_pos + num_tokens <= cache_max_seq_len,
             "tl_attention_batch_prefill: positions [%d,%d) exceed the cache (%d)", start_pos, start_pos + num_tokens,
             cache_max_seq_len);
  TL_REQUIRE(head_dim > 0 && head_dim <= 256 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0,
             "tl_attention_batch_prefill: bad head configuration");
  if (num_tokens == 0) return 0;
  if (tl_attn_prefill_mfma_supported(num_q_heads, num_kv_heads, head_dim))  // tiled, any context length
    return tl_launch_attn_prefill_mfma(q_batch_dev, k_cache_dev, v_cache_dev, out_batch_dev, num_tokens, start_pos,
                                       num_q_heads, num_kv_heads, head_dim, scale, static_cast<hipStream_t>(stream));
  const size_t lds = sizeof(float) * (32 + 4 * (size_t)head_dim + (size_t)(start_pos + num_tokens));
  TL_REQUIRE(lds <= 160 * 1024, "tl_attention_batch_prefill: sequence too long for LDS scores (%zu B)", lds);
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(attn_prefill_kernel, 160 * 1024, lds_raised));
  attn_prefill_kernel<<<num_tokens * num_q_heads, 256, lds, static_cast<hipStream_t>(stream)>>>(
      q_batch_dev, k_cache_dev, v_cache_dev, out_batch_dev, num_tokens, start_pos, num_q_heads, num_kv_heads,
      head_dim, scale);
  TL_HIP(hipGetLastError());
  return 0;
}

// ---- element-wise ------------------------------------------------------------------------------------
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
extern "C" int tl_add_residual(const float* a_dev, const float* b_dev, float* out_dev, size_t n, void* stream) {
  TL_REQUIRE(a_dev && b_dev && out_dev, "tl_add_residual: null argument");
  if (n == 0) return 0;
  add_kernel<<<(unsigned)((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(a_dev, b_dev, out_dev, n);
  TL_HIP(hipGetLastError());
  return 0;
}

__global__ void swiglu_kernel(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ o, size_t n,
                              int batch_form) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = g[i];
  if (batch_form) o[i] = (x / (1.0f + expf(-x))) * u[i];          // cpu_batch_processor.cpp:252-256
  else o[i] = (x * (1.0f / (1.0f + expf(-x)))) * u[i];            // utils.cpp:704-705, model.cpp:735
}
extern "C" int tl_swiglu(const float* gate_dev, const float* up_dev, float* out_dev, size_t n, int batch_form,
                         void* stream) {
  TL_REQUIRE(gate_dev && up_dev && out_dev, "tl_swiglu: null argument");
  if (n == 0) return 0;
  swiglu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(gate_dev, up_dev, out_dev,
                                                                                           n, batch_form);
  TL_HIP(hipGetLastError());
  return 0;
}

// ---- sampling: repetition penalty (api.cpp:86-97) + first-max arg-max (api.cpp:79,135) ------------------
// One workgroup. hist: ring of generated ids; penalty applied in place to logits for the last <=64 ids.
__global__ __launch_bounds__(1024) void argmax_kernel(float* __restrict__ logits, int n, int* __restrict__ result,
                                                      const int* __restrict__ hist, const int* __restrict__ n_hist_dev,
                                                      float penalty) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  if (penalty > 1.0f && hist && n_hist_dev) {
    if (threadIdx.x == 0) {  // sequential on purpose: a repeated id is penalised once per occurrence, in order
      const int nh = *n_hist_dev;
      const int start = nh > 64 ? nh - 64 : 0;
      for (int i = start; i < nh; ++i) {
        const int id = hist[i];
        if (id < 0 || id >= n) continue;
        const float l = logits[id];
        logits[id] = (l > 0.0f) ? l / penalty : l * penalty;
      }
    }
    __syncthreads();
  }
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = logits[i];
    if (v > best) {  // strictly greater keeps the first maximum within a thread's stride
      best = v;
      idx = i;
    }
  }
  // combine: larger value wins, ties -> smaller index
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(idx, o);
    if (ov > best || (ov == best && oi < idx)) {
      best = ov;
      idx = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    bv[threadIdx.x >> 6] = best;
    bi[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
        best = bv[w];
        idx = bi[w];
      }
    *result = idx == 0x7fffffff ? 0 : idx;  // all -inf/NaN: std::max_element returns the first element
  }
}

int tl_launch_argmax(float* logits, int n, int* result_dev, const int* hist, const int* n_hist_dev, float penalty,
                     hipStream_t s) {
  argmax_kernel<<<1, 1024, 0, s>>>(logits, n, result_dev, hist, n_hist_dev, penalty);
  TL_HIP(hipGetLastError());
  return 0;
}

extern "C" int tl_argmax(const float* logits_dev, int n, int* result_dev, void* stream) {
  TL_REQUIRE(logits_dev && result_dev && n > 0, "tl_argmax: bad argument");
  return tl_launch_argmax(const_cast<float*>(logits_dev), n, result_dev, nullptr, nullptr, 1.0f,
                          static_cast<hipStream_t>(stream));
}
