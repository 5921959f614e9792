// tl_sample.hip — temperature / top-k / top-p sampling on the device, the non-greedy half of
// sample_top_k_top_p_temperature (api.cpp:126-216) + apply_repetition_penalty (api.cpp:86-97).
//
// The reference copies vocab * 4 B to the host every token and sorts the whole vocabulary there (std::sort over
// 128 k pairs, ~10 ms: six decode steps' worth). Here the logits never leave HBM: one workgroup
//   1. applies the repetition penalty in place (in order, once per occurrence),
//   2. softmax of (logit - max) / T with the exponentials and their sum in FP64, each probability rounded to FP32
//      (api.cpp:143-175) into a scratch vector,
//   3. selects the top_k largest (probability, id) pairs — descending by probability, ties by HIGHER id, as
//      std::greater<std::pair<float,int>> orders them — with a radix select over the 64-bit key (prob bits << 32 | id),
//   4. sorts those <= 1024 candidates in LDS (bitonic), walks the FP32 cumulative sum up to top_p, renormalises,
//   5. draws with the inverse CDF in FP64 the way std::discrete_distribution does (normalised partial sums, first
//      entry whose partial sum exceeds u), writes the id and advances the step state.
// The uniform variate comes from a counter-based generator (splitmix64 of seed + step); the reference seeds a
// std::mt19937 from std::random_device, so its streams are not reproducible either: parity is on the candidate set
// and its probabilities (tests/test_sample_gpu.py), not on the random stream.
// Limit: effective top_k <= 1024 (top_k <= 0 or >= vocab means "no top-k" in the reference; that form stays on the
// host path, tl_session.cpp).
#include <math.h>

#include "tl_internal.h"

namespace {

constexpr int SK_MAX = TL_SAMPLE_MAX_TOP_K;
constexpr int SK_THREADS = 1024;
typedef unsigned long long u64;

__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ float block_max_f(float v, float* red) {
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < SK_THREADS / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < SK_THREADS / 64; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ u64 key_of(float p, int id) { return ((u64)__float_as_uint(p) << 32) | (unsigned)id; }

// logits [n] (penalised in place when state != NULL), probs: scratch [n]. state: the engine's step state
// {token, pos, n_hist, penalty, history...} or NULL for the operator-level call.
__global__ __launch_bounds__(SK_THREADS) void sample_topk_kernel(float* __restrict__ logits, int n,
                                                                 float* __restrict__ probs, int* __restrict__ state,
                                                                 int state_header, int hist_cap, TlSampleParams* sp,
                                                                 int* __restrict__ out_id, int* __restrict__ cand_ids,
                                                                 float* __restrict__ cand_probs,
                                                                 int* __restrict__ n_cand) {
  __shared__ float redf[16];
  __shared__ double redd[16];
  __shared__ unsigned hist[256];
  __shared__ u64 keys[SK_MAX];
  __shared__ u64 s_prefix;
  __shared__ int s_remaining, s_done, s_count, s_pick;
  const int tid = threadIdx.x;
  const float temperature = sp->temperature, top_p = sp->top_p;
  // ---- 1. repetition penalty (sequential on purpose: a repeated id is penalised once per occurrence, in order) ----
  if (state && tid == 0) {
    const float penalty = __int_as_float(state[3]);
    const int n_hist = state[2];
    const int* h = state + state_header;
    if (penalty > 1.0f) {
      for (int i = n_hist > 64 ? n_hist - 64 : 0; i < n_hist; ++i) {
        const int id = h[i];
        if (id < 0 || id >= n) continue;
        const float l = logits[id];
        logits[id] = (l > 0.0f) ? l / penalty : l * penalty;
      }
    }
  }
  __syncthreads();
  // ---- 2. max, FP64 softmax, FP32 probabilities ----------------------------------------------------------------------
  float mx = -INFINITY;
  for (int i = tid; i < n; i += SK_THREADS) mx = fmaxf(mx, logits[i]);
  mx = block_max_f(mx, redf);
  int pick = -1;
  if (temperature < 0.05f) {  // greedy: first maximum (api.cpp:134-136)
    int idx = 0x7fffffff;
    for (int i = tid; i < n; i += SK_THREADS)
      if (logits[i] == mx && i < idx) idx = i;
    for (int o = 32; o >= 1; o >>= 1) idx = min(idx, __shfl_xor(idx, o));
    __shared__ int redi[16];
    if ((tid & 63) == 0) redi[tid >> 6] = idx;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < SK_THREADS / 64; ++w) idx = min(idx, redi[w]);
      s_pick = idx == 0x7fffffff ? 0 : idx;
      if (n_cand) *n_cand = 0;
    }
    __syncthreads();
    pick = s_pick;
  } else {
    const float inv_t = 1.0f / temperature;
    double part = 0.0;
    for (int i = tid; i < n; i += SK_THREADS) part += exp((double)((logits[i] - mx) * inv_t));
    const double total = block_sum_d(part, redd);
    for (int i = tid; i < n; i += SK_THREADS) {
      const double e = exp((double)((logits[i] - mx) * inv_t));
      probs[i] = (float)(total > 0.0 ? e / total : 1.0 / n);
    }
    __syncthreads();
    // ---- 3. radix select of the k largest keys (prob bits, id), most significant byte first -----------------------
    int k = sp->top_k;
    if (k <= 0 || k > n) k = n;
    if (k > SK_MAX) k = SK_MAX;  // (the host entry points reject this case)
    if (tid == 0) {
      s_prefix = 0;
      s_remaining = k;
      s_done = 0;
    }
    __syncthreads();
    int nbytes = 0;  // bytes of the key fixed so far (from the top)
    for (int byte = 7; byte >= 0; --byte) {
      if (s_done) break;
      for (int b = tid; b < 256; b += SK_THREADS) hist[b] = 0;
      __syncthreads();
      const u64 prefix = s_prefix;
      const int sh = 8 * byte;
      for (int i = tid; i < n; i += SK_THREADS) {
        const u64 key = key_of(probs[i], i);
        if (byte == 7 || (key >> (sh + 8)) == (prefix >> (sh + 8))) atomicAdd(&hist[(unsigned)(key >> sh) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int rem = s_remaining, b = 255;
        unsigned matching = 0;
        for (int q = 0; q < 256; ++q) matching += hist[q];
        if ((int)matching == rem) {  // everything that shares the prefix so far is selected
          s_done = 1;
        } else {
          while (b > 0 && (int)hist[b] < rem) rem -= hist[b--];
          s_prefix = prefix | ((u64)b << sh);
          s_remaining = rem;
        }
      }
      __syncthreads();
      if (!s_done) nbytes = 8 - byte;
    }
    // ---- gather the selected keys into LDS ----------------------------------------------------------------------
    if (tid == 0) s_count = 0;
    for (int i = tid; i < SK_MAX; i += SK_THREADS) keys[i] = 0;  // 0 = empty slot (real keys are stored + 1)
    __syncthreads();
    {
      const int sh = 64 - 8 * nbytes;  // bits below the fixed part
      const u64 prefix = s_prefix;
      for (int i = tid; i < n; i += SK_THREADS) {
        const u64 key = key_of(probs[i], i);
        // above the prefix on the fixed bytes: selected; equal: selected (either everything that matches was taken,
        // or all 8 bytes are fixed and the key is the k-th one itself)
        const bool sel = nbytes == 0 ? true : (key >> sh) >= (prefix >> sh);
        if (sel) {
          const int slot = atomicAdd(&s_count, 1);
          if (slot < SK_MAX) keys[slot] = key + 1;
        }
      }
    }
    __syncthreads();
    const int nsel = s_count < SK_MAX ? s_count : SK_MAX;
    // ---- 4. bitonic sort, descending ---------------------------------------------------------------------------------
    for (int size = 2; size <= SK_MAX; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < SK_MAX; i += SK_THREADS) {
          const int j = i ^ stride;
          if (j > i) {
            const bool desc = (i & size) == 0;
            const u64 a = keys[i], b = keys[j];
            if (desc ? a < b : a > b) {
              keys[i] = b;
              keys[j] = a;
            }
          }
        }
        __syncthreads();
      }
    // ---- 5. top-p cut, renormalisation, draw (one thread: <= 1024 sequential FP32 adds, as the reference does) ---------
    if (tid == 0) {
      float cum = 0.f;
      int keep = nsel;
      for (int i = 0; i < nsel; ++i) {
        cum += __uint_as_float((unsigned)((keys[i] - 1) >> 32));
        if (cum >= top_p) {
          keep = i + 1;
          break;
        }
      }
      float kept = 0.f;
      for (int i = 0; i < keep; ++i) kept += __uint_as_float((unsigned)((keys[i] - 1) >> 32));
      double u;
      if (sp->use_fixed_u) {
        u = (double)sp->fixed_u;
      } else {
        const u64 r = splitmix64(sp->seed + 0xD1B54A32D192ED03ull * (u64)sp->counter);
        u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
        sp->counter = sp->counter + 1;
      }
      // std::discrete_distribution: weights normalised in FP64, pick the first index whose partial sum exceeds u
      double wsum = 0.0;
      for (int i = 0; i < keep; ++i) {
        const float p = __uint_as_float((unsigned)((keys[i] - 1) >> 32));
        wsum += (double)(kept > 0.f ? p / kept : 1.0f / (float)keep);
      }
      double acc = 0.0;
      int chosen = keep - 1;
      for (int i = 0; i < keep; ++i) {
        const float p = __uint_as_float((unsigned)((keys[i] - 1) >> 32));
        const float fp = kept > 0.f ? p / kept : 1.0f / (float)keep;
        acc += (double)fp / wsum;
        if (cand_ids) cand_ids[i] = (int)((keys[i] - 1) & 0xffffffffu);
        if (cand_probs) cand_probs[i] = fp;
        if (u < acc && chosen == keep - 1 && i < keep - 1) chosen = i;
      }
      if (n_cand) *n_cand = keep;
      s_pick = keep > 0 ? (int)((keys[chosen] - 1) & 0xffffffffu) : 0;
    }
    __syncthreads();
    pick = s_pick;
  }
  if (tid == 0) {
    if (out_id) *out_id = pick;
    if (state) {  // advance the step state exactly as the greedy tail does
      const int n_hist = state[2];
      state[0] = pick;
      state[1] = state[1] + 1;
      if (n_hist < hist_cap) state[state_header + n_hist] = pick;
      state[2] = n_hist + 1;
    }
  }
}

}  // namespace

int tl_launch_sample_topk(float* logits, int n, float* probs_scratch, int* state, int state_header, int hist_cap,
                          TlSampleParams* params_dev, int* out_id, int* cand_ids, float* cand_probs, int* n_cand,
                          hipStream_t s) {
  sample_topk_kernel<<<1, SK_THREADS, 0, s>>>(logits, n, probs_scratch, state, state_header, hist_cap, params_dev, out_id,
                                              cand_ids, cand_probs, n_cand);
  TL_HIP(hipGetLastError());
  return 0;
}

// operator-level call: logits are NOT modified (no penalty); u in [0,1) is the caller's uniform variate
extern "C" int tl_sample_top_k_top_p(const float* logits_dev, int n, float temperature, int top_k, float top_p, float u,
                                     int* id_dev, int* cand_ids_dev, float* cand_probs_dev, int* n_cand_dev,
                                     void* stream) {
  TL_REQUIRE(logits_dev && id_dev && n > 0, "tl_sample_top_k_top_p: bad argument");
  TL_REQUIRE(u >= 0.0f && u < 1.0f, "tl_sample_top_k_top_p: u = %g outside [0,1)", (double)u);
  const int k = (top_k <= 0 || top_k > n) ? n : top_k;
  TL_REQUIRE(temperature < 0.05f || k <= SK_MAX,
             "tl_sample_top_k_top_p: effective top_k %d exceeds the device sampler's limit %d (use the host path)", k, SK_MAX);
  hipStream_t s = static_cast<hipStream_t>(stream);
  void *scratch_v = nullptr, *params_v = nullptr;  // per-device operator workspace (tl_scratch)
  TL_TRY(tl_scratch(TL_SCRATCH_SAMPLE_PROBS, sizeof(float) * (size_t)n, &scratch_v));
  TL_TRY(tl_scratch(TL_SCRATCH_SAMPLE_PARAMS, sizeof(TlSampleParams), &params_v));
  float* scratch = static_cast<float*>(scratch_v);
  TlSampleParams* params = static_cast<TlSampleParams*>(params_v);
  TlSampleParams h = {};
  h.temperature = temperature;
  h.top_k = top_k;
  h.top_p = top_p;
  h.use_fixed_u = 1;
  h.fixed_u = u;
  TL_HIP(hipMemcpyAsync(params, &h, sizeof(h), hipMemcpyHostToDevice, s));
  TL_HIP(hipStreamSynchronize(s));  // h lives on this stack frame
  return tl_launch_sample_topk(const_cast<float*>(logits_dev), n, scratch, nullptr, 0, 0, params, id_dev, cand_ids_dev,
                               cand_probs_dev, n_cand_dev, s);
}
