// tl_attn.hip — decode attention over the device-resident KV cache, split over the sequence ("flash-decoding"),
// with RoPE of q/k and the KV-cache write of the new position fused in for the engine's decode step.
//
// Replaces attention_cuda / attention_cuda_optimized (cuda_kernels.h:259,607: one block per head, scores for the
// whole context in shared memory) + rope_cuda (:239) + update_kv_cache_cuda (:344, one launch per kv head per K/V per
// layer). The math is the CPU path's (model.cpp:609-663, utils.cpp:462-486, 675-698): adjacent-pair RoPE from the
// host-computed table, scores = (q . K[t]) / sqrt(hd), softmax with max-subtraction and 1/(sum + 1e-9), out = sum p V.
//
// Shape (HBM/latency-bound byte work, no MFMA):
//   - grid (kv head, sequence split, head-group part): one workgroup serves the query heads of a kv head (GQA group), so
//     every K/V byte is read once; groups of 8 heads (TinyLlama, the 70B shape) are served by two workgroups of 4 (their
//     K/V rows are read twice, the per-row softmax work per workgroup halves: +11 % tokens/s on TinyLlama-1.1B); a split is max(32, pow2(len / 32)) positions (tl_attn_split_len): at most 32 splits, so that
//     8 kv heads fill 256 CUs from 1024 positions on while short contexts still spread over tens of CUs (one CU only
//     pulls ~25-50 GB/s);
//   - a K/V row is read by HD/4 lanes x 16 B (two or four rows per wave load); the K AND V rows of a batch (up to 16 of
//     each per lane) are in flight together and folded into a running (max, sum, out) per lane group (online softmax),
//     so a split of 256 positions costs one memory round trip;
//   - every split writes a (max, sum, unnormalised out) partial per head; the partials are combined at the next kernel
//     boundary: in the x-staging prologue of the attn_output GEMV (tl_gemv.hip), or by attn_combine_kernel for the
//     operator-level API. (An in-launch combine by the last-arriving workgroup costs a release fence, an atomic and an
//     acquire fence, ~4 us, more than the attention itself at short context.)
//   - FUSE: the workgroup rotates its query heads into LDS; the split that owns the new position rotates k, writes
//     K/V of that position into the cache and takes that position's score/value from LDS instead of from memory.
#include <math.h>

#include <type_traits>

#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ATT_NW = 8;  // waves per workgroup (512 threads: 256 VGPRs per lane for the rows in flight)
constexpr int ATT_THREADS = ATT_NW * 64;

__device__ __forceinline__ float safe_exp_diff(float m, float m_new) {  // exp(m - m_new), 0 when m is -inf
  return m == -INFINITY ? 0.f : expf(m - m_new);
}

// adjacent-pair rotation (utils.cpp:462-486); contraction off: each product is rounded separately as on the host
__device__ __forceinline__ void rope2(float a, float b, float c, float s, float& o0, float& o1) {
#pragma clang fp contract(off)
  o0 = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
  o1 = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
}

template <int HD, int G, bool FUSE>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_split_kernel(AttnArgs a) {
  constexpr int LP = HD / 4;             // lanes per K/V row
  constexpr int PP = 64 / LP;            // rows per wave load
  constexpr int U = 16;                  // K and V rows in flight per lane: 256 rows per batch and workgroup at HD 128
  constexpr int RECL = HD + 4;           // partial record: out[HD], max, sum, 2 pad (16-B aligned)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm;                        // [G][HD]
  float* kr = qs + G * HD;               // [HD] rotated k of the new position (FUSE owner)
  float* vr = kr + HD;                   // [HD]
  float* red = vr + HD;                  // 64 floats: M[g] at g, 1/L or L at 8 + g, flag at 32
  float* part = red + 64;                // [ATT_NW][G][RECL]
  const int kvh = blockIdx.x, split = blockIdx.y;
  const int head0 = (kvh * gridDim.z + blockIdx.z) * G;  // gridDim.z workgroups share a kv head, G query heads each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = a.pos_dev ? *a.pos_dev : a.seq_len - 1;
  const int len = pos + 1;
  const int SL = tl_attn_split_len(len);
  const int active = (len + SL - 1) / SL;
  if (split >= active || pos < 0 || pos >= a.max_seq) return;  // (the combine only reads splits < active)
  const int kv_dim = a.nkv * HD;
  const int t0 = split * SL;
  const int t1 = len < t0 + SL ? len : t0 + SL;
  const bool owner = FUSE && split == active - 1;  // this split holds the new position, still only in registers
  const int ncache = owner ? t1 - t0 - 1 : t1 - t0;  // positions of this split that come from the cache

  // ---- phase 0: query heads (and the new k / v) into LDS ---------------------------------------------------
  constexpr int half = HD / 2;
  if (FUSE) {
    const float* tp = a.table + 2 * (size_t)pos * half;
    for (int i = tid; i < G * half; i += ATT_THREADS) {
      const int g = i / half, p = i % half;
      const float* qh = a.q + (size_t)(head0 + g) * HD;
      rope2(qh[2 * p], qh[2 * p + 1], tp[2 * p], tp[2 * p + 1], qs[g * HD + 2 * p], qs[g * HD + 2 * p + 1]);
    }
    if (owner) {
      for (int p = tid; p < half; p += ATT_THREADS) {
        const float* kh = a.k + (size_t)kvh * HD;
        float o0, o1;
        rope2(kh[2 * p], kh[2 * p + 1], tp[2 * p], tp[2 * p + 1], o0, o1);
        kr[2 * p] = o0;
        kr[2 * p + 1] = o1;
        if (blockIdx.z == 0) {  // one of the workgroups that share the kv head writes the cache
          float* dst = a.kc + (size_t)pos * kv_dim + (size_t)kvh * HD;
          dst[2 * p] = o0;
          dst[2 * p + 1] = o1;
        }
      }
      for (int d = tid; d < HD; d += ATT_THREADS) {
        const float x = a.v[(size_t)kvh * HD + d];
        vr[d] = x;
        if (blockIdx.z == 0) a.vc[(size_t)pos * kv_dim + (size_t)kvh * HD + d] = x;
      }
    }
  } else {
    for (int i = tid; i < G * HD; i += ATT_THREADS) qs[i] = a.q[(size_t)head0 * HD + i];
  }
  __syncthreads();

  // ---- phase 1: one pass over the rows with a running (max, sum, out) per lane group (online softmax) ----------
  const int sub = lane / LP, ln = lane % LP;
  f32x4 qf[G], acc[G];
  float m[G], l[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    qf[g] = *reinterpret_cast<const f32x4*>(qs + g * HD + 4 * ln);
    acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[g] = -INFINITY;
    l[g] = 0.f;
  }
  // rows [r0, r0 + cnt) of `kk`/`vv` hold valid[u]; folds them into the running state
  auto fold = [&](const f32x4* kk, const f32x4* vv, const bool* valid, auto cnt_c) {
    constexpr int cnt = decltype(cnt_c)::value;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s[cnt];
      float mx = m[g];
#pragma unroll
      for (int u = 0; u < cnt; ++u) {
        float d = (qf[g][0] * kk[u][0] + qf[g][1] * kk[u][1]) + (qf[g][2] * kk[u][2] + qf[g][3] * kk[u][3]);
#pragma unroll
        for (int o = LP / 2; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        s[u] = valid[u] ? d * a.scale : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      if (mx == -INFINITY) continue;  // nothing valid yet (uniform over the lane group)
      const float corr = safe_exp_diff(m[g], mx);
      m[g] = mx;
      l[g] *= corr;
      acc[g][0] *= corr; acc[g][1] *= corr; acc[g][2] *= corr; acc[g][3] *= corr;
#pragma unroll
      for (int u = 0; u < cnt; ++u) {
        const float p = valid[u] ? expf(s[u] - mx) : 0.f;
        l[g] += p;
        acc[g][0] = fmaf(p, vv[u][0], acc[g][0]);
        acc[g][1] = fmaf(p, vv[u][1], acc[g][1]);
        acc[g][2] = fmaf(p, vv[u][2], acc[g][2]);
        acc[g][3] = fmaf(p, vv[u][3], acc[g][3]);
      }
    }
  };
  const float* kbase = a.kc + (size_t)t0 * kv_dim + (size_t)kvh * HD + 4 * ln;
  const float* vbase = a.vc + (size_t)t0 * kv_dim + (size_t)kvh * HD + 4 * ln;
  // one batch of UC row slots per lane group: slot u holds row base + (u * NW + wave) * PP + sub (rows go round the
  // waves first, so short splits use few slots in every wave). UC is compile-time: 2, 4, 8 or U.
  auto batch = [&](int base, auto uc_c) {
    constexpr int UC = decltype(uc_c)::value;
    f32x4 kk[UC], vv[UC];
    bool valid[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {  // unconditional (clamped) loads: all K and V rows of the batch in flight together
      const int idx = base + (u * ATT_NW + wave) * PP + sub;
      valid[u] = idx < ncache;
      const size_t off = (size_t)(valid[u] ? idx : ncache - 1) * kv_dim;
      kk[u] = *reinterpret_cast<const f32x4*>(kbase + off);
      vv[u] = *reinterpret_cast<const f32x4*>(vbase + off);
    }
    fold(kk, vv, valid, uc_c);
  };
  for (int base = 0; base < ncache; base += ATT_NW * PP * U) {
    const int mine = ncache - base - wave * PP;  // rows at or after this wave's first slot (wave-uniform)
    if (mine <= 0) continue;
    const int slots = (mine + ATT_NW * PP - 1) / (ATT_NW * PP);
    if (slots <= 2) batch(base, std::integral_constant<int, 2>{});
    else if (slots <= 4) batch(base, std::integral_constant<int, 4>{});
    else if (U > 8 && slots <= 8) batch(base, std::integral_constant<int, 8>{});
    else batch(base, std::integral_constant<int, U>{});
  }
  if (owner && wave == ATT_NW - 1) {  // the new position, from LDS (the last wave has the fewest cache rows)
    f32x4 kk[1], vv[1];
    bool valid[1];
    kk[0] = *reinterpret_cast<const f32x4*>(kr + 4 * ln);
    vv[0] = *reinterpret_cast<const f32x4*>(vr + 4 * ln);
    valid[0] = sub == 0;
    fold(kk, vv, valid, std::integral_constant<int, 1>{});
  }
  // fold the PP lane groups of a wave (shuffles), then the waves (LDS)
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int o = 32; o >= LP; o >>= 1) {
      const float mo = __shfl_xor(m[g], o);
      const float mn = fmaxf(m[g], mo);
      const float c = safe_exp_diff(m[g], mn), co = safe_exp_diff(mo, mn);
      l[g] = l[g] * c + __shfl_xor(l[g], o) * co;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[g][e] = acc[g][e] * c + __shfl_xor(acc[g][e], o) * co;
      m[g] = mn;
    }
    if (sub == 0) {
      float* rec = part + (size_t)(wave * G + g) * RECL;
      *reinterpret_cast<f32x4*>(rec + 4 * ln) = acc[g];
      if (ln == 0) {
        rec[HD] = m[g];
        rec[HD + 1] = l[g];
      }
    }
  }
  __syncthreads();
  if (tid < G) {  // split-wide max and sum per head
    float M = -INFINITY;
    for (int w = 0; w < ATT_NW; ++w) M = fmaxf(M, part[(size_t)(w * G + tid) * RECL + HD]);
    float L = 0.f;
    for (int w = 0; w < ATT_NW; ++w) {
      const float* rec = part + (size_t)(w * G + tid) * RECL;
      L += rec[HD + 1] * safe_exp_diff(rec[HD], M);
    }
    red[tid] = M;
    red[8 + tid] = L;
  }
  __syncthreads();

  // ---- phase 2: the split's partial per head: ws[head][split] = {out[HD] (unnormalised), max, sum} -----------------
  for (int i = tid; i < G * HD; i += ATT_THREADS) {
    const int g = i / HD, d = i % HD;
    const float M = red[g];
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_NW; ++w) {
      const float* rec = part + (size_t)(w * G + g) * RECL;
      o += rec[d] * safe_exp_diff(rec[HD], M);
    }
    float* rec = a.ws + ((size_t)(head0 + g) * TL_ATTN_MAX_SPLITS + split) * RECL;
    rec[d] = o;
    if (d == 0) {
      rec[HD] = M;
      rec[HD + 1] = red[8 + g];
    }
  }
}

// out[h][:] = sum_s o_s e^(m_s - M) / (sum_s l_s e^(m_s - M) + 1e-9): one workgroup per head (operator-level API).
// The (max, sum) pairs of all splits are loaded in one go and the partial rows eight at a time (unrolled, clamped:
// independent loads instead of one dependent round trip per split); the sums run in split order.
__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ ws, float* __restrict__ out, int hd,
                                                           int seq_len, const int* __restrict__ pos_dev) {
  constexpr int NS = TL_ATTN_MAX_SPLITS;
  const int len = pos_dev ? *pos_dev + 1 : seq_len;
  const int SL = tl_attn_split_len(len);
  const int active = (len + SL - 1) / SL;
  const int h = blockIdx.x, d = threadIdx.x;
  if (d >= hd) return;
  const int recl = hd + 4;
  const float* rec0 = ws + (size_t)h * NS * recl;
  float ms[NS], ls[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float* rec = rec0 + (size_t)(s < active ? s : active - 1) * recl;
    ms[s] = rec[hd];
    ls[s] = rec[hd + 1];
  }
  float M = -INFINITY;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < active) M = fmaxf(M, ms[s]);
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < NS; s0 += 8) {
    if (s0 >= active) break;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rec0[(size_t)(s0 + j < active ? s0 + j : active - 1) * recl + d];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < active) {
        const float w = expf(ms[s0 + j] - M);
        L = fmaf(ls[s0 + j], w, L);
        O = fmaf(o[j], w, O);
      }
  }
  out[(size_t)h * hd + d] = O * (1.0f / (L + 1e-9f));  // utils.cpp:692
}

template <int HD, int G>
int launch_hg(const AttnArgs& a, bool fuse, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)G * HD + 2 * HD + 64 + (size_t)ATT_NW * G * (HD + 4));
  const dim3 grid(a.nkv, TL_ATTN_MAX_SPLITS, (a.nh / a.nkv) / G);
  if (fuse) {
    static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
    TL_TRY(tl_raise_lds_limit(attn_decode_split_kernel<HD, G, true>, 160 * 1024, lds_raised));
    attn_decode_split_kernel<HD, G, true><<<grid, ATT_THREADS, lds, s>>>(a);
  } else {
    static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
    TL_TRY(tl_raise_lds_limit(attn_decode_split_kernel<HD, G, false>, 160 * 1024, lds_raised));
    attn_decode_split_kernel<HD, G, false><<<grid, ATT_THREADS, lds, s>>>(a);
  }
  TL_HIP(hipGetLastError());
  return 0;
}

template <int HD>
int launch_h(const AttnArgs& a, int G, bool fuse, hipStream_t s) {
  switch (G) {
    case 1: return launch_hg<HD, 1>(a, fuse, s);
    case 2: return launch_hg<HD, 2>(a, fuse, s);
    case 3: return launch_hg<HD, 3>(a, fuse, s);
    case 4: return launch_hg<HD, 4>(a, fuse, s);
    // 8 query heads per kv head (TinyLlama): two workgroups of 4 heads per (kv head, split). K/V rows are read twice
    // (the cache rows of such models are short), the per-row softmax work of a workgroup halves
    case 8: return launch_hg<HD, 4>(a, fuse, s);
    default: tl_set_error("attention: %d query heads per kv head is not supported by the split kernel", G); return 2;
  }
}

}  // namespace

bool tl_attn_split_supported(int nh, int nkv, int hd) {
  if (nkv <= 0 || nh % nkv != 0) return false;
  const int G = nh / nkv;
  return (hd == 64 || hd == 128) && (G == 1 || G == 2 || G == 3 || G == 4 || G == 8);
}

size_t tl_attn_ws_floats(int nh, int hd) { return (size_t)nh * TL_ATTN_MAX_SPLITS * (hd + 4); }

// the split kernel alone: leaves partials in a.ws (combined by the consumer)
int tl_launch_attn_split(const AttnArgs& a, bool fuse, hipStream_t s) {
  TL_REQUIRE(tl_attn_split_supported(a.nh, a.nkv, a.hd), "attention: unsupported head configuration %d/%d/%d", a.nh,
             a.nkv, a.hd);
  TL_REQUIRE(a.ws, "attention: workspace missing");
  const int G = a.nh / a.nkv;
  return a.hd == 64 ? launch_h<64>(a, G, fuse, s) : launch_h<128>(a, G, fuse, s);
}

int tl_launch_attn_combine(const AttnArgs& a, hipStream_t s) {
  attn_combine_kernel<<<a.nh, 128, 0, s>>>(a.ws, a.out, a.hd, a.seq_len, a.pos_dev);
  TL_HIP(hipGetLastError());
  return 0;
}
