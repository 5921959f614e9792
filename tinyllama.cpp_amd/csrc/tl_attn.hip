// tl_attn.hip — decode attention over the device-resident KV cache, split over the sequence ("flash-decoding"),
// with RoPE of q/k and the KV-cache write of the new position fused in for the engine's decode step.
//
// Replaces attention_cuda / attention_cuda_optimized (cuda_kernels.h:259,607: one block per head, scores for the
// whole context in shared memory) + rope_cuda (:239) + update_kv_cache_cuda (:344, one launch per kv head per K/V per
// layer). The math is the CPU path's (model.cpp:609-663, utils.cpp:462-486, 675-698): adjacent-pair RoPE from the
// host-computed table, scores = (q . K[t]) / sqrt(hd), softmax with max-subtraction and 1/(sum + 1e-9), out = sum p V.
//
// Shape (HBM/latency-bound byte work, no MFMA):
//   - grid (kv head, sequence split): one workgroup serves ALL query heads of a kv head (GQA group), so every K/V
//     byte is read once; a split is SL consecutive positions, chosen so that kv_heads x splits ~ the CU count at the
//     longest context the cache holds;
//   - a K/V row is read by HD/4 lanes x 16 B (one or two 256/512-B rows per wave load), 4 rows per lane in flight;
//   - splits that hold positions write (max, sum, unnormalised out) partials; the last workgroup of a kv head to arrive
//     (agent-scope release / counter / acquire) combines them. A single active split writes the result directly;
//   - FUSE: the workgroup rotates its query heads into LDS; the split that owns the new position rotates k, writes
//     K/V of that position into the cache and takes that position's score/value from LDS instead of from memory.
#include <math.h>

#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ATT_NW = 8;           // waves per workgroup
constexpr int ATT_THREADS = ATT_NW * 64;
constexpr int ATT_U = 4;            // K/V rows in flight per lane

__device__ __forceinline__ float wmax64(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// adjacent-pair rotation (utils.cpp:462-486); contraction off: each product is rounded separately as on the host
__device__ __forceinline__ void rope2(float a, float b, float c, float s, float& o0, float& o1) {
#pragma clang fp contract(off)
  o0 = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
  o1 = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
}

template <int HD, int G, bool FUSE>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_split_kernel(AttnArgs a) {
  constexpr int LP = HD / 4;    // lanes per K/V row
  constexpr int PP = 64 / LP;   // rows per wave load
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int SL = a.split_len;
  float* qs = sm;                      // [G][HD]
  float* kr = qs + G * HD;             // [HD] rotated k of the new position (FUSE owner)
  float* vr = kr + HD;                 // [HD]
  float* red = vr + HD;                // 64 floats: m[g] at g, l[g] at 8 + g, flags at 32
  float* sc = red + 64;                // [G][SL]
  float* part = sc + G * SL;           // [ATT_NW][G][HD]
  const int kvh = blockIdx.x, split = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = a.pos_dev ? *a.pos_dev : a.seq_len - 1;
  const int len = pos + 1;
  const int active = (len + SL - 1) / SL;
  if (split >= active || pos < 0 || pos >= a.max_seq) return;
  const int kv_dim = a.nkv * HD;
  const int t0 = split * SL;
  const int t1 = len < t0 + SL ? len : t0 + SL;
  const bool owner = FUSE && split == active - 1;  // this split holds the new position, still only in registers
  const int n = t1 - t0;                           // positions of this split
  const int ncache = owner ? n - 1 : n;            // ... of which come from the cache

  // ---- phase 0: query heads (and the new k / v) into LDS ---------------------------------------------------
  constexpr int half = HD / 2;
  if (FUSE) {
    const float* tp = a.table + 2 * (size_t)pos * half;
    for (int i = tid; i < G * half; i += ATT_THREADS) {
      const int g = i / half, p = i % half;
      const float* qh = a.q + (size_t)(kvh * G + g) * HD;
      rope2(qh[2 * p], qh[2 * p + 1], tp[2 * p], tp[2 * p + 1], qs[g * HD + 2 * p], qs[g * HD + 2 * p + 1]);
    }
    if (owner) {
      for (int p = tid; p < half; p += ATT_THREADS) {
        const float* kh = a.k + (size_t)kvh * HD;
        float o0, o1;
        rope2(kh[2 * p], kh[2 * p + 1], tp[2 * p], tp[2 * p + 1], o0, o1);
        kr[2 * p] = o0;
        kr[2 * p + 1] = o1;
        float* dst = a.kc + (size_t)pos * kv_dim + (size_t)kvh * HD;
        dst[2 * p] = o0;
        dst[2 * p + 1] = o1;
      }
      for (int d = tid; d < HD; d += ATT_THREADS) {
        const float x = a.v[(size_t)kvh * HD + d];
        vr[d] = x;
        a.vc[(size_t)pos * kv_dim + (size_t)kvh * HD + d] = x;
      }
    }
  } else {
    for (int i = tid; i < G * HD; i += ATT_THREADS) qs[i] = a.q[(size_t)kvh * G * HD + i];
  }
  __syncthreads();

  // ---- phase 1: scores ------------------------------------------------------------------------------------
  const int sub = lane / LP, ln = lane % LP;
  f32x4 qf[G];
#pragma unroll
  for (int g = 0; g < G; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qs + g * HD + 4 * ln);
  const float* kbase = a.kc + (size_t)t0 * kv_dim + (size_t)kvh * HD + 4 * ln;
  const float* vbase = a.vc + (size_t)t0 * kv_dim + (size_t)kvh * HD + 4 * ln;
  for (int base = 0; base < ncache; base += ATT_NW * PP * ATT_U) {
    f32x4 kk[ATT_U];
    int idx[ATT_U];
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {  // unconditional (clamped) loads: all ATT_U rows are in flight together
      idx[u] = base + (u * ATT_NW + wave) * PP + sub;
      const int ic = idx[u] < ncache ? idx[u] : ncache - 1;
      kk[u] = *reinterpret_cast<const f32x4*>(kbase + (size_t)ic * kv_dim);
    }
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float s = (qf[g][0] * kk[u][0] + qf[g][1] * kk[u][1]) + (qf[g][2] * kk[u][2] + qf[g][3] * kk[u][3]);
#pragma unroll
        for (int o = LP / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (ln == 0 && idx[u] < ncache) sc[g * SL + idx[u]] = s * a.scale;
      }
    }
  }
  if (owner && wave == 0) {  // the new position, from LDS
    const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + 4 * ln);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s = (qf[g][0] * kk[0] + qf[g][1] * kk[1]) + (qf[g][2] * kk[2] + qf[g][3] * kk[3]);
#pragma unroll
      for (int o = LP / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) sc[g * SL + (n - 1)] = s * a.scale;
    }
  }
  __syncthreads();

  // ---- phase 2: per-head max / exp / sum over the split --------------------------------------------------------
  for (int g = wave; g < G; g += ATT_NW) {
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, sc[g * SL + i]);
    m = wmax64(m);
    float l = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float e = expf(sc[g * SL + i] - m);
      sc[g * SL + i] = e;
      l += e;
    }
    l = wsum64(l);
    if (lane == 0) {
      red[g] = m;
      red[8 + g] = l;
    }
  }
  __syncthreads();

  // ---- phase 3: out[g][:] = sum_t p[g][t] V[t][:] ------------------------------------------------------------------
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < ncache; base += ATT_NW * PP * ATT_U) {
    f32x4 vv[ATT_U];
    int idx[ATT_U];
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      idx[u] = base + (u * ATT_NW + wave) * PP + sub;
      const int ic = idx[u] < ncache ? idx[u] : ncache - 1;
      vv[u] = *reinterpret_cast<const f32x4*>(vbase + (size_t)ic * kv_dim);
    }
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      const bool ok = idx[u] < ncache;
      const int ic = ok ? idx[u] : 0;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float p = ok ? sc[g * SL + ic] : 0.f;
        acc[g][0] = fmaf(p, vv[u][0], acc[g][0]);
        acc[g][1] = fmaf(p, vv[u][1], acc[g][1]);
        acc[g][2] = fmaf(p, vv[u][2], acc[g][2]);
        acc[g][3] = fmaf(p, vv[u][3], acc[g][3]);
      }
    }
  }
  if (owner && wave == 0 && sub == 0) {
    const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + 4 * ln);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float p = sc[g * SL + (n - 1)];
      acc[g][0] = fmaf(p, vv[0], acc[g][0]);
      acc[g][1] = fmaf(p, vv[1], acc[g][1]);
      acc[g][2] = fmaf(p, vv[2], acc[g][2]);
      acc[g][3] = fmaf(p, vv[3], acc[g][3]);
    }
  }
  // fold the PP row slots of a wave, then the waves through LDS
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int o = 32; o >= LP; o >>= 1) {
      acc[g][0] += __shfl_xor(acc[g][0], o);
      acc[g][1] += __shfl_xor(acc[g][1], o);
      acc[g][2] += __shfl_xor(acc[g][2], o);
      acc[g][3] += __shfl_xor(acc[g][3], o);
    }
    if (sub == 0) *reinterpret_cast<f32x4*>(part + (size_t)(wave * G + g) * HD + 4 * ln) = acc[g];
  }
  __syncthreads();

  // ---- phase 4: result or partial + last-arriver combine -------------------------------------------------------------
  const int NS = a.nsplit_max;
  constexpr int REC = HD + 2;  // partial record: out[HD], max, sum
  float* rec0 = a.ws + ((size_t)kvh * NS) * G * REC;
  for (int i = tid; i < G * HD; i += ATT_THREADS) {
    const int g = i / HD, d = i % HD;
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_NW; ++w) o += part[(size_t)(w * G + g) * HD + d];
    if (active == 1) {
      a.out[(size_t)(kvh * G + g) * HD + d] = o * (1.0f / (red[8 + g] + 1e-9f));  // utils.cpp:692
    } else {
      float* rec = rec0 + ((size_t)split * G + g) * REC;
      rec[d] = o;
      if (d == 0) {
        rec[HD] = red[g];
        rec[HD + 1] = red[8 + g];
      }
    }
  }
  if (active == 1) return;
  // publish: every storing wave drains its stores, the workgroup meets, ONE lane releases at agent scope and arrives
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(a.counters + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == active - 1;
    if (last) {
      __hip_atomic_store(a.counters + kvh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    red[32] = __int_as_float(last);
  }
  __syncthreads();
  if (__float_as_int(red[32]) == 0) return;
  // the last arriver of this kv head: combine the `active` partials in split order
  for (int i = tid; i < G * HD; i += ATT_THREADS) {
    const int g = i / HD, d = i % HD;
    float M = -INFINITY;
    for (int s = 0; s < active; ++s) M = fmaxf(M, rec0[((size_t)s * G + g) * REC + HD]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < active; ++s) {
      const float* rec = rec0 + ((size_t)s * G + g) * REC;
      const float w = expf(rec[HD] - M);
      L = fmaf(rec[HD + 1], w, L);
      O = fmaf(rec[d], w, O);
    }
    a.out[(size_t)(kvh * G + g) * HD + d] = O * (1.0f / (L + 1e-9f));
  }
}

template <int HD, int G>
int launch_hg(const AttnArgs& a, bool fuse, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)G * HD + 2 * HD + 64 + (size_t)G * a.split_len + (size_t)ATT_NW * G * HD);
  TL_REQUIRE(lds <= 160 * 1024, "attention: split length %d needs %zu B of LDS", a.split_len, lds);
  const dim3 grid(a.nkv, a.nsplit_max);
  if (fuse) {
    static bool attr = false;
    if (!attr) {
      TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_split_kernel<HD, G, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    attn_decode_split_kernel<HD, G, true><<<grid, ATT_THREADS, lds, s>>>(a);
  } else {
    static bool attr = false;
    if (!attr) {
      TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_split_kernel<HD, G, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
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
    case 8: return launch_hg<HD, 8>(a, fuse, s);
    default: tl_set_error("attention: %d query heads per kv head is not supported by the split kernel", G); return 2;
  }
}

}  // namespace

bool tl_attn_split_supported(int nh, int nkv, int hd) {
  if (nkv <= 0 || nh % nkv != 0) return false;
  const int G = nh / nkv;
  return (hd == 64 || hd == 128) && (G == 1 || G == 2 || G == 3 || G == 4 || G == 8);
}

void tl_attn_geometry(int max_seq, int nkv, int* split_len, int* nsplit) {
  // kv_heads x splits ~ 256 workgroups at the longest context; a split is 64..1024 positions (power of two)
  long want = ((long)max_seq * nkv + 255) / 256;
  int sl = 64;
  while (sl < want && sl < 1024) sl <<= 1;
  *split_len = sl;
  *nsplit = (max_seq + sl - 1) / sl;
}

size_t tl_attn_ws_floats(int nh, int hd, int nsplit) { return (size_t)nh * nsplit * (hd + 2); }

int tl_launch_attn_split(const AttnArgs& a, bool fuse, hipStream_t s) {
  TL_REQUIRE(tl_attn_split_supported(a.nh, a.nkv, a.hd), "attention: unsupported head configuration %d/%d/%d", a.nh,
             a.nkv, a.hd);
  TL_REQUIRE(a.ws && a.counters && a.split_len > 0 && a.nsplit_max > 0, "attention: workspace missing");
  TL_REQUIRE((size_t)a.split_len * a.nsplit_max >= (size_t)(a.pos_dev ? a.max_seq : a.seq_len),
             "attention: %d splits of %d positions do not cover the sequence", a.nsplit_max, a.split_len);
  const int G = a.nh / a.nkv;
  return a.hd == 64 ? launch_h<64>(a, G, fuse, s) : launch_h<128>(a, G, fuse, s);
}
