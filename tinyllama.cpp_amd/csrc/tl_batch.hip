// tl_batch.hip — the position-dependent kernels of the MULTI-SEQUENCE decode step: B sequences advance one
// token each per step; the projections run as skinny GEMMs on the matrix cores (tl_gemm.hip, the weights are read once
// for all B tokens), and what depends on (sequence, position) lives here:
//   - RoPE of q / k at each token's own position + K/V write into the SEQUENCE-MAJOR cache
//     [seq][max_seq][n_kv_heads][head_dim] (forward_cpu_batch_generation, model.cpp:1302-1320: apply_rope_vector per
//     token, kv_offset = (seq_idx * max_seq_len + pos) * kv_dim);
//   - attention of every token over ITS sequence's cache rows 0..pos (model.cpp:1325-1382; the math is the
//     single-sequence path's: scores scaled by 1/sqrt(hd), max-subtracted softmax, 1/(sum + 1e-9));
//   - first-maximum arg-max of every token's logits row + the state advance of the device-resident generation loop
//     (api.cpp:1069-1095; the reference applies no repetition penalty on this path, api.cpp:993,1084).
// Semantics note: the reference's parallel path reads history_len = batch_seq_lens[seq] BEFORE that step's update
// (model.cpp:1327, 1494-1496), i.e. a token does not attend to itself there, unlike in its own sequential fallback
// (api.cpp:1116-1130 -> generate()). The target here is the sequential fallback: every sequence decodes exactly as it
// would alone, which is also what the parity tests check against the single-sequence oracle.
#include <math.h>

#include <type_traits>

#include "tl_chunk.h"
#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ float exp_diff(float m, float m_new) { return m == -INFINITY ? 0.f : expf(m - m_new); }

// adjacent-pair rotation (utils.cpp:462-486), products rounded separately as on the host
__device__ __forceinline__ void rot2(float a, float b, float c, float s, float& o0, float& o1) {
#pragma clang fp contract(off)
  o0 = __fsub_rn(__fmul_rn(a, c), __fmul_rn(b, s));
  o1 = __fadd_rn(__fmul_rn(a, s), __fmul_rn(b, c));
}

// q [T][nq][hd], k [T][nkv][hd] rotated in place at pos[t]; k, v of token t written to cache row (seq[t], pos[t])
// pq / pk / pv: S != 0 = that projection is still the partial tiles of a K-split GEMM and is summed here, in slice order
// (what ksplit_reduce_kernel would have left in q / k / v)
__device__ __forceinline__ float ld_parts(const float* base, const GemmParts& g, size_t o) {
  if (!g.S) return base[o];
  float acc = g.part[o];
  for (int z = 1; z < g.S; ++z) acc += g.part[(size_t)z * g.n + o];
  return acc;
}
__global__ void batch_rope_kv_kernel(float* __restrict__ q, float* __restrict__ k, const float* __restrict__ v,
                                     float* __restrict__ kc, float* __restrict__ vc, size_t seq_stride,
                                     const float* __restrict__ table, const int* __restrict__ pos,
                                     const int* __restrict__ seq, int T, int nq, int nkv, int hd, int max_seq, int n_seq,
                                     GemmParts pq, GemmParts pk, GemmParts pv, int start_pos) {
  const int half = hd >> 1, kv_dim = nkv * hd;
  const int per_tok = (nq + nkv) * half + kv_dim;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)T * per_tok) return;
  const int t = (int)(gid / per_tok), rem = (int)(gid % per_tok);
  const int p = pos ? pos[t] : start_pos + t, sq = seq ? seq[t] : 0;  // (no arrays: one sequence at consecutive positions)
  if (p < 0 || p >= max_seq || sq < 0 || sq >= n_seq) return;  // (validated on the host; never writes out of range)
  const size_t row = (size_t)sq * seq_stride + (size_t)p * kv_dim;
  if (rem < (nq + nkv) * half) {
    const int h = rem / half, i = rem % half;
    const float* tp = table + 2 * (size_t)p * half;
    if (h < nq) {
      float* x = q + ((size_t)t * nq + h) * hd;
      const size_t o = ((size_t)t * nq + h) * hd + 2 * i;
      float o0, o1;
      rot2(ld_parts(q, pq, o), ld_parts(q, pq, o + 1), tp[2 * i], tp[2 * i + 1], o0, o1);
      x[2 * i] = o0;
      x[2 * i + 1] = o1;
    } else {
      float* x = k + ((size_t)t * nkv + (h - nq)) * hd;
      const size_t o = ((size_t)t * nkv + (h - nq)) * hd + 2 * i;
      float o0, o1;
      rot2(ld_parts(k, pk, o), ld_parts(k, pk, o + 1), tp[2 * i], tp[2 * i + 1], o0, o1);
      x[2 * i] = o0;
      x[2 * i + 1] = o1;
      float* dst = kc + row + (size_t)(h - nq) * hd;
      dst[2 * i] = o0;
      dst[2 * i + 1] = o1;
    }
  } else {
    const int d = rem - (nq + nkv) * half;
    vc[row + d] = ld_parts(v, pv, (size_t)t * kv_dim + d);
  }
}

struct BatchAttnArgs {
  const float* q;   // [T][nh][hd], rotated
  const float* kc;  // layer base of sequence 0: [n_seq][max_seq][nkv][hd]
  const float* vc;
  size_t seq_stride;
  float* out;       // [T][nh][hd]
  const int* pos;
  const int* seq;
  int nh, nkv, max_seq, n_seq;
  float scale;
  // planes != NULL: the output goes straight into the activation planes of the attn_output GEMM (what
  // split_kernel<SPLIT_PLAIN> would make of `out`, bit for bit: bf16 planes [np][plane_stride] of the [T][nh * hd]
  // matrix + the sums of its 32-column blocks [nh * hd / 32][tpad]); `out` is then not written
  __bf16* planes;
  size_t plane_stride;
  float* xsum;
  int np, tpad;
};

// One workgroup per (kv head, token, head-group part): the G query heads of the group share every K/V row read.
// A K/V row is read by 16 lanes (one DPP row) x 16 B x HD/64 loads, so the dot's reduction is four DPP adds (the first
// version used HD/4 lanes per row and __shfl_xor, which is ds_bpermute on this target); U rows of K and of V per lane are
// in flight together and folded into a running (max, sum, out) per lane group (online softmax), then the four lane
// groups of a wave and the waves are folded.
__device__ __forceinline__ float batch_row_sum16(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror
  return v;
}
template <int HD, int G>
__global__ __launch_bounds__(256) void attn_batch_decode_kernel(BatchAttnArgs a) {
  constexpr int NW = 4, LP = 16, NV = HD / 64, PP = 64 / LP, U = 8 / NV, RECL = HD + 4;
  constexpr int NR = NW * PP;  // row groups per workgroup (= one DPP row of lanes in the weights phase)
  static_assert(NR == 16, "the weights of a head's row groups are reduced inside one DPP row");
  __shared__ __attribute__((aligned(16))) float part[NR * G * RECL];
  __shared__ float red[16], wts[G * NR];
  __shared__ __attribute__((aligned(16))) float outbuf[G * HD];
  const int kvh = blockIdx.x, tok = blockIdx.y;
  const int head0 = (kvh * gridDim.z + blockIdx.z) * G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane / LP, ln = lane % LP;
  const int pos = a.pos[tok], sq = a.seq[tok];
  if (pos < 0 || pos >= a.max_seq || sq < 0 || sq >= a.n_seq) return;  // uniform
  const int len = pos + 1;
  const int kv_dim = a.nkv * HD;
  const float* kbase = a.kc + (size_t)sq * a.seq_stride + (size_t)kvh * HD + 4 * ln;
  const float* vbase = a.vc + (size_t)sq * a.seq_stride + (size_t)kvh * HD + 4 * ln;
  f32x4 qf[G][NV], acc[G][NV];
  float m[G], l[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      qf[g][v] = *reinterpret_cast<const f32x4*>(a.q + ((size_t)tok * a.nh + head0 + g) * HD + 64 * v + 4 * ln);
      acc[g][v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    m[g] = -INFINITY;
    l[g] = 0.f;
  }
  for (int base = 0; base < len; base += NW * PP * U) {
    f32x4 kk[U][NV], vv[U][NV];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // unconditional, clamped loads: the whole batch in flight together
      const int idx = base + (u * NW + wave) * PP + sub;
      valid[u] = idx < len;
      const size_t off = (size_t)(valid[u] ? idx : len - 1) * kv_dim;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        kk[u][v] = *reinterpret_cast<const f32x4*>(kbase + off + 64 * v);
        vv[u][v] = *reinterpret_cast<const f32x4*>(vbase + off + 64 * v);
      }
    }
    if constexpr (G == 4) {
      // the exps of the four heads once per quad (lane l: head l % 4, DPP quad broadcasts) instead of identically in all 16
      // lanes of a row: as in tl_attn.hip's fold (same values, a quarter of the exps)
      float s[G][U], mx[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        mx[g] = m[g];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v)
            d += (qf[g][v][0] * kk[u][v][0] + qf[g][v][1] * kk[u][v][1]) + (qf[g][v][2] * kk[u][v][2] + qf[g][v][3] * kk[u][v][3]);
          d = batch_row_sum16(d);
          s[g][u] = valid[u] ? d * a.scale : -INFINITY;
          mx[g] = fmaxf(mx[g], s[g][u]);
        }
      }
      const int gq = lane & 3;
      auto sel4 = [&](float x0, float x1, float x2, float x3) { return gq == 0 ? x0 : gq == 1 ? x1 : gq == 2 ? x2 : x3; };
      const float mxs = sel4(mx[0], mx[1], mx[2], mx[3]);
      const float cs = exp_diff(sel4(m[0], m[1], m[2], m[3]), mxs);  // (0 while nothing is valid: the state stays empty)
      float ps[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ps[u] = valid[u] ? expf(sel4(s[0][u], s[1][u], s[2][u], s[3][u]) - mxs) : 0.f;
      auto apply = [&](auto g_c) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value;
        constexpr int QP = g * 0x55;  // quad_perm: every lane of the quad reads lane g
        auto bcast = [&](float x) {
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), QP, 0xf, 0xf, false));
        };
        const float corr = bcast(cs);
        m[g] = mx[g];
        l[g] *= corr;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[g][v] *= corr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float p = bcast(ps[u]);
          l[g] += p;
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][v][e] = fmaf(p, vv[u][v][e], acc[g][v][e]);
        }
      };
      apply(std::integral_constant<int, 0>{});
      apply(std::integral_constant<int, 1>{});
      apply(std::integral_constant<int, 2>{});
      apply(std::integral_constant<int, 3>{});
      continue;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s[U];
      float mx = m[g];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v)
          d += (qf[g][v][0] * kk[u][v][0] + qf[g][v][1] * kk[u][v][1]) + (qf[g][v][2] * kk[u][v][2] + qf[g][v][3] * kk[u][v][3]);
        d = batch_row_sum16(d);
        s[u] = valid[u] ? d * a.scale : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      if (mx == -INFINITY) continue;  // nothing valid yet (uniform over the lane group)
      const float corr = exp_diff(m[g], mx);
      m[g] = mx;
      l[g] *= corr;
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[g][v] *= corr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float p = valid[u] ? expf(s[u] - mx) : 0.f;
        l[g] += p;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[g][v][e] = fmaf(p, vv[u][v][e], acc[g][v][e]);
      }
    }
  }
  // every row group's state -> an LDS record (folding the four groups of a wave in registers first is 2 x 10 LDS-path
  // shuffles per head and lane, and four exps per output element on top: as in tl_attn_body.h)
  const int rg = wave * PP + sub;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* rec = part + (size_t)(rg * G + g) * RECL;
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(rec + 64 * v + 4 * ln) = acc[g][v];
    if (ln == 0) {
      rec[HD] = m[g];
      rec[HD + 1] = l[g];
    }
  }
  __syncthreads();
  if (tid < G * NR) {  // lane 16 g + r: the weight exp(m_r - M) of row group r of head g, once; 1 / (L + 1e-9)
    const int g = tid / NR, r = tid % NR;
    const float* rec = part + (size_t)(r * G + g) * RECL;
    const float mr = rec[HD];
    auto dmax = [](float v, auto ctrl_c) {
      constexpr int CTRL = decltype(ctrl_c)::value;
      const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
      return fmaxf(v, __builtin_bit_cast(float, t));
    };
    float M = dmax(mr, std::integral_constant<int, 0xB1>{});
    M = dmax(M, std::integral_constant<int, 0x4E>{});
    M = dmax(M, std::integral_constant<int, 0x141>{});
    M = dmax(M, std::integral_constant<int, 0x140>{});
    const float w = exp_diff(mr, M);  // (0 for a row group without rows)
    const float L = batch_row_sum16(rec[HD + 1] * w);
    wts[tid] = w;
    if (r == 0) red[8 + g] = 1.0f / (L + 1e-9f);  // utils.cpp:692
  }
  __syncthreads();
  for (int i = tid; i < G * (HD / 4); i += 256) {
    const int g = i / (HD / 4), c4 = i % (HD / 4);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; ++r) {  // row-group order
      const f32x4 x = *reinterpret_cast<const f32x4*>(part + (size_t)(r * G + g) * RECL + 4 * c4);
      const float w = wts[g * NR + r];
      o[0] = fmaf(x[0], w, o[0]); o[1] = fmaf(x[1], w, o[1]); o[2] = fmaf(x[2], w, o[2]); o[3] = fmaf(x[3], w, o[3]);
    }
    const float inv = red[8 + g];
    o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
    if (a.planes) *reinterpret_cast<f32x4*>(outbuf + g * HD + 4 * c4) = o;
    else *reinterpret_cast<f32x4*>(a.out + ((size_t)tok * a.nh + head0 + g) * HD + 4 * c4) = o;
  }
  if (!a.planes) return;  // uniform
  __syncthreads();
  // split_kernel<SPLIT_PLAIN>'s body for this workgroup's G * HD consecutive columns: 8 columns per thread, the sum of a
  // 32-column block over 4 neighbouring threads
  const int K = a.nh * HD;
  if (tid < G * HD / 8) {
    const int i0 = head0 * HD + tid * 8;
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(outbuf + tid * 8), hi = *reinterpret_cast<const f32x4*>(outbuf + tid * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = lo[e];
      v[4 + e] = hi[e];
    }
    float bs = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    bs += __shfl_xor(bs, 1);
    bs += __shfl_xor(bs, 2);
    if ((tid & 3) == 0) a.xsum[(size_t)(i0 >> 5) * a.tpad + tok] = bs;
    for (int p = 0; p < a.np; ++p) {
      bf16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        b[e] = (__bf16)v[e];
        v[e] -= (float)b[e];
      }
      *reinterpret_cast<bf16x8*>(a.planes + p * a.plane_stride + (size_t)tok * K + i0) = b;
    }
  }
}

template <int HD>
int launch_attn_h(const BatchAttnArgs& a, int T, hipStream_t s) {
  const int grp = a.nh / a.nkv;
  auto go = [&](auto g_c, int parts) {
    constexpr int G = decltype(g_c)::value;
    attn_batch_decode_kernel<HD, G><<<dim3(a.nkv, T, parts), 256, 0, s>>>(a);
  };
  switch (grp) {
    case 1: go(std::integral_constant<int, 1>{}, 1); break;
    case 2: go(std::integral_constant<int, 2>{}, 1); break;
    case 3: go(std::integral_constant<int, 3>{}, 1); break;
    case 4: go(std::integral_constant<int, 4>{}, 1); break;
    case 8: go(std::integral_constant<int, 4>{}, 2); break;
    default: tl_set_error("batched decode attention: %d query heads per kv head is not supported", grp); return 2;
  }
  TL_HIP(hipGetLastError());
  return 0;
}

// first maximum of every logits row (lowest index on ties, api.cpp:79,135 / std::max_element) + state advance
__global__ __launch_bounds__(1024) void batch_argmax_advance_kernel(const float* __restrict__ logits, int V,
                                                                    int* __restrict__ tokens, int* __restrict__ pos,
                                                                    int* __restrict__ ids_out) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int t = blockIdx.x;
  const float* row = logits + (size_t)t * V;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += 1024) {
    const float v = row[i];
    if (v > best) {
      best = v;
      idx = i;
    }
  }
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
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
        best = bv[w];
        idx = bi[w];
      }
    if (idx == 0x7fffffff) idx = 0;  // a row of NaN / -inf only
    if (ids_out) ids_out[t] = idx;
    if (tokens) tokens[t] = idx;
    if (pos) pos[t] += 1;
  }
}

}  // namespace

int tl_launch_batch_rope_kv(float* q, float* k, const float* v, float* kc, float* vc, size_t seq_stride, const float* table,
                            const int* pos_dev, const int* seq_dev, int T, int nq, int nkv, int hd, int max_seq, int n_seq,
                            hipStream_t s, const GemmParts* pq, const GemmParts* pk, const GemmParts* pv, int start_pos) {
  TL_REQUIRE(q && k && v && kc && vc && table && ((pos_dev && seq_dev) || (!pos_dev && !seq_dev && start_pos >= 0)) && T > 0 &&
                 hd % 2 == 0, "batch rope/kv: bad argument");
  const GemmParts none = {nullptr, 0, 0};
  const GemmParts gq = pq ? *pq : none, gk = pk ? *pk : none, gv = pv ? *pv : none;
  TL_REQUIRE((!gq.S || gq.n == (size_t)T * nq * hd) && (!gk.S || gk.n == (size_t)T * nkv * hd) &&
                 (!gv.S || gv.n == (size_t)T * nkv * hd), "batch rope/kv: pending sums of another shape");
  const size_t n = (size_t)T * ((size_t)(nq + nkv) * (hd / 2) + (size_t)nkv * hd);
  batch_rope_kv_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(q, k, v, kc, vc, seq_stride, table, pos_dev, seq_dev, T, nq,
                                                                   nkv, hd, max_seq, n_seq, gq, gk, gv, start_pos);
  TL_HIP(hipGetLastError());
  return 0;
}

bool tl_batch_attn_supported(int nh, int nkv, int hd) {
  if (nkv <= 0 || nh % nkv != 0) return false;
  const int g = nh / nkv;
  return (hd == 64 || hd == 128) && (g == 1 || g == 2 || g == 3 || g == 4 || g == 8);
}

int tl_launch_batch_attn(const float* q, const float* kc, const float* vc, size_t seq_stride, float* out, const int* pos_dev,
                         const int* seq_dev, int T, int nh, int nkv, int hd, int max_seq, int n_seq, float scale,
                         hipStream_t s, void* planes, size_t plane_stride, int terms) {
  TL_REQUIRE(tl_batch_attn_supported(nh, nkv, hd), "batched decode attention: unsupported head configuration %d/%d/%d", nh,
             nkv, hd);
  TL_REQUIRE(q && kc && vc && out && pos_dev && seq_dev && T > 0 && T <= 65535, "batched decode attention: bad argument");
  BatchAttnArgs a = {q, kc, vc, seq_stride, out, pos_dev, seq_dev, nh, nkv, max_seq, n_seq, scale, nullptr, 0, nullptr, 0, 0};
  if (planes) {  // layout as tl_launch_split makes it: three bf16 planes, then the block sums
    TL_REQUIRE(plane_stride >= tl_gemm_padded_tokens(T) * (size_t)nh * hd, "batched decode attention: activation planes too small");
    a.planes = static_cast<__bf16*>(planes);
    a.plane_stride = plane_stride;
    a.xsum = reinterpret_cast<float*>(a.planes + 3 * plane_stride);
    a.np = terms == 1 ? 1 : terms == 3 ? 2 : 3;
    a.tpad = (int)tl_gemm_padded_tokens(T);
  }
  return hd == 64 ? launch_attn_h<64>(a, T, s) : launch_attn_h<128>(a, T, s);
}

int tl_launch_batch_argmax_advance(const float* logits, int T, int V, int* tokens_dev, int* pos_dev, int* ids_out_dev,
                                   hipStream_t s) {
  TL_REQUIRE(logits && T > 0 && V > 0, "batch argmax: bad argument");
  batch_argmax_advance_kernel<<<T, 1024, 0, s>>>(logits, V, tokens_dev, pos_dev, ids_out_dev);
  TL_HIP(hipGetLastError());
  return 0;
}
