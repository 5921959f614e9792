// tl_attn_body.h -- the split decode-attention workgroup body (see tl_attn.hip for the design), shared by the
// launch-per-attention kernel (tl_attn.hip) and the attn_output launch that carries its layer's attention (tl_gemv.hip).
#pragma once
#include <math.h>

#include <type_traits>

#include "tl_chunk.h"

namespace tl_attn {

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

// sum over the 16 lanes of a DPP row, result in every lane of the row (no LDS traffic)
__device__ __forceinline__ float row_sum16(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror
  return v;
}
// max / sum over 32 consecutive lanes (two DPP rows), result in every one of them: four DPP steps inside the rows and ONE
// LDS-path exchange between them (the xor butterfly is five dependent ds_bpermute round trips per reduction)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, t));
}
__device__ __forceinline__ float max32(float v) {
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  v = dpp_max<0x141>(v);
  v = dpp_max<0x140>(v);
  return fmaxf(v, __shfl_xor(v, 16));
}
__device__ __forceinline__ float sum32(float v) {
  v = row_sum16(v);
  return v + __shfl_xor(v, 16);
}
// agent-scope accesses for data handed between workgroups inside one launch (the L2s of the 8 XCDs are not coherent with
// each other for ordinary accesses)
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dot4(f32x4 x, f32x4 y) { return (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]); }

// The body of the split kernel for workgroup (kvh, split, zpart) of an ATT_THREADS-thread workgroup; `sm` = its LDS
// (attn_split_lds_floats). U_ = row slots in flight per lane (0: 16 / NV, the launch-per-attention kernel's 128 VGPRs of K
// and V); SC1OUT: the partial records are stored with agent-scope (written-through) stores, for a consumer in the SAME
// launch (tl_gemv.hip, the attention folded into the attn_output launch).
template <int HD, int G, bool FUSE, int U_ = 0, bool SC1OUT = false>
__device__ __forceinline__ void attn_split_body(const AttnArgs& a, const int kvh, const int split, const int zpart, float* sm) {
#pragma clang fp contract(off)  // (tl_attn.hip is built without contraction; the same arithmetic wherever this is included)
  constexpr int LP = 16;                 // lanes per K/V row = one DPP row: the dot's reduction is four DPP adds
  constexpr int NV = HD / 64;            // 16-byte loads per lane and row: lane ln holds d in [4 ln, 4 ln + 4) (+ 64)
  constexpr int PP = 64 / LP;            // rows per wave load = row groups per wave
  constexpr int NR = ATT_NW * PP;        // row groups per workgroup, each with its own running softmax state
  constexpr int U = U_ > 0 ? U_ : 16 / NV;  // row slots in flight per lane (16 / NV: 128 VGPRs of K and V, 256 rows per workgroup at hd 128)
  constexpr int SB = 4;                  // slots folded at a time, in arrival order, while the later ones are in flight
  constexpr int RECL = HD + 4;           // partial record: out[HD], max, sum, 2 pad (16-B aligned)
  float* red = sm;                       // 64 floats: M[g] at g, L at 8 + g
  float* wts = red + 64;                 // [G][NR] running max of every row group's record (the merge phase: split weights)
  float* lls = wts + NR * G;             // [G][NR] running sum of every row group's record
  float* wl = lls + NR * G;              // [ATT_NW][NR] per wave: the weights of its head's row groups (output loop)
  float* part = wl + ATT_NW * NR;        // [NR][G][RECL]
  const int head0 = (kvh * a.zparts + zpart) * G;  // a.zparts workgroups share a kv head, G query heads each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = a.pos_dev ? *a.pos_dev : a.seq_len - 1;
  const int len = pos + 1;
  const int SL = tl_attn_split_len(len);
  const int active = (len + SL - 1) / SL;
  if (split >= active || pos < 0 || pos >= a.max_seq) return;  // (the combine only reads splits < active)
  const int kv_dim = a.kv_row;
  const size_t hoff = (size_t)kvh * (size_t)a.kv_head;
  const int t0 = split * SL;
  const int t1 = len < t0 + SL ? len : t0 + SL;
  const bool owner = FUSE && split == active - 1;  // this split holds the new position, which is not in the cache yet
  const int ncache = owner ? t1 - t0 - 1 : t1 - t0;  // positions of this split that come from the cache
  const int sub = lane / LP, ln = lane % LP;

  // ---- loads, in the order they are needed (vmcnt retires in issue order): the query heads (+ the RoPE row), then the
  //      first batch of K/V rows. No LDS staging and no barrier in front of the rows: every lane rotates the 4 * NV
  //      elements of each query head it needs itself (the row groups repeat that, from L1/L2). Staging the heads through
  //      LDS instead (one pair per thread, rows in flight across the barrier) measured slower at every length.
  f32x4 qf[G][NV], tq[NV];
  if (FUSE) {
    const float* tp = a.table + (size_t)pos * HD;  // [hd/2][2]: pair p at 2p; the pairs of d = 4 ln .. 4 ln + 3 at 4 ln
#pragma unroll
    for (int v = 0; v < NV; ++v) tq[v] = *reinterpret_cast<const f32x4*>(tp + 64 * v + 4 * ln);
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int v = 0; v < NV; ++v) qf[g][v] = *reinterpret_cast<const f32x4*>(a.q + (size_t)(head0 + g) * HD + 64 * v + 4 * ln);
  // FUSE: k and v of the new position, asked for here with everything else (every workgroup loads them, the owner's last
  // wave uses them: loading them where they are used put one more round trip behind that wave's fold, on the path of
  // the workgroup every launch waits for)
  f32x4 kn[NV], vn[NV];
  if (FUSE) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      kn[v] = *reinterpret_cast<const f32x4*>(a.k + (size_t)kvh * HD + 64 * v + 4 * ln);
      vn[v] = *reinterpret_cast<const f32x4*>(a.v + (size_t)kvh * HD + 64 * v + 4 * ln);
    }
  }

  const float* kbase = a.kc + (size_t)t0 * kv_dim + hoff + 4 * ln;
  const float* vbase = a.vc + (size_t)t0 * kv_dim + hoff + 4 * ln;
  f32x4 kk[U * NV], vv[U * NV];
  unsigned valid = 0;  // bit u: slot u of the batch in flight holds a row of this split
  // slot u of a batch holds row base + (u * NW + wave) * PP + sub (rows go round the waves first, so short splits use
  // few slots in every wave); unconditional (clamped) loads: all K and V rows of the batch in flight together
  auto issue = [&](int base, auto uc_c) __attribute__((always_inline)) {
    constexpr int UC = decltype(uc_c)::value;
    valid = 0;
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int idx = base + (u * ATT_NW + wave) * PP + sub;
      const bool ok = idx < ncache;
      valid |= ok ? 1u << u : 0u;
      const size_t off = (size_t)(ok ? idx : ncache - 1) * kv_dim;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        kk[u * NV + v] = *reinterpret_cast<const f32x4*>(kbase + off + 64 * v);
        vv[u * NV + v] = *reinterpret_cast<const f32x4*>(vbase + off + 64 * v);
      }
    }
  };
  // slot class of a batch for this wave (wave-uniform): 0 (no rows), 2, 4, 8 or U
  auto slot_class = [&](int base) __attribute__((always_inline)) {
    const int mine = ncache - base - wave * PP;  // rows at or after this wave's first slot
    if (mine <= 0) return 0;
    const int slots = (mine + ATT_NW * PP - 1) / (ATT_NW * PP);
    if constexpr (U <= 2) return U;
    else if constexpr (U <= 4) return slots <= 2 ? 2 : U;
    else return slots <= 2 ? 2 : slots <= 4 ? 4 : (slots <= 8 || U == 8) ? 8 : U;
  };
  // ---- one pass over the rows with a running (max, sum, out) per 16-lane row group (online softmax) -------------
  f32x4 acc[G][NV];
  float m[G], l[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[g][v] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[g] = -INFINITY;
    l[g] = 0.f;
  }
  // folds `cnt` row slots (K rows kk_, V rows vv_, validity bits valid_) into the running state
  auto fold = [&](const f32x4* kk_, const f32x4* vv_, unsigned valid_, auto cnt_c) __attribute__((always_inline)) {
    constexpr int cnt = decltype(cnt_c)::value;
    if constexpr (G >= 2) {
      // Several heads per workgroup: the 16 lanes of a row hold the SAME scores of all of them after the row sums, and the
      // exps (a correction and one weight per slot and head: 4 (1 + cnt) of ~15 instructions each, the largest single
      // item of this kernel's arithmetic, which at long context is what bounds it) were computed identically in every
      // lane. Here lane ln computes them for head ln % 4 only (with fewer than four heads the spare lanes repeat
      // the last one) and the quad hands them round with DPP broadcasts: the same values, 1 / G of the exps.
      float s[G][cnt], mx[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        mx[g] = m[g];
#pragma unroll
        for (int u = 0; u < cnt; ++u) {
          float d = dot4(qf[g][0], kk_[u * NV]);
          if (NV == 2) d += dot4(qf[g][NV - 1], kk_[u * NV + NV - 1]);
          d = row_sum16(d);
          s[g][u] = (valid_ >> u & 1u) ? d * a.scale : -INFINITY;
          mx[g] = fmaxf(mx[g], s[g][u]);
        }
      }
      const int gq = lane & 3;
      auto sel4 = [&](float x0, float x1, float x2, float x3) { return gq == 0 ? x0 : gq == 1 ? x1 : gq == 2 ? x2 : x3; };
      constexpr int G1 = 1, G2 = G > 2 ? 2 : G - 1, G3 = G > 3 ? 3 : G - 1;  // (heads of lanes 1, 2, 3 of a quad)
      const float mxs = sel4(mx[0], mx[G1], mx[G2], mx[G3]);
      const float cs = safe_exp_diff(sel4(m[0], m[G1], m[G2], m[G3]), mxs);  // (0 while nothing is valid: the state stays empty)
      float ps[cnt];
#pragma unroll
      for (int u = 0; u < cnt; ++u)
        ps[u] = (valid_ >> u & 1u) ? expf(sel4(s[0][u], s[G1][u], s[G2][u], s[G3][u]) - mxs) : 0.f;
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
        for (int v = 0; v < NV; ++v) {
          acc[g][v][0] *= corr; acc[g][v][1] *= corr; acc[g][v][2] *= corr; acc[g][v][3] *= corr;
        }
#pragma unroll
        for (int u = 0; u < cnt; ++u) {
          const float p = bcast(ps[u]);
          l[g] += p;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc[g][v][0] = fmaf(p, vv_[u * NV + v][0], acc[g][v][0]);
            acc[g][v][1] = fmaf(p, vv_[u * NV + v][1], acc[g][v][1]);
            acc[g][v][2] = fmaf(p, vv_[u * NV + v][2], acc[g][v][2]);
            acc[g][v][3] = fmaf(p, vv_[u * NV + v][3], acc[g][v][3]);
          }
        }
      };
      apply(std::integral_constant<int, 0>{});
      apply(std::integral_constant<int, 1>{});
      if constexpr (G > 2) apply(std::integral_constant<int, 2>{});
      if constexpr (G > 3) apply(std::integral_constant<int, 3>{});
      return;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s[cnt];
      float mx = m[g];
#pragma unroll
      for (int u = 0; u < cnt; ++u) {
        float d = dot4(qf[g][0], kk_[u * NV]);
        if (NV == 2) d += dot4(qf[g][NV - 1], kk_[u * NV + NV - 1]);
        d = row_sum16(d);
        s[u] = (valid_ >> u & 1u) ? d * a.scale : -INFINITY;
        mx = fmaxf(mx, s[u]);
      }
      if (mx == -INFINITY) continue;  // nothing valid yet (uniform over the row group)
      const float corr = safe_exp_diff(m[g], mx);
      m[g] = mx;
      l[g] *= corr;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc[g][v][0] *= corr; acc[g][v][1] *= corr; acc[g][v][2] *= corr; acc[g][v][3] *= corr;
      }
#pragma unroll
      for (int u = 0; u < cnt; ++u) {
        const float p = (valid_ >> u & 1u) ? expf(s[u] - mx) : 0.f;
        l[g] += p;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          acc[g][v][0] = fmaf(p, vv_[u * NV + v][0], acc[g][v][0]);
          acc[g][v][1] = fmaf(p, vv_[u * NV + v][1], acc[g][v][1]);
          acc[g][v][2] = fmaf(p, vv_[u * NV + v][2], acc[g][v][2]);
          acc[g][v][3] = fmaf(p, vv_[u * NV + v][3], acc[g][v][3]);
        }
      }
    }
  };
  auto fold_batch = [&](auto uc_c) __attribute__((always_inline)) {
    constexpr int UC = decltype(uc_c)::value;
    constexpr int C = UC < SB ? UC : SB;
#pragma unroll
    for (int u0 = 0; u0 < UC; u0 += C) fold(kk + u0 * NV, vv + u0 * NV, valid >> u0, std::integral_constant<int, C>{});
  };
  // The first batch of NW * PP * U rows (the only one unless len > 8192 at hd 128): its loads are issued BEFORE the query
  // rotation, which then waits for the q loads only; the row slots are folded in arrival order.
  const int uc0 = slot_class(0);
  constexpr int C4 = U >= 4 ? 4 : U, C8 = U >= 8 ? 8 : U;  // (slot classes never exceed U)
  if (uc0 == 2) issue(0, std::integral_constant<int, 2>{});
  else if (uc0 == 4) issue(0, std::integral_constant<int, C4>{});
  else if (uc0 == 8) issue(0, std::integral_constant<int, C8>{});
  else if (uc0 == U) issue(0, std::integral_constant<int, U>{});
  if (FUSE) {  // adjacent-pair rotation of the query elements in registers
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        f32x4 r;
        float o0, o1;
        rope2(qf[g][v][0], qf[g][v][1], tq[v][0], tq[v][1], o0, o1);
        r[0] = o0; r[1] = o1;
        rope2(qf[g][v][2], qf[g][v][3], tq[v][2], tq[v][3], o0, o1);
        r[2] = o0; r[3] = o1;
        qf[g][v] = r;
      }
#pragma unroll
    for (int v = 0; v < NV; ++v) {  // ... and of the new position's key
      f32x4 r;
      float o0, o1;
      rope2(kn[v][0], kn[v][1], tq[v][0], tq[v][1], o0, o1);
      r[0] = o0; r[1] = o1;
      rope2(kn[v][2], kn[v][3], tq[v][2], tq[v][3], o0, o1);
      r[2] = o0; r[3] = o1;
      kn[v] = r;
    }
  }
  if (uc0 == 2) fold_batch(std::integral_constant<int, 2>{});
  else if (uc0 == 4) fold_batch(std::integral_constant<int, C4>{});
  else if (uc0 == 8) fold_batch(std::integral_constant<int, C8>{});
  else if (uc0 == U) fold_batch(std::integral_constant<int, U>{});
  for (int base = ATT_NW * PP * U; base < ncache; base += ATT_NW * PP * U) {  // further batches: always the full slot class
    if (slot_class(base) == 0) continue;
    issue(base, std::integral_constant<int, U>{});
    fold_batch(std::integral_constant<int, U>{});
  }
  if (owner && wave == ATT_NW - 1) {
    // the new position (the last wave has the fewest cache rows): its rotated k and its v (loaded and rotated at the top)
    // go into the cache (one of the workgroups that share the kv head writes) and are folded from registers
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (sub == 0 && zpart == 0) {
        *reinterpret_cast<f32x4*>(a.kc + (size_t)pos * kv_dim + hoff + 64 * v + 4 * ln) = kn[v];
        *reinterpret_cast<f32x4*>(a.vc + (size_t)pos * kv_dim + hoff + 64 * v + 4 * ln) = vn[v];
      }
    }
    fold(kn, vn, sub == 0 ? 1u : 0u, std::integral_constant<int, 1>{});
  }

  // ---- the NR row groups' states -> LDS records; weights exp(m_r - M) per (row group, head); the split's partial ------
  // (merging the row groups of a wave in registers first costs 2 x 10 cross-row exchanges per head and lane: more than
  //  the whole fold of a short split)
  const int rg = wave * PP + sub;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* rec = part + (size_t)(rg * G + g) * RECL;
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(rec + 64 * v + 4 * ln) = acc[g][v];
    if (ln == 0) {
      wts[g * NR + rg] = m[g];
      lls[g * NR + rg] = l[g];
    }
  }
  __syncthreads();
  static_assert(NR == 32, "a head's 32 row-group states are read as eight float4");
  const bool merge = a.done_cnt != nullptr && active >= TL_ATTN_MERGE_MIN;
  // ws[head][split] = {out[HD] (unnormalised), max, sum}: four lanes share an output quad, each sums a quarter of the
  // row groups (in a fixed order), then the quad is summed with two DPP steps. Every thread takes the split-wide max of its
  // head from the 32 row-group maxima itself (eight broadcast LDS reads) and the weights exp(m_r - M) of its own eight row
  // groups: a phase of its own for them (32 lanes per head, two reductions, one more barrier) was 0.3 us of every launch.
  for (int i = tid; i < G * HD; i += ATT_THREADS) {
    const int rq = i & 3, d4 = (i >> 2) % (HD / 4), g = i / HD;
    float M = -INFINITY;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(wts + g * NR + 4 * q);
      M = fmaxf(fmaxf(M, fmaxf(t[0], t[1])), fmaxf(t[2], t[3]));
    }
    // a wave's 64 threads share one head (HD >= 64): lane l computes the weight of row group l % 32 once and leaves it in
    // the wave's LDS strip; everybody reads back its own eight (LDS operations of a wave execute in order: no barrier).
    // Eight exps per thread instead were ~240 instructions in every wave of the workgroup.
    static_assert(HD % 64 == 0, "one head per wave in the output loop");
    wl[wave * NR + (lane & 31)] = safe_exp_diff(wts[g * NR + (lane & 31)], M);
    __builtin_amdgcn_wave_barrier();
    f32x4 wv[2];
    wv[0] = *reinterpret_cast<const f32x4*>(wl + wave * NR + rq * (NR / 4));
    wv[1] = *reinterpret_cast<const f32x4*>(wl + wave * NR + rq * (NR / 4) + 4);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
#pragma unroll
    for (int j = 0; j < NR / 4; ++j) {
      const int r = rq * (NR / 4) + j;
      const f32x4 x = *reinterpret_cast<const f32x4*>(part + (size_t)(r * G + g) * RECL + 4 * d4);
      const float w = wv[j >> 2][j & 3];
      L = fmaf(lls[g * NR + r], w, L);
      o[0] = fmaf(x[0], w, o[0]); o[1] = fmaf(x[1], w, o[1]); o[2] = fmaf(x[2], w, o[2]); o[3] = fmaf(x[3], w, o[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = dpp_add<0x4E, 0xf>(dpp_add<0xB1, 0xf>(o[e]));
    L = dpp_add<0x4E, 0xf>(dpp_add<0xB1, 0xf>(L));
    if (rq == 0) {
      float* rec = a.ws + ((size_t)(head0 + g) * TL_ATTN_MAX_SPLITS + split) * RECL;
      if (merge || SC1OUT) {
        // agent-scope stores (written through): the merging workgroup may sit on another XCD, behind another L2
#pragma unroll
        for (int e = 0; e < 4; ++e) st_agent(rec + 4 * d4 + e, o[e]);
        if (d4 == 0) {
          st_agent(rec + HD, M);
          st_agent(rec + HD + 1, L);
        }
      } else {
        *reinterpret_cast<f32x4*>(rec + 4 * d4) = o;
        if (d4 == 0) {
          rec[HD] = M;
          rec[HD + 1] = L;
        }
      }
    }
  }
  if (!merge) return;  // the consumer combines the records itself

  // ---- many splits: the LAST workgroup of this head group to get here merges the group's records into record 0, so
  //      that the consumer (every workgroup of the attn_output GEMV re-reads the partials) finds one record per head.
  //      Hand-off: written-through record stores, vmcnt(0), barrier, one relaxed agent-scope counter add per workgroup;
  //      the last one reads the records with agent-scope loads. The counter is left at zero for the next launch.
  //      What this rests on: it is the form MI355X_MICROARCH.md lists as measured valid on gfx950 without fences ("Valid
  //      forms", table row 1: every stored byte sc1 and drained by its wave before the workgroup barrier, ONE lane adds to
  //      ONE unsharded counter, the workgroup whose add came last -- told by the returned value -- loads with sc1 loads after
  //      its add has returned, its other waves after a barrier that lane joins). It is a property of this chip's caches, not
  //      of the C++ memory model: an agent-scope release / acquire pair instead (buffer_wbl2 + buffer_inv, ~1.7 us each)
  //      would add ~3.4 us to every merging launch (32 per step at long context, ~5 % of an 8k step). The compiler cannot
  //      reorder across it: the asm wait carries a memory clobber and __syncthreads() orders the atomics around it.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's record stores have been acknowledged
  __syncthreads();
  if (tid == 0) {
    unsigned* c = a.done_cnt + kvh * a.zparts + zpart;
    const unsigned before = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = before + 1u == (unsigned)active;
    if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[32] = last ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[32] == 0.f) return;
  static_assert(G * HD <= ATT_THREADS && G * TL_ATTN_MAX_SPLITS <= ATT_THREADS, "one element per thread in the merge");
  const bool mine = tid < G * HD;
  const int g = mine ? tid / HD : 0, d = tid % HD;
  float* rec0 = a.ws + (size_t)(head0 + g) * TL_ATTN_MAX_SPLITS * RECL;
  float o[TL_ATTN_MAX_SPLITS];
  if (mine) {
#pragma unroll
    for (int sp = 0; sp < TL_ATTN_MAX_SPLITS; ++sp)  // all records in flight together (clamped, unconditional)
      o[sp] = ld_agent(rec0 + (size_t)(sp < active ? sp : active - 1) * RECL + d);
  }
  // meanwhile: (max, sum) of every record once, by 32 consecutive lanes per head -> weights in LDS (wts is free again)
  if (tid < G * TL_ATTN_MAX_SPLITS) {
    const int gg = tid / TL_ATTN_MAX_SPLITS, sp = tid % TL_ATTN_MAX_SPLITS;
    const float* rec = a.ws + ((size_t)(head0 + gg) * TL_ATTN_MAX_SPLITS + (sp < active ? sp : active - 1)) * RECL;
    const float ms = sp < active ? ld_agent(rec + HD) : -INFINITY;
    const float ls = ld_agent(rec + HD + 1);
    static_assert(TL_ATTN_MAX_SPLITS == 32, "the reductions below run over 32 consecutive lanes per head");
    const float M = max32(ms);
    const float w = sp < active ? expf(ms - M) : 0.f;
    wts[tid] = w;
    const float L = sum32(ls * w);
    if (sp == 0) {
      red[gg] = M;
      red[8 + gg] = L;
    }
  }
  __syncthreads();  // weights ready; record 0 has been read (into registers) by everyone before it is overwritten
  if (mine) {
    float O = 0.f;
#pragma unroll
    for (int sp = 0; sp < TL_ATTN_MAX_SPLITS; ++sp) O = fmaf(o[sp], wts[g * TL_ATTN_MAX_SPLITS + sp], O);  // split order
    rec0[d] = O;
    if (d == 0) {
      rec0[HD] = red[g];
      rec0[HD + 1] = red[8 + g];
    }
  }
}


// LDS floats of one attn_split_body workgroup
template <int HD, int G>
constexpr size_t attn_split_lds_floats() {
  return 64 + 2 * (size_t)ATT_NW * 4 * G + (size_t)ATT_NW * ATT_NW * 4 + (size_t)ATT_NW * 4 * G * (HD + 4);
}

}  // namespace tl_attn
