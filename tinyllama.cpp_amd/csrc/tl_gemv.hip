// tl_gemv.hip — block-dequant GEMV over the plane layout (tl_internal.h): y[r] = sum_c deq(W[r][c]) * x[c].
//
// What it replaces: the reference never multiplies by quantized weights on a GPU — it expands every layer
// to FP32 on the host and calls cublasSgemm with N=1 (weight_management.cpp:199-435, cuda_kernels.cu:291-314);
// its CPU path is dequantize-block + Kahan-double dot (utils.cpp:293-397, 784-888). The numerics target is the
// CPU path; this kernel accumulates in FP32 (tolerance stated in tests/test_gpu_parity.py).
//
// Shape of the kernel (HBM-bound streaming; no MFMA on purpose):
//   - ONE workgroup of up to 16 waves per CU, every workgroup resident at once (grid <= #CUs): the input vector is
//     staged into LDS once per CU (optionally with the RMSNorm of utils.cpp:648-673 fused in), in 36-float strided
//     chunks: 32 values + the two 16-element partial sums that fold the "-min" / "-32" / "-128" offsets out of the
//     inner loop. The 16 B/lane LDS reads are bank-conflict free;
//   - a lane owns one 32-weight chunk of a 2048-column span: one 16 B/lane load = 1 KiB contiguous per wave, plus
//     the 16 B scale header of its super-block (8 lanes share one);
//   - a wave's step is TWO weight streams x one span sharing the same x chunk: two consecutive rows, or the same
//     row of ffn_gate and ffn_up. Steps are register double-buffered: the loads of step i+1 are issued before step i
//     is consumed, and a wave's first loads go out before x is staged, so the HBM round trip overlaps the prologue;
//   - the block format is a COMPILE-TIME parameter of a wave's loop (a type switch inside the loop makes the compiler
//     wait for every load right where it is issued). A launch may mix two formats (attn_q/attn_k Q4_K + attn_v Q6_K in
//     the Q4_K_M mix): the waves of a workgroup are split between the two classes in proportion to their bytes;
//   - row groups are dealt round-robin over all waves of the chip (consecutive groups on different CUs);
//   - weights are loaded non-temporal (each byte is read once per token);
//   - nibble/byte unpack is v_and/v_lshr + v_cvt_f32_ubyteN, FP32 FMA chains, DPP wave reduction;
//   - epilogues: plain store, residual add, or SwiGLU of a gate/up row pair.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "tl_internal.h"

#include "tl_chunk.h"
#include "tl_attn_body.h"

// the vector-side arguments of a launch (the segments travel as separate kernel parameters)
struct GemvVec {
  const float* x;
  const float* norm_w;
  const float* resid;
  float eps;
  int cols;
  unsigned long long* stamps;  // diagnostic instantiation only: 10 u64 per wave (tl_debug_gemv_stamps)
  // GEMV_ATTN_OUT_COMBINE: x is the attention output still in split partials (tl_attn.hip), combined while staging
  const float* attn_ws;
  const int* attn_pos;
  int attn_hd;
  int attn_merged;
  // the launch geometry as EXPLICIT arguments: blockDim / gridDim live in the hidden argument block, which hipcc reads
  // with separate scalar loads where first used: two more dependent round trips (~0.25 us each, cold scalar cache)
  // between kernel entry and a wave's first weight load
  int nthr, grid;
  const TlPeerDev* peer;  // PEER instantiations only: the exchange this launch carries (tl_internal.h)
  int peer_gpw;           // ... and the row groups a wave has at most: its slice of the LDS stash holds 2 floats for each
};

// ---- chained launches (tl_launch_chain below): several GEMVs of the decode step inside ONE launch ---------------------
// A chain launch is a 2-D grid (workgroups of a task, tasks): the workgroups of task j+1 are dispatched behind those of
// task j and take a CU as soon as one has room, issue their first weight loads (constants) and only then wait for task j,
// so that kernel boundary, argument fetch and the first HBM round trip of a launch overlap the tail of its predecessor.
// Hand-off (MI355X_MICROARCH.md, "Valid forms", table row 1): every activation word that crosses a task edge is stored
// with an agent-scope (sc1, write-through) store and loaded with sc1 loads; every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane adds 1 to the task's arrival counter (8 shards on
// lines of their own: 256 arrivals on one word serialise at ~12 ns each); the consumer's wave 0 polls the shards with sc1
// loads (bounded by a deadline: a hand-off that never comes becomes an error word, not a hung GPU) and releases the other
// waves through a workgroup barrier. Counters only grow: the target is (generation + 1) x arrivals per step, with the
// generation word advanced once per step by a kernel of its own (no reset, no dependence on stale zeros).
// Deadlock freedom rests on in-order dispatch inside one grid (x fastest, then y): a waiting workgroup can only hold a CU
// if every workgroup of the task it waits for was placed before it.
constexpr int CHAIN_SHARDS = 8;
constexpr int CHAIN_SHARD_WORDS = 32;                                // 128 B apart
constexpr int CHAIN_TASK_WORDS = CHAIN_SHARDS * CHAIN_SHARD_WORDS;  // counter words per task
constexpr unsigned long long CHAIN_TIMEOUT_TICKS = 200000000ull;    // 2 s of the 100 MHz realtime counter
struct ChainWait {
  const unsigned* cnt;  // the predecessor task's arrival counters (CHAIN_SHARDS shards); nullptr: nothing to wait for
  unsigned gen1;        // generation + 1
  int pred_nwg;         // workgroups of the predecessor task
  unsigned* err;        // sticky error word (a poll ran into its deadline)
  // diagnostic run only (tl_chain_set_stamps): 8 realtime stamps (100 MHz) of this workgroup, written by thread 0:
  // [0] entry, [1] first weight loads issued, [2] predecessor seen, [3] x staged, [4] last loads issued, [5] all stores
  // acknowledged and the workgroup met, [6] arrival added
  unsigned long long* st;
  // != nullptr: this launch carries its layer's attention (gemv_attn_kernel below): the wait is for that, not for a task
  const struct FusedAttn* fa;
  float* attn_smem;  // LDS of the attention body (behind the x image)
  int bid, grid;
};
// The attention of a layer folded into the layer's attn_output launch. Workgroup b < n_kv_heads x zparts x active splits
// runs the split-attention body for (split b / (nkv zparts), kv head, part) while every wave's first weight loads are
// already in flight, stores its partial records written-through and arrives at `count`; the last arrival resets the
// counter and publishes the launch's epoch (CHAIN_SHARDS copies on lines of their own), which wave 0 of EVERY workgroup
// polls for before the x staging combines the records (sc1 loads). epoch = generation x layers + layer + 1: a value no
// earlier launch has published, so nothing needs clearing; the generation word is advanced once per step by a kernel of
// its own. All workgroups of the launch are resident (one per CU), so every poller's producer is running or done.
struct FusedAttn {
  AttnArgs at;
  unsigned* count;
  unsigned* epoch;
  const unsigned* gen;
  unsigned tag, tags_per_gen;
  unsigned* err;
  // diagnostic run only (tl_chain_set_stamps): 8 realtime stamps per workgroup, written by thread 0: [0] entry, [1] first
  // weight loads issued, [2] all attention records published, [3] x staged, [4] last loads issued, [7] thread 0 done; for
  // the workgroups with an attention task: [5] its records acknowledged, [6] its arrival added
  unsigned long long* st;
};
__device__ __forceinline__ void chain_stamp(const ChainWait* cw, int k) {
  if (cw->st != nullptr && threadIdx.x == 0) cw->st[k] = __builtin_amdgcn_s_memrealtime();
}
template <int HD, int G>
__device__ __forceinline__ void fused_attention_phase(const ChainWait& cw, int wave, int lane) {
  const FusedAttn& f = *cw.fa;
  const int pos = *f.at.pos_dev;  // (uniform: a scalar load)
  const int len = pos + 1;
  const int SL = tl_attn_split_len(len), active = (len + SL - 1) / SL;
  const int per = f.at.nkv * f.at.zparts;
  const int ntasks = (pos >= 0 && pos < f.at.max_seq) ? per * active : 0;
  if (ntasks == 0) return;  // (the position is out of range: the step's other kernels ignore it too)
  for (int t = cw.bid; t < ntasks; t += cw.grid) {
    const int split = t / per, r = t - split * per, z = r / f.at.nkv, kvh = r - z * f.at.nkv;
    tl_attn::attn_split_body<HD, G, true, 2, true>(f.at, kvh, split, z, cw.attn_smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have been acknowledged
    __syncthreads();                                  // ... every wave's; the body's LDS is free again
  }
  const unsigned target = *f.gen * f.tags_per_gen + f.tag;
  if (cw.bid < ntasks) chain_stamp(&cw, 5);
  if (cw.bid < ntasks && threadIdx.x == 0) {
    const unsigned arrivals = (unsigned)(ntasks < cw.grid ? ntasks : cw.grid);
    const unsigned before = __hip_atomic_fetch_add(f.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == arrivals) {
      __hip_atomic_store(f.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int c = 0; c < CHAIN_SHARDS; ++c)
        __hip_atomic_store(f.epoch + c * CHAIN_SHARD_WORDS, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (cw.st != nullptr) cw.st[6] = __builtin_amdgcn_s_memrealtime();
  }
  if (wave == 0) {
    const unsigned* p = f.epoch + (cw.bid & (CHAIN_SHARDS - 1)) * CHAIN_SHARD_WORDS;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + CHAIN_TIMEOUT_TICKS;
    for (unsigned it = 1;; ++it) {
      const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane(v) == target) break;
      if ((it & 255u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {
        if (lane == 0) atomicOr(f.err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void chain_wait(const ChainWait& cw, int wave, int lane) {
  if (cw.cnt == nullptr) return;  // (uniform)
  if (wave == 0) {
    const int sh = lane & (CHAIN_SHARDS - 1);
    const unsigned want = cw.gen1 * (unsigned)((cw.pred_nwg + CHAIN_SHARDS - 1 - sh) / CHAIN_SHARDS);
    const unsigned* p = cw.cnt + sh * CHAIN_SHARD_WORDS;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + CHAIN_TIMEOUT_TICKS;
    for (unsigned it = 1;; ++it) {
      const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dword sc1
      const bool ok = (int)(v - want) >= 0;
      if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
      if ((it & 255u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {
        if (lane == 0) atomicOr(cw.err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
// activation loads / stores across a task edge: sc1 (agent scope) in a chain launch, the ordinary forms otherwise
template <bool CHAIN>
__device__ __forceinline__ f32x4 ldx4(const void* base, unsigned byte_off) {
  if constexpr (CHAIN) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 1 << 4));  // sc1
  } else {
    return ldc<f32x4>(base, byte_off);
  }
}
template <bool CHAIN>
__device__ __forceinline__ float ldx1(const float* p) {
  if constexpr (CHAIN) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return ldg_f32(p);
}
template <bool CHAIN>
__device__ __forceinline__ void stx1(float* p, float v) {
  if constexpr (CHAIN) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else stg_f32(p, v);
}

// ---- x staging: (optional RMSNorm, utils.cpp:648-673) -> strided LDS image + per-chunk half sums --------------
// Split in two so that the x loads are the FIRST loads a wave issues: loads return in order, and staging must not
// wait behind the first weight loads (an HBM round trip) that are issued in between.
struct XRegs {
  f32x4 v[4];  // the first four float4 of a thread (covers cols <= 16 * blockDim)
  f32x4 g[2];  // norm weights of the first two (norm launches have cols = hidden_size <= 8 * blockDim in practice)
  // XATTN: the first XA_CHUNK split partials of the thread's first float4: out, (max, sum); and the active split count
  f32x4 ao[6];
  float am[6], al[6];
  int active;
};

constexpr int XA_CHUNK = 6;
// issue the loads of splits [s0, s0 + XA_CHUNK) (clamped to the active ones) of float4 index i
template <bool CHAIN = false>
__device__ __forceinline__ void xattn_issue(const GemvVec& a, int i, int s0, int active, f32x4* ao, float* am, float* al) {
  const int hd = a.attn_hd, recl = hd + 4;
  const int e = i << 2, h = e / hd, d = e - h * hd;
  const float* rec0 = a.attn_ws + (size_t)h * TL_ATTN_MAX_SPLITS * recl;
#pragma unroll
  for (int k = 0; k < XA_CHUNK; ++k) {
    const int s = s0 + k < active ? s0 + k : active - 1;  // unconditional loads (counted waits), clamped
    const float* rec = rec0 + (size_t)s * recl;
    if constexpr (CHAIN) {  // the records may have been written in THIS launch (attention folded in): sc1 loads
      const unsigned ro = (unsigned)((h * TL_ATTN_MAX_SPLITS + s) * recl);  // (wave-uniform base, per-lane offset)
      ao[k] = ldx4<true>(a.attn_ws, (ro + (unsigned)d) * 4u);
      am[k] = ldx1<true>(a.attn_ws + ro + hd);
      al[k] = ldx1<true>(a.attn_ws + ro + hd + 1);
    } else {
      ao[k] = *reinterpret_cast<const f32x4*>(rec + d);
      am[k] = rec[hd];
      al[k] = rec[hd + 1];
    }
  }
}
// fold a chunk into the running (M, L, O) of a float4
__device__ __forceinline__ void xattn_fold(int s0, int active, const f32x4* ao, const float* am, const float* al, float& M,
                                           float& L, f32x4& O) {
  float mx = M;
#pragma unroll
  for (int k = 0; k < XA_CHUNK; ++k) mx = fmaxf(mx, s0 + k < active ? am[k] : -INFINITY);
  const float corr = M == -INFINITY ? 0.f : expf(M - mx);
  L *= corr;
  O[0] *= corr; O[1] *= corr; O[2] *= corr; O[3] *= corr;
#pragma unroll
  for (int k = 0; k < XA_CHUNK; ++k) {
    const float w = s0 + k < active ? expf(am[k] - mx) : 0.f;
    L = fmaf(al[k], w, L);
    O[0] = fmaf(ao[k][0], w, O[0]); O[1] = fmaf(ao[k][1], w, O[1]);
    O[2] = fmaf(ao[k][2], w, O[2]); O[3] = fmaf(ao[k][3], w, O[3]);
  }
  M = mx;
}
// Unconditional (index-clamped) loads: a load under a branch makes the compiler wait with vmcnt(0) at the join,
// which would also wait for the weight loads issued after these.
template <bool XATTN, bool CHAIN = false>
__device__ __forceinline__ void stage_x_issue(const GemvVec& a, XRegs& r) {
  const int n4 = a.cols >> 2;
  if (XATTN) {
    int pos;  // scalar load: its own counter, so the partial loads below need not wait behind anything
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pos) : "s"(a.attn_pos) : "memory");
    r.active = tl_attn_records(pos + 1, a.attn_merged != 0);
    const int i = threadIdx.x < (unsigned)n4 ? threadIdx.x : n4 - 1;
    xattn_issue<CHAIN>(a, i, 0, r.active, r.ao, r.am, r.al);
    return;
  }
  const void* wbase = a.norm_w ? a.norm_w : a.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + k * a.nthr;
    r.v[k] = ldx4<CHAIN>(a.x, (unsigned)(i < n4 ? i : n4 - 1) * 16u);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = threadIdx.x + k * a.nthr;
    r.g[k] = ldc<f32x4>(wbase, (unsigned)(i < n4 ? i : n4 - 1) * 16u);
  }
}
__device__ __forceinline__ float ssq4(f32x4 t) { return (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]); }
template <bool XATTN, bool CHAIN = false>
__device__ __forceinline__ void stage_x_finish(const GemvVec& a, const XRegs& r, float* smem) {
  float* xs = smem + XS_SCRATCH;
  const int tid = threadIdx.x, nthr = a.nthr;
  const int n4 = a.cols >> 2;
  if (XATTN) {
    // x = combined attention output: out = sum_s o_s e^(m_s - M) / (sum_s l_s e^(m_s - M) + 1e-9), utils.cpp:692
    for (int i = tid; i < n4; i += nthr) {
      float M = -INFINITY, L = 0.f;
      f32x4 O = {0.f, 0.f, 0.f, 0.f};
      if (i == tid) {
        xattn_fold(0, r.active, r.ao, r.am, r.al, M, L, O);
      } else {
        f32x4 ao[XA_CHUNK];
        float am[XA_CHUNK], al[XA_CHUNK];
        xattn_issue<CHAIN>(a, i, 0, r.active, ao, am, al);
        xattn_fold(0, r.active, ao, am, al, M, L, O);
      }
      for (int s0 = XA_CHUNK; s0 < r.active; s0 += XA_CHUNK) {
        f32x4 ao[XA_CHUNK];
        float am[XA_CHUNK], al[XA_CHUNK];
        xattn_issue<CHAIN>(a, i, s0, r.active, ao, am, al);
        xattn_fold(s0, r.active, ao, am, al, M, L, O);
      }
      const float inv = 1.0f / (L + 1e-9f);
      f32x4 t = {O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv};
      const int idx = i << 2;
      float* dst = xs + (idx >> 5) * XS_STRIDE;
      *reinterpret_cast<f32x4*>(dst + (idx & 31)) = t;
      float sm4 = (t[0] + t[1]) + (t[2] + t[3]);
      sm4 = dpp_add<0x4E, 0xf>(dpp_add<0xB1, 0xf>(sm4));  // quad sum in every lane of the quad (DPP, no LDS path)
      if ((i & 3) == 0) dst[32 + ((i >> 2) & 1)] = sm4;
    }
    __syncthreads();
    return;
  }
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(a.norm_w);
  float nf = 1.0f;
  if (a.norm_w) {
    float ssq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) ssq += (tid + k * nthr < n4) ? ssq4(r.v[k]) : 0.f;
    for (int i = tid + 4 * nthr; i < n4; i += nthr) ssq += ssq4(CHAIN ? ldx4<CHAIN>(a.x, (unsigned)i * 16u) : x4[i]);
    ssq = wave_sum_lane63(ssq);  // DPP only (the xor butterfly is six dependent LDS-path shuffles here)
    if ((tid & 63) == 63) smem[tid >> 6] = ssq;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) tot += smem[w];
    const float e = a.eps < 1e-5f ? 1e-5f : a.eps;  // SAFE_MAX(eps, MIN_NORM_EPS)
    nf = 1.0f / sqrtf(tot / (float)a.cols + e);
  }
  auto put = [&](int i, f32x4 t, f32x4 g) {
    if (a.norm_w) {
      t[0] = t[0] * nf * g[0]; t[1] = t[1] * nf * g[1]; t[2] = t[2] * nf * g[2]; t[3] = t[3] * nf * g[3];
    }
    const int idx = i << 2;
    float* dst = xs + (idx >> 5) * XS_STRIDE;
    *reinterpret_cast<f32x4*>(dst + (idx & 31)) = t;
    // 4 consecutive threads cover 16 consecutive x: their sum is one of the chunk's two half sums
    float s = (t[0] + t[1]) + (t[2] + t[3]);
    s = dpp_add<0x4E, 0xf>(dpp_add<0xB1, 0xf>(s));  // quad sum in every lane of the quad (DPP, no LDS path)
    if ((i & 3) == 0) dst[32 + ((i >> 2) & 1)] = s;
  };
  // n4 is a multiple of 8 and blockDim of 64, so quads of threads are all-in or all-out of every branch
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + k * nthr;
    if (i < n4) put(i, r.v[k], k < 2 ? r.g[k] : (a.norm_w ? w4[i] : one));
  }
  for (int i = tid + 4 * nthr; i < n4; i += nthr) put(i, CHAIN ? ldx4<CHAIN>(a.x, (unsigned)i * 16u) : x4[i], a.norm_w ? w4[i] : one);
  __syncthreads();
}

// ---- the kernel ------------------------------------------------------------------------------------------------
// The segments of one format class as plain scalars (no array: a dynamically indexed by-value kernel argument is
// copied to scratch and drags every address computation into VGPRs).
struct Segs {
  QW w0, w1, w2;
  float *o0, *o1, *o2;
  int t0, t1;  // row groups in segments 0 and 1
};

// row group `rg` of a class: rows (2*lt, 2*lt+1) of one of its segments, or row lt of the gate/up pair
template <bool PAIR>
__device__ __forceinline__ void group_of(const Segs& g, int rg, QW& wa, QW& wb, int& ra, int& rb, float*& out) {
  if (PAIR) {  // wa / wb / out were set once by the caller and never change
    ra = rb = rg;
    return;
  }
  int seg = 0, lt = rg;
  if (lt >= g.t0) { lt -= g.t0; seg = 1; }
  if (seg == 1 && lt >= g.t1) { lt -= g.t1; seg = 2; }
  // 3-way select by mask arithmetic: a select chain over kernel arguments is turned into a scratch-memory table
  const unsigned long long m0 = seg == 0 ? ~0ull : 0ull, m1 = seg == 1 ? ~0ull : 0ull, m2 = seg == 2 ? ~0ull : 0ull;
  auto pick = [&](const void* x0, const void* x1, const void* x2) {
    return reinterpret_cast<const uint8_t*>(((unsigned long long)x0 & m0) | ((unsigned long long)x1 & m1) |
                                            ((unsigned long long)x2 & m2));
  };
  wa.p0 = pick(g.w0.p0, g.w1.p0, g.w2.p0);
  wa.p1 = pick(g.w0.p1, g.w1.p1, g.w2.p1);
  wa.p2 = pick(g.w0.p2, g.w1.p2, g.w2.p2);
  wa.p3 = pick(g.w0.p3, g.w1.p3, g.w2.p3);
  wa.rows = (int)(((unsigned)g.w0.rows & (unsigned)m0) | ((unsigned)g.w1.rows & (unsigned)m1) |
                  ((unsigned)g.w2.rows & (unsigned)m2));
  out = reinterpret_cast<float*>(const_cast<uint8_t*>(pick(g.o0, g.o1, g.o2)));
  wb.p0 = wa.p0;
  wb.p1 = wa.p1;
  wb.p2 = wa.p2;
  wb.p3 = wa.p3;
  wb.rows = wa.rows;
  ra = 2 * lt;
  rb = ra + 1 < wa.rows ? ra + 1 : wa.rows - 1;  // odd tail: computed twice, stored once
}

// ---- the exchange folded into the epilogue (PEER instantiations; protocol and reasons in tl_peer.hip) -----------------
typedef unsigned long long pu64;
typedef pu64 __attribute__((address_space(1))) * peer_gptr;
__device__ __forceinline__ void peer_store(pu64* p, pu64 w) {
  __hip_atomic_store((peer_gptr)(uintptr_t)p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ pu64 peer_load(const pu64* p) {
  return __hip_atomic_load((peer_gptr)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// row `row` of this rank, value v, epoch e: one 8-byte word into slot [e & 1][rank][row] of every rank's area
// (ALL: of this rank's own area too -- the all-reduce then reads every addend the same way)
template <bool ALL>
__device__ __forceinline__ void peer_push(const TlPeerDev* pd, unsigned e, int row, float v) {
  const int tp = pd->tp, rank = pd->rank, n = pd->n;
  const size_t at = ((size_t)(e & 1u) * (size_t)tp + (size_t)rank) * (size_t)n + (size_t)row;
  const pu64 w = ((pu64)e << 32) | (pu64)__float_as_uint(v);
#pragma unroll
  for (int p = 0; p < TL_PEER_MAX_RANKS; ++p)
    if (p < tp && (ALL || p != rank)) peer_store(pd->rx[p] + at, w);
}
// End of a wave's work: its finished rows wait in the wave's LDS stash (2 floats per row group, in the order the wave
// worked through first, first + stride, ...: rows 2 rg and 2 rg + 1 of a single-matrix launch). Pushing them from the
// epilogue of every row group instead measured much slower where a wave has many row groups (lm_head): the pushes are
// uncached stores, stores retire through the same in-order counter as the loads, and the step loop's counted waits then
// sit behind a store's slow acknowledgement once per row group. Here: one row per lane, 32 row groups per pass: push the
// row's word to every rank, wait for every rank's word of the row (bounded by the deadline), then
// REDUCE: out[row] = sum in rank order 0 .. tp-1 (every rank computes the same bits); else: out[(q - rank) n + row] = rank
// q's value for q != rank (out points at this rank's rows of the gathered vector, which the epilogue stored directly).
template <bool REDUCE>
__device__ __forceinline__ void peer_collect(const TlPeerDev* pd, unsigned e, float* out, int rows, int first, int stride,
                                             int ngroups, const float* stash, int lane) {
  const int tp = pd->tp, rank = pd->rank, n = pd->n;
  const pu64* mine = pd->rx[0];
#pragma unroll
  for (int q = 1; q < TL_PEER_MAX_RANKS; ++q) mine = q == rank ? pd->rx[q] : mine;
  mine += (size_t)(e & 1u) * (size_t)tp * (size_t)n;
  const pu64 deadline = __builtin_amdgcn_s_memrealtime() + pd->timeout_ticks;
  bool ok = true;
  for (int j0 = 0; j0 < ngroups; j0 += 32) {
    const int j = j0 + (lane >> 1);
    const int row = 2 * (first + j * stride) + (lane & 1);
    if (j < ngroups && row < rows) {
      peer_push<REDUCE>(pd, e, row, stash[2 * j + (lane & 1)]);
      const pu64* at = mine + row;
      pu64 w[TL_PEER_MAX_RANKS];
#pragma unroll
      for (int q = 0; q < TL_PEER_MAX_RANKS; ++q)  // the first round of polls of all ranks in flight together
        w[q] = (q < tp && (REDUCE || q != rank)) ? peer_load(at + (size_t)q * n) : 0ull;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < TL_PEER_MAX_RANKS; ++q) {
        if (q >= tp || (!REDUCE && q == rank)) continue;
        pu64 x = w[q];
        for (unsigned it = 1; (unsigned)(x >> 32) != e; ++it) {
          if ((it & 63u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          x = peer_load(at + (size_t)q * n);
        }
        const float v = __uint_as_float((unsigned)x);
        if (REDUCE) acc += v;
        else stg_f32(out + ((long)(q - rank) * n + row), v);
      }
      if (REDUCE) stg_f32(out + row, acc);
    }
  }
  if (!ok) atomicOr(pd->err, 1u);
}

template <int TYPE>
struct Step {
  Chunk<TYPE> c0, c1;
  float r0, r1;  // residual values of the two rows (RESID launches only)
};

// what the consumer of a step needs to know about it (wave-uniform)
struct StepMeta {
  int sp, ra, rows;
  float* out;
};

// all row groups rg = first, first + stride, ... < total of one format class, for one wave.
// Two cursors: the ISSUE cursor (irg, isp, wa/wb/ra/rb/out) runs one step ahead of the step being consumed, whose
// facts travel in a StepMeta next to its register buffer. Every consume() directly follows an issue() in straight-line
// code (the loop is peeled instead of making the prefetch conditional), so the compiler can wait with a COUNTED
// vmcnt that leaves the prefetch in flight.
// RESID: out[r] = resid[r] + y[r]. The two residual values are loaded at ISSUE time, ahead of the step's weight loads:
// loads complete in order, so a load in the epilogue would drain the whole prefetch queue once per row group.
// CHAIN (a task of a chain launch): x and the first residual values are loaded only AFTER the wait for the predecessor
// task, which follows the first weight loads; activation loads and stores take their sc1 forms.
// AHD > 0 (with CHAIN): the launch carries its layer's attention (fused_attention_phase<AHD, AG>) in place of the wait.
template <int TYPE, bool PAIR, bool RESID, bool XATTN, bool STAMP, bool PEER = false, bool CHAIN = false, int AHD = 0,
          int AG = 0>
__device__ __forceinline__ void run_class(const Segs& g, const GemvVec& a, XRegs& xr, float* smem, int first,
                                          int stride, int total, int lane, unsigned long long* ts, unsigned pe = 0,
                                          int peer_wave = 0, const ChainWait* cw = nullptr) {
  const float* xs = smem + XS_SCRATCH;
  const int nchunks = a.cols >> 5;
  const int nspans = (nchunks + 63) >> 6;
  // PEER: this wave's slice of the stash behind the x image, and the row groups it has finished
  float* stash = smem + XS_SCRATCH + nchunks * XS_STRIDE + 2 * peer_wave * a.peer_gpw;
  int ndone = 0;
  if (first >= total) {  // a wave without work only helps staging x
    if (STAMP) ts[3] = __builtin_readcyclecounter();
    if constexpr (CHAIN) {
      if constexpr (AHD > 0) fused_attention_phase<AHD, AG>(*cw, peer_wave, lane);
      else chain_wait(*cw, peer_wave, lane);
      stage_x_issue<XATTN, CHAIN>(a, xr);
    }
    stage_x_finish<XATTN, CHAIN>(a, xr, smem);
    if (STAMP) ts[4] = __builtin_readcyclecounter();
    return;
  }
  int irg = first, isp = 0;
  QW wa = g.w0, wb = PAIR ? g.w1 : g.w0;
  int ra = 0, rb = 0;
  float* out = g.o0;
  group_of<PAIR>(g, irg, wa, wb, ra, rb, out);
  float acc0 = 0.f, acc1 = 0.f;
  bool with_resid = !CHAIN;  // (CHAIN: the first step's residual values follow the wait)

  auto issue = [&](Step<TYPE>& st, StepMeta& m) {
    m.sp = isp;
    m.ra = ra;
    m.rows = wa.rows;
    m.out = out;
    const int chunk = (isp << 6) + lane;
    const unsigned cc = chunk < nchunks ? chunk : nchunks - 1;
    if (RESID && (!CHAIN || with_resid)) {
      st.r0 = ldx1<CHAIN>(a.resid + ra);
      st.r1 = ldx1<CHAIN>(a.resid + rb);
    }
    st.c0.load(wa, ra, cc);
    st.c1.load(wb, rb, cc);
    if (++isp == nspans) {  // advance the issue cursor (scalar work, overlaps the loads)
      isp = 0;
      irg += stride;
      if (irg < total) group_of<PAIR>(g, irg, wa, wb, ra, rb, out);
    }
  };
  auto consume = [&](const Step<TYPE>& st, const StepMeta& m) {
    const int chunk = (m.sp << 6) + lane;
    const bool act = chunk < nchunks;
    const int cc = act ? chunk : nchunks - 1;
    float x[32];
    const f32x4* xp = reinterpret_cast<const f32x4*>(xs + cc * XS_STRIDE);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 t = xp[i];
      x[4 * i] = t[0]; x[4 * i + 1] = t[1]; x[4 * i + 2] = t[2]; x[4 * i + 3] = t[3];
    }
    const f32x4 sums = xp[8];
    const float d0 = st.c0.dot(x, sums[0], sums[1], cc & 7);
    const float d1 = st.c1.dot(x, sums[0], sums[1], cc & 7);
    acc0 += act ? d0 : 0.0f;
    acc1 += act ? d1 : 0.0f;
    if (m.sp == nspans - 1) {  // row group complete: reduce, epilogue, store
      const float y0 = wave_sum_lane63(acc0);
      const float y1 = wave_sum_lane63(acc1);
      if (lane == 63) {
        if (PAIR) {
          const float sg = 1.0f / (1.0f + expf(-y0));  // silu_cpu (utils.cpp:700-707), then * up (model.cpp:735)
          stx1<CHAIN>(m.out + m.ra, (y0 * sg) * y1);
        } else if (PEER) {  // into the wave's stash; pushed and collected at the end of the wave's work (peer_collect)
          stash[2 * ndone] = RESID ? st.r0 + y0 : y0;
          stash[2 * ndone + 1] = RESID ? st.r1 + y1 : y1;
          if (!RESID) {
            stg_f32(m.out + m.ra, y0);
            if (m.ra + 1 < m.rows) stg_f32(m.out + m.ra + 1, y1);
          }
        } else {
          stx1<CHAIN>(m.out + m.ra, RESID ? st.r0 + y0 : y0);
          if (m.ra + 1 < m.rows) stx1<CHAIN>(m.out + m.ra + 1, RESID ? st.r1 + y1 : y1);
        }
      }
      acc0 = 0.f;
      acc1 = 0.f;
      if (PEER) ++ndone;
    }
  };

  // Software pipeline of depth one: the loads of step i+1 are in flight while step i is consumed. (Alternating two
  // named buffers instead of the copy makes hipcc insert vmcnt(0) waits in front of unrelated register writes.)
  Step<TYPE> cur, nxt;
  StepMeta mc, mn;
  const int ra0 = ra, rb0 = rb;
  issue(cur, mc);  // the first step's loads go out before x is staged: the HBM round trip overlaps the prologue
  // (Issuing a SECOND step ahead of the staging was tried twice and measured slower both times: in this loop the copy
  //  form then waits for its own fresh loads, -10 %; as a loop-free variant for the launches with two steps per wave,
  //  attn_output and QKV, -1.7 %: the x loads of the waves that start late queue behind twice as many weight loads of
  //  the early ones, and staging ends with a barrier.)
  if (STAMP) ts[3] = __builtin_readcyclecounter();
  if constexpr (CHAIN) {  // the predecessor task's output exists from here on
    chain_stamp(cw, 1);
    if constexpr (AHD > 0) fused_attention_phase<AHD, AG>(*cw, peer_wave, lane);
    else chain_wait(*cw, peer_wave, lane);
    chain_stamp(cw, 2);
    stage_x_issue<XATTN, CHAIN>(a, xr);
    if (RESID) {
      cur.r0 = ldx1<CHAIN>(a.resid + ra0);
      cur.r1 = ldx1<CHAIN>(a.resid + rb0);
    }
    with_resid = true;
  }
  stage_x_finish<XATTN, CHAIN>(a, xr, smem);
  if constexpr (CHAIN) chain_stamp(cw, 3);
  if (STAMP) ts[4] = __builtin_readcyclecounter();
  while (irg < total) {
    issue(nxt, mn);
    consume(cur, mc);
    if (STAMP && ts[5] == 0) ts[5] = __builtin_readcyclecounter();
    cur = nxt;
    mc = mn;
  }
  if (STAMP) ts[6] = __builtin_readcyclecounter();  // every load of this wave has been issued
  if constexpr (CHAIN) chain_stamp(cw, 4);
  consume(cur, mc);
  if (STAMP && ts[5] == 0) ts[5] = __builtin_readcyclecounter();
  if (PEER) peer_collect<RESID>(a.peer, pe, g.o0, g.w0.rows, first, stride, ndone, stash, lane);
}


// ---- flattened steps: column counts that are not a multiple of the 2048-column span ---------------------------------
// With cols = 3072 (Llama-3.2-3B) a row is 96 chunks: spans of 64 leave half the lanes of every second step idle
// (25 % of the load slots); 5632 and 11008 columns waste 8 % and 10 %. Here a row group is RA consecutive rows per
// weight stream (stream 0: rows ra .. ra+RA-1, stream 1: the same rows of the second matrix of a pair, or rows
// ra+RA .. ra+2RA-1), and its RA * nchunks (row, chunk) items are dealt to the lanes as ONE sequence, 64 per step:
// only the last step of a group can be partly empty (none at all when 64 divides RA * nchunks). Within a step the
// lanes sit in at most two consecutive rows (nchunks > 64), so a lane's row is j0 or j0 + 1 and its chunk follows from
// a running offset; the row is a per-lane value here (address arithmetic in VGPRs), which is why this is a variant
// and not the default. Both streams still share one x chunk per lane. 2 * RA accumulators per lane.
template <int TYPE, int RA>
struct StepF {
  Chunk<TYPE> c0, c1;
  float r[2 * RA];  // residual values of the group's rows (RESID launches only)
};
struct StepMetaF {
  int s, j0, off0, ra, rows;
  float* out;
};

template <bool PAIR, int RPG>
__device__ __forceinline__ void group_of_flat(const Segs& g, int rg, QW& wa, QW& wb, int& ra, float*& out) {
  if (PAIR) {  // wa / wb / out were set once by the caller and never change
    ra = rg * RPG;
    return;
  }
  int seg = 0, lt = rg;
  if (lt >= g.t0) { lt -= g.t0; seg = 1; }
  if (seg == 1 && lt >= g.t1) { lt -= g.t1; seg = 2; }
  const unsigned long long m0 = seg == 0 ? ~0ull : 0ull, m1 = seg == 1 ? ~0ull : 0ull, m2 = seg == 2 ? ~0ull : 0ull;
  auto pick = [&](const void* x0, const void* x1, const void* x2) {
    return reinterpret_cast<const uint8_t*>(((unsigned long long)x0 & m0) | ((unsigned long long)x1 & m1) |
                                            ((unsigned long long)x2 & m2));
  };
  wa.p0 = pick(g.w0.p0, g.w1.p0, g.w2.p0);
  wa.p1 = pick(g.w0.p1, g.w1.p1, g.w2.p1);
  wa.p2 = pick(g.w0.p2, g.w1.p2, g.w2.p2);
  wa.p3 = pick(g.w0.p3, g.w1.p3, g.w2.p3);
  wa.rows = (int)(((unsigned)g.w0.rows & (unsigned)m0) | ((unsigned)g.w1.rows & (unsigned)m1) |
                  ((unsigned)g.w2.rows & (unsigned)m2));
  out = reinterpret_cast<float*>(const_cast<uint8_t*>(pick(g.o0, g.o1, g.o2)));
  wb.p0 = wa.p0;
  wb.p1 = wa.p1;
  wb.p2 = wa.p2;
  wb.p3 = wa.p3;
  wb.rows = wa.rows;
  ra = RPG * lt;
}

template <int TYPE, bool PAIR, bool RESID, bool XATTN, int RA>
__device__ __forceinline__ void run_class_flat(const Segs& g, const GemvVec& a, const XRegs& xr, float* smem, int first,
                                               int stride, int total, int lane) {
  constexpr int RPG = PAIR ? RA : 2 * RA;  // output rows per group
  const float* xs = smem + XS_SCRATCH;
  const int nchunks = a.cols >> 5;                  // > 64 (host)
  const int nsteps = (RA * nchunks + 63) >> 6;
  if (first >= total) {
    stage_x_finish<XATTN>(a, xr, smem);
    return;
  }
  int irg = first, is = 0, ij0 = 0, ioff0 = 0;  // issue cursor: group, step, first row / chunk offset of the step
  QW wa = g.w0, wb = PAIR ? g.w1 : g.w0;
  int ra = 0;
  float* out = g.o0;
  group_of_flat<PAIR, RPG>(g, irg, wa, wb, ra, out);
  float acc[2 * RA];
#pragma unroll
  for (int k = 0; k < 2 * RA; ++k) acc[k] = 0.f;

  auto issue = [&](StepF<TYPE, RA>& st, StepMetaF& m) {
    m.s = is;
    m.j0 = ij0;
    m.off0 = ioff0;
    m.ra = ra;
    m.rows = wa.rows;
    m.out = out;
    const int off = ioff0 + lane;
    const bool hi = off >= nchunks;
    int j = ij0 + (hi ? 1 : 0);
    int chunk = hi ? off - nchunks : off;
    if (j >= RA) {  // past the group's items (last step only): any valid address, contributes nothing
      j = RA - 1;
      chunk = nchunks - 1;
    }
    const int last = wa.rows - 1;
    if (RESID) {
#pragma unroll
      for (int k = 0; k < 2 * RA; ++k) st.r[k] = ldg_f32(a.resid + (ra + k < last ? ra + k : last));
    }
    const int rowA = ra + j < last ? ra + j : last;
    const int rowB = PAIR ? rowA : (ra + RA + j < last ? ra + RA + j : last);
    st.c0.load(wa, rowA, (unsigned)chunk);
    st.c1.load(wb, rowB, (unsigned)chunk);
    ioff0 += 64;  // advance the issue cursor (scalar work, overlaps the loads)
    if (ioff0 >= nchunks) {
      ioff0 -= nchunks;
      ++ij0;
    }
    if (++is == nsteps) {
      is = 0;
      ij0 = 0;
      ioff0 = 0;
      irg += stride;
      if (irg < total) group_of_flat<PAIR, RPG>(g, irg, wa, wb, ra, out);
    }
  };
  auto consume = [&](const StepF<TYPE, RA>& st, const StepMetaF& m) {
    const int off = m.off0 + lane;
    const bool hi = off >= nchunks;
    const int j = m.j0 + (hi ? 1 : 0);
    const int chunk = hi ? off - nchunks : off;
    const bool act = j < RA;
    const int cc = act ? chunk : nchunks - 1;
    float x[32];
    const f32x4* xp = reinterpret_cast<const f32x4*>(xs + cc * XS_STRIDE);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 t = xp[i];
      x[4 * i] = t[0]; x[4 * i + 1] = t[1]; x[4 * i + 2] = t[2]; x[4 * i + 3] = t[3];
    }
    const f32x4 sums = xp[8];
    const float d0 = st.c0.dot(x, sums[0], sums[1], cc & 7);
    const float d1 = st.c1.dot(x, sums[0], sums[1], cc & 7);
#pragma unroll
    for (int k = 0; k < RA; ++k) {
      const bool hit = act && j == k;
      acc[k] += hit ? d0 : 0.0f;
      acc[RA + k] += hit ? d1 : 0.0f;
    }
    if (m.s == nsteps - 1) {  // row group complete: reduce, epilogue, store
      float y[2 * RA];
#pragma unroll
      for (int k = 0; k < 2 * RA; ++k) {
        y[k] = wave_sum_lane63(acc[k]);
        acc[k] = 0.f;
      }
      if (lane == 63) {
        if (PAIR) {
#pragma unroll
          for (int k = 0; k < RA; ++k) {
            const float sg = 1.0f / (1.0f + expf(-y[k]));  // silu_cpu (utils.cpp:700-707), then * up (model.cpp:735)
            if (m.ra + k < m.rows) stg_f32(m.out + m.ra + k, (y[k] * sg) * y[RA + k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 2 * RA; ++k)
            if (m.ra + k < m.rows) stg_f32(m.out + m.ra + k, RESID ? st.r[k] + y[k] : y[k]);
        }
      }
    }
  };

  StepF<TYPE, RA> cur, nxt;  // depth-one software pipeline in the copy form (see run_class)
  StepMetaF mc, mn;
  issue(cur, mc);
  stage_x_finish<XATTN>(a, xr, smem);
  while (irg < total) {
    issue(nxt, mn);
    consume(cur, mc);
    cur = nxt;
    mc = mn;
  }
  consume(cur, mc);
}

// TA: format of class A = segments [0, nsegA); TB: format of class B = segment nsegA (TL_NONE: no class B).
// Waves [0, wavesA) of every workgroup run class A, the others class B.
// ROLE only names the instantiation (tl_internal.h GemvRole) so that profiles separate the per-layer launches.
// RA > 0: the flattened step scheme above with RA rows per stream and group (the host sizes t0..t2 accordingly).
// PEER: the launch carries its tensor-parallel exchange (a.peer; single matrix, RA = 0): all-reduce for the residual roles,
// all-gather otherwise.
// The body of a launch; `bid` is the workgroup's index among the a.grid workgroups of the launch (blockIdx.x), or of its
// task in a chain launch (CHAIN; cw says what to wait for; RA = 0, no PEER).
template <int TA, int TB, bool PAIR, int ROLE, bool STAMP = false, int RA = 0, bool PEER = false, bool CHAIN = false,
          int AHD = 0, int AG = 0>
__device__ __forceinline__ void gemv_body(const GemvSeg& sg0, const GemvSeg& sg1, const GemvSeg& sg2, const GemvVec& a,
                                          int t0, int t1, int t2, int nsegA, int wavesA, int bid, float* smem,
                                          const ChainWait* cw = nullptr) {
  // the roles launched with a residual / with x still in attention partials
  constexpr bool RESID = ROLE == GEMV_ATTN_OUT || ROLE == GEMV_DOWN || ROLE == GEMV_ATTN_OUT_COMBINE;
  constexpr bool XATTN = ROLE == GEMV_ATTN_OUT_COMBINE;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt0 = 0;
  if (STAMP) {
    ts[0] = __builtin_readcyclecounter();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  // All kernel arguments into SGPRs with ONE batch of scalar loads: left alone, hipcc loads them where first used,
  // in four dependent round trips to a cold scalar cache (~2000 cycles before a wave's first weight load goes out).
  asm volatile("" ::"s"(sg0.w.p0), "s"(sg0.w.p1), "s"(sg0.w.p2), "s"(sg0.w.p3), "s"(sg0.w.rows), "s"(sg0.out),
               "s"(a.x), "s"(a.norm_w), "s"(a.resid), "s"(a.eps), "s"(a.cols), "s"(t0), "s"(t1), "s"(t2),
               "s"(nsegA), "s"(wavesA), "s"(a.nthr), "s"(a.grid));
  if (ROLE == GEMV_ATTN_OUT_COMBINE) asm volatile("" ::"s"(a.attn_ws), "s"(a.attn_pos), "s"(a.attn_hd));
  if (PAIR || ROLE == GEMV_QKV)
    asm volatile("" ::"s"(sg1.w.p0), "s"(sg1.w.p1), "s"(sg1.w.p2), "s"(sg1.w.p3), "s"(sg1.w.rows), "s"(sg1.out));
  if (ROLE == GEMV_QKV)
    asm volatile("" ::"s"(sg2.w.p0), "s"(sg2.w.p1), "s"(sg2.w.p2), "s"(sg2.w.p3), "s"(sg2.w.rows), "s"(sg2.out));
  if (STAMP) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the arguments have arrived
    ts[1] = __builtin_readcyclecounter();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = a.nthr >> 6;
  XRegs xr;
  if constexpr (!CHAIN) stage_x_issue<XATTN>(a, xr);  // (CHAIN: after the wait for the predecessor task, in run_class)
  if (STAMP) ts[2] = __builtin_readcyclecounter();
  unsigned pe = 0;  // the epoch this launch's words carry: one more than the exchanges completed (0 = cleared area)
  if (PEER) {
    pe = __hip_atomic_load(a.peer->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    pe = __builtin_amdgcn_readfirstlane(pe == 0u ? 1u : pe);
  }
  if (TB == TL_NONE || wave < wavesA) {
    const int wA = TB == TL_NONE ? nwaves : wavesA;
    const int total = PAIR ? t0 : (nsegA == 1 ? t0 : (nsegA == 2 ? t0 + t1 : t0 + t1 + t2));
    const Segs g = {sg0.w, sg1.w, sg2.w, sg0.out, sg1.out, sg2.out, t0, nsegA >= 2 ? t1 : 0x3fffffff};
    if constexpr (RA > 0)
      run_class_flat<TA, PAIR, RESID, XATTN, RA>(g, a, xr, smem, wave * a.grid + bid, a.grid * wA, total, lane);
    else
      run_class<TA, PAIR, RESID, XATTN, STAMP, PEER, CHAIN, AHD, AG>(g, a, xr, smem, wave * a.grid + bid, a.grid * wA, total,
                                                                     lane, ts, pe, wave, cw);
  } else if (TB != TL_NONE) {
    constexpr int TBB = TB == TL_NONE ? TA : TB;
    const GemvSeg sb = nsegA == 1 ? sg1 : sg2;  // wave-uniform, two-way: stays in SGPRs
    const int tb = nsegA == 1 ? t1 : t2;
    const Segs g = {sb.w, sb.w, sb.w, sb.out, sb.out, sb.out, 0x3fffffff, 0x3fffffff};
    if constexpr (RA > 0)
      run_class_flat<TBB, false, false, XATTN, RA>(g, a, xr, smem, (wave - wavesA) * a.grid + bid,
                                                   a.grid * (nwaves - wavesA), tb, lane);
    else
      run_class<TBB, false, false, XATTN, STAMP, false, CHAIN>(g, a, xr, smem, (wave - wavesA) * a.grid + bid,
                                                               a.grid * (nwaves - wavesA), tb, lane, ts, 0, wave, cw);
  }
  if (PEER) {  // every wave has read the epoch and collected its rows: the last workgroup through advances the epoch
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned before = __hip_atomic_fetch_add(a.peer->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (unsigned)a.grid - 1u) {
        __hip_atomic_store(a.peer->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.peer->epoch, pe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (STAMP && lane == 0 && ts[5] != 0) {
    ts[7] = __builtin_readcyclecounter();
    unsigned long long* o = a.stamps + 10 * (size_t)(wave * a.grid + bid);
    for (int i = 0; i < 8; ++i) o[i] = ts[i];
    o[8] = __builtin_amdgcn_s_memrealtime();
    o[9] = rt0;
  }
}

// Since the launch geometry travels as explicit arguments these kernels have NO hidden argument block: their kernarg
// segment would be exactly the 276 bytes of explicit arguments, not a multiple of 8. rocprofv3's dispatch callback faulted
// twice in round 2 under dispatches of this kernel family (a memmove off the end of a mapping; profiles/README.md,
// "profiler aborts": not explained by the argument block alone -- other kernels of the library share both properties and
// have run under the profiler -- but rounding the segment up to 320 bytes (5 x 64) with an unread tail argument costs
// nothing).
struct GemvPad {
  int unused[11];
};
template <int TA, int TB, bool PAIR, int ROLE, bool STAMP = false, int RA = 0, bool PEER = false>
__global__ __launch_bounds__(GEMV_MAX_THREADS) void gemv_kernel(GemvSeg sg0, GemvSeg sg1, GemvSeg sg2, GemvVec a,
                                                                int t0, int t1, int t2, int nsegA, int wavesA, GemvPad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gemv_body<TA, TB, PAIR, ROLE, STAMP, RA, PEER>(sg0, sg1, sg2, a, t0, t1, t2, nsegA, wavesA, blockIdx.x, smem);
}

// ---- the attn_output launch that carries its layer's attention (RoPE + KV write + split attention, tl_attn_body.h) -------
// 512 threads (the attention body's geometry: 8 waves, up to 256 VGPRs), one workgroup per CU. Every wave first issues the
// weight loads of its first step, then the workgroups that have an attention task run it, everybody waits for the last of
// them (fused_attention_phase) and the x staging combines the partial records: the attention's launch boundary, prologue
// and tail are gone and what is left of its latency overlaps the first HBM round trip of the projection. For sequences of
// at most FUSED_ATTN_MAX_LEN positions (16 splits of 64: beyond that the consumer's combine of up to 32 records per head
// costs more than the split kernel's in-kernel merge, and the step keeps its own attention launch).
template <int TA, int HD, int G>
__global__ __launch_bounds__(tl_attn::ATT_THREADS) void gemv_attn_kernel(GemvSeg sg0, GemvVec a, int t0, FusedAttn fa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nchunks = a.cols >> 5;
  const ChainWait cw = {nullptr, 0u, 0, fa.err, fa.st ? fa.st + 8 * (size_t)blockIdx.x : nullptr, &fa,
                        smem + XS_SCRATCH + nchunks * XS_STRIDE, (int)blockIdx.x, a.grid};
  chain_stamp(&cw, 0);
  gemv_body<TA, TL_NONE, false, GEMV_ATTN_OUT_COMBINE, false, 0, false, true, HD, G>(sg0, sg0, sg0, a, t0, 0, 0, 1, a.nthr >> 6,
                                                                                        blockIdx.x, smem, &cw);
  chain_stamp(&cw, 7);
}

// ---- the chain launch: grid (workgroups per task, tasks); every workgroup runs the body its task names -------------------
struct alignas(128) ChainTask {
  GemvSeg sg[3];
  GemvVec a;
  int t[3];
  int nsegA, wavesA;
  int code;      // chain_code(): role and format(s) of the body
  int nwg;       // workgroups of this task that work (the others of the row leave at once)
  int pred_nwg;  // ... of the task before it in the launch (0: first task, nothing to wait for)
  int pad;
};
static_assert(sizeof(ChainTask) == TL_CHAIN_TASK_BYTES, "TL_CHAIN_TASK_BYTES (tl_internal.h) is the stride of a task array");
constexpr int CHAIN_ROLE_QKV = 0, CHAIN_ROLE_COMBINE = 1, CHAIN_ROLE_GATE_UP = 2, CHAIN_ROLE_DOWN = 3, CHAIN_ROLE_LM_HEAD = 4;
constexpr int CHAIN_QKV_MIXED = 3;  // type slot of the Q4_K + Q6_K launch
__host__ __device__ constexpr int chain_code(int role, int tslot) { return role * 4 + tslot; }
static int chain_type_slot(int type) { return type == TL_Q4_K ? 0 : type == TL_Q6_K ? 1 : type == TL_Q8_0 ? 2 : -1; }

template <int TA, int TB, bool PAIR, int ROLE>
__device__ __forceinline__ void chain_body(const ChainTask& T, int bid, float* smem, const ChainWait* cw) {
  gemv_body<TA, TB, PAIR, ROLE, false, 0, false, true>(T.sg[0], T.sg[1], T.sg[2], T.a, T.t[0], T.t[1], T.t[2], T.nsegA,
                                                       T.wavesA, bid, smem, cw);
}

// Persistent form: ONE workgroup of 16 waves per CU walks through the launch's tasks. A wave that has finished its rows
// of task j drains its stores, adds itself to the workgroup's LDS counter (the last one adds the workgroup's arrival to
// the task's counter) and goes straight on to task j + 1: descriptor, first weight loads, and only then the workgroup
// barrier behind which the predecessor is polled for and x is staged -- so the tail of a task (its slowest waves, ~3 us)
// is covered by the next task's first HBM round trip, wave by wave, and no workgroup ever leaves or enters a CU between
// tasks. Every wave passes the same barriers in the same order (waves without rows only help staging), and the barrier
// of task j + 1's wait is also what makes the x image in LDS free to overwrite.
__global__ __launch_bounds__(GEMV_MAX_THREADS) void chain_kernel(const ChainTask* __restrict__ tasks, int ntasks,
                                                                 unsigned* sync, const unsigned* __restrict__ gen,
                                                                 unsigned* err, unsigned long long* stamps, int stamp_stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the workgroup's count of waves that are through with a task: the last word of the reduction scratch in front of the x
  // image, which no chain body uses (a static __shared__ variable would come off the 160 KiB the x image may take)
  unsigned& wg_waves_done = *(reinterpret_cast<unsigned*>(smem) + (XS_SCRATCH - 1));
  const int bid = blockIdx.x;
  if (threadIdx.x == 0) wg_waves_done = 0u;
  __syncthreads();
  const unsigned gen1 = *gen + 1u;
  unsigned served = 0;  // tasks this workgroup has worked on
  for (int task = 0; task < ntasks; ++task) {
    const ChainTask T = tasks[task];  // (uniform address, read-only: scalar loads)
    if (bid >= T.nwg) continue;       // (uniform over the workgroup)
    unsigned long long* st = stamps ? stamps + 8 * ((size_t)task * stamp_stride + bid) : nullptr;
    const ChainWait cw = {T.pred_nwg > 0 ? sync + (size_t)(task - 1) * CHAIN_TASK_WORDS : nullptr, gen1, T.pred_nwg, err, st};
    chain_stamp(&cw, 0);
    switch (T.code) {
      case chain_code(CHAIN_ROLE_QKV, 0): chain_body<TL_Q4_K, TL_NONE, false, GEMV_QKV>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_QKV, 1): chain_body<TL_Q6_K, TL_NONE, false, GEMV_QKV>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_QKV, 2): chain_body<TL_Q8_0, TL_NONE, false, GEMV_QKV>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_QKV, CHAIN_QKV_MIXED): chain_body<TL_Q4_K, TL_Q6_K, false, GEMV_QKV>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_COMBINE, 0): chain_body<TL_Q4_K, TL_NONE, false, GEMV_ATTN_OUT_COMBINE>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_COMBINE, 1): chain_body<TL_Q6_K, TL_NONE, false, GEMV_ATTN_OUT_COMBINE>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_COMBINE, 2): chain_body<TL_Q8_0, TL_NONE, false, GEMV_ATTN_OUT_COMBINE>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_GATE_UP, 0): chain_body<TL_Q4_K, TL_NONE, true, GEMV_GATE_UP>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_GATE_UP, 1): chain_body<TL_Q6_K, TL_NONE, true, GEMV_GATE_UP>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_GATE_UP, 2): chain_body<TL_Q8_0, TL_NONE, true, GEMV_GATE_UP>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_DOWN, 0): chain_body<TL_Q4_K, TL_NONE, false, GEMV_DOWN>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_DOWN, 1): chain_body<TL_Q6_K, TL_NONE, false, GEMV_DOWN>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_DOWN, 2): chain_body<TL_Q8_0, TL_NONE, false, GEMV_DOWN>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_LM_HEAD, 0): chain_body<TL_Q4_K, TL_NONE, false, GEMV_LM_HEAD>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_LM_HEAD, 1): chain_body<TL_Q6_K, TL_NONE, false, GEMV_LM_HEAD>(T, bid, smem, &cw); break;
      case chain_code(CHAIN_ROLE_LM_HEAD, 2): chain_body<TL_Q8_0, TL_NONE, false, GEMV_LM_HEAD>(T, bid, smem, &cw); break;
      default: break;
    }
    // this wave's stores have been acknowledged; the last wave of the workgroup to get here adds the workgroup's arrival
    ++served;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) {
      const unsigned before = __hip_atomic_fetch_add(&wg_waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (before + 1u == served * (unsigned)(GEMV_MAX_THREADS / 64)) {
        if (st) st[5] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(sync + (size_t)task * CHAIN_TASK_WORDS + (bid & (CHAIN_SHARDS - 1)) * CHAIN_SHARD_WORDS, 1u,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st) st[6] = __builtin_amdgcn_s_memrealtime();
      }
    }
  }
}

// ---- cold path for F32 matrices (only norms are F32 in real GGUFs; kept for completeness of the type list) ------
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                       const float* __restrict__ resid, float* __restrict__ out,
                                                       int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* wr = w + (size_t)row * cols;
  float acc = 0.f;
  for (int i = lane; i < cols; i += 64) acc = fmaf(wr[i], x[i], acc);
  acc = wave_sum_lane63(acc);
  if (lane == 63) out[row] = resid ? resid[row] + acc : acc;
}

static int g_gemv_setup_done = 0;
static int g_num_cus = 256;
static int g_max_waves = 16;

// TL_GEMV_CUS: at most this many workgroups per GEMV launch. For rehearsals of tensor parallelism with several ranks on
// ONE device: a launch that carries its exchange waits for the same launch of the other ranks, so all of them have to fit
// on the chip together (ranks x cap <= CUs); on a device per rank the question does not arise.
static int g_cu_cap = 0;
static int gemv_setup() {
  g_num_cus = tl_num_cus();  // of the CURRENT device (cached per ordinal): a process may drive unlike devices
  if (g_cu_cap > 0 && g_num_cus > g_cu_cap) g_num_cus = g_cu_cap;
  if (g_gemv_setup_done) return 0;
  if (const char* e = getenv("TL_GEMV_CUS")) {
    const int c = atoi(e);
    if (c >= 1) g_cu_cap = c;
    if (g_cu_cap > 0 && g_num_cus > g_cu_cap) g_num_cus = g_cu_cap;
  }
  if (const char* e = getenv("TL_GEMV_WAVES")) {  // tuning knob (experiments only)
    const int w = atoi(e);
    if (w >= 1 && w <= 16) g_max_waves = w;
  }
  g_gemv_setup_done = 1;
  return 0;
}

struct Launch {
  GemvSeg seg[3];
  GemvVec v;
  int t[3];
  int nsegA, wavesA;
  int grid, threads;
  size_t lds;
  int ra;  // 0: span steps; 2 / 4: flattened steps with that many rows per stream and group
};

template <int TA, int TB, bool PAIR, int ROLE, bool STAMP = false, int RA = 0, bool PEER = false>
static int gemv_launch_ra(const Launch& L, hipStream_t s) {
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemv_kernel<TA, TB, PAIR, ROLE, STAMP, RA, PEER>, 160 * 1024, lds_raised));
  GemvVec v = L.v;
  v.nthr = L.threads;
  v.grid = L.grid;
  gemv_kernel<TA, TB, PAIR, ROLE, STAMP, RA, PEER><<<L.grid, L.threads, L.lds, s>>>(L.seg[0], L.seg[1], L.seg[2], v, L.t[0],
                                                                                     L.t[1], L.t[2], L.nsegA, L.wavesA, GemvPad{});
  return 0;
}
template <int TA, int TB, bool PAIR, int ROLE, bool STAMP = false>
static int gemv_launch_one(const Launch& L, hipStream_t s) {
  if constexpr (!STAMP && !PAIR && TB == TL_NONE &&
                (ROLE == GEMV_ATTN_OUT || ROLE == GEMV_DOWN || ROLE == GEMV_ATTN_OUT_COMBINE || ROLE == GEMV_LM_HEAD)) {
    if (L.v.peer) return gemv_launch_ra<TA, TB, PAIR, ROLE, false, 0, true>(L, s);  // (the host keeps these on spans)
  }
  if constexpr (!STAMP && TA != TL_Q8_K) {  // (the host never picks the flattened scheme for Q8_K or the stamped run)
    if (L.ra == 2) return gemv_launch_ra<TA, TB, PAIR, ROLE, false, 2>(L, s);
  }
  return gemv_launch_ra<TA, TB, PAIR, ROLE, STAMP, 0>(L, s);
}

template <int TA, int ROLE>
static int gemv_launch_plain(const Launch& L, hipStream_t s) {
  return gemv_launch_one<TA, TL_NONE, false, ROLE>(L, s);
}

template <int ROLE>
static int gemv_launch_type(int ta, const Launch& L, hipStream_t s) {
  switch (ta) {
    case TL_Q4_K: return gemv_launch_plain<TL_Q4_K, ROLE>(L, s);
    case TL_Q6_K: return gemv_launch_plain<TL_Q6_K, ROLE>(L, s);
    case TL_Q8_0: return gemv_launch_plain<TL_Q8_0, ROLE>(L, s);
    case TL_Q8_K: return gemv_launch_plain<TL_Q8_K, ROLE>(L, s);
    default: tl_set_error("gemv: unsupported weight type %d", ta); return 2;
  }
}

static int gemv_launch_pair(int ta, const Launch& L, hipStream_t s) {
  switch (ta) {
    case TL_Q4_K: return gemv_launch_one<TL_Q4_K, TL_NONE, true, GEMV_GATE_UP>(L, s);
    case TL_Q6_K: return gemv_launch_one<TL_Q6_K, TL_NONE, true, GEMV_GATE_UP>(L, s);
    case TL_Q8_0: return gemv_launch_one<TL_Q8_0, TL_NONE, true, GEMV_GATE_UP>(L, s);
    case TL_Q8_K: return gemv_launch_one<TL_Q8_K, TL_NONE, true, GEMV_GATE_UP>(L, s);
    default: tl_set_error("gemv: unsupported weight type %d", ta); return 2;
  }
}

__global__ void swiglu_inplace_kernel(float* __restrict__ g, const float* __restrict__ u, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = g[i];
  g[i] = (x * (1.0f / (1.0f + expf(-x)))) * u[i];  // utils.cpp:704-705, model.cpp:735
}

// Step scheme of a launch: 0 = spans of 64 chunks; 2 = flattened steps with two rows per stream and group, when that
// lowers the share of idle load slots AND the launch still has a row group for every wave of the chip: a flattened
// group is twice the rows, and below 16 groups per CU the lost waves cost more than the idle slots (measured:
// Llama-3.2-3B gate/up 14.2 -> 12.4 us and lm_head 71.8 -> 65.5 us, but attn_output 7.6 -> 8.1 us with 768 groups; four
// rows per stream made TinyLlama's ffn_down, 256 groups, twice as slow). TL_GEMV_FLAT=0 switches it off for A/B runs.
static int gemv_flat_rows(int cols, bool allowed, long total_rows, bool pair) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("TL_GEMV_FLAT");
    enabled = e ? atoi(e) != 0 : 1;
  }
  const int nchunks = cols / 32;
  if (!allowed || !enabled || nchunks <= 64 || nchunks % 64 == 0) return 0;
  if (total_rows / (pair ? 2 : 4) < 16L * g_num_cus) return 0;
  auto padded = [&](int ra) { return (double)((ra * nchunks + 63) / 64 * 64) / (double)(ra * nchunks); };
  const double span = (double)((nchunks + 63) / 64 * 64) / (double)nchunks;
  int best = 0;
  double best_pad = span - 0.02;  // must save at least 2 % of the slots
  for (int ra : {2})
    if (padded(ra) < best_pad - 1e-9) {
      best = ra;
      best_pad = padded(ra);
    }
  return best;
}

static int gemv_geometry(Launch& L, int total) {
  // one workgroup per CU, all resident at once; small problems use fewer CUs / fewer waves per workgroup
  L.grid = total < g_num_cus ? total : g_num_cus;
  int waves = (total + L.grid - 1) / L.grid;
  if (waves > g_max_waves) waves = g_max_waves;
  if (waves < 4) waves = 4;  // the x staging pass wants a few waves
  L.threads = waves * 64;
  return waves;
}

// Plans a fused launch: geometry, row groups, format classes. *plain = false: the arguments need one of the unfused routes
// of tl_launch_gemv (F32 matrices, a gate/up pair in two formats, other format mixtures) and L is not filled in.
// role_out: the instantiation (GemvRole); classB_out: Q4_K segments followed by one Q6_K segment.
static int gemv_plan(const GemvArgs& a, Launch& L, int* role_out, bool* classB_out, bool* plain) {
  *plain = false;
  TL_TRY(gemv_setup());
  TL_REQUIRE(a.nseg >= 1 && a.nseg <= 3, "gemv: nseg %d", a.nseg);
  TL_REQUIRE(a.cols > 0 && a.cols % 32 == 0, "gemv: cols %d must be a positive multiple of 32", a.cols);
  for (int i = 0; i < a.nseg; ++i) {
    TL_REQUIRE(a.seg[i].w.cols == a.cols, "gemv: segment %d has cols %d, expected %d", i, a.seg[i].w.cols, a.cols);
    TL_REQUIRE(a.seg[i].w.rows > 0, "gemv: segment %d has no rows", i);
  }
  if (a.pair)
    TL_REQUIRE(a.nseg == 2 && a.seg[0].w.rows == a.seg[1].w.rows, "gemv: pair mode needs two equal-shaped matrices");
  TL_REQUIRE(!a.resid || a.nseg == 1, "gemv: residual epilogue needs a single matrix");
  for (int i = 0; i < a.nseg; ++i)
    if (a.seg[i].w.type == TL_F32) return 0;
  L = Launch{};
  L.lds = sizeof(float) * (XS_SCRATCH + (size_t)(a.cols / 32) * XS_STRIDE);
  TL_REQUIRE(L.lds <= 160 * 1024, "gemv: cols %d needs %zu B of LDS (> 160 KiB)", a.cols, L.lds);
  L.v = GemvVec{a.x, a.norm_w, a.resid, a.eps, a.cols, nullptr, a.attn_ws, a.attn_pos, a.attn_hd, a.attn_merged, 0, 0, a.peer, 0};
  if (a.peer)
    TL_REQUIRE(a.nseg == 1 && !a.pair && (a.resid || a.role == GEMV_LM_HEAD),
               "gemv: a launch that carries its exchange is one matrix with a residual (all-reduce) or the lm_head (all-gather)");
  for (int i = 0; i < 3; ++i) L.seg[i] = a.seg[i < a.nseg ? i : 0];
  const int ta = a.seg[0].w.type;
  bool flat_ok = true;
  for (int i = 0; i < a.nseg; ++i) flat_ok &= a.seg[i].w.type != TL_Q8_K;
  long total_rows = 0;
  for (int i = 0; i < (a.pair ? 1 : a.nseg); ++i) total_rows += a.seg[i].w.rows;
  L.ra = a.peer ? 0 : gemv_flat_rows(a.cols, flat_ok, total_rows, a.pair != 0);  // (the collect phase walks span row groups)
  *classB_out = false;
  if (a.pair) {
    if (a.seg[1].w.type != ta) return 0;  // gate and up in different formats: unfused
    L.t[0] = L.ra ? (a.seg[0].w.rows + L.ra - 1) / L.ra : a.seg[0].w.rows;
    L.nsegA = 2;
    L.wavesA = gemv_geometry(L, L.t[0]);
    *role_out = GEMV_GATE_UP;
    *plain = true;
    return 0;
  }
  // format classes: segments [0, nsegA) share seg[0]'s format; at most ONE trailing segment may differ
  int nsegA = 1;
  while (nsegA < a.nseg && a.seg[nsegA].w.type == ta) ++nsegA;
  const bool classB = nsegA < a.nseg;
  if (classB && !(nsegA == a.nseg - 1 && ta == TL_Q4_K && a.seg[nsegA].w.type == TL_Q6_K)) return 0;  // other mixtures
  double bytesA = 0.0, bytesB = 0.0;
  int totalA = 0;
  for (int i = 0; i < a.nseg; ++i) {
    L.t[i] = L.ra ? (a.seg[i].w.rows + 2 * L.ra - 1) / (2 * L.ra) : (a.seg[i].w.rows + 1) / 2;
    const double b = (double)a.seg[i].w.rows * a.cols / tl_type_block_elems(a.seg[i].w.type) *
                     (double)tl_type_block_bytes(a.seg[i].w.type);
    if (i < nsegA) {
      bytesA += b;
      totalA += L.t[i];
    } else {
      bytesB += b;
    }
  }
  L.nsegA = nsegA;
  const int total = classB ? totalA + L.t[nsegA] : totalA;
  const int waves = gemv_geometry(L, total);
  L.wavesA = waves;
  if (a.peer) {  // the waves' stashes of finished rows behind the x image
    L.v.peer_gpw = (total + L.grid * waves - 1) / (L.grid * waves);
    L.lds += sizeof(float) * 2 * (size_t)L.v.peer_gpw * (size_t)waves;
    TL_REQUIRE(L.lds <= 160 * 1024, "gemv: %d row groups per wave do not fit the LDS stash of a launch that carries its exchange",
               L.v.peer_gpw);
  }
  *classB_out = classB;
  if (classB) {
    int wa = (int)(waves * bytesA / (bytesA + bytesB) + 0.5);
    if (wa < 1) wa = 1;
    if (wa > waves - 1) wa = waves - 1;
    L.wavesA = wa;
    *role_out = GEMV_QKV;
    *plain = true;
    return 0;
  }
  int role = a.role;  // the residual epilogue is compiled into the ATTN_OUT / DOWN instantiations only
  if (a.resid && role != GEMV_ATTN_OUT && role != GEMV_DOWN) role = GEMV_ATTN_OUT;
  if (!a.resid && (role == GEMV_ATTN_OUT || role == GEMV_DOWN)) role = GEMV_GENERIC;
  if (a.attn_ws) {
    TL_REQUIRE(a.resid && a.attn_pos && !a.norm_w && a.nseg == 1 && a.attn_hd > 0 && a.attn_hd % 4 == 0 &&
                   a.cols % a.attn_hd == 0,
               "gemv: attention-partials input needs a single residual launch without norm");
    role = GEMV_ATTN_OUT_COMBINE;
  }
  *role_out = role;
  *plain = true;
  return 0;
}

int tl_launch_gemv(const GemvArgs& a, hipStream_t s) {
  Launch L;
  int role = GEMV_GENERIC;
  bool classB = false, plain = false;
  TL_TRY(gemv_plan(a, L, &role, &classB, &plain));
  const int ta = a.seg[0].w.type;
  if (!plain) {
    bool any_f32 = false;
    for (int i = 0; i < a.nseg; ++i) any_f32 |= a.seg[i].w.type == TL_F32;
    if (any_f32) {
      TL_REQUIRE(!a.pair && !a.norm_w && !a.peer, "gemv: F32 matrices are not supported in fused launches");
      for (int i = 0; i < a.nseg; ++i) {
        TL_REQUIRE(a.seg[i].w.type == TL_F32, "gemv: mixed F32 / quantized segments are not supported");
        const QW& w = a.seg[i].w;
        gemv_f32_kernel<<<(w.rows + 3) / 4, 256, 0, s>>>(reinterpret_cast<const float*>(w.p0), a.x, a.resid,
                                                          a.seg[i].out, w.rows, w.cols);
      }
      TL_HIP(hipGetLastError());
      return 0;
    }
    if (a.pair) {
      // gate and up in different formats (not produced by any mix the reference loads): two launches + SwiGLU.
      // The up projection goes through a.seg[1].out, which the caller must then provide.
      TL_REQUIRE(a.seg[1].out, "gemv: a mixed-format gate/up pair needs a scratch output for the up projection");
      GemvArgs g1 = a, g2 = a;
      g1.pair = g2.pair = 0;
      g1.nseg = g2.nseg = 1;
      g2.seg[0] = a.seg[1];
      g1.role = g2.role = GEMV_GENERIC;
      TL_TRY(tl_launch_gemv(g1, s));
      TL_TRY(tl_launch_gemv(g2, s));
      const int n = a.seg[0].w.rows;
      swiglu_inplace_kernel<<<(n + 255) / 256, 256, 0, s>>>(a.seg[0].out, a.seg[1].out, n);
      TL_HIP(hipGetLastError());
      return 0;
    }
    // any other format mixture: one launch per segment (correct, just not fused)
    for (int i = 0; i < a.nseg; ++i) {
      GemvArgs one = a;
      one.nseg = 1;
      one.seg[0] = a.seg[i];
      TL_TRY(tl_launch_gemv(one, s));
    }
    return 0;
  }
  int rc = 0;
  if (a.pair) {
    rc = gemv_launch_pair(ta, L, s);
  } else if (classB) {
    rc = gemv_launch_one<TL_Q4_K, TL_Q6_K, false, GEMV_QKV>(L, s);
  } else {
    switch (role) {
      case GEMV_ATTN_OUT_COMBINE: rc = gemv_launch_type<GEMV_ATTN_OUT_COMBINE>(ta, L, s); break;
      case GEMV_QKV: rc = gemv_launch_type<GEMV_QKV>(ta, L, s); break;
      case GEMV_ATTN_OUT: rc = gemv_launch_type<GEMV_ATTN_OUT>(ta, L, s); break;
      case GEMV_DOWN: rc = gemv_launch_type<GEMV_DOWN>(ta, L, s); break;
      case GEMV_LM_HEAD: rc = gemv_launch_type<GEMV_LM_HEAD>(ta, L, s); break;
      default: rc = gemv_launch_type<GEMV_GENERIC>(ta, L, s); break;
    }
  }
  TL_TRY(rc);
  TL_HIP(hipGetLastError());
  return 0;
}

static unsigned long long* g_chain_stamps = nullptr;
static size_t g_chain_stamps_cap = 0, g_chain_stamps_used = 0;
// ---- the attn_output launch that carries its layer's attention: host side ---------------------------------------------
static_assert(TL_SYNC_LINE_WORDS == CHAIN_SHARD_WORDS, "one counter per 128-byte line");
size_t tl_fused_attn_sync_words() { return (size_t)(2 + CHAIN_SHARDS) * CHAIN_SHARD_WORDS; }  // count | generation | epochs

template <int TA, int HD, int G>
static int gemv_attn_launch(const Launch& L, const FusedAttn& fa, hipStream_t s) {
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemv_attn_kernel<TA, HD, G>, 160 * 1024, lds_raised));
  GemvVec v = L.v;
  v.nthr = L.threads;
  v.grid = L.grid;
  const size_t lds = L.lds + sizeof(float) * tl_attn::attn_split_lds_floats<HD, G>();
  TL_REQUIRE(lds <= 160 * 1024, "gemv+attention: %zu B of LDS (> 160 KiB)", lds);
  gemv_attn_kernel<TA, HD, G><<<L.grid, L.threads, lds, s>>>(L.seg[0], v, L.t[0], fa);
  return 0;
}
template <int TA>
static int gemv_attn_launch_shape(const Launch& L, const FusedAttn& fa, int G, hipStream_t s) {
  if (fa.at.hd == 64) return G == 4 ? gemv_attn_launch<TA, 64, 4>(L, fa, s) : gemv_attn_launch<TA, 64, 3>(L, fa, s);
  return G == 4 ? gemv_attn_launch<TA, 128, 4>(L, fa, s) : gemv_attn_launch<TA, 128, 3>(L, fa, s);
}
// query heads per workgroup of the folded attention (tl_attn.hip's rule); 0: no folded form for this head configuration
static int fused_attn_heads(const AttnArgs& t) {
  if (!(t.hd == 64 || t.hd == 128) || t.nkv <= 0 || t.nh % t.nkv != 0) return 0;
  const int group = t.nh / t.nkv;
  return group % 4 == 0 ? 4 : (group % 3 == 0 ? 3 : 0);
}
bool tl_gemv_attn_supported(const GemvArgs& a, const AttnArgs& t) {
  Launch L;
  int role = GEMV_GENERIC;
  bool classB = false, plain = false;
  if (gemv_plan(a, L, &role, &classB, &plain) != 0) return false;
  const int ta = a.seg[0].w.type;
  if (!plain || role != GEMV_ATTN_OUT_COMBINE || L.ra != 0 || a.peer || a.attn_merged) return false;
  if (!(ta == TL_Q4_K || ta == TL_Q6_K || ta == TL_Q8_0)) return false;
  const int G = fused_attn_heads(t);
  if (G == 0 || a.cols != t.nh * t.hd) return false;
  const size_t att = sizeof(float) * (t.hd == 64 ? (G == 4 ? tl_attn::attn_split_lds_floats<64, 4>() : tl_attn::attn_split_lds_floats<64, 3>())
                                                 : (G == 4 ? tl_attn::attn_split_lds_floats<128, 4>() : tl_attn::attn_split_lds_floats<128, 3>()));
  return L.lds + att <= 160 * 1024;
}
// a = the attn_output launch (attention partials as x, attn_merged = 0), t = its layer's attention (done_cnt unused);
// sync = tl_fused_attn_sync_words() zeroed words shared by the launches of a model, tag in [1, tags_per_gen)
int tl_launch_gemv_attn(const GemvArgs& a, const AttnArgs& t, unsigned* sync, unsigned tag, unsigned tags_per_gen,
                        unsigned* err, hipStream_t s) {
  TL_REQUIRE(tl_gemv_attn_supported(a, t), "gemv+attention: this launch has no folded form");
  Launch L;
  int role = GEMV_GENERIC;
  bool classB = false, plain = false;
  TL_TRY(gemv_plan(a, L, &role, &classB, &plain));
  L.grid = g_num_cus;  // every CU: the attention tasks want workgroups of their own even where the projection has few rows
  L.threads = tl_attn::ATT_THREADS;
  L.wavesA = L.threads / 64;
  FusedAttn fa = {};
  fa.at = t;
  fa.at.done_cnt = nullptr;
  fa.at.ws = const_cast<float*>(a.attn_ws);
  const int G = fused_attn_heads(t);
  fa.at.zparts = (t.nh / t.nkv) / G;
  if (fa.at.kv_row == 0) {
    fa.at.kv_row = t.nkv * t.hd;
    fa.at.kv_head = t.hd;
  }
  fa.count = sync;
  fa.gen = sync + CHAIN_SHARD_WORDS;
  fa.epoch = sync + 2 * CHAIN_SHARD_WORDS;
  fa.tag = tag;
  fa.tags_per_gen = tags_per_gen;
  fa.err = err;
  if (g_chain_stamps && g_chain_stamps_used + (size_t)L.grid * 8 <= g_chain_stamps_cap) {  // diagnostic run
    fa.st = g_chain_stamps + g_chain_stamps_used;
    g_chain_stamps_used += 256 * 8;
  }
  int rc = 0;
  switch (a.seg[0].w.type) {
    case TL_Q4_K: rc = gemv_attn_launch_shape<TL_Q4_K>(L, fa, G, s); break;
    case TL_Q6_K: rc = gemv_attn_launch_shape<TL_Q6_K>(L, fa, G, s); break;
    default: rc = gemv_attn_launch_shape<TL_Q8_0>(L, fa, G, s); break;
  }
  TL_TRY(rc);
  TL_HIP(hipGetLastError());
  return 0;
}

// ---- chain launches: host side -------------------------------------------------------------------------------------------
// tl_chain_plan turns the arguments of one fused launch into a task of a chain launch (*ok = false: this launch has no chain
// form -- an unfused route, flattened steps, a format without a chain body, a carried exchange); tasks of one chain are
// uploaded by the caller (device array of ChainTask, TL_CHAIN_TASK_BYTES each) and run by tl_launch_chain.
int tl_chain_plan(const GemvArgs& a, void* task_out, size_t* lds_out, int pred_nwg, int* nwg_out, bool* ok) {
  *ok = false;
  Launch L;
  int role = GEMV_GENERIC;
  bool classB = false, plain = false;
  TL_TRY(gemv_plan(a, L, &role, &classB, &plain));
  if (!plain || L.ra != 0 || a.peer) return 0;
  int crole = -1;
  switch (role) {
    case GEMV_QKV: crole = CHAIN_ROLE_QKV; break;
    case GEMV_ATTN_OUT_COMBINE: crole = CHAIN_ROLE_COMBINE; break;
    case GEMV_GATE_UP: crole = CHAIN_ROLE_GATE_UP; break;
    case GEMV_DOWN: crole = CHAIN_ROLE_DOWN; break;
    case GEMV_LM_HEAD: crole = CHAIN_ROLE_LM_HEAD; break;
    default: return 0;
  }
  const int slot = classB ? CHAIN_QKV_MIXED : chain_type_slot(a.seg[0].w.type);
  if (slot < 0) return 0;
  ChainTask t = {};
  for (int i = 0; i < 3; ++i) {
    t.sg[i] = L.seg[i];
    t.t[i] = L.t[i];
  }
  t.a = L.v;
  // every wave of the persistent workgroup takes part in every task (same barriers for all). A launch of its own with
  // fewer waves has at most one row group per wave, and wave w takes the same one here (the others only help staging
  // x); a row's sum does not depend on the wave that forms it
  t.a.nthr = GEMV_MAX_THREADS;
  t.a.grid = L.grid;
  t.nsegA = L.nsegA;
  t.wavesA = L.wavesA;
  t.code = chain_code(crole, slot);
  t.nwg = L.grid;
  t.pred_nwg = pred_nwg;
  memcpy(task_out, &t, sizeof(t));
  *lds_out = L.lds;
  *nwg_out = L.grid;
  *ok = true;
  return 0;
}

// tasks_dev: ntasks ChainTask records (TL_CHAIN_TASK_BYTES each, 128-byte aligned); sync: ntasks * CHAIN_TASK_WORDS
// counter words of THIS launch (zero at allocation, never reset); gen: the generation word; err: the sticky error word
int tl_launch_chain(const void* tasks_dev, int ntasks, int max_nwg, size_t lds, unsigned* sync, const unsigned* gen,
                    unsigned* err, hipStream_t s) {
  TL_REQUIRE(tasks_dev && ntasks > 0 && max_nwg > 0 && sync && gen && err, "tl_launch_chain: bad argument");
  static TlDeviceOnce lds_raised;
  TL_TRY(tl_raise_lds_limit(chain_kernel, 160 * 1024, lds_raised));
  unsigned long long* st = g_chain_stamps;
  if (st) {  // diagnostic run: consecutive launches fill consecutive regions of the buffer
    if (g_chain_stamps_used + (size_t)ntasks * 256 * 8 > g_chain_stamps_cap) st = nullptr;
    else {
      st += g_chain_stamps_used;
      g_chain_stamps_used += (size_t)ntasks * 256 * 8;
    }
  }
  chain_kernel<<<max_nwg, GEMV_MAX_THREADS, lds, s>>>(static_cast<const ChainTask*>(tasks_dev), ntasks, sync, gen, err, st, 256);
  TL_HIP(hipGetLastError());
  return 0;
}
// Diagnostic hook (no counterpart in the reference): from now on every chain launch writes 8 realtime stamps per workgroup
// (ChainWait::st) into buf_dev, [launch][task][256 workgroups][8] u64 in launch order, until cap_u64 words are used; nullptr
// switches the stamps off. Launch eagerly (TL_NO_GRAPH=1): a captured graph keeps the pointer it was captured with.
extern "C" int tl_chain_set_stamps(unsigned long long* buf_dev, size_t cap_u64) {
  g_chain_stamps = buf_dev;
  g_chain_stamps_cap = buf_dev ? cap_u64 : 0;
  g_chain_stamps_used = 0;
  return 0;
}
size_t tl_chain_sync_words(int ntasks) { return (size_t)ntasks * CHAIN_TASK_WORDS; }
__global__ void chain_gen_bump_kernel(unsigned* gen) { *gen = *gen + 1u; }
int tl_launch_chain_gen_bump(unsigned* gen, hipStream_t s) {
  chain_gen_bump_kernel<<<1, 1, 0, s>>>(gen);
  TL_HIP(hipGetLastError());
  return 0;
}

// Diagnostic hook (no counterpart in the reference): one tl_matvec_q-shaped launch of the stamped instantiation;
// stamps_dev receives 10 u64 per wave slot (grid * waves-per-block slots, <= 256 * 16): shader-clock stamps at kernel
// entry, arguments in SGPRs, x loads issued, first weight loads issued, x staged, first step done, last loads issued,
// wave end (last dot, reduction and store done); [8], [9] = realtime at wave end / entry. Q4_K and Q6_K only.
extern "C" int tl_debug_gemv_stamps(const tl_qweight* w, const float* x_dev, float* y_dev, unsigned long long* stamps_dev,
                                    int* grid_out, int* waves_out, void* stream) {
  TL_REQUIRE(w && x_dev && y_dev && stamps_dev && (w->type == TL_Q4_K || w->type == TL_Q6_K),
             "tl_debug_gemv_stamps: bad argument (Q4_K / Q6_K only)");
  TL_TRY(gemv_setup());
  hipStream_t s = static_cast<hipStream_t>(stream);
  Launch L = {};
  L.lds = sizeof(float) * (XS_SCRATCH + (size_t)(w->cols / 32) * XS_STRIDE);
  TL_REQUIRE(L.lds <= 160 * 1024, "tl_debug_gemv_stamps: cols %d too large", w->cols);
  L.v = GemvVec{x_dev, nullptr, nullptr, 0.f, w->cols, stamps_dev, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0};
  for (int i = 0; i < 3; ++i) L.seg[i] = GemvSeg{qw_view(w), y_dev};
  L.t[0] = (w->rows + 1) / 2;
  L.nsegA = 1;
  L.wavesA = gemv_geometry(L, L.t[0]);
  TL_HIP(hipMemsetAsync(stamps_dev, 0, sizeof(unsigned long long) * 10 * (size_t)L.grid * (L.threads / 64), s));
  if (w->type == TL_Q4_K) TL_TRY((gemv_launch_one<TL_Q4_K, TL_NONE, false, GEMV_GENERIC, true>(L, s)));
  else TL_TRY((gemv_launch_one<TL_Q6_K, TL_NONE, false, GEMV_GENERIC, true>(L, s)));
  TL_HIP(hipGetLastError());
  if (grid_out) *grid_out = L.grid;
  if (waves_out) *waves_out = L.threads / 64;
  return 0;
}

extern "C" int tl_matvec_q(const tl_qweight* w, const float* x_dev, float* y_dev, void* stream) {
  TL_REQUIRE(w && x_dev && y_dev, "tl_matvec_q: null argument");
  GemvArgs a = {};
  a.seg[0].w = qw_view(w);
  a.seg[0].out = y_dev;
  a.nseg = 1;
  a.cols = w->cols;
  a.x = x_dev;
  return tl_launch_gemv(a, static_cast<hipStream_t>(stream));
}

// ---- Infinity-Cache prefetch ------------------------------------------------------------------------------------------
// Reads [p, p + bytes) with default-policy 16 B loads and throws the data away: the lines land in the 256 MiB
// memory-side Infinity Cache, from where a later (non-temporal) weight stream reads them faster than from HBM. Meant
// to run on a second stream / graph branch BESIDE the kernel that precedes the consumer of these bytes.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[i + k * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
  }
  for (; i < n16; i += stride) {
    const u32x4 v = p[i];
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;  // keeps the loads alive; practically never taken
}

int tl_launch_prefetch(const void* p, size_t bytes, int blocks, unsigned* sink, hipStream_t s) {
  if (bytes < 16 || blocks <= 0) return 0;
  prefetch_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<const u32x4*>(p), bytes / 16, sink);
  TL_HIP(hipGetLastError());
  return 0;
}

// Measurement hook (experiment): like tl_bench_matvec, but a second stream prefetches matrix i+1 into the Infinity Cache
// while the GEMV over matrix i runs; the whole chain is captured into one graph (two branches) and replayed.
extern "C" int tl_bench_matvec_prefetch(const tl_qweight* const* ws, int nw, const float* x_dev, float* y_dev, int reps,
                                        int prefetch_blocks, float* avg_us_out) {
  TL_REQUIRE(ws && nw > 1 && x_dev && y_dev && reps > 0 && avg_us_out, "tl_bench_matvec_prefetch: bad argument");
  hipStream_t sa, sb;
  TL_HIP(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  TL_HIP(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned* sink = nullptr;
  TL_HIP(hipMalloc(&sink, 4));
  TL_TRY(tl_matvec_q(ws[0], x_dev, y_dev, sa));  // attributes set outside capture
  TL_HIP(hipStreamSynchronize(sa));
  std::vector<hipEvent_t> evs;
  auto new_event = [&]() {
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    evs.push_back(e);
    return e;
  };
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  TL_HIP(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < nw && rc == 0; ++i) {
    if (prefetch_blocks > 0) {  // branch: prefetch of matrix i+1 starts when GEMV i-1 has finished (= GEMV i starts)
      hipEvent_t e = new_event();
      hipEventRecord(e, sa);
      hipStreamWaitEvent(sb, e, 0);
      const tl_qweight* nxt = ws[(i + 1) % nw];
      rc = tl_launch_prefetch(nxt->base, nxt->bytes, prefetch_blocks, sink, sb);
    }
    if (rc == 0) rc = tl_matvec_q(ws[i], x_dev, y_dev, sa);
  }
  if (prefetch_blocks > 0) {  // join the branch
    hipEvent_t e = new_event();
    hipEventRecord(e, sb);
    hipStreamWaitEvent(sa, e, 0);
  }
  hipError_t ce = hipStreamEndCapture(sa, &graph);
  if (rc != 0) return rc;
  TL_HIP(ce);
  TL_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  TL_HIP(hipEventCreate(&e0));
  TL_HIP(hipEventCreate(&e1));
  TL_HIP(hipGraphLaunch(exec, sa));  // warm-up
  TL_HIP(hipEventRecord(e0, sa));
  for (int r = 0; r < reps; ++r) TL_HIP(hipGraphLaunch(exec, sa));
  TL_HIP(hipEventRecord(e1, sa));
  TL_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TL_HIP(hipEventElapsedTime(&ms, e0, e1));
  *avg_us_out = ms * 1e3f / ((float)reps * nw);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  for (hipEvent_t e : evs) hipEventDestroy(e);
  hipGraphExecDestroy(exec);
  hipGraphDestroy(graph);
  hipFree(sink);
  hipStreamDestroy(sa);
  hipStreamDestroy(sb);
  return 0;
}

// Measurement hook: `iters` back-to-back GEMVs cycling through `nw` equally-shaped matrices (more than the
// 256 MiB Infinity Cache holds, so that every launch streams from HBM), timed with HIP events on `stream`.
extern "C" int tl_bench_matvec(const tl_qweight* const* ws, int nw, const float* x_dev, float* y_dev, int iters,
                               int warmup, float* avg_us_out, void* stream) {
  TL_REQUIRE(ws && nw > 0 && x_dev && y_dev && iters > 0 && avg_us_out, "tl_bench_matvec: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  TL_HIP(hipEventCreate(&e0));
  TL_HIP(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) TL_TRY(tl_matvec_q(ws[i % nw], x_dev, y_dev, stream));
  TL_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) TL_TRY(tl_matvec_q(ws[i % nw], x_dev, y_dev, stream));
  TL_HIP(hipEventRecord(e1, s));
  TL_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  TL_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_us_out = ms * 1e3f / iters;
  return 0;
}

// ---- multi-vector GEMV: a FEW input vectors against ONE pass over the weights --------------------------------------------
// The multi-sequence decode step with 2..8 sequences (tl_model_batch_*): too few tokens for the matrix-core GEMM (whose
// cost per step does not depend on the token count below 128), too many to stream the weights once per token. Same
// layout, chunk loads, per-format dot and depth-one register pipeline as the kernel above; the chunk in registers is
// multiplied with the staged chunk of EVERY vector (the unpack work is shared, one FMA chain per vector), 2 * NV
// accumulators per lane. x is [nv][cols] (RMSNorm per vector fused into the staging when norm_w is given), outputs are
// [nv][rows of the segment], the residual is [nv][rows]. One format per launch (the host splits mixed QKV launches).
// LDS holds NV staged vectors: (64 + NV * cols/32 * 36) floats must fit 160 KiB (the host splits the vectors otherwise).
struct NvVec {
  const float* x;
  const float* norm_w;
  const float* resid;
  float eps;
  int cols, nv;
};

template <int NV>
__device__ __forceinline__ void stage_x_nv(const NvVec& a, float* smem) {
  float* xs = smem + XS_SCRATCH;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n4 = a.cols >> 2, nchunks = a.cols >> 5;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(a.norm_w);
  float nf[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) nf[v] = 1.0f;
  if (a.norm_w) {
    float ssq[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      ssq[v] = 0.f;
      if (v < a.nv) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)v * a.cols);
        for (int i = tid; i < n4; i += nthr) ssq[v] += ssq4(x4[i]);
      }
      ssq[v] = wave_sum_all(ssq[v]);
      if ((tid & 63) == 0) smem[v * 16 + (tid >> 6)] = ssq[v];  // NV * 16 = 64 floats = XS_SCRATCH at most
    }
    __syncthreads();
    const float e = a.eps < 1e-5f ? 1e-5f : a.eps;  // SAFE_MAX(eps, MIN_NORM_EPS)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float tot = 0.f;
      for (int w = 0; w < (nthr >> 6); ++w) tot += smem[v * 16 + w];
      nf[v] = 1.0f / sqrtf(tot / (float)a.cols + e);
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)(v < a.nv ? v : 0) * a.cols);
    float* xv = xs + (size_t)v * nchunks * XS_STRIDE;
    for (int i = tid; i < n4; i += nthr) {  // n4 is a multiple of 8 and nthr of 64: quads are all-in or all-out
      f32x4 t = x4[i];
      if (a.norm_w) {
        const f32x4 g = w4[i];
        t[0] = t[0] * nf[v] * g[0]; t[1] = t[1] * nf[v] * g[1]; t[2] = t[2] * nf[v] * g[2]; t[3] = t[3] * nf[v] * g[3];
      }
      const int idx = i << 2;
      float* dst = xv + (idx >> 5) * XS_STRIDE;
      *reinterpret_cast<f32x4*>(dst + (idx & 31)) = t;
      float s = (t[0] + t[1]) + (t[2] + t[3]);
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if ((i & 3) == 0) dst[32 + ((i >> 2) & 1)] = s;
    }
  }
  __syncthreads();
}

template <int TYPE, int NV>
struct StepNv {
  Chunk<TYPE> c0, c1;
  float r[2 * NV];
};

// MODE 0: plain, up to three segments of one format; 1: gate/up pair + SwiGLU; 2: one segment + residual
// (NV = 4 runs 8 waves per workgroup: 2 * 4 accumulators, the residuals and two chunk buffers need more than the 128
// VGPRs a lane has at 16 waves)
template <int TYPE, int NV, int MODE>
__global__ __launch_bounds__(NV <= 2 ? GEMV_MAX_THREADS : GEMV_MAX_THREADS / 2) void gemv_nv_kernel(GemvSeg sg0, GemvSeg sg1, GemvSeg sg2, NvVec a, int t0,
                                                                   int t1, int t2, int nseg) {
  constexpr bool PAIR = MODE == 1, RESID = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const float* xs = smem + XS_SCRATCH;
  const int nchunks = a.cols >> 5;
  const int nspans = (nchunks + 63) >> 6;
  const int total = PAIR ? t0 : (nseg == 1 ? t0 : (nseg == 2 ? t0 + t1 : t0 + t1 + t2));
  const Segs g = {sg0.w, sg1.w, sg2.w, sg0.out, sg1.out, sg2.out, t0, nseg >= 2 ? t1 : 0x3fffffff};
  const int first = wave * gridDim.x + blockIdx.x, stride = gridDim.x * nwaves;
  if (first >= total) {
    stage_x_nv<NV>(a, smem);
    return;
  }
  int irg = first, isp = 0;
  QW wa = g.w0, wb = PAIR ? g.w1 : g.w0;
  int ra = 0, rb = 0;
  float* out = g.o0;
  group_of<PAIR>(g, irg, wa, wb, ra, rb, out);
  float acc[2 * NV];
#pragma unroll
  for (int k = 0; k < 2 * NV; ++k) acc[k] = 0.f;

  auto issue = [&](StepNv<TYPE, NV>& st, StepMeta& m) {
    m.sp = isp;
    m.ra = ra;
    m.rows = wa.rows;
    m.out = out;
    const int chunk = (isp << 6) + lane;
    const unsigned cc = chunk < nchunks ? chunk : nchunks - 1;
    if (RESID) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const size_t vo = (size_t)(v < a.nv ? v : 0) * wa.rows;
        st.r[2 * v] = ldg_f32(a.resid + vo + ra);
        st.r[2 * v + 1] = ldg_f32(a.resid + vo + rb);
      }
    }
    st.c0.load(wa, ra, cc);
    st.c1.load(wb, rb, cc);
    if (++isp == nspans) {
      isp = 0;
      irg += stride;
      if (irg < total) group_of<PAIR>(g, irg, wa, wb, ra, rb, out);
    }
  };
  auto consume = [&](const StepNv<TYPE, NV>& st, const StepMeta& m) {
    const int chunk = (m.sp << 6) + lane;
    const bool act = chunk < nchunks;
    const int cc = act ? chunk : nchunks - 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float x[32];
      const f32x4* xp = reinterpret_cast<const f32x4*>(xs + ((size_t)v * nchunks + cc) * XS_STRIDE);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 t = xp[i];
        x[4 * i] = t[0]; x[4 * i + 1] = t[1]; x[4 * i + 2] = t[2]; x[4 * i + 3] = t[3];
      }
      const f32x4 sums = xp[8];
      const float d0 = st.c0.dot(x, sums[0], sums[1], cc & 7);
      const float d1 = st.c1.dot(x, sums[0], sums[1], cc & 7);
      acc[2 * v] += act ? d0 : 0.0f;
      acc[2 * v + 1] += act ? d1 : 0.0f;
    }
    if (m.sp == nspans - 1) {  // row group complete: reduce, epilogue, store
      float y[2 * NV];
#pragma unroll
      for (int k = 0; k < 2 * NV; ++k) {
        y[k] = wave_sum_lane63(acc[k]);
        acc[k] = 0.f;
      }
      if (lane == 63) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (v >= a.nv) continue;
          float* o = m.out + (size_t)v * m.rows;
          if (PAIR) {
            const float sg = 1.0f / (1.0f + expf(-y[2 * v]));  // silu_cpu (utils.cpp:700-707), then * up (model.cpp:735)
            stg_f32(o + m.ra, (y[2 * v] * sg) * y[2 * v + 1]);
          } else {
            stg_f32(o + m.ra, RESID ? st.r[2 * v] + y[2 * v] : y[2 * v]);
            if (m.ra + 1 < m.rows) stg_f32(o + m.ra + 1, RESID ? st.r[2 * v + 1] + y[2 * v + 1] : y[2 * v + 1]);
          }
        }
      }
    }
  };

  StepNv<TYPE, NV> cur, nxt;
  StepMeta mc, mn;
  issue(cur, mc);  // the first weight loads go out before x is staged
  stage_x_nv<NV>(a, smem);
  while (irg < total) {
    issue(nxt, mn);
    consume(cur, mc);
    cur = nxt;
    mc = mn;
  }
  consume(cur, mc);
}

template <int TYPE, int NV, int MODE>
static int gemv_nv_launch_inst(const Launch& L, const NvVec& v, int nseg, hipStream_t s) {
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemv_nv_kernel<TYPE, NV, MODE>, 160 * 1024, lds_raised));
  gemv_nv_kernel<TYPE, NV, MODE><<<L.grid, L.threads, L.lds, s>>>(L.seg[0], L.seg[1], L.seg[2], v, L.t[0], L.t[1], L.t[2], nseg);
  return 0;
}
template <int TYPE, int NV>
static int gemv_nv_launch_mode(int mode, const Launch& L, const NvVec& v, int nseg, hipStream_t s) {
  if (mode == 1) return gemv_nv_launch_inst<TYPE, NV, 1>(L, v, nseg, s);
  if (mode == 2) return gemv_nv_launch_inst<TYPE, NV, 2>(L, v, nseg, s);
  return gemv_nv_launch_inst<TYPE, NV, 0>(L, v, nseg, s);
}
template <int TYPE>
static int gemv_nv_launch_nv(int nvt, int mode, const Launch& L, const NvVec& v, int nseg, hipStream_t s) {
  if (nvt == 2) return gemv_nv_launch_mode<TYPE, 2>(mode, L, v, nseg, s);
  return gemv_nv_launch_mode<TYPE, 4>(mode, L, v, nseg, s);
}

// most vectors one launch can stage for this column count (0: not even two)
int tl_gemv_nv_capacity(int cols) {
  const size_t per = sizeof(float) * (size_t)(cols / 32) * XS_STRIDE, avail = 160 * 1024 - sizeof(float) * XS_SCRATCH;
  const int fit = (int)(avail / per);
  return fit >= 4 ? 4 : (fit >= 2 ? 2 : 0);
}

// a: as for tl_launch_gemv, with x [nv][cols], resid [nv][rows], every segment's out [nv][rows of the segment];
// segments of one launch must share one format (Q4_K / Q6_K / Q8_0); no attention-partials input.
int tl_launch_gemv_nv(const GemvArgs& a, int nv, hipStream_t s) {
  TL_TRY(gemv_setup());
  TL_REQUIRE(nv >= 1 && a.nseg >= 1 && a.nseg <= 3 && a.cols > 0 && a.cols % 32 == 0 && !a.attn_ws, "gemv_nv: bad argument");
  const int ta = a.seg[0].w.type;
  TL_REQUIRE(ta == TL_Q4_K || ta == TL_Q6_K || ta == TL_Q8_0, "gemv_nv: weight type %d has no multi-vector kernel", ta);
  for (int i = 0; i < a.nseg; ++i)
    TL_REQUIRE(a.seg[i].w.type == ta && a.seg[i].w.cols == a.cols && a.seg[i].w.rows > 0 && a.seg[i].out,
               "gemv_nv: segment %d does not match the launch", i);
  TL_REQUIRE(!a.pair || (a.nseg == 2 && a.seg[0].w.rows == a.seg[1].w.rows), "gemv_nv: pair mode needs two equal-shaped matrices");
  TL_REQUIRE(!a.resid || (a.nseg == 1 && !a.pair), "gemv_nv: residual epilogue needs a single matrix");
  const int cap = tl_gemv_nv_capacity(a.cols);
  TL_REQUIRE(cap >= 2, "gemv_nv: %d columns do not fit two staged vectors", a.cols);
  const int mode = a.pair ? 1 : (a.resid ? 2 : 0);
  for (int v0 = 0; v0 < nv; v0 += cap) {
    const int n = nv - v0 < cap ? nv - v0 : cap;
    const int nvt = n <= 2 ? 2 : 4;
    Launch L = {};
    L.lds = sizeof(float) * (XS_SCRATCH + (size_t)nvt * (a.cols / 32) * XS_STRIDE);
    if (L.lds < sizeof(float) * (XS_SCRATCH + 128)) L.lds = sizeof(float) * (XS_SCRATCH + 128);
    int total = 0;
    for (int i = 0; i < 3; ++i) {
      const GemvSeg& sg = a.seg[i < a.nseg ? i : 0];
      L.seg[i] = GemvSeg{sg.w, sg.out + (size_t)v0 * sg.w.rows};
      L.t[i] = a.pair ? sg.w.rows : (sg.w.rows + 1) / 2;
      if (i < a.nseg && !(a.pair && i > 0)) total += L.t[i];
    }
    gemv_geometry(L, total);
    if (nvt > 2 && L.threads > GEMV_MAX_THREADS / 2) L.threads = GEMV_MAX_THREADS / 2;
    const NvVec vec = {a.x + (size_t)v0 * a.cols, a.norm_w, a.resid ? a.resid + (size_t)v0 * a.seg[0].w.rows : nullptr,
                       a.eps, a.cols, n};
    int rc = 0;
    switch (ta) {
      case TL_Q4_K: rc = gemv_nv_launch_nv<TL_Q4_K>(nvt, mode, L, vec, a.nseg, s); break;
      case TL_Q6_K: rc = gemv_nv_launch_nv<TL_Q6_K>(nvt, mode, L, vec, a.nseg, s); break;
      default: rc = gemv_nv_launch_nv<TL_Q8_0>(nvt, mode, L, vec, a.nseg, s); break;
    }
    TL_TRY(rc);
    TL_HIP(hipGetLastError());
  }
  return 0;
}
