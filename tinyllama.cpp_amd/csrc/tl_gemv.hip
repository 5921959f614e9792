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
//   - row groups are dealt round-robin over all waves of the chip (consecutive groups on different CUs);
//   - weights are loaded non-temporal (each byte is read once per token);
//   - nibble/byte unpack is v_and/v_lshr + v_cvt_f32_ubyteN, FP32 FMA chains, DPP wave reduction;
//   - epilogues: plain store, residual add, or SwiGLU of a gate/up row pair.
#include <stdlib.h>

#include "tl_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XS_STRIDE = 36;   // floats per staged chunk: 32 x + sum(x[0..15]) + sum(x[16..31]) + 2 pad
constexpr int XS_SCRATCH = 64;  // floats reserved at the start of dynamic LDS for block reductions
constexpr int GEMV_MAX_THREADS = 1024;

// ---- wave reduction with DPP (no LDS traffic): the total lands in lane 63 --------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float ub0(unsigned v) { return (float)(v & 0xffu); }
__device__ __forceinline__ float ub1(unsigned v) { return (float)((v >> 8) & 0xffu); }
__device__ __forceinline__ float ub2(unsigned v) { return (float)((v >> 16) & 0xffu); }
__device__ __forceinline__ float ub3(unsigned v) { return (float)(v >> 24); }

__device__ __forceinline__ float half_to_float(unsigned h) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);  // v_cvt_f32_f16
}
// Q8 scale fields: |d| for finite negatives (quantization.cpp:85-87)
__device__ __forceinline__ float scale_field(unsigned h) {
  const float d = half_to_float(h);
  return (d < 0.0f && d > -INFINITY) ? -d : d;
}

// dot of 4 unpacked bytes with 4 consecutive x
__device__ __forceinline__ float dot4(unsigned v, const float* x, float acc) {
  acc = fmaf(ub0(v), x[0], acc);
  acc = fmaf(ub1(v), x[1], acc);
  acc = fmaf(ub2(v), x[2], acc);
  acc = fmaf(ub3(v), x[3], acc);
  return acc;
}

// ---- one 32-weight chunk in registers, format-agnostic container (9 VGPRs) ---------------------------------
//   Q4_K: a = payload, b = super-block header          Q6_K: a = ql', b = {qh'.x, qh'.y, sc pair, d}
//   Q8_0/Q8_K: a,b = 32 payload bytes, c = d           F32 is handled by a separate (cold) path
struct Raw {
  u32x4 a, b;
  unsigned c;
};

// one weight stream of a step: a matrix and a (wave-uniform) row of it
struct Stream {
  QW w;
  int row;
};

template <bool NT, typename T>
__device__ __forceinline__ T ldg(const uint8_t* base, unsigned byte_off) {
  const T* p = reinterpret_cast<const T*>(base + byte_off);
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// chunk c of the stream's row; all base addresses are wave-uniform, the per-lane part is a 32-bit byte offset
template <bool NT>
__device__ __forceinline__ void chunk_load(const Stream& s, unsigned c, Raw& r) {
  const size_t row = (size_t)s.row;
  const size_t cols = (size_t)s.w.cols;
  switch (s.w.type) {  // wave-uniform
    case TL_Q4_K:
      r.a = ldg<NT, u32x4>(s.w.p0 + row * (cols >> 1), c * 16u);
      r.b = ldg<NT, u32x4>(s.w.p1 + row * (cols >> 4), (c >> 3) * 16u);
      break;
    case TL_Q6_K: {
      r.a = ldg<NT, u32x4>(s.w.p0 + row * (cols >> 1), c * 16u);
      const u32x2 h = ldg<NT, u32x2>(s.w.p1 + row * (cols >> 2), c * 8u);
      r.b[0] = h[0];
      r.b[1] = h[1];
      r.b[2] = ldg<NT, unsigned short>(s.w.p2 + row * (cols >> 4), c * 2u);
      r.b[3] = ldg<NT, unsigned short>(s.w.p3 + row * (cols >> 7), (c >> 3) * 2u);
      break;
    }
    case TL_Q8_0:
      r.a = ldg<NT, u32x4>(s.w.p0 + row * cols, c * 32u);
      r.b = ldg<NT, u32x4>(s.w.p0 + row * cols, c * 32u + 16u);
      r.c = ldg<NT, unsigned short>(s.w.p1 + row * (cols >> 4), c * 2u);
      break;
    case TL_Q8_K:
      r.a = ldg<NT, u32x4>(s.w.p0 + row * cols, c * 32u);
      r.b = ldg<NT, u32x4>(s.w.p0 + row * cols, c * 32u + 16u);
      r.c = ldg<NT, unsigned short>(s.w.p1 + row * (cols >> 7), (c >> 3) * 2u);
      break;
    default: break;
  }
}

// (d*sc) * sum(q*x) - (dmin*m) * sum(x)  ==  sum((d*sc*q - dmin*m) * x)   (quantization.cpp:200-215)
__device__ __forceinline__ float dot_q4k(const Raw& r, const float* x, float s0, float s1, int j) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned lo = r.a[w] & 0x0F0F0F0Fu, hi = (r.a[w] >> 4) & 0x0F0F0F0Fu;
    a0 = dot4(lo, x + 8 * w, a0);
    a1 = dot4(hi, x + 8 * w + 4, a1);
  }
  const float d = half_to_float(r.b[0] & 0xffffu), dmin = half_to_float(r.b[0] >> 16);
  const unsigned sc = ((r.b[1] >> (4 * j)) & 15u) | (((r.b[3] >> (2 * j)) & 3u) << 4);
  const unsigned mn = ((r.b[2] >> (4 * j)) & 15u) | (((r.b[3] >> (16 + 2 * j)) & 3u) << 4);
  return (d * (float)sc) * (a0 + a1) - (dmin * (float)mn) * (s0 + s1);
}

// d*sc*(q6-32) summed per 16-weight sub-block (quantization.cpp:255-263)
__device__ __forceinline__ float dot_q6k(const Raw& r, const float* x, float s0, float s1) {
  float acc[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const unsigned h = r.b[v];
    const unsigned l0 = r.a[2 * v], l1 = r.a[2 * v + 1];
    const unsigned c0 = (l0 & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
    const unsigned c1 = ((l0 >> 4) & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
    const unsigned c2 = (l1 & 0x0F0F0F0Fu) | (h & 0x30303030u);
    const unsigned c3 = ((l1 >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
    float a = 0.f, b = 0.f;
    a = dot4(c0, x + 16 * v, a);
    b = dot4(c1, x + 16 * v + 4, b);
    a = dot4(c2, x + 16 * v + 8, a);
    b = dot4(c3, x + 16 * v + 12, b);
    acc[v] = a + b;
  }
  const float d = half_to_float(r.b[3]);
  const float sc0 = (float)(int)(signed char)(r.b[2] & 0xffu), sc1 = (float)(int)(signed char)((r.b[2] >> 8) & 0xffu);
  return (d * sc0) * (acc[0] - 32.0f * s0) + (d * sc1) * (acc[1] - 32.0f * s1);
}

// |d| * sum(q*x): bytes are biased by +128 for the unsigned byte->float convert, folded back with sum(x)
__device__ __forceinline__ float dot_q8(const Raw& r, const float* x, float s0, float s1) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    a0 = dot4(r.a[w] ^ 0x80808080u, x + 4 * w, a0);
    a1 = dot4(r.b[w] ^ 0x80808080u, x + 16 + 4 * w, a1);
  }
  return scale_field(r.c) * ((a0 + a1) - 128.0f * (s0 + s1));
}

__device__ __forceinline__ float chunk_dot(int type, const Raw& r, const float* x, float s0, float s1, int j) {
  switch (type) {  // wave-uniform
    case TL_Q4_K: return dot_q4k(r, x, s0, s1, j);
    case TL_Q6_K: return dot_q6k(r, x, s0, s1);
    default: return dot_q8(r, x, s0, s1);
  }
}

// the vector-side arguments of a launch (the segments travel as separate kernel parameters)
struct GemvVec {
  const float* x;
  const float* norm_w;
  const float* resid;
  float eps;
  int cols;
};

// ---- x staging: (optional RMSNorm, utils.cpp:648-673) -> strided LDS image + per-chunk half sums --------------
__device__ __forceinline__ void stage_x(const GemvVec a, float* smem) {
  float* xs = smem + XS_SCRATCH;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n4 = a.cols >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(a.norm_w);
  // the first two float4 of a thread stay in registers across the reduction (covers cols <= 8 * blockDim)
  const int i0 = tid, i1 = tid + nthr;
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
  if (i0 < n4) v0 = x4[i0];
  if (i1 < n4) v1 = x4[i1];
  float nf = 1.0f;
  if (a.norm_w) {
    float ssq = (v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3]);
    ssq += (v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]);
    for (int i = tid + 2 * nthr; i < n4; i += nthr) {
      const f32x4 t = x4[i];
      ssq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
    ssq = wave_sum_all(ssq);
    if ((tid & 63) == 0) smem[tid >> 6] = ssq;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) tot += smem[w];
    const float e = a.eps < 1e-5f ? 1e-5f : a.eps;  // SAFE_MAX(eps, MIN_NORM_EPS)
    nf = 1.0f / sqrtf(tot / (float)a.cols + e);
  }
  auto put = [&](int i, f32x4 t) {
    if (a.norm_w) {
      const f32x4 g = w4[i];
      t[0] = t[0] * nf * g[0]; t[1] = t[1] * nf * g[1]; t[2] = t[2] * nf * g[2]; t[3] = t[3] * nf * g[3];
    }
    const int idx = i << 2;
    float* dst = xs + (idx >> 5) * XS_STRIDE;
    *reinterpret_cast<f32x4*>(dst + (idx & 31)) = t;
    // 4 consecutive threads cover 16 consecutive x: their sum is one of the chunk's two half sums
    float s = (t[0] + t[1]) + (t[2] + t[3]);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if ((i & 3) == 0) dst[32 + ((i >> 2) & 1)] = s;
  };
  // n4 is a multiple of 8 and blockDim of 64, so quads of threads are all-in or all-out of every branch
  if (i0 < n4) put(i0, v0);
  if (i1 < n4) put(i1, v1);
  for (int i = tid + 2 * nthr; i < n4; i += nthr) put(i, x4[i]);
  __syncthreads();
}

// ---- the kernel ------------------------------------------------------------------------------------------------
struct Step {
  Raw r0, r1;
};

// the segments of a launch as plain scalars (no array: a dynamically indexed by-value kernel argument is copied to
// scratch and drags every address computation into VGPRs)
struct Segs {
  QW w0, w1, w2;
  float *o0, *o1, *o2;
  const float* resid;
  int t0, t1;
};

// the two streams of row group `rg`: rows (2*lt, 2*lt+1) of one segment, or row lt of the gate/up pair
template <bool PAIR>
__device__ __forceinline__ void streams_of(const Segs& g, int rg, Stream& s0, Stream& s1, float*& out, int& row0) {
  if (PAIR) {
    s0.w = g.w0;
    s1.w = g.w1;
    s0.row = s1.row = row0 = rg;
    out = g.o0;
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
  auto picki = [&](int x0, int x1, int x2) {
    return (int)(((unsigned)x0 & (unsigned)m0) | ((unsigned)x1 & (unsigned)m1) | ((unsigned)x2 & (unsigned)m2));
  };
  s0.w.p0 = pick(g.w0.p0, g.w1.p0, g.w2.p0);
  s0.w.p1 = pick(g.w0.p1, g.w1.p1, g.w2.p1);
  s0.w.p2 = pick(g.w0.p2, g.w1.p2, g.w2.p2);
  s0.w.p3 = pick(g.w0.p3, g.w1.p3, g.w2.p3);
  s0.w.rows = picki(g.w0.rows, g.w1.rows, g.w2.rows);
  s0.w.type = picki(g.w0.type, g.w1.type, g.w2.type);
  s0.w.cols = g.w0.cols;
  out = reinterpret_cast<float*>(const_cast<uint8_t*>(pick(g.o0, g.o1, g.o2)));
  s1.w = s0.w;
  row0 = 2 * lt;
  s0.row = row0;
  s1.row = row0 + 1 < s0.w.rows ? row0 + 1 : s0.w.rows - 1;  // odd tail: computed twice, stored once
}

template <bool NT>
__device__ __forceinline__ void step_load(Step& st, const Stream& s0, const Stream& s1, int span, int nchunks,
                                          int lane) {
  const int chunk = (span << 6) + lane;
  const unsigned cc = chunk < nchunks ? chunk : nchunks - 1;
  chunk_load<NT>(s0, cc, st.r0);
  chunk_load<NT>(s1, cc, st.r1);
}

__device__ __forceinline__ void step_dot(const Step& st, int type0, int type1, int span, int nchunks, int lane,
                                         const float* xs, float& acc0, float& acc1) {
  const int chunk = (span << 6) + lane;
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
  const float d0 = chunk_dot(type0, st.r0, x, sums[0], sums[1], cc & 7);
  const float d1 = chunk_dot(type1, st.r1, x, sums[0], sums[1], cc & 7);
  acc0 += act ? d0 : 0.0f;
  acc1 += act ? d1 : 0.0f;
}

// ROLE only names the instantiation (tl_internal.h GemvRole) so that profiles separate the per-layer launches.
template <bool PAIR, bool NT, int ROLE>
__global__ __launch_bounds__(GEMV_MAX_THREADS) void gemv_kernel(GemvSeg sg0, GemvSeg sg1, GemvSeg sg2, GemvVec a,
                                                                int t0, int t1, int t2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float* xs = smem + XS_SCRATCH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunks = a.cols >> 5;
  const int nspans = (nchunks + 63) >> 6;
  const int total = PAIR ? t0 : (t0 + t1 + t2);
  const int stride = gridDim.x * (blockDim.x >> 6);
  const Segs g = {sg0.w, sg1.w, sg2.w, sg0.out, sg1.out, sg2.out, a.resid, t0, t1};

  // wave-uniform cursor: row group rg (streams s0/s1), span sp
  int rg = wave * gridDim.x + blockIdx.x, sp = 0;
  Stream s0, s1;
  float* out = nullptr;
  int row0 = 0;
  Step bufA, bufB;
  const bool any = rg < total;
  if (any) {  // first step's loads go out before x is staged: the HBM round trip overlaps the prologue
    streams_of<PAIR>(g, rg, s0, s1, out, row0);
    step_load<NT>(bufA, s0, s1, 0, nchunks, lane);
  }
  stage_x(a, smem);
  if (!any) return;

  float acc0 = 0.f, acc1 = 0.f;
  // one step: prefetch the next step into `nxt`, consume `cur`; returns false when the wave is done
  auto step = [&](Step& cur, Step& nxt) -> bool {
    int nsp = sp + 1, nrg = rg;
    if (nsp == nspans) {
      nsp = 0;
      nrg = rg + stride;
    }
    Stream n0 = s0, n1 = s1;
    float* nout = out;
    int nrow0 = row0;
    const bool more = nrg < total;
    if (more) {
      if (nrg != rg) streams_of<PAIR>(g, nrg, n0, n1, nout, nrow0);
      step_load<NT>(nxt, n0, n1, nsp, nchunks, lane);
    }
    step_dot(cur, s0.w.type, s1.w.type, sp, nchunks, lane, xs, acc0, acc1);
    if (sp == nspans - 1) {  // row group complete: reduce, epilogue, store
      const float y0 = wave_sum_lane63(acc0);
      const float y1 = wave_sum_lane63(acc1);
      if (lane == 63) {
        if (PAIR) {
          const float sg = 1.0f / (1.0f + expf(-y0));  // silu_cpu (utils.cpp:700-707), then * up (model.cpp:735)
          out[row0] = (y0 * sg) * y1;
        } else {
          out[row0] = g.resid ? g.resid[row0] + y0 : y0;
          if (row0 + 1 < s0.w.rows) out[row0 + 1] = g.resid ? g.resid[row0 + 1] + y1 : y1;
        }
      }
      acc0 = 0.f;
      acc1 = 0.f;
    }
    rg = nrg;
    sp = nsp;
    s0 = n0;
    s1 = n1;
    out = nout;
    row0 = nrow0;
    return more;
  };
  for (;;) {
    if (!step(bufA, bufB)) break;
    if (!step(bufB, bufA)) break;
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
static int g_nt = 1, g_max_waves = 16;

static int gemv_setup() {
  if (g_gemv_setup_done) return 0;
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TL_HIP(hipGetDeviceProperties(&prop, dev));
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("TL_GEMV_NT")) g_nt = atoi(e) != 0;  // tuning knobs (experiments only)
  if (const char* e = getenv("TL_GEMV_WAVES")) {
    const int w = atoi(e);
    if (w >= 1 && w <= 16) g_max_waves = w;
  }
  g_gemv_setup_done = 1;
  return 0;
}

template <bool PAIR, bool NT, int ROLE>
static int gemv_launch_one(const GemvArgs& a, int grid, int threads, size_t lds, const int* t, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_kernel<PAIR, NT, ROLE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const GemvVec v = {a.x, a.norm_w, a.resid, a.eps, a.cols};
  gemv_kernel<PAIR, NT, ROLE><<<grid, threads, lds, s>>>(a.seg[0], a.seg[1], a.seg[2], v, t[0], t[1], t[2]);
  return 0;
}

template <bool NT>
static int gemv_launch_role(const GemvArgs& a, int grid, int threads, size_t lds, const int* t, hipStream_t s) {
  if (a.pair) return gemv_launch_one<true, NT, GEMV_GATE_UP>(a, grid, threads, lds, t, s);
  switch (a.role) {
    case GEMV_QKV: return gemv_launch_one<false, NT, GEMV_QKV>(a, grid, threads, lds, t, s);
    case GEMV_ATTN_OUT: return gemv_launch_one<false, NT, GEMV_ATTN_OUT>(a, grid, threads, lds, t, s);
    case GEMV_DOWN: return gemv_launch_one<false, NT, GEMV_DOWN>(a, grid, threads, lds, t, s);
    case GEMV_LM_HEAD: return gemv_launch_one<false, NT, GEMV_LM_HEAD>(a, grid, threads, lds, t, s);
    default: return gemv_launch_one<false, NT, GEMV_GENERIC>(a, grid, threads, lds, t, s);
  }
}

int tl_launch_gemv(const GemvArgs& a, hipStream_t s) {
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
  bool any_f32 = false;
  for (int i = 0; i < a.nseg; ++i) any_f32 |= a.seg[i].w.type == TL_F32;
  if (any_f32) {
    TL_REQUIRE(!a.pair && !a.norm_w, "gemv: F32 matrices are not supported in fused launches");
    for (int i = 0; i < a.nseg; ++i) {
      TL_REQUIRE(a.seg[i].w.type == TL_F32, "gemv: mixed F32 / quantized segments are not supported");
      const QW& w = a.seg[i].w;
      gemv_f32_kernel<<<(w.rows + 3) / 4, 256, 0, s>>>(reinterpret_cast<const float*>(w.p0), a.x, a.resid,
                                                        a.seg[i].out, w.rows, w.cols);
    }
    TL_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = sizeof(float) * (XS_SCRATCH + (size_t)(a.cols / 32) * XS_STRIDE);
  TL_REQUIRE(lds <= 160 * 1024, "gemv: cols %d needs %zu B of LDS (> 160 KiB)", a.cols, lds);
  int t[3] = {0, 0, 0};
  for (int i = 0; i < a.nseg; ++i) t[i] = a.pair ? a.seg[i].w.rows : (a.seg[i].w.rows + 1) / 2;
  const int total = a.pair ? t[0] : t[0] + t[1] + t[2];
  // one workgroup per CU, all resident at once; small problems use fewer CUs / fewer waves per workgroup
  int grid = total < g_num_cus ? total : g_num_cus;
  int waves = (total + grid - 1) / grid;
  if (waves > g_max_waves) waves = g_max_waves;
  if (waves < 4) waves = 4;  // the x staging pass wants a few waves
  const int threads = waves * 64;
  TL_TRY(g_nt ? gemv_launch_role<true>(a, grid, threads, lds, t, s) : gemv_launch_role<false>(a, grid, threads, lds, t, s));
  TL_HIP(hipGetLastError());
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
