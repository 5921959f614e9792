// tl_gemv.hip — block-dequant GEMV over the plane layout (tl_internal.h): y[r] = sum_c deq(W[r][c]) * x[c].
//
// What it replaces: the reference never multiplies by quantized weights on a GPU — it expands every layer
// to FP32 on the host and calls cublasSgemm with N=1 (weight_management.cpp:199-435, cuda_kernels.cu:291-314);
// its CPU path is dequantize-block + Kahan-double dot (utils.cpp:293-397, 784-888). The numerics target is the
// CPU path; this kernel accumulates in FP32 (tolerance stated in tests/test_gpu_gemv.py).
//
// Shape of the kernel (HBM-bound; no MFMA on purpose):
//   - a lane owns one 32-weight chunk per 2048-column span: one 16 B/lane load = 1 KiB contiguous per wave
//     (Q4_K/Q6_K payload), plus the shared 16 B scale header of its super-block (8 lanes, same address);
//   - a wave owns R rows at a time and walks the spans; all R*(payload+header) loads are issued before the
//     first dot so that >= R KiB per wave are in flight; weights are loaded non-temporal (read once);
//   - x is staged once per workgroup into LDS in 36-float strided chunks (32 values + the two 16-element
//     partial sums used to fold the "-min" / "-32" / "-128" offsets), which makes the 16 B/lane LDS reads
//     bank-conflict free; an optional RMSNorm of x is fused into that staging pass;
//   - nibble/byte unpack is v_and/v_lshr + v_cvt_f32_ubyteN, FP32 FMA chains, DPP/permute wave reduction;
//   - epilogues: plain store, residual add, or SwiGLU of a gate/up row pair.
#include "tl_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XS_STRIDE = 36;    // floats per staged chunk: 32 x + sum(x[0..15]) + sum(x[16..31]) + 2 pad
constexpr int XS_SCRATCH = 64;   // floats reserved at the start of dynamic LDS for block reductions
constexpr int GEMV_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
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

template <int TYPE>
struct Chunk;

template <>
struct Chunk<TL_Q4_K> {
  u32x4 q, h;
  __device__ __forceinline__ void load(const QW& w, size_t c) {
    q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w.p0) + c);
    h = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w.p1) + (c >> 3));
  }
  // (d*sc) * sum(q*x) - (dmin*m) * sum(x)   ==  sum((d*sc*q - dmin*m) * x)   (quantization.cpp:200-215)
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int j) const {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned lo = q[w] & 0x0F0F0F0Fu, hi = (q[w] >> 4) & 0x0F0F0F0Fu;
      a0 = dot4(lo, x + 8 * w, a0);
      a1 = dot4(hi, x + 8 * w + 4, a1);
    }
    const float d = half_to_float(h.x & 0xffffu), dmin = half_to_float(h.x >> 16);
    const unsigned sc = ((h.y >> (4 * j)) & 15u) | (((h.w >> (2 * j)) & 3u) << 4);
    const unsigned mn = ((h.z >> (4 * j)) & 15u) | (((h.w >> (16 + 2 * j)) & 3u) << 4);
    return (d * (float)sc) * (a0 + a1) - (dmin * (float)mn) * (s0 + s1);
  }
};

template <>
struct Chunk<TL_Q6_K> {
  u32x4 q;
  u32x2 hq;
  unsigned sc2, dh;
  __device__ __forceinline__ void load(const QW& w, size_t c) {
    q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w.p0) + c);
    hq = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(w.p1) + c);
    sc2 = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(w.p2) + c);
    dh = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(w.p3) + (c >> 3));
  }
  // d*sc*(q6-32) summed per 16-weight sub-block (quantization.cpp:255-263)
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int) const {
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const unsigned h = hq[v];
      const unsigned l0 = q[2 * v], l1 = q[2 * v + 1];
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
    const float d = half_to_float(dh);
    const float sc0 = (float)(int)(signed char)(sc2 & 0xffu), sc1 = (float)(int)(signed char)(sc2 >> 8);
    return (d * sc0) * (acc[0] - 32.0f * s0) + (d * sc1) * (acc[1] - 32.0f * s1);
  }
};

template <int TYPE>
struct ChunkQ8 {
  u32x4 a, b;
  unsigned dh;
  __device__ __forceinline__ void load(const QW& w, size_t c) {
    const u32x4* p = reinterpret_cast<const u32x4*>(w.p0) + 2 * c;
    a = __builtin_nontemporal_load(p);
    b = __builtin_nontemporal_load(p + 1);
    dh = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(w.p1) + (TYPE == TL_Q8_0 ? c : (c >> 3)));
  }
  // |d| * sum(q*x): bytes are biased by +128 for the unsigned byte->float convert, folded back with sum(x)
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int) const {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a0 = dot4(a[w] ^ 0x80808080u, x + 4 * w, a0);
      a1 = dot4(b[w] ^ 0x80808080u, x + 16 + 4 * w, a1);
    }
    return scale_field(dh) * ((a0 + a1) - 128.0f * (s0 + s1));
  }
};
template <>
struct Chunk<TL_Q8_0> : ChunkQ8<TL_Q8_0> {};
template <>
struct Chunk<TL_Q8_K> : ChunkQ8<TL_Q8_K> {};

template <>
struct Chunk<TL_F32> {
  f32x4 v[8];
  __device__ __forceinline__ void load(const QW& w, size_t c) {
    const f32x4* p = reinterpret_cast<const f32x4*>(w.p0) + 8 * c;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(p + i);
  }
  __device__ __forceinline__ float dot(const float* x, float, float, int) const {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a = fmaf(v[i][0], x[4 * i], a);
      a = fmaf(v[i][1], x[4 * i + 1], a);
      a = fmaf(v[i][2], x[4 * i + 2], a);
      a = fmaf(v[i][3], x[4 * i + 3], a);
    }
    return a;
  }
};

// R rows [row0, row0+R) of one matrix against the staged x: per-lane partial sums in acc[R]
template <int TYPE, int R>
__device__ __forceinline__ void rows_dot(const QW& w, int row0, const float* xs, int nchunks, int lane,
                                         float (&acc)[R]) {
  const int nspans = (nchunks + 63) >> 6;
  size_t rbase[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = (row0 + r < w.rows) ? row0 + r : w.rows - 1;  // tail rows are computed but not stored
    rbase[r] = (size_t)row * nchunks;
  }
  for (int s = 0; s < nspans; ++s) {
    const int chunk = (s << 6) + lane;
    const bool act = chunk < nchunks;
    const int cc = act ? chunk : nchunks - 1;
    Chunk<TYPE> ck[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ck[r].load(w, rbase[r] + cc);
    float x[32];
    const f32x4* xp = reinterpret_cast<const f32x4*>(xs + cc * XS_STRIDE);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 t = xp[i];
      x[4 * i] = t[0]; x[4 * i + 1] = t[1]; x[4 * i + 2] = t[2]; x[4 * i + 3] = t[3];
    }
    const f32x4 sums = xp[8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float d = ck[r].dot(x, sums[0], sums[1], cc & 7);
      acc[r] += act ? d : 0.0f;
    }
  }
}

template <int R>
__device__ __forceinline__ void rows_dot_any(const QW& w, int row0, const float* xs, int nchunks, int lane,
                                             float (&acc)[R]) {
  switch (w.type) {  // wave-uniform
    case TL_Q4_K: rows_dot<TL_Q4_K, R>(w, row0, xs, nchunks, lane, acc); break;
    case TL_Q6_K: rows_dot<TL_Q6_K, R>(w, row0, xs, nchunks, lane, acc); break;
    case TL_Q8_0: rows_dot<TL_Q8_0, R>(w, row0, xs, nchunks, lane, acc); break;
    case TL_Q8_K: rows_dot<TL_Q8_K, R>(w, row0, xs, nchunks, lane, acc); break;
    default: rows_dot<TL_F32, R>(w, row0, xs, nchunks, lane, acc); break;
  }
}

// stage x (optionally RMS-normalised, utils.cpp:648-673) into the strided LDS image + per-chunk half sums
__device__ __forceinline__ void stage_x(const GemvArgs& a, float* smem) {
  float* xs = smem + XS_SCRATCH;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int n4 = a.cols >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
  float nf = 1.0f;
  if (a.norm_w) {
    float ssq = 0.f;
    for (int i = tid; i < n4; i += nthreads) {
      const f32x4 v = x4[i];
      ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ssq = wave_sum(ssq);
    if ((tid & 63) == 0) smem[tid >> 6] = ssq;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (nthreads >> 6); ++i) tot += smem[i];
    const float e = a.eps < 1e-5f ? 1e-5f : a.eps;  // SAFE_MAX(eps, MIN_NORM_EPS)
    nf = 1.0f / sqrtf(tot / (float)a.cols + e);
  }
  const f32x4* w4 = reinterpret_cast<const f32x4*>(a.norm_w);
  for (int i = tid; i < n4; i += nthreads) {
    f32x4 v = x4[i];
    if (a.norm_w) {
      const f32x4 g = w4[i];
      v[0] = v[0] * nf * g[0]; v[1] = v[1] * nf * g[1]; v[2] = v[2] * nf * g[2]; v[3] = v[3] * nf * g[3];
    }
    const int idx = i << 2;
    *reinterpret_cast<f32x4*>(xs + (idx >> 5) * XS_STRIDE + (idx & 31)) = v;
  }
  __syncthreads();
  const int nchunks = a.cols >> 5;
  for (int c = tid; c < nchunks; c += nthreads) {
    const f32x4* p = reinterpret_cast<const f32x4*>(xs + c * XS_STRIDE);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 u = p[i], v = p[4 + i];
      s0 += (u[0] + u[1]) + (u[2] + u[3]);
      s1 += (v[0] + v[1]) + (v[2] + v[3]);
    }
    f32x4 sums = {s0, s1, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(xs + c * XS_STRIDE + 32) = sums;
  }
  __syncthreads();
}

// ROLE only names the instantiation (tl_internal.h GemvRole) so that profiles separate the per-layer launches.
template <int R, int ROLE>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs a, int t0, int t1, int t2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  stage_x(a, smem);
  const float* xs = smem + XS_SCRATCH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int nchunks = a.cols >> 5;
  const int total = a.pair ? t0 : (t0 + t1 + t2);
  for (int task = blockIdx.x * nwaves + wave; task < total; task += gridDim.x * nwaves) {
    int seg = 0, lt = task;
    if (!a.pair) {
      if (lt >= t0) { lt -= t0; seg = 1; }
      if (seg == 1 && lt >= t1) { lt -= t1; seg = 2; }
    }
    const int row0 = lt * R;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    if (a.pair) {
      float accu[R];
#pragma unroll
      for (int r = 0; r < R; ++r) accu[r] = 0.f;
      rows_dot_any<R>(a.seg[0].w, row0, xs, nchunks, lane, acc);
      rows_dot_any<R>(a.seg[1].w, row0, xs, nchunks, lane, accu);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float g = wave_sum(acc[r]), u = wave_sum(accu[r]);
        if (lane == 0 && row0 + r < a.seg[0].w.rows) {
          const float sg = 1.0f / (1.0f + expf(-g));  // silu_cpu (utils.cpp:700-707) then * up (model.cpp:735)
          a.seg[0].out[row0 + r] = (g * sg) * u;
        }
      }
    } else {
      // seg is wave-uniform; pick the matrix without indexing the by-value struct dynamically
      const QW& w = seg == 0 ? a.seg[0].w : (seg == 1 ? a.seg[1].w : a.seg[2].w);
      float* out = seg == 0 ? a.seg[0].out : (seg == 1 ? a.seg[1].out : a.seg[2].out);
      rows_dot_any<R>(w, row0, xs, nchunks, lane, acc);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float y = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < w.rows) out[row0 + r] = a.resid ? a.resid[row0 + r] + y : y;
      }
    }
  }
}

static int g_gemv_attr_done = 0;
static int g_gemv_R_override = -1;
static int g_num_cus = 256;

template <int R, int ROLE>
static int gemv_launch_one(const GemvArgs& a, int grid, size_t lds, const int* t, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_kernel<R, ROLE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  gemv_kernel<R, ROLE><<<grid, GEMV_THREADS, lds, s>>>(a, t[0], t[1], t[2]);
  return 0;
}

template <int R>
static int gemv_launch_role(const GemvArgs& a, int grid, size_t lds, const int* t, hipStream_t s) {
  switch (a.role) {
    case GEMV_QKV: return gemv_launch_one<R, GEMV_QKV>(a, grid, lds, t, s);
    case GEMV_ATTN_OUT: return gemv_launch_one<R, GEMV_ATTN_OUT>(a, grid, lds, t, s);
    case GEMV_GATE_UP: return gemv_launch_one<R, GEMV_GATE_UP>(a, grid, lds, t, s);
    case GEMV_DOWN: return gemv_launch_one<R, GEMV_DOWN>(a, grid, lds, t, s);
    case GEMV_LM_HEAD: return gemv_launch_one<R, GEMV_LM_HEAD>(a, grid, lds, t, s);
    default: return gemv_launch_one<R, GEMV_GENERIC>(a, grid, lds, t, s);
  }
}

static int gemv_setup() {
  if (g_gemv_attr_done) return 0;
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TL_HIP(hipGetDeviceProperties(&prop, dev));
  g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const char* e = getenv("TL_GEMV_R");
  if (e) g_gemv_R_override = atoi(e);
  g_gemv_attr_done = 1;
  return 0;
}

int tl_launch_gemv(const GemvArgs& a, hipStream_t s) {
  TL_TRY(gemv_setup());
  TL_REQUIRE(a.nseg >= 1 && a.nseg <= 3, "gemv: nseg %d", a.nseg);
  TL_REQUIRE(a.cols > 0 && a.cols % 32 == 0, "gemv: cols %d must be a positive multiple of 32", a.cols);
  int rows_total = 0;
  for (int i = 0; i < a.nseg; ++i) {
    TL_REQUIRE(a.seg[i].w.cols == a.cols, "gemv: segment %d has cols %d, expected %d", i, a.seg[i].w.cols, a.cols);
    TL_REQUIRE(a.seg[i].w.rows > 0, "gemv: segment %d has no rows", i);
    rows_total += a.seg[i].w.rows;
  }
  if (a.pair) {
    TL_REQUIRE(a.nseg == 2 && a.seg[0].w.rows == a.seg[1].w.rows, "gemv: pair mode needs two equal-shaped matrices");
    rows_total = a.seg[0].w.rows;
  }
  TL_REQUIRE(!a.resid || a.nseg == 1, "gemv: residual epilogue needs a single matrix");
  const size_t lds = sizeof(float) * (XS_SCRATCH + (size_t)(a.cols / 32) * XS_STRIDE);
  TL_REQUIRE(lds <= 160 * 1024, "gemv: cols %d needs %zu B of LDS (> 160 KiB)", a.cols, lds);
  // rows per wave: more rows per wave = more bytes in flight per wave, fewer waves; keep >= ~8 wave-tasks per CU
  int R = rows_total >= 8 * 1024 ? 2 : 1;  // R=4 is compiled but needs 256 VGPRs today (opt-in via TL_GEMV_R)
  if (a.pair && R > 2) R = 2;
  if (g_gemv_R_override == 1 || g_gemv_R_override == 2 || g_gemv_R_override == 4) R = g_gemv_R_override;
  int t[3] = {0, 0, 0};
  for (int i = 0; i < a.nseg; ++i) t[i] = (a.seg[i].w.rows + R - 1) / R;
  const int total = a.pair ? t[0] : t[0] + t[1] + t[2];
  const int waves = GEMV_THREADS / 64;
  int blocks_per_cu = (int)((160 * 1024) / lds);
  if (blocks_per_cu > 8) blocks_per_cu = 8;
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  int grid = (total + waves - 1) / waves;
  const int cap = g_num_cus * blocks_per_cu;
  if (grid > cap) grid = cap;
  switch (R) {
    case 1: TL_TRY((gemv_launch_role<1>(a, grid, lds, t, s))); break;
    case 2: TL_TRY((gemv_launch_role<2>(a, grid, lds, t, s))); break;
    default: TL_TRY((gemv_launch_role<4>(a, grid, lds, t, s))); break;
  }
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
