// tl_gemm.hip — batched-prefill GEMM over the quantized plane layout on the matrix cores:
//   Y[T][rows] (+ resid) = X[T][cols] . W[rows][cols]^T
//
// What it replaces: smart_gemm_batch_cuda -> cublasSgemm / cublasGemmEx(BF16) over FP32-expanded weights
// (model.cpp:2141-2249, cuda_kernels.cu:2541-2701); the numerics target is the CPU path, matmul_f32_f32_batch_cpu
// (utils.cpp:730-782: per token the Kahan-double dot of the exactly dequantized row).
//
// This is the one dense contraction of the path, so it runs on MFMA (v_mfma_f32_32x32x16_bf16, FP32 accumulate):
//   - the weights never exist in HBM in expanded form: a workgroup loads the raw plane bytes of a 64-row x 32-column
//     tile, dequantizes them in registers with the reference's exact FP32 expression (the same values as
//     tl_dequantize_rows), and writes them to LDS as bf16;
//   - precision: a bf16 has 8 significant bits, an FP32 weight or activation 24. Both operands are therefore split
//     into up to three bf16 "planes" v = p0 + p1 + p2 (p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1); the
//     subtractions are exact), and a product is assembled from 1, 3 or 6 MFMAs:
//         terms = 1:  x0.w0                                  (plain bf16, error ~2^-9 per product)
//         terms = 3:  + x0.w1 + x1.w0                        (~2^-17)
//         terms = 6:  + x1.w1 + x0.w2 + x2.w0                (~2^-24: FP32-grade, the default)
//     at 1/6 of the bf16 rate the 6-term form is still ~2.7x the chip's FP32 MFMA/VALU rate;
//   - the activation planes are produced ONCE per GEMM input by the producer kernel (RMSNorm, SwiGLU or a plain
//     split, below), not by every workgroup that consumes them;
//   - tile: 128 tokens x 64 weight rows x 32 columns per step; XOR-swizzled 64-B LDS rows (fragment reads and staging
//     writes bank-conflict free), two LDS buffers and one barrier per step. Default kernel: wave-specialised, 8 waves =
//     4 producers (global loads three steps ahead in registers, dequantize + split, LDS writes) + 4 consumers (fragment
//     reads + MFMAs, 64 tokens x 32 rows each), gemm_q_ws_kernel; the unified 4-wave kernel (gemm_q_kernel, also with
//     128-row tiles) is kept for A/B runs (TL_GEMM_WS=0): it measured 13 % slower end to end;
//   - workgroup -> tile mapping is XCD-aware: the 8 XCDs (round-robin over blockIdx) each walk a contiguous range of
//     weight-row tiles of ONE token panel, so a panel's activation planes (<= 3 MB) stay in that XCD's L2.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "tl_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int GB_N = 128;   // tokens per workgroup tile
constexpr int GB_K = 32;    // columns per step = one 32-weight chunk of the plane layout
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ float h2f(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
__device__ __forceinline__ float h2f_scale(unsigned h) {  // |d| for finite negatives (quantization.cpp:85-87)
  const float d = h2f(h);
  return (d < 0.0f && d > -INFINITY) ? -d : d;
}

// ---- 8 consecutive weights (quarter `q` of 32-weight chunk `c` of `row`): raw bytes, then exact FP32 values --------
struct Raw8 {
  unsigned a, b;  // payload
  u32x4 h;        // Q4_K super-block header / scale fields
};

template <int TYPE>
__device__ __forceinline__ Raw8 load8(const QW& w, int row, int c, int q) {
  Raw8 r = {};
  const size_t R = (size_t)row, cols = (size_t)w.cols;
  if (TYPE == TL_Q4_K) {
    r.a = *reinterpret_cast<const unsigned*>(w.p0 + R * (cols >> 1) + (size_t)c * 16 + q * 4);
    r.h = *reinterpret_cast<const u32x4*>(w.p1 + R * (cols >> 4) + (size_t)(c >> 3) * 16);
  } else if (TYPE == TL_Q6_K) {
    r.a = *reinterpret_cast<const unsigned*>(w.p0 + R * (cols >> 1) + (size_t)c * 16 + q * 4);
    r.b = *reinterpret_cast<const unsigned*>(w.p1 + R * (cols >> 2) + (size_t)c * 8 + (q >> 1) * 4);
    r.h[0] = *reinterpret_cast<const uint8_t*>(w.p2 + R * (cols >> 4) + (size_t)c * 2 + (q >> 1));
    r.h[1] = *reinterpret_cast<const uint16_t*>(w.p3 + (R * (cols >> 8) + (size_t)(c >> 3)) * 2);
  } else {  // Q8_0 / Q8_K
    const u32x2 v = *reinterpret_cast<const u32x2*>(w.p0 + R * cols + (size_t)c * 32 + q * 8);
    r.a = v[0];
    r.b = v[1];
    const size_t di = TYPE == TL_Q8_0 ? R * (cols >> 5) + (size_t)c : R * (cols >> 8) + (size_t)(c >> 3);
    r.h[0] = *reinterpret_cast<const uint16_t*>(w.p1 + di * 2);
  }
  return r;
}

// the reference's expressions with every product / difference rounded separately (quantization.cpp:171-266,
// 1009-1052): the values are bit-identical to tl_dequantize_rows
template <int TYPE>
__device__ __forceinline__ void deq8(const Raw8& r, int c, int q, float* y) {
#pragma clang fp contract(off)
  if (TYPE == TL_Q4_K) {
    const int j = c & 7;
    const float d = h2f(r.h[0] & 0xffffu), dmin = h2f(r.h[0] >> 16);
    const unsigned sc = ((r.h[1] >> (4 * j)) & 15u) | (((r.h[3] >> (2 * j)) & 3u) << 4);
    const unsigned mn = ((r.h[2] >> (4 * j)) & 15u) | (((r.h[3] >> (16 + 2 * j)) & 3u) << 4);
    const float d1 = __fmul_rn(d, (float)sc), m1 = __fmul_rn(dmin, (float)mn);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned byte = (r.a >> (8 * t)) & 0xffu;
      y[t] = __fsub_rn(__fmul_rn(d1, (float)(byte & 15u)), m1);
      y[4 + t] = __fsub_rn(__fmul_rn(d1, (float)(byte >> 4)), m1);
    }
  } else if (TYPE == TL_Q6_K) {
    const float ds = __fmul_rn(h2f(r.h[1]), (float)(int)(signed char)(r.h[0] & 0xffu));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int part = i >> 2, t = i & 3, g = 2 * (q & 1) + part;
      const unsigned lo = (r.a >> (8 * t + 4 * part)) & 15u;
      const unsigned hi = (r.b >> (8 * t + 2 * g)) & 3u;
      y[i] = __fmul_rn(ds, (float)((int)(lo | (hi << 4)) - 32));
    }
  } else {
    const float d = h2f_scale(r.h[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned v = i < 4 ? r.a : r.b;
      y[i] = __fmul_rn(d, (float)(int)(signed char)((v >> (8 * (i & 3))) & 0xffu));
    }
  }
}

// v = p0 + p1 + p2 with bf16 parts (round-to-nearest-even converts; the remainders are exact in FP32)
template <int NP>
__device__ __forceinline__ void split8(const float* v, bf16x8* planes) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = v[i];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const __bf16 b = (__bf16)r;
      planes[p][i] = b;
      r -= (float)b;
    }
  }
}

// BM = weight rows per workgroup tile (64 or 128): 4 waves as 2 (rows) x 2 (tokens), a wave owns 64 tokens x BM/2 rows
// = 2 x BM/64 accumulators of 32x32.
//
// LDS image: rows of 32 bf16 (64 B, no padding), two buffers. The four 16-B chunks of a row are stored at chunk index
// c ^ ((row >> 2) & 3): a 16-lane group of a fragment read (16 consecutive rows, one chunk) and of a staging write
// (4 consecutive rows x 4 chunks) then both cover all 64 banks exactly once.
// Pipeline per 32-column step, ONE barrier: write step k+1 (dequantized from the registers fetched a step earlier) into
// the other buffer, fetch step k+2 into registers, run step k's MFMAs; the staging VALU work and the MFMAs of a wave are
// independent instruction streams that the scheduler interleaves.
constexpr int LROW = 32;
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * LROW + ((chunk ^ ((row >> 2) & 3)) << 3); }

template <int NP, int BM>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * NP * (BM + GB_N) * LROW * sizeof(__bf16); }

template <int TYPE, int NP, int NT, int BM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_q_kernel(QW w, const __bf16* __restrict__ xp, size_t plane_stride,
                                                              float* __restrict__ y, const float* __restrict__ resid,
                                                              int T, int mtiles, int ntiles) {
  constexpr int MI = BM / 64;  // 32-row weight fragments per wave, and staged (row, quarter) items per thread
  constexpr int WPL = BM * LROW, XPL = GB_N * LROW;      // elements per plane
  constexpr int BUF = NP * (WPL + XPL);                  // elements per buffer: W planes, then X planes
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int M = w.rows, K = w.cols;
  const int total = mtiles * ntiles;
  int tile = blockIdx.x;
  if ((total & 7) == 0) tile = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);  // contiguous range per XCD
  const int nt = tile / mtiles, mt = tile - nt * mtiles;
  const int m0 = mt * BM, n0 = nt * GB_N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // staging roles: weights (row, quarter of the chunk), MI rows per thread; activations: two 16-B pieces of the
  // 128 x 64 B tile per plane
  const int wrow = tid >> 2, wq = tid & 3;
  int grow[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) grow[j] = m0 + wrow + 64 * j < M ? m0 + wrow + 64 * j : M - 1;
  const int xr0 = tid >> 2, xr1 = xr0 + 64, xs = tid & 3;
  const __bf16* xg0 = xp + (size_t)(n0 + xr0) * K + xs * 8;
  const __bf16* xg1 = xp + (size_t)(n0 + xr1) * K + xs * 8;
  const int nk = K / GB_K;

  Raw8 rw[MI];
  bf16x8 rx[NP][2];
  auto fetch = [&](int kn) {  // global -> registers
#pragma unroll
    for (int j = 0; j < MI; ++j) rw[j] = load8<TYPE>(w, grow[j], kn, wq);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      rx[p][0] = *reinterpret_cast<const bf16x8*>(xg0 + p * plane_stride + (size_t)kn * GB_K);
      rx[p][1] = *reinterpret_cast<const bf16x8*>(xg1 + p * plane_stride + (size_t)kn * GB_K);
    }
  };
  auto stage = [&](int buf, int kn) {  // registers (holding step kn) -> LDS buffer; weights dequantized + split on the way
    __bf16* wb = lds + buf * BUF;
    __bf16* xb = wb + NP * WPL;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      float wv[8];
      deq8<TYPE>(rw[j], kn, wq, wv);
      bf16x8 wp[NP];
      split8<NP>(wv, wp);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x8*>(wb + p * WPL + lds_off(wrow + 64 * j, wq)) = wp[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<bf16x8*>(xb + p * XPL + lds_off(xr0, xs)) = rx[p][0];
      *reinterpret_cast<bf16x8*>(xb + p * XPL + lds_off(xr1, xs)) = rx[p][1];
    }
  };
  f32x16 acc[2][MI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  fetch(0);
  stage(0, 0);
  fetch(nk > 1 ? 1 : 0);
  __syncthreads();
  for (int kk = 0; kk < nk; ++kk) {
    const int cur = kk & 1;
    {  // step kk+1 into the other buffer, step kk+2 into registers (clamped at the end: unconditional, waits stay counted)
      const int k1 = kk + 1 < nk ? kk + 1 : nk - 1, k2 = kk + 2 < nk ? kk + 2 : nk - 1;
      stage(cur ^ 1, k1);
      fetch(k2);
    }
    const __bf16* wb = lds + cur * BUF;
    const __bf16* xb = wb + NP * WPL;
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      bf16x8 a[NP][2], b[NP][MI];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        a[p][0] = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(wn * 64 + fr, k16 * 2 + fh));
        a[p][1] = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(wn * 64 + 32 + fr, k16 * 2 + fh));
#pragma unroll
        for (int j = 0; j < MI; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(wb + p * WPL + lds_off(wm * (BM / 2) + 32 * j + fr, k16 * 2 + fh));
      }
      // (x plane, w plane) pairs, smallest contributions first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 6 - NT; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t] < NP ? PA[t] : 0][i], b[PB[t] < NP ? PB[t] : 0][j],
                                                                acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[token][row]: column (weight row) on the lane, token rows in the 16 registers
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * (BM / 2) + 32 * j + fr;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (t < T) {
          const size_t o = (size_t)t * M + m;
          y[o] = resid ? resid[o] + acc[i][j][e] : acc[i][j][e];
        }
      }
  }
}

// ---- the wave-specialised form: 8 waves, 4 producers + 4 consumers ---------------------------------------------------------
// In the kernel above a wave alternates between staging (global loads, dequantize + split on the VALU, LDS writes) and
// MFMAs, and the two only overlap across co-resident workgroups. Here the roles are split: waves 4-7 do nothing but
// staging, with loads two steps ahead in registers (they hold no accumulators), waves 0-3 nothing but fragment reads and
// MFMAs; every SIMD hosts one wave of each kind, so its VALU and its matrix pipe run side by side. Same tile (64 weight
// rows x 128 tokens x 32 columns per step), same two swizzled LDS buffers, one workgroup barrier per step.
constexpr int WS_THREADS = 512;

// BN: tokens per tile, 128 or 64. At most 64 tokens (a multi-sequence decode step, a short prompt) would leave the
// second token half of every 128-token tile empty: the 64-token form gives each consumer wave one 32-token fragment
// instead of two and halves the activation staging.
template <int TYPE, int NP, int NT, int BM, int BN>
__global__ __launch_bounds__(WS_THREADS) void gemm_q_ws_kernel(QW w, const __bf16* __restrict__ xp, size_t plane_stride,
                                                               float* __restrict__ y, const float* __restrict__ resid,
                                                               int T, int mtiles, int ntiles) {
  constexpr int MI = BM / 64;  // weight rows per producer thread; 32-row fragments per consumer wave
  constexpr int NI = BN / 64;  // 64-token groups staged per producer thread; 32-token fragments per consumer wave
  constexpr int WPL = BM * LROW, XPL = BN * LROW;
  constexpr int BUF = NP * (WPL + XPL);
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int M = w.rows, K = w.cols;
  const int total = mtiles * ntiles;
  int tile = blockIdx.x;
  if ((total & 7) == 0) tile = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int nt = tile / mtiles, mt = tile - nt * mtiles;
  const int m0 = mt * BM, n0 = nt * BN;
  // K split: slice blockIdx.y of gridDim.y covers steps [kbase, kbase + nk) and writes its partial tile to slice
  // blockIdx.y of y (then a [gridDim.y][T][M] workspace without residual; summed by ksplit_reduce_kernel)
  const int nk_all = K / GB_K, kper = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int kbase = (int)blockIdx.y * kper;
  const int nk = nk_all - kbase < kper ? nk_all - kbase : kper, last = kbase + nk - 1;
  const int nk3 = (nk + 2) / 3 * 3;  // both roles run a multiple of three steps (the tail steps only meet at the barrier)
  if (gridDim.y > 1) y += (size_t)blockIdx.y * (size_t)T * (size_t)M;
  if (threadIdx.x >= 256) {
    // ---------------- producers (waves 4-7: with round-robin wave placement one per SIMD, next to one consumer) ----------
    const int tid = threadIdx.x - 256;
    const int wrow = tid >> 2, wq = tid & 3;
    int grow[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) grow[j] = m0 + wrow + 64 * j < M ? m0 + wrow + 64 * j : M - 1;
    const int xr0 = tid >> 2, xs = tid & 3;  // activation rows xr0 (+ 64): token n0 + xr0 (+ 64)
    const __bf16* xg0 = xp + (size_t)(n0 + xr0) * K + xs * 8;
    struct Regs {
      Raw8 rw[MI];
      bf16x8 rx[NP][NI];
    };
    // three register sets in rotation, the loop unrolled by three so that the rotation is register renaming and not
    // copying (a copy of a set with loads in flight would wait for them): at step kk the set (kk+1)%3 holds step kk+1
    // (staged now), (kk+2)%3 holds step kk+2, and the set that held step kk is refilled with step kk+3.
    Regs s0, s1, s2;
    auto fetch = [&](Regs& g, int kn) {
      kn += kbase;
      kn = kn < last ? kn : last;  // clamped, unconditional: the waits stay counted
#pragma unroll
      for (int j = 0; j < MI; ++j) g.rw[j] = load8<TYPE>(w, grow[j], kn, wq);
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          g.rx[p][i] = *reinterpret_cast<const bf16x8*>(xg0 + (size_t)(64 * i) * K + p * plane_stride + (size_t)kn * GB_K);
    };
    auto stage = [&](const Regs& g, int buf, int kn) {
      kn += kbase;
      kn = kn < last ? kn : last;
      __bf16* wb = lds + buf * BUF;
      __bf16* xb = wb + NP * WPL;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        float wv[8];
        deq8<TYPE>(g.rw[j], kn, wq, wv);
        bf16x8 wp[NP];
        split8<NP>(wv, wp);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x8*>(wb + p * WPL + lds_off(wrow + 64 * j, wq)) = wp[p];
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<bf16x8*>(xb + p * XPL + lds_off(xr0 + 64 * i, xs)) = g.rx[p][i];
    };
    fetch(s0, 0);
    fetch(s1, 1);
    fetch(s2, 2);
    stage(s0, 0, 0);
    __syncthreads();  // buffer 0 holds step 0
    for (int kk = 0; kk < nk3; kk += 3) {
      fetch(s0, kk + 3);
      stage(s1, (kk + 1) & 1, kk + 1);
      __syncthreads();
      fetch(s1, kk + 4);
      stage(s2, (kk + 2) & 1, kk + 2);
      __syncthreads();
      fetch(s2, kk + 5);
      stage(s0, (kk + 3) & 1, kk + 3);
      __syncthreads();
    }
    return;
  }
  // ---------------- consumers: wave = BN/2 tokens x BM/2 rows ----------------
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  __syncthreads();
  for (int kk = 0; kk < nk3; ++kk) {
    if (kk >= nk) {  // uniform
      __syncthreads();
      continue;
    }
    const __bf16* wb = lds + (kk & 1) * BUF;
    const __bf16* xb = wb + NP * WPL;
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      bf16x8 a[NP][NI], b[NP][MI];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
          a[p][i] = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(wn * (32 * NI) + 32 * i + fr, k16 * 2 + fh));
#pragma unroll
        for (int j = 0; j < MI; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(wb + p * WPL + lds_off(wm * (BM / 2) + 32 * j + fr, k16 * 2 + fh));
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 6 - NT; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t] < NP ? PA[t] : 0][i], b[PB[t] < NP ? PB[t] : 0][j],
                                                                acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * (BM / 2) + 32 * j + fr;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = n0 + wn * (32 * NI) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (t < T) {
          const size_t o = (size_t)t * M + m;
          y[o] = resid ? resid[o] + acc[i][j][e] : acc[i][j][e];
        }
      }
  }
}


// ---- the integer-plane form (default) --------------------------------------------------------------------------------------
// A quantized weight is (scale of its sub-block) x (small integer) [- (min of its sub-block)], and the integers (Q4_K
// 0..15, Q6_K -32..31, Q8 -128..127) are EXACT in bf16. So the weights need no FP32 dequantization and no three-way split:
// the producers unpack the integers into ONE bf16 plane, the matrix cores form  S[row][tok] = sum_k q[row][k] x[tok][k]
// over the 16 columns of an MFMA (Q6_K's scale granularity; Q4_K / Q8 blocks are 32 = two of them) with the activation
// split into its 1..3 bf16 planes -- 3 MFMAs per FP32-grade product instead of 6 -- and the consumers fold the scales in
// on the VALU:  acc += (d*sc)[row] * S   and, for Q4_K,  acc -= (dmin*m)[row] * sum_k x[tok][k]  per 32-column step (the
// block sums of x come from split_kernel). This is the decode GEMV's factorisation (tl_gemv.hip Chunk::dot), so batched
// and token-by-token paths round alike; d*sc and dmin*m are the reference's separately rounded FP32 products.
// Producers shrink to a byte unpack (no exact-dequant arithmetic, one LDS plane instead of three).
// LDS per buffer: W [64][32] bf16 | X [NPX][BN][32] bf16 | sA [64][2] f32 (scale per row and 16-column half) |
// sB [64] f32 (Q4_K: dmin*m) | sx [BN] f32 (Q4_K: block sums of x).
template <int TYPE>
__device__ __forceinline__ bf16x8 unpack8(const Raw8& r, int q) {
  bf16x8 o;
  if (TYPE == TL_Q4_K) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned byte = (r.a >> (8 * t)) & 0xffu;
      o[t] = (__bf16)(float)(byte & 15u);
      o[4 + t] = (__bf16)(float)(byte >> 4);
    }
  } else if (TYPE == TL_Q6_K) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int part = i >> 2, t = i & 3, g = 2 * (q & 1) + part;
      const unsigned lo = (r.a >> (8 * t + 4 * part)) & 15u;
      const unsigned hi = (r.b >> (8 * t + 2 * g)) & 3u;
      o[i] = (__bf16)(float)((int)(lo | (hi << 4)) - 32);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned v = i < 4 ? r.a : r.b;
      o[i] = (__bf16)(float)(int)(signed char)((v >> (8 * (i & 3))) & 0xffu);
    }
  }
  return o;
}

template <int NPX, int BN, int BM>
constexpr size_t gemm_qi_lds_bytes() {
  return (size_t)2 * ((BM + NPX * BN) * LROW * sizeof(__bf16) + (BM * 3 + BN) * sizeof(float));
}

// BM = 128 (two 32-row fragments per consumer wave) halves the activation-fragment reads per MFMA: with half the MFMAs of
// the exact-dequantization kernel this form is bound by LDS reads at BM = 64; its LDS image is small enough for two
// workgroups per CU at BM = 128 (the three-weight-plane kernel's is not).
#ifndef TL_QI_WAVES
#define TL_QI_WAVES 2
#endif
#ifndef TL_QI_SETS
#define TL_QI_SETS 3
#endif
// NSEG = 3: up to three matrices that share the activation planes (q | k | v, gate | up: same K, same format) in ONE
// launch; the row tiles of the matrices follow each other in the tile index and every tile computes exactly what it would
// in a launch of its own: MSegs::mt_end = cumulative row-tile counts, S = K slices of each matrix (gridDim.y is the
// largest; workgroups beyond a matrix's own count leave at once). No residual in that form.
struct MSegs {
  QW w[3];
  float* y[3];
  int mt_end[3];
  int S[3];
};
template <int TYPE, int NPX, int BN, int BM, int NSEG>
__global__ __launch_bounds__(WS_THREADS, TL_QI_WAVES) void gemm_qi_ws_kernel(MSegs segs, const __bf16* __restrict__ xp, size_t plane_stride,
                                                                const float* __restrict__ xsum, int tpad,
                                                                const float* __restrict__ resid,
                                                                int T, int mtiles, int ntiles) {
  constexpr int NI = BN / 64, MI = BM / 64;
  constexpr bool PER16 = TYPE == TL_Q6_K;  // scale granularity: 16 columns (one MFMA) or the whole 32-column step
  constexpr int WPL = BM * LROW, XPL = BN * LROW;
  constexpr size_t BUF_BYTES = (WPL + NPX * XPL) * sizeof(__bf16) + (BM * 3 + BN) * sizeof(float);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int total = mtiles * ntiles;
  int tile = blockIdx.x;
  if ((total & 7) == 0) tile = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int nt = tile / mtiles;
  int mt = tile - nt * mtiles, seg = 0;
  if (NSEG > 1) {
    seg = (mt >= segs.mt_end[0]) + (mt >= segs.mt_end[1]);
    if (seg) mt -= segs.mt_end[seg - 1];
  }
  const int nslices = NSEG > 1 ? segs.S[seg] : (int)gridDim.y;
  if ((int)blockIdx.y >= nslices) return;  // (uniform over the workgroup, before any barrier)
  const QW w = segs.w[seg];
  float* __restrict__ y = segs.y[seg];
  const int M = w.rows, K = w.cols;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk_all = K / GB_K, kper = (nk_all + nslices - 1) / nslices;
  const int kbase = (int)blockIdx.y * kper;
  const int nk = nk_all - kbase < kper ? nk_all - kbase : kper, last = kbase + nk - 1;
  const int nk3 = (nk + TL_QI_SETS - 1) / TL_QI_SETS * TL_QI_SETS;  // both roles run a multiple of the rotation length
  if (nslices > 1) y += (size_t)blockIdx.y * (size_t)T * (size_t)M;
  auto wbuf = [&](int b) { return reinterpret_cast<__bf16*>(lds_raw + (size_t)b * BUF_BYTES); };
  auto fbuf = [&](int b) {
    return reinterpret_cast<float*>(lds_raw + (size_t)b * BUF_BYTES + (WPL + NPX * XPL) * sizeof(__bf16));
  };
  if (threadIdx.x >= 256) {
    // ---------------- producers ----------------
    const int tid = threadIdx.x - 256;
    const int wrow = tid >> 2, wq = tid & 3;
    int grow[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) grow[j] = m0 + wrow + 64 * j < M ? m0 + wrow + 64 * j : M - 1;
    const int xr0 = tid >> 2, xs = tid & 3;
    const __bf16* xg0 = xp + (size_t)(n0 + xr0) * K + xs * 8;
    const float* sg = xsum + n0 + (tid < BN ? tid : 0);  // block sums of x: [K / 32][tpad]
    struct Regs {
      Raw8 rw[MI];
      bf16x8 rx[NPX][NI];
      float sx;
    };
#if TL_QI_SETS == 3
    Regs s0, s1, s2;
#else
    Regs s0, s1;
#endif
    auto fetch = [&](Regs& g, int kn) {
      kn += kbase;
      kn = kn < last ? kn : last;
#pragma unroll
      for (int j = 0; j < MI; ++j) g.rw[j] = load8<TYPE>(w, grow[j], kn, wq);
#pragma unroll
      for (int p = 0; p < NPX; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          g.rx[p][i] = *reinterpret_cast<const bf16x8*>(xg0 + (size_t)(64 * i) * K + p * plane_stride + (size_t)kn * GB_K);
      if (TYPE == TL_Q4_K) g.sx = sg[(size_t)kn * tpad];
    };
    auto stage = [&](const Regs& g, int buf, int kn) {
      kn += kbase;
      kn = kn < last ? kn : last;
      __bf16* wb = wbuf(buf);
      __bf16* xb = wb + WPL;
      float* fb = fbuf(buf);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
#pragma clang fp contract(off)
        const int row = wrow + 64 * j;
        const Raw8& r = g.rw[j];
        *reinterpret_cast<bf16x8*>(wb + lds_off(row, wq)) = unpack8<TYPE>(r, wq);
        if (TYPE == TL_Q4_K) {
          if (wq == 0) {
            const int c = kn & 7;
            const float d = h2f(r.h[0] & 0xffffu), dmin = h2f(r.h[0] >> 16);
            const unsigned sc = ((r.h[1] >> (4 * c)) & 15u) | (((r.h[3] >> (2 * c)) & 3u) << 4);
            const unsigned mn = ((r.h[2] >> (4 * c)) & 15u) | (((r.h[3] >> (16 + 2 * c)) & 3u) << 4);
            const float d1 = __fmul_rn(d, (float)sc);
            fb[2 * row] = d1;
            fb[2 * row + 1] = d1;
            fb[2 * BM + row] = __fmul_rn(dmin, (float)mn);
          }
        } else if (TYPE == TL_Q6_K) {
          if ((wq & 1) == 0)  // quarters 0 and 2 carry the scale bytes of the two 16-column halves
            fb[2 * row + (wq >> 1)] = __fmul_rn(h2f(r.h[1]), (float)(int)(signed char)(r.h[0] & 0xffu));
        } else {
          if (wq == 0) {
            const float d = h2f_scale(r.h[0]);
            fb[2 * row] = d;
            fb[2 * row + 1] = d;
          }
        }
      }
      if (TYPE == TL_Q4_K && tid < BN) fb[3 * BM + tid] = g.sx;
#pragma unroll
      for (int p = 0; p < NPX; ++p)
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<bf16x8*>(xb + p * XPL + lds_off(xr0 + 64 * i, xs)) = g.rx[p][i];
    };
#if TL_QI_SETS == 3
    fetch(s0, 0);
    fetch(s1, 1);
    fetch(s2, 2);
    stage(s0, 0, 0);
    __syncthreads();
    for (int kk = 0; kk < nk3; kk += 3) {
      fetch(s0, kk + 3);
      stage(s1, (kk + 1) & 1, kk + 1);
      __syncthreads();
      fetch(s1, kk + 4);
      stage(s2, (kk + 2) & 1, kk + 2);
      __syncthreads();
      fetch(s2, kk + 5);
      stage(s0, (kk + 3) & 1, kk + 3);
      __syncthreads();
    }
#else  // two register sets: loads two steps ahead (fewer registers: two workgroups per CU)
    fetch(s0, 0);
    fetch(s1, 1);
    stage(s0, 0, 0);
    __syncthreads();
    for (int kk = 0; kk < nk3; kk += 2) {
      fetch(s0, kk + 2);
      stage(s1, (kk + 1) & 1, kk + 1);
      __syncthreads();
      fetch(s1, kk + 3);
      stage(s0, kk & 1, kk + 2);
      __syncthreads();
    }
#endif
    return;
  }
  // ---------------- consumers: wave = BN/2 tokens x BM/2 rows ----------------
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int lrow0 = wm * (BM / 2) + fr;  // this lane's weight row of fragment j: lrow0 + 32 * j
  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int kk = 0; kk < nk3; ++kk) {
    if (kk >= nk) {  // uniform
      __syncthreads();
      continue;
    }
    const __bf16* wb = wbuf(kk & 1);
    const __bf16* xb = wb + WPL;
    const float* fb = fbuf(kk & 1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x16 t[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) t[j] = zero;
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        bf16x8 a[NPX];
#pragma unroll
        for (int p = 0; p < NPX; ++p)
          a[p] = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(wn * (32 * NI) + 32 * i + fr, k16 * 2 + fh));
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(wb + lds_off(lrow0 + 32 * j, k16 * 2 + fh));
#pragma unroll
          for (int p = NPX - 1; p >= 0; --p) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], b, t[j], 0, 0, 0);  // smallest plane first
          if (PER16) {
            const float sa = fb[2 * (lrow0 + 32 * j) + k16];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(sa, t[j][e], acc[i][j][e]);
            t[j] = zero;
          }
        }
      }
      if (!PER16) {
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const float sa = fb[2 * (lrow0 + 32 * j)];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(sa, t[j][e], acc[i][j][e]);
        }
      }
      if (TYPE == TL_Q4_K) {
        const float* sx = fb + 3 * BM;
        f32x4 v[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) v[g4] = *reinterpret_cast<const f32x4*>(sx + wn * (32 * NI) + 32 * i + 8 * g4 + 4 * fh);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const float nsb = -fb[2 * BM + lrow0 + 32 * j];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(nsb, v[e >> 2][e & 3], acc[i][j][e]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + lrow0 + 32 * j;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = n0 + wn * (32 * NI) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (t < T) {
          const size_t o = (size_t)t * M + m;
          y[o] = resid ? resid[o] + acc[i][j][e] : acc[i][j][e];
        }
      }
  }
}

// ---- the expanded-plane form: prefill planes resident in HBM, the matrix cores fed by LDS-DMA ------------------------------
// Measured (profiles/r03_d): in gemm_qi_ws_kernel the producers' unpack and the consumers' MFMA + scale FMAs share each
// SIMD's issue slots; either role alone needs ~75 % of the step. The reference keeps an FP32 expansion of every matrix on the
// device for its batched path (ensure_f32_concatenated_weights_loaded, weight_management.cpp; cuBLAS reads 4 bytes per
// weight); with 288 GB of HBM this backend can afford the same trade at half the bytes: tl_gemm_expand() writes, once per
// matrix, exactly what the producers of gemm_qi_ws_kernel would write into LDS for every (256-row tile, 32-column step) --
// the integers as ONE bf16 plane, chunk swizzle applied, and the FP32 scale words -- as contiguous 16 KB + 3 KB records.
// The kernel's producer waves then issue nothing but global_load_lds (LDS-DMA: no VGPRs, no VALU, no LDS write
// instructions), three steps deep in a ring of LDS buffers, and the eight consumer waves run the same MFMA / scale
// sequence as gemm_qi_ws_kernel on the same operands: the results are bit-identical to that kernel's.
// Measured and not kept (profiles/r03_g_prefill_planes.txt): an eight-wave workgroup in which every wave is a consumer (256
// VGPRs; the DMAs issued through inline assembly, fragment reads a quarter step ahead of their MFMAs across the step
// barrier): -4 %; the same with two sets of step sums and every MFMA carrying its share of the scale FMAs (packed) in its
// shadow (sched_group_barrier): -13 %. The matrix cores are busy 0.42 of the kernel's cycles here (0.31 in
// gemm_qi_ws_kernel); what a step still waits for is not LDS latency inside a wave.
// Producers and consumers stay separate waves for a second reason: hipcc makes every LDS read that follows an LDS-DMA
// issue in the same wave wait for that DMA (it cannot tell the buffers apart), which would serialise a wave that did both.
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;
constexpr int PX_BM = 256, PX_BN = 128, PX_NB = 3;
constexpr int PX_PW = 4, PX_CW = 8;                  // producer / consumer waves
constexpr int PX_THREADS = 64 * (PX_PW + PX_CW);
constexpr int PX_SC = 3 * PX_BM;                     // scale words per (row tile, step): sA [256][2] | sB [256]
constexpr size_t PX_WBYTES = (size_t)PX_BM * LROW * sizeof(__bf16);   // 16 KB
constexpr size_t PX_XBYTES = (size_t)PX_BN * LROW * sizeof(__bf16);   // 8 KB per activation plane
template <int NPX>
constexpr size_t px_buf_bytes() { return PX_WBYTES + NPX * PX_XBYTES + PX_SC * sizeof(float) + 1024; }  // (+ sx: 512 B used)

// one thread per row of a 256-row tile; grid (steps, row tiles)
template <int TYPE>
__global__ __launch_bounds__(PX_BM) void px_expand_kernel(QW w, __bf16* __restrict__ q, float* __restrict__ sc, int nk) {
#pragma clang fp contract(off)
  const int k = blockIdx.x, mt = blockIdx.y, r = threadIdx.x;
  const int grow = mt * PX_BM + r;
  __bf16* qb = q + ((size_t)mt * nk + k) * (PX_BM * LROW);
  float* fb = sc + ((size_t)mt * nk + k) * PX_SC;
  if (grow >= w.rows) {  // padding rows: zeros
    const bf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
    for (int wq = 0; wq < 4; ++wq) *reinterpret_cast<bf16x8*>(qb + lds_off(r, wq)) = z;
    fb[2 * r] = 0.f;
    fb[2 * r + 1] = 0.f;
    fb[2 * PX_BM + r] = 0.f;
    return;
  }
  fb[2 * PX_BM + r] = 0.f;
#pragma unroll
  for (int wq = 0; wq < 4; ++wq) {
    const Raw8 raw = load8<TYPE>(w, grow, k, wq);
    *reinterpret_cast<bf16x8*>(qb + lds_off(r, wq)) = unpack8<TYPE>(raw, wq);
    if (TYPE == TL_Q4_K) {  // (the same separately rounded products as gemm_qi_ws_kernel's producers)
      if (wq == 0) {
        const int c = k & 7;
        const float d = h2f(raw.h[0] & 0xffffu), dmin = h2f(raw.h[0] >> 16);
        const unsigned scv = ((raw.h[1] >> (4 * c)) & 15u) | (((raw.h[3] >> (2 * c)) & 3u) << 4);
        const unsigned mn = ((raw.h[2] >> (4 * c)) & 15u) | (((raw.h[3] >> (16 + 2 * c)) & 3u) << 4);
        const float d1 = __fmul_rn(d, (float)scv);
        fb[2 * r] = d1;
        fb[2 * r + 1] = d1;
        fb[2 * PX_BM + r] = __fmul_rn(dmin, (float)mn);
      }
    } else if (TYPE == TL_Q6_K) {
      if ((wq & 1) == 0) fb[2 * r + (wq >> 1)] = __fmul_rn(h2f(raw.h[1]), (float)(int)(signed char)(raw.h[0] & 0xffu));
    } else {
      if (wq == 0) {
        const float d = h2f_scale(raw.h[0]);
        fb[2 * r] = d;
        fb[2 * r + 1] = d;
      }
    }
  }
}

struct PSegs {
  const __bf16* q[3];
  const float* sc[3];
  float* y[3];
  int M[3];
  int mt_end[3];  // cumulative 256-row tile counts
  int S[3];       // K slices of each matrix
};
// BM = 256: a workgroup takes a whole record (consumer waves 4 (rows) x 2 (tokens), 64 x 64 each); BM = 128: half a record
// (2 x 4 waves of 64 rows x 32 tokens) -- twice the workgroups for the matrices with few row tiles (px_plan).
#ifdef TL_PX_STAMP
// diagnostic build only (make EXTRA=-DTL_PX_STAMP; tools/px_stamps.py): shader-clock stamps of workgroup (0, 0), lane 0 of
// every wave, for the first PX_STAMP_STEPS steps: consumers [0] past the step barrier, [1] / [3] the first / second token
// half's MFMAs issued, [2] / [4] its sums folded, [7] about to meet; producers [0] past the barrier, [1] DMAs of step kk + 2
// issued, [2] step kk + 1 landed (about to meet). The stamps cost time themselves (a stamped step is ~1.5x a plain one).
constexpr int PX_STAMP_STEPS = 48;
__device__ unsigned long long px_stamps[PX_PW + PX_CW][PX_STAMP_STEPS][8];
#define PX_STAMP(k, slot)                                                                                       \
  do {                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (k) < PX_STAMP_STEPS) px_stamps[wave][(k)][(slot)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  } while (0)
#else
#define PX_STAMP(k, slot) do { } while (0)
#endif
// BN = 64 (with BM = 256 only): 64-token tiles for launches of 33..64 tokens (4 x 2 waves of 64 rows x 32 tokens).
template <int TYPE, int NPX, int BM, int BN = PX_BN>
__global__ __launch_bounds__(PX_THREADS) void gemm_px_kernel(PSegs segs, const __bf16* __restrict__ xp, size_t plane_stride,
                                                             const float* __restrict__ xsum, int tpad,
                                                             const float* __restrict__ resid, int T, int K, int mtiles,
                                                             int ntiles) {
  static_assert(BM == 256 || BM == 128, "a record or half a record");
  static_assert(BN == 128 || (BN == 64 && BM == 256), "64-token tiles with whole records only");
  constexpr int NI = BM == 256 ? BN / 64 : 1, MI = 2, HALVES = PX_BM / BM;
  constexpr bool PER16 = TYPE == TL_Q6_K;
  constexpr size_t BUF = px_buf_bytes<NPX>();
  constexpr int XPL = PX_BN * LROW;  // (the LDS image keeps room for 128 tokens per plane)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int total = mtiles * ntiles;
  int tile = blockIdx.x;
  if ((total & 7) == 0) tile = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int nt = tile / mtiles;
  int mt = tile - nt * mtiles;
  const int seg = (mt >= segs.mt_end[0]) + (mt >= segs.mt_end[1]);
  if (seg) mt -= segs.mt_end[seg - 1];
  const int nslices = segs.S[seg];
  if ((int)blockIdx.y >= nslices) return;  // (uniform over the workgroup, before any barrier)
  const int M = segs.M[seg];
  float* __restrict__ y = segs.y[seg];
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk_all = K / GB_K, kper = (nk_all + nslices - 1) / nslices;
  const int kbase = (int)blockIdx.y * kper;
  const int nk = nk_all - kbase < kper ? nk_all - kbase : kper;
  if (nslices > 1) y += (size_t)blockIdx.y * (size_t)T * (size_t)M;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= PX_CW) {
    // ---------------- producers: LDS-DMA only ----------------
    const int pw = wave - PX_CW;
    constexpr int XG = BN / 16;       // sixteen-token groups of a plane
    constexpr int XI = NPX * XG / 4;  // activation DMAs per wave and step (NPX planes x XG groups over 4 waves)
    constexpr int WI = BM / 64;       // weight DMAs per wave and step (1 KB = 16 rows each)
    constexpr int PER = WI + XI + 1;  // DMAs per wave and step
    const int rec = mt / HALVES, half = mt - rec * HALVES;  // the 256-row record of this tile, and which half of it
    const __bf16* wsrc = segs.q[seg] + ((size_t)rec * nk_all + kbase) * (PX_BM * LROW) + (size_t)half * (BM * LROW) +
                         (size_t)(pw * WI) * 512 + lane * 8;
    // the scale words: BM = 256: waves 0..2 a KB each of sA | sB; BM = 128: wave 0 (and 2, again: every wave issues the same
    // number of DMAs) this half's sA, wave 1 this half's sB (512 B; the upper lanes repeat it into the slack behind);
    // wave 3: the block sums of x
    const float* scb = segs.sc[seg] + ((size_t)rec * nk_all + kbase) * PX_SC;
    const float* msrc = pw == 3 ? xsum + (size_t)kbase * tpad + n0 + (lane & (BN / 4 - 1)) * 4  // (BN block sums; the other lanes repeat them)
                        : BM == 256 ? scb + pw * 256 + lane * 4
                        : pw == 1   ? scb + 2 * PX_BM + half * BM + (lane & 31) * 4
                                    : scb + half * (2 * BM) + lane * 4;
    const size_t mstep = pw < 3 ? (size_t)PX_SC : (size_t)tpad;
    const __bf16* xsrc[XI];
    int xdst[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int idx = pw * XI + j, p = idx / XG, i = idx % XG;
      const int row = i * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
      xsrc[j] = xp + (size_t)p * plane_stride + (size_t)(n0 + row) * K + (size_t)kbase * GB_K + chunk * 8;
      xdst[j] = (int)(PX_WBYTES + p * PX_XBYTES + i * 1024);
    }
    const int mdst = (int)(PX_WBYTES + NPX * PX_XBYTES) +
                     (pw == 3 ? PX_SC * 4 : BM == 256 ? pw * 1024 : pw == 1 ? 2 * BM * 4 : 0);
    auto issue = [&](int kk) {
      unsigned char* b = lds_raw + (size_t)(kk % PX_NB) * BUF;
#pragma unroll
      for (int i = 0; i < WI; ++i)
        __builtin_amdgcn_global_load_lds((glb_vptr)(wsrc + (size_t)kk * (PX_BM * LROW) + i * 512), (lds_vptr)(b + (pw * WI + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < XI; ++j)
        __builtin_amdgcn_global_load_lds((glb_vptr)(xsrc[j] + (size_t)kk * GB_K), (lds_vptr)(b + xdst[j]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_vptr)(msrc + (size_t)kk * mstep), (lds_vptr)(b + mdst), 16, 0, 0);
    };
    issue(0);
    if (nk > 1) issue(1);
    for (int kk = 0; kk < nk; ++kk) {
      // step kk has landed (the DMAs of step kk + 1 may still be in flight); meet the consumers, who are through with step
      // kk - 1: its buffer takes step kk + 2
      if (kk + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PX_STAMP(kk, 2);
      asm volatile("s_barrier" ::: "memory");
      PX_STAMP(kk, 0);
      if (kk + 2 < nk) issue(kk + 2);
      PX_STAMP(kk, 1);
    }
    return;
  }
  // ---------------- consumers: wave = 64 rows x 32 NI tokens ----------------
  const int wm = BM == 256 ? wave & 3 : wave & 1, wn = BM == 256 ? wave >> 2 : wave >> 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int lrow0 = wm * 64 + fr;
  const int tb = wn * (32 * NI);  // the wave's first token of the tile
  f32x16 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < nk; ++kk) {
    PX_STAMP(kk, 7);
    __syncthreads();
    PX_STAMP(kk, 0);
    const unsigned char* b = lds_raw + (size_t)(kk % PX_NB) * BUF;
    const __bf16* wb = reinterpret_cast<const __bf16*>(b);
    const __bf16* xb = reinterpret_cast<const __bf16*>(b + PX_WBYTES);
    const float* fb = reinterpret_cast<const float*>(b + PX_WBYTES + NPX * PX_XBYTES);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x16 t[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) t[j] = zero;
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        bf16x8 a[NPX];
#pragma unroll
        for (int p = 0; p < NPX; ++p)
          a[p] = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(tb + 32 * i + fr, k16 * 2 + fh));
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const bf16x8 bq = *reinterpret_cast<const bf16x8*>(wb + lds_off(lrow0 + 32 * j, k16 * 2 + fh));
#pragma unroll
          for (int p = NPX - 1; p >= 0; --p) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], bq, t[j], 0, 0, 0);  // smallest plane first
          if (PER16) {
            const float sa = fb[2 * (lrow0 + 32 * j) + k16];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(sa, t[j][e], acc[i][j][e]);
            t[j] = zero;
          }
        }
      }
      PX_STAMP(kk, 1 + 2 * i);
      if (!PER16) {
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const float sa = fb[2 * (lrow0 + 32 * j)];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(sa, t[j][e], acc[i][j][e]);
        }
      }
      if (TYPE == TL_Q4_K) {
        const float* sx = fb + PX_SC;
        f32x4 v[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) v[g4] = *reinterpret_cast<const f32x4*>(sx + tb + 32 * i + 8 * g4 + 4 * fh);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const float nsb = -fb[2 * BM + lrow0 + 32 * j];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(nsb, v[e >> 2][e & 3], acc[i][j][e]);
        }
      }
      PX_STAMP(kk, 2 + 2 * i);
    }
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + lrow0 + 32 * j;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = n0 + tb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (t < T) {
          const size_t o = (size_t)t * M + m;
          y[o] = resid ? resid[o] + acc[i][j][e] : acc[i][j][e];
        }
      }
  }
}

// ---- the direct form for few tokens (multi-sequence decode steps, short prompts) ----------------------------------------
// With <= 64 tokens a weight byte meets so few MFMAs that the kernel above is bound by its four producer waves (byte
// unpack + LDS write of every weight: the whole weight set passes in ~8 ms whatever the token count). Here the weights
// never touch LDS: EVERY wave loads the raw bytes of its own 32 rows straight into registers (QD_SETS steps ahead),
// unpacks them into the MFMA operand it feeds itself, and applies the scales of its own rows from the same registers;
// only the activation tile (NPX planes x 32*NI tokens, + the block sums for Q4_K) is staged through LDS, cooperatively,
// one step ahead, one barrier per step. A step is 64 columns = two 32-weight chunks: lane (row fr, half fh) loads the
// WHOLE chunk 2s + fh of its row with one 16-byte load (two for Q8), and v_permlane32_swap hands the quarters round so
// that both halves of the wave hold the operand of one chunk (an MFMA must not mix chunks: their scales differ).
// Same arithmetic per (row, token) as gemm_qi_ws_kernel: S per 16 columns on the matrix cores, acc += (d*sc) * S,
// acc -= (dmin*m) * sum(x) on the VALU. Workgroup = NW waves x 32 rows; K split over gridDim.y slices as above.
// K must be a multiple of 64.
#ifndef TL_QD_SETS
#define TL_QD_SETS 3
#endif
constexpr int QD_SETS = TL_QD_SETS;  // weight register sets = 64-column steps a wave's loads run ahead
__device__ __forceinline__ u32x2 half_swap(unsigned a, unsigned b) {  // -> {[a.lo | b.lo], [a.hi | b.hi]} (32-lane halves)
  return __builtin_amdgcn_permlane32_swap(a, b, false, false);
}
// NSEG = 3: several matrices in one launch as in gemm_qi_ws_kernel (MSegs; here all with gridDim.y slices)
template <int TYPE, int NPX, int NI, int NW, int NSEG>
__global__ __launch_bounds__(64 * NW) void gemm_qd_kernel(MSegs segs, const __bf16* __restrict__ xp, size_t plane_stride,
                                                          const float* __restrict__ xsum, int tpad,
                                                          const float* __restrict__ resid, int T, int mtiles, int ntiles) {
  constexpr int BN = 32 * NI, NT = 64 * NW, BMD = 32 * NW;
  constexpr bool PER16 = TYPE == TL_Q6_K;
  constexpr int XPL = BN * LROW;                   // one plane of one chunk
  constexpr int XPIECES = 2 * NPX * BN * 4;        // 16-byte pieces of one step's activation tile
  constexpr int XR = (XPIECES + NT - 1) / NT;      // pieces per thread
  constexpr size_t BUF_BYTES = (size_t)2 * NPX * XPL * sizeof(__bf16) + 2 * BN * sizeof(float);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tile = blockIdx.x;
  const int nt = tile / mtiles;
  int mt = tile - nt * mtiles, seg = 0;
  if (NSEG > 1) {
    seg = (mt >= segs.mt_end[0]) + (mt >= segs.mt_end[1]);
    if (seg) mt -= segs.mt_end[seg - 1];
  }
  const QW w = segs.w[seg];
  float* __restrict__ y = segs.y[seg];
  const int M = w.rows, K = w.cols;
  const int m0 = mt * BMD, n0 = nt * BN;
  const int ns_all = K / 64, sper = (ns_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int sbase = (int)blockIdx.y * sper;
  const int ns = ns_all - sbase < sper ? ns_all - sbase : sper;
  const int nsr = (ns + QD_SETS - 1) / QD_SETS * QD_SETS;  // a multiple of the register-set rotation
  auto phys = [&](int ss) { return sbase + (ss < ns ? ss : ns - 1); };  // (steps past the end reload the last one)
  if (gridDim.y > 1) y += (size_t)blockIdx.y * (size_t)T * (size_t)M;
  auto xbuf = [&](int b) { return reinterpret_cast<__bf16*>(lds_raw + (size_t)b * BUF_BYTES); };
  auto sbuf = [&](int b) { return reinterpret_cast<float*>(lds_raw + (size_t)b * BUF_BYTES + (size_t)2 * NPX * XPL * sizeof(__bf16)); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int row = m0 + 32 * wave + fr;
  const size_t R = (size_t)(row < M ? row : M - 1);

  // this thread's pieces of the activation tile: piece = (chunk, plane, token row, quarter). No lane is masked off (a
  // load in a divergent branch makes hipcc drain the whole prefetch queue at the join): where the pieces do not divide
  // evenly, the spare threads copy a piece a second time (same bytes to the same LDS address).
  const __bf16* xg[XR];
  int xl[XR];
#pragma unroll
  for (int r = 0; r < XR; ++r) {
    const int id = (tid + r * NT) % XPIECES;
    const int cp = id / (BN * 4), rem = id - cp * (BN * 4), tr = rem >> 2, q = rem & 3;
    const int c = cp / NPX, p = cp - c * NPX;
    xg[r] = xp + (size_t)p * plane_stride + (size_t)(n0 + tr) * K + c * 32 + q * 8;
    xl[r] = cp * XPL + lds_off(tr, q);
  }
  const int sxi = tid % (2 * BN);  // block sums of x: [chunk][token]
  const float* sg = xsum + (size_t)(sxi / BN) * tpad + n0 + sxi % BN;
  struct XRegs {
    bf16x8 v[XR];
    float sx;
  };
  auto fetch_x = [&](XRegs& g, int ss) {
    const int sn = phys(ss);
#pragma unroll
    for (int r = 0; r < XR; ++r) g.v[r] = *reinterpret_cast<const bf16x8*>(xg[r] + (size_t)sn * 64);
    if (TYPE == TL_Q4_K) g.sx = sg[(size_t)(2 * sn) * tpad];
  };
  auto write_x = [&](const XRegs& g, int buf) {
    __bf16* xb = xbuf(buf);
#pragma unroll
    for (int r = 0; r < XR; ++r) *reinterpret_cast<bf16x8*>(xb + xl[r]) = g.v[r];
    if (TYPE == TL_Q4_K) sbuf(buf)[sxi] = g.sx;
  };
  struct WRegs {
    u32x4 q0, q1;  // the payload of chunk 2s + fh (q1: second half of a Q8 chunk)
    u32x2 hi;      // Q6_K: its high bits
    u32x4 hdr;     // Q4_K: super-block header; Q6_K: [0] = the four scale bytes of both chunks
    unsigned d;    // Q6_K / Q8_K: d of the super-block; Q8_0: d of both chunks
  };
  auto fetch_w = [&](WRegs& g, int ss) {
    const int sn = phys(ss);
    const size_t c0 = (size_t)2 * sn, cm = c0 + fh, cols = (size_t)K;
    if (TYPE == TL_Q4_K) {
      g.q0 = *reinterpret_cast<const u32x4*>(w.p0 + R * (cols >> 1) + cm * 16);
      g.hdr = *reinterpret_cast<const u32x4*>(w.p1 + R * (cols >> 4) + (c0 >> 3) * 16);
    } else if (TYPE == TL_Q6_K) {
      g.q0 = *reinterpret_cast<const u32x4*>(w.p0 + R * (cols >> 1) + cm * 16);
      g.hi = *reinterpret_cast<const u32x2*>(w.p1 + R * (cols >> 2) + cm * 8);
      g.hdr[0] = *reinterpret_cast<const unsigned*>(w.p2 + R * (cols >> 4) + c0 * 2);
      g.d = *reinterpret_cast<const uint16_t*>(w.p3 + (R * (cols >> 8) + (c0 >> 3)) * 2);
    } else {
      g.q0 = *reinterpret_cast<const u32x4*>(w.p0 + R * cols + cm * 32);
      g.q1 = *reinterpret_cast<const u32x4*>(w.p0 + R * cols + cm * 32 + 16);
      if (TYPE == TL_Q8_0) g.d = *reinterpret_cast<const unsigned*>(w.p1 + (R * (cols >> 5) + c0) * 2);
      else g.d = *reinterpret_cast<const uint16_t*>(w.p1 + (R * (cols >> 8) + (c0 >> 3)) * 2);
    }
  };

  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const WRegs& g, int buf, int ss) {
#pragma clang fp contract(off)
    const int c0 = 2 * phys(ss);
    // operands [chunk][16-column half]: quarter 2 * k16 + fh of chunk c0 + chunk in every lane
    Raw8 op[2][2] = {};
    if (TYPE == TL_Q4_K || TYPE == TL_Q6_K) {
      const u32x2 s01 = half_swap(g.q0[0], g.q0[1]), s23 = half_swap(g.q0[2], g.q0[3]);
      op[0][0].a = s01[0]; op[1][0].a = s01[1];
      op[0][1].a = s23[0]; op[1][1].a = s23[1];
      if (TYPE == TL_Q6_K) {
        const u32x2 h0 = half_swap(g.hi[0], g.hi[0]), h1 = half_swap(g.hi[1], g.hi[1]);
        op[0][0].b = h0[0]; op[1][0].b = h0[1];
        op[0][1].b = h1[0]; op[1][1].b = h1[1];
      }
    } else {
      const u32x2 s02 = half_swap(g.q0[0], g.q0[2]), s13 = half_swap(g.q0[1], g.q0[3]);
      const u32x2 s46 = half_swap(g.q1[0], g.q1[2]), s57 = half_swap(g.q1[1], g.q1[3]);
      op[0][0].a = s02[0]; op[0][0].b = s13[0]; op[1][0].a = s02[1]; op[1][0].b = s13[1];
      op[0][1].a = s46[0]; op[0][1].b = s57[0]; op[1][1].a = s46[1]; op[1][1].b = s57[1];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const __bf16* xb = xbuf(buf) + c * NPX * XPL;
      float s16[2] = {0.f, 0.f}, m1 = 0.f;
      if (TYPE == TL_Q4_K) {
        const int j = (c0 + c) & 7;
        const float d = h2f(g.hdr[0] & 0xffffu), dmin = h2f(g.hdr[0] >> 16);
        const unsigned sc = ((g.hdr[1] >> (4 * j)) & 15u) | (((g.hdr[3] >> (2 * j)) & 3u) << 4);
        const unsigned mn = ((g.hdr[2] >> (4 * j)) & 15u) | (((g.hdr[3] >> (16 + 2 * j)) & 3u) << 4);
        s16[0] = __fmul_rn(d, (float)sc);
        m1 = __fmul_rn(dmin, (float)mn);
      } else if (TYPE == TL_Q6_K) {
        const float d = h2f(g.d);
        s16[0] = __fmul_rn(d, (float)(int)(signed char)((g.hdr[0] >> (16 * c)) & 0xffu));
        s16[1] = __fmul_rn(d, (float)(int)(signed char)((g.hdr[0] >> (16 * c + 8)) & 0xffu));
      } else if (TYPE == TL_Q8_0) {
        s16[0] = h2f_scale((g.d >> (16 * c)) & 0xffffu);
      } else {
        s16[0] = h2f_scale(g.d);
      }
      f32x16 t[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) t[i] = zero;
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        const bf16x8 b = unpack8<TYPE>(op[c][k16], 2 * k16 + fh);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
          for (int p = NPX - 1; p >= 0; --p) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(xb + p * XPL + lds_off(32 * i + fr, k16 * 2 + fh));
            t[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, t[i], 0, 0, 0);
          }
          if (PER16) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = fmaf(s16[k16], t[i][e], acc[i][e]);
            t[i] = zero;
          }
        }
      }
      if (!PER16) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][e] = fmaf(s16[0], t[i][e], acc[i][e]);
      }
      if (TYPE == TL_Q4_K) {
        const float* sx = sbuf(buf) + c * BN;
        const float nsb = -m1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          f32x4 v[4];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) v[g4] = *reinterpret_cast<const f32x4*>(sx + 32 * i + 8 * g4 + 4 * fh);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][e] = fmaf(nsb, v[e >> 2][e & 3], acc[i][e]);
        }
      }
    }
  };

  // Both prefetch queues live in registers, indexed by unrolled constants only. The activation pieces run three steps
  // ahead too (an L2 round trip per step would otherwise bound the step time); the LDS image stays double-buffered.
  static_assert(QD_SETS % 3 == 0, "the unrolled rotation must cover the three activation register sets");
  WRegs sets[QD_SETS];
  XRegs xs[3];
#pragma unroll
  for (int u = 0; u < QD_SETS; ++u) fetch_w(sets[u], u);
  fetch_x(xs[0], 0);
  fetch_x(xs[1], 1);
  fetch_x(xs[2], 2);
  write_x(xs[0], 0);
  fetch_x(xs[0], 3);
  __syncthreads();
  for (int ss = 0; ss < nsr; ss += QD_SETS) {
#pragma unroll
    for (int u = 0; u < QD_SETS; ++u) {
      write_x(xs[(u + 1) % 3], (ss + u + 1) & 1);  // step ss+u+1, fetched three iterations ago
      fetch_x(xs[(u + 1) % 3], ss + u + 4);
      if (ss + u < ns) compute(sets[u], (ss + u) & 1, ss + u);
      fetch_w(sets[u], ss + u + QD_SETS);
      __syncthreads();
    }
  }
  if (row < M) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int t = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (t < T) {
          const size_t o = (size_t)t * M + row;
          y[o] = resid ? resid[o] + acc[i][e] : acc[i][e];
        }
      }
  }
}

// ---- producers of the activation planes ----------------------------------------------------------------------------
// planes[p][t][k] bf16, t < Tpad (rows >= T are never written: a token row only ever meets its own output row)
enum SplitMode { SPLIT_PLAIN = 0, SPLIT_RMSNORM = 1, SPLIT_SWIGLU = 2 };

// One workgroup per (token, 2048-column slice): 8 consecutive columns per thread, 16-byte plane stores. RMSNORM:
// rmsnorm_batch_cpu (utils.cpp:613-646): x * rsqrt(mean(x^2) + eps) * w; every slice's workgroup sums the squares of the
// whole row itself (same order in all of them). SWIGLU: the batched form of cpu_batch_processor.cpp:252-256:
// (g / (1 + exp(-g))) * u. K is a multiple of 32.
constexpr int SPLIT_SLICE = 2048;
// DEF: an input is still the partial sums of a K-split GEMM (GemmParts: the reduction that ksplit_reduce_kernel would
// have run is done here, same order: residual + (slice 0 + slice 1 + ...)). RMSNORM: x is the residual stream, d.a its
// pending projection; the new residual stream goes to d.xout (a different buffer: the other column slice's workgroup of
// the same token still reads x). SWIGLU: d.a replaces x (gate), d.b replaces aux (up).
struct SplitDefer {
  GemmParts a, b;
  float* xout;
};
__device__ __forceinline__ float parts_sum(const GemmParts& g, size_t o) {
  float acc = g.part[o];
  for (int z = 1; z < g.S; ++z) acc += g.part[(size_t)z * g.n + o];
  return acc;
}
__device__ __forceinline__ f32x4 parts_sum4(const GemmParts& g, size_t o) {
  f32x4 acc = *reinterpret_cast<const f32x4*>(g.part + o);
  for (int z = 1; z < g.S; ++z) acc += *reinterpret_cast<const f32x4*>(g.part + (size_t)z * g.n + o);
  return acc;
}
template <int MODE, bool DEF>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, const float* __restrict__ aux,
                                                    __bf16* __restrict__ planes, size_t plane_stride, int K, int np,
                                                    float eps, float* __restrict__ xsum, int tpad, SplitDefer d) {
  __shared__ float red[4];
  const int t = blockIdx.x;
  const float* xt = x + (size_t)t * K;
  float nf = 1.f;
  if (MODE == SPLIT_RMSNORM) {
    float ssq = 0.f;
    if (DEF) {
      for (int i = threadIdx.x; i < K; i += 256) {
        const float v = xt[i] + parts_sum(d.a, (size_t)t * K + i);
        ssq += v * v;
      }
    } else {
      for (int i = threadIdx.x; i < K; i += 256) ssq += xt[i] * xt[i];
    }
    for (int o = 32; o >= 1; o >>= 1) ssq += __shfl_xor(ssq, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ssq;
    __syncthreads();
    nf = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
  }
  const int i0 = blockIdx.y * SPLIT_SLICE + threadIdx.x * 8;
  if (i0 >= K) return;
  float v[8];
  f32x4 lo, hi;
  const size_t o0 = (size_t)t * K + i0;
  if (DEF && MODE == SPLIT_SWIGLU && d.a.S) {
    lo = parts_sum4(d.a, o0);
    hi = parts_sum4(d.a, o0 + 4);
  } else {
    lo = *reinterpret_cast<const f32x4*>(xt + i0);
    hi = *reinterpret_cast<const f32x4*>(xt + i0 + 4);
    if (DEF && MODE == SPLIT_RMSNORM) {
      lo = lo + parts_sum4(d.a, o0);
      hi = hi + parts_sum4(d.a, o0 + 4);
      *reinterpret_cast<f32x4*>(d.xout + o0) = lo;
      *reinterpret_cast<f32x4*>(d.xout + o0 + 4) = hi;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] = lo[e];
    v[4 + e] = hi[e];
  }
  if (MODE != SPLIT_PLAIN) {
    f32x4 alo, ahi;
    if (DEF && MODE == SPLIT_SWIGLU && d.b.S) {
      alo = parts_sum4(d.b, o0);
      ahi = parts_sum4(d.b, o0 + 4);
    } else {
      const float* ap = MODE == SPLIT_RMSNORM ? aux + i0 : aux + (size_t)t * K + i0;
      alo = *reinterpret_cast<const f32x4*>(ap);
      ahi = *reinterpret_cast<const f32x4*>(ap + 4);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = e < 4 ? alo[e] : ahi[e - 4];
      if (MODE == SPLIT_RMSNORM) v[e] = v[e] * nf * a;
      else v[e] = (v[e] / (1.0f + expf(-v[e]))) * a;
    }
  }
  // sum of the 32-column block (4 threads): the integer-plane GEMM folds Q4_K's block minimum in through it
  float bs = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  bs += __shfl_xor(bs, 1);
  bs += __shfl_xor(bs, 2);
  if ((threadIdx.x & 3) == 0) xsum[(size_t)(i0 >> 5) * tpad + t] = bs;
  for (int p = 0; p < np; ++p) {
    bf16x8 b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      b[e] = (__bf16)v[e];
      v[e] -= (float)b[e];
    }
    *reinterpret_cast<bf16x8*>(planes + p * plane_stride + (size_t)t * K + i0) = b;
  }
}

// RMSNORM over a residual stream whose last projection is still pending as K-split partial tiles: one workgroup of 1024
// threads per token. Phase A sums residual + slices (vector loads, all in flight together), writes the new residual
// stream (in place is fine: the row belongs to this workgroup) and keeps the row in LDS; phase B is split_kernel's sum of
// squares, thread for thread (256 accumulators over columns j, j + 256, ..., butterfly per wave, four wave sums in
// order); phase C is split_kernel's plane write with its thread -> 8 columns mapping. Bit-identical to
// ksplit_reduce_kernel followed by split_kernel<SPLIT_RMSNORM>. K is a multiple of 32; LDS: K floats.
__global__ __launch_bounds__(1024) void split_rmsnorm_deferred_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      __bf16* __restrict__ planes, size_t plane_stride, int K,
                                                                      int np, float eps, float* __restrict__ xsum, int tpad,
                                                                      GemmParts a, float* __restrict__ xout) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)t * K;
  for (int i0 = tid * 4; i0 < K; i0 += 4096) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(x + base + i0) + parts_sum4(a, base + i0);
    *reinterpret_cast<f32x4*>(xout + base + i0) = r;
    *reinterpret_cast<f32x4*>(rowbuf + i0) = r;
  }
  __syncthreads();
  if (tid < 256) {
    float ssq = 0.f;
    for (int i = tid; i < K; i += 256) ssq += rowbuf[i] * rowbuf[i];
    for (int o = 32; o >= 1; o >>= 1) ssq += __shfl_xor(ssq, o);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
  }
  __syncthreads();
  const float nf = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
  for (int i0 = tid * 8; i0 < K; i0 += 8192) {
    float v[8];
    const f32x4 lo = *reinterpret_cast<const f32x4*>(rowbuf + i0), hi = *reinterpret_cast<const f32x4*>(rowbuf + i0 + 4);
    const f32x4 alo = *reinterpret_cast<const f32x4*>(w + i0), ahi = *reinterpret_cast<const f32x4*>(w + i0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xv = e < 4 ? lo[e] : hi[e - 4], av = e < 4 ? alo[e] : ahi[e - 4];
      v[e] = xv * nf * av;
    }
    float bs = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    bs += __shfl_xor(bs, 1);
    bs += __shfl_xor(bs, 2);
    if ((tid & 3) == 0) xsum[(size_t)(i0 >> 5) * tpad + t] = bs;
    for (int p = 0; p < np; ++p) {
      bf16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        b[e] = (__bf16)v[e];
        v[e] -= (float)b[e];
      }
      *reinterpret_cast<bf16x8*>(planes + p * plane_stride + base + i0) = b;
    }
  }
}

int planes_of_terms(int terms) { return terms == 1 ? 1 : terms == 3 ? 2 : 3; }

// y[o] = (resid ? resid[o] : 0) + sum over the K slices, in slice order
__global__ void ksplit_reduce_kernel(const float* __restrict__ part, int S, size_t n, const float* __restrict__ resid,
                                     float* __restrict__ y) {
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  float acc = part[o];
  for (int z = 1; z < S; ++z) acc += part[(size_t)z * n + o];
  y[o] = resid ? resid[o] + acc : acc;
}
static bool ksplit_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("TL_GEMM_KSPLIT");  // 0: never split K (A/B runs)
    on = e ? atoi(e) != 0 : 1;
  }
  return on != 0;
}
static int gemm_num_cus() { return tl_num_cus(); }
// set by tl_launch_gemm_q around its dispatch when the caller takes over the reduction (GemmDefer): a K-split launch then
// leaves its partial tiles in the caller's buffer and launches no ksplit_reduce_kernel
static thread_local GemmDefer* t_defer = nullptr;
// the workspace of a stream's K-split GEMMs: launches of one stream are ordered, launches of different streams (q | k, v
// and gate | up run side by side) must not share one. Eight slots; a ninth stream takes over the slot used longest
// ago after a device-wide sync (streams of freed models leave their slots behind); -1 only if that sync fails.
static int ksplit_slot(hipStream_t s) {
  static std::mutex mu;
  static hipStream_t seen[8];
  static unsigned long long used[8];
  static unsigned long long tick = 0;
  static int n = 0;
  std::lock_guard<std::mutex> lock(mu);
  ++tick;
  for (int i = 0; i < n; ++i)
    if (seen[i] == s) {
      used[i] = tick;
      return TL_SCRATCH_GEMM_PART0 + i;
    }
  int slot = n;
  if (n == 8) {
    slot = 0;
    for (int i = 1; i < 8; ++i)
      if (used[i] < used[slot]) slot = i;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
  } else {
    ++n;
  }
  seen[slot] = s;
  used[slot] = tick;
  return TL_SCRATCH_GEMM_PART0 + slot;
}

// where a K-split launch of S slices puts its partial tiles: the caller's buffer when the reduction is deferred (and the
// buffer is large enough), else the stream's workspace slot; S = 1 when neither is to be had
static int ksplit_workspace(hipStream_t s, int T, int rows, int* S, float** part, bool* deferred) {
  *part = nullptr;
  *deferred = false;
  if (*S <= 1) return 0;
  const size_t floats = (size_t)*S * T * rows;
  if (t_defer && t_defer->buf && t_defer->cap_floats >= floats) {
    *part = t_defer->buf;
    *deferred = true;
    return 0;
  }
  const int slot = ksplit_slot(s);
  if (slot < 0) {
    *S = 1;
    return 0;
  }
  void* p = nullptr;
  TL_TRY(tl_scratch(slot, sizeof(float) * floats, &p));
  *part = static_cast<float*>(p);
  return 0;
}
static void ksplit_finish(const float* part, int S, size_t n, const float* resid, float* y, bool deferred, hipStream_t s) {
  if (deferred) {
    t_defer->got = {part, n, S};
    return;
  }
  ksplit_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(part, S, n, resid, y);
}

template <int TYPE, int NP, int NT, int BM>
int launch_gemm_inst(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  const int mtiles = (w.rows + BM - 1) / BM, ntiles = (T + GB_N - 1) / GB_N;
  constexpr size_t lds = gemm_lds_bytes<NP, BM>();
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemm_q_kernel<TYPE, NP, NT, BM>, (int)lds, lds_raised));
  gemm_q_kernel<TYPE, NP, NT, BM><<<dim3(mtiles * ntiles), dim3(GEMM_THREADS), lds, s>>>(w, xp, ps, y, resid, T, mtiles, ntiles);
  TL_HIP(hipGetLastError());
  return 0;
}

template <int TYPE, int NP, int NT, int BM, int BN>
int launch_gemm_ws_bn(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  const int mtiles = (w.rows + BM - 1) / BM, ntiles = (T + BN - 1) / BN;
  constexpr size_t lds = (size_t)2 * NP * (BM + BN) * LROW * sizeof(__bf16);
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemm_q_ws_kernel<TYPE, NP, NT, BM, BN>, (int)lds, lds_raised));
  // Skinny problems (few token panels: the multi-sequence decode step, short prompts) have fewer tiles than the chip has
  // workgroup slots (2 per CU): split K over up to 8 slices so that every CU streams weights, partial tiles into a
  // per-stream workspace, summed in slice order (deterministic) with the residual by one small kernel.
  const int tiles = mtiles * ntiles, nk_all = w.cols / GB_K;
  int S = 1;
  if (ksplit_enabled() && tiles < 2 * gemm_num_cus()) {
    S = (2 * gemm_num_cus() + tiles - 1) / tiles;
    if (S > 8) S = 8;
    if (S > nk_all / 12) S = nk_all / 12;  // at least a dozen steps per slice
    if (S < 1) S = 1;
    const int kper = (nk_all + S - 1) / S;
    S = (nk_all + kper - 1) / kper;  // no empty slice
  }
  float* part = nullptr;
  bool deferred = false;
  TL_TRY(ksplit_workspace(s, T, w.rows, &S, &part, &deferred));
  if (S == 1) {
    gemm_q_ws_kernel<TYPE, NP, NT, BM, BN><<<dim3(tiles), dim3(WS_THREADS), lds, s>>>(w, xp, ps, y, resid, T, mtiles, ntiles);
  } else {
    gemm_q_ws_kernel<TYPE, NP, NT, BM, BN><<<dim3(tiles, S), dim3(WS_THREADS), lds, s>>>(w, xp, ps, part, nullptr, T, mtiles, ntiles);
    ksplit_finish(part, S, (size_t)T * w.rows, resid, y, deferred, s);
  }
  TL_HIP(hipGetLastError());
  return 0;
}
// token tile by problem size: 64 tokens when there are no more (TL_GEMM_BN=128 keeps the wide tile for A/B runs)
template <int TYPE, int NP, int NT, int BM>
int launch_gemm_ws(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  static int narrow = -1;
  if (narrow < 0) {
    const char* e = getenv("TL_GEMM_BN");
    narrow = !(e && atoi(e) == 128);
  }
  if (narrow && T <= 64) return launch_gemm_ws_bn<TYPE, NP, NT, BM, 64>(w, xp, ps, y, resid, T, s);
  return launch_gemm_ws_bn<TYPE, NP, NT, BM, GB_N>(w, xp, ps, y, resid, T, s);
}

// K slices of an integer-plane launch of `tiles` tiles
// (workgroups wanted on the chip: one per CU up to 128 tokens, two beyond. Measured, whole steps: 64 / 128 sequences 7.79 /
// 12.03 ms with one against 7.94 / 12.49 with two, a 128-token prefill 11.56 against 12.01 ms, but 512 tokens 35.4
// against 34.8 ms)
static int qi_slices(int tiles, int nk_all, int T) {
  int S = 1;
  const int want = (T <= 128 ? 1 : 2) * gemm_num_cus();
  if (ksplit_enabled() && tiles < want) {
    S = (want + tiles - 1) / tiles;
    if (S > 8) S = 8;
    if (S > nk_all / 12) S = nk_all / 12;
    if (S < 1) S = 1;
    const int kper = (nk_all + S - 1) / S;
    S = (nk_all + kper - 1) / kper;
  }
  return S;
}
static bool qi_narrow_always() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("TL_GEMM_QI_BN");  // 64: 64-token tiles for every problem size (A/B runs)
    v = e && atoi(e) == 64;
  }
  return v != 0;
}
// ---- expanded prefill planes: registry (keyed by the matrix's first plane pointer) and launch ------------------------------
struct PxEntry {
  __bf16* q;
  float* sc;
  size_t bytes;
};
static std::mutex g_px_mu;
static std::atomic<long> g_px_launches{0};
static std::unordered_map<const void*, PxEntry> g_px;
static bool px_enabled() {  // TL_GEMM_PX=0: ignore the expanded planes (A/B runs; read at every launch)
  const char* e = getenv("TL_GEMM_PX");
  return e ? atoi(e) != 0 : true;
}
static bool px_lookup(const QW& w, PxEntry* out) {
  std::lock_guard<std::mutex> lock(g_px_mu);
  auto it = g_px.find(w.p0);
  if (it == g_px.end()) return false;
  *out = it->second;
  return true;
}
static bool px_ready(const QW* w, int n) {
  if (!px_enabled()) return false;
  PxEntry e;
  for (int i = 0; i < n; ++i)
    if (!px_lookup(w[i], &e)) return false;
  return true;
}
// Row-tile size and K slices of an expanded-plane launch of n matrices over `ntiles` token tiles and nk_all 32-column steps.
// One workgroup per CU, so the cost of a launch is (rounds of workgroups over the chip) x (steps of a workgroup + the
// ring fill and the tile store, ~8 steps' worth), plus the partial tiles the consumer has to sum; a 128-row step costs
// ~0.6 of a 256-row one (measured), and is taken where 256-row tiles leave CUs idle or slices short (attn_output / ffn_down /
// q | k | v at 128 tokens). TL_GEMM_PX_SLICES=0: 256 rows and the unpacking kernel's slice counts, which make the two
// kernels agree bit for bit (parity runs).
struct PxPlan {
  int bm, S;
};
static PxPlan px_plan(const QW* w, int n, int ntiles, int nk_all, int bn = PX_BN) {
  const int cus = gemm_num_cus();
  PxPlan best = {PX_BM, 1};
  long best_cost = -1;
  for (int bm : {256, 128}) {
    if (bn == 64 && bm == 128) continue;  // (64-token tiles take whole records)
    long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long)((w[i].rows + bm - 1) / bm) * ntiles;
    for (int S = 1; S <= (ksplit_enabled() ? 8 : 1); ++S) {
      const int kper = (nk_all + S - 1) / S;
      if (S > 1 && (kper < 12 || (nk_all + kper - 1) / kper != S)) continue;  // a dozen steps per slice at least, no empty slice
      const long rounds = (tiles * S + cus - 1) / cus;
      const long cost = rounds * (bm == 256 && bn == PX_BN ? 10L * kper + 80 : 6L * kper + 60) + (S > 1 ? 20L * S : 0);
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best = {bm, S};
      }
    }
  }
  return best;
}
static bool px_own_slices() {
  const char* e = getenv("TL_GEMM_PX_SLICES");  // (read at every launch)
  return e ? atoi(e) != 0 : true;
}
// the expanded-plane launch of n <= 3 matrices (same format and width) with S[i] K slices each; false: not every matrix has
// its planes (the caller takes the unpacking kernel)
template <int TYPE, int NPX, int BM, int BN = PX_BN>
int launch_gemm_px_bm(const PSegs& g, int mt, int ntiles, int smax, const __bf16* xp, size_t ps, const float* xsum, int tpad,
                      const float* resid, int T, int K, hipStream_t s) {
  constexpr size_t lds = PX_NB * px_buf_bytes<NPX>();
  static_assert(lds <= 160 * 1024, "three LDS buffers per workgroup");
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemm_px_kernel<TYPE, NPX, BM, BN>, (int)lds, lds_raised));
  gemm_px_kernel<TYPE, NPX, BM, BN><<<dim3(mt * ntiles, smax), dim3(PX_THREADS), lds, s>>>(g, xp, ps, xsum, tpad, resid, T, K, mt, ntiles);
  return 0;
}
template <int TYPE, int NPX>
int launch_gemm_px(const QW* w, int n, const __bf16* xp, size_t ps, float* const* out, const int* S, const float* resid, int T,
                   hipStream_t s, bool* done, int bm = PX_BM, int bn = PX_BN) {
  *done = false;
  if (!px_enabled()) return 0;
  PxEntry e[3];
  for (int i = 0; i < n; ++i)
    if (!px_lookup(w[i], &e[i])) return 0;
  if (const char* f = getenv("TL_GEMM_PX_BM")) bm = atoi(f) == 128 ? 128 : PX_BM;  // (parity / A/B runs; read at every launch)
  const float* xsum = reinterpret_cast<const float*>(xp + 3 * ps);
  if (bn == 64) bm = PX_BM;
  const int tpad = (int)tl_gemm_padded_tokens(T), ntiles = (T + bn - 1) / bn;
  PSegs g = {};
  int mt = 0, smax = 1;
  for (int i = 0; i < 3; ++i) {
    const int k = i < n ? i : n - 1;  // (unused slots repeat the last matrix; their tile range is empty)
    g.q[i] = e[k].q;
    g.sc[i] = e[k].sc;
    g.y[i] = out[k];
    g.M[i] = w[k].rows;
    g.S[i] = S[k];
    if (i < n) mt += (w[i].rows + bm - 1) / bm;
    g.mt_end[i] = mt;
    if (S[k] > smax) smax = S[k];
  }
  if (bn == 64) TL_TRY((launch_gemm_px_bm<TYPE, NPX, 256, 64>(g, mt, ntiles, smax, xp, ps, xsum, tpad, resid, T, w[0].cols, s)));
  else if (bm == 128) TL_TRY((launch_gemm_px_bm<TYPE, NPX, 128>(g, mt, ntiles, smax, xp, ps, xsum, tpad, resid, T, w[0].cols, s)));
  else TL_TRY((launch_gemm_px_bm<TYPE, NPX, 256>(g, mt, ntiles, smax, xp, ps, xsum, tpad, resid, T, w[0].cols, s)));
  TL_HIP(hipGetLastError());
  ++g_px_launches;
  *done = true;
  return 0;
}

// n matrices in one launch (gemm_qi_ws_kernel NSEG = 3); S[i] slices for matrix i, outputs / partial tiles in out[i]
template <int TYPE, int NPX, int BN, int BM>
int launch_gemm_qi_multi(const QW* w, int n, const __bf16* xp, size_t ps, float* const* out, const int* S, int T, hipStream_t s) {
  if constexpr (BM == 128) {  // the same sums from the expanded planes, where every matrix has them
    bool done = false;
    const int bm = px_own_slices() && px_ready(w, n) ? px_plan(w, n, (T + BN - 1) / BN, w[0].cols / GB_K, BN).bm : PX_BM;
    TL_TRY((launch_gemm_px<TYPE, NPX>(w, n, xp, ps, out, S, nullptr, T, s, &done, bm, BN)));
    if (done) return 0;
  }
  constexpr size_t lds = gemm_qi_lds_bytes<NPX, BN, BM>();
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemm_qi_ws_kernel<TYPE, NPX, BN, BM, 3>, (int)lds, lds_raised));
  const float* xsum = reinterpret_cast<const float*>(xp + 3 * ps);
  const int tpad = (int)tl_gemm_padded_tokens(T), ntiles = (T + BN - 1) / BN;
  MSegs g = {};
  int mt = 0, smax = 1;
  for (int i = 0; i < 3; ++i) {
    const int k = i < n ? i : n - 1;  // (unused slots repeat the last matrix; their tile range is empty)
    g.w[i] = w[k];
    g.y[i] = out[k];
    g.S[i] = S[k];
    if (i < n) mt += (w[i].rows + BM - 1) / BM;
    g.mt_end[i] = mt;
    if (S[k] > smax) smax = S[k];
  }
  gemm_qi_ws_kernel<TYPE, NPX, BN, BM, 3><<<dim3(mt * ntiles, smax), dim3(WS_THREADS), lds, s>>>(g, xp, ps, xsum, tpad, nullptr, T, mt, ntiles);
  TL_HIP(hipGetLastError());
  return 0;
}
template <int TYPE, int NPX, int BN, int BM>
int launch_gemm_qi_bn(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  const int mtiles = (w.rows + BM - 1) / BM, ntiles = (T + BN - 1) / BN;
  constexpr size_t lds = gemm_qi_lds_bytes<NPX, BN, BM>();
  static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
  TL_TRY(tl_raise_lds_limit(gemm_qi_ws_kernel<TYPE, NPX, BN, BM, 1>, (int)lds, lds_raised));
  const float* xsum = reinterpret_cast<const float*>(xp + 3 * ps);
  const int tpad = (int)tl_gemm_padded_tokens(T);
  const int tiles = mtiles * ntiles;
  int S = qi_slices(tiles, w.cols / GB_K, T);  // split K as in launch_gemm_ws_bn
  int px_bm = PX_BM;
  if constexpr (BM == 128) {
    if (px_own_slices() && px_ready(&w, 1)) {
      const PxPlan pl = px_plan(&w, 1, ntiles, w.cols / GB_K, BN);
      S = pl.S;
      px_bm = pl.bm;
    }
  }
  float* part = nullptr;
  bool deferred = false;
  TL_TRY(ksplit_workspace(s, T, w.rows, &S, &part, &deferred));
  if constexpr (BM == 128) {  // the same sums (same K slices) from the expanded planes, if this matrix has them
    bool done = false;
    float* o = S == 1 ? y : part;
    TL_TRY((launch_gemm_px<TYPE, NPX>(&w, 1, xp, ps, &o, &S, S == 1 ? resid : nullptr, T, s, &done, px_bm, BN)));
    if (done) {
      if (S > 1) ksplit_finish(part, S, (size_t)T * w.rows, resid, y, deferred, s);
      TL_HIP(hipGetLastError());
      return 0;
    }
  }
  MSegs g = {};
  g.w[0] = w;
  g.y[0] = S == 1 ? y : part;
  if (S == 1) {
    gemm_qi_ws_kernel<TYPE, NPX, BN, BM, 1><<<dim3(tiles), dim3(WS_THREADS), lds, s>>>(g, xp, ps, xsum, tpad, resid, T, mtiles, ntiles);
  } else {
    gemm_qi_ws_kernel<TYPE, NPX, BN, BM, 1><<<dim3(tiles, S), dim3(WS_THREADS), lds, s>>>(g, xp, ps, xsum, tpad, nullptr, T, mtiles, ntiles);
    ksplit_finish(part, S, (size_t)T * w.rows, resid, y, deferred, s);
  }
  TL_HIP(hipGetLastError());
  return 0;
}
template <int TYPE, int NPX, int BM>
int launch_gemm_qi_np(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  const bool narrow_always = qi_narrow_always();
  return (T <= 64 || narrow_always) ? launch_gemm_qi_bn<TYPE, NPX, 64, BM>(w, xp, ps, y, resid, T, s)
                                    : launch_gemm_qi_bn<TYPE, NPX, GB_N, BM>(w, xp, ps, y, resid, T, s);
}
template <int TYPE>
int launch_gemm_qi(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int terms, int bm,
                   hipStream_t s) {
  if (bm == 128) {
    if (terms == 3) return launch_gemm_qi_np<TYPE, 2, 128>(w, xp, ps, y, resid, T, s);
    return launch_gemm_qi_np<TYPE, 3, 128>(w, xp, ps, y, resid, T, s);
  }
  if (terms == 3) return launch_gemm_qi_np<TYPE, 2, 64>(w, xp, ps, y, resid, T, s);
  return launch_gemm_qi_np<TYPE, 3, 64>(w, xp, ps, y, resid, T, s);
}

// the direct kernel: 32*NI-token tiles, NW waves (32 rows each) per workgroup, K split so that every CU has ~2 workgroups
template <int TYPE, int NPX, int NI, int NW>
int launch_gemm_qd_inst(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int S, hipStream_t s) {
  constexpr int BN = 32 * NI, BMD = 32 * NW;
  const int mtiles = (w.rows + BMD - 1) / BMD, ntiles = (T + BN - 1) / BN;
  constexpr size_t lds = 2 * ((size_t)2 * NPX * BN * LROW * sizeof(__bf16) + 2 * BN * sizeof(float));
  const float* xsum = reinterpret_cast<const float*>(xp + 3 * ps);
  const int tpad = (int)tl_gemm_padded_tokens(T);
  const int tiles = mtiles * ntiles;
  float* part = nullptr;
  bool deferred = false;
  TL_TRY(ksplit_workspace(s, T, w.rows, &S, &part, &deferred));
  MSegs g = {};
  g.w[0] = w;
  g.y[0] = S == 1 ? y : part;
  if (S == 1) {
    gemm_qd_kernel<TYPE, NPX, NI, NW, 1><<<dim3(tiles), dim3(64 * NW), lds, s>>>(g, xp, ps, xsum, tpad, resid, T, mtiles, ntiles);
  } else {
    gemm_qd_kernel<TYPE, NPX, NI, NW, 1><<<dim3(tiles, S), dim3(64 * NW), lds, s>>>(g, xp, ps, xsum, tpad, nullptr, T, mtiles, ntiles);
    ksplit_finish(part, S, (size_t)T * w.rows, resid, y, deferred, s);
  }
  TL_HIP(hipGetLastError());
  return 0;
}
// K slices of a direct-kernel launch of `tiles` tiles that should put about `want` workgroups on the chip
static int qd_slices(int tiles, int nk_all, int want) {
  int S = 1;
  if (ksplit_enabled() && tiles < want) {
    S = (want + tiles - 1) / tiles;
    if (S > 8) S = 8;
    if (S > nk_all / 6) S = nk_all / 6;  // at least six 64-column steps per slice
    if (S < 1) S = 1;
    const int kper = (nk_all + S - 1) / S;
    S = (nk_all + kper - 1) / kper;
  }
  return S;
}
// set by tl_launch_gemm_q_multi while it launches the matrices of a group one by one (merging switched off, or no
// buffers for the partial tiles): the slice count it chose for the group, so that the sums do not depend on the route
static thread_local int t_group_slices = 0;
// waves per workgroup and K slices of the direct kernel for one matrix: eight waves (256 rows) share a staged activation
// tile when that still gives every CU two workgroups, else four waves and about ONE workgroup per CU (measured on the
// 32-sequence step: gate | up at 2 slices 5.22 ms, 3: 5.29, 5: 5.55; attn_output / ffn_down, 32 tiles each, want their 8)
static void qd_shape(int rows, int cols, int ntiles, int* NW, int* S) {
  const int nk_all = cols / 64, cus = gemm_num_cus();
  if (t_group_slices > 0) {
    *NW = 4;
    *S = t_group_slices;
    return;
  }
  const int tiles8 = (rows + 255) / 256 * ntiles, tiles4 = (rows + 127) / 128 * ntiles;
  if (tiles4 >= cus) {  // (lm_head: unsplit four-wave workgroups; 32 sequences 5.27 -> 5.22 ms per step, eight waves 5.26)
    *NW = 4;
    *S = 1;
    return;
  }
  const int S8 = qd_slices(tiles8, nk_all, 2 * cus);
  if (tiles8 * S8 >= 2 * cus) {
    *NW = 8;
    *S = S8;
  } else {
    *NW = 4;
    *S = qd_slices(tiles4, nk_all, cus);
  }
}
template <int TYPE, int NPX, int NI>
int launch_gemm_qd_ni(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, hipStream_t s) {
  int NW = 4, S = 1;
  qd_shape(w.rows, w.cols, (T + 32 * NI - 1) / (32 * NI), &NW, &S);

  if (NW == 8) return launch_gemm_qd_inst<TYPE, NPX, NI, 8>(w, xp, ps, y, resid, T, S, s);
  return launch_gemm_qd_inst<TYPE, NPX, NI, 4>(w, xp, ps, y, resid, T, S, s);
}
// n matrices in one launch (gemm_qd_kernel NSEG = 3), four waves per workgroup; S slices for every one of them
template <int TYPE, int NPX, int NI, int NW>
int launch_gemm_qd_multi_nw(const QW* w, int n, const __bf16* xp, size_t ps, float* const* out, int T, int S, hipStream_t s) {
  constexpr int BN = 32 * NI, BMD = 32 * NW;
  constexpr size_t lds = 2 * ((size_t)2 * NPX * BN * LROW * sizeof(__bf16) + 2 * BN * sizeof(float));
  const float* xsum = reinterpret_cast<const float*>(xp + 3 * ps);
  const int tpad = (int)tl_gemm_padded_tokens(T), ntiles = (T + BN - 1) / BN;
  MSegs g = {};
  int mt = 0;
  for (int i = 0; i < 3; ++i) {
    const int k = i < n ? i : n - 1;  // (unused slots repeat the last matrix; their tile range is empty)
    g.w[i] = w[k];
    g.y[i] = out[k];
    if (i < n) mt += (w[i].rows + BMD - 1) / BMD;
    g.mt_end[i] = mt;
  }
  gemm_qd_kernel<TYPE, NPX, NI, NW, 3><<<dim3(mt * ntiles, S), dim3(64 * NW), lds, s>>>(g, xp, ps, xsum, tpad, nullptr, T, mt, ntiles);
  TL_HIP(hipGetLastError());
  return 0;
}
// four waves per workgroup. Eight (the waves of a workgroup share the staged activation tile) measured slower on merged
// launches even where they still fill the chip: 32 sequences 5.67 ms per step against 5.56 (TL_GEMM_MERGE_NW=8 for A/B
// runs; the choice changes no bit)
template <int TYPE, int NPX, int NI>
int launch_gemm_qd_multi(const QW* w, int n, const __bf16* xp, size_t ps, float* const* out, int T, int S, hipStream_t s) {
  static int force = -1;
  if (force < 0) {
    const char* e = getenv("TL_GEMM_MERGE_NW");
    force = e ? atoi(e) : 0;
  }
  const bool eight = force == 8;
  return eight ? launch_gemm_qd_multi_nw<TYPE, NPX, NI, 8>(w, n, xp, ps, out, T, S, s)
               : launch_gemm_qd_multi_nw<TYPE, NPX, NI, 4>(w, n, xp, ps, out, T, S, s);
}
template <int TYPE>
int launch_gemm_qd(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int terms, hipStream_t s) {
  const bool wide = T > 32 && T <= 64;  // one pass with 64-token tiles, else 32-token tiles
  if (terms == 3) {
    return wide ? launch_gemm_qd_ni<TYPE, 2, 2>(w, xp, ps, y, resid, T, s) : launch_gemm_qd_ni<TYPE, 2, 1>(w, xp, ps, y, resid, T, s);
  }
  return wide ? launch_gemm_qd_ni<TYPE, 3, 2>(w, xp, ps, y, resid, T, s) : launch_gemm_qd_ni<TYPE, 3, 1>(w, xp, ps, y, resid, T, s);
}
// token counts up to which the direct kernel is taken (TL_GEMM_DIRECT: 0 = never, at most 64; read at every launch so
// that one process can compare the routes)
static int gemm_direct_max() {
  const char* e = getenv("TL_GEMM_DIRECT");
  const int v = e ? atoi(e) : 32;
  return v < 0 ? 0 : (v > 64 ? 64 : v);
}

template <int TYPE, int BM>
int launch_gemm_ws_bm(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int terms,
                      hipStream_t s) {
  if (terms == 1) return launch_gemm_ws<TYPE, 1, 1, BM>(w, xp, ps, y, resid, T, s);
  if (terms == 3) return launch_gemm_ws<TYPE, 2, 3, BM>(w, xp, ps, y, resid, T, s);
  return launch_gemm_ws<TYPE, 3, 6, BM>(w, xp, ps, y, resid, T, s);
}

template <int TYPE, int BM>
int launch_gemm_bm(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int terms,
                   hipStream_t s) {
  if (terms == 1) return launch_gemm_inst<TYPE, 1, 1, BM>(w, xp, ps, y, resid, T, s);
  if (terms == 3) return launch_gemm_inst<TYPE, 2, 3, BM>(w, xp, ps, y, resid, T, s);
  return launch_gemm_inst<TYPE, 3, 6, BM>(w, xp, ps, y, resid, T, s);
}

template <int TYPE>
int launch_gemm_typed(const QW& w, const __bf16* xp, size_t ps, float* y, const float* resid, int T, int terms,
                      hipStream_t s) {
  // 128-row tiles when they still give every CU two workgroups' worth of tiles, else 64-row tiles
  static int forced = -1, ws = -1, integer_planes = 1;
  if (forced < 0) {
    const char* q = getenv("TL_GEMM_INT");  // 0: the exact-dequantization kernels (three weight planes, 6 / 3 / 1 MFMAs)
    integer_planes = q ? atoi(q) != 0 : 1;
    const char* e = getenv("TL_GEMM_BM");
    forced = e ? atoi(e) : 0;
    const char* v = getenv("TL_GEMM_WS");
    ws = v ? atoi(v) : 1;  // default: the wave-specialised kernel; TL_GEMM_WS=0 selects the unified one (A/B runs)
  }
  // (one MFMA per product leaves nothing to save on the matrix cores: the plain bf16 form stays on the dequantizing kernel)
  if (integer_planes && terms != 1 && T <= gemm_direct_max() && w.cols % 64 == 0) return launch_gemm_qd<TYPE>(w, xp, ps, y, resid, T, terms, s);
  if (integer_planes && terms != 1) return launch_gemm_qi<TYPE>(w, xp, ps, y, resid, T, terms, forced == 64 ? 64 : 128, s);
  const long tiles128 = (long)((w.rows + 127) / 128) * ((T + GB_N - 1) / GB_N);
  if (ws) {  // 128-row tiles halve the LDS fragment traffic per MFMA but leave one workgroup per CU: measured 14 % slower
    const bool big = forced == 128;
    return big ? launch_gemm_ws_bm<TYPE, 128>(w, xp, ps, y, resid, T, terms, s)
               : launch_gemm_ws_bm<TYPE, 64>(w, xp, ps, y, resid, T, terms, s);
  }
  const bool big = forced ? forced == 128 : tiles128 >= 2 * 256;
  return big ? launch_gemm_bm<TYPE, 128>(w, xp, ps, y, resid, T, terms, s)
             : launch_gemm_bm<TYPE, 64>(w, xp, ps, y, resid, T, terms, s);
}

int g_gemm_terms = 6;

}  // namespace

int tl_gemm_default_terms() { return g_gemm_terms; }

// ---- expanded prefill planes (see gemm_px_kernel) --------------------------------------------------------------------------
size_t tl_gemm_expanded_bytes(int rows, int cols) {
  const size_t mtiles = (size_t)(rows + PX_BM - 1) / PX_BM, nk = (size_t)cols / GB_K;
  return mtiles * nk * (PX_WBYTES + PX_SC * sizeof(float));
}
bool tl_gemm_expandable(const tl_qweight* w) {
  return w && (w->type == TL_Q4_K || w->type == TL_Q6_K || w->type == TL_Q8_0 || w->type == TL_Q8_K) && w->cols % 64 == 0 &&
         w->rows > 0;
}
int tl_gemm_expand(const tl_qweight* w, hipStream_t s) {
  TL_REQUIRE(tl_gemm_expandable(w), "tl_gemm_expand: this matrix has no expanded form");
  {
    std::lock_guard<std::mutex> lock(g_px_mu);
    if (g_px.count(w->p0)) return 0;
  }
  const int mtiles = (w->rows + PX_BM - 1) / PX_BM, nk = w->cols / GB_K;
  const size_t qbytes = (size_t)mtiles * nk * PX_WBYTES, bytes = tl_gemm_expanded_bytes(w->rows, w->cols);
  void* base = nullptr;
  hipError_t e = hipMalloc(&base, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    tl_set_error("tl_gemm_expand: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return 1;
  }
  PxEntry ent = {static_cast<__bf16*>(base), reinterpret_cast<float*>(static_cast<unsigned char*>(base) + qbytes), bytes};
  const QW v = qw_view(w);
  const dim3 grid(nk, mtiles);
  switch (w->type) {
    case TL_Q4_K: px_expand_kernel<TL_Q4_K><<<grid, PX_BM, 0, s>>>(v, ent.q, ent.sc, nk); break;
    case TL_Q6_K: px_expand_kernel<TL_Q6_K><<<grid, PX_BM, 0, s>>>(v, ent.q, ent.sc, nk); break;
    case TL_Q8_0: px_expand_kernel<TL_Q8_0><<<grid, PX_BM, 0, s>>>(v, ent.q, ent.sc, nk); break;
    default: px_expand_kernel<TL_Q8_K><<<grid, PX_BM, 0, s>>>(v, ent.q, ent.sc, nk); break;
  }
  e = hipGetLastError();
  if (e != hipSuccess) {
    hipFree(base);
    tl_set_error("tl_gemm_expand: launch failed: %s", hipGetErrorString(e));
    return 1;
  }
  std::lock_guard<std::mutex> lock(g_px_mu);
  g_px[w->p0] = ent;
  return 0;
}
// called by tl_qweight_free: drops the matrix's expanded planes, if any
void tl_gemm_expand_release(const tl_qweight* w) {
  if (!w) return;
  PxEntry ent = {};
  {
    std::lock_guard<std::mutex> lock(g_px_mu);
    auto it = g_px.find(w->p0);
    if (it == g_px.end()) return;
    ent = it->second;
    g_px.erase(it);
  }
  hipFree(ent.q);
}
bool tl_gemm_expanded(const tl_qweight* w) {
  std::lock_guard<std::mutex> lock(g_px_mu);
  return w && g_px.count(w->p0) != 0;
}
size_t tl_gemm_padded_tokens(int T) { return (size_t)((T + GB_N - 1) / GB_N) * GB_N; }
// bytes of an activation-plane buffer with plane_stride = padded tokens x columns: three bf16 planes + the FP32 sums of
// every 32-column block
size_t tl_gemm_planes_bytes(size_t plane_stride) { return 3 * plane_stride * sizeof(__bf16) + plane_stride / 32 * sizeof(float); }

int tl_launch_split(int mode, const float* x, const float* aux, void* planes, size_t plane_stride, int T, int K,
                    int terms, float eps, hipStream_t s, const GemmParts* pa, const GemmParts* pb, float* xout) {
  if (T == 0) return 0;
  const int np = planes_of_terms(terms);
  __bf16* pl = static_cast<__bf16*>(planes);
  TL_REQUIRE(K > 0 && K % 8 == 0 && T <= 65535, "tl_launch_split: bad shape %d x %d", T, K);
  const dim3 grid(T, (K + SPLIT_SLICE - 1) / SPLIT_SLICE);
  float* xsum = reinterpret_cast<float*>(pl + 3 * plane_stride);  // [K / 32][padded tokens], behind the three planes
  const int tpad = (int)tl_gemm_padded_tokens(T);
  SplitDefer d = {};
  if (pa && pa->S) d.a = *pa;
  if (pb && pb->S) d.b = *pb;
  d.xout = xout;
  const bool def = d.a.S || d.b.S;
  if (def) {
    const size_t n = (size_t)T * K;
    TL_REQUIRE(mode != SPLIT_PLAIN && (!d.a.S || d.a.n == n) && (!d.b.S || d.b.n == n), "tl_launch_split: deferred sums of another shape");
    TL_REQUIRE(mode != SPLIT_RMSNORM || (d.a.S && !d.b.S && xout && xout != x), "tl_launch_split: RMSNorm over a pending projection needs a second residual buffer");
  }
  if (mode == SPLIT_RMSNORM) {
    if (def && K % 32 == 0 && (size_t)K * sizeof(float) <= 48 * 1024)
      split_rmsnorm_deferred_kernel<<<T, 1024, (size_t)K * sizeof(float), s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d.a, xout);
    else if (def) split_kernel<SPLIT_RMSNORM, true><<<grid, 256, 0, s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d);
    else split_kernel<SPLIT_RMSNORM, false><<<grid, 256, 0, s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d);
  } else if (mode == SPLIT_SWIGLU) {
    if (def) split_kernel<SPLIT_SWIGLU, true><<<grid, 256, 0, s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d);
    else split_kernel<SPLIT_SWIGLU, false><<<grid, 256, 0, s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d);
  } else {
    split_kernel<SPLIT_PLAIN, false><<<grid, 256, 0, s>>>(x, aux, pl, plane_stride, K, np, eps, xsum, tpad, d);
  }
  TL_HIP(hipGetLastError());
  return 0;
}

// planes: [planes_of_terms(terms)][plane_stride] bf16 with plane_stride >= tl_gemm_padded_tokens(T) * w.cols
// defer != NULL: if the launch splits K and defer->buf holds the partial tiles, no reduction is launched and defer->got
// describes them (the consumer adds `resid` itself); otherwise defer->got.S = 0 and y is complete as usual
static int launch_gemm_dispatch(const QW& w, const __bf16* xp, size_t plane_stride, float* y, const float* resid, int T,
                                int terms, hipStream_t s);
int tl_launch_gemm_q(const QW& w, const void* planes, size_t plane_stride, float* y, const float* resid, int T,
                     int terms, hipStream_t s, GemmDefer* defer) {
  TL_REQUIRE(terms == 1 || terms == 3 || terms == 6, "tl_launch_gemm_q: terms must be 1, 3 or 6 (got %d)", terms);
  TL_REQUIRE(w.cols % GB_K == 0 && w.rows > 0, "tl_launch_gemm_q: bad weight shape %d x %d", w.rows, w.cols);
  TL_REQUIRE(plane_stride >= tl_gemm_padded_tokens(T) * (size_t)w.cols, "tl_launch_gemm_q: activation planes too small");
  if (defer) defer->got = {nullptr, 0, 0};
  if (T == 0) return 0;
  const __bf16* xp = static_cast<const __bf16*>(planes);
  t_defer = defer;
  const int rc = launch_gemm_dispatch(w, xp, plane_stride, y, resid, T, terms, s);
  t_defer = nullptr;
  return rc;
}
int tl_launch_ksplit_reduce(const GemmParts& g, const float* resid, float* y, hipStream_t s) {
  if (!g.S) return 0;
  ksplit_reduce_kernel<<<(unsigned)((g.n + 255) / 256), 256, 0, s>>>(g.part, g.S, g.n, resid, y);
  TL_HIP(hipGetLastError());
  return 0;
}
static int launch_gemm_dispatch(const QW& w, const __bf16* xp, size_t plane_stride, float* y, const float* resid, int T,
                                int terms, hipStream_t s) {
  switch (w.type) {
    case TL_Q4_K: return launch_gemm_typed<TL_Q4_K>(w, xp, plane_stride, y, resid, T, terms, s);
    case TL_Q6_K: return launch_gemm_typed<TL_Q6_K>(w, xp, plane_stride, y, resid, T, terms, s);
    case TL_Q8_0: return launch_gemm_typed<TL_Q8_0>(w, xp, plane_stride, y, resid, T, terms, s);
    case TL_Q8_K: return launch_gemm_typed<TL_Q8_K>(w, xp, plane_stride, y, resid, T, terms, s);
    default: TL_REQUIRE(false, "tl_launch_gemm_q: weight type %d has no MFMA path", w.type);
  }
  return 0;
}

// Up to three matrices over the same activation planes (q | k | v, gate | up) in one launch of the direct small-token
// kernel or of the integer-plane kernel. Taken only when that changes no bit: all matrices of one format and width, and
// for every matrix the number of K slices its own launch would use (the direct kernel needs them equal; then every tile
// computes what it would alone). Outputs: with one slice y[i] directly; with more, the partial tiles go to defer[i] (required) for the consumer
// to sum. *merged = false: nothing was launched, the caller runs the matrices one by one.
int tl_launch_gemm_q_multi(const QW* w, int n, const void* planes, size_t plane_stride, float* const* y, int T, int terms,
                           hipStream_t s, GemmDefer* const* defer, bool* merged) {
  *merged = false;
  const char* e = getenv("TL_GEMM_MERGE");  // 0: one launch per matrix (A/B runs; read at every call)
  const bool on = e ? atoi(e) != 0 : true;
  const char* q = getenv("TL_GEMM_INT");
  const char* bm = getenv("TL_GEMM_BM");
  if ((q && atoi(q) == 0) || n < 2 || n > 3 || T <= 0 || terms == 1) return 0;
  for (int i = 0; i < n; ++i)
    if (w[i].type != w[0].type || w[i].cols != w[0].cols || w[i].cols % 64 != 0 || w[i].rows <= 0) return 0;
  const bool direct = T <= gemm_direct_max();
  if (!direct && (!on || (bm && atoi(bm) == 64))) return 0;  // (the merged integer-plane launch exists for 128-row tiles only)
  // K slices of every matrix as its own launch would choose them
  const int NI = T > 32 ? 2 : 1, BN = (T <= 64 || qi_narrow_always()) ? 64 : GB_N;
  const int ntiles = direct ? (T + 32 * NI - 1) / (32 * NI) : (T + BN - 1) / BN;
  int S[3] = {1, 1, 1};
  if (direct) {  // one slice count for the group, from the tiles of the whole launch (four-wave workgroups)
    int tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (w[i].rows + 127) / 128 * ntiles;
    S[0] = S[1] = S[2] = qd_slices(tiles, w[0].cols / 64, gemm_num_cus());
  } else {
    for (int i = 0; i < n; ++i) S[i] = qi_slices((w[i].rows + 127) / 128 * ntiles, w[i].cols / GB_K, T);
    if (px_own_slices() && px_ready(w, n))  // the expanded-plane kernel: one slice count from the launch's tiles
      S[0] = S[1] = S[2] = px_plan(w, n, ntiles, w[0].cols / GB_K, BN).S;
  }
  TL_REQUIRE(plane_stride >= tl_gemm_padded_tokens(T) * (size_t)w[0].cols, "tl_launch_gemm_q_multi: activation planes too small");
  float* out[3] = {nullptr, nullptr, nullptr};
  bool can_merge = on;
  for (int i = 0; i < n; ++i) {
    if (S[i] == 1) {
      out[i] = y[i];
    } else {
      const size_t floats = (size_t)S[i] * T * w[i].rows;
      if (!defer || !defer[i] || !defer[i]->buf || defer[i]->cap_floats < floats) can_merge = false;
      else out[i] = defer[i]->buf;
    }
  }
  if (!can_merge) {
    if (!direct) return 0;  // (integer-plane route: every matrix keeps its own slice count, the caller's launches are the same)
    // the direct route one matrix at a time, with the group's slice count: the same sums as the merged launch
    for (int i = 0; i < n; ++i) {
      t_group_slices = S[0];
      const int rc1 = tl_launch_gemm_q(w[i], planes, plane_stride, y[i], nullptr, T, terms, s, defer ? defer[i] : nullptr);
      t_group_slices = 0;
      if (rc1) return rc1;
    }
    *merged = true;
    return 0;
  }
  const __bf16* xp = static_cast<const __bf16*>(planes);
  int rc = 0;
#define TL_QD_MULTI(TYPE)                                                                                           \
  rc = terms == 3 ? (NI == 2 ? launch_gemm_qd_multi<TYPE, 2, 2>(w, n, xp, plane_stride, out, T, S[0], s)            \
                             : launch_gemm_qd_multi<TYPE, 2, 1>(w, n, xp, plane_stride, out, T, S[0], s))           \
                  : (NI == 2 ? launch_gemm_qd_multi<TYPE, 3, 2>(w, n, xp, plane_stride, out, T, S[0], s)            \
                             : launch_gemm_qd_multi<TYPE, 3, 1>(w, n, xp, plane_stride, out, T, S[0], s))
#define TL_QI_MULTI(TYPE)                                                                                           \
  rc = terms == 3 ? (BN == 64 ? launch_gemm_qi_multi<TYPE, 2, 64, 128>(w, n, xp, plane_stride, out, S, T, s)        \
                              : launch_gemm_qi_multi<TYPE, 2, GB_N, 128>(w, n, xp, plane_stride, out, S, T, s))     \
                  : (BN == 64 ? launch_gemm_qi_multi<TYPE, 3, 64, 128>(w, n, xp, plane_stride, out, S, T, s)        \
                              : launch_gemm_qi_multi<TYPE, 3, GB_N, 128>(w, n, xp, plane_stride, out, S, T, s))
#define TL_MULTI(TYPE)        \
  if (direct) {               \
    TL_QD_MULTI(TYPE);        \
  } else {                    \
    TL_QI_MULTI(TYPE);        \
  }
  switch (w[0].type) {
    case TL_Q4_K: TL_MULTI(TL_Q4_K) break;
    case TL_Q6_K: TL_MULTI(TL_Q6_K) break;
    case TL_Q8_0: TL_MULTI(TL_Q8_0) break;
    case TL_Q8_K: TL_MULTI(TL_Q8_K) break;
    default: return 0;
  }
#undef TL_MULTI
#undef TL_QI_MULTI
#undef TL_QD_MULTI
  if (rc) return rc;
  for (int i = 0; i < n; ++i)
    if (defer && defer[i]) defer[i]->got = S[i] > 1 ? GemmParts{out[i], (size_t)T * w[i].rows, S[i]} : GemmParts{nullptr, 0, 0};
  *merged = true;
  return 0;
}

extern "C" int tl_qweight_expand(tl_qweight* w, size_t* expanded_bytes, void* stream) {
  TL_REQUIRE(w, "tl_qweight_expand: null handle");
  if (expanded_bytes) *expanded_bytes = 0;
  if (!tl_gemm_expandable(w)) return 0;
  TL_TRY(tl_gemm_expand(w, static_cast<hipStream_t>(stream)));
  if (expanded_bytes) *expanded_bytes = tl_gemm_expanded_bytes(w->rows, w->cols);
  return 0;
}

#ifdef TL_PX_STAMP
extern "C" int tl_debug_px_stamps(unsigned long long* host_out, int clear) {
  if (host_out) TL_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(px_stamps), sizeof(px_stamps)));
  if (clear) {
    static unsigned long long zeros[PX_PW + PX_CW][PX_STAMP_STEPS][8];
    TL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(px_stamps), zeros, sizeof(zeros)));
  }
  return 0;
}
#endif
// Diagnostic hook (include/tlhip_debug.h): launches of the expanded-plane kernel so far in this process
extern "C" int tl_debug_gemm_px_launches(long long* count_out) {
  TL_REQUIRE(count_out, "tl_debug_gemm_px_launches: null argument");
  *count_out = g_px_launches.load();
  return 0;
}

extern "C" int tl_gemm_set_terms(int terms) {
  TL_REQUIRE(terms == 1 || terms == 3 || terms == 6, "tl_gemm_set_terms: terms must be 1, 3 or 6 (got %d)", terms);
  g_gemm_terms = terms;
  return 0;
}

// ---- operator-level batched mat-mul: Y[T][rows] = X[T][cols] W^T ----------------------------------------------------
// T < 8 (or an FP32 matrix): T GEMV launches, each bit-identical to tl_matvec_q. Otherwise the MFMA kernel above with
// the process-wide term count (tl_gemm_set_terms); the activation planes live in the per-device operator workspace
// (tl_scratch): calls that overlap on one device must be serialised by the caller (one host thread per stream).
extern "C" int tl_matmul_q(const tl_qweight* w, const float* x_dev, float* y_dev, int num_tokens, void* stream) {
  TL_REQUIRE(w && x_dev && y_dev && num_tokens >= 0, "tl_matmul_q: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 2..7 tokens: the multi-vector GEMV (one pass over the weights for up to four tokens, tl_gemv.hip)
  if (num_tokens >= 2 && num_tokens < 8 && (w->type == TL_Q4_K || w->type == TL_Q6_K || w->type == TL_Q8_0) &&
      w->cols % 32 == 0 && tl_gemv_nv_capacity(w->cols) >= 2) {
    GemvArgs a = {};
    a.seg[0] = {qw_view(w), y_dev};
    a.nseg = 1;
    a.cols = w->cols;
    a.x = x_dev;
    return tl_launch_gemv_nv(a, num_tokens, s);
  }
  if (num_tokens < 8 || w->type == TL_F32) {
    for (int t = 0; t < num_tokens; ++t)
      TL_TRY(tl_matvec_q(w, x_dev + (size_t)t * w->cols, y_dev + (size_t)t * w->rows, stream));
    return 0;
  }
  const int terms = g_gemm_terms, np = planes_of_terms(terms);
  const size_t ps = tl_gemm_padded_tokens(num_tokens) * (size_t)w->cols;
  void* ws = nullptr;
  (void)np;
  TL_TRY(tl_scratch(TL_SCRATCH_GEMM_PLANES, tl_gemm_planes_bytes(ps), &ws));
  TL_TRY(tl_launch_split(SPLIT_PLAIN, x_dev, nullptr, ws, ps, num_tokens, w->cols, terms, 0.f, s));
  return tl_launch_gemm_q(qw_view(w), ws, ps, y_dev, nullptr, num_tokens, terms, s);
}
