// tl_chunk.h — the per-format 32-weight "chunk" of the plane layout (tl_internal.h) in registers: its loads and its dot
// with 32 input values. Shared by the launch-per-projection GEMV (tl_gemv.hip) and the ring-loader engine (tl_stream.hip).
#pragma once
#include "tl_internal.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XS_STRIDE = 36;   // floats per staged chunk: 32 x + sum(x[0..15]) + sum(x[16..31]) + 2 pad
constexpr int XS_SCRATCH = 64;  // floats reserved at the start of dynamic LDS for block reductions
constexpr int GEMV_MAX_THREADS = 1024;
constexpr int TL_NONE = -1;     // "no second format class" marker for the kernel template

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
  // opaque to the optimiser: with the nibble masks visible it narrows "(v >> 8) & 0xff" to "& 0x0f" and loses the
  // v_cvt_f32_ubyte1..3 byte-select patterns (a shift + an and + ubyte0 per weight instead of one convert)
  asm("" : "+v"(v));
  acc = fmaf(ub0(v), x[0], acc);
  acc = fmaf(ub1(v), x[1], acc);
  acc = fmaf(ub2(v), x[2], acc);
  acc = fmaf(ub3(v), x[3], acc);
  return acc;
}

// Non-temporal load through an explicitly GLOBAL pointer: the segment pointers pass through integer arithmetic
// (group_of), after which the compiler no longer knows their address space and would emit flat_load, whose
// completion is unordered and forces vmcnt(0) waits.
// acc + f16(lo / hi half of p) * x, all in FP32 (v_fma_mix_f32: op_sel_hi marks the fp16 source, op_sel its half)
__device__ __forceinline__ float fma_mix_lo(unsigned p, float x, float acc) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(p), "v"(x));
  return acc;
}
__device__ __forceinline__ float fma_mix_hi(unsigned p, float x, float acc) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(p), "v"(x));
  return acc;
}

template <typename T>
__device__ __forceinline__ T ldg(const uint8_t* base, unsigned byte_off) {
  typedef const T __attribute__((address_space(1))) * gptr;
  return __builtin_nontemporal_load((gptr)(unsigned long long)(base + byte_off));
}
// cached load at a wave-uniform base + 32-bit per-lane byte offset (scalar base + vector offset addressing: no 64-bit
// vector arithmetic per load)
template <typename T>
__device__ __forceinline__ T ldc(const void* base, unsigned byte_off) {
  typedef const T __attribute__((address_space(1))) * gptr;
  return *(gptr)(unsigned long long)(static_cast<const uint8_t*>(base) + byte_off);
}
__device__ __forceinline__ float ldg_f32(const float* p) {
  typedef const float __attribute__((address_space(1))) * gptr;
  return *(gptr)(unsigned long long)p;
}
__device__ __forceinline__ void stg_f32(float* p, float v) {
  typedef float __attribute__((address_space(1))) * gptr;
  *(gptr)(unsigned long long)p = v;
}

// ---- per-format: one 32-weight chunk in registers, its loads and its dot with 32 x ----------------------------
// `base*` arguments are wave-uniform row bases, the per-lane part is a 32-bit byte offset
// a * b - c and the Q6_K tail with the fusing fixed (contraction off around explicit fmaf): see Chunk<TL_Q4_K>::dot
__device__ __forceinline__ float chunk_fms(float a, float b, float c) {
#pragma clang fp contract(off)
  return fmaf(a, b, -c);
}
__device__ __forceinline__ float chunk_q6_tail(float w0, float a0, float s0, float w1, float a1, float s1) {
#pragma clang fp contract(off)
  const float t0 = fmaf(-32.0f, s0, a0), t1 = fmaf(-32.0f, s1, a1);
  return fmaf(w0, t0, w1 * t1);
}

template <int TYPE>
struct Chunk;

// (d*sc) * sum(q*x) - (dmin*m) * sum(x)  ==  sum((d*sc*q - dmin*m) * x)   (quantization.cpp:200-215)
template <>
struct Chunk<TL_Q4_K> {
  u32x4 a, b;  // payload, super-block header
  __device__ __forceinline__ void load(const QW& w, int row, unsigned c) {
    // the row's byte offset inside a plane is a 32-bit product (tl_qweight_upload refuses planes of 4 GiB and more)
    // added to the 64-bit plane base: three scalar instructions per plane; the full 64-bit product cost ten in front of
    // a wave's first weight load. (Carrying the row offset in the LANE offset instead was measured slower: it puts a
    // vector add per load into the steady loop, whose VALU is the scarce resource.)
    const unsigned r = (unsigned)row, cols = (unsigned)w.cols;
    a = ldg<u32x4>(w.p0 + (size_t)(r * (cols >> 1)), c * 16u);
    b = ldg<u32x4>(w.p1 + (size_t)(r * (cols >> 4)), (c >> 3) * 16u);
  }
  // Nibbles never become integers-to-convert: a masked dword IS two fp16 subnormals (n * 2^-24 in each half when
  // the nibble sits in bits 0-3 of the half, n * 2^-20 in bits 4-7), and v_fma_mix_f32 multiplies an fp16 half by an
  // fp32 x into an fp32 accumulator exactly. 5 bit-ops + 8 FMAs per 8 weights instead of 3 + 8 converts + 8 FMAs.
  // Nibble i of dword w is weight 8w + i/2 (i even) or 8w + 4 + i/2 (i odd), see tl_internal.h.
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int j) const {
    float aA = 0.f, aB = 0.f;  // aA: nibbles scaled 2^-24, aB: nibbles scaled 2^-20
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned v = a[w], s = v >> 8;
      const unsigned m0 = v & 0x000F000Fu, m1 = v & 0x00F000F0u, m2 = s & 0x000F000Fu, m3 = s & 0x00F000F0u;
      const float* xw = x + 8 * w;
      aA = fma_mix_lo(m0, xw[0], aA);
      aB = fma_mix_lo(m1, xw[4], aB);
      aA = fma_mix_lo(m2, xw[1], aA);
      aB = fma_mix_lo(m3, xw[5], aB);
      aA = fma_mix_hi(m0, xw[2], aA);
      aB = fma_mix_hi(m1, xw[6], aB);
      aA = fma_mix_hi(m2, xw[3], aA);
      aB = fma_mix_hi(m3, xw[7], aB);
    }
    const float q = fmaf(aB, 0.0625f, aA) * 16777216.0f;  // sum(nibble * x), exact power-of-two rescale
    const float d = half_to_float(b[0] & 0xffffu), dmin = half_to_float(b[0] >> 16);
    const unsigned sc = ((b[1] >> (4 * j)) & 15u) | (((b[3] >> (2 * j)) & 3u) << 4);
    const unsigned mn = ((b[2] >> (4 * j)) & 15u) | (((b[3] >> (16 + 2 * j)) & 3u) << 4);
    // ONE spelled-out fused form: left to the compiler, which of the two products is fused into the subtraction depends
    // on the code around the inlined call, and kernels that must agree bit for bit (gemv_kernel's step schemes, the
    // multi-vector GEMV) would differ in the last place
    return chunk_fms(d * (float)sc, q, (dmin * (float)mn) * (s0 + s1));
  }
};

// d*sc*(q6-32) summed per 16-weight sub-block (quantization.cpp:255-263)
template <>
struct Chunk<TL_Q6_K> {
  u32x4 a;  // ql'
  u32x2 h;  // qh'
  unsigned sc, d;
  __device__ __forceinline__ void load(const QW& w, int row, unsigned c) {
    const unsigned r = (unsigned)row, cols = (unsigned)w.cols;  // (32-bit offsets: see Chunk<TL_Q4_K>::load)
    a = ldg<u32x4>(w.p0 + (size_t)(r * (cols >> 1)), c * 16u);
    h = ldg<u32x2>(w.p1 + (size_t)(r * (cols >> 2)), c * 8u);
    sc = ldg<unsigned short>(w.p2 + (size_t)(r * (cols >> 4)), c * 2u);
    d = ldg<unsigned short>(w.p3 + (size_t)(r * (cols >> 7)), (c >> 3) * 2u);
  }
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int) const {
    float acc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const unsigned hh = h[v];
      const unsigned l0 = a[2 * v], l1 = a[2 * v + 1];
      const unsigned c0 = (l0 & 0x0F0F0F0Fu) | ((hh << 4) & 0x30303030u);
      const unsigned c1 = ((l0 >> 4) & 0x0F0F0F0Fu) | ((hh << 2) & 0x30303030u);
      const unsigned c2 = (l1 & 0x0F0F0F0Fu) | (hh & 0x30303030u);
      const unsigned c3 = ((l1 >> 4) & 0x0F0F0F0Fu) | ((hh >> 2) & 0x30303030u);
      float p = 0.f, q = 0.f;
      p = dot4(c0, x + 16 * v, p);
      q = dot4(c1, x + 16 * v + 4, q);
      p = dot4(c2, x + 16 * v + 8, p);
      q = dot4(c3, x + 16 * v + 12, q);
      acc[v] = p + q;
    }
    const float df = half_to_float(d);
    const float sc0 = (float)(int)(signed char)(sc & 0xffu), sc1 = (float)(int)(signed char)((sc >> 8) & 0xffu);
    return chunk_q6_tail(df * sc0, acc[0], s0, df * sc1, acc[1], s1);  // (spelled-out fused form, see Chunk<TL_Q4_K>::dot)
  }
};

// |d| * sum(q*x): bytes are biased by +128 for the unsigned byte->float convert, folded back with sum(x)
template <int TYPE>
struct ChunkQ8 {
  u32x4 a, b;
  unsigned d;
  __device__ __forceinline__ void load(const QW& w, int row, unsigned c) {
    const unsigned r = (unsigned)row, cols = (unsigned)w.cols;  // (32-bit offsets: see Chunk<TL_Q4_K>::load)
    a = ldg<u32x4>(w.p0 + (size_t)(r * cols), c * 32u);
    b = ldg<u32x4>(w.p0 + (size_t)(r * cols), c * 32u + 16u);
    d = TYPE == TL_Q8_0 ? ldg<unsigned short>(w.p1 + (size_t)(r * (cols >> 4)), c * 2u)
                        : ldg<unsigned short>(w.p1 + (size_t)(r * (cols >> 7)), (c >> 3) * 2u);
  }
  __device__ __forceinline__ float dot(const float* x, float s0, float s1, int) const {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a0 = dot4(a[w] ^ 0x80808080u, x + 4 * w, a0);
      a1 = dot4(b[w] ^ 0x80808080u, x + 16 + 4 * w, a1);
    }
    return scale_field(d) * chunk_fms(-128.0f, s0 + s1, -(a0 + a1));  // (a0 + a1) - 128 (s0 + s1), fused form spelled out
  }
};
template <>
struct Chunk<TL_Q8_0> : ChunkQ8<TL_Q8_0> {};
template <>
struct Chunk<TL_Q8_K> : ChunkQ8<TL_Q8_K> {};

