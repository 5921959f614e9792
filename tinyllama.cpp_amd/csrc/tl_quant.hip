// tl_quant.hip — upload/re-layout of GGUF super-blocks into the plane layout (tl_internal.h) and the
// bit-exact dequantizer over that layout.
//
// Reference behaviour restated (not ported): block formats quantization.h:57-124; dequantizers
// quantization.cpp:171-266 (Q4_K, Q6_K), :1009-1052 (Q8_K, Q8_0); half->float :47-90.
#include "tl_internal.h"

size_t tl_type_block_bytes(int type) {
  switch (type) {
    case TL_F32: return 4;
    case TL_Q8_0: return 34;
    case TL_Q4_K: return 144;
    case TL_Q6_K: return 210;
    case TL_Q8_K: return 290;
    default: return 0;
  }
}
int tl_type_block_elems(int type) {
  switch (type) {
    case TL_F32: return 1;
    case TL_Q8_0: return 32;
    case TL_Q4_K:
    case TL_Q6_K:
    case TL_Q8_K: return 256;
    default: return 0;
  }
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct PlaneSizes {
  size_t s0, s1, s2, s3;
};
static PlaneSizes plane_sizes(int type, size_t rows, size_t cols) {
  switch (type) {
    case TL_Q4_K: return {rows * cols / 2, rows * (cols / 256) * 16, 0, 0};
    case TL_Q6_K: return {rows * cols / 2, rows * cols / 4, rows * cols / 16, rows * (cols / 256) * 2};
    case TL_Q8_0: return {rows * cols, rows * (cols / 32) * 2, 0, 0};
    case TL_Q8_K: return {rows * cols, rows * (cols / 256) * 2, 0, 0};
    case TL_F32: return {rows * cols * 4, 0, 0, 0};
    default: return {0, 0, 0, 0};
  }
}
size_t tl_qweight_device_bytes(int type, int rows, int cols) {
  PlaneSizes p = plane_sizes(type, (size_t)rows, (size_t)cols);
  return align256(p.s0) + align256(p.s1) + align256(p.s2) + align256(p.s3);
}

// ---------------------------------------------------------------------------------------------------
// re-layout kernels: one thread per 32-weight chunk. `src` = raw GGUF blocks of the whole matrix.
// ---------------------------------------------------------------------------------------------------
__global__ void relayout_q4k(const uint8_t* __restrict__ src, uint8_t* __restrict__ qs, uint8_t* __restrict__ hdr,
                             size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const size_t blk = c >> 3;
  const int j = (int)(c & 7);
  const uint8_t* b = src + blk * 144;
  const uint8_t* q = b + 16 + 32 * (j >> 1);  // 32 source bytes shared by sub-blocks 2c' and 2c'+1
  const int hi = j & 1;
  uint32_t out[4];
  for (int w = 0; w < 4; ++w) {
    uint32_t v = 0;
    for (int t = 0; t < 4; ++t) {
      const uint8_t a = q[8 * w + t], bb = q[8 * w + 4 + t];
      const uint32_t n0 = hi ? (a >> 4) : (a & 15), n1 = hi ? (bb >> 4) : (bb & 15);
      v |= (n0 | (n1 << 4)) << (8 * t);
    }
    out[w] = v;
  }
  reinterpret_cast<uint4*>(qs)[c] = make_uint4(out[0], out[1], out[2], out[3]);
  if (j == 0) {
    const uint8_t* s = b + 4;
    uint32_t lo_sc = 0, lo_m = 0, hi2 = 0;
    for (int k = 0; k < 8; ++k) {  // get_scale_min_k4, quantization.cpp:145-155
      uint32_t sc, m;
      if (k < 4) {
        sc = s[k] & 63;
        m = s[k + 4] & 63;
      } else {
        sc = (s[k + 4] & 0x0F) | ((s[k - 4] >> 6) << 4);
        m = (s[k + 4] >> 4) | ((s[k] >> 6) << 4);
      }
      lo_sc |= (sc & 15) << (4 * k);
      lo_m |= (m & 15) << (4 * k);
      hi2 |= (sc >> 4) << (2 * k);
      hi2 |= (m >> 4) << (16 + 2 * k);
    }
    const uint32_t dd = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    reinterpret_cast<uint4*>(hdr)[blk] = make_uint4(dd, lo_sc, lo_m, hi2);
  }
}

// weight n of a Q6_K block (quantization.cpp:243-263): 6-bit code before the -32 offset
__device__ inline uint32_t q6k_code(const uint8_t* b, int n) {
  const int half = n >> 7, g = (n >> 5) & 3, l = n & 31;
  const uint8_t qlb = b[64 * half + 32 * (g & 1) + l];
  const uint8_t qhb = b[128 + 32 * half + l];
  const uint32_t lo = (g >= 2) ? (qlb >> 4) : (qlb & 15);
  return lo | (((qhb >> (2 * g)) & 3) << 4);
}

__global__ void relayout_q6k(const uint8_t* __restrict__ src, uint8_t* __restrict__ ql, uint8_t* __restrict__ qh,
                             uint8_t* __restrict__ sc, uint8_t* __restrict__ d, size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const size_t blk = c >> 3;
  const int j = (int)(c & 7);
  const uint8_t* b = src + blk * 210;
  uint32_t lo[4] = {0, 0, 0, 0}, hi[2] = {0, 0};
  for (int i = 0; i < 32; ++i) {
    const uint32_t code = q6k_code(b, 32 * j + i);
    const int w = i >> 3, part = (i >> 2) & 1, t = i & 3;
    lo[w] |= (code & 15) << (8 * t + 4 * part);
    const int v = i >> 4, g = (i >> 2) & 3;
    hi[v] |= (code >> 4) << (8 * t + 2 * g);
  }
  reinterpret_cast<uint4*>(ql)[c] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  reinterpret_cast<uint2*>(qh)[c] = make_uint2(hi[0], hi[1]);
  reinterpret_cast<uint16_t*>(sc)[c] = (uint16_t)b[192 + 2 * j] | ((uint16_t)b[192 + 2 * j + 1] << 8);
  if (j == 0) reinterpret_cast<uint16_t*>(d)[blk] = (uint16_t)b[208] | ((uint16_t)b[209] << 8);
}

// Q8_0 (block = 34 B = one chunk) and Q8_K (block = 290 B = 8 chunks)
__global__ void relayout_q8(const uint8_t* __restrict__ src, uint8_t* __restrict__ qs, uint8_t* __restrict__ d,
                            size_t nchunks, int is_q8k) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const uint8_t* q;
  if (is_q8k) {
    const uint8_t* b = src + (c >> 3) * 290;
    q = b + 2 + 32 * (c & 7);
    if ((c & 7) == 0) reinterpret_cast<uint16_t*>(d)[c >> 3] = (uint16_t)b[0] | ((uint16_t)b[1] << 8);
  } else {
    const uint8_t* b = src + c * 34;
    q = b + 2;
    reinterpret_cast<uint16_t*>(d)[c] = (uint16_t)b[0] | ((uint16_t)b[1] << 8);
  }
  uint32_t o[8];
  for (int w = 0; w < 8; ++w)
    o[w] = (uint32_t)q[4 * w] | ((uint32_t)q[4 * w + 1] << 8) | ((uint32_t)q[4 * w + 2] << 16) |
           ((uint32_t)q[4 * w + 3] << 24);
  uint4* dst = reinterpret_cast<uint4*>(qs) + 2 * c;
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

int tl_qweight_build(int type, const void* host_blocks, int rows, int cols, tl_qweight* w, hipStream_t s) {
  const int be = tl_type_block_elems(type);
  TL_REQUIRE(be > 0, "tl_qweight: unsupported ggml type %d", type);
  TL_REQUIRE(rows > 0 && cols > 0, "tl_qweight: bad shape %d x %d", rows, cols);
  const int col_mult = (type == TL_F32 || type == TL_Q8_0) ? 32 : 256;
  TL_REQUIRE(cols % col_mult == 0, "tl_qweight: cols (%d) must be a multiple of %d for type %d", cols, col_mult,
             type);
  PlaneSizes ps = plane_sizes(type, (size_t)rows, (size_t)cols);
  // (the GEMV / GEMM kernels address a plane with 32-bit byte offsets; + 1280: the whole-KiB over-read slack)
  TL_REQUIRE(type == TL_F32 || ps.s0 + 1280 < (1ull << 32),
             "tl_qweight: a %d x %d matrix of type %d has a plane of %zu bytes (4 GiB and more are not addressable)", rows,
             cols, type, ps.s0);
  uint8_t* base = static_cast<uint8_t*>(w->base);
  w->type = type;
  w->rows = rows;
  w->cols = cols;
  w->p0 = base;
  w->p1 = w->p0 + align256(ps.s0);
  w->p2 = w->p1 + align256(ps.s1);
  w->p3 = w->p2 + align256(ps.s2);
  const size_t nblocks = (size_t)rows * cols / be;
  const size_t raw = nblocks * tl_type_block_bytes(type);
  if (type == TL_F32) {
    TL_HIP(hipMemcpyAsync(w->p0, host_blocks, raw, hipMemcpyHostToDevice, s));
    TL_HIP(hipStreamSynchronize(s));
    return 0;
  }
  uint8_t* staging = nullptr;
  TL_HIP(hipMalloc(&staging, raw));
  hipError_t e = hipMemcpyAsync(staging, host_blocks, raw, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const size_t nchunks = (size_t)rows * cols / 32;
    const unsigned grid = (unsigned)((nchunks + 255) / 256);
    switch (type) {
      case TL_Q4_K: relayout_q4k<<<grid, 256, 0, s>>>(staging, w->p0, w->p1, nchunks); break;
      case TL_Q6_K: relayout_q6k<<<grid, 256, 0, s>>>(staging, w->p0, w->p1, w->p2, w->p3, nchunks); break;
      case TL_Q8_0: relayout_q8<<<grid, 256, 0, s>>>(staging, w->p0, w->p1, nchunks, 0); break;
      case TL_Q8_K: relayout_q8<<<grid, 256, 0, s>>>(staging, w->p0, w->p1, nchunks, 1); break;
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipError_t e2 = hipFree(staging);
  if (e == hipSuccess) e = e2;
  if (e != hipSuccess) {
    tl_set_error("tl_qweight upload failed: %s", hipGetErrorString(e));
    return 1;
  }
  return 0;
}

extern "C" int tl_qweight_upload(int type, const void* host_blocks, int rows, int cols, tl_qweight** out,
                                 void* stream) {
  TL_REQUIRE(out && host_blocks, "tl_qweight_upload: null argument");
  TL_REQUIRE(tl_type_block_elems(type) > 0, "tl_qweight_upload: unsupported ggml type %d", type);
  TL_REQUIRE(rows > 0 && cols > 0, "tl_qweight_upload: bad shape %d x %d", rows, cols);
  tl_qweight* w = new tl_qweight();
  w->bytes = tl_qweight_device_bytes(type, rows, cols);
  // (+ 1280: the ring-loader engine copies whole KiB; the last one of the last plane may read up to 1 KiB past it, tl_stream.hip)
  hipError_t e = hipMalloc(&w->base, w->bytes + 1280);
  if (e != hipSuccess) {
    delete w;
    tl_set_error("tl_qweight_upload: hipMalloc(%zu) failed: %s", w->bytes, hipGetErrorString(e));
    return 1;
  }
  int rc = tl_qweight_build(type, host_blocks, rows, cols, w, static_cast<hipStream_t>(stream));
  if (rc != 0) {
    hipFree(w->base);
    delete w;
    return rc;
  }
  *out = w;
  return 0;
}

extern "C" int tl_qweight_free(tl_qweight* w) {
  if (!w) return 0;
  tl_gemm_expand_release(w);
  hipError_t e = hipFree(w->base);
  delete w;
  TL_HIP(e);
  return 0;
}

extern "C" int tl_qweight_info(const tl_qweight* w, int* type, int* rows, int* cols, size_t* device_bytes) {
  TL_REQUIRE(w, "tl_qweight_info: null handle");
  if (type) *type = w->type;
  if (rows) *rows = w->rows;
  if (cols) *cols = w->cols;
  if (device_bytes) *device_bytes = w->bytes;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// bit-exact dequantizer over the plane layout. Every float op is a separately rounded IEEE op in the
// reference's order: FMA contraction is switched off inside the kernel (HIP's __fmul_rn alone does not
// prevent hipcc from contracting a*b-c).
// ---------------------------------------------------------------------------------------------------
// quantization.cpp:47-90 restated with integer ops (v_cvt_f32_f16 is not used so that the result cannot
// depend on the wave's f16 denormal mode)
__device__ inline float half_bits_to_float(uint32_t h, bool scale_field) {
  const uint32_t sign = (h >> 15) & 1u, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) {
      bits = sign << 31;
    } else {
      const int lz = __clz(m) - 21;  // shifts needed to bring the leading 1 to bit 10
      const uint32_t mm = (m << lz) & 0x3ffu;
      bits = (sign << 31) | ((uint32_t)(113 - lz) << 23) | (mm << 13);
    }
  } else if (e == 0x1f) {
    bits = (sign << 31) | 0x7f800000u | (m << 13);
  } else {
    bits = (sign << 31) | ((e + 112u) << 23) | (m << 13);
  }
  if (scale_field && sign && e != 0x1f && (e | m) != 0) bits &= 0x7fffffffu;  // |d| for finite negatives
  return __uint_as_float(bits);
}

// row of output line lr: row0 + lr; with row_index: row_index[0] + lr, or (gather) row_index[lr], where an id outside
// the table yields a zero line (lookup_embedding, model_utils.cpp:15-19)
__global__ void dequant_rows_kernel(QW w, const int* __restrict__ row_index, int row0, int nrows,
                                    float* __restrict__ out, int gather) {
#pragma clang fp contract(off)
  const int cpr = w.cols >> 5;  // chunks per row
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)nrows * cpr) return;
  const int lr = (int)(gid / cpr), cc = (int)(gid % cpr);
  int row = row_index ? (gather ? row_index[lr] : row_index[0] + lr) : row0 + lr;
  if (row < 0 || row >= w.rows) {
    if (gather) {
      float4* z = reinterpret_cast<float4*>(out + ((size_t)lr * cpr + cc) * 32);
      for (int i = 0; i < 8; ++i) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    row = 0;  // callers validate; never read out of bounds
  }
  const size_t c = (size_t)row * cpr + cc;  // global chunk index
  float y[32];
  switch (w.type) {
    case TL_Q4_K: {
      const uint4 q = reinterpret_cast<const uint4*>(w.p0)[c];
      const uint4 h = reinterpret_cast<const uint4*>(w.p1)[c >> 3];
      const int j = (int)(c & 7);
      const float d = half_bits_to_float(h.x & 0xffffu, false), dmin = half_bits_to_float(h.x >> 16, false);
      const uint32_t sc = ((h.y >> (4 * j)) & 15u) | (((h.w >> (2 * j)) & 3u) << 4);
      const uint32_t mn = ((h.z >> (4 * j)) & 15u) | (((h.w >> (16 + 2 * j)) & 3u) << 4);
      const float d1 = __fmul_rn(d, (float)sc), m1 = __fmul_rn(dmin, (float)mn);
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
      for (int ww = 0; ww < 4; ++ww)
        for (int t = 0; t < 4; ++t) {
          const uint32_t byte = (qq[ww] >> (8 * t)) & 0xffu;
          y[8 * ww + t] = __fsub_rn(__fmul_rn(d1, (float)(byte & 15u)), m1);
          y[8 * ww + 4 + t] = __fsub_rn(__fmul_rn(d1, (float)(byte >> 4)), m1);
        }
      break;
    }
    case TL_Q6_K: {
      const uint4 q = reinterpret_cast<const uint4*>(w.p0)[c];
      const uint2 hh = reinterpret_cast<const uint2*>(w.p1)[c];
      const uint32_t s2 = reinterpret_cast<const uint16_t*>(w.p2)[c];
      const float d = half_bits_to_float(reinterpret_cast<const uint16_t*>(w.p3)[c >> 3], false);
      const float ds[2] = {__fmul_rn(d, (float)(int8_t)(s2 & 0xff)), __fmul_rn(d, (float)(int8_t)(s2 >> 8))};
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
      const uint32_t hq[2] = {hh.x, hh.y};
      for (int i = 0; i < 32; ++i) {
        const int ww = i >> 3, part = (i >> 2) & 1, t = i & 3, v = i >> 4, g = (i >> 2) & 3;
        const uint32_t lo = (qq[ww] >> (8 * t + 4 * part)) & 15u;
        const uint32_t hi = (hq[v] >> (8 * t + 2 * g)) & 3u;
        const int code = (int)(lo | (hi << 4)) - 32;
        y[i] = __fmul_rn(ds[v], (float)code);
      }
      break;
    }
    case TL_Q8_0:
    case TL_Q8_K: {
      const uint4* qp = reinterpret_cast<const uint4*>(w.p0) + 2 * c;
      const uint4 a = qp[0], b = qp[1];
      const size_t di = (w.type == TL_Q8_0) ? c : (c >> 3);
      const float d = half_bits_to_float(reinterpret_cast<const uint16_t*>(w.p1)[di], true);
      const uint32_t qq[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      for (int i = 0; i < 32; ++i) y[i] = __fmul_rn(d, (float)(int8_t)((qq[i >> 2] >> (8 * (i & 3))) & 0xff));
      break;
    }
    default: {  // TL_F32
      const float4* p = reinterpret_cast<const float4*>(w.p0) + 8 * c;
      for (int i = 0; i < 8; ++i) {
        const float4 f = p[i];
        y[4 * i] = f.x; y[4 * i + 1] = f.y; y[4 * i + 2] = f.z; y[4 * i + 3] = f.w;
      }
      break;
    }
  }
  float4* o = reinterpret_cast<float4*>(out + ((size_t)lr * cpr + cc) * 32);
  for (int i = 0; i < 8; ++i) o[i] = make_float4(y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]);
}

int tl_launch_dequant_rows(const tl_qweight* w, const int* row_index_dev, int row0, int nrows, float* out,
                           hipStream_t s) {
  const size_t n = (size_t)nrows * (w->cols / 32);
  if (n == 0) return 0;
  dequant_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(qw_view(w), row_index_dev, row0, nrows, out, 0);
  TL_HIP(hipGetLastError());
  return 0;
}

// out[i][:] = dequantized row ids_dev[i] of the table, i < n (batched embedding lookup)
int tl_launch_dequant_gather(const tl_qweight* w, const int* ids_dev, int n, float* out, hipStream_t s) {
  const size_t total = (size_t)n * (w->cols / 32);
  if (total == 0) return 0;
  dequant_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(qw_view(w), ids_dev, 0, n, out, 1);
  TL_HIP(hipGetLastError());
  return 0;
}

extern "C" int tl_dequantize_rows(const tl_qweight* w, int row0, int nrows, float* out_dev, void* stream) {
  TL_REQUIRE(w && out_dev, "tl_dequantize_rows: null argument");
  TL_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= w->rows, "tl_dequantize_rows: rows [%d,%d) outside [0,%d)",
             row0, row0 + nrows, w->rows);
  return tl_launch_dequant_rows(w, nullptr, row0, nrows, out_dev, static_cast<hipStream_t>(stream));
}

extern "C" int tl_lookup_embedding(const tl_qweight* table, int token_id, float* out_dev, void* stream) {
  TL_REQUIRE(table && out_dev, "tl_lookup_embedding: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (token_id < 0 || token_id >= table->rows) {  // model_utils.cpp:15-19: out-of-range id -> zero vector
    TL_HIP(hipMemsetAsync(out_dev, 0, sizeof(float) * (size_t)table->cols, s));
    return 0;
  }
  return tl_launch_dequant_rows(table, nullptr, token_id, 1, out_dev, s);
}

// ---------------------------------------------------------------------------------------------------
// FP32 -> Q8_K blocks on the device (quantize_fp32_to_q8_K, quantization.cpp:719-770), bit-exact: per 256 values
// d = max|x| / 127 stored through the reference's own float->half rule (quantization.cpp:92-143: the mantissa is bumped
// by one half-ulp step whenever bit 13 is set, which is not IEEE round-to-nearest-even), q = clamp(roundf(x * (1/d))),
// bsums = the sixteen 16-element sums of q. One wave per block, output in the GGUF block layout (290 B).
// ---------------------------------------------------------------------------------------------------
__device__ inline uint32_t float_to_half_bits_ref(float f) {
  const uint32_t x = __float_as_uint(f);
  const uint32_t sign = (x >> 31) & 1u, e32 = (x >> 23) & 0xffu;
  uint32_t m = x & 0x7fffffu;
  if (e32 == 0xff) return (sign << 15) | 0x7c00u | (m != 0 ? 0x200u : 0u);
  int e16 = (int)e32 - 112;
  if (e16 >= 0x1f) return (sign << 15) | 0x7c00u;
  if (e16 <= 0) {
    if (e16 < -10) return sign << 15;
    m = (m | 0x800000u) >> (1 - e16);
    if ((m >> 13) & 1u) m += 1u << 13;
    return (sign << 15) | (m >> 13);
  }
  if ((m >> 13) & 1u) {
    m += 1u << 13;
    if ((m >> 23) == 1u) {
      m = 0;
      if (++e16 >= 0x1f) return (sign << 15) | 0x7c00u;
    }
  }
  return (sign << 15) | ((uint32_t)e16 << 10) | (m >> 13);
}

__global__ __launch_bounds__(64) void quantize_q8k_kernel(const float* __restrict__ x, uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x;
  const float* xb = x + (size_t)blockIdx.x * 256;
  uint8_t* ob = out + (size_t)blockIdx.x * 290;
  const float4 v = reinterpret_cast<const float4*>(xb)[lane];  // values 4*lane .. 4*lane+3
  float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const float d = __fdiv_rn(amax, 127.0f);
  const float id = d != 0.f ? __fdiv_rn(1.0f, d) : 0.0f;
  const float xs[4] = {v.x, v.y, v.z, v.w};
  int q[4], sum = 0;
  for (int i = 0; i < 4; ++i) {
    float r = roundf(__fmul_rn(xs[i], id));
    r = fminf(127.0f, r);
    r = fmaxf(-128.0f, r);
    q[i] = (int)r;
    sum += q[i];
  }
  // 16-element sums: 4 lanes each
  sum += __shfl_xor(sum, 1);
  sum += __shfl_xor(sum, 2);
  if (lane == 0) {
    const uint32_t h = float_to_half_bits_ref(d);
    ob[0] = (uint8_t)(h & 0xff);
    ob[1] = (uint8_t)(h >> 8);
  }
  for (int i = 0; i < 4; ++i) ob[2 + 4 * lane + i] = (uint8_t)(int8_t)q[i];
  if ((lane & 3) == 0) {
    const int16_t s16 = (int16_t)sum;
    ob[258 + 2 * (lane >> 2)] = (uint8_t)((uint16_t)s16 & 0xff);
    ob[259 + 2 * (lane >> 2)] = (uint8_t)((uint16_t)s16 >> 8);
  }
}

extern "C" int tl_quantize_q8_k(const float* x_dev, size_t n, void* blocks_dev, void* stream) {
  TL_REQUIRE(x_dev && blocks_dev, "tl_quantize_q8_k: null argument");
  TL_REQUIRE(n % 256 == 0, "tl_quantize_q8_k: %zu values are not a multiple of the 256-element block", n);
  if (n == 0) return 0;
  quantize_q8k_kernel<<<(unsigned)(n / 256), 64, 0, static_cast<hipStream_t>(stream)>>>(x_dev, static_cast<uint8_t*>(blocks_dev));
  TL_HIP(hipGetLastError());
  return 0;
}
