// ref_shim.cpp — extern "C" doorway into the REAL reference code (test infrastructure only).
//
// This file is ours; it contains no reference code. It is compiled TOGETHER with the reference's
// own sources where they lie under /root/reference (oracle/Makefile target `ref`: quantization.cpp,
// utils.cpp, logger.cpp, gguf_parser.cpp — the "Tier-1" subset that needs no external library), into
// oracle/_ref/libtlref.so. The result is used (a) to validate the C restatement in tl_oracle.c and
// (b) by oracle/make_golden.py to produce the committed fixtures under tests/golden/.
// Each wrapper only marshals plain buffers into the std::vector signatures the reference uses.
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "logger.h"
#include "quantization.h"
#include "utils.h"

namespace {
template <class B>
std::vector<B> as_blocks(const void* p, size_t n) {
  std::vector<B> v(n);
  std::memcpy(static_cast<void*>(v.data()), p, n * sizeof(B));
  return v;
}
std::vector<std::pair<float, float>> as_pairs(const float* table, size_t n_pairs) {
  std::vector<std::pair<float, float>> v(n_pairs);
  for (size_t i = 0; i < n_pairs; ++i) v[i] = {table[2 * i], table[2 * i + 1]};
  return v;
}
}  // namespace

extern "C" {

void ref_init() { Logger::set_level(Logger::Level::OFF); }

float ref_fp16_to_fp32(uint16_t h, int is_scale) { return fp16_to_fp32(h, is_scale != 0); }
uint16_t ref_fp32_to_fp16(float f) { return fp32_to_fp16(f); }

// type ids = GGMLType values
int ref_dequantize(int type, const void* blocks, size_t nblocks, float* out) {
  std::vector<float> o;
  switch (type) {
    case GGML_TYPE_Q4_K: {
      auto b = as_blocks<block_q4_K>(blocks, nblocks);
      dequantize_vector_q4k_to_f32(b, o, nblocks * GGML_QK_K, 0);
      break;
    }
    case GGML_TYPE_Q6_K: {
      auto b = as_blocks<block_q6_K>(blocks, nblocks);
      dequantize_vector_q6k_to_f32(b, o, nblocks * GGML_QK_K, 0);
      break;
    }
    case GGML_TYPE_Q8_0: {
      auto b = as_blocks<block_q8_0>(blocks, nblocks);
      dequantize_vector_q8_0_to_f32(b, o, nblocks * GGML_QK8_0, 0);
      break;
    }
    case GGML_TYPE_Q8_K: {
      auto b = as_blocks<block_q8_K>(blocks, nblocks);
      o.resize(nblocks * GGML_QK_K);
      dequantize_q8_k(b, o, static_cast<int>(nblocks * GGML_QK_K), false);
      break;
    }
    default: return -1;
  }
  std::memcpy(out, o.data(), o.size() * sizeof(float));
  return 0;
}

int ref_quantize_q8_k(const float* x, size_t n, void* blocks_out) {
  std::vector<float> v(x, x + n);
  auto q = quantize_fp32_to_q8_K(v);
  std::memcpy(blocks_out, q.data(), q.size() * sizeof(block_q8_K));
  return static_cast<int>(sizeof(block_q8_K));
}

int ref_matvec_q(int type, const void* blocks, const float* x, float* y, int rows, int cols) {
  std::vector<float> xv(x, x + cols), out;
  switch (type) {
    case GGML_TYPE_Q4_K:
      matvec_q4k_f32_vector_cpu(as_blocks<block_q4_K>(blocks, (size_t)rows * cols / GGML_QK_K), xv, out, rows, cols, false);
      break;
    case GGML_TYPE_Q6_K:
      matvec_q6k_f32_vector_cpu(as_blocks<block_q6_K>(blocks, (size_t)rows * cols / GGML_QK_K), xv, out, rows, cols, false);
      break;
    case GGML_TYPE_Q8_0:
      matvec_q8_0_f32_vector_cpu(as_blocks<block_q8_0>(blocks, (size_t)rows * cols / GGML_QK8_0), xv, out, rows, cols, false);
      break;
    case GGML_TYPE_Q8_K: {
      // NOT matvec_q8k_f32_vector_cpu: it hands dequantize_q8_k an EMPTY output vector (utils.cpp:418-419
      // -> quantization.cpp:1028 writes out of bounds; it segfaults when called). Its evident intent —
      // dequantize, then the f32 Kahan mat-vec (utils.cpp:421) — is what is exercised here instead.
      auto b = as_blocks<block_q8_K>(blocks, (size_t)rows * cols / GGML_QK_K);
      std::vector<float> w((size_t)rows * cols);
      dequantize_q8_k(b, w, rows * cols, false);
      matvec_f32_f32_vector_cpu(w, xv, out, rows, cols);
      break;
    }
    case GGML_TYPE_F32: {
      std::vector<float> w(static_cast<const float*>(blocks), static_cast<const float*>(blocks) + (size_t)rows * cols);
      matvec_f32_f32_vector_cpu(w, xv, out, rows, cols);
      break;
    }
    default: return -1;
  }
  std::memcpy(y, out.data(), out.size() * sizeof(float));
  return 0;
}

int ref_matmul_batch(int type, const void* blocks, const float* x, float* y, int T, int rows, int cols) {
  std::vector<float> xv(x, x + (size_t)T * cols), out;
  switch (type) {
    case GGML_TYPE_Q4_K:
      matmul_q4k_f32_batch_cpu(as_blocks<block_q4_K>(blocks, (size_t)rows * cols / GGML_QK_K), xv, out, T, rows, cols);
      break;
    case GGML_TYPE_Q6_K:
      matmul_q6k_f32_batch_cpu(as_blocks<block_q6_K>(blocks, (size_t)rows * cols / GGML_QK_K), xv, out, T, rows, cols);
      break;
    case GGML_TYPE_Q8_0:
      matmul_q8_0_f32_batch_cpu(as_blocks<block_q8_0>(blocks, (size_t)rows * cols / GGML_QK8_0), xv, out, T, rows, cols);
      break;
    case GGML_TYPE_F32: {
      std::vector<float> w(static_cast<const float*>(blocks), static_cast<const float*>(blocks) + (size_t)rows * cols);
      matmul_f32_f32_batch_cpu(w, xv, out, T, rows, cols);
      break;
    }
    default: return -1;
  }
  std::memcpy(y, out.data(), out.size() * sizeof(float));
  return 0;
}

void ref_rmsnorm_vector(const float* x, const float* w, float* out, int n, float eps) {
  std::vector<float> xv(x, x + n), wv(w, w + n), o;
  rmsnorm_vector_cpu(xv, wv, o, eps);
  std::memcpy(out, o.data(), (size_t)n * sizeof(float));
}

void ref_rmsnorm_batch(const float* x, const float* w, float* out, int T, int n, float eps) {
  std::vector<float> xv(x, x + (size_t)T * n), wv(w, w + n), o;
  rmsnorm_batch_cpu(xv, wv, o, T, n, eps);
  std::memcpy(out, o.data(), (size_t)T * n * sizeof(float));
}

void ref_rope_apply(float* x, int num_heads, int head_dim, int pos, const float* table, int max_pos,
                    int adjacent) {
  std::vector<float> xv(x, x + (size_t)num_heads * head_dim);
  auto tb = as_pairs(table, (size_t)max_pos * head_dim / 2);
  apply_rope_vector(xv, num_heads, head_dim, pos, tb, max_pos, adjacent != 0);
  std::memcpy(x, xv.data(), xv.size() * sizeof(float));
}

void ref_rope_batch(float* q, float* k, int T, int nq, int nkv, int head_dim, int start_pos,
                    const float* table, int max_pos, int adjacent) {
  std::vector<float> qv(q, q + (size_t)T * nq * head_dim), kv(k, k + (size_t)T * nkv * head_dim);
  auto tb = as_pairs(table, (size_t)max_pos * head_dim / 2);
  apply_rope_batch_cpu(qv, kv, T, nq, nkv, head_dim, start_pos, tb, max_pos, adjacent != 0);
  std::memcpy(q, qv.data(), qv.size() * sizeof(float));
  std::memcpy(k, kv.data(), kv.size() * sizeof(float));
}

void ref_softmax(const float* x, float* out, int n) {
  std::vector<float> xv(x, x + n), o;
  softmax_vector_cpu(xv, o);
  std::memcpy(out, o.data(), (size_t)n * sizeof(float));
}

void ref_silu(const float* x, float* out, int n) {
  std::vector<float> xv(x, x + n), o(n);
  silu_cpu(xv, o);
  std::memcpy(out, o.data(), (size_t)n * sizeof(float));
}

}  // extern "C"
