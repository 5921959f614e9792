// ref_shim.cpp — extern "C" doorway into the REAL reference code (test infrastructure only).
//
// This file is ours; it contains no reference code. It is compiled TOGETHER with the reference's
// own sources where they lie under /root/reference (oracle/Makefile target `ref`: quantization.cpp,
// utils.cpp, logger.cpp, gguf_parser.cpp — the "Tier-1" subset that needs no external library), into
// oracle/_ref/libtlref.so. The result is used (a) to validate the C restatement in tl_oracle.c and
// (b) by oracle/make_golden.py to produce the committed fixtures under tests/golden/.
// Each wrapper only marshals plain buffers into the std::vector signatures the reference uses.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "gguf_parser.h"
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
    case GGML_TYPE_BF16: {  // what the CPU forward runs on BF16 matrices (model.cpp:583-749)
      std::vector<uint16_t> w(static_cast<const uint16_t*>(blocks), static_cast<const uint16_t*>(blocks) + (size_t)rows * cols);
      matvec_bf16_f32_vector_cpu(w, xv, out, rows, cols);
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

// ---- the reference's own GGUF parser (load_gguf_meta, gguf_parser.cpp:123) over a file on disk ------------------------
// One file at a time. ref_gguf_open returns the tensor count, or -1 with the parser's exception text in ref_last_error().
static std::unique_ptr<GGUFData> g_gguf;
static std::vector<std::pair<std::string, GGUFMetadataValue>> g_meta;
static std::string g_err;

const char* ref_last_error() { return g_err.c_str(); }

long ref_gguf_open(const char* path, int use_mmap) {
  g_gguf.reset();
  g_meta.clear();
  g_err.clear();
  try {
    g_gguf = std::make_unique<GGUFData>(load_gguf_meta(path, use_mmap != 0));
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
  for (const auto& kv : g_gguf->metadata) g_meta.emplace_back(kv.first, kv.second);
  return static_cast<long>(g_gguf->tensor_infos.size());
}
void ref_gguf_close() {
  g_gguf.reset();
  g_meta.clear();
}
unsigned long long ref_gguf_alignment() { return g_gguf ? g_gguf->data_alignment : 0; }
long ref_gguf_meta_count() { return g_gguf ? static_cast<long>(g_meta.size()) : -1; }

// tensor i as the reference recorded it; returns 0, or -1 for a bad index
int ref_gguf_tensor(long i, char* name, size_t name_len, unsigned* type, unsigned* n_dims, unsigned long long dims[4],
                    unsigned long long* offset, unsigned long long* num_elements, unsigned long long* size_in_bytes) {
  if (!g_gguf || i < 0 || (size_t)i >= g_gguf->tensor_infos.size()) return -1;
  const GGUFTensorInfo& t = g_gguf->tensor_infos[(size_t)i];
  std::snprintf(name, name_len, "%s", t.name.c_str());
  *type = static_cast<unsigned>(t.type);
  *n_dims = static_cast<unsigned>(t.shape.size());
  for (size_t d = 0; d < 4; ++d) dims[d] = d < t.shape.size() ? t.shape[d] : 0;
  *offset = t.offset;
  *num_elements = t.num_elements;
  *size_in_bytes = t.size_in_bytes;
  return 0;
}

// the bytes of tensor i where the reference's consumers read them (weight_management.cpp:856-875: mapped block +
// offset_diff_for_mmap + offset, or the bulk-read vector + offset); returns the byte count copied (<= cap), -1 on error
long ref_gguf_tensor_bytes(long i, void* out, size_t cap) {
  if (!g_gguf || i < 0 || (size_t)i >= g_gguf->tensor_infos.size()) return -1;
  const GGUFTensorInfo& t = g_gguf->tensor_infos[(size_t)i];
  const uint8_t* base = nullptr;
  size_t avail = 0;
  if (g_gguf->mapped_tensor_data && g_gguf->mapped_tensor_data != GGUFData::MMapFailure) {
    base = static_cast<const uint8_t*>(g_gguf->mapped_tensor_data) + g_gguf->offset_diff_for_mmap;
    avail = g_gguf->mapped_tensor_data_size - g_gguf->offset_diff_for_mmap;
  } else {
    base = g_gguf->tensor_data.data();
    avail = g_gguf->tensor_data.size();
  }
  if (!base || t.offset + t.size_in_bytes > avail) return -1;
  const size_t n = t.size_in_bytes < cap ? t.size_in_bytes : cap;
  std::memcpy(out, base + t.offset, n);
  return static_cast<long>(n);
}

// metadata entry i (map order): kind 0 unsigned, 1 signed, 2 float, 3 bool, 4 string, 5 array (u = length)
int ref_gguf_meta(long i, char* key, size_t key_len, int* kind, unsigned long long* u, long long* s, double* f, char* str,
                  size_t str_len) {
  if (!g_gguf || i < 0 || (size_t)i >= g_meta.size()) return -1;
  std::snprintf(key, key_len, "%s", g_meta[(size_t)i].first.c_str());
  *u = 0;
  *s = 0;
  *f = 0.0;
  if (str_len) str[0] = 0;
  std::visit(
      [&](const auto& v) {
        using T = std::decay_t<decltype(v)>;
        if constexpr (std::is_same_v<T, bool>) {
          *kind = 3;
          *u = v ? 1 : 0;
        } else if constexpr (std::is_same_v<T, std::string>) {
          *kind = 4;
          std::snprintf(str, str_len, "%s", v.c_str());
        } else if constexpr (std::is_same_v<T, GGUFArray>) {
          *kind = 5;
          *u = v.len;
        } else if constexpr (std::is_floating_point_v<T>) {
          *kind = 2;
          *f = static_cast<double>(v);
        } else if constexpr (std::is_signed_v<T>) {
          *kind = 1;
          *s = static_cast<long long>(v);
        } else {
          *kind = 0;
          *u = static_cast<unsigned long long>(v);
        }
      },
      g_meta[(size_t)i].second);
  return 0;
}

// ---- timing of the reference's own decode-path pair on one tensor, with no marshalling inside the timed region ------
// dequantize_vector_q{4k,6k,8_0}_to_f32 (quantization.cpp:1054-1223; what ensure_*_dequantized runs once per token and
// tensor, weight_management.cpp:44-196) followed by matvec_f32_f32_vector_cpu (utils.cpp:349): seconds of each, best of
// `reps`. Returns -1 for a type the reference has no whole-tensor dequantizer for.
}  // extern "C" (the helper below is a template)
namespace {
template <class B, class F>
int time_pair(const void* blocks, size_t nblocks, size_t block_elems, const float* x, float* y, int rows, int cols, int reps,
              double* dequant_s, double* matvec_s, F dequant) {
  const std::vector<B> b = as_blocks<B>(blocks, nblocks);
  const std::vector<float> xv(x, x + cols);
  std::vector<float> out;
  double best_d = 1e30, best_m = 1e30;
  for (int r = 0; r < reps; ++r) {
    std::vector<float> w;  // a fresh vector per repetition, as after clear_layer_dequantized_weights
    const auto t0 = std::chrono::steady_clock::now();
    dequant(b, w, nblocks * block_elems, 0);
    const auto t1 = std::chrono::steady_clock::now();
    matvec_f32_f32_vector_cpu(w, xv, out, rows, cols);
    const auto t2 = std::chrono::steady_clock::now();
    const double d = std::chrono::duration<double>(t1 - t0).count(), m = std::chrono::duration<double>(t2 - t1).count();
    if (d < best_d) best_d = d;
    if (m < best_m) best_m = m;
  }
  std::memcpy(y, out.data(), out.size() * sizeof(float));
  *dequant_s = best_d;
  *matvec_s = best_m;
  return 0;
}
}  // namespace

extern "C" {
int ref_time_dequant_matvec(int type, const void* blocks, const float* x, float* y, int rows, int cols, int reps,
                            double* dequant_s, double* matvec_s) {
  const size_t n = (size_t)rows * cols;
  switch (type) {
    case GGML_TYPE_Q4_K:
      return time_pair<block_q4_K>(blocks, n / GGML_QK_K, GGML_QK_K, x, y, rows, cols, reps, dequant_s, matvec_s,
                                   dequantize_vector_q4k_to_f32);
    case GGML_TYPE_Q6_K:
      return time_pair<block_q6_K>(blocks, n / GGML_QK_K, GGML_QK_K, x, y, rows, cols, reps, dequant_s, matvec_s,
                                   dequantize_vector_q6k_to_f32);
    case GGML_TYPE_Q8_0:
      return time_pair<block_q8_0>(blocks, n / GGML_QK8_0, GGML_QK8_0, x, y, rows, cols, reps, dequant_s, matvec_s,
                                   dequantize_vector_q8_0_to_f32);
    default: return -1;
  }
}

}  // extern "C"
