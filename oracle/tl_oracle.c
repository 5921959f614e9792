/*
 * tl_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY, see tl_oracle.h).
 *
 * A restatement in plain C of what the reference computes on its CPU path; every function names
 * the reference file:line it follows. Float expressions keep the reference's evaluation order and
 * are compiled with -ffp-contract=off so that no multiply-add is fused (the reference's canonical
 * build is baseline x86-64 without FMA).
 */
#include "tl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * half <-> float (quantization.cpp:47-136)
 * ---------------------------------------------------------------------------------------- */
float tlo_fp16_to_fp32(uint16_t h, int is_gguf_scale_field) {
  const uint32_t sign = (uint32_t)(h >> 15) & 1u;
  uint32_t e = (uint32_t)(h >> 10) & 0x1fu;
  uint32_t m = (uint32_t)h & 0x3ffu;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) {
      bits = sign << 31;
    } else {
      /* subnormal half: renormalise (unsigned wrap-around of e is intended, as in the reference) */
      e = 1;
      while ((m & 0x400u) == 0) {
        m <<= 1;
        e--;
      }
      m &= ~0x400u;
      bits = (sign << 31) | ((e - 15u + 127u) << 23) | (m << 13);
    }
  } else if (e == 0x1f) {
    bits = (sign << 31) | (0xffu << 23) | (m << 13);
  } else {
    bits = (sign << 31) | ((e - 15u + 127u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &bits, sizeof f);
  /* quantization.cpp:85-87: scale fields lose their sign when finite and negative */
  if (is_gguf_scale_field && f < 0.0f && !(isnan(f) || isinf(f))) f = fabsf(f);
  return f;
}

uint16_t tlo_fp32_to_fp16(float f) {
  uint32_t x;
  memcpy(&x, &f, sizeof x);
  const uint32_t sign = (x >> 31) & 1u;
  const uint32_t e32 = (x >> 23) & 0xffu;
  uint32_t m32 = x & 0x7fffffu;
  if (e32 == 0xff) return (uint16_t)((sign << 15) | 0x7c00u | (m32 != 0 ? 0x200u : 0u));
  int e16 = (int)e32 - 127 + 15;
  if (e16 >= 0x1f) return (uint16_t)((sign << 15) | 0x7c00u);
  if (e16 <= 0) {
    if (e16 < -10) return (uint16_t)(sign << 15);
    m32 = (m32 | 0x800000u) >> (1 - e16);
    if ((m32 >> 13) & 1u) m32 += (1u << 13); /* the reference's (non-IEEE) rounding rule */
    return (uint16_t)((sign << 15) | (m32 >> 13));
  }
  if ((m32 >> 13) & 1u) {
    m32 += (1u << 13);
    if ((m32 >> 23) == 1u) {
      m32 = 0;
      e16++;
      if (e16 >= 0x1f) return (uint16_t)((sign << 15) | 0x7c00u);
    }
  }
  return (uint16_t)((sign << 15) | ((uint32_t)e16 << 10) | (m32 >> 13));
}

/* ------------------------------------------------------------------------------------------
 * block formats (quantization.h:57-124), byte offsets written out instead of packed structs
 *   Q4_K 144 B: d:f16 @0, dmin:f16 @2, scales[12] @4, qs[128] @16
 *   Q6_K 210 B: ql[128] @0, qh[64] @128, scales[16]:i8 @192, d:f16 @208
 *   Q8_0  34 B: d:f16 @0, qs[32]:i8 @2
 *   Q8_K 290 B: d:f16 @0, qs[256]:i8 @2, bsums[16]:i16 @258
 * ---------------------------------------------------------------------------------------- */
size_t tlo_type_block_bytes(int type) {
  switch (type) {
    case TLO_F32: return 4;
    case TLO_Q8_0: return 34;
    case TLO_Q4_K: return 144;
    case TLO_Q6_K: return 210;
    case TLO_Q8_K: return 290;
    default: return 0;
  }
}

size_t tlo_type_block_elems(int type) {
  switch (type) {
    case TLO_F32: return 1;
    case TLO_Q8_0: return TLO_QK8_0;
    case TLO_Q4_K:
    case TLO_Q6_K:
    case TLO_Q8_K: return TLO_QK_K;
    default: return 0;
  }
}

static inline uint16_t rd_u16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* get_scale_min_k4, quantization.cpp:145-155 */
static inline void q4k_scale_min(int j, const uint8_t* s, uint8_t* sc, uint8_t* mn) {
  if (j < 4) {
    *sc = s[j] & 63;
    *mn = s[j + 4] & 63;
  } else {
    *sc = (uint8_t)((s[j + 4] & 0x0F) | ((s[j - 4] >> 6) << 4));
    *mn = (uint8_t)((s[j + 4] >> 4) | ((s[j] >> 6) << 4));
  }
}

/* dequantize_q4_k_m, quantization.cpp:171-221 */
static void deq_q4k_block(const uint8_t* b, float* y) {
  const float d = tlo_fp16_to_fp32(rd_u16(b + 0), 0);
  const float dmin = tlo_fp16_to_fp32(rd_u16(b + 2), 0);
  const uint8_t* scales = b + 4;
  const uint8_t* q = b + 16;
  for (int c = 0; c < 4; ++c) { /* 64 weights per pass, sub-blocks 2c and 2c+1 */
    uint8_t sc1, m1, sc2, m2;
    q4k_scale_min(2 * c, scales, &sc1, &m1);
    q4k_scale_min(2 * c + 1, scales, &sc2, &m2);
    const float d1 = d * (float)sc1, mm1 = dmin * (float)m1;
    const float d2 = d * (float)sc2, mm2 = dmin * (float)m2;
    for (int l = 0; l < 32; ++l) y[64 * c + l] = d1 * (float)(q[32 * c + l] & 0x0F) - mm1;
    for (int l = 0; l < 32; ++l) y[64 * c + 32 + l] = d2 * (float)(q[32 * c + l] >> 4) - mm2;
  }
}

/* dequantize_q6_k, quantization.cpp:223-266 */
static void deq_q6k_block(const uint8_t* b, float* y) {
  const float d = tlo_fp16_to_fp32(rd_u16(b + 208), 0);
  for (int half = 0; half < 2; ++half) {
    const uint8_t* ql = b + 64 * half;
    const uint8_t* qh = b + 128 + 32 * half;
    const int8_t* sc = (const int8_t*)(b + 192 + 8 * half);
    float* yo = y + 128 * half;
    for (int l = 0; l < 32; ++l) {
      const int is = l / 16;
      const int8_t q1 = (int8_t)((int8_t)((ql[l] & 0x0F) | (((qh[l] >> 0) & 3) << 4)) - 32);
      const int8_t q2 = (int8_t)((int8_t)((ql[l + 32] & 0x0F) | (((qh[l] >> 2) & 3) << 4)) - 32);
      const int8_t q3 = (int8_t)((int8_t)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32);
      const int8_t q4 = (int8_t)((int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
      /* d * sc * q evaluates left to right: (d*float(sc))*float(q) */
      yo[l + 0] = d * (float)sc[is + 0] * (float)q1;
      yo[l + 32] = d * (float)sc[is + 2] * (float)q2;
      yo[l + 64] = d * (float)sc[is + 4] * (float)q3;
      yo[l + 96] = d * (float)sc[is + 6] * (float)q4;
    }
  }
}

/* dequantize_q8_0_block, quantization.cpp:1047-1052 */
static void deq_q80_block(const uint8_t* b, float* y) {
  const float d = tlo_fp16_to_fp32(rd_u16(b), 1);
  const int8_t* q = (const int8_t*)(b + 2);
  for (int i = 0; i < TLO_QK8_0; ++i) y[i] = d * (float)q[i];
}

/* dequantize_q8_k, quantization.cpp:1009-1045 */
static void deq_q8k_block(const uint8_t* b, float* y) {
  const float d = tlo_fp16_to_fp32(rd_u16(b), 1);
  const int8_t* q = (const int8_t*)(b + 2);
  for (int i = 0; i < TLO_QK_K; ++i) y[i] = d * (float)q[i];
}

static void deq_block(int type, const uint8_t* b, float* y) {
  switch (type) {
    case TLO_Q4_K: deq_q4k_block(b, y); break;
    case TLO_Q6_K: deq_q6k_block(b, y); break;
    case TLO_Q8_0: deq_q80_block(b, y); break;
    case TLO_Q8_K: deq_q8k_block(b, y); break;
    case TLO_F32: memcpy(y, b, 4); break;
    default: break;
  }
}

/* dequantize_vector_*_to_f32, quantization.cpp:1054-1223: serial over blocks */
void tlo_dequantize(int type, const void* blocks, size_t nblocks, float* out) {
  const size_t bb = tlo_type_block_bytes(type), be = tlo_type_block_elems(type);
  const uint8_t* p = (const uint8_t*)blocks;
  for (size_t i = 0; i < nblocks; ++i) deq_block(type, p + i * bb, out + i * be);
}

/* quantize_fp32_to_q8_K, quantization.cpp:719-770 */
void tlo_quantize_q8_k(const float* x, size_t n, void* blocks_out) {
  uint8_t* out = (uint8_t*)blocks_out;
  const size_t nb = n / TLO_QK_K;
  for (size_t i = 0; i < nb; ++i, x += TLO_QK_K, out += 290) {
    float amax = 0.0f;
    for (int j = 0; j < TLO_QK_K; ++j) {
      const float a = fabsf(x[j]);
      amax = amax < a ? a : amax; /* std::max(amax, a) */
    }
    const float d = amax / 127.0f;
    const float id = (d != 0.f) ? 1.0f / d : 0.0f;
    const uint16_t dh = tlo_fp32_to_fp16(d);
    out[0] = (uint8_t)(dh & 0xff);
    out[1] = (uint8_t)(dh >> 8);
    int16_t bsum[16] = {0};
    for (int j = 0; j < TLO_QK_K; ++j) {
      float r = roundf(x[j] * id);
      r = 127.0f < r ? 127.0f : r;   /* std::min(127, r) */
      r = -128.0f < r ? r : -128.0f; /* std::max(-128, .) */
      const int8_t q = (int8_t)r;
      out[2 + j] = (uint8_t)q;
      bsum[j / 16] = (int16_t)(bsum[j / 16] + q);
    }
    memcpy(out + 258, bsum, sizeof bsum);
  }
}

/* ------------------------------------------------------------------------------------------
 * mat-vec with double Kahan accumulation (utils.cpp:293-397, 784-888)
 * ---------------------------------------------------------------------------------------- */
#define KAHAN_STEP(sum, comp, term)      \
  do {                                   \
    const double y_ = (term) - (comp);   \
    const double t_ = (sum) + y_;        \
    (comp) = (t_ - (sum)) - y_;          \
    (sum) = t_;                          \
  } while (0)

void tlo_matvec_f32(const float* w, const float* x, float* y, int rows, int cols) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    const float* wr = w + (size_t)r * cols;
    double s = 0.0, c = 0.0;
    for (int i = 0; i < cols; ++i) KAHAN_STEP(s, c, (double)wr[i] * (double)x[i]);
    y[r] = (float)s;
  }
}

void tlo_matvec_q(int type, const void* blocks, const float* x, float* y, int rows, int cols) {
  if (type == TLO_F32) {
    tlo_matvec_f32((const float*)blocks, x, y, rows, cols);
    return;
  }
  const size_t bb = tlo_type_block_bytes(type), be = tlo_type_block_elems(type);
  const size_t bpr = (size_t)cols / be;
  const uint8_t* base = (const uint8_t*)blocks;
#pragma omp parallel for
  for (long r = 0; r < rows; ++r) {
    float tmp[TLO_QK_K];
    double s = 0.0, c = 0.0;
    const uint8_t* row = base + (size_t)r * bpr * bb;
    for (size_t b = 0; b < bpr; ++b) {
      deq_block(type, row + b * bb, tmp);
      const float* xb = x + b * be;
      for (size_t i = 0; i < be; ++i) KAHAN_STEP(s, c, (double)tmp[i] * (double)xb[i]);
    }
    y[r] = (float)s;
  }
}

/* matmul_*_batch_cpu: a loop of per-token mat-vecs, OpenMP over tokens (utils.cpp:730-782, 890-926) */
void tlo_matmul_q_batch(int type, const void* blocks, const float* x, float* y, int T, int rows,
                        int cols) {
  const size_t bb = tlo_type_block_bytes(type), be = tlo_type_block_elems(type);
  const size_t bpr = (size_t)cols / be;
  const uint8_t* base = (const uint8_t*)blocks;
#pragma omp parallel for schedule(static)
  for (int t = 0; t < T; ++t) {
    const float* xt = x + (size_t)t * cols;
    float* yt = y + (size_t)t * rows;
    float tmp[TLO_QK_K];
    for (int r = 0; r < rows; ++r) {
      double s = 0.0, c = 0.0;
      if (type == TLO_F32) {
        const float* wr = (const float*)blocks + (size_t)r * cols;
        for (int i = 0; i < cols; ++i) KAHAN_STEP(s, c, (double)wr[i] * (double)xt[i]);
      } else {
        const uint8_t* row = base + (size_t)r * bpr * bb;
        for (size_t b = 0; b < bpr; ++b) {
          deq_block(type, row + b * bb, tmp);
          const float* xb = xt + b * be;
          for (size_t i = 0; i < be; ++i) KAHAN_STEP(s, c, (double)tmp[i] * (double)xb[i]);
        }
      }
      yt[r] = (float)s;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * glue ops
 * ---------------------------------------------------------------------------------------- */
/* rmsnorm_vector_cpu, utils.cpp:648-673 (eps floor numeric::MIN_NORM_EPS = 1e-5) */
void tlo_rmsnorm_vector(const float* x, const float* w, float* out, int n, float eps) {
  double ssq = 0.0;
  for (int i = 0; i < n; ++i) ssq += (double)x[i] * (double)x[i];
  ssq /= (double)(size_t)n;
  const float e = eps < 1e-5f ? 1e-5f : eps;
  const float nf = 1.0f / sqrtf((float)ssq + e);
  for (int i = 0; i < n; ++i) out[i] = x[i] * nf * w[i];
}

/* rmsnorm_batch_cpu, utils.cpp:613-646 (plain eps) */
void tlo_rmsnorm_batch(const float* x, const float* w, float* out, int T, int n, float eps) {
  for (int t = 0; t < T; ++t) {
    const float* xt = x + (size_t)t * n;
    float* ot = out + (size_t)t * n;
    double ssq = 0.0;
    for (int i = 0; i < n; ++i) ssq += (double)xt[i] * (double)xt[i];
    const double mean = ssq / n;
    const float nf = 1.0f / sqrtf((float)mean + eps);
    for (int i = 0; i < n; ++i) ot[i] = xt[i] * nf * w[i];
  }
}

/* gpu_initialization.cpp:48-58 */
void tlo_rope_table(float* table, int max_pos, int head_dim, float theta) {
  for (int pos = 0; pos < max_pos; ++pos) {
    for (int i = 0; i < head_dim; i += 2) {
      const float freq = powf(theta, -((float)i) / head_dim);
      const float angle = pos * freq;
      float* e = table + 2 * ((size_t)pos * (head_dim / 2) + i / 2);
      e[0] = cosf(angle);
      e[1] = sinf(angle);
    }
  }
}

/* apply_rope_vector, utils.cpp:428-489 */
void tlo_rope_apply(float* x, int num_heads, int head_dim, int pos, const float* table,
                    int max_pos, int adjacent) {
  if (pos < 0 || pos >= max_pos) return;
  const int half = head_dim / 2;
  const float* tp = table + 2 * (size_t)pos * half;
  for (int h = 0; h < num_heads; ++h) {
    float* xh = x + (size_t)h * head_dim;
    for (int i = 0; i < half; ++i) {
      const float c = tp[2 * i], s = tp[2 * i + 1];
      const int i0 = adjacent ? 2 * i : i;
      const int i1 = adjacent ? 2 * i + 1 : i + half;
      const float a = xh[i0], b = xh[i1];
      xh[i0] = a * c - b * s;
      xh[i1] = a * s + b * c;
    }
  }
}

/* softmax_vector_cpu, utils.cpp:675-698 */
void tlo_softmax(const float* x, float* out, int n) {
  if (n <= 0) return;
  float mx = x[0];
  for (int i = 1; i < n; ++i)
    if (x[i] > mx) mx = x[i];
  float sum = 0.0f;
  for (int i = 0; i < n; ++i) {
    out[i] = expf(x[i] - mx);
    sum += out[i];
  }
  const float inv = 1.0f / (sum + 1e-9f);
  for (int i = 0; i < n; ++i) out[i] *= inv;
}

/* silu_cpu, utils.cpp:700-707 */
void tlo_silu(const float* x, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    const float sg = 1.0f / (1.0f + expf(-x[i]));
    out[i] = x[i] * sg;
  }
}

void tlo_swiglu(const float* gate, const float* up, float* out, int n, int batch_form) {
  if (batch_form) { /* cpu_batch_processor.cpp:252-256 */
    for (int i = 0; i < n; ++i) {
      const float g = gate[i];
      const float s = g / (1.0f + expf(-g));
      out[i] = s * up[i];
    }
  } else { /* model.cpp:734-735 */
    for (int i = 0; i < n; ++i) {
      const float sg = 1.0f / (1.0f + expf(-gate[i]));
      const float a = gate[i] * sg;
      out[i] = a * up[i];
    }
  }
}

/* model.cpp:632-663 */
void tlo_attention_decode(const float* q, const float* kcache, const float* vcache, float* out,
                          int n_heads, int n_kv_heads, int head_dim, int pos) {
  const float scale = 1.0f / sqrtf((float)head_dim);
  const int kv_dim = n_kv_heads * head_dim;
  const int group = n_heads / n_kv_heads;
  const int len = pos + 1;
  float* scores = (float*)malloc(sizeof(float) * (size_t)len);
  for (int h = 0; h < n_heads; ++h) {
    const float* qh = q + (size_t)h * head_dim;
    const int kvh = h / group;
    for (int t = 0; t < len; ++t) {
      const float* kt = kcache + (size_t)t * kv_dim + (size_t)kvh * head_dim;
      float s = 0.0f;
      for (int d = 0; d < head_dim; ++d) s += qh[d] * kt[d];
      scores[t] = s * scale;
    }
    tlo_softmax(scores, scores, len);
    float* oh = out + (size_t)h * head_dim;
    for (int d = 0; d < head_dim; ++d) oh[d] = 0.0f;
    for (int t = 0; t < len; ++t) {
      const float* vt = vcache + (size_t)t * kv_dim + (size_t)kvh * head_dim;
      for (int d = 0; d < head_dim; ++d) oh[d] += scores[t] * vt[d];
    }
  }
  free(scores);
}

/* attention_batch_cpu, cpu_attention.cpp:91-201 */
void tlo_attention_batch(const float* q, const float* kcache, const float* vcache, float* out,
                         int T, int start_pos, int n_heads, int n_kv_heads, int head_dim) {
  const float scale = 1.0f / sqrtf((float)head_dim);
  const int kv_dim = n_kv_heads * head_dim;
  const int group = n_heads / n_kv_heads;
  float* scores = (float*)malloc(sizeof(float) * (size_t)(start_pos + T + 1));
  memset(out, 0, sizeof(float) * (size_t)T * n_heads * head_dim);
  for (int t = 0; t < T; ++t) {
    const int len = start_pos + t + 1;
    if (len <= 0) continue;
    for (int h = 0; h < n_heads; ++h) {
      const float* qh = q + ((size_t)t * n_heads + h) * head_dim;
      const int kvh = h / group;
      for (int p = 0; p < len; ++p) {
        const float* kt = kcache + (size_t)p * kv_dim + (size_t)kvh * head_dim;
        float s = 0.0f;
        for (int d = 0; d < head_dim; ++d) s += qh[d] * kt[d];
        scores[p] = s * scale;
      }
      tlo_softmax(scores, scores, len);
      float* oh = out + ((size_t)t * n_heads + h) * head_dim;
      for (int p = 0; p < len; ++p) {
        if (scores[p] == 0.0f) continue; /* cpu_attention.cpp:182 */
        const float* vt = vcache + (size_t)p * kv_dim + (size_t)kvh * head_dim;
        for (int d = 0; d < head_dim; ++d) oh[d] += scores[p] * vt[d];
      }
    }
  }
  free(scores);
}

/* api.cpp:73-80 / 134-136: first maximal element */
int tlo_argmax(const float* v, int n) {
  if (n <= 0) return -1;
  int best = 0;
  for (int i = 1; i < n; ++i)
    if (v[best] < v[i]) best = i;
  return best;
}

/* api.cpp:86-97 */
void tlo_apply_repetition_penalty(float* logits, int vocab, const int* generated, int n_generated,
                                  float penalty, int window) {
  if (penalty <= 1.0f || n_generated <= 0) return;
  const int start = n_generated > window ? n_generated - window : 0;
  for (int i = start; i < n_generated; ++i) {
    const int id = generated[i];
    if (id < 0 || id >= vocab) continue;
    float* l = &logits[id];
    *l = (*l > 0.0f) ? *l / penalty : *l * penalty;
  }
}

/* api.cpp:126-216 sample_top_k_top_p_temperature, up to (not including) the random draw: the candidate set in draw
 * order and its renormalised probabilities. Returns the candidate count; temperature < 0.05 is the greedy branch
 * (api.cpp:134-136): one candidate, the first maximum, probability 1. ids / probs need room for `vocab` entries.
 * Order of operations as in the reference: (l - max) * (1/T) in float; exp and the sum in double; /sum in double, then
 * rounded to float; std::sort with std::greater on (prob, id) pairs, i.e. ties broken towards the HIGHER id; top_k cut;
 * float running sum until >= top_p; float renormalisation. */
typedef struct {
  float p;
  int id;
} tlo_prob_id;
static int tlo_prob_id_greater(const void* a, const void* b) {
  const tlo_prob_id *x = (const tlo_prob_id*)a, *y = (const tlo_prob_id*)b;
  if (x->p != y->p) return x->p > y->p ? -1 : 1;
  if (x->id != y->id) return x->id > y->id ? -1 : 1;
  return 0;
}
int tlo_sample_candidates(const float* logits, int vocab, float temperature, int top_k, float top_p, int* ids,
                          float* probs) {
  if (vocab <= 0) return 0;
  if (temperature < 0.05f) {
    ids[0] = tlo_argmax(logits, vocab);
    probs[0] = 1.0f;
    return 1;
  }
  if (top_k > vocab) top_k = vocab;
  if (top_k <= 0) top_k = vocab;
  float max_logit = -INFINITY;
  for (int i = 0; i < vocab; ++i)
    if (logits[i] > max_logit) max_logit = logits[i]; /* std::max(max_logit, logit) */
  const float scale = 1.0f / temperature;
  double* pd = (double*)malloc(sizeof(double) * (size_t)vocab);
  tlo_prob_id* pi = (tlo_prob_id*)malloc(sizeof(tlo_prob_id) * (size_t)vocab);
  double sum_exp = 0.0;
  for (int i = 0; i < vocab; ++i) {
    const float scaled = (logits[i] - max_logit) * scale;
    pd[i] = exp((double)scaled);
    sum_exp += pd[i];
  }
  for (int i = 0; i < vocab; ++i) pd[i] = sum_exp > 0.0 ? pd[i] / sum_exp : 1.0 / vocab;
  for (int i = 0; i < vocab; ++i) {
    pi[i].p = (float)pd[i];
    pi[i].id = i;
  }
  qsort(pi, (size_t)vocab, sizeof(tlo_prob_id), tlo_prob_id_greater); /* a total order: any sort gives std::sort's result */
  int n = top_k < vocab ? top_k : vocab;
  float cumulative = 0.0f;
  int last = 0;
  for (int i = 0; i < n; ++i) {
    cumulative += pi[i].p;
    last = i;
    if (cumulative >= top_p) break;
  }
  n = last + 1;
  float final_sum = 0.0f;
  for (int i = 0; i < n; ++i) final_sum += pi[i].p;
  for (int i = 0; i < n; ++i) {
    ids[i] = pi[i].id;
    probs[i] = final_sum > 0.0f ? pi[i].p / final_sum : 1.0f / (float)n;
  }
  free(pd);
  free(pi);
  return n;
}

/* The draw of std::discrete_distribution<int> (api.cpp:211-213) as libstdc++ 11 implements it (bits/random.tcc:2656-2713):
 * weights summed and normalised in double, partial sums, the last forced to 1, then the FIRST index whose cumulative
 * probability is >= u (std::lower_bound) for u = generate_canonical<double> in [0, 1). A single weight always draws 0.
 * (The reference seeds its std::mt19937 from std::random_device, so u itself is not reproducible.) */
int tlo_discrete_draw(const float* probs, int n, double u) {
  if (n < 2) return 0;
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += (double)probs[i];
  double cp = 0.0;
  for (int i = 0; i < n; ++i) {
    cp += (double)probs[i] / sum;
    const double c = i == n - 1 ? 1.0 : cp;
    if (c >= u) return i;
  }
  return n - 1;
}

/* ------------------------------------------------------------------------------------------
 * model-level
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int type;
  const void* data;
  int rows, cols;
  float* f32; /* ref_structure: lazily dequantized copy */
} tlo_tensor;

struct tlo_model {
  int hidden, inter, n_heads, n_kv_heads, n_layers, vocab, max_pos, head_dim;
  float eps, theta;
  int ref_structure;
  tlo_tensor* lt; /* [n_layers][9] */
  tlo_tensor tok, onorm, out;
  float* rope; /* [max_pos][hd/2][2] */
  float** kc;
  float** vc;
};

tlo_model* tlo_model_create(int hidden, int inter, int n_heads, int n_kv_heads, int n_layers,
                            int vocab, int max_pos, float eps, float rope_theta) {
  tlo_model* m = (tlo_model*)calloc(1, sizeof *m);
  m->hidden = hidden;
  m->inter = inter;
  m->n_heads = n_heads;
  m->n_kv_heads = n_kv_heads;
  m->n_layers = n_layers;
  m->vocab = vocab;
  m->max_pos = max_pos;
  m->head_dim = hidden / n_heads;
  m->eps = eps;
  m->theta = rope_theta;
  m->lt = (tlo_tensor*)calloc((size_t)n_layers * 9, sizeof(tlo_tensor));
  m->rope = (float*)malloc(sizeof(float) * (size_t)max_pos * m->head_dim);
  tlo_rope_table(m->rope, max_pos, m->head_dim, rope_theta);
  m->kc = (float**)calloc((size_t)n_layers, sizeof(float*));
  m->vc = (float**)calloc((size_t)n_layers, sizeof(float*));
  const size_t kvn = (size_t)max_pos * n_kv_heads * m->head_dim;
  for (int l = 0; l < n_layers; ++l) {
    m->kc[l] = (float*)calloc(kvn, sizeof(float));
    m->vc[l] = (float*)calloc(kvn, sizeof(float));
  }
  return m;
}

static void tensor_evict(tlo_tensor* t) {
  free(t->f32);
  t->f32 = NULL;
}

void tlo_model_free(tlo_model* m) {
  if (!m) return;
  for (int i = 0; i < m->n_layers * 9; ++i) tensor_evict(&m->lt[i]);
  tensor_evict(&m->out);
  for (int l = 0; l < m->n_layers; ++l) {
    free(m->kc[l]);
    free(m->vc[l]);
  }
  free(m->kc);
  free(m->vc);
  free(m->rope);
  free(m->lt);
  free(m);
}

int tlo_model_set_tensor(tlo_model* m, int kind, int layer, int type, const void* data, int rows,
                         int cols) {
  tlo_tensor* t;
  if (kind == TLO_T_TOKEN_EMBD) t = &m->tok;
  else if (kind == TLO_T_OUTPUT_NORM) t = &m->onorm;
  else if (kind == TLO_T_OUTPUT) t = &m->out;
  else if (kind >= 0 && kind <= TLO_T_FFN_NORM && layer >= 0 && layer < m->n_layers)
    t = &m->lt[layer * 9 + kind];
  else return -1;
  t->type = type;
  t->data = data;
  t->rows = rows;
  t->cols = cols;
  t->f32 = NULL;
  return 0;
}

void tlo_model_set_ref_structure(tlo_model* m, int ref_structure) { m->ref_structure = ref_structure; }

void tlo_model_reset_kv(tlo_model* m) {
  const size_t kvn = (size_t)m->max_pos * m->n_kv_heads * m->head_dim;
  for (int l = 0; l < m->n_layers; ++l) {
    memset(m->kc[l], 0, kvn * sizeof(float));
    memset(m->vc[l], 0, kvn * sizeof(float));
  }
}

const float* tlo_model_kcache(tlo_model* m, int layer) { return m->kc[layer]; }
const float* tlo_model_vcache(tlo_model* m, int layer) { return m->vc[layer]; }

/* lookup_embedding, model_utils.cpp:11-182: dequantize one row */
void tlo_lookup_embedding(tlo_model* m, int token, float* out) {
  const int hs = m->hidden;
  if (token < 0 || token >= m->vocab) {
    memset(out, 0, sizeof(float) * (size_t)hs);
    return;
  }
  const tlo_tensor* t = &m->tok;
  if (t->type == TLO_F32) {
    memcpy(out, (const float*)t->data + (size_t)token * hs, sizeof(float) * (size_t)hs);
    return;
  }
  const size_t bb = tlo_type_block_bytes(t->type), be = tlo_type_block_elems(t->type);
  const size_t bpr = (size_t)hs / be;
  tlo_dequantize(t->type, (const uint8_t*)t->data + (size_t)token * bpr * bb, bpr, out);
}

/* dequantize_vector_q{4k,6k,8_0}_to_f32 (quantization.cpp:1054-1223) in the reference's structure, for the CPU baseline:
 * f32_weights.resize(n) value-initialises the fresh vector (a pass of zeros over the whole FP32 tensor) before the serial
 * block loop fills it. Same values as tlo_dequantize; the extra pass is part of what the reference pays per tensor and
 * token (tools/validate_cpu_baseline.py times this against the reference's own function). */
void tlo_dequantize_whole_tensor(int type, const void* blocks, size_t nblocks, float* out) {
  memset(out, 0, nblocks * tlo_type_block_elems(type) * sizeof(float));
  tlo_dequantize(type, blocks, nblocks, out);
}

/* one projection y = W x through either the fused path or (ref_structure) ensure_*_dequantized +
 * matvec_f32_f32_vector_cpu — the two are bit-identical */
static void project(tlo_model* m, tlo_tensor* t, const float* x, float* y) {
  if (m->ref_structure && t->type != TLO_F32) {
    if (!t->f32) {
      const size_t n = (size_t)t->rows * t->cols;
      t->f32 = (float*)malloc(n * sizeof(float));
      tlo_dequantize_whole_tensor(t->type, t->data, n / tlo_type_block_elems(t->type), t->f32);
    }
    tlo_matvec_f32(t->f32, x, y, t->rows, t->cols);
  } else {
    tlo_matvec_q(t->type, t->data, x, y, t->rows, t->cols);
  }
}

static void project_batch(tlo_model* m, tlo_tensor* t, const float* x, float* y, int T) {
  if (m->ref_structure && t->type != TLO_F32) {
    if (!t->f32) {
      const size_t n = (size_t)t->rows * t->cols;
      t->f32 = (float*)malloc(n * sizeof(float));
      tlo_dequantize_whole_tensor(t->type, t->data, n / tlo_type_block_elems(t->type), t->f32);
    }
    tlo_matmul_q_batch(TLO_F32, t->f32, x, y, T, t->rows, t->cols);
  } else {
    tlo_matmul_q_batch(t->type, t->data, x, y, T, t->rows, t->cols);
  }
}

/* TinyLlamaModel::forward, model.cpp:536-807 */
void tlo_forward(tlo_model* m, float* x, int pos, float* logits, int n_layers_limit) {
  const int hs = m->hidden, is = m->inter, nh = m->n_heads, nkv = m->n_kv_heads, hd = m->head_dim;
  const int kvd = nkv * hd;
  const int nl = (n_layers_limit < 0 || n_layers_limit > m->n_layers) ? m->n_layers : n_layers_limit;
  float* xn = (float*)malloc(sizeof(float) * (size_t)hs);
  float* q = (float*)malloc(sizeof(float) * (size_t)hs);
  float* k = (float*)malloc(sizeof(float) * (size_t)kvd);
  float* v = (float*)malloc(sizeof(float) * (size_t)kvd);
  float* att = (float*)malloc(sizeof(float) * (size_t)hs);
  float* prj = (float*)malloc(sizeof(float) * (size_t)hs);
  float* g = (float*)malloc(sizeof(float) * (size_t)is);
  float* u = (float*)malloc(sizeof(float) * (size_t)is);
  float* h = (float*)malloc(sizeof(float) * (size_t)is);
  for (int l = 0; l < nl; ++l) {
    tlo_tensor* lt = &m->lt[l * 9];
    tlo_rmsnorm_vector(x, (const float*)lt[TLO_T_ATTN_NORM].data, xn, hs, m->eps);
    project(m, &lt[TLO_T_ATTN_Q], xn, q);
    project(m, &lt[TLO_T_ATTN_K], xn, k);
    project(m, &lt[TLO_T_ATTN_V], xn, v);
    tlo_rope_apply(q, nh, hd, pos, m->rope, m->max_pos, 1);
    tlo_rope_apply(k, nkv, hd, pos, m->rope, m->max_pos, 1);
    if (pos < m->max_pos) { /* model.cpp:615-625 */
      memcpy(m->kc[l] + (size_t)pos * kvd, k, sizeof(float) * (size_t)kvd);
      memcpy(m->vc[l] + (size_t)pos * kvd, v, sizeof(float) * (size_t)kvd);
    }
    tlo_attention_decode(q, m->kc[l], m->vc[l], att, nh, nkv, hd, pos);
    project(m, &lt[TLO_T_ATTN_O], att, prj);
    for (int i = 0; i < hs; ++i) x[i] = x[i] + prj[i];
    tlo_rmsnorm_vector(x, (const float*)lt[TLO_T_FFN_NORM].data, xn, hs, m->eps);
    project(m, &lt[TLO_T_FFN_GATE], xn, g);
    project(m, &lt[TLO_T_FFN_UP], xn, u);
    tlo_swiglu(g, u, h, is, 0);
    project(m, &lt[TLO_T_FFN_DOWN], h, prj);
    for (int i = 0; i < hs; ++i) x[i] = x[i] + prj[i];
    if (m->ref_structure && l >= 2) { /* model.cpp:771-777 */
      for (int kd = 0; kd <= TLO_T_FFN_DOWN; ++kd) tensor_evict(&m->lt[(l - 2) * 9 + kd]);
    }
  }
  if (logits && nl == m->n_layers) {
    tlo_rmsnorm_vector(x, (const float*)m->onorm.data, xn, hs, m->eps);
    project(m, &m->out, xn, logits);
  }
  free(xn); free(q); free(k); free(v); free(att); free(prj); free(g); free(u); free(h);
}

/* forward_cpu_batch (cpu_batch_processor.cpp:11-281) + forward_cpu_logits_batch (model.cpp:1094-1156) */
void tlo_forward_batch(tlo_model* m, float* x, int T, int start_pos, float* logits) {
  const int hs = m->hidden, is = m->inter, nh = m->n_heads, nkv = m->n_kv_heads, hd = m->head_dim;
  const int kvd = nkv * hd;
  const size_t th = (size_t)T * hs, ti = (size_t)T * is, tk = (size_t)T * kvd;
  float* xn = (float*)malloc(sizeof(float) * th);
  float* q = (float*)malloc(sizeof(float) * th);
  float* k = (float*)malloc(sizeof(float) * tk);
  float* v = (float*)malloc(sizeof(float) * tk);
  float* att = (float*)malloc(sizeof(float) * th);
  float* prj = (float*)malloc(sizeof(float) * th);
  float* g = (float*)malloc(sizeof(float) * ti);
  float* u = (float*)malloc(sizeof(float) * ti);
  float* h = (float*)malloc(sizeof(float) * ti);
  for (int l = 0; l < m->n_layers; ++l) {
    tlo_tensor* lt = &m->lt[l * 9];
    tlo_rmsnorm_batch(x, (const float*)lt[TLO_T_ATTN_NORM].data, xn, T, hs, m->eps);
    project_batch(m, &lt[TLO_T_ATTN_Q], xn, q, T);
    project_batch(m, &lt[TLO_T_ATTN_K], xn, k, T);
    project_batch(m, &lt[TLO_T_ATTN_V], xn, v, T);
    for (int t = 0; t < T; ++t) { /* apply_rope_batch_cpu, utils.cpp:491-611 */
      tlo_rope_apply(q + (size_t)t * hs, nh, hd, start_pos + t, m->rope, m->max_pos, 1);
      tlo_rope_apply(k + (size_t)t * kvd, nkv, hd, start_pos + t, m->rope, m->max_pos, 1);
    }
    for (int t = 0; t < T; ++t) { /* update_kv_cache_batch_cpu, cpu_attention.cpp:7-89 */
      const int p = start_pos + t;
      if (p >= m->max_pos) continue;
      memcpy(m->kc[l] + (size_t)p * kvd, k + (size_t)t * kvd, sizeof(float) * (size_t)kvd);
      memcpy(m->vc[l] + (size_t)p * kvd, v + (size_t)t * kvd, sizeof(float) * (size_t)kvd);
    }
    tlo_attention_batch(q, m->kc[l], m->vc[l], att, T, start_pos, nh, nkv, hd);
    project_batch(m, &lt[TLO_T_ATTN_O], att, prj, T);
    for (size_t i = 0; i < th; ++i) x[i] = x[i] + prj[i];
    tlo_rmsnorm_batch(x, (const float*)lt[TLO_T_FFN_NORM].data, xn, T, hs, m->eps);
    project_batch(m, &lt[TLO_T_FFN_GATE], xn, g, T);
    project_batch(m, &lt[TLO_T_FFN_UP], xn, u, T);
    tlo_swiglu(g, u, h, (int)ti, 1);
    project_batch(m, &lt[TLO_T_FFN_DOWN], h, prj, T);
    for (size_t i = 0; i < th; ++i) x[i] = x[i] + prj[i];
  }
  if (logits) {
    tlo_rmsnorm_batch(x, (const float*)m->onorm.data, xn, T, hs, m->eps);
    project_batch(m, &m->out, xn, logits, T);
  }
  free(xn); free(q); free(k); free(v); free(att); free(prj); free(g); free(u); free(h);
}

/* TinyLlamaSession::generate on token ids, greedy (api.cpp:585-813) */
int tlo_generate(tlo_model* m, const int* prompt, int n_prompt, int max_new, float penalty,
                 int eos_id, int* out_ids, float* logits_dump) {
  const int P = n_prompt, V = m->vocab, hs = m->hidden;
  const int total_steps = P + max_new - 1;
  int n_gen = 0, next = -1, start = 0;
  float* logits = (float*)malloc(sizeof(float) * (size_t)V);
  float* x = (float*)malloc(sizeof(float) * (size_t)hs);
  tlo_model_reset_kv(m);
  if (P >= 32) { /* batched prefill, api.cpp:601-694 */
    float* xb = (float*)malloc(sizeof(float) * (size_t)P * hs);
    float* lb = (float*)malloc(sizeof(float) * (size_t)P * V);
    for (int i = 0; i < P; ++i) tlo_lookup_embedding(m, prompt[i], xb + (size_t)i * hs);
    tlo_forward_batch(m, xb, P, 0, lb);
    memcpy(logits, lb + (size_t)(P - 1) * V, sizeof(float) * (size_t)V);
    free(xb);
    free(lb);
    if (logits_dump) memcpy(logits_dump + (size_t)n_gen * V, logits, sizeof(float) * (size_t)V);
    tlo_apply_repetition_penalty(logits, V, out_ids, n_gen, penalty, 64);
    next = tlo_argmax(logits, V);
    out_ids[n_gen++] = next;
    start = P;
  }
  for (int pos = start; pos < total_steps; ++pos) {
    if (pos >= m->max_pos) break;
    const int input = (pos < P && start == 0) ? prompt[pos] : next;
    tlo_lookup_embedding(m, input, x);
    tlo_forward(m, x, pos, logits, -1);
    if (pos == P - 1 || pos >= P) {
      if (logits_dump) memcpy(logits_dump + (size_t)n_gen * V, logits, sizeof(float) * (size_t)V);
      tlo_apply_repetition_penalty(logits, V, out_ids, n_gen, penalty, 64);
      next = tlo_argmax(logits, V);
      out_ids[n_gen++] = next;
    }
    if (next == eos_id && pos >= P) break;
    if (n_gen >= max_new) break;
  }
  free(logits);
  free(x);
  return n_gen;
}
