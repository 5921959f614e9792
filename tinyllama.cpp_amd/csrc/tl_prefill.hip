// tl_prefill.hip — causal prefill attention of T new tokens over the device-resident KV cache, on the FP32 matrix
// instruction (v_mfma_f32_32x32x2_f32: exact FP32 products, FP32 accumulate — the CPU path's arithmetic class).
//
// Replaces attention_batch_prefill_cuda (cuda_kernels.h:498; one block per (token, head) with the whole context's
// scores in shared memory). The math is attention_batch_cpu (cpu_attention.cpp:91-201): token i (absolute position
// start_pos + i) attends positions 0..start_pos+i of its kv head's cache rows; scores = (q . k) * scale, softmax
// with max-subtraction and 1/(sum + 1e-9), out = sum p v.
//
// Shape: one workgroup of PF_NW waves per (32-query tile, head); the key tiles of 32 positions are dealt round-robin to the
// waves, each walks its tiles with an online softmax of its own, and the waves' (max, sum, O) are merged through LDS in
// wave order at the end (deterministic).
//   S^T[pos][q] = K Q^T : A = K tile, B = Q^T. The summation index d is permuted so that lane half h owns
//     d in [HD/2*h, HD/2*(h+1)): a lane's K operand is then ONE contiguous half row of the cache (16-B loads straight
//     from global memory, no LDS), and its Q operand a half row kept in registers for the whole kernel.
//   The result has the query on the lane and 16 positions in registers: row max / sum are in-register reductions plus
//     one exchange between the lane halves.
//   O[q][d] = P V : the S^T accumulator registers ARE the A operand of the second product (the 32x32 result layout
//     puts the summed-over position index where the next MFMA's k index lives), V rows are read 128 B per half wave.
#include <math.h>

#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// position (row of the 32x32 result) held in accumulator register e of lane half h
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

constexpr int PF_NW = 4;

template <int HD>
__global__ __launch_bounds__(64 * PF_NW) void attn_prefill_mfma_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                               const float* __restrict__ vc, float* __restrict__ out,
                                                               int T, int start_pos, int nh, int nkv, float scale) {
  constexpr int HH = HD / 2;   // d range of a lane half
  constexpr int DB = HD / 32;  // 32-wide d blocks of the output
  __shared__ float sm_m[PF_NW][32], sm_l[PF_NW][32];
  __shared__ __attribute__((aligned(16))) float sm_o[32][HD];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * 32, head = blockIdx.y;
  const int kvh = head / (nh / nkv);
  const int kv_dim = nkv * HD;
  const int len = start_pos + T;  // positions present in the cache
  const int qi = q0 + r < T ? q0 + r : T - 1;  // this lane's query (clamped; rows >= T are not stored)
  const int qpos = start_pos + qi;

  float qreg[HH];
  {
    const float* qp = q + ((size_t)qi * nh + head) * HD + HH * h;
#pragma unroll
    for (int i = 0; i < HH / 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * i);
      qreg[4 * i] = v[0]; qreg[4 * i + 1] = v[1]; qreg[4 * i + 2] = v[2]; qreg[4 * i + 3] = v[3];
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int b = 0; b < DB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[b][e] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;  // per query (lane r); lrun is this half's share of the sum

  const int last_q = q0 + 31 < T ? q0 + 31 : T - 1;
  const int ntile = (start_pos + last_q) / 32 + 1;  // tiles holding positions <= the last query's position
  for (int tile = wave; tile < ntile; tile += PF_NW) {
    const int p0 = tile * 32;
    // ---- S^T = K Q^T over d ---------------------------------------------------------------------------------
    const int krow = p0 + r < len ? p0 + r : len - 1;
    const float* kp = kc + (size_t)krow * kv_dim + (size_t)kvh * HD + HH * h;
    float kreg[HH];
#pragma unroll
    for (int i = 0; i < HH / 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(kp + 4 * i);
      kreg[4 * i] = v[0]; kreg[4 * i + 1] = v[1]; kreg[4 * i + 2] = v[2]; kreg[4 * i + 3] = v[3];
    }
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int i = 0; i < HH; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[i], qreg[i], s, 0, 0, 0);
    // ---- mask, online softmax (query on the lane, 16 positions in registers per half) ------------------------
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = p0 + acc_row(e, h);
      s[e] = p <= qpos ? s[e] * scale : -INFINITY;
      tmax = fmaxf(tmax, s[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(mrun, tmax);  // may still be -inf: a wave's first tiles can lie wholly behind a query
    const float corr = mrun == -INFINITY ? 0.f : expf(mrun - mnew);
    mrun = mnew;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = s[e] == -INFINITY ? 0.f : expf(s[e] - mnew);  // (s finite implies mnew finite)
      psum += s[e];
    }
    lrun = lrun * corr + psum;
    // ---- rescale O (its rows are queries: fetch each row's factor from the lane that owns the query) -----------
    if (__any(corr != 1.f)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float c = __shfl(corr, acc_row(e, h));
#pragma unroll
        for (int b = 0; b < DB; ++b) o[b][e] *= c;
      }
    }
    // ---- O += P V : A = the S^T registers (k = position), B = V[position][d block] ------------------------------
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int vrow = p0 + acc_row(e, h);
      vrow = vrow < len ? vrow : len - 1;  // masked positions carry p = 0; keep their loads inside the cache
      const float* vp = vc + (size_t)vrow * kv_dim + (size_t)kvh * HD + r;
#pragma unroll
      for (int b = 0; b < DB; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[e], vp[32 * b], o[b], 0, 0, 0);
    }
  }
  // ---- merge the waves: M = max_w m_w, L = sum_w l_w e^(m_w - M), O = sum_w O_w e^(m_w - M), in wave order -----------
  const float ltot = lrun + __shfl_xor(lrun, 32);
  if (h == 0) {
    sm_m[wave][r] = mrun;
    sm_l[wave][r] = ltot;
  }
  __syncthreads();
  float fac[16];  // this wave's factor for the 16 queries whose O rows this lane holds
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = acc_row(e, h);
    float M = sm_m[0][row];
#pragma unroll
    for (int w = 1; w < PF_NW; ++w) M = fmaxf(M, sm_m[w][row]);
    const float mw = sm_m[wave][row];
    fac[e] = mw == -INFINITY ? 0.f : expf(mw - M);  // M is finite: position 0 is visible to every query (wave 0, tile 0)
  }
  for (int w = 0; w < PF_NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = acc_row(e, h);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
          const float v = o[b][e] * fac[e];
          sm_o[row][32 * b + r] = w == 0 ? v : sm_o[row][32 * b + r] + v;
        }
      }
    }
    __syncthreads();
  }
  // ---- out[q][d] = O / (L + 1e-9): 256 threads over 32 x HD values ---------------------------------------------
  for (int i = threadIdx.x; i < 32 * HD; i += 64 * PF_NW) {
    const int row = i / HD, d = i % HD;
    if (q0 + row >= T) continue;
    float M = sm_m[0][row];
#pragma unroll
    for (int w = 1; w < PF_NW; ++w) M = fmaxf(M, sm_m[w][row]);
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < PF_NW; ++w) L += sm_m[w][row] == -INFINITY ? 0.f : sm_l[w][row] * expf(sm_m[w][row] - M);
    out[((size_t)(q0 + row) * nh + head) * HD + d] = sm_o[row][d] * (1.0f / (L + 1e-9f));
  }
}

}  // namespace

bool tl_attn_prefill_mfma_supported(int nh, int nkv, int hd) {
  return (hd == 64 || hd == 128) && nkv > 0 && nh % nkv == 0;
}

// q [T][nh][hd] (rotated), caches [max_seq][nkv][hd] already holding positions 0..start_pos+T-1, out [T][nh][hd]
int tl_launch_attn_prefill_mfma(const float* q, const float* kc, const float* vc, float* out, int T, int start_pos,
                                int nh, int nkv, int hd, float scale, hipStream_t s) {
  TL_REQUIRE(tl_attn_prefill_mfma_supported(nh, nkv, hd), "tl_launch_attn_prefill_mfma: unsupported head shape");
  if (T == 0) return 0;
  const dim3 grid((T + 31) / 32, nh), block(64 * PF_NW);
  if (hd == 128) attn_prefill_mfma_kernel<128><<<grid, block, 0, s>>>(q, kc, vc, out, T, start_pos, nh, nkv, scale);
  else attn_prefill_mfma_kernel<64><<<grid, block, 0, s>>>(q, kc, vc, out, T, start_pos, nh, nkv, scale);
  TL_HIP(hipGetLastError());
  return 0;
}
