// tl_prefill.hip — causal prefill attention of T new tokens over the device-resident KV cache, on the FP32 matrix
// instruction (v_mfma_f32_32x32x2_f32: exact FP32 products, FP32 accumulate — the CPU path's arithmetic class).
//
// Replaces attention_batch_prefill_cuda (cuda_kernels.h:498; one block per (token, head) with the whole context's
// scores in shared memory). The math is attention_batch_cpu (cpu_attention.cpp:91-201): token i (absolute position
// start_pos + i) attends positions 0..start_pos+i of its kv head's cache rows; scores = (q . k) * scale, softmax
// with max-subtraction and 1/(sum + 1e-9), out = sum p v.
//
// Shape: one workgroup of PF_NW = 8 waves per (head, PAIR of 32-query tiles p and ntq - 1 - p): a query tile's work grows
// with its position (causal), a pair is the same work for every p, and every XCD gets whole heads (see the kernel). Per
// query tile the key tiles of 32 positions are dealt round-robin to the waves, each walks its tiles with an online softmax
// of its own, and the waves' (max, sum, O) are merged through LDS slots in wave order (deterministic).
//   S^T[pos][q] = K Q^T : A = K tile, B = Q^T. The summation index d is permuted so that lane half h owns
//     d = 8 i + 4 h + (0..3): a lane's K operand is its own cache row, 16 bytes per load straight from global memory (no
//     LDS; the two halves share 32-byte pieces), asked for one tile AHEAD into the registers the previous products have
//     just read, so the round trip runs under that tile's softmax and P V. The Q operand is read from the query tile in
//     LDS (with Q in registers too there is no room for the prefetch at two waves per SIMD).
//   The result has the query on the lane and 16 positions in registers: row max / sum are in-register reductions plus
//     one exchange between the lane halves.
//   O[q][d] = P V : the S^T accumulator registers ARE the A operand of the second product (the 32x32 result layout
//     puts the summed-over position index where the next MFMA's k index lives), V rows are read 128 B per half wave.
// Round 3 (profiles/r03_k_prefill_attention.txt): 88 -> 47 us per layer at 512 tokens (8B shape). What the first form lost:
// 296 registers (one wave per SIMD, nothing to overlap a wave's load round trips with), a (tile, head) grid that gave XCD 0
// 2.4 x the work of XCD 7, two rounds of unequal workgroups, K loads exposed in front of every tile, a serial five-barrier
// merge with four exps per output element.
#include <math.h>

#include "tl_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// position (row of the 32x32 result) held in accumulator register e of lane half h
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

#ifdef TL_PF_STAMP  // diagnostic build (tools/pf_stamps.py): 100 MHz wall-clock stamps per wave
__device__ unsigned long long pf_stamps[256 * 8 * 32];
#define PFS(k) do { if (lane == 0 && blockIdx.x < 256) pf_stamps[(blockIdx.x * 8 + wave) * 32 + (k)] = wall_clock64(); } while (0)
#define PF_FORCE(x) do { float dummy_; asm volatile("v_mov_b32 %0, %1" : "=v"(dummy_) : "v"(x)); } while (0)
#else
#define PFS(k) do {} while (0)
#define PF_FORCE(x) do {} while (0)
#endif

constexpr int PF_NW = 8;  // waves per workgroup (two per SIMD: 256 VGPRs each)

// LDS floats of a workgroup: (max, sum) per wave and query, 1 / (L + 1e-9) and the waves' own factors, the query tile
// (rows padded by 4 floats: a lane reads its own row), one O slot per wave
template <int HD>
constexpr size_t pf_lds_floats() { return (size_t)PF_NW * 32 * 3 + 32 + 32 * (size_t)(HD + 4) + (size_t)PF_NW * 32 * HD; }

// planes != NULL: the output goes straight into the activation planes of the attn_output GEMM (what split_kernel<SPLIT_PLAIN>
// would make of `out`, bit for bit: np bf16 planes [plane_stride] of the [T][nh * hd] matrix + the sums of its 32-column
// blocks [nh * hd / 32][tpad]); `out` is then not written
struct PfPlanes {
  __bf16* planes;
  size_t plane_stride;
  float* xsum;
  int np, tpad;
};
typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));

struct PfTile {  // one 32-query tile as this lane sees it
  int q0, qi, qpos, ntile, nact;
};

template <int HD>
__global__ __launch_bounds__(64 * PF_NW) void attn_prefill_mfma_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                               const float* __restrict__ vc, float* __restrict__ out,
                                                               int T, int start_pos, int nh, int nkv, float scale, PfPlanes pl) {
  constexpr int HH = HD / 2;   // d range of a lane half
  constexpr int DB = HD / 32;  // 32-wide d blocks of the output
  extern __shared__ __attribute__((aligned(16))) float pf_sm[];
  float* sm_m = pf_sm;                    // [PF_NW][32] running max of every wave
  float* sm_l = sm_m + PF_NW * 32;        // [PF_NW][32] sum
  float* sm_f = sm_l + PF_NW * 32;        // [PF_NW][32] the wave's factor exp(m_w - M) per query
  float* sm_inv = sm_f + PF_NW * 32;      // [32] 1 / (L + 1e-9)
  float* sm_q = sm_inv + 32;              // [32][HD + 4] the query tile
  float* sm_o = sm_q + 32 * (HD + 4);     // [PF_NW][32][HD]
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  // Workgroup -> (head, PAIR of query tiles p and ntq - 1 - p). A query tile's work grows with its position (causal), so
  // a pair is the same work for every p: one round of equal workgroups (512 tokens x 32 heads: 256 of them) instead of two
  // rounds of unequal ones. Workgroup L runs on XCD L mod 8: every XCD gets nh / 8 whole heads (with 4 heads per kv head:
  // one kv head's K/V per L2) -- a (tile, head) grid handed XCD 0 the heaviest tile of every head and XCD 7 the lightest
  // (2.4 x the work; the launch lasted as long as XCD 0).
  int pair, head;
  const int ntq = (T + 31) / 32;
  {
    const int L = blockIdx.x;
    if (nh % 8 == 0) {
      const int hp = nh / 8, xcd = L & 7, j = L >> 3;
      head = xcd * hp + j % hp;
      pair = j / hp;
    } else {
      head = L % nh;
      pair = L / nh;
    }
  }
  const int kvh = head / (nh / nkv);
  const int kv_dim = nkv * HD;
  const int len = start_pos + T;  // positions present in the cache
  auto tile_of = [&](int qt) {
    PfTile t;
    t.q0 = qt * 32;
    t.qi = t.q0 + r < T ? t.q0 + r : T - 1;  // this lane's query (clamped; rows >= T are not stored)
    t.qpos = start_pos + t.qi;
    const int last_q = t.q0 + 31 < T ? t.q0 + 31 : T - 1;
    t.ntile = (start_pos + last_q) / 32 + 1;  // key tiles holding positions <= the last query's position
    t.nact = t.ntile < PF_NW ? t.ntile : PF_NW;
    return t;
  };
  const bool two = pair < ntq - 1 - pair;  // (odd tile count: the middle tile is its own pair)
  const PfTile tA = tile_of(ntq - 1 - pair), tB = tile_of(pair);
  // waves without a key tile leave at once (a barrier counts the surviving waves only) and give their registers back: the
  // early query tiles of a prompt run on one to seven waves. The early tile never has more waves than the late one.
  if (wave >= tA.nact) return;
  PFS(0);

  // The query tile lives in LDS (the MFMA operand is read from there, 16 bytes at a time): with it in registers there is
  // no room for the next tile's K rows beside this tile's (256 VGPRs at two waves per SIMD). The waves of the tile load it
  // together, coalesced: quad c of row w at thread w * (HD / 4) + c.
  constexpr int NQ4 = 32 * (HD / 4);                                   // quads of a query tile
  constexpr int QPT = (NQ4 + 64 * PF_NW - 1) / (64 * PF_NW);           // quads per thread when all waves are there
  auto q_src = [&](const PfTile& t, int i) {
    const int row = i / (HD / 4), c4 = i % (HD / 4);
    const int qrow = t.q0 + row < T ? t.q0 + row : T - 1;
    return reinterpret_cast<const f32x4*>(q + ((size_t)qrow * nh + head) * HD + 4 * c4);
  };
  auto q_dst = [&](int i) { return reinterpret_cast<f32x4*>(sm_q + (i / (HD / 4)) * (HD + 4) + 4 * (i % (HD / 4))); };
  // issue: the first QPT quads of this thread into registers (their round trip runs under whatever follows)
  auto load_q = [&](const PfTile& t, int nthreads, f32x4* tmp) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const int i = threadIdx.x + j * nthreads;
      tmp[j] = *q_src(t, i < NQ4 ? i : NQ4 - 1);
    }
  };
  // ... into LDS; with fewer than PF_NW waves the rest of the tile is copied in place (short tiles only)
  auto store_q = [&](const PfTile& t, int nthreads, const f32x4* tmp) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
      const int i = threadIdx.x + j * nthreads;
      if (i < NQ4) *q_dst(i) = tmp[j];
    }
    for (int i = threadIdx.x + QPT * nthreads; i < NQ4; i += nthreads) *q_dst(i) = *q_src(t, i);
  };
  float kreg[HH];
  auto load_k = [&](int tile) __attribute__((always_inline)) {
    const int krow = tile * 32 + r < len ? tile * 32 + r : len - 1;
    const float* kp = kc + (size_t)krow * kv_dim + (size_t)kvh * HD + 4 * h;
#pragma unroll
    for (int i = 0; i < HH / 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(kp + 8 * i);
      kreg[4 * i] = v[0]; kreg[4 * i + 1] = v[1]; kreg[4 * i + 2] = v[2]; kreg[4 * i + 3] = v[3];
    }
  };
  {
    f32x4 tmp[QPT];
    load_q(tA, 64 * tA.nact, tmp);
    load_k(wave);
    store_q(tA, 64 * tA.nact, tmp);
  }
  __syncthreads();
  const float* qrow_s = sm_q + r * (HD + 4) + 4 * h;  // this lane's share of its query row (d = 8 i + 4 h + 0..3)

  for (int ph = 0; ph < 2; ++ph) {  // the late (heavy) tile of the pair, then the early one
    const PfTile t = ph == 0 ? tA : tB;
    if (ph == 1 && (!two || wave >= tB.nact)) return;
    const int nact = t.nact;
    f32x16 o[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[b][e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;  // per query (lane r); lrun is this half's share of the sum
#ifdef TL_PF_STAMP
    int tcount = 0;
#endif

    for (int tile = wave; tile < t.ntile; tile += PF_NW) {
      const int p0 = tile * 32;
      // ---- S^T = K Q^T over d (q and this tile's K rows were asked for a tile ago) ------------------------------------
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int i4 = 0; i4 < HH / 4; ++i4) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qrow_s + 8 * i4);
#pragma unroll
        for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[4 * i4 + c], qv[c], s, 0, 0, 0);
      }
      // ---- the next tile's K rows (or the first ones of the pair's second query tile) into the registers the products
      //      above have just read: their round trip runs under this tile's softmax and P V ----------------------------
      if (tile + PF_NW < t.ntile) {
        load_k(tile + PF_NW);
      } else if (ph == 0 && two && wave < tB.nact) {
        load_k(wave);
      }
      // ---- mask, online softmax (query on the lane, 16 positions in registers per half) ------------------------
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int p = p0 + acc_row(e, h);
        s[e] = p <= t.qpos ? s[e] * scale : -INFINITY;
        tmax = fmaxf(tmax, s[e]);
      }
#ifdef TL_PF_STAMP
      PF_FORCE(tmax);
      if (tcount == 0) PFS(3 + 12 * ph);
#endif
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
#ifdef TL_PF_STAMP
      PF_FORCE(lrun);
      if (tcount == 0) PFS(4 + 12 * ph);
#endif
      // ---- rescale O (its rows are queries: fetch each row's factor from the lane that owns the query) -----------
      if (__any(corr != 1.f)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float c = __shfl(corr, acc_row(e, h));
#pragma unroll
          for (int b = 0; b < DB; ++b) o[b][e] *= c;
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // (V loads hoisted above the softmax do not fit beside the prefetched K rows)
      // ---- O += P V : A = the S^T registers (k = position), B = V[position][d block] ------------------------------
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int vrow = p0 + acc_row(e, h);
        vrow = vrow < len ? vrow : len - 1;  // masked positions carry p = 0; keep their loads inside the cache
        const float* vp = vc + (size_t)vrow * kv_dim + (size_t)kvh * HD + r;
#pragma unroll
        for (int b = 0; b < DB; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[e], vp[32 * b], o[b], 0, 0, 0);
      }
#ifdef TL_PF_STAMP
      PF_FORCE(o[DB - 1][15]);
      if (tcount == 0) PFS(5 + 12 * ph);
      ++tcount;
#endif
    }
    PFS(6 + 12 * ph);
    // ---- merge the active waves: M = max_w m_w, L = sum_w l_w e^(m_w - M), O = sum_w O_w e^(m_w - M), in wave order ----
    const float ltot = lrun + __shfl_xor(lrun, 32);
    __syncthreads();  // (second tile of the pair: everybody is done reading the first tile's slots)
    PFS(7 + 12 * ph);
    // every wave is past its last K Q^T of this query tile: the pair's second query tile may replace it (loads issued
    // here, stored in front of the merge's last barrier)
    f32x4 qtmp[QPT];
    const bool next_q = ph == 0 && two;
    if (next_q) load_q(tB, 64 * nact, qtmp);
    if (h == 0) {
      sm_m[wave * 32 + r] = mrun;
      sm_l[wave * 32 + r] = ltot;
    }
    __syncthreads();
    {
      // lane r (both halves compute the same) takes query r's overall max once; its own wave's factor goes through the
      // wave's LDS strip to the lanes that hold that query's O rows (a wave's LDS operations execute in order)
      float M = sm_m[r];
      for (int w = 1; w < nact; ++w) M = fmaxf(M, sm_m[w * 32 + r]);  // finite: position 0 is visible to every query (wave 0)
      const float mw = sm_m[wave * 32 + r];
      sm_f[wave * 32 + r] = mw == -INFINITY ? 0.f : expf(mw - M);
      if (wave == 0 && h == 0) {
        float L = 0.f;
        for (int w = 0; w < nact; ++w) {
          const float m2 = sm_m[w * 32 + r];
          L += m2 == -INFINITY ? 0.f : sm_l[w * 32 + r] * expf(m2 - M);
        }
        sm_inv[r] = 1.0f / (L + 1e-9f);
      }
    }
    __builtin_amdgcn_wave_barrier();
    PFS(8 + 12 * ph);
    float* slot = sm_o + (size_t)wave * 32 * HD;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = acc_row(e, h);
      const float f = sm_f[wave * 32 + row];
#pragma unroll
      for (int b = 0; b < DB; ++b) slot[row * HD + 32 * b + r] = o[b][e] * f;
    }
    if (next_q) store_q(tB, 64 * nact, qtmp);
    __syncthreads();
    PFS(9 + 12 * ph);
    // ---- out[q][d] = O / (L + 1e-9): the surviving threads over 32 x HD / 8 column octets, slots summed in wave order; or
    //      the GEMM planes of those eight columns (split_kernel<SPLIT_PLAIN>'s body: 8 columns per thread, the sum of a
    //      32-column block over 4 neighbouring threads) ---------------------------------------------------------------
    for (int i = threadIdx.x; i < 32 * (HD / 8); i += 64 * nact) {
      const int row = i / (HD / 8), c8 = i % (HD / 8);
      const bool live = t.q0 + row < T;
      f32x4 lo = *reinterpret_cast<const f32x4*>(sm_o + row * HD + 8 * c8);
      f32x4 hi = *reinterpret_cast<const f32x4*>(sm_o + row * HD + 8 * c8 + 4);
      for (int w = 1; w < nact; ++w) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(sm_o + (size_t)w * 32 * HD + row * HD + 8 * c8);
        const f32x4 y = *reinterpret_cast<const f32x4*>(sm_o + (size_t)w * 32 * HD + row * HD + 8 * c8 + 4);
        lo[0] += x[0]; lo[1] += x[1]; lo[2] += x[2]; lo[3] += x[3];
        hi[0] += y[0]; hi[1] += y[1]; hi[2] += y[2]; hi[3] += y[3];
      }
      const float inv = sm_inv[row];
      lo[0] *= inv; lo[1] *= inv; lo[2] *= inv; lo[3] *= inv;
      hi[0] *= inv; hi[1] *= inv; hi[2] *= inv; hi[3] *= inv;
      if (!pl.planes) {
        if (live) {
          float* op = out + ((size_t)(t.q0 + row) * nh + head) * HD + 8 * c8;
          *reinterpret_cast<f32x4*>(op) = lo;
          *reinterpret_cast<f32x4*>(op + 4) = hi;
        }
        continue;  // (uniform: pl is a kernel argument)
      }
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      float bs = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      bs += __shfl_xor(bs, 1);  // (the four threads of a block are neighbours of one row: all here together)
      bs += __shfl_xor(bs, 2);
      if (!live) continue;
      const int tok = t.q0 + row, K = nh * HD, i0 = head * HD + 8 * c8;
      if ((c8 & 3) == 0) pl.xsum[(size_t)(i0 >> 5) * pl.tpad + tok] = bs;
      for (int p = 0; p < pl.np; ++p) {
        pf_bf16x8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          b[e] = (__bf16)v[e];
          v[e] -= (float)b[e];
        }
        *reinterpret_cast<pf_bf16x8*>(pl.planes + p * pl.plane_stride + (size_t)tok * K + i0) = b;
      }
    }
    PFS(10 + 12 * ph);
  }
}

}  // namespace

#ifdef TL_PF_STAMP
extern "C" int tl_debug_pf_stamps(unsigned long long* out_host) {
  TL_HIP(hipDeviceSynchronize());
  TL_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(pf_stamps), sizeof(unsigned long long) * 256 * 8 * 32));
  return 0;
}
#endif

bool tl_attn_prefill_mfma_supported(int nh, int nkv, int hd) {
  return (hd == 64 || hd == 128) && nkv > 0 && nh % nkv == 0;
}

// q [T][nh][hd] (rotated), caches [max_seq][nkv][hd] already holding positions 0..start_pos+T-1, out [T][nh][hd]
int tl_launch_attn_prefill_mfma(const float* q, const float* kc, const float* vc, float* out, int T, int start_pos,
                                int nh, int nkv, int hd, float scale, hipStream_t s, void* planes, size_t plane_stride,
                                int terms) {
  TL_REQUIRE(tl_attn_prefill_mfma_supported(nh, nkv, hd), "tl_launch_attn_prefill_mfma: unsupported head shape");
  if (T == 0) return 0;
  PfPlanes pl = {nullptr, 0, nullptr, 0, 0};
  if (planes) {  // layout as tl_launch_split makes it: three bf16 planes, then the block sums
    TL_REQUIRE(plane_stride >= tl_gemm_padded_tokens(T) * (size_t)nh * hd, "prefill attention: activation planes too small");
    pl.planes = static_cast<__bf16*>(planes);
    pl.plane_stride = plane_stride;
    pl.xsum = reinterpret_cast<float*>(pl.planes + 3 * plane_stride);
    pl.np = terms == 1 ? 1 : terms == 3 ? 2 : 3;
    pl.tpad = (int)tl_gemm_padded_tokens(T);
  }
  const dim3 grid((((T + 31) / 32 + 1) / 2) * nh), block(64 * PF_NW);
  if (hd == 128) {
    static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
    TL_TRY(tl_raise_lds_limit(attn_prefill_mfma_kernel<128>, 160 * 1024, lds_raised));
    attn_prefill_mfma_kernel<128><<<grid, block, sizeof(float) * pf_lds_floats<128>(), s>>>(q, kc, vc, out, T, start_pos, nh, nkv, scale, pl);
  } else {
    static TlDeviceOnce lds_raised;
    TL_TRY(tl_raise_lds_limit(attn_prefill_mfma_kernel<64>, 160 * 1024, lds_raised));
    attn_prefill_mfma_kernel<64><<<grid, block, sizeof(float) * pf_lds_floats<64>(), s>>>(q, kc, vc, out, T, start_pos, nh, nkv, scale, pl);
  }
  TL_HIP(hipGetLastError());
  return 0;
}
