// tl_attn.hip — decode attention over the device-resident KV cache, split over the sequence ("flash-decoding"),
// with RoPE of q/k and the KV-cache write of the new position fused in for the engine's decode step.
//
// Replaces attention_cuda / attention_cuda_optimized (cuda_kernels.h:259,607: one block per head, scores for the
// whole context in shared memory) + rope_cuda (:239) + update_kv_cache_cuda (:344, one launch per kv head per K/V per
// layer). The math is the CPU path's (model.cpp:609-663, utils.cpp:462-486, 675-698): adjacent-pair RoPE from the
// host-computed table, scores = (q . K[t]) / sqrt(hd), softmax with max-subtraction and 1/(sum + 1e-9), out = sum p V.
//
// Shape (HBM/latency-bound byte work, no MFMA):
//   - grid (kv head, sequence split, head-group part): one workgroup serves the query heads of a kv head (GQA group), so
//     every K/V byte is read once; groups of 8 heads (TinyLlama, the 70B shape) are served by two workgroups of 4 (their
//     K/V rows are read twice, the per-row softmax work per workgroup halves: +11 % tokens/s on TinyLlama-1.1B); a split
//     is max(64, pow2(len / 32)) positions (tl_attn_split_len): at most 32 splits, so that 8 kv heads fill 256 CUs from
//     2048 positions on while short contexts still spread over tens of CUs;
//   - a K/V row is read by 16 lanes (one DPP row) x 16 B x hd/64 loads, four rows per wave load; the dot's reduction is
//     four DPP adds (quad_perm, row_half_mirror, row_mirror) and every lane of the row ends up with the score. (The first
//     version used 32 lanes per row and __shfl_xor, which is ds_bpermute here: 640 LDS-path shuffles per workgroup pass.)
//     All K AND V rows of a batch (128 VGPRs per lane, 256 rows per workgroup at hd 128) are in flight together and are
//     folded four slots at a time in arrival order into a running (max, sum, out) per row group (online softmax), so the
//     fold of the early rows overlaps the arrival of the late ones;
//   - the 32 row groups of a workgroup leave their states in LDS records; 32 lanes per head compute the weights
//     exp(m_r - M), and the records are summed by quads of lanes (a register merge of the 4 row groups of a wave costs
//     2 x 10 cross-row exchanges per head: +1.4 us at short context);
//   - every split writes a (max, sum, unnormalised out) partial per head. Up to 16 splits the partials are combined at
//     the next kernel boundary: in the x-staging prologue of the attn_output GEMV (tl_gemv.hip), or by
//     attn_combine_kernel for the operator-level API. From 17 splits on (AttnArgs::done_cnt set) the last workgroup of a
//     head group to finish merges the group's partials into record 0 (written-through stores, one counter add per
//     workgroup, agent-scope loads): +3.9 us on this kernel, but the GEMV, whose 256 workgroups each re-read every
//     record, drops from 18.6 to 10.2 us at 32 splits;
//   - FUSE: every lane rotates the query elements it needs in registers (no LDS staging, no barrier in front of the
//     rows); the split that owns the new position rotates k, writes K/V of that position into the cache and folds that
//     position from registers instead of from memory.
#include "tl_attn_body.h"

namespace {
using namespace tl_attn;

template <int HD, int G, bool FUSE>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_split_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  attn_split_body<HD, G, FUSE>(a, blockIdx.x, blockIdx.y, blockIdx.z, sm);
}

// out[h][:] = sum_s o_s e^(m_s - M) / (sum_s l_s e^(m_s - M) + 1e-9): one workgroup per head (operator-level API).
// The (max, sum) pairs of all splits are loaded in one go and the partial rows eight at a time (unrolled, clamped:
// independent loads instead of one dependent round trip per split); the sums run in split order.
__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ ws, float* __restrict__ out, int hd,
                                                           int seq_len, const int* __restrict__ pos_dev, int merged) {
  constexpr int NS = TL_ATTN_MAX_SPLITS;
  const int len = pos_dev ? *pos_dev + 1 : seq_len;
  const int active = tl_attn_records(len, merged != 0);  // (1 where the split kernel merged its partials itself)
  const int h = blockIdx.x, d = threadIdx.x;
  if (d >= hd) return;
  const int recl = hd + 4;
  const float* rec0 = ws + (size_t)h * NS * recl;
  float ms[NS], ls[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float* rec = rec0 + (size_t)(s < active ? s : active - 1) * recl;
    ms[s] = rec[hd];
    ls[s] = rec[hd + 1];
  }
  float M = -INFINITY;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < active) M = fmaxf(M, ms[s]);
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int s0 = 0; s0 < NS; s0 += 8) {
    if (s0 >= active) break;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rec0[(size_t)(s0 + j < active ? s0 + j : active - 1) * recl + d];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < active) {
        const float w = expf(ms[s0 + j] - M);
        L = fmaf(ls[s0 + j], w, L);
        O = fmaf(o[j], w, O);
      }
  }
  out[(size_t)h * hd + d] = O * (1.0f / (L + 1e-9f));  // utils.cpp:692
}

template <int HD, int G>
int launch_hg(const AttnArgs& a_in, bool fuse, hipStream_t s) {
  AttnArgs a = a_in;
  a.zparts = (a.nh / a.nkv) / G;
  const size_t lds = sizeof(float) * (64 + 2 * (size_t)ATT_NW * 4 * G + (size_t)ATT_NW * ATT_NW * 4 + (size_t)ATT_NW * 4 * G * (HD + 4));
  const dim3 grid(a.nkv, TL_ATTN_MAX_SPLITS, (a.nh / a.nkv) / G);
  if (fuse) {
    static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
    TL_TRY(tl_raise_lds_limit(attn_decode_split_kernel<HD, G, true>, 160 * 1024, lds_raised));
    attn_decode_split_kernel<HD, G, true><<<grid, ATT_THREADS, lds, s>>>(a);
  } else {
    static TlDeviceOnce lds_raised;  // per kernel instantiation, per device
    TL_TRY(tl_raise_lds_limit(attn_decode_split_kernel<HD, G, false>, 160 * 1024, lds_raised));
    attn_decode_split_kernel<HD, G, false><<<grid, ATT_THREADS, lds, s>>>(a);
  }
  TL_HIP(hipGetLastError());
  return 0;
}

// query heads of a kv head served by one workgroup: the largest divisor of the group size that is <= 4 (4 for groups of
// 4 and 8, 3 for 3 and 6, 1 for 5 and 7: then every head has its own workgroup and the K/V rows are read once per head,
// which is what the unsplit kernel does too)
int heads_per_workgroup(int group) {
  for (int p = 4; p >= 1; --p)
    if (group % p == 0) return p;
  return 1;
}

template <int HD>
int launch_h(const AttnArgs& a, int P, bool fuse, hipStream_t s) {
  switch (P) {
    case 1: return launch_hg<HD, 1>(a, fuse, s);
    case 2: return launch_hg<HD, 2>(a, fuse, s);
    case 3: return launch_hg<HD, 3>(a, fuse, s);
    default: return launch_hg<HD, 4>(a, fuse, s);
  }
}

}  // namespace

bool tl_attn_split_supported(int nh, int nkv, int hd) {
  if (nkv <= 0 || nh % nkv != 0) return false;
  return hd == 64 || hd == 128;  // any group size: see heads_per_workgroup
}

size_t tl_attn_ws_floats(int nh, int hd) { return (size_t)nh * TL_ATTN_MAX_SPLITS * (hd + 4); }

// the split kernel alone: leaves partials in a.ws (combined by the consumer)
int tl_launch_attn_split(const AttnArgs& a, bool fuse, hipStream_t s) {
  TL_REQUIRE(tl_attn_split_supported(a.nh, a.nkv, a.hd), "attention: unsupported head configuration %d/%d/%d", a.nh,
             a.nkv, a.hd);
  TL_REQUIRE(a.ws, "attention: workspace missing");
  const int P = heads_per_workgroup(a.nh / a.nkv);
  AttnArgs b = a;
  if (b.kv_row == 0) {
    b.kv_row = a.nkv * a.hd;
    b.kv_head = a.hd;
  }
  return a.hd == 64 ? launch_h<64>(b, P, fuse, s) : launch_h<128>(b, P, fuse, s);
}

int tl_launch_attn_combine(const AttnArgs& a, hipStream_t s) {
  attn_combine_kernel<<<a.nh, 128, 0, s>>>(a.ws, a.out, a.hd, a.seq_len, a.pos_dev, a.done_cnt != nullptr);
  TL_HIP(hipGetLastError());
  return 0;
}
