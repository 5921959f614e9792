// tl_internal.h — shared declarations inside libtlhip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "tlhip.h"
#include "tlhip_debug.h"

// ---- error plumbing -------------------------------------------------------------------------------
void tl_set_error(const char* fmt, ...);
#define TL_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      tl_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
#define TL_REQUIRE(cond, ...)    \
  do {                           \
    if (!(cond)) {               \
      tl_set_error(__VA_ARGS__); \
      return 2;                  \
    }                            \
  } while (0)
#define TL_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != 0) return rc_; \
  } while (0)

// hipFuncSetAttribute is per DEVICE (every device has its own copy of the code object): remember it per device ordinal,
// not per process, so that a process driving several GPUs raises the LDS limit on each of them. `done` is the call
// site's own static mask (one per kernel instantiation).
struct TlDeviceOnce {
  unsigned long long mask[2] = {0, 0};  // ordinals 0..127
};
template <class K>
static inline int tl_raise_lds_limit(K kernel, int bytes, TlDeviceOnce& done) {
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  TL_REQUIRE(dev >= 0 && dev < 128, "device ordinal %d out of range", dev);
  unsigned long long& m = done.mask[dev >> 6];
  if (m >> (dev & 63) & 1ull) return 0;
  TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  m |= 1ull << (dev & 63);
  return 0;
}
// CU count of the current device (cached per ordinal)
static inline int tl_num_cus() {
  static int cus[128] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 128) return 256;
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}

// lazily grown per-(device, slot) workspace of the operator-level entry points (tl_runtime.hip)
enum TlScratchSlot {
  TL_SCRATCH_ATTN = 0,
  TL_SCRATCH_GEMM_PLANES,
  TL_SCRATCH_SAMPLE_PROBS,
  TL_SCRATCH_SAMPLE_PARAMS,
  TL_SCRATCH_GEMM_PART0,  // + 0..7: K-split partial tiles, one workspace per stream (tl_gemm.hip)
  TL_SCRATCH_GEMM_PART7 = TL_SCRATCH_GEMM_PART0 + 7
};
int tl_scratch(int slot, size_t bytes, void** out);
// tl_ops.hip: looks up the device address of tl_attention_noalloc's static workspace (called by tl_init, device current)
int tl_ops_init_device(int dev);

// ---- device layout of one quantized matrix -----------------------------------------------------------
// The GGUF blocks (quantization.h:57-124) are split at upload into 16-byte-aligned planes, row-major by
// output row, so that one wave-wide 16 B/lane load is a contiguous, aligned 1 KiB and every lane's 16 B
// of payload is exactly one 32-weight "chunk" over 32 CONSECUTIVE columns:
//
//   Q4_K  p0 qs'  [rows][cols/2]        chunk = 16 B: dword w byte t = (weight 8w+t) | (weight 8w+4+t)<<4
//         p1 hdr' [rows][cols/256][16]  {d:f16 | dmin:f16<<16, sc_lo4[8 x 4b], m_lo4[8 x 4b],
//                                         sc_hi2[8 x 2b] | m_hi2[8 x 2b]<<16}
//   Q6_K  p0 ql'  [rows][cols/2]        low nibbles, same nibble order as Q4_K
//         p1 qh'  [rows][cols/4]        chunk = 8 B: dword v byte t bits 2g..2g+1 = high 2 bits of weight 16v+4g+t
//         p2 sc   [rows][cols/16] i8    natural order
//         p3 d    [rows][cols/256] f16
//   Q8_0  p0 qs   [rows][cols] i8, p1 d [rows][cols/32] f16
//   Q8_K  p0 qs   [rows][cols] i8, p1 d [rows][cols/256] f16        (bsums are not needed and not kept)
//   F32   p0      [rows][cols] f32
// Total device bytes == GGUF bytes (Q8_K: minus the 32 B of bsums per block).
struct tl_qweight {
  int type, rows, cols;
  size_t bytes;  // device bytes held
  void* base;    // single allocation
  uint8_t *p0, *p1, *p2, *p3;
};

struct QW {  // by-value kernel argument
  const uint8_t *p0, *p1, *p2, *p3;
  int rows, cols, type;
};
static inline QW qw_view(const tl_qweight* w) {
  return QW{w->p0, w->p1, w->p2, w->p3, w->rows, w->cols, w->type};
}

size_t tl_type_block_bytes(int type);
int tl_type_block_elems(int type);
// device bytes of a [rows x cols] matrix of `type` in the plane layout
size_t tl_qweight_device_bytes(int type, int rows, int cols);
// builds a tl_qweight over caller-provided storage [base, base+bytes) from raw GGUF blocks on the host
int tl_qweight_build(int type, const void* host_blocks, int rows, int cols, tl_qweight* w, hipStream_t s);

// ---- launchers shared between the operator-level C-ABI and the model engine ---------------------------
// GEMV over 1..3 matrices that share the input vector (q|k|v in one launch), with optional fused
// RMSNorm prologue, residual epilogue, or gate/up SwiGLU pairing; see tl_gemv.hip
// One kind of exchange folded into a GEMV's epilogue (tl_peer.hip owns the object, in device memory; tl_gemv.hip's PEER
// instantiations read it): the words of a row go out to every rank as soon as the row is reduced, and the wave that
// pushed them polls this rank's own area for the same rows of the others before it ends. See tl_peer.hip.
constexpr int TL_PEER_MAX_RANKS = 8;
struct TlPeerDev {
  unsigned long long* rx[TL_PEER_MAX_RANKS];  // every rank's receive area of this kind as mapped here: u64 [2 slots][tp][n]
  unsigned* epoch;                            // exchanges of this kind completed on this rank (the words carry epoch + 1)
  unsigned* done;                             // workgroups of the running launch that are through (the last one advances epoch)
  unsigned* err;                              // sticky: a poll gave up at its deadline
  unsigned long long timeout_ticks;           // of the 100 MHz realtime counter
  int tp, rank, n;
};
enum GemvRole { GEMV_GENERIC = 0, GEMV_QKV, GEMV_ATTN_OUT, GEMV_GATE_UP, GEMV_DOWN, GEMV_LM_HEAD, GEMV_ATTN_OUT_COMBINE };
struct GemvSeg {
  QW w;
  float* out;
};
struct GemvArgs {
  GemvSeg seg[3];
  int nseg;
  int pair;             // 1: out0[r] = swiglu(seg0[r].x, seg1[r].x)  (decode form, model.cpp:734-735)
  int cols;             // shared K
  const float* x;       // [cols]
  const float* norm_w;  // != NULL: x := rmsnorm_vector(x, norm_w, eps) first (utils.cpp:648-673)
  float eps;
  const float* resid;   // != NULL: out0[r] = resid[r] + y[r]   (single segment only)
  // != NULL: x is the attention output still in split partials (AttnArgs::ws); the staging pass combines them.
  // x is then ignored; cols = attn_nh * attn_hd.
  const float* attn_ws;
  const int* attn_pos;  // device position (sequence length - 1)
  int attn_hd;
  int attn_merged;      // 1: the split kernel ran with AttnArgs::done_cnt (one record per head from TL_ATTN_MERGE_MIN splits on)
  int role;             // GemvRole: selects a separately named instantiation (profiling only)
  // != NULL (tensor parallelism, single quantized segment): the exchange is part of this launch. With resid (row-split
  // projection): out[r] = sum over ranks, in rank order, of (resid[r] + y[r]) -- an all-reduce; without (lm_head): this
  // rank's rows go to out[r] and rank q's rows to out[(q - rank) * n + r] -- an all-gather in place.
  const TlPeerDev* peer;
};
int tl_launch_gemv(const GemvArgs& a, hipStream_t s);
// chain launches (tl_gemv.hip): several fused GEMV launches of the decode step as tasks of ONE launch, handed over through
// arrival counters instead of kernel boundaries
constexpr size_t TL_CHAIN_TASK_BYTES = 384;  // sizeof(ChainTask) (checked in tl_gemv.hip), the stride of a task array
int tl_chain_plan(const GemvArgs& a, void* task_out, size_t* lds_out, int pred_nwg, int* nwg_out, bool* ok);
int tl_launch_chain(const void* tasks_dev, int ntasks, int max_nwg, size_t lds, unsigned* sync, const unsigned* gen,
                    unsigned* err, hipStream_t s);
size_t tl_chain_sync_words(int ntasks);
int tl_launch_chain_gen_bump(unsigned* gen, hipStream_t s);
int tl_launch_prefetch(const void* p, size_t bytes, int blocks, unsigned* sink, hipStream_t s);

int tl_launch_rmsnorm(const float* x, const float* w, float* out, int T, int n, float eps, int floor_eps,
                      hipStream_t s);
int tl_launch_dequant_rows(const tl_qweight* w, const int* row_index_dev, int row0, int nrows, float* out,
                           hipStream_t s);
int tl_launch_dequant_gather(const tl_qweight* w, const int* ids_dev, int n, float* out, hipStream_t s);
int tl_launch_rope_kv(float* q, float* k, const float* v, float* kcache, float* vcache, const float* table, int nq,
                      int nkv, int hd, const int* pos_dev, int max_pos, hipStream_t s);
int tl_launch_attn_decode(const float* q, const float* kc, const float* vc, float* out, int nh, int nkv, int hd,
                          int seq_len, const int* pos_dev, int max_seq, float scale, hipStream_t s);
// split decode attention (tl_attn.hip)
constexpr int TL_ATTN_MAX_SPLITS = 32;
// positions per split for a sequence of `len` positions: max(64, pow2ceil(len / 32)) -> at most 32 splits. 64 rows are
// two row slots per wave of the split kernel, the smallest unit it issues anyway; 32-row splits only doubled the records
// the consumer has to combine (len 500: attention + attn_output 22.8 -> 20.3 us per layer).
__host__ __device__ inline int tl_attn_split_len(int len) {
  int sl = 64;
  while (sl * TL_ATTN_MAX_SPLITS < len) sl <<= 1;
  return sl;
}
// Number of splits from which the split kernel merges its own partials (the last workgroup of a head group to finish
// does it, when AttnArgs::done_cnt is set) instead of leaving `active` records per head to the consumer: the attn_output
// GEMV re-reads every record in all 256 workgroups: measured 9.7 us at 5 splits, 18.6 us at 32 (+0.33 us per split)
// against +3.9 us for the in-kernel merge, which therefore pays from 17 splits on.
constexpr int TL_ATTN_MERGE_MIN = 17;
// records per head the consumer of the partials has to combine for a sequence of `len` positions
__host__ __device__ inline int tl_attn_records(int len, bool merged) {
  const int sl = tl_attn_split_len(len);
  const int active = (len + sl - 1) / sl;
  return merged && active >= TL_ATTN_MERGE_MIN ? 1 : active;
}
struct AttnArgs {
  const float* q;        // [nh][hd]; FUSE: un-rotated (rotated into LDS only)
  const float* k;        // FUSE only: un-rotated k [nkv][hd] of the new position
  const float* v;        // FUSE only: v [nkv][hd] of the new position
  float* kc;             // K cache [max_seq][nkv][hd] (FUSE: position pos is written)
  float* vc;             // V cache
  const float* table;    // FUSE only: RoPE table [max_seq][hd/2][2]
  float* out;            // [nh][hd] (combine only)
  float* ws;             // partials [nh][TL_ATTN_MAX_SPLITS][hd + 4]: out[hd] unnormalised, max, sum, pad
  int nh, nkv, hd;
  int seq_len;           // positions 0..seq_len-1 when pos_dev == NULL
  const int* pos_dev;    // != NULL: the new position lives on the device (graph replay); seq_len = *pos_dev + 1
  int max_seq;
  float scale;
  // != NULL: one zero-initialised counter per head group (nh / G of them; the kernel leaves them at zero). From
  // TL_ATTN_MERGE_MIN active splits on, the last workgroup of a head group merges the group's partials into record 0
  unsigned* done_cnt;
  // K/V addressing in floats: position t of kv head h starts at t * kv_row + h * kv_head. 0 / 0 = the reference's
  // [max_seq][nkv][hd] (kv_row = nkv * hd, kv_head = hd), filled in by the launcher
  int kv_row;
  long long kv_head;
  int zparts;  // = gridDim.z, set by the launcher (an explicit argument: gridDim comes from the hidden argument block
               // with a scalar load of its own)
};
bool tl_attn_split_supported(int nh, int nkv, int hd);
size_t tl_attn_ws_floats(int nh, int hd);
int tl_launch_attn_split(const AttnArgs& a, bool fuse, hipStream_t s);
int tl_launch_attn_combine(const AttnArgs& a, hipStream_t s);
// the attn_output GEMV that carries its layer's attention (tl_gemv.hip, gemv_attn_kernel)
bool tl_gemv_attn_supported(const GemvArgs& a, const AttnArgs& t);
// sync: tl_fused_attn_sync_words() zeroed words, in lines of TL_SYNC_LINE_WORDS: arrival counter | generation (advanced
// once per step: tl_launch_chain_gen_bump) | epoch copies
constexpr int TL_SYNC_LINE_WORDS = 32;
size_t tl_fused_attn_sync_words();
int tl_launch_gemv_attn(const GemvArgs& a, const AttnArgs& t, unsigned* sync, unsigned tag, unsigned tags_per_gen,
                        unsigned* err, hipStream_t s);

// batched-prefill GEMM on the matrix cores (tl_gemm.hip): activation planes = the FP32 activations split into
// 1..3 bf16 planes [plane][padded tokens][cols]; terms = MFMAs per product (1, 3 or 6)
int tl_gemm_default_terms();
// expanded prefill planes (tl_gemm.hip, gemm_px_kernel): a resident bf16 integer plane + FP32 scale words per matrix
// (2.375 bytes per weight), from which the batched GEMM is fed by LDS-DMA instead of unpacking the quantized planes in
// every workgroup; same results. tl_qweight_free releases them.
size_t tl_gemm_expanded_bytes(int rows, int cols);
bool tl_gemm_expandable(const tl_qweight* w);
bool tl_gemm_expanded(const tl_qweight* w);
int tl_gemm_expand(const tl_qweight* w, hipStream_t s);
void tl_gemm_expand_release(const tl_qweight* w);
size_t tl_gemm_padded_tokens(int T);
size_t tl_gemm_planes_bytes(size_t plane_stride);  // three bf16 planes + FP32 block sums
// mode 0: planes = split(x[T][K]); 1: split(rmsnorm_batch(x, aux = weight[K], eps)); 2: split(swiglu(x, aux[T][K]))
// partial sums a K-split GEMM left for its consumer: value[o] = part[o] + part[n + o] + ... (S slices, in that order);
// S == 0: nothing pending
struct GemmParts {
  const float* part;
  size_t n;
  int S;
};
// handed to tl_launch_gemm_q by a caller whose next kernel does the reduction itself: buf / cap_floats in, got out
struct GemmDefer {
  float* buf;
  size_t cap_floats;
  GemmParts got;
};
// pa / pb: pending sums (RMSNORM: pa = the projection to add to the residual stream x, the sum goes to xout != x;
// SWIGLU: pa replaces x, pb replaces aux)
int tl_launch_split(int mode, const float* x, const float* aux, void* planes, size_t plane_stride, int T, int K,
                    int terms, float eps, hipStream_t s, const GemmParts* pa = nullptr, const GemmParts* pb = nullptr,
                    float* xout = nullptr);
int tl_launch_gemm_q(const QW& w, const void* planes, size_t plane_stride, float* y, const float* resid, int T,
                     int terms, hipStream_t s, GemmDefer* defer = nullptr);
// up to three matrices over the same planes in one launch when that changes no bit (tl_gemm.hip); *merged = false: nothing
// was launched
int tl_launch_gemm_q_multi(const QW* w, int n, const void* planes, size_t plane_stride, float* const* y, int T, int terms,
                           hipStream_t s, GemmDefer* const* defer, bool* merged);
// y = (resid ? resid : 0) + the pending sums (what the GEMM launch would have run itself)
int tl_launch_ksplit_reduce(const GemmParts& g, const float* resid, float* y, hipStream_t s);

// causal prefill attention on the FP32 matrix instruction (tl_prefill.hip)
bool tl_attn_prefill_mfma_supported(int nh, int nkv, int hd);
// planes != NULL: the output as the attn_output GEMM's activation planes (tl_launch_split's layout) instead of `out`
int tl_launch_attn_prefill_mfma(const float* q, const float* kc, const float* vc, float* out, int T, int start_pos,
                                int nh, int nkv, int hd, float scale, hipStream_t s, void* planes = nullptr,
                                size_t plane_stride = 0, int terms = 0);

// on-device temperature / top-k / top-p sampler (tl_sample.hip)
constexpr int TL_SAMPLE_MAX_TOP_K = 1024;
struct TlSampleParams {  // lives in device memory; `counter` is advanced by the kernel
  float temperature;
  int top_k;
  float top_p;
  int use_fixed_u;
  float fixed_u;
  unsigned counter;
  unsigned long long seed;
};
int tl_launch_sample_topk(float* logits, int n, float* probs_scratch, int* state, int state_header, int hist_cap,
                          TlSampleParams* params_dev, int* out_id, int* cand_ids, float* cand_probs, int* n_cand,
                          hipStream_t s);

int tl_launch_argmax(float* logits, int n, int* result_dev, const int* hist, const int* n_hist_dev, float penalty,
                     hipStream_t s);

// ---- tl_peer.hip: one-shot peer-to-peer exchange for the tensor-parallel decode step (hipIpc-mapped receive areas,
//      8-byte {value, epoch} words, bounded polls). tl_peer_create makes this rank's area and its 64-byte IPC handle;
//      tl_peer_connect takes the handles of all tp ranks (rank order); tl_peer_selftest is collective.
struct TlPeer;
int tl_peer_create(int tp, int rank, int n_reduce, int n_gather, TlPeer** out, void* handle_out_64);
int tl_peer_connect(TlPeer* p, const void* handles);
int tl_peer_selftest(TlPeer* p, hipStream_t s, int* ok_out);
int tl_peer_all_reduce(TlPeer* p, const float* in, float* out, hipStream_t s);
int tl_peer_all_gather(TlPeer* p, float* buf, hipStream_t s);
int tl_peer_error_async(TlPeer* p, unsigned* flag_host, hipStream_t s);
int tl_peer_clear_error(TlPeer* p, hipStream_t s);  // the flag is sticky until cleared
void tl_peer_destroy(TlPeer* p);
// the device-side descriptions of the exchanges that a GEMV launch carries itself (GemvArgs::peer); valid after connect.
// They have areas and epochs of their own: launches of tl_peer_all_reduce / all_gather may be mixed in freely.
const TlPeerDev* tl_peer_fused_reduce(TlPeer* p);
const TlPeerDev* tl_peer_fused_gather(TlPeer* p);

// ---- tl_batch.hip: position-dependent kernels of the multi-sequence decode step (sequence-major KV cache) ------------
int tl_launch_batch_rope_kv(float* q, float* k, const float* v, float* kc, float* vc, size_t seq_stride, const float* table,
                            const int* pos_dev, const int* seq_dev, int T, int nq, int nkv, int hd, int max_seq, int n_seq,
                            hipStream_t s, const GemmParts* pq = nullptr, const GemmParts* pk = nullptr,
                            const GemmParts* pv = nullptr, int start_pos = 0);
bool tl_batch_attn_supported(int nh, int nkv, int hd);
int tl_launch_batch_attn(const float* q, const float* kc, const float* vc, size_t seq_stride, float* out, const int* pos_dev,
                         const int* seq_dev, int T, int nh, int nkv, int hd, int max_seq, int n_seq, float scale,
                         hipStream_t s, void* planes = nullptr, size_t plane_stride = 0, int terms = 0);
int tl_launch_batch_argmax_advance(const float* logits, int T, int V, int* tokens_dev, int* pos_dev, int* ids_out_dev,
                                   hipStream_t s);

// multi-vector GEMV (tl_gemv.hip): nv input vectors [nv][cols] against one pass over the weights; outputs [nv][rows]
int tl_gemv_nv_capacity(int cols);
int tl_launch_gemv_nv(const GemvArgs& a, int nv, hipStream_t s);
