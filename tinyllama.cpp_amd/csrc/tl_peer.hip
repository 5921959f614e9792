// Peer-to-peer exchange for the tensor-parallel decode step: one-shot "push" all-reduce / all-gather over memory that
// every rank of the group maps from every other rank (hipIpc handles; xGMI between the GPUs of one node).
//
// Why not the library collective: the decode step exchanges f32[hidden] (16 KB for Llama-3-8B) 2 x layers times per
// token. Messages of that size are pure latency, and a general-purpose ring / tree collective costs several kernel
// phases and flag round trips. Here one small kernel per exchange does everything in ONE trip over the links:
//   - every value travels as an 8-byte word {f32 bits, epoch}: the word validates itself, so no fence, no separate flag
//     and no second round trip are needed (8-byte aligned stores are single transactions) -- the "LL" idea of
//     NCCL/RCCL's low-latency protocol;
//   - rank r stores its words straight into slot [r] of every peer's receive area (xGMI is point to point: 7 links per
//     GPU, one per peer, all used at once), then polls its OWN area (local memory) for the words of the others and
//     sums them in rank order, so that every rank computes bit-identical results (they all sample the same token);
//   - two receive slots alternate: a sender can only be two exchanges ahead of a receiver after having received that
//     receiver's words of the exchange in between, which the receiver sends after it finished reading (stream order);
//   - all accesses to the shared areas are system-scope atomics (cache-bypassing loads / write-through stores) on
//     uncached allocations, so neither the writer's nor the reader's L2 (8 per GPU) ever holds a stale word;
//   - every poll is bounded by a wall-clock deadline: a missing peer raises an error flag, never a hung GPU.
// The kernels are plain launches: they are captured into the per-rank step graph like every other kernel of the step.
// Folded form: the same protocol carried by the GEMV launches on the producing side of an exchange
// (tl_gemv.hip, PEER instantiations; the object they read is TlPeerDev): the lane that holds a finished row pushes its
// word to every rank, and at the end of its work the wave polls this rank's area for its own rows of all ranks, sums them
// in rank order (row-split projections) or files them (lm_head) and writes the result where the next launch reads it: no
// exchange launch at all, 2 x layers + 1 fewer launches per token. These exchanges have areas and epochs of their own
// (one epoch word per kind, advanced by the last workgroup of a launch), so both forms can be used side by side.
// No counterpart in the reference (its multi-GPU story is a mock MPI coordinator).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "tl_internal.h"

namespace {

typedef unsigned long long u64;
constexpr int PEER_THREADS = 256;

// (global address space: a generic-pointer access compiles to flat_* and makes every wait a full one)
typedef u64 __attribute__((address_space(1))) * gptr64;
__device__ __forceinline__ void ll_store(u64* p, u64 w) {
  __hip_atomic_store((gptr64)(uintptr_t)p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ u64 ll_load(const u64* p) {
  return __hip_atomic_load((gptr64)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// rx.p[r]: base of THIS exchange kind's receive area on rank r as mapped into this process: u64 [2 slots][tp][n]
// (by value and indexed by unrolled constants only: the pointers arrive in SGPRs with the other kernel arguments).
struct PeerTable {
  u64* p[TL_PEER_MAX_RANKS];
};
// ctr[blockIdx.x]: number of exchanges of this kind this workgroup has completed (the epoch travels in the words).
template <bool GATHER>
__global__ __launch_bounds__(PEER_THREADS) void peer_exchange_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                     PeerTable rx, unsigned* __restrict__ ctr,
                                                                     unsigned* __restrict__ err, int tp, int rank, int n,
                                                                     u64 timeout_ticks) {
  const int i = blockIdx.x * PEER_THREADS + threadIdx.x;
  const bool act = i < n;
  unsigned e = ctr[blockIdx.x] + 1u;
  if (e == 0u) e = 1u;  // 0 is the cleared state of the areas
  const size_t slot = (size_t)(e & 1u) * (size_t)tp * (size_t)n;
  const float mine = act ? in[i] : 0.f;
  if (act) {
    const u64 w = ((u64)e << 32) | (u64)__float_as_uint(mine);
#pragma unroll
    for (int p = 0; p < TL_PEER_MAX_RANKS; ++p)
      if (p < tp && p != rank) ll_store(rx.p[p] + slot + (size_t)rank * n + i, w);
  }
  bool ok = true;
  if (act) {
    const u64* mine_rx = rx.p[0];
#pragma unroll
    for (int q = 1; q < TL_PEER_MAX_RANKS; ++q) mine_rx = q == rank ? rx.p[q] : mine_rx;
    const u64* mybase = mine_rx + slot + i;
    u64 w[TL_PEER_MAX_RANKS];
#pragma unroll
    for (int q = 0; q < TL_PEER_MAX_RANKS; ++q)  // all polls of the first round in flight together
      w[q] = (q < tp && q != rank) ? ll_load(mybase + (size_t)q * n) : 0ull;
    const u64 deadline = __builtin_amdgcn_s_memrealtime() + timeout_ticks;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < TL_PEER_MAX_RANKS; ++q) {
      if (q >= tp) continue;
      float v = mine;
      if (q != rank) {
        u64 x = w[q];
        for (unsigned it = 1; (unsigned)(x >> 32) != e; ++it) {
          if ((it & 63u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) {
            ok = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          x = ll_load(mybase + (size_t)q * n);
        }
        v = __uint_as_float((unsigned)x);
        if (GATHER) out[(size_t)q * n + i] = v;
      }
      acc += v;  // rank order 0 .. tp-1 on every rank
    }
    if (!GATHER) out[i] = acc;
  }
  if (!ok) atomicOr(err, 1u);
  __syncthreads();
  if (threadIdx.x == 0) ctr[blockIdx.x] = e;
}

}  // namespace

struct TlPeer {
  int tp = 0, rank = 0, n_reduce = 0, n_gather = 0;
  u64* area = nullptr;                 // this rank's receive areas: reduce [2][tp][n_reduce], gather [2][tp][n_gather]
  size_t area_words = 0;
  std::vector<void*> mapped;           // peers' areas as opened here (nullptr for the own rank)
  PeerTable rx_reduce, rx_gather;      // every rank's area of each kind, as mapped here
  unsigned* ctr_reduce = nullptr;      // per-workgroup exchange counters
  unsigned* ctr_gather = nullptr;
  unsigned* err = nullptr;
  u64 timeout_ticks = 0;
  bool connected = false;
  // folded form: area sections 3 and 4 (same shapes as 1 and 2), {epoch, done} words per kind, device descriptions
  unsigned* fused_words = nullptr;     // [0,1] reduce epoch / done, [2,3] gather epoch / done
  TlPeerDev* fused_dev = nullptr;      // [0] reduce, [1] gather
};

static int peer_blocks(int n) { return (n + PEER_THREADS - 1) / PEER_THREADS; }

void tl_peer_destroy(TlPeer* p) {
  if (!p) return;
  for (void* q : p->mapped)
    if (q) (void)hipIpcCloseMemHandle(q);
  (void)hipFree(p->ctr_reduce);
  (void)hipFree(p->ctr_gather);
  (void)hipFree(p->err);
  (void)hipFree(p->fused_words);
  (void)hipFree(p->fused_dev);
  (void)hipFree(p->area);
  delete p;
}

int tl_peer_create(int tp, int rank, int n_reduce, int n_gather, TlPeer** out, void* handle_out_64) {
  TL_REQUIRE(out && handle_out_64, "tl_peer_create: null argument");
  TL_REQUIRE(tp >= 2 && tp <= TL_PEER_MAX_RANKS && rank >= 0 && rank < tp, "tl_peer_create: %d ranks (2..%d), rank %d", tp,
             TL_PEER_MAX_RANKS, rank);
  TL_REQUIRE(n_reduce > 0 && n_gather > 0, "tl_peer_create: empty exchange");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  TlPeer* p = new TlPeer();
  p->tp = tp;
  p->rank = rank;
  p->n_reduce = n_reduce;
  p->n_gather = n_gather;
  p->mapped.assign(tp, nullptr);
  p->area_words = 2 * 2 * (size_t)tp * ((size_t)n_reduce + (size_t)n_gather);  // (launch form + folded form)
  double timeout_ms = 5000.0;
  if (const char* e = getenv("TL_TP_PEER_TIMEOUT_MS")) {
    const double v = atof(e);
    if (v >= 1.0 && v <= 60000.0) timeout_ms = v;
  }
  p->timeout_ticks = (u64)(timeout_ms * 1e5);  // s_memrealtime counts at 100 MHz
  auto fail = [&](const char* what, hipError_t e) {
    tl_set_error("tl_peer_create: %s failed: %s", what, hipGetErrorString(e));
    tl_peer_destroy(p);
    return 1;
  };
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&p->area), p->area_words * sizeof(u64), hipDeviceMallocUncached);
  if (e != hipSuccess) return fail("hipExtMallocWithFlags(uncached)", e);
  if ((e = hipMemset(p->area, 0, p->area_words * sizeof(u64))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMalloc(&p->ctr_reduce, sizeof(unsigned) * peer_blocks(n_reduce))) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&p->ctr_gather, sizeof(unsigned) * peer_blocks(n_gather))) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMalloc(&p->err, sizeof(unsigned))) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMemset(p->ctr_reduce, 0, sizeof(unsigned) * peer_blocks(n_reduce))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMemset(p->ctr_gather, 0, sizeof(unsigned) * peer_blocks(n_gather))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMemset(p->err, 0, sizeof(unsigned))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMalloc(&p->fused_words, sizeof(unsigned) * 4)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipMemset(p->fused_words, 0, sizeof(unsigned) * 4)) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMalloc(&p->fused_dev, sizeof(TlPeerDev) * 2)) != hipSuccess) return fail("hipMalloc", e);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail("hipDeviceSynchronize", e);
  hipIpcMemHandle_t h;
  if ((e = hipIpcGetMemHandle(&h, p->area)) != hipSuccess) return fail("hipIpcGetMemHandle", e);
  memcpy(handle_out_64, &h, sizeof(h));
  *out = p;
  return 0;
}

int tl_peer_connect(TlPeer* p, const void* handles) {
  TL_REQUIRE(p && handles, "tl_peer_connect: null argument");
  TL_REQUIRE(!p->connected, "tl_peer_connect: already connected");
  for (int q = 0; q < TL_PEER_MAX_RANKS; ++q) p->rx_reduce.p[q] = p->rx_gather.p[q] = p->area;  // (unused entries stay valid)
  for (int q = 0; q < p->tp; ++q) {
    u64* base = p->area;
    if (q != p->rank) {
      hipIpcMemHandle_t h;
      memcpy(&h, static_cast<const char*>(handles) + (size_t)q * sizeof(h), sizeof(h));
      void* ptr = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess || !ptr) {
        tl_set_error("tl_peer_connect: hipIpcOpenMemHandle of rank %d's area failed: %s", q, hipGetErrorString(e));
        (void)hipGetLastError();
        return 1;
      }
      p->mapped[q] = ptr;
      base = static_cast<u64*>(ptr);
    }
    p->rx_reduce.p[q] = base;
    p->rx_gather.p[q] = base + 2 * (size_t)p->tp * (size_t)p->n_reduce;
  }
  TlPeerDev d[2] = {};
  const size_t half = 2 * (size_t)p->tp * ((size_t)p->n_reduce + (size_t)p->n_gather);
  for (int q = 0; q < TL_PEER_MAX_RANKS; ++q) {
    d[0].rx[q] = p->rx_reduce.p[q] + half;
    d[1].rx[q] = p->rx_gather.p[q] + half;
  }
  for (int k = 0; k < 2; ++k) {
    d[k].epoch = p->fused_words + 2 * k;
    d[k].done = p->fused_words + 2 * k + 1;
    d[k].err = p->err;
    d[k].timeout_ticks = p->timeout_ticks;
    d[k].tp = p->tp;
    d[k].rank = p->rank;
    d[k].n = k == 0 ? p->n_reduce : p->n_gather;
  }
  TL_HIP(hipMemcpy(p->fused_dev, d, sizeof(d), hipMemcpyHostToDevice));
  p->connected = true;
  return 0;
}

const TlPeerDev* tl_peer_fused_reduce(TlPeer* p) { return p && p->connected ? p->fused_dev : nullptr; }
const TlPeerDev* tl_peer_fused_gather(TlPeer* p) { return p && p->connected ? p->fused_dev + 1 : nullptr; }

int tl_peer_all_reduce(TlPeer* p, const float* in, float* out, hipStream_t s) {
  TL_REQUIRE(p && p->connected && in && out && in != out, "tl_peer_all_reduce: not connected / bad buffers");
  peer_exchange_kernel<false><<<peer_blocks(p->n_reduce), PEER_THREADS, 0, s>>>(in, out, p->rx_reduce, p->ctr_reduce, p->err,
                                                                                p->tp, p->rank, p->n_reduce, p->timeout_ticks);
  TL_HIP(hipGetLastError());
  return 0;
}

// in place: this rank's n_gather floats already sit at buf + rank * n_gather
int tl_peer_all_gather(TlPeer* p, float* buf, hipStream_t s) {
  TL_REQUIRE(p && p->connected && buf, "tl_peer_all_gather: not connected / null buffer");
  peer_exchange_kernel<true><<<peer_blocks(p->n_gather), PEER_THREADS, 0, s>>>(buf + (size_t)p->rank * p->n_gather, buf,
                                                                               p->rx_gather, p->ctr_gather, p->err, p->tp,
                                                                               p->rank, p->n_gather, p->timeout_ticks);
  TL_HIP(hipGetLastError());
  return 0;
}

// copies the device error flag (asynchronously, into *flag_host) behind the work enqueued on s
int tl_peer_error_async(TlPeer* p, unsigned* flag_host, hipStream_t s) {
  TL_HIP(hipMemcpyAsync(flag_host, p->err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  return 0;
}

int tl_peer_clear_error(TlPeer* p, hipStream_t s) {
  TL_HIP(hipMemsetAsync(p->err, 0, sizeof(unsigned), s));
  return 0;
}

// Collective self-check (every rank of the group calls it once after tl_peer_connect): three all-reduces (both slots
// and one reuse) and two all-gathers of rank-dependent patterns with known results. Returns 0 and *ok_out = 1 when
// every value arrived exactly; *ok_out = 0 on a timeout or a wrong value (the caller then keeps the library collective).
int tl_peer_selftest(TlPeer* p, hipStream_t s, int* ok_out) {
  TL_REQUIRE(p && p->connected && ok_out, "tl_peer_selftest: not connected");
  *ok_out = 0;
  const int nr = p->n_reduce, ng = p->n_gather, tp = p->tp;
  float *in = nullptr, *out = nullptr, *gbuf = nullptr;
  TL_HIP(hipMalloc(&in, sizeof(float) * nr));
  TL_HIP(hipMalloc(&out, sizeof(float) * nr));
  TL_HIP(hipMalloc(&gbuf, sizeof(float) * (size_t)ng * tp));
  std::vector<float> h(nr), r(nr), g((size_t)ng * tp);
  bool good = true;
  int rc = 0;
  for (int round = 0; round < 3 && rc == 0; ++round) {
    for (int i = 0; i < nr; ++i) h[i] = (float)((p->rank + 1) * ((i + round) % 251 + 1));  // small integers: exact sums
    if (hipMemcpy(in, h.data(), sizeof(float) * nr, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    if (rc == 0) rc = tl_peer_all_reduce(p, in, out, s);
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = 1;
    if (rc == 0 && hipMemcpy(r.data(), out, sizeof(float) * nr, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    const float ranks_sum = (float)(tp * (tp + 1) / 2);
    for (int i = 0; i < nr && rc == 0; ++i) good &= r[i] == ranks_sum * (float)((i + round) % 251 + 1);
    if (!good) break;  // (a timed-out poll leaves a wrong value behind: no point in waiting for the next one)
  }
  for (int round = 0; round < 2 && rc == 0 && good; ++round) {
    std::fill(g.begin(), g.end(), -1.0f);
    for (int i = 0; i < ng; ++i) g[(size_t)p->rank * ng + i] = (float)(p->rank * 1000 + (i + round) % 997);
    if (hipMemcpy(gbuf, g.data(), sizeof(float) * g.size(), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    if (rc == 0) rc = tl_peer_all_gather(p, gbuf, s);
    if (rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = 1;
    if (rc == 0 && hipMemcpy(g.data(), gbuf, sizeof(float) * g.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    for (int q = 0; q < tp && rc == 0; ++q)
      for (int i = 0; i < ng; ++i) good &= g[(size_t)q * ng + i] == (float)(q * 1000 + (i + round) % 997);
  }
  unsigned flag = 1;
  if (rc == 0 && hipMemcpy(&flag, p->err, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
  (void)hipFree(in);
  (void)hipFree(out);
  (void)hipFree(gbuf);
  if (rc > 1) return rc;
  TL_REQUIRE(rc == 0, "tl_peer_selftest: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
  *ok_out = good && flag == 0;
  return 0;
}
