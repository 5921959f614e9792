// tl_stream.hip — the ring-loader decode engine: ONE persistent launch walks a list of phases (the projections of a decode
// step, model.cpp:854-1069) with the weight stream decoupled from everything that waits.
//
// Why (profiles/r02_a_gemv_stamps.txt): a launch-per-projection GEMV spends 1.3-1.9 us between a wave's entry and its
// first weight load (argument and x round trips, cold instruction cache), ramps up over another 1-2 us, and ends with a
// 2-4 us tail in which half of the waves have run out of rows — 5 of the 7-16 us of a launch move no weights. Here:
//   - wave 0 of every workgroup (one workgroup per CU, all resident) is a LOADER: it walks the phase list and copies this
//     CU's share of every weight matrix HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no
//     registers), into a ring of NSLOTS item slots, as far ahead as the ring allows. It waits for nothing but ring space:
//     not for the input vector, not for the previous phase, not for other CUs — so HBM keeps streaming through norm,
//     reduction, hand-off and attention;
//   - waves 1..8 are CONSUMERS: an item is `rpi` rows of a matrix; consumer w owns one 2048-column span of the input
//     vector IN REGISTERS for the whole phase (32 values per lane: no LDS traffic for x at all) and multiplies it with its
//     (row, span) unit of every item; per-row partial sums of the spans meet in LDS, the wave that arrives last runs the
//     epilogue (residual / SwiGLU) and stores the row;
//   - loader and consumers talk through LDS words only (FULL sequence numbers, DONE counters); no s_barrier after the
//     prologue (the loader never joins one).
// The chunk arithmetic is tl_chunk.h's, so every row is computed exactly as the launch-per-projection GEMV computes it up
// to the order in which the span partials are added.
#include <stdio.h>
#include <stdlib.h>

#include "tl_chunk.h"

namespace {

constexpr int NCONS = 8;                      // consumer waves
constexpr int NLOAD = 2;                      // loader waves: a wave counts at most 63 copies in flight (63 KiB), two hold
                                              // what the memory latency needs; loader j copies the items seq % NLOAD == j
constexpr int NTHREADS = (NCONS + NLOAD) * 64;
constexpr int RING_BYTES = 128 * 1024;        // weight ring: items of 8-20 KiB, placed back to back (1-KiB granules)
constexpr int ITEM_MAX_BYTES = 20 * 1024;
constexpr int NCTRL = 32;                     // control entries, indexed by item sequence number
constexpr int MAX_IN_RING = 20;               // items between the oldest unreleased and the newest issued (< NCTRL - NCONS:
                                              // a control entry is reused only after its epilogue has certainly run)
constexpr int MAX_OUTSTANDING = 60;           // LDS-DMA instructions in flight per loader (vmcnt counts to 63)
constexpr int MAX_UNPUBLISHED = 7;            // their per-item counts live in one 64-bit register, a byte each
constexpr int MAX_UNITS = 2 * NCONS;          // (row, span) units per item: at most two per consumer wave
constexpr unsigned SPIN_LIMIT = 1u << 22;     // bounded spins: a stuck hand-off ends the kernel with an error word

struct SegDesc {
  const uint8_t* p[4];  // planes (tl_internal.h)
  float* out;
  int rows, type;
};

// one GEMV phase: out_s[r] = epilogue(sum_c W_s[r][c] x[c]) for up to three matrices that share x
struct PhaseDesc {
  SegDesc seg[3];
  int nseg;
  int pair;      // 1: seg0 = ffn_gate, seg1 = ffn_up; out0[r] = silu(g[r]) * u[r]
  int cols;
  int nspans;    // 2048-column spans of x
  int rpi[3];    // rows per item of each segment (pair: of the pair, counted once)
  int items[3];  // items of each segment
  int n_items;
  const float* x;
  const float* norm_w;  // != NULL: x := rmsnorm(x) * norm_w first (utils.cpp:648-673)
  const float* resid;   // != NULL: out0[r] = resid[r] + y[r]
  float eps;
};

struct Ctrl {  // LDS control block behind the ring
  unsigned full[NCTRL];    // sequence number + 1 of the item that has landed
  unsigned done[NCTRL];    // NCONS arrivals per use of the entry: every consumer has read its units and stored its partials
  unsigned start[NLOAD][NCTRL];  // private to each loader: ring offset of every item (its own view of the ring)
  unsigned fail;           // a spin ran out
  unsigned pad[3];
  float partial[NCTRL][MAX_UNITS];  // [entry][unit row * nspans + span]
  unsigned long long stamps[128][8];  // diagnostic: per-item shader-clock stamps (dumped by workgroup 0 when asked for)
};

// per-format plane geometry: bytes per row of plane p for `cols` columns (0: no such plane)
__host__ __device__ inline int plane_row_bytes(int type, int p, int cols) {
  switch (type) {
    case TL_Q4_K: return p == 0 ? cols / 2 : (p == 1 ? cols / 16 : 0);
    case TL_Q6_K: return p == 0 ? cols / 2 : (p == 1 ? cols / 4 : (p == 2 ? cols / 16 : cols / 128));
    case TL_Q8_0: return p == 0 ? cols : (p == 1 ? cols / 16 : 0);
    case TL_Q8_K: return p == 0 ? cols : (p == 1 ? cols / 128 : 0);
    default: return 0;
  }
}
__host__ __device__ inline int round1k(int b) { return (b + 1023) & ~1023; }
// ring bytes of `rpi` rows of one matrix: every plane rounded up to whole 1-KiB LDS-DMA instructions
__host__ __device__ inline int rows_ring_bytes(int type, int cols, int rpi) {
  int b = 0;
  for (int p = 0; p < 4; ++p) b += round1k(rpi * plane_row_bytes(type, p, cols));
  return b;
}
// where the next item of `sz` bytes goes (loader and consumers run the same recurrence): back to back, never split
__device__ __forceinline__ unsigned ring_place(unsigned head, unsigned sz) { return head + sz > RING_BYTES ? 0u : head; }

// ---- LDS words shared by loader and consumers ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {  // (wave-uniform address: the value goes to an SGPR)
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned lds_add(unsigned* p, unsigned v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spins until *p == want (wave-uniform); false when the spin limit ran out or another wave already failed
__device__ __forceinline__ bool lds_wait_eq(const unsigned* p, unsigned want, Ctrl* c) {
  for (unsigned spins = 0;; ++spins) {
    if (lds_ld(p) == want) return true;
    if ((spins & 63u) == 63u && (lds_ld(&c->fail) != 0u || spins > SPIN_LIMIT)) {
      lds_st(&c->fail, 1u);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// ---- loader ---------------------------------------------------------------------------------------------------------------
// one LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses into LDS [lds_dst, lds_dst + 1024). M0 carries
// the LDS address and is compiler-reserved: it is written and restored inside the statement that uses it.
__device__ __forceinline__ void dma_1k(const uint8_t* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
      : "memory");
}
// N consecutive KiB from a WAVE-UNIFORM global base (SGPR pair) + this lane's fixed offset lane * 16 (one VGPR, set once)
// with ONE M0: the instruction's immediate offset moves the global and the LDS address alike. No vector arithmetic per
// copy: the loader's whole item loop runs on the scalar unit.
template <int N>
__device__ __forceinline__ void dma_nk(const uint8_t* sbase, unsigned lane16, unsigned lds_dst) {
  static_assert(N >= 1 && N <= 4, "immediate offsets reach 3 KiB");
  unsigned keep;
  const unsigned long long sb = (unsigned long long)sbase;
  const unsigned d = __builtin_amdgcn_readfirstlane(lds_dst);
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(d), "s"(sb) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(d), "s"(sb) : "memory");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(d), "s"(sb) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(d), "s"(sb) : "memory");
}
// `ni` whole KiB from the wave-uniform base to LDS lds_dst
__device__ __forceinline__ void dma_run(const uint8_t* sbase, unsigned lane16, unsigned lds_dst, int ni) {
  for (; ni >= 4; ni -= 4) {
    dma_nk<4>(sbase, lane16, lds_dst);
    sbase += 4096;
    lds_dst += 4096u;
  }
  if (ni == 3) dma_nk<3>(sbase, lane16, lds_dst);
  else if (ni == 2) dma_nk<2>(sbase, lane16, lds_dst);
  else if (ni == 1) dma_nk<1>(sbase, lane16, lds_dst);
}

// waits until at most n & ~3 (n wave-uniform, 0..63) of this wave's vector-memory operations are outstanding: rounding
// down only waits for a few loads more. One hand-written branch tree (four bit tests on the way to the wait): a C++
// switch over 61 immediates became a structurised chain of ~50 branches per call.
__device__ __forceinline__ void wait_vmcnt(unsigned n) {
  asm volatile(
      "s_bitcmp1_b32 %0, 5\n\t"
      "s_cbranch_scc1 11f\n\t"
      "s_bitcmp1_b32 %0, 4\n\t"
      "s_cbranch_scc1 12f\n\t"
      "s_bitcmp1_b32 %0, 3\n\t"
      "s_cbranch_scc1 13f\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 14f\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_branch 9f\n\t"
      "14:\n\t"
      "s_waitcnt vmcnt(4)\n\t"
      "s_branch 9f\n\t"
      "13:\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 15f\n\t"
      "s_waitcnt vmcnt(8)\n\t"
      "s_branch 9f\n\t"
      "15:\n\t"
      "s_waitcnt vmcnt(12)\n\t"
      "s_branch 9f\n\t"
      "12:\n\t"
      "s_bitcmp1_b32 %0, 3\n\t"
      "s_cbranch_scc1 16f\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 17f\n\t"
      "s_waitcnt vmcnt(16)\n\t"
      "s_branch 9f\n\t"
      "17:\n\t"
      "s_waitcnt vmcnt(20)\n\t"
      "s_branch 9f\n\t"
      "16:\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 18f\n\t"
      "s_waitcnt vmcnt(24)\n\t"
      "s_branch 9f\n\t"
      "18:\n\t"
      "s_waitcnt vmcnt(28)\n\t"
      "s_branch 9f\n\t"
      "11:\n\t"
      "s_bitcmp1_b32 %0, 4\n\t"
      "s_cbranch_scc1 19f\n\t"
      "s_bitcmp1_b32 %0, 3\n\t"
      "s_cbranch_scc1 20f\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 21f\n\t"
      "s_waitcnt vmcnt(32)\n\t"
      "s_branch 9f\n\t"
      "21:\n\t"
      "s_waitcnt vmcnt(36)\n\t"
      "s_branch 9f\n\t"
      "20:\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 22f\n\t"
      "s_waitcnt vmcnt(40)\n\t"
      "s_branch 9f\n\t"
      "22:\n\t"
      "s_waitcnt vmcnt(44)\n\t"
      "s_branch 9f\n\t"
      "19:\n\t"
      "s_bitcmp1_b32 %0, 3\n\t"
      "s_cbranch_scc1 23f\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 24f\n\t"
      "s_waitcnt vmcnt(48)\n\t"
      "s_branch 9f\n\t"
      "24:\n\t"
      "s_waitcnt vmcnt(52)\n\t"
      "s_branch 9f\n\t"
      "23:\n\t"
      "s_bitcmp1_b32 %0, 2\n\t"
      "s_cbranch_scc1 25f\n\t"
      "s_waitcnt vmcnt(56)\n\t"
      "s_branch 9f\n\t"
      "25:\n\t"
      "s_waitcnt vmcnt(60)\n\t"
      "s_branch 9f\n\t"
      "9:"
      :
      : "s"(__builtin_amdgcn_readfirstlane(n))
      : "memory", "scc");
}

template <bool ST>
struct Loader {
  unsigned ring;             // LDS byte address of the ring
  Ctrl* c;
  unsigned lid;              // which loader this is: it copies the items with seq % NLOAD == lid, and walks all of them
  unsigned seq;              // items placed so far (this CU's item sequence over all phases)
  unsigned pubseq;           // the oldest item of this loader that is not published yet
  unsigned inflight;         // its unpublished items
  unsigned outstanding;      // their DMA instructions
  unsigned long long nq;     // their instruction counts, one byte each, newest in the low byte
  unsigned head;             // ring offset behind the newest item
  unsigned tail_seq;         // oldest item the consumers have not released yet
  unsigned tail_pos;         // its ring offset (head when the ring is empty)
  bool wrapped;              // the newest item sits below the oldest one (free space: [head, tail_pos))
  int lane;
  bool skip;                 // experiment: issue nothing
  bool nofree;               // experiment: never wait for ring space
  unsigned long long* st;    // diagnostic: per-item shader-clock stamps of workgroup 0 (NULL otherwise)

  // publishes the oldest unpublished item once its bytes have landed (loads complete in order: wait for all but the
  // instructions issued after it)
  __device__ __forceinline__ void publish_one() {
    const unsigned n_old = (unsigned)(nq >> (8u * (inflight - 1u))) & 0xffu;
    outstanding -= n_old;
    wait_vmcnt(outstanding);
    lds_st(&c->full[pubseq % NCTRL], pubseq + 1u);
    if (ST && st && lane == 0 && pubseq < 128u) st[pubseq * 8u + 2u] = __builtin_readcyclecounter();
    pubseq += NLOAD;
    --inflight;
  }
  // releases the oldest item if every consumer has arrived for it
  __device__ __forceinline__ bool try_release() {
    const unsigned e = tail_seq % NCTRL;
    const unsigned d = __hip_atomic_load(&c->done[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned nxt = __hip_atomic_load(&c->start[lid][(tail_seq + 1u) % NCTRL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(d) != NCONS * (tail_seq / NCTRL + 1u)) return false;
    ++tail_seq;
    const unsigned np = tail_seq < seq ? __builtin_amdgcn_readfirstlane(nxt) : head;
    if (tail_seq == seq || np < tail_pos) wrapped = false;
    tail_pos = np;
    return true;
  }
  __device__ __forceinline__ bool fits(unsigned pos, unsigned sz) const {
    if (nofree) return true;
    if (tail_seq == seq) return true;                       // empty ring
    if (seq - tail_seq >= (unsigned)MAX_IN_RING) return false;
    if (pos == head) return wrapped ? pos + sz <= tail_pos : true;  // (ring_place saw to pos + sz <= RING_BYTES)
    return !wrapped && sz <= tail_pos;                      // wrapping to offset 0: below the oldest item
  }

  // the items [a, b) of one segment (PAIR: of ffn_gate and ffn_up together). The format is a compile-time parameter:
  // an item is NP (x 2) straight copies whose sizes are fixed per segment
  template <int TYPE, bool PAIR>
  __device__ __forceinline__ bool run_seg(const SegDesc s, const SegDesc s2, int cols, int rpi, int a, int b) {
    constexpr int NP = TYPE == TL_Q6_K ? 4 : 2;
    int rb[NP], ni[NP], nins = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      rb[p] = plane_row_bytes(TYPE, p, cols);
      ni[p] = (rpi * rb[p] + 1023) >> 10;
      nins += ni[p];
    }
    const unsigned half = (unsigned)(nins << 10), sz = PAIR ? 2u * half : half;
    if (PAIR) nins *= 2;
    const unsigned lane16 = (unsigned)lane * 16u;
    for (int li = a; li < b; ++li) {
      const bool mine = seq % NLOAD == lid;  // (the other loaders' items are placed too: every loader keeps the whole picture)
      // the queue stays inside what vmcnt and the count register can hold
      while (mine && inflight > 0u && (outstanding + (unsigned)nins > MAX_OUTSTANDING || inflight >= (unsigned)MAX_UNPUBLISHED)) publish_one();
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 0u] = __builtin_readcyclecounter();
      const unsigned pos = ring_place(head, sz);
      if (!fits(pos, sz)) {
        while (try_release() && !fits(pos, sz)) {}
        if (!fits(pos, sz)) {
          while (inflight > 0u) publish_one();  // nothing can be issued anyway: hand over everything that is in flight
          for (unsigned spins = 0; !fits(pos, sz); ++spins) {
            if (try_release()) continue;
            if ((spins & 63u) == 63u && (lds_ld(&c->fail) != 0u || spins > SPIN_LIMIT)) {
              lds_st(&c->fail, 1u);
              return false;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (tail_seq == seq) {  // empty ring: the item becomes the oldest one wherever it goes
        tail_pos = pos;
        wrapped = false;
      } else if (pos != head) {
        wrapped = true;
      }
      lds_st(&c->start[lid][seq % NCTRL], pos);
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 1u] = __builtin_readcyclecounter();
      unsigned dst = ring + pos;
      const int row0 = li * rpi;
      const int rows_valid = s.rows - row0 < rpi ? s.rows - row0 : rpi;
#pragma unroll
      for (int hf = 0; hf < (PAIR ? 2 : 1); ++hf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const uint8_t* base = (hf ? s2.p[p] : s.p[p]) + (size_t)row0 * rb[p];
          if (skip || !mine) {
          } else if (rows_valid == rpi) {
            // whole item: straight copies; the lanes of a last, partial KiB read on into the next rows (or, behind the
            // last row of the last plane, into the slack tl_qweight_upload allocates)
            dma_run(base, lane16, dst, ni[p]);
          } else {
            const int bytes = rpi * rb[p], valid = rows_valid * rb[p];
            for (int o = 0; o < bytes; o += 1024) {  // last item of a matrix: lanes past its rows re-read its first bytes
              int lo = o + lane * 16;
              if (lo >= valid) lo = 0;
              dma_1k(base + lo, dst + (unsigned)o);
            }
          }
          dst += (unsigned)(ni[p] << 10);
        }
      }
      head = pos + sz;
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 3u] = __builtin_readcyclecounter();
      if (mine) {
        nq = (nq << 8) | (unsigned)nins;
        outstanding += (unsigned)nins;
        ++inflight;
      }
      ++seq;
    }
    return true;
  }

  __device__ __forceinline__ bool run_phase(const PhaseDesc& ph, int cu, int ncu) {
    const int n_items = ph.n_items, cols = ph.cols, nseg = ph.pair ? 1 : ph.nseg;
    const bool pair = ph.pair != 0;
    const int first = (int)((long)n_items * cu / ncu), last = (int)((long)n_items * (cu + 1) / ncu);
    int sb = 0;  // first item of the segment
    bool ok = true;
    for (int sg = 0; sg < nseg && ok; ++sg) {
      const int cnt = ph.items[sg];
      const int a = first > sb ? first - sb : 0, b = last - sb < cnt ? last - sb : cnt;
      if (a < b) {
        const int rpi = ph.rpi[sg];
        if (pair) {
          switch (ph.seg[sg].type) {
            case TL_Q4_K: ok = run_seg<TL_Q4_K, true>(ph.seg[0], ph.seg[1], cols, rpi, a, b); break;
            case TL_Q6_K: ok = run_seg<TL_Q6_K, true>(ph.seg[0], ph.seg[1], cols, rpi, a, b); break;
            case TL_Q8_0: ok = run_seg<TL_Q8_0, true>(ph.seg[0], ph.seg[1], cols, rpi, a, b); break;
            default: ok = run_seg<TL_Q8_K, true>(ph.seg[0], ph.seg[1], cols, rpi, a, b); break;
          }
        } else {
          switch (ph.seg[sg].type) {
            case TL_Q4_K: ok = run_seg<TL_Q4_K, false>(ph.seg[sg], ph.seg[sg], cols, rpi, a, b); break;
            case TL_Q6_K: ok = run_seg<TL_Q6_K, false>(ph.seg[sg], ph.seg[sg], cols, rpi, a, b); break;
            case TL_Q8_0: ok = run_seg<TL_Q8_0, false>(ph.seg[sg], ph.seg[sg], cols, rpi, a, b); break;
            default: ok = run_seg<TL_Q8_K, false>(ph.seg[sg], ph.seg[sg], cols, rpi, a, b); break;
          }
        }
      }
      sb += cnt;
    }
    return ok;
  }
  __device__ __forceinline__ void drain() {
    while (inflight > 0u) publish_one();
  }
};

// ---- consumers ------------------------------------------------------------------------------------------------------------
// one chunk out of an item slot in LDS (same fields as Chunk<T>::load reads from global memory)
template <int TYPE>
__device__ __forceinline__ void chunk_from_lds(Chunk<TYPE>& k, const uint8_t* item, int cols, int rpi, int r, unsigned c) {
  if constexpr (TYPE == TL_Q4_K) {
    const uint8_t* p0 = item + r * (cols >> 1);
    const uint8_t* p1 = item + round1k(rpi * (cols >> 1)) + r * (cols >> 4);
    k.a = *reinterpret_cast<const u32x4*>(p0 + c * 16u);
    k.b = *reinterpret_cast<const u32x4*>(p1 + (c >> 3) * 16u);
  } else if constexpr (TYPE == TL_Q6_K) {
    const int o1 = round1k(rpi * (cols >> 1)), o2 = o1 + round1k(rpi * (cols >> 2)), o3 = o2 + round1k(rpi * (cols >> 4));
    k.a = *reinterpret_cast<const u32x4*>(item + r * (cols >> 1) + c * 16u);
    k.h = *reinterpret_cast<const u32x2*>(item + o1 + r * (cols >> 2) + c * 8u);
    k.sc = *reinterpret_cast<const unsigned short*>(item + o2 + r * (cols >> 4) + c * 2u);
    k.d = *reinterpret_cast<const unsigned short*>(item + o3 + r * (cols >> 7) + (c >> 3) * 2u);
  } else {
    const int o1 = round1k(rpi * cols);
    k.a = *reinterpret_cast<const u32x4*>(item + r * cols + c * 32u);
    k.b = *reinterpret_cast<const u32x4*>(item + r * cols + c * 32u + 16u);
    k.d = TYPE == TL_Q8_0 ? *reinterpret_cast<const unsigned short*>(item + o1 + r * (cols >> 4) + c * 2u)
                          : *reinterpret_cast<const unsigned short*>(item + o1 + r * (cols >> 7) + (c >> 3) * 2u);
  }
}

struct XSpan {  // a consumer's span of the input vector: this lane's chunk and its two half sums
  float x[32];
  float s0, s1;
  bool act;  // the lane's chunk exists (cols is not always a multiple of 2048)
};

template <bool ST>
struct Consumer {
  const uint8_t* ring;
  Ctrl* c;
  unsigned seq;    // items consumed so far (this CU's item sequence over all phases)
  unsigned head;   // ring offset behind the newest item consumed (the loader's recurrence)
  int wave, lane;  // wave: 0..NCONS-1
  bool skip;       // experiment: no arithmetic
  bool nofull;     // experiment: never wait for an item to land
  unsigned long long* st;  // diagnostic stamps (wave 0 of workgroup 0)
  // the epilogue this wave still owes (item seq % NCONS == wave): run one item later, when the other waves' partials
  // are almost surely in and the wait costs a single LDS read
  struct Pending {
    bool on;
    unsigned entry, want;
    float* out;
    const float* resid;
    int row0, rows, rpi, nspans;
    bool pair;
  } pe;

  __device__ __forceinline__ bool flush_epilogue() {
    if (!pe.on) return true;
    pe.on = false;
    if (!lds_wait_eq(&c->done[pe.entry], pe.want, c)) return false;  // every consumer has stored its partials
    asm volatile("" ::: "memory");
    const int row = pe.row0 + lane;
    if (lane < pe.rpi && row < pe.rows) {
      const float* pp = c->partial[pe.entry];
      float y = 0.f, u = 0.f;
      for (int sp = 0; sp < pe.nspans; ++sp) y += pp[lane * pe.nspans + sp];
      if (pe.pair) {
        for (int sp = 0; sp < pe.nspans; ++sp) u += pp[(pe.rpi + lane) * pe.nspans + sp];
        const float sg = 1.0f / (1.0f + expf(-y));  // silu_cpu (utils.cpp:700-707), then * up (model.cpp:735)
        y = (y * sg) * u;
      } else if (pe.resid) {
        y += ldg_f32(pe.resid + row);
      }
      stg_f32(pe.out + row, y);
    }
    return true;
  }

  // the items [a, b) of one segment; `xs` is this wave's span of x. Waves are grouped by span (g = NCONS / nspans waves
  // each); wave gi of a group takes the unit rows gi and gi + g of every item.
  template <int TYPE>
  __device__ __forceinline__ bool run_seg(const XSpan& xs, float* out, const float* resid, int rows, bool pair, int cols,
                                          int rpi, int nspans, int a, int b) {
    const int g = NCONS / nspans;
    const int nur = pair ? 2 * rpi : rpi;                // unit rows of an item
    const int span = wave % nspans, gi = wave / nspans;
    const bool in_group = gi < g && !skip;
    const int ur0 = gi, ur1 = gi + g;
    const bool act0 = in_group && ur0 < nur, act1 = in_group && ur1 < nur;
    const int nchunks = cols >> 5;
    int chunk = span * 64 + lane;
    if (chunk >= nchunks) chunk = nchunks - 1;
    const unsigned half = (unsigned)rows_ring_bytes(TYPE, cols, rpi), sz = pair ? 2u * half : half;
    // pair: unit rows [0, rpi) are ffn_gate rows, [rpi, 2 rpi) the same rows of ffn_up (second half of the item)
    const unsigned off0 = (pair && ur0 >= rpi) ? half : 0u, off1 = (pair && ur1 >= rpi) ? half : 0u;
    const int r0 = (pair && ur0 >= rpi) ? ur0 - rpi : ur0, r1 = (pair && ur1 >= rpi) ? ur1 - rpi : ur1;
    for (int li = a; li < b; ++li) {
      const unsigned entry = seq % NCTRL;
      const unsigned pos = ring_place(head, sz);
      head = pos + sz;
      const uint8_t* item = ring + pos;
      // the FULL word and the units' bytes are read in one go: LDS serves a wave's reads in order, so bytes read behind
      // a FULL word that matches are the item's
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 4u] = __builtin_readcyclecounter();
      Chunk<TYPE> k0, k1;
      unsigned f = __hip_atomic_load(&c->full[entry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");
      if (act0) chunk_from_lds<TYPE>(k0, item + off0, cols, rpi, r0, (unsigned)chunk);
      if (act1) chunk_from_lds<TYPE>(k1, item + off1, cols, rpi, r1, (unsigned)chunk);
      if (!nofull && __builtin_amdgcn_readfirstlane(f) != seq + 1u) {  // not there yet: wait on the word alone, then read again
        if (!lds_wait_eq(&c->full[entry], seq + 1u, c)) return false;
        asm volatile("" ::: "memory");
        if (act0) chunk_from_lds<TYPE>(k0, item + off0, cols, rpi, r0, (unsigned)chunk);
        if (act1) chunk_from_lds<TYPE>(k1, item + off1, cols, rpi, r1, (unsigned)chunk);
      }
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 5u] = __builtin_readcyclecounter();
      if (act0) {
        const float d = k0.dot(xs.x, xs.s0, xs.s1, chunk & 7);
        const float y = wave_sum_lane63(xs.act ? d : 0.0f);
        if (lane == 63) c->partial[entry][ur0 * nspans + span] = y;
      }
      if (act1) {
        const float d = k1.dot(xs.x, xs.s0, xs.s1, chunk & 7);
        const float y = wave_sum_lane63(xs.act ? d : 0.0f);
        if (lane == 63) c->partial[entry][ur1 * nspans + span] = y;
      }
      asm volatile("" ::: "memory");
      if (lane == 0) lds_add(&c->done[entry], 1u);  // (after the partials in program order: LDS keeps a wave's order)
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 6u] = __builtin_readcyclecounter();
      if (!flush_epilogue()) return false;          // the item before this one, if its epilogue was this wave's
      if ((int)(seq % NCONS) == wave)
        pe = Pending{true, entry, NCONS * (seq / NCTRL + 1u), out, resid, li * rpi, rows, rpi, nspans, pair};
      if (ST && st && lane == 0 && seq < 128u) st[seq * 8u + 7u] = __builtin_readcyclecounter();
      ++seq;
    }
    return true;
  }

  __device__ __forceinline__ bool run_phase(const PhaseDesc& ph, const XSpan& xs, int cu, int ncu) {
    const int n_items = ph.n_items, cols = ph.cols, nspans = ph.nspans, nseg = ph.pair ? 1 : ph.nseg;
    const bool pair = ph.pair != 0;
    const float* resid = ph.resid;
    const int first = (int)((long)n_items * cu / ncu), last = (int)((long)n_items * (cu + 1) / ncu);
    int sb = 0;
    bool ok = true;
    for (int sg = 0; sg < nseg && ok; ++sg) {
      const int cnt = ph.items[sg];
      const int a = first > sb ? first - sb : 0, b = last - sb < cnt ? last - sb : cnt;
      if (a < b) {
        float* out = ph.seg[sg].out;
        const int rows = ph.seg[sg].rows, rpi = ph.rpi[sg];
        switch (ph.seg[sg].type) {
          case TL_Q4_K: ok = run_seg<TL_Q4_K>(xs, out, resid, rows, pair, cols, rpi, nspans, a, b); break;
          case TL_Q6_K: ok = run_seg<TL_Q6_K>(xs, out, resid, rows, pair, cols, rpi, nspans, a, b); break;
          case TL_Q8_0: ok = run_seg<TL_Q8_0>(xs, out, resid, rows, pair, cols, rpi, nspans, a, b); break;
          default: ok = run_seg<TL_Q8_K>(xs, out, resid, rows, pair, cols, rpi, nspans, a, b); break;
        }
      }
      sb += cnt;
    }
    return ok && flush_epilogue();
  }
};

// this wave's span of x (optionally RMS-normalised) into registers
__device__ __forceinline__ void load_span(XSpan& xs, const float* x, const float* norm_w, float nf, int cols, int span, int lane) {
  const int nchunks = cols >> 5;
  int chunk = span * 64 + lane;
  xs.act = chunk < nchunks;
  if (!xs.act) chunk = nchunks - 1;
  typedef const f32x4 __attribute__((address_space(1))) * g4;  // (pointers out of a descriptor: say that they are global)
  const g4 xp = (g4)(unsigned long long)(x + (size_t)chunk * 32);
  const g4 wp = (g4)(unsigned long long)((norm_w ? norm_w : x) + (size_t)chunk * 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x4 t = xp[i];
    if (norm_w) {
      const f32x4 g = wp[i];
      t[0] = t[0] * nf * g[0]; t[1] = t[1] * nf * g[1]; t[2] = t[2] * nf * g[2]; t[3] = t[3] * nf * g[3];
    }
    xs.x[4 * i] = t[0]; xs.x[4 * i + 1] = t[1]; xs.x[4 * i + 2] = t[2]; xs.x[4 * i + 3] = t[3];
  }
  // sum of 16 = ((q0 + q1) + (q2 + q3)) over the quads of four, as four staging threads fold them with two shuffles
  auto q = [&](int i) { return (xs.x[4 * i] + xs.x[4 * i + 1]) + (xs.x[4 * i + 2] + xs.x[4 * i + 3]); };
  xs.s0 = (q(0) + q(1)) + (q(2) + q(3));
  xs.s1 = (q(4) + q(5)) + (q(6) + q(7));
}

// One launch = phases [first, first + count) of `descs`, back to back. (Stage 1: no cross-CU hand-off between phases —
// every phase reads an x that was complete before the launch.)
// dbg (experiments): bit 0 consumers skip the arithmetic, bit 1 the loader publishes items without copying them
template <bool ST>
__global__ __launch_bounds__(NTHREADS) void stream_kernel(const PhaseDesc* __restrict__ descs, int first, int count,
                                                         unsigned* __restrict__ err, int dbg,
                                                         unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  Ctrl* c = reinterpret_cast<Ctrl*>(lds + RING_BYTES);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (unsigned i = threadIdx.x; i < sizeof(Ctrl) / 4; i += NTHREADS) reinterpret_cast<unsigned*>(c)[i] = 0u;
  __syncthreads();  // the only workgroup barrier: from here on the loader runs on its own
  const int cu = blockIdx.x, ncu = gridDim.x;
  bool ok = true;
  if (wave >= NCONS) {
    // the loaders go ahead of the consumers in issue arbitration: they are the youngest waves of the workgroup and would
    // otherwise get the slots the consumers leave (measured: 23 -> 20 us for a 66 MB phase)
    if (!(dbg & 4)) __builtin_amdgcn_s_setprio(3);
    const unsigned lid = (unsigned)(wave - NCONS);
    Loader<ST> L = {(unsigned)(unsigned long long)lds, c, lid, 0u, lid, 0u, 0u, 0ull, 0u, 0u, 0u, false, lane, (dbg & 2) != 0,
                    (dbg & 32) != 0, (blockIdx.x == 0 && stamps && lid == 0u) ? &c->stamps[0][0] : nullptr};
    for (int p = first; p < first + count && ok; ++p) ok = L.run_phase(descs[p], cu, ncu);
    L.drain();
  } else {
    const int cw = wave;
    Consumer<ST> C = {lds, c, 0u, 0u, cw, lane, (dbg & 1) != 0, (dbg & 16) != 0, (blockIdx.x == 0 && cw == 0 && stamps) ? &c->stamps[0][0] : nullptr, {}};
    for (int p = first; p < first + count && ok; ++p) {
      const PhaseDesc& ph = descs[p];
      XSpan xs;
      load_span(xs, ph.x, nullptr, 1.0f, ph.cols, cw % ph.nspans, lane);
      ok = C.run_phase(ph, xs, cu, ncu);
    }
  }
  if (!ok && lane == 0) atomicExch(err, 1u);
  if (ST && stamps && blockIdx.x == 0) {  // diagnostic run: every wave of the workgroup is through, then the stamps go out
    __syncthreads();
    for (unsigned i = threadIdx.x; i < 128u * 8u; i += NTHREADS) stamps[i] = (&c->stamps[0][0])[i];
  }
}

int g_attr_done[64] = {};

}  // namespace

// geometry of a segment: rows per item such that the consumers get at most two units each and an item stays under
// ITEM_MAX_BYTES of ring (pair: counts both matrices)
static int stream_plan(int type, int cols, int pair, int* rpi_out, int* nspans_out) {
  const int nchunks = cols / 32, nspans = (nchunks + 63) / 64;
  TL_REQUIRE(cols % 32 == 0 && nspans >= 1 && nspans <= NCONS, "stream: %d columns need %d spans (at most %d)", cols, nspans, NCONS);
  const int g = NCONS / nspans;
  auto ok = [&](int r) {
    if (rows_ring_bytes(type, cols, r) * (pair ? 2 : 1) > ITEM_MAX_BYTES) return false;
    for (int p = 0; p < 4; ++p)
      if ((r * plane_row_bytes(type, p, cols)) % 16 != 0) return false;
    return true;
  };
  int rpi = pair ? g : 2 * g;
  while (rpi >= 1 && !ok(rpi)) --rpi;
  TL_REQUIRE(rpi >= 1, "stream: no item geometry for %d columns of type %d (an item must fit %d bytes of ring in planes of "
             "whole 16-byte pieces)", cols, type, ITEM_MAX_BYTES);
  *rpi_out = rpi;
  *nspans_out = nspans;
  return 0;
}

// Measurement / parity hook (no counterpart in the reference): y = W x through the ring-loader engine, `nw` equally
// shaped matrices as `nw` phases of ONE launch (phase i: ys_dev + i * rows), timed with HIP events when avg_us_out != NULL.
extern "C" int tl_stream_matvec(const tl_qweight* const* ws, int nw, const float* x_dev, float* ys_dev, int reps,
                                float* avg_us_out, void* stream) {
  TL_REQUIRE(ws && nw > 0 && x_dev && ys_dev && reps > 0, "tl_stream_matvec: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TL_HIP(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const size_t lds = (size_t)RING_BYTES + sizeof(Ctrl);
  TL_REQUIRE(lds <= 160 * 1024, "tl_stream_matvec: %zu bytes of LDS", lds);
  if (dev < 64 && !g_attr_done[dev]) {
    TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    TL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    g_attr_done[dev] = 1;
  }
  PhaseDesc* h = new PhaseDesc[nw]();
  for (int i = 0; i < nw; ++i) {
    const tl_qweight* w = ws[i];
    PhaseDesc& d = h[i];
    int rc = stream_plan(w->type, w->cols, 0, &d.rpi[0], &d.nspans);
    if (rc != 0) {
      delete[] h;
      return rc;
    }
    d.seg[0] = SegDesc{{w->p0, w->p1, w->p2, w->p3}, ys_dev + (size_t)i * w->rows, w->rows, w->type};
    d.nseg = 1;
    d.cols = w->cols;
    d.items[0] = (w->rows + d.rpi[0] - 1) / d.rpi[0];
    d.n_items = d.items[0];
    d.x = x_dev;
  }
  PhaseDesc* ddev = nullptr;
  unsigned* err = nullptr;
  TL_HIP(hipMalloc(&ddev, sizeof(PhaseDesc) * nw));
  TL_HIP(hipMalloc(&err, 4));
  TL_HIP(hipMemcpy(ddev, h, sizeof(PhaseDesc) * nw, hipMemcpyHostToDevice));
  TL_HIP(hipMemset(err, 0, 4));
  delete[] h;
  hipEvent_t e0, e1;
  TL_HIP(hipEventCreate(&e0));
  TL_HIP(hipEventCreate(&e1));
  const int dbg = getenv("TL_STREAM_DBG") ? atoi(getenv("TL_STREAM_DBG")) : 0;
  unsigned long long* stamps = nullptr;
  const char* stamp_path = getenv("TL_STREAM_STAMPS");
  if (stamp_path) {
    TL_HIP(hipMalloc(&stamps, 128 * 8 * 8));
    TL_HIP(hipMemset(stamps, 0, 128 * 8 * 8));
  }
  stream_kernel<false><<<ncu, NTHREADS, lds, s>>>(ddev, 0, nw, err, dbg, nullptr);  // warm-up
  TL_HIP(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) stream_kernel<false><<<ncu, NTHREADS, lds, s>>>(ddev, 0, nw, err, dbg, nullptr);
  TL_HIP(hipEventRecord(e1, s));
  if (stamps) stream_kernel<true><<<ncu, NTHREADS, lds, s>>>(ddev, 0, nw, err, dbg, stamps);  // (the stamped build, untimed)
  TL_HIP(hipEventSynchronize(e1));
  TL_HIP(hipStreamSynchronize(s));
  TL_HIP(hipGetLastError());
  float ms = 0.f;
  TL_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (avg_us_out) *avg_us_out = ms * 1e3f / ((float)reps * nw);
  unsigned herr = 0;
  TL_HIP(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (stamps) {
    unsigned long long* hs = new unsigned long long[128 * 8];
    hipMemcpy(hs, stamps, 128 * 8 * 8, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(stamp_path, "w")) {
      for (int i = 0; i < 128 && hs[i * 8] != 0; ++i) {
        for (int k = 0; k < 8; ++k) fprintf(f, "%llu ", hs[i * 8 + k] ? hs[i * 8 + k] - hs[0] : 0ull);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    delete[] hs;
    hipFree(stamps);
  }
  hipFree(ddev);
  hipFree(err);
  TL_REQUIRE(herr == 0, "tl_stream_matvec: a wave of the engine gave up waiting (ring protocol error)");
  return 0;
}
