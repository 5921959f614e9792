/* tlhip_debug.h -- measurement and diagnostic hooks of libtlhip.so. NOT part of the drop-in boundary (include/tlhip.h is
 * what a maintainer of the reference binds in place of cuda_kernels.h:104-643); none of these has a counterpart in the
 * reference. bench.py, tools/ and the tests use them through tinyllama.cpp_amd/capi.py (DEBUG_SIGNATURES). */
#ifndef TLHIP_DEBUG_H
#define TLHIP_DEBUG_H
#include "tlhip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* From now on every chain launch of the decode step (tl_gemv.hip, chain_kernel) writes 8 realtime stamps (100 MHz) per
 * workgroup into buf_dev, laid out [launch][task][256 workgroups][8] u64 in launch order, until cap_u64 words are used:
 * [0] workgroup entry, [1] first weight loads issued, [2] predecessor task seen complete, [3] x staged, [4] last loads
 * issued, [5] all stores acknowledged and the workgroup met, [6] arrival added. NULL switches the stamps off. Run the step
 * eagerly (TL_NO_GRAPH=1): a captured graph keeps the pointer it was captured with. */
int tl_chain_set_stamps(unsigned long long* buf_dev, size_t cap_u64);

/* Measurement hook (no counterpart in the reference): `iters` back-to-back tl_matvec_q launches cycling through
 * `nw` equally-shaped matrices (use more than the 256 MiB Infinity Cache holds), timed with HIP events on `stream`. */
int tl_bench_matvec(const tl_qweight* const* ws, int nw, const float* x_dev, float* y_dev, int iters, int warmup,
                    float* avg_us_out, void* stream);
/* Measurement hook (experiment, no counterpart in the reference): the tl_bench_matvec chain captured into a two-branch
 * graph in which a second stream pulls matrix i+1 into the Infinity Cache while the GEMV over matrix i runs
 * (prefetch_blocks workgroups of 256 threads; 0 = no prefetch branch, same graph otherwise). */
int tl_bench_matvec_prefetch(const tl_qweight* const* ws, int nw, const float* x_dev, float* y_dev, int reps,
                             int prefetch_blocks, float* avg_us_out);
/* Diagnostic hook (no counterpart in the reference): one tl_matvec_q-shaped launch of an instrumented build of the
 * GEMV kernel; stamps_dev[10 * grid * waves] u64 receives per-wave shader-clock stamps (see tl_gemv.hip). */
int tl_debug_gemv_stamps(const tl_qweight* w, const float* x_dev, float* y_dev, unsigned long long* stamps_dev,
                         int* grid_out, int* waves_out, void* stream);
/* Measurement hook (no counterpart in the reference): runs n_steps greedy decode steps EAGERLY (no graph) on the
 * model's stream with a HIP event after every kernel, and accumulates per-kernel-kind milliseconds and launch
 * counts into kind_ms_out[9] / kind_launches_out[9]: 0 embed, 1 norm+QKV GEMV, 2 RoPE+KV write, 3 attention,
 * 4 O GEMV+residual, 5 norm+gate/up GEMV+SwiGLU, 6 down GEMV+residual, 7 norm+lm_head GEMV, 8 sample+advance. */
int tl_model_profile_decode(tl_model* m, int first_token, int start_pos, int n_steps, float* kind_ms_out,
                            int* kind_launches_out);
/* Measurement hook (no counterpart in the reference): `iters` back-to-back launches of ONE kind of the decode step's
 * GEMV kernels (kinds as above: 1, 4, 5, 6, 7), cycling through the layers so that every launch streams from HBM,
 * timed with HIP events on the engine's stream. Leaves activations / logits in an unspecified state. kind 9: the
 * step's all-reduce of f32[hidden] of a linked tensor-parallel shard (a collective call: every rank makes it). */
int tl_model_bench_kernel(tl_model* m, int kind, int iters, int warmup, float* avg_us_out, double* bytes_per_launch_out);

/* Diagnostic hook (no counterpart in the reference): launches of the expanded-plane GEMM kernel (tl_qweight_expand) so far
 * in this process: lets a test prove which kernel formed a result. */
int tl_debug_gemm_px_launches(long long* count_out);

/* ---- libtlhip_tools.so (csrc/tl_stream.hip; built beside libtlhip.so, not part of it): a measured negative result kept
 * for the record -- the persistent ring-loader engine of round 2. */
/* Measurement / parity hook (no counterpart in the reference): y_i = W_i x through the ring-loader engine (tl_stream.hip),
 * the nw equally shaped matrices as nw phases of ONE persistent launch (ys_dev: [nw][rows]); avg_us_out (may be NULL) =
 * microseconds per matrix over `reps` launches. */
int tl_stream_matvec(const tl_qweight* const* ws, int nw, const float* x_dev, float* ys_dev, int reps, float* avg_us_out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
