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

#ifdef __cplusplus
}
#endif
#endif
