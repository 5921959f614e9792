cd $GRAFT_REPO_ROOT
for d in ${DBGS:-3 0}; do
TL_STREAM_DBG=$d TL_STREAM_STAMPS=gpurun_out/stream_stamps_$d.txt timeout -k 10 100 python3 - <<PY
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from _pkg import load_package
pkg = load_package(); S = pkg.synth
from tinyllama_cpp_amd import engine
from tinyllama_cpp_amd.capi import check, lib
engine.init(0)
rng = np.random.default_rng(0)
rows, cols = 28672, 4096
blocks = S.random_blocks(S.Q4_K, rows * cols // 256, rng, w_std=0.02)
ws = [engine.QWeight(S.Q4_K, blocks, rows, cols) for _ in range(9)]
x = engine.DeviceBuffer.from_numpy(rng.standard_normal(cols).astype(np.float32))
ys = engine.DeviceBuffer(4 * rows * 9)
arr = (C.c_void_p * 9)(*[w.h for w in ws])
us = C.c_float()
check(lib().tl_stream_matvec(arr, 9, x.ptr, ys.ptr, 5, C.byref(us), None), "stream")
print("dbg $d us/matrix", us.value)
PY
echo "== stamps dbg $d: item: t_begin t_space_ok t_pub t_issue_end | cons0: t_poll_begin t_full_seen t_arrived t_item_end"
sed -n 20,32p gpurun_out/stream_stamps_$d.txt
echo "== loader fine: before-publish-loop-of-item after-fits(space ok) ... "
sed -n 84,96p gpurun_out/stream_stamps_$d.txt
done
