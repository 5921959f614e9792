"""tinyllama.cpp_amd — MI355X (gfx950) backend for tinyllama.cpp's quantized transformer forward pass.

Import through `_pkg.load_package()` (alias `tinyllama_cpp_amd`). Sub-modules:
  capi   ctypes binding of the C-ABI (include/tlhip.h) -> csrc/libtlhip.so; raises if it is missing
  engine thin host-side objects over the C-ABI (QWeight, Model)
  tp     tensor-parallel sharding of the model's tensors (host-side numpy only)
  synth  synthetic random-init models / GGUF writer (host-side numpy only)
"""
from . import synth  # noqa: F401
