"""commet_amd — MI355X-native index_and_search hot path of Commet.

Only what the path needs lives here:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI library + the C++ host tool
  lib.py     ctypes binding of include/commet_hip.h (fails loudly without the built .so)
  api.py     thin Python mirror of the C ABI (Context / ReadSet)
  synth.py   the synthetic read sets of SURVEY §8d
  build.py   hipcc / g++ build recipes used by __graft_entry__.build()
"""
from .api import Context, ReadSet, CommetError, device_count, device_cache_trim, device_cache_bytes, device_pooled_bytes, device_alloc_stats  # noqa: F401

__all__ = ["Context", "ReadSet", "CommetError", "device_count", "device_cache_trim", "device_cache_bytes", "device_pooled_bytes", "device_alloc_stats"]
