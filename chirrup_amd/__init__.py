"""chirrup_amd -- MI355X (gfx950) native RWKV-7 decode hot path for leonsama/chirrup.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/chirrup_amd.h)
  lib.py     ctypes loader (no CPU fallback: a missing library raises)
  ops.py     operator layer with the reference's op names / schemas (boundary B1)
  rwkv7.py   host-side model mirror of Albatross/rwkv7.py's RWKV_x070 (boundary B3)
"""
from .lib import ChirrupAmdError, LIB_PATH, build, load  # noqa: F401

__all__ = ["ChirrupAmdError", "LIB_PATH", "build", "load"]
