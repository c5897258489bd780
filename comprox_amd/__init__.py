"""comprox_amd — MI355X-native block codec behind comprox's data_block_t entry points.

The product is libcrgpu.so (hand-written HIP for gfx950 + a plain C ABI, include/crgpu.h).
This package only holds its sources (csrc/), the build recipe and a thin ctypes mirror of the
ABI used by the tests and bench.py. There is no CPU fallback: without the compiled library or
without a gfx950 device every call raises.
"""
from . import api  # noqa: F401
from .api import CrGpu, CrDict, CrMulti, CrGpuError, load_library, CODEC_ROP, CODEC_ROX, CODEC_ROLZ, bound  # noqa: F401
