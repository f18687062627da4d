"""speedy_amd — MI355X-native implementation of google/speedy's hot path
(analysis -> tension -> speed -> pitch-synchronous overlap-add).

The product is the HIP library speedy_amd/lib/libspeedy_hip.so (C-ABI: include/speedy_hip.h and the
reference-compatible include/sonic2.h).  This package is the thin Python mirror used by tests and bench:
ctypes over the C-ABI, torch only for device memory / streams / torch.distributed.
There is no CPU fallback: without the library or a GPU, calls raise.
"""
from ._lib import lib, build, LIB_PATH  # noqa: F401
