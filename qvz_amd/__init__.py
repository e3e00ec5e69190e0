"""qvz_amd: MI355X-native (gfx950) implementation of qvz's encode hot path.

The product is the native code under qvz_amd/csrc (hand-written HIP kernels + host C) behind the C-ABI of
include/qvz_hip.h; this package is the Python plumbing used by the tests and bench.py (device memory, streams,
torch.distributed). Importing kernels requires the built library: there is no CPU fallback.
"""
__all__ = ["kernels"]
