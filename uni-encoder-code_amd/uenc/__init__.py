"""uenc — MI355X-native (gfx950) implementation of the unified-encoder hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd bookkeeping, torch.distributed
over RCCL).  Arithmetic: hand-written HIP kernels in `libuenc_hip.so`, reached through the C ABI
declared in `include/uenc.h` (ctypes, raw device pointers).  Importing `uenc.capi` raises when the
library has not been built: there is no CPU or eager-PyTorch fallback for the kernels.
"""
__version__ = "0.1.0"
