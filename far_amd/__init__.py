"""far_amd: MI355X-native (gfx950) implementation of FAR's pose-estimation hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the hot operators are
hand-written HIP kernels in far_amd/lib/libfar_hip.so behind the C ABI of include/far_hip.h.
"""
__version__ = '0.1.0'
