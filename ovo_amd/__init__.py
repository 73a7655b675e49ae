"""ovo_amd -- MI355X-native implementation of OVO's per-frame open-vocabulary feature path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the path
runs in libovo_hip.so (hand-written HIP for gfx950, C ABI in include/ovo_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
