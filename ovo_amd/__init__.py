"""ovo_amd -- MI355X-native implementation of OVO's per-frame open-vocabulary feature path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the path
runs in libovo_hip.so (hand-written HIP for gfx950, C ABI in include/ovo_hip.h).  See DESIGN.md.
"""
import os as _os

import torch as _torch

__version__ = "0.1.0"

# The only CPU torch ops on the path are 8x4 / 4x4 set-up products per frame.  With a many-core host torch's
# OpenMP/MKL pools (128+ threads) spin-wait after every tiny parallel region and starve the HIP runtime's helper
# threads: measured 78 ms -> 26 ms per keyframe on a 128-core box just from this setting.  Override with OVO_CPU_THREADS.
_torch.set_num_threads(int(_os.environ.get("OVO_CPU_THREADS", "1")))
