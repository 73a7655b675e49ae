"""Side streams of the pipeline.  A stream may be restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask) through an
environment variable -- a measurement knob: OVO_SAM_CUS / OVO_VIT_CUS = "xcd:0,1,2" (CU-mask bits i with i % 8 in the set: on MI300-class
parts consecutive mask bits go round the 8 XCDs) or "range:0-127" (consecutive mask bits)."""
from __future__ import annotations

import ctypes as C
import os

import torch

_hip = None


def _mask_words(spec: str, n_cus: int = 256):
    kind, _, arg = spec.partition(":")
    bits = [False] * n_cus
    if kind == "xcd":
        keep = {int(v) for v in arg.split(",")}
        bits = [(i % 8) in keep for i in range(n_cus)]
    elif kind == "range":
        lo, hi = (int(v) for v in arg.split("-"))
        bits = [lo <= i <= hi for i in range(n_cus)]
    else:
        raise ValueError(f"CU spec '{spec}': expected xcd:<list> or range:<lo>-<hi>")
    words = [0] * ((n_cus + 31) // 32)
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    return words


def side_stream(device, env: str, priority: int = 0) -> torch.cuda.Stream:
    """A new stream on `device`; CU-masked when the environment variable `env` holds a CU spec."""
    spec = os.environ.get(env)
    if not spec:
        return torch.cuda.Stream(device=device, priority=priority)
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    words = _mask_words(spec)
    arr = (C.c_uint32 * len(words))(*words)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = _hip.hipExtStreamCreateWithCUMask(C.byref(handle), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc}) for {env}={spec}")
    return torch.cuda.ExternalStream(handle.value, device=device)
