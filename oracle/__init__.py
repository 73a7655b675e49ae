"""CPU oracle for the OVO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package
(prompt rule ③).  `ovo_amd/` never imports it; the product fails loudly when its HIP library is
missing instead of falling back to anything in here.

Pinning status
  * geometry / back-projection / tracking / fusion / similarity / TextRegion pooling / mask NMS:
    PINNED -- checked bit-exactly (integers) or to 1e-6 (floats) against tests/golden/*.npz, which
    `tools/gen_golden.py` produced by running the reference's own functions on CPU.
  * ViT / SAM2 forward passes: "parity unpinned" by the reference (its model dependencies are
    un-vendored).  `oracle/vit.py` and `oracle/hiera.py` restate the published
    architectures and are pinned only against HuggingFace transformers 5.15 (independent code, random
    weights) through tests/golden/hf_*.npz.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libovo_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ovo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma",
               "-o", _SO, src, "-lm"]
        subprocess.check_call(cmd)
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        i64, i32, f32, p = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p
        _lib.orc_frustum_ids.restype = i64
        _lib.orc_frustum_ids.argtypes = [p, i64, p, p, p]
        _lib.orc_project.restype = None
        _lib.orc_project.argtypes = [p, i64, i32, p, p, p]
        _lib.orc_match.restype = i64
        _lib.orc_match.argtypes = [p, i64, i32, p, p, p, i32, i32, f32, p, p]
        _lib.orc_backproject.restype = i64
        _lib.orc_backproject.argtypes = [p, p, p, i32, i32, i32, i32, p, p, p, p]
        _lib.orc_similarity.restype = None
        _lib.orc_similarity.argtypes = [p, i64, p, i32, i32, i32, f32, f32, p]
        _lib.orc_mask_intersections.restype = None
        _lib.orc_mask_intersections.argtypes = [p, i32, i64, p]
    return _lib
