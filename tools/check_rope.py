"""For a maintainer who HAS perception_models installed (it is not vendored in the reference and not available offline): everything
needed to diff this build's 2-D rotary embedding against upstream in one call.

    python tools/check_rope.py [out.npz]

writes, for EACH of the four convention combinations (cls_offset in {1, 0} x axis_order in {"xy", "yx"}), the cos / sin tables and OUR
rotation of one probe q [1, heads, T, head_dim] (keys `cos_<off>_<order>`, `sin_...`, `q_rotated_...`), plus `q` and `default` (the
combination ViTSpec ships with: cls_offset 1, axis_order "xy").  Rotate `q` with upstream `core.vision_encoder.rope.Rope2D` for a
24 x 24 grid with a class token and see which `q_rotated_*` it equals; if it is not the default, set ViTSpec.rope_cls_offset /
rope_axis_order of the PE cards in ovo_amd/encoders/vit.py accordingly (the oracle follows the spec: oracle/vit.py:rope_for).

`ovo_amd.encoders.vit.rope_tables` is the only place the product defines the embedding (the HIP path consumes its output in the QKV GEMM's
rope epilogue, csrc/gemm_common.h:math4); `oracle/vit.py:rope2d_tables` restates it independently for the tests.
What is ASSUMED of upstream (PE-Core-L14-336: head_dim 64, 24 x 24 patches, theta 10000), beyond the two switches:
  * half of a head's channels rotate with one grid axis, the other half with the other (axial split);
  * within a half, frequency i is theta^(-i / (head_dim / 4)), i = 0 .. head_dim/4 - 1, shared by the adjacent channel pair (2i, 2i+1);
  * a pair rotates as (x0, x1) -> (x0 cos - x1 sin, x1 cos + x0 sin); the class token is not rotated.
If upstream differs in any of these, descriptors of real PE checkpoints would differ while every test here stays green ("parity
unpinned" for this one function, DESIGN.md section 0 row c)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ovo_amd.encoders.vit import SPECS, rope_tables


def _rope(x, cos, sin):
    """(x0, x1) -> (x0 c - x1 s, x1 c + x0 s) on adjacent channel pairs; x [B, H, T, hd], cos / sin [T, hd]."""
    x0, x1 = x[..., 0::2], x[..., 1::2]
    return x * cos + torch.stack([-x1, x0], dim=-1).flatten(-2) * sin


spec = SPECS["PE-Core-L14-336"]
q = torch.randn(1, spec.heads, spec.tokens, spec.width // spec.heads, generator=torch.Generator().manual_seed(0))
out = sys.argv[1] if len(sys.argv) > 1 else "rope_pe_l14_336.npz"
arrays = {"q": q.numpy(), "grid": spec.grid, "theta": 10000.0, "default": f"{spec.rope_cls_offset}_{spec.rope_axis_order}"}
for off in (1, 0):
    for order in ("xy", "yx"):
        cos, sin = rope_tables(spec, cls_offset=off, axis_order=order)
        assert bool((cos[0] == 1).all() and (sin[0] == 0).all())          # the class-token row is the identity rotation
        arrays[f"cos_{off}_{order}"], arrays[f"sin_{off}_{order}"] = cos.numpy(), sin.numpy()
        arrays[f"q_rotated_{off}_{order}"] = _rope(q, cos, sin).numpy()
np.savez(out, **arrays)
print(f"wrote {out}: 4 convention combinations, tables {tuple(cos.shape)}, probe q {tuple(q.shape)}; default = {arrays['default']}")
