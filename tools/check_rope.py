"""For a maintainer who HAS perception_models installed (it is not vendored in the reference and not available offline): everything
needed to diff this build's 2-D rotary embedding against upstream in one call.

    python tools/check_rope.py [out.npz]       # writes cos / sin tables, a probe q [1, heads, T, head_dim] and OUR rotation of it

`ovo_amd.encoders.vit.rope_tables(spec)` is the ONLY place the PE rotary embedding is defined here -- the oracle (oracle/vit.py:_rope)
and the HIP path (rope epilogue of the QKV GEMM, csrc/gemm_common.h:math4) both consume its output -- so one comparison covers both:
rotate `q` with upstream `core.vision_encoder.rope.Rope2D` for a 24 x 24 grid with a class token and compare with `q_rotated`.
What is ASSUMED of upstream (PE-Core-L14-336: head_dim 64, 24 x 24 patches, theta 10000):
  * half of a head's channels rotate with the patch ROW index, the other half with the COLUMN index (axial split, rows first);
  * within a half, frequency i is theta^(-i / (head_dim / 4)), i = 0 .. head_dim/4 - 1, shared by the adjacent channel pair (2i, 2i+1);
  * a pair rotates as (x0, x1) -> (x0 cos - x1 sin, x1 cos + x0 sin); positions are the integer patch indices; the class token is not rotated.
If upstream differs in any of these, descriptors of real PE checkpoints would differ while every test here stays green (the oracle
shares the assumption -- "parity unpinned" for this one function, DESIGN.md section 0 row c)."""
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
cos, sin = rope_tables(spec)
q = torch.randn(1, spec.heads, spec.tokens, spec.width // spec.heads, generator=torch.Generator().manual_seed(0))
out = sys.argv[1] if len(sys.argv) > 1 else "rope_pe_l14_336.npz"
np.savez(out, cos=cos.numpy(), sin=sin.numpy(), q=q.numpy(), q_rotated=_rope(q, cos, sin).numpy(), grid=spec.grid, theta=10000.0)
print(f"wrote {out}: cos/sin {tuple(cos.shape)}, probe q {tuple(q.shape)}; class-token row rotated by identity: {bool((cos[0] == 1).all() and (sin[0] == 0).all())}")
