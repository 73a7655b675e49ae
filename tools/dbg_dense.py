import os, sys
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
from ovo_amd.utils import clip_utils
KW = dict(vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=300_000, track_th=40, k_top_views=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
pipe = FramePipeline("cuda:0", **KW)
frames = synthetic_frames(8, "cuda:0", scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
for i, f in enumerate(frames):
    out = pipe.step(f, frames[i + 1:])
    torch.cuda.synchronize()
    n = out["n_points"]
    _, cls, conf = clip_utils.similarity(pipe.acc[:n], pipe.texts, cnt=pipe.cnt[:n], want_sim=False, want_argmax=True)
    bad = (cls != pipe.dense_cls[:n]).nonzero().reshape(-1)
    seg = pipe.ovo.last_point_seg
    print(i, "n", n, "touched", pipe.n_touched.tolist(), "parity", pipe._touch_parity, "mismatch", bad.numel(), bad[:6].tolist(),
          "cnt", pipe.cnt[:n][bad[:6]].tolist(), "seg", seg[bad[:6]].tolist() if bad.numel() else None, "rows", pipe.ovo.last_mask_rows)
