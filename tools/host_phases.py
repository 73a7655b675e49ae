"""Host-side wall clock of each phase of FramePipeline.step WITHOUT extra device syncs (is the frame host-bound?)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd import _lib as L
from ovo_amd.pipeline import FramePipeline, synthetic_frames
from ovo_amd.utils import clip_utils
dev = torch.device("cuda", 0)
N = 30
pipe = FramePipeline(dev, extra_capacity=(N + 8) * 72000)
frames = synthetic_frames(N + 4, dev)
for f in frames[:4]: pipe.step(f)
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
lib = L.load()
T0 = time.perf_counter()
for f in frames[4:]:
    t = time.perf_counter()
    pipe.masks.frames = {f.index: f}
    fd = [f.index, f.rgb_lr, f.depth, f.c2w]
    pipe.slam.track_camera(fd); c2w = pipe.slam._c2w_host[f.index]; pipe.slam.map(fd, c2w); t = tick("map", t)
    with torch.cuda.stream(pipe.sam_stream):
        pipe.sam_out = pipe.sam.forward(pipe.sam.preprocess(f.rgb.permute(2, 0, 1).contiguous()))
    t = tick("sam_launch", t)
    pipe.ovo.prefetch_image_features(f.rgb); t = tick("vit_launch", t)
    upd = pipe.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, (1.0, 1.0, 12)], pipe.slam.get_map(), c2w); t = tick("track(sync inside)", t)
    pipe.slam.update_pcd_obj_ids(upd); t = tick("writeback", t)
    pipe.ovo.compute_semantic_info(); t = tick("pool+fuse", t)
    nn = pipe.slam._n
    rows = torch.tensor(pipe.ovo.last_mask_rows, dtype=torch.int32).to(dev, non_blocking=True)
    L.check(lib.ovo_scatter_accum(L.ptr(pipe.ovo.last_point_seg), pipe.ovo.last_point_seg.shape[0], L.ptr(rows), rows.shape[0],
                                  L.ptr(pipe.ovo.last_clip_embeds), pipe.D, L.ptr(pipe.acc), L.ptr(pipe.cnt), L.stream())); t = tick("scatter", t)
    table = pipe.ovo.get_objs_clips(); t = tick("gather", t)
    clip_utils.similarity(table, pipe.texts, want_argmax=True); t = tick("query_inst", t)
    clip_utils.similarity(pipe.acc[:nn], pipe.texts, cnt=pipe.cnt[:nn], want_sim=False, want_argmax=True); t = tick("query_dense", t)
torch.cuda.synchronize()
total = time.perf_counter() - T0
print({k: round(1e3 * v / N, 3) for k, v in acc.items()}, "host sum ms", round(1e3 * sum(acc.values()) / N, 2), "wall ms/frame", round(1e3 * total / N, 2))
