"""What a rank pays per keyframe it does NOT own in a multi-GPU round (FramePipeline.step_round): the replicated, order-dependent part --
map update, tracking + instance decisions, plan, store + fuse of gathered descriptors, dense scatter / query of its rows -- on an
otherwise idle GPU (no encoders).  N x this is the serial term of a round.  usage: python tools/replicated_cost.py [frames]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd import _lib as L
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = FramePipeline(dev, sam_card=None, extra_capacity=(N + 8) * 72000)
pipe.prefetch = False
frames = synthetic_frames(N + 4, dev)
lib = L.load()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
def one(f, timed):
    t = time.perf_counter()
    pipe.masks.frames = {f.index: f}
    fd = [f.index, f.rgb_lr, f.depth, f.c2w]
    pipe.slam.track_camera(fd); c2w = pipe.slam._c2w_host[f.index]; pipe.slam.map(fd, c2w)
    if timed: t = tick("map (1 sync)", t)
    upd = pipe.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, (1.0, 1.0, pipe.crop_edge)], pipe.slam.get_map(), c2w)
    if upd is not None: pipe.slam.update_pcd_obj_ids(upd)
    if timed: t = tick("track (1 sync)", t)
    plan = pipe.ovo._plan_semantic_info() if len(pipe.ovo.keyframes_queue) > 0 else None
    if timed: t = tick("plan", t)
    if plan is None: return
    d = torch.zeros((len(plan["matched_ins_ids"]), pipe.D), dtype=torch.float32, device=dev); d[:, 0] = 1.0     # stands in for the gathered descriptors
    pipe.ovo._apply_semantic_plan(plan, d)
    if timed: t = tick("store+fuse", t)
    point_seg, mask_rows = pipe.ovo.last_point_seg, pipe.ovo.last_mask_rows
    rows = torch.tensor(mask_rows, dtype=torch.int32).to(dev, non_blocking=True)
    k = pipe._touch_parity; pipe._touch_parity ^= 1
    touched, n_cur, n_nxt = L.ptr(pipe.touched), pipe.n_touched[k:].data_ptr(), pipe.n_touched[k ^ 1:].data_ptr()
    L.check(lib.ovo_scatter_accum_touched(L.ptr(point_seg), point_seg.shape[0], L.ptr(rows), rows.shape[0], L.ptr(d), pipe.D, L.ptr(pipe.acc), L.ptr(pipe.cnt),
                                          touched, n_cur, n_nxt, 0, 1, pipe.SHARD_BLOCK, L.stream()))
    L.check(lib.ovo_similarity_rows(L.ptr(pipe.acc), 0, touched, n_cur, min(point_seg.shape[0], pipe.rows_local), pipe.D, L.ptr(pipe.texts), pipe.texts.shape[0],
                                    L.ptr(pipe.cnt), 0, 0.0, 0.0, 0.0, L.ptr(pipe.dense_cls), L.ptr(pipe.dense_conf), L.stream()))
    if timed: t = tick("dense scatter+query (launch)", t)
for f in frames[:4]: one(f, False)
torch.cuda.synchronize()
T0 = time.perf_counter()
for f in frames[4:4 + N]: one(f, True)
torch.cuda.synchronize()
total = time.perf_counter() - T0
print({k: round(1e3 * v / N, 3) for k, v in acc.items()})
print(f"replicated cost per keyframe: {1e3 * total / N:.3f} ms wall (host + its syncs, idle GPU)")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for f in frames[4:4 + N]:
        f2 = type(f)(f.index + 1000, f.rgb, f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks)
        one(f2, False)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
