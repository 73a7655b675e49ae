"""What a rank pays per keyframe it does NOT own in a multi-GPU round (FramePipeline.step_round): the replicated, order-dependent part --
map update, tracking + instance decisions, plan, store + fuse of gathered descriptors, dense scatter / query of its rows -- on an
otherwise idle GPU (no encoders).  N x this is the serial term of a round.  Keyframes are processed the way step_round does it: a round
of R keyframes is QUEUED (ovo_map_step + ovo_track_step per keyframe, nothing read back), then finished in order.
usage: python tools/replicated_cost.py [frames] [round]        (round = 1: one keyframe at a time; HOST=1: host-decision path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
from ovo_amd.pipeline import FramePipeline, synthetic_frames
from ovo_amd.utils import geometry_utils as G
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pipe = FramePipeline(dev, sam_card=None, extra_capacity=(N + 16) * 72000)
pipe.prefetch = False
if os.environ.get("HOST"):
    pipe.ovo.config["host_decisions"] = True
frames = synthetic_frames(N + R, dev)
lib = L.load()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
ratio = (1.0, 1.0, pipe.crop_edge)
def round_(group, timed):
    t = time.perf_counter()
    pipe.masks.frames = {f.index: f for f in group}
    native = all(pipe.ovo._native_ok(f.masks) for f in group)
    if native:
        G.prepare_frame_cameras([(f.depth, f.c2w) for f in group], pipe.slam._K_host)
        if timed: t = tick("cameras (batched)", t)
        pend = []
        for f in group:
            fd = [f.index, f.rgb_lr, f.depth, f.c2w]
            pipe.slam.track_camera(fd); c2w = pipe.slam._c2w_host[f.index]; pipe.slam.map_launch(fd, c2w)
            pend.append(pipe.ovo.detect_and_track_launch([f.index, f.rgb, f.depth, ratio], pipe.slam, c2w))
        if timed: t = tick("launch map + track chains", t)
    for k, f in enumerate(group):
        if native:
            pipe.ovo.detect_and_track_finish(pend[k])
            if timed: t = tick("wait + bookkeeping", t)
        else:
            fd = [f.index, f.rgb_lr, f.depth, f.c2w]
            pipe.slam.track_camera(fd); c2w = pipe.slam._c2w_host[f.index]; pipe.slam.map(fd, c2w)
            if timed: t = tick("map (1 sync)", t)
            upd = pipe.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, ratio], pipe.slam.get_map(), c2w)
            if upd is not None: pipe.slam.update_pcd_obj_ids(upd)
            if timed: t = tick("track (1 sync)", t)
        plan = pipe.ovo._plan_semantic_info() if len(pipe.ovo.keyframes_queue) > 0 else None
        if timed: t = tick("plan", t)
        if plan is None: continue
        d = torch.zeros((len(plan["matched_ins_ids"]), pipe.D), dtype=torch.float32, device=dev); d[:, 0] = 1.0     # stands in for the gathered descriptors
        pipe.ovo._apply_semantic_plan(plan, d)
        if timed: t = tick("store+fuse", t)
        point_seg, mask_rows = pipe.ovo.last_point_seg, pipe.ovo.last_mask_rows
        rows = torch.tensor(mask_rows, dtype=torch.int32).to(dev, non_blocking=True)
        k2 = pipe._touch_parity; pipe._touch_parity ^= 1
        touched, n_cur, n_nxt = L.ptr(pipe.touched), pipe.n_touched[k2:].data_ptr(), pipe.n_touched[k2 ^ 1:].data_ptr()
        L.check(lib.ovo_scatter_accum_touched(L.ptr(point_seg), point_seg.shape[0], L.ptr(rows), rows.shape[0], L.ptr(d), pipe.D, L.ptr(pipe.acc), L.ptr(pipe.cnt),
                                              touched, n_cur, n_nxt, 0, 1, pipe.SHARD_BLOCK, L.stream()))
        L.check(lib.ovo_similarity_rows(L.ptr(pipe.acc), 0, touched, n_cur, min(point_seg.shape[0], pipe.rows_local), pipe.D, L.ptr(pipe.texts), pipe.texts.shape[0],
                                        L.ptr(pipe.cnt), 0, 0.0, 0.0, 0.0, L.ptr(pipe.dense_cls), L.ptr(pipe.dense_conf), L.stream()))
        if timed: t = tick("dense scatter+query (launch)", t)
round_(frames[:R], False)
torch.cuda.synchronize()
T0 = time.perf_counter()
for s in range(R, R + N, R): round_(frames[s:min(s + R, R + N)], True)
torch.cuda.synchronize()
total = time.perf_counter() - T0
print(f"round of {R} keyframes, {'host' if os.environ.get('HOST') else 'device'} decisions")
print({k: round(1e3 * v / N, 3) for k, v in acc.items()})
print(f"replicated cost per keyframe: {1e3 * total / N:.3f} ms wall (host + its waits, idle GPU)")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    more = synthetic_frames(N, dev, start=N + R)
    pr = cProfile.Profile(); pr.enable()
    for s in range(0, N, R): round_(more[s:s + R], False)
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
