import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
pipe = FramePipeline(dev, extra_capacity=45 * 72000)
frames = synthetic_frames(40, dev)
worst = (0, None, -1)
for i, f in enumerate(frames[:38]):
    pipe.masks.frames = {f.index: f}
    fd = [f.index, f.rgb_lr, f.depth, f.c2w]
    pipe.slam.track_camera(fd); c2w = pipe.slam._c2w_host[f.index]; pipe.slam.map(fd, c2w)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    upd = pipe.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, (1.0, 1.0, 12)], pipe.slam.get_map(), c2w)
    torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
    if i > 3 and dt > worst[0]: worst = (dt, pr, i)
    pipe.slam.update_pcd_obj_ids(upd); pipe.ovo.compute_semantic_info()
print("worst track step", worst[2], round(worst[0] * 1e3, 2), "ms")
s = io.StringIO(); pstats.Stats(worst[1], stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2500])
