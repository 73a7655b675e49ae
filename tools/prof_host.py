"""cProfile of the host side of FramePipeline.step_round (one process): which Python functions the per-keyframe work -- replicated on
every rank in a multi-GPU round -- spends its host time in.  usage: python tools/prof_host.py [rounds]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pipe = FramePipeline(dev, extra_capacity=(N + 16) * 72000, encoder_batch=8)
frames = synthetic_frames(N + 8, dev)
pipe.prime(*frames[0].rgb.shape[:2])
for i in range(8):
    pipe.step_round(frames[i:i + 1], frames[i + 1:8])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(8, 8 + N):
    pipe.step_round(frames[i:i + 1], frames[i + 1:8 + N])
pr.disable()
torch.cuda.synchronize()
print("ms/step (profiled)", (time.perf_counter() - t0) / N * 1e3)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
