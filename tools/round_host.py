import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from ovo_amd.pipeline import Frame, FramePipeline, synthetic_frames
N = int(sys.argv[1]); R = 24; EB = 12
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
base = synthetic_frames(48, dev)
need = (R * 2 + 2 * EB) * N
pool = base * (need // len(base) + 1)
stream = [Frame(100_000 + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:need])]
pipe = FramePipeline(dev, sam_card=None if os.environ.get("NOSAM") else "hiera_b+", extra_capacity=(need + 2) * 72_000, encoder_batch=EB, emulate=(0, N))
if os.environ.get("NOVIT"): pipe.prefetch = False
pipe.prime(*stream[0].rgb.shape[:2])
pos = 0
def run(rounds):
    global pos
    end = pos + rounds * N
    for _ in range(rounds):
        g = stream[pos:pos + N]; pos += N
        pipe.step_round(g, stream[pos:end])
run(EB); torch.cuda.synchronize()
t0 = time.perf_counter(); run(R); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"world {N} NOSAM={os.environ.get('NOSAM')}: host {1e3*(t1-t0)/R:.3f} ms/round, total {1e3*(t2-t0)/R:.3f} ms/round")
