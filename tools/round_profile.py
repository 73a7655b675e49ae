"""cProfile of the host side of an emulated N-rank round (FramePipeline(emulate=(0, N))): where the Python time of a round goes, and how
much of it is waiting (PinnedRing.wait = the device has not finished a chain yet).  usage: python tools/round_profile.py [N] [rounds]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ovo_amd.pipeline import Frame, FramePipeline, synthetic_frames
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
EB = int(os.environ.get("EB", "12"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
base = synthetic_frames(48, dev)
need = (R + 2 * EB) * N
pool = base * (need // len(base) + 1)
stream = [Frame(100_000 + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:need])]
pipe = FramePipeline(dev, sam_card=None if os.environ.get("NOSAM") else "hiera_b+", extra_capacity=(need + 2) * 72_000, encoder_batch=EB, emulate=(0, N))
pipe.prime(*stream[0].rgb.shape[:2])
pos = 0
def run(rounds):
    global pos
    end = pos + rounds * N
    for _ in range(rounds):
        g = stream[pos:pos + N]; pos += N
        pipe.step_round(g, stream[pos:end])
run(EB)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
run(R)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"world {N}: {1e3 * (t1 - t0) / R:.3f} ms host per round, +{1e3 * (t2 - t1):.2f} ms drain at the end ({R} rounds)")
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
