"""Time of the fused tracking pass (ovo_track_project) on the bench workload's map.  Diagnosis tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
pipe = FramePipeline(dev, n_map=int(os.environ.get("N_MAP", 1_000_000)), extra_capacity=2_000_000)
frames = synthetic_frames(8, dev)
for f in frames[:4]: pipe.step(f)
pipe.join() if hasattr(pipe, "join") else None
torch.cuda.synchronize()
from ovo_amd import _lib as L
import ctypes as C
lib = L.load()
L.check(lib.ovo_profile_start())
for f in frames[4:]: pipe.step(f)
if hasattr(pipe, "join"): pipe.join()
ms, work, n = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_int64 * 9)()
L.check(lib.ovo_profile_stop(ms, work, n, 9))
print(f"track_project: {n[2]} launches, {1e3 * ms[2] / max(n[2], 1):.1f} us each, {work[2] / ms[2] / 1e6:.1f} GB/s algorithmic, points {pipe.slam._n if hasattr(pipe, 'slam') else '?'}")
