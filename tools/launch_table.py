"""Per-shape GEMM / attention time of one frame: OVO_PROF_DUMP lines (kind M N K work ms) -> table.  Diagnosis tool."""
import collections, os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
dump = "/tmp/ovo_prof_dump.txt"
if os.path.exists(dump): os.remove(dump)
os.environ["OVO_PROF_DUMP"] = dump
from ovo_amd import _lib as L
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
frames_n = 4
pipe = FramePipeline(dev, n_map=1_000_000, extra_capacity=2_000_000)
frames = synthetic_frames(3 + frames_n, dev)
for f in frames[:3]: pipe.step(f)
torch.cuda.synchronize()
lib = L.load()
L.check(lib.ovo_profile_start())
for f in frames[3:]: pipe.step(f)
ms, work, n = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_int64 * 9)()
L.check(lib.ovo_profile_stop(ms, work, n, 9))
agg = collections.OrderedDict()
for line in open(dump):
    k, a, b, c, w, t = line.split()[:6]
    key = (int(k), int(a), int(b), int(c))
    e = agg.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += float(t); e[2] += float(w)
tiles = {1: "attn", 4: "128x128", 5: "128x64", 6: "64x128", 7: "64x64"}
tot = 0.0
print("%-8s %-24s %6s %9s %9s %8s" % ("kind", "shape", "n/frm", "avg us", "ms/frame", "TF"))
for (k, a, b, c), (cnt, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if k not in tiles: continue
    tot += t / frames_n
    print("%-8s %-24s %6.1f %9.1f %9.3f %8.0f" % (tiles[k], (a, b, c), cnt / frames_n, 1e3 * t / cnt, t / frames_n, w / t / 1e9))
print("total ms/frame", tot)
