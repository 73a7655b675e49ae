"""Per-shape table of the GEMM / attention launches of ONE batched encoder forward (the library's hipEvent profiler, streams folded):
python tools/enc_table.py {vit|sam} B"""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")    # this tool flips OVO_* knobs between launches
import ctypes as C, os, sys, tempfile
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.vit import SPECS as VS, HipViT
which, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
if which == "vit":
    enc = HipViT(VS["PE-Core-L14-336"], None, dev, 0); x = torch.randn(2 * B, 3, 336, 336, device=dev); fn = lambda: enc.forward(x, tokens=True)
else:
    enc = HipHiera(HS[os.environ.get("SAM", "hiera_b+")], None, dev, 0); x = torch.randn(B, 3, 1024, 1024, device=dev); fn = lambda: enc.forward(x)
for _ in range(3): fn()
torch.cuda.synchronize()
dump = tempfile.mktemp()
os.environ["OVO_PROF_DUMP"] = dump
lib = L.load()
REP = 5
L.check(lib.ovo_profile_start())
for _ in range(REP): fn()
ms, work, n = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_int64 * 9)()
L.check(lib.ovo_profile_stop(ms, work, n, 9))
rows = defaultdict(lambda: [0, 0.0, 0.0])
for line in open(dump):
    k, a, b, c, w, t = line.split()[:6]
    e = rows[(int(k), int(a), int(b), int(c))]
    e[0] += 1; e[1] += float(t); e[2] += float(w)
names = {0: "256x128", 3: "256x256", 4: "128x128", 5: "128x64", 6: "64x128", 7: "64x64", 8: "stream", 1: "attn", 2: "track"}
tot = 0.0
print("%-8s %-26s %7s %9s %9s %8s" % ("kind", "shape", "n/fwd", "avg us", "ms/frame", "TF"))
for (k, a, b, c), (cnt, t, w) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tot += t / REP / B
    print("%-8s %-26s %7.1f %9.1f %9.3f %8.0f" % (names.get(k, k), str((a, b, c)), cnt / REP, 1e3 * t / cnt, t / REP / B, w / t / 1e9))
print("total GEMM + attention ms/frame", tot)
