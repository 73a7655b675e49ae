"""SAM2 automatic mask generator at the reference's settings (points_per_side 16 -> 256 clicks, 640x480 frame): time per frame
of encoder / decoder / post-processing.  Diagnosis tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import synthetic as syn
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.sam_decoder import SPECS as DS, HipSamDecoder
from ovo_amd.entities.sam_amg import HipSam2AutomaticMaskGenerator
dev = "cuda"
pps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
enc = HipHiera(HS[os.environ.get("ENC", "hiera_b+")], None, device=dev)
dec = HipSamDecoder(DS["sam2"], None, device=dev)
amg = HipSam2AutomaticMaskGenerator(enc, dec, points_per_side=pps, pred_iou_thresh=0.5, stability_score_thresh=0.5)
img = torch.from_numpy(syn.render_rgb(480, 640, 1)).to(dev)
def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    print(f"{name:30s} {1e3 * (time.perf_counter() - t0) / n:8.2f} ms")
    return out
emb = timed("encoder", lambda: enc.encode_frame(img.permute(2, 0, 1).contiguous()), n=1 if os.environ.get("DEC_ONLY") else 5)
amg._set_grid()
f0, f1 = emb["high_res_feats"]
timed(f"decoder ({pps * pps} clicks)", lambda: dec.forward(emb["image_embed"][0], f1[0], f0[0]))
if os.environ.get("DEC_ONLY"): sys.exit(0)      # rocprofv3 of the decoder alone (1 encoder + 6 decoder forwards)
r = timed("generate_device (all)", lambda: amg.generate_device(img))
print("kept", r["masks"].shape[0], "of", pps * pps * 3, "| peak memory GB", torch.cuda.max_memory_allocated() / 2**30)
if os.environ.get("OVO_PROF_DUMP"):
    import ctypes as C, collections
    from ovo_amd import _lib as L
    lib = L.load(); open(os.environ["OVO_PROF_DUMP"], "w").close()
    L.check(lib.ovo_profile_start()); dec.forward(emb["image_embed"][0], f1[0], f0[0])
    ms, work, n = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_int64 * 9)()
    L.check(lib.ovo_profile_stop(ms, work, n, 9))
    agg = collections.OrderedDict()
    for line in open(os.environ["OVO_PROF_DUMP"]):
        k, a, b, c, w, t = line.split()[:6]; e = agg.setdefault((int(k), int(a), int(b), int(c)), [0, 0.0, 0.0]); e[0] += 1; e[1] += float(t); e[2] += float(w)
    for (k, a, b, c), (cnt, t, w) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"kind {k} shape {(a, b, c)!s:26s} x{cnt:3d} {1e3 * t / cnt:9.1f} us  total {t:7.3f} ms  {w / t / 1e9:7.0f} TF")
    print("gemm+attn ms:", sum(ms))
# ---- where generate_device's time goes: launch half (encoder + decoder + statistics, device) vs finish half (host filters,
# box NMS, binarise), at the candidate counts the thresholds let through
import numpy as np
for thr in ((0.5, 0.5), (0.0, 0.0)):
    amg.pred_iou_thresh, amg.stability_score_thresh = thr
    t = {}
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = amg.generate_launch(img)
        t1 = time.perf_counter()
        h["done"].synchronize(); t2 = time.perf_counter()
        r = amg.generate_finish(h)
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        if rep:
            for k, v in (("launch (host)", t1 - t0), ("wait for device", t2 - t1), ("finish (host: filter + box NMS + binarise launch)", t3 - t2), ("binarise (device tail)", t4 - t3)):
                t[k] = t.get(k, 0.0) + v / 5
    print(f"thresholds {thr}: kept {r['masks'].shape[0]:4d}  " + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in t.items()))
