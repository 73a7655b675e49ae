import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import synthetic as syn
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.sam_decoder import SPECS as DS, HipSamDecoder
from ovo_amd.entities.sam_amg import HipSam2AutomaticMaskGenerator
dev = "cuda"
enc = HipHiera(HS["hiera_b+"], None, device=dev)
dec = HipSamDecoder(DS["sam2"], None, device=dev)
amg = HipSam2AutomaticMaskGenerator(enc, dec, points_per_side=16, pred_iou_thresh=0.5, stability_score_thresh=0.5)
img = torch.from_numpy(syn.render_rgb(480, 640, 1)).to(dev)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = amg.generate_launch(img)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = amg.generate_finish(h)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"iter {i}: launch {1e3*(t1-t0):.2f} ms (host), gpu done +{1e3*(t2-t1):.2f}, finish {1e3*(t3-t2):.2f} | device allocs {st['num_device_alloc']} frees {st['num_device_free']} retries {st['num_alloc_retries']}")
