"""BASELINE.json configs[1]: CLIP ViT-B/16 per-frame forward on one MI355X, bf16 (SURVEY.md section 8d "Config 2"), and the crop embed
types' keyframe (embed_type: vanilla = one forward of N mask crops; the fused types 1 + 2 N, clip_generator.py:139-150).
Per batch size: images/s of preprocess (open_clip's kept transforms on a 640x480 frame) + forward (pooled, projected descriptor), and the
fraction of the dense bf16 MFMA peak.  usage: python tools/vit_bench.py [card ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders.vit import SPECS, HipViT
PEAK = 2500e12
dev = torch.device("cuda", 0)
cards = sys.argv[1:] or ["ViT-B-16-qg", "ViT-L-14-qg"]
frame = (torch.rand(3, 480, 640, device=dev) * 255).to(torch.uint8)
for card in cards:
    spec = SPECS[card]
    vit = HipViT(spec, None, dev, 0)
    gflop = spec.flops_per_image() / 1e9
    for b in (1, 2, 8, 65, 129):                         # 65 = embed_type vanilla with 65 masks' crops / 1 + 2 x 32; 129 = 1 + 2 x 64
        crops = frame[None].expand(b, -1, -1, -1).contiguous()
        batch = torch.empty((b, 3, spec.image_size, spec.image_size), dtype=torch.float32, device=dev)
        out = torch.empty((b, spec.out_dim), dtype=torch.float32, device=dev)
        def step():
            vit.preprocess_clip(crops, scale=1 / 255.0, out=batch)
            vit.forward(batch, out=out)
        for _ in range(3): step()
        torch.cuda.synchronize()
        reps = max(5, 200 // b)
        t0 = time.perf_counter()
        for _ in range(reps): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): vit.forward(batch, out=out)
        e1.record(); torch.cuda.synchronize()
        fwd = e0.elapsed_time(e1) / reps * 1e-3
        print(json.dumps({"card": card, "batch": b, "images_per_s": round(b / dt, 1), "ms_per_batch": round(1e3 * dt, 3), "forward_only_ms": round(1e3 * fwd, 3),
                          "gflop_per_image": round(gflop, 1), "tflops_forward": round(b * gflop / fwd / 1e3, 1), "frac_of_mfma_peak": round(b * gflop * 1e9 / fwd / PEAK, 4)}))
