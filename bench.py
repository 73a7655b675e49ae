#!/usr/bin/env python3
"""bench.py -- frames/s of OVO's per-frame open-vocabulary feature path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one 640x480 RGB-D keyframe through the whole path on each rank (ovo_amd/pipeline.py): back-projection,
SAM2 image encoder, cull/project/match/vote/assign against the point map, ViT tokens + region pooling, multi-view
fusion, dense per-point fusion, instance query and dense-map query.  Every frame is a semantic keyframe
(map_every = segment_every = 1); the reference's own `fps` divides by segment_every = 10 (ovomapping.py:218).
Inputs are synthetic (seeded) and resident in HBM before the timed region; weights are seeded random init of the
named architectures (no checkpoints offline) -- throughput is weight independent.

Rank 0 prints ONE JSON line: the driver contract fields plus `roofline` (dominant kernel = the bf16 MFMA GEMM
instantiation with the most time, measured with hipEvents around every launch in a second, profiled pass -- as run on
three concurrent streams, and again with the streams folded under `isolated`; `traffic` = HBM bytes per launch from the
committed PMC reduction profiles/pmc_traffic.json) and `cpu_baseline` (the CPU oracle of the same path on one frame,
N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vit", default="PE-Core-L14-336", help="ViT card (ovo_amd.encoders.vit.SPECS)")
    ap.add_argument("--sam", default="hiera_b+", help="SAM2 image encoder card, or 'none'")
    ap.add_argument("--map-points", type=int, default=1_000_000)
    ap.add_argument("--texts", type=int, default=10)
    ap.add_argument("--no-dense", action="store_true", help="skip the dense per-point accumulate / query")
    ap.add_argument("--encoder-batch", type=int, default=int(os.environ.get("OVO_ENCODER_BATCH", "4")),
                    help="keyframes whose SAM2 / ViT forwards run as ONE batched forward each (encoder look-ahead; 1 = per frame). "
                         "The reference defers a keyframe's descriptors by kf_queue_delay = 10 keyframes (ovo.yaml:53), so results do not change")
    ap.add_argument("--sam-full", action="store_true", help="not the headline workload: also run SAM2's mask decoder on a 16x16 click grid and the "
                    "automatic-mask-generator filters every frame (SURVEY.md f1); tracking still consumes the synthetic masks, because "
                    "random-init SAM2 weights keep no mask")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=8)
    return ap.parse_args()


def cpu_baseline(pipe_args, frame_np, map_xyz, texts):
    """The CPU oracle (a port of the reference's algorithm, oracle/) on ONE frame of the same workload."""
    from oracle import features as OF, geometry as OG, hiera as OH, semantic as OS, vit as OV
    from ovo_amd import synthetic as syn
    from ovo_amd.encoders import hiera as EH, vit as EV
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # more threads only add OpenMP spin on these small ops
    spec = EV.SPECS[pipe_args.vit]
    sd = EV.random_state(spec, 0)
    rope = EV.rope_tables(spec) if spec.use_rope else None
    K = syn.scannet_intrinsics(1.0)
    rgb, rgb_lr, depth, c2w, seg, masks = frame_np
    t0 = time.time()
    pm = OS.PointMap(K)
    pm.xyz, pm.next_id = map_xyz, map_xyz.shape[0]
    pm.ids = np.arange(pm.next_id, dtype=np.int32)[:, None]
    pm.ins = np.full(pm.next_id, -1, np.int32)
    pm.rgb = np.zeros((pm.next_id, 3), np.uint8)
    pm.integrate(rgb_lr, depth, c2w)
    if pipe_args.sam != "none":
        hs = EH.SPECS[pipe_args.sam]
        x = OV.resize_normalize(torch.from_numpy(rgb.transpose(2, 0, 1).copy()), hs.image_size, EH.IMAGENET_MEAN, EH.IMAGENET_STD, None, 1 / 255.0)
        OH.hiera_forward(EH.random_state(hs, 0), x[None], stages=hs.stages, heads=hs.heads, window_spec=hs.window_spec,
                         global_blocks=hs.global_blocks)
    tr = OS.SemanticTracker(K, 0.05, 100, True, 10000)
    matched, fused, _, upd = tr.step(depth, (1.0, 1.0, 12), pm.xyz, pm.ids, pm.ins, c2w, seg, masks)
    H, W = rgb.shape[:2]
    img = torch.from_numpy(rgb.transpose(2, 0, 1).copy())
    nh, nw = max(H // spec.image_size, 1), max(W // spec.image_size, 1)
    ch, cw = -(-H // nh), -(-W // nw)
    crops = [(0, 0, H, W)] + [(max(min(i * ch + ch, H) - ch, 0), max(min(j * cw + cw, W) - cw, 0), ch, cw) for i in range(nh) for j in range(nw)]
    batch = torch.stack([OV.resize_normalize(img, spec.image_size, spec.mean, spec.std, c, 1 / 255.0) for c in crops])
    tok = OV.vit_forward(sd, batch, patch=spec.patch, heads=spec.heads, act=spec.act, rope=rope, tokens=True)
    P = spec.grid
    xs = OF.stitch_tokens(tok[:, 1:].numpy(), P, P * nh, P * nw, nh, nw)
    fm = OF.feature_masks(fused if len(fused) else masks, P * nh, P * nw)
    d = spec.width
    desc = OF.region_pool(xs, fm, sd["attn_pool.attn.in_proj_weight"][2 * d:], sd["attn_pool.attn.in_proj_bias"][2 * d:],
                          sd["attn_pool.attn.out_proj.weight"], sd["attn_pool.attn.out_proj.bias"], sd["proj"])
    OF.classify(OF.similarity(np.nan_to_num(desc), texts))
    return 1.0 / (time.time() - t0), torch.get_num_threads()


def pmc_traffic(tile: str):
    """HBM bytes per launch of the dominant GEMM instantiation from the committed PMC reduction (profiles/pmc_traffic.json,
    made by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    bm, bn = tile.split(",")
    with open(path) as fh:
        kernels = json.load(fh)["kernels"]
    hits = [v for k, v in kernels.items() if (f"k_gemmILi{bm}ELi{bn}ELi64E" in k or f"k_gemm8pILi{bm}ELi{bn}E" in k) and "DF16b" in k]
    n = sum(v["launches"] for v in hits)
    return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n) if n else None


def profile_pass(pipe, it, steps, lib):
    """hipEvent pairs around every GEMM / attention / track_project launch of `steps` frames (on the launching streams)."""
    from ovo_amd import _lib as L
    L.check(lib.ovo_profile_start())
    it.region(steps)
    for _ in range(steps):
        f, upcoming = next(it)
        pipe.step(f, upcoming)
    ms, work, n = (C.c_double * 8)(), (C.c_double * 8)(), (C.c_int64 * 8)()
    L.check(lib.ovo_profile_stop(ms, work, n, 8))
    tiles = {3: "256,256", 0: "256,128", 4: "128,128", 5: "128,64", 6: "64,128", 7: "64,64"}     # 256-row tiles: the ping-pong kernel (gemm8p.hip)
    dom = max(tiles, key=lambda k: ms[k])                      # the GEMM instantiation with the most time = dominant kernel
    tf = work[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    gemm_ms, gemm_work = sum(ms[k] for k in tiles), sum(work[k] for k in tiles)
    name = f"k_gemm8p<{tiles[dom]},64,bf16> (ovo_amd/csrc/gemm8p.hip)" if dom in (0, 3) else f"k_gemm<{tiles[dom]},64,bf16> (ovo_amd/csrc/gemm.hip)"
    return {"bound": "mfma", "kernel": name, "achieved": round(tf, 1),
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": pmc_traffic(tiles[dom]),
            "launches_per_frame": n[dom] / steps, "avg_launch_us": round(1e3 * ms[dom] / max(n[dom], 1), 2),
            "gemm_tiles_ms_per_frame": {tiles[k]: round(ms[k] / steps, 3) for k in tiles},
            "gemm_all_ms_per_frame": round(gemm_ms / steps, 3),
            "gemm_all_tflops": round(gemm_work / (gemm_ms * 1e-3) / 1e12, 1) if gemm_ms > 0 else 0.0,
            "attention_ms_per_frame": round(ms[1] / steps, 3),
            "attention_tflops": round(work[1] / (ms[1] * 1e-3) / 1e12, 1) if ms[1] > 0 else 0.0,
            "track_project_ms_per_frame": round(ms[2] / steps, 4),
            "track_project_gbs": round(work[2] / (ms[2] * 1e-3) / 1e9, 1) if ms[2] > 0 else 0.0,
            "hbm_peak_gbs": HBM_PEAK_GBS}


def main():
    args = parse()
    from ovo_amd import _lib as L, parallel
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    rank, local_rank, world = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU (ovo_amd has no CPU path)"
    dev_index = parallel.local_device(local_rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = L.load()

    total = args.warmup + args.steps + (0 if args.no_roofline else 2 * args.profile_steps)
    sam = None if args.sam == "none" else args.sam
    pipe = FramePipeline(dev, vit_card=args.vit, sam_card=sam, n_map=args.map_points, n_text=args.texts, dense=not args.no_dense, sam_full=args.sam_full,
                         extra_capacity=(total + 2) * 72_000, seed=0, encoder_batch=args.encoder_batch)
    frames = synthetic_frames(total, dev, seed=100 * rank + int(os.environ.get("OVO_BENCH_SEED", "0")))   # each rank streams its own frames (weak scaling)
    H, W = frames[0].rgb.shape[:2]
    map0 = pipe.slam.pcd.cpu().numpy().copy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    class Feed:
        """Frames in order; a step also sees the frames that follow it INSIDE the same region (warm-up / timed / profiled), so the
        encoder look-ahead never does work of a timed frame outside the timed region, nor work of later frames inside it."""
        def __init__(self, frames):
            self.frames, self.pos, self.end = frames, 0, 0

        def region(self, n):
            self.end = self.pos + n

        def __next__(self):
            f = self.frames[self.pos]
            self.pos += 1
            return f, self.frames[self.pos:self.end]

    it = Feed(frames)

    def run(n):
        it.region(n)
        for _ in range(n):
            f, upcoming = next(it)
            pipe.step(f, upcoming)

    run(args.warmup)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    if os.environ.get("OVO_BENCH_PER_STEP"):                       # diagnosis only: sync + stamp every step
        stamps = []
        it.region(args.steps)
        for _ in range(args.steps):
            f, upcoming = next(it)
            pipe.step(f, upcoming)
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())
        print("per-step ms:", [round(1e3 * (b - a), 2) for a, b in zip([t0] + stamps, stamps)], file=sys.stderr)
    else:
        run(args.steps)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    merge_calls = 0
    t_merge = 0.0
    if world > 1:                                                 # dense accumulator merge: once per batch, reported separately
        tm = time.perf_counter()
        merge_calls = pipe.merge_dense()
        torch.cuda.synchronize()
        t_merge = parallel.max_over_ranks(time.perf_counter() - tm, dev)

    roof = None
    if not args.no_roofline and args.profile_steps > 0:
        roof = profile_pass(pipe, it, args.profile_steps, lib)                 # as timed: three concurrent HIP streams
        torch.cuda.synchronize()
        if args.encoder_batch > 1:
            pipe.serial = True                                                 # one stream: every kernel has the chip to itself
            iso = profile_pass(pipe, it, args.profile_steps, lib)
            pipe.serial = False
        else:
            streams = (pipe.sam_stream, pipe.prefetch)
            pipe.sam_stream, pipe.prefetch = None, False
            iso = profile_pass(pipe, it, args.profile_steps, lib)
            pipe.sam_stream, pipe.prefetch = streams
        roof["isolated"] = {k: iso[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "gemm_all_ms_per_frame", "gemm_all_tflops",
                                                "attention_ms_per_frame", "attention_tflops", "track_project_gbs")}
        roof["isolated"]["note"] = "same kernels, second profiled pass with the SAM2 / ViT-prefetch streams folded into one"

    cpu = None
    if map0 is not None:
        f = frames[args.warmup]
        fnp = (f.rgb.cpu().numpy(), f.rgb_lr.cpu().numpy(), f.depth.cpu().numpy(), f.c2w, f.seg_map.cpu().numpy(), f.masks.cpu().numpy())
        try:
            v, cores = cpu_baseline(args, fnp, map0, pipe.texts.cpu().numpy())
            cpu = {"value": round(v, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                   "sample": "1 frame of the same workload through oracle/ (fp32 torch-CPU encoders + C geometry), after the GPU run"}
        except Exception as e:                                     # the baseline must never take the bench down
            cpu = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        fl = pipe.flops_per_frame(H, W)
        ms_step = 1e3 * elapsed / args.steps
        line = {
            "metric": "frames/s (CLIP+SAM2+fusion+query), 640x480 ScanNet, 1/2/4/8 MI355X",
            "value": round(world * args.steps / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[2]: {args.vit} TextRegion descriptors (2 x 336^2 crops of a 640x480 frame) + "
                                   f"SAM2 {args.sam} image encoder @1024^2 + back-projection + cull/project/match/vote on a "
                                   f"{args.map_points}-point map + multi-view fusion + dense per-point fusion + {args.texts}-prompt "
                                   f"instance and dense-map query; every frame a keyframe; masks from the precomputed-mask seam (32/frame)",
                       "frames_per_step_per_gpu": 1, "map_points": args.map_points, "texts": args.texts, "masks_per_frame": int(frames[0].masks.shape[0]),
                       "sam2": "image encoder + mask decoder (256 clicks) + automatic-mask-generator filters" if args.sam_full else "image encoder",
                       "parallelism": f"frame-sharded x{world}, per-step RCCL all-reduce of descriptor accumulators" if world > 1 else "single GPU",
                       "gflop_per_frame": {k: round(v / 1e9, 1) for k, v in fl.items()},
                       "points_end": pipe.last.get("n_points"), "instances_end": pipe.last.get("n_instances")},
            "model_tflops_effective": round(sum(fl.values()) / (ms_step * 1e-3) / 1e12, 1),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if world > 1:
            line["dense_merge"] = {"collectives": merge_calls, "ms": round(1e3 * t_merge, 2)}
        print(json.dumps(line))
    parallel.barrier()


if __name__ == "__main__":
    main()
