#!/usr/bin/env python3
"""bench.py -- frames/s of OVO's per-frame open-vocabulary feature path on MI355X (BASELINE.json metric).

    python bench.py                                   (= --gpus 1 --steps 24 --warmup 3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one ROUND of N keyframes (N = number of GPUs; one 640x480 RGB-D keyframe per GPU) through the whole path
(ovo_amd/pipeline.py): back-projection, SAM2 image encoder, cull/project/match/vote/assign against the point map, ViT tokens + region
pooling, multi-view fusion, dense per-point fusion, instance query and dense-map query.  Every frame is a semantic keyframe
(map_every = segment_every = 1); the reference's own `fps` divides by segment_every = 10 (ovomapping.py:218).
Inputs are synthetic (seeded) and resident in HBM before the timed region; weights are seeded random init of the named
architectures (no checkpoints offline) -- throughput is weight independent.  With several GPUs every rank holds the whole frame
stream (the order-dependent integer passes run replicated), rank k owns keyframe k of a round for the encoders, and the round's one
exchange (all-gather of the owners' descriptors over RCCL) sits INSIDE the step.

Rank 0 prints ONE JSON line: the driver contract fields plus
  roofline      dominant kernel = the bf16 MFMA GEMM instantiation with the most time, hipEvents around every launch in a second,
                profiled pass -- as run on three concurrent streams, and with the streams folded under `isolated`; `traffic` = HBM
                bytes per launch from the committed PMC reduction profiles/pmc_traffic.json; `frame_frac` = the whole frame's
                flops / step time / peak; `measured_peaks` = stream copy and 8192^3 GEMM measured in this job
  cpu_baseline  the CPU oracle of the same path: 1 warm-up + median of 3 frames, N = 1 only
  parity        GPU vs oracle on one frame of this workload: max |descriptor error|, per-point instance-id mismatches
  per_step_ms / sustained   step cadence statistics of the timed steps and a >= 2 s continuation of the same stream
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
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vit", default="PE-Core-L14-336", help="ViT card (ovo_amd.encoders.vit.SPECS)")
    ap.add_argument("--sam", default="hiera_b+", help="SAM2 image encoder card, or 'none'")
    ap.add_argument("--map-points", type=int, default=1_000_000)
    ap.add_argument("--texts", type=int, default=10)
    ap.add_argument("--no-dense", action="store_true", help="skip the dense per-point accumulate / query")
    ap.add_argument("--encoder-batch", type=int, default=int(os.environ.get("OVO_ENCODER_BATCH", "14")),
                    help="keyframes (per GPU) whose SAM2 / ViT forwards run as ONE batched forward each (encoder look-ahead; 1 = per frame; 14 keyframes = "
                         "28 crops x 577 tokens = 16 156 rows make every ViT product a whole number of 256-CU rounds of 256 x 256 tiles: +1 % over 12, "
                         "profiles/r05c_encoder_batch_sweep.txt). "
                         "The reference defers a keyframe's descriptors by kf_queue_delay = 10 keyframes (ovo.yaml:53), so results do not change")
    ap.add_argument("--sam-full", action="store_true", help="not the headline workload: also run SAM2's mask decoder on a 16x16 click grid and the "
                    "automatic-mask-generator filters every frame (SURVEY.md f1); tracking still consumes the synthetic masks, because "
                    "random-init SAM2 weights keep no mask")
    ap.add_argument("--sam-own-masks", action="store_true",
                    help="with --sam-full (implied): the masks SAM2's generator keeps drive the tracking of their keyframe (the reference's default path, "
                         "mask_generator.py:102-120) instead of the frame's precomputed masks.  The weights are random-init (no checkpoint offline), whose "
                         "predicted IoU / stability scores are noise: the generator's two score thresholds and the mask-NMS score threshold are set to 0 "
                         "(--sam-thresholds) so that a realistic number of masks (~30 per frame) survives the box / mask NMS")
    ap.add_argument("--sam-thresholds", default="0.0,0.0,0.0", help="pred_iou, stability, mask-NMS score thresholds used by --sam-own-masks")
    ap.add_argument("--sam-min-own", type=int, default=8,
                    help="--sam-own-masks: a keyframe whose generator keeps fewer masks than this is tracked with the frame's precomputed masks -- after "
                         "the whole generator chain ran and was waited for (random-init weights keep ~3 degenerate masks: see the help above)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--census", default=None, metavar="FILE",
                    help="measurement runs (tools/gpu_measure.sh: the rocprofv3 --pmc passes): profile EVERY launch of the process and write, per GEMM tile, "
                         "the launch count and the algorithmic bytes per launch -- the denominator of the PMC traffic of exactly these launches; implies --no-roofline")
    ap.add_argument("--dense-merge", choices=("auto", "none", "reduce"), default="auto",
                    help="N > 1: also time north_star's literal collective -- ONE bucketed RCCL sum-reduce of whole per-GPU dense accumulators "
                         "f32[map_points, D] (parallel.allreduce_dense_) -- beside the sharded design the keyframe path uses (auto = reduce when N > 1)")
    ap.add_argument("--profile-steps", type=int, default=24,
                    help="steps of each of the two profiled passes behind `roofline` (as run / isolated); two whole look-ahead groups at the default "
                         "--encoder-batch 12, so the profiled launches have the timed region's shapes (8 steps profiled 4-keyframe forwards: all GEMMs "
                         "555 TFLOP/s isolated against 603 at the timed shapes on one box)")
    ap.add_argument("--projection-world", type=int, default=8, help="N = 1 only: after the timed legs, ONE GPU emulates rank 0 of a job of this many GPUs "
                    "(its own keyframe's encoders + every keyframe's replicated passes + 1/N of the dense rows, all-gather replaced by a local copy) "
                    "and reports the round time under `projection` (0 = off)")
    ap.add_argument("--no-shared-crops", action="store_true", help="skip the `shared_crops` leg (identical TextRegion crops encoded once; not the headline)")
    ap.add_argument("--no-online", action="store_true", help="skip the `online` leg (the same workload with no encoder look-ahead)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="after the timed steps: keep stepping the same stream for this long (0 = off)")
    return ap.parse_args()


def cpu_frame(pipe_args, frame_np, map_xyz, texts):
    """The CPU oracle (a port of the reference's algorithm, oracle/) on ONE frame of the same workload.
    -> (seconds, descriptors f32[n, D], the masks they were pooled from, updated per-point instance ids, the oracle's point map)."""
    from oracle import features as OF, hiera as OH, semantic as OS, vit as OV
    from ovo_amd import synthetic as syn
    from ovo_amd.encoders import hiera as EH, vit as EV
    spec = EV.SPECS[pipe_args.vit]
    sd = EV.random_state(spec, 0)
    rope = OV.rope_for(spec)
    K = syn.scannet_intrinsics(1.0)
    rgb, rgb_lr, depth, c2w, seg, masks = frame_np
    t0 = time.time()
    pm = OS.PointMap(K)
    pm.xyz, pm.next_id = map_xyz.copy(), map_xyz.shape[0]
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
    used = fused if len(fused) else masks
    fm = OF.feature_masks(used, P * nh, P * nw)
    d = spec.width
    desc = OF.region_pool(xs, fm, sd["attn_pool.attn.in_proj_weight"][2 * d:], sd["attn_pool.attn.in_proj_bias"][2 * d:],
                          sd["attn_pool.attn.out_proj.weight"], sd["attn_pool.attn.out_proj.bias"], sd["proj"])
    OF.classify(OF.similarity(np.nan_to_num(desc), texts))
    return time.time() - t0, desc, used, upd, pm


def cpu_baseline(args, frames_np, map_xyz, texts):
    """1 warm-up frame + the median of 3 timed frames on the host cores."""
    # thread count: probed, not assumed -- one frame each at 32 / 64 / 128 threads (whatever the host has), the fastest is used for the timed frames
    # (round 4 fixed 32: "more threads only add OpenMP spin on these small ops" was never measured on the driver's 256-core host)
    host = os.cpu_count() or 1
    probe = {}
    torch.set_num_threads(min(32, host))
    cpu_frame(args, frames_np[0], map_xyz, texts)                  # warm-up (first-touch of the weights, thread pools)
    for threads in sorted({min(t, host) for t in (32, 64, 128)}):
        torch.set_num_threads(threads)
        probe[threads] = cpu_frame(args, frames_np[0], map_xyz, texts)[0]
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    times, last = [], None
    for f in frames_np[1:4]:
        last = cpu_frame(args, f, map_xyz, texts)
        times.append(last[0])
    times.sort()
    cpu_baseline.probe = {str(k): round(v, 2) for k, v in probe.items()}
    return 1.0 / times[len(times) // 2], threads, times, last


def parity_block(pipe, frame, map_xyz, oracle_out):
    """GPU vs oracle on ONE frame of this workload (BASELINE.md section 2 "Reported"): TextRegion descriptors of the same masks
    (max |unit-descriptor error|, north_star bound 1e-3) and the per-point instance ids after back-projection + tracking against the
    initial map (bit-exact: mismatches must be 0)."""
    from ovo_amd import synthetic as syn
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.pipeline import ResidentMasks
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    _, desc_ref, used_masks, upd_ref, pm_ref = oracle_out
    dev = pipe.device
    tr = pipe.clip.textregion
    img = frame.rgb.permute(2, 0, 1).contiguous()
    desc = tr.predict(img, torch.from_numpy(np.ascontiguousarray(used_masks)).to(dev), scale=1.0 / 255.0).cpu().numpy()
    ok = ~np.isnan(desc_ref).any(1)
    err = float(np.abs(desc[ok] - desc_ref[ok]).max()) if ok.any() else None
    # the same descriptors through the BATCHED forward the timed region runs (one frame alone is 1154 token rows: the small-batch kernels, LayerNorm launches; the
    # look-ahead groups of the bench are 16 156 rows: 256-row ping-pong kernels with the LayerNorms folded into the products): this frame's crops 14 times over
    masks_dev = torch.from_numpy(np.ascontiguousarray(used_masks)).to(dev)
    one = tr.vlm.preprocess(img, tr.forward_crops(img.shape[1], img.shape[2]), scale=1.0 / 255.0)
    feats = tr.vlm.forward(one.repeat(14, 1, 1, 1), tokens=True)[:one.shape[0]]
    desc_b = tr.pe_value_with_sam2_attn(tr.get_features_mask(masks_dev), feats).cpu().numpy()
    err_b = float(np.abs(desc_b[ok] - desc_ref[ok]).max()) if ok.any() else None
    K = torch.from_numpy(syn.scannet_intrinsics(1.0)).to(dev)
    vm = VanillaMapper({"device": str(dev), "mapping": {"k_pooling": 3}}, K)
    n0 = map_xyz.shape[0]
    vm.set_map_dict({"xyz": torch.from_numpy(map_xyz), "obj_ids": torch.full((n0, 1), -1, dtype=torch.int32),
                     "ids": torch.arange(n0, dtype=torch.int32)[:, None], "max_id": n0, "color": torch.zeros((n0, 3), dtype=torch.uint8)})
    seam = ResidentMasks()
    seam.frames = {frame.index: frame}
    cfg = {"match_distance_th": 0.05, "track_th": 100, "depth_filter": True, "log": False, "kf_queue_delay": 0, "debug_info": False,
           "clip": {"embed_type": "TextRegion", "k_top_views": 10000, "fusion": "avg_pooling"}, "sam": {"precomputed": True}}
    ovo = OVO(cfg, None, "parity", K, device=str(dev), clip_generator=pipe.clip, mask_generator=seam)
    fd = [frame.index, frame.rgb_lr, frame.depth, frame.c2w]
    vm.track_camera(fd)
    vm.map(fd, vm._c2w_host[frame.index])
    upd = ovo.detect_and_track_objects([frame.index, frame.rgb, frame.depth, (1.0, 1.0, pipe.crop_edge)], vm.get_map(), vm._c2w_host[frame.index])
    upd = upd.cpu().numpy().reshape(-1)
    same_points = bool(vm.pcd.shape[0] == pm_ref.xyz.shape[0] and np.array_equal(vm.pcd.cpu().numpy(), pm_ref.xyz))
    mism = int((upd != upd_ref.reshape(-1)).sum()) if upd.shape == upd_ref.reshape(-1).shape else -1
    return {"max_abs_desc_err": None if err is None else round(err, 6), "max_abs_desc_err_batched_forward": None if err_b is None else round(err_b, 6),
            "descriptors": int(ok.sum()), "index_mismatches": mism,
            "points": int(upd.shape[0]), "map_points_identical": same_points,
            "note": "one frame of this workload: GPU TextRegion descriptors vs the fp32 oracle on the same masks (alone = 1154 token rows, and as the first of 14 "
                    "copies = the 16 156-row batched forward of the timed region: 256-row kernels, LayerNorms folded into the products); per-point instance ids after "
                    "back-projection + tracking on the initial map vs the oracle (bit-exact expected)"}


def pmc_traffic(tile: str):
    """HBM bytes per launch of the dominant GEMM instantiation from the committed PMC reduction (profiles/pmc_traffic.json,
    made by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    if tile == "stream":
        with open(path) as fh:
            hits = [v for k, v in json.load(fh)["kernels"].items() if "k_gemm_stream" in k]
        n = sum(v["launches"] for v in hits)
        return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n) if n else None
    bm, bn = tile.split(",")
    with open(path) as fh:
        kernels = json.load(fh)["kernels"]
    hits = [v for k, v in kernels.items() if (f"k_gemmILi{bm}ELi{bn}ELi64E" in k or f"k_gemm8pILi{bm}ELi{bn}E" in k or f"k_gemm8p<{bm}, {bn}" in k)
            and ("DF16b" in k or "__bf16" in k or "bool _Accum" in k)]
    n = sum(v["launches"] for v in hits)
    return round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n) if n else None


def pmc_traffic_census(tile: str):
    """Algorithmic bytes per launch of the SAME launches `pmc_traffic` averages over (tools/gpu_measure.sh runs the PMC passes with --census and
    tools/pmc_traffic.py stores the result): the bench's own `algorithmic_bytes_per_launch` describes its profiled pass (14 + 10-frame groups), the PMC
    passes also contain the priming forwards and the warm-up group -- two populations with different means."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        c = (json.load(fh).get("census") or {}).get("tiles", {}).get(tile)
    return None if not c or not c.get("algorithmic_bytes_per_launch") else {"launches": c["launches"], "algorithmic_bytes_per_launch": round(c["algorithmic_bytes_per_launch"])}


def pmc_traffic_stamp():
    """Where `roofline.traffic` comes from: the committed PMC reduction carries the hash of the kernel sources it was measured on (tools/pmc_traffic.py);
    this says whether the sources of THIS run are the same."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        measured = json.load(fh).get("csrc_sha16")
    root = os.path.join(ROOT, "ovo_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(root, f), "rb") as fh:
                h.update(fh.read())
    here = h.hexdigest()[:16]
    return {"source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not collected in this run)",
            "measured_on_csrc_sha16": measured, "this_run_csrc_sha16": here, "same_kernel_sources": bool(measured) and measured == here}


def profile_pass(pipe, feed, rounds, lib):
    """hipEvent pairs around every GEMM / attention / track_project launch of `rounds` steps (on the launching streams)."""
    from ovo_amd import _lib as L
    L.check(lib.ovo_profile_start())
    feed.run(pipe, rounds)
    ms, work, n = (C.c_double * 10)(), (C.c_double * 10)(), (C.c_int64 * 10)()
    L.check(lib.ovo_profile_stop(ms, work, n, 10))
    nbytes = (C.c_double * 10)()
    L.check(lib.ovo_profile_bytes(nbytes, 10))
    steps = rounds                                                # per frame of THIS rank: one owned keyframe per round
    tiles = {3: "256,256", 0: "256,128", 4: "128,128", 5: "128,64", 6: "64,128", 7: "64,64", 8: "stream"}     # 256-row tiles: the ping-pong kernel (gemm8p.hip); stream: gemm_stream.hip
    dom = max(tiles, key=lambda k: ms[k])                      # the GEMM instantiation with the most time = dominant kernel
    tf = work[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    gemm_ms, gemm_work = sum(ms[k] for k in tiles), sum(work[k] for k in tiles)
    name = f"k_gemm8p<{tiles[dom]},64,bf16> (ovo_amd/csrc/gemm8p.hip)" if dom in (0, 3) else \
        ("k_gemm_stream<bf16> (ovo_amd/csrc/gemm_stream.hip)" if dom == 8 else f"k_gemm<{tiles[dom]},64,bf16> (ovo_amd/csrc/gemm.hip)")
    fold = os.environ.get("OVO_VIT_LNFOLD", "1") != "0" and getattr(getattr(getattr(pipe.clip, "textregion", None), "vlm", None), "ln_fold", False)
    return {"bound": "mfma", "kernel": name, "achieved": round(tf, 1),
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
            "layernorm_fold": bool(fold),       # True: this kernel's launches also carry the ViT's LayerNorm work (bf16 copy + row statistics out, statistics in;
                                                # DESIGN.md section 9 item 9); OVO_VIT_LNFOLD=0 runs the LayerNorm kernels beside a lighter GEMM (frac + 0.01)
            "traffic": pmc_traffic(tiles[dom]), "traffic_stamp": pmc_traffic_stamp(), "traffic_population": pmc_traffic_census(tiles[dom]),
            "algorithmic_bytes_per_launch": round(nbytes[dom] / max(n[dom], 1)),
            "launches_per_frame": n[dom] / steps, "avg_launch_us": round(1e3 * ms[dom] / max(n[dom], 1), 2),
            "gemm_tiles_ms_per_frame": {tiles[k]: round(ms[k] / steps, 3) for k in tiles},
            "gemm_tiles_tflops": {tiles[k]: round(work[k] / (ms[k] * 1e-3) / 1e12, 1) for k in tiles if ms[k] > 0},
            "gemm_all_ms_per_frame": round(gemm_ms / steps, 3),
            "gemm_all_tflops": round(gemm_work / (gemm_ms * 1e-3) / 1e12, 1) if gemm_ms > 0 else 0.0,
            "attention_ms_per_frame": round(ms[1] / steps, 3),
            "attention_tflops": round(work[1] / (ms[1] * 1e-3) / 1e12, 1) if ms[1] > 0 else 0.0,
            "track_project_ms_per_frame": round(ms[2] / steps / max(pipe.world, 1), 4),
            "track_project_gbs": round(work[2] / (ms[2] * 1e-3) / 1e9, 1) if ms[2] > 0 else 0.0,
            # the dense scatter-reduce fusion of one keyframe.  Round 6 (fusion.hip: k_scatter_query): ONE launch from the tracking pass's hit list that also
            # re-queries the changed rows (what was scan + apply + ovo_similarity_rows); algorithmic bytes = changed rows x (8 D + 12)
            "scatter_form": "fused accumulate + re-query (ovo_scatter_accum_query)" if getattr(pipe.ovo, "hit_shard", None) is not None else "scan + apply (+ separate query launch)",
            "scatter_accum_us": round(1e3 * ms[9] / max(n[9], 1), 2), "scatter_accum_mb": round(work[9] / max(n[9], 1) / 1e6, 1),
            "scatter_accum_gbs": round(work[9] / (ms[9] * 1e-3) / 1e9, 1) if ms[9] > 0 else 0.0,
            "hbm_peak_gbs": HBM_PEAK_GBS}


def measured_peaks(dev, lib):
    """Stream copy and a plain 8192^3 bf16 GEMM measured in THIS job, beside the nominal peaks."""
    from ovo_amd import _lib as L
    n = 1 << 28                                                    # 1 GiB of f32
    a, b = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    a.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b.copy_(a)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    b.zero_()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        b.zero_()
    e1.record()
    torch.cuda.synchronize()
    write_gbs = 5 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    s_ = a.sum()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        s_ = a.sum()
    e1.record()
    torch.cuda.synchronize()
    read_gbs = 5 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    m = 8192
    x = torch.randn(m, m, device=dev).to(torch.bfloat16)
    w = torch.randn(m, m, device=dev).to(torch.bfloat16)
    o = torch.empty(m, m, dtype=torch.bfloat16, device=dev)
    g = L.Gemm()
    g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, g.add, g.ld_add = x.data_ptr(), m, w.data_ptr(), m, None, o.data_ptr(), m, None, 0
    g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, m, m, 2, 2, 0, 1.0
    for _ in range(5):
        L.check(lib.ovo_gemm(C.byref(g), L.stream()))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        L.check(lib.ovo_gemm(C.byref(g), L.stream()))
    e1.record()
    torch.cuda.synchronize()
    tf = 20 * 2.0 * m ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return {"hbm_copy_gbs": round(copy_gbs, 1), "hbm_copy_frac_of_8tbs": round(copy_gbs / HBM_PEAK_GBS, 3),
            "hbm_write_only_gbs": round(write_gbs, 1), "hbm_read_only_gbs": round(read_gbs, 1),
            "gemm_8192_bf16_tflops": round(tf, 1), "gemm_8192_frac_of_peak": round(tf / MFMA_BF16_PEAK_TFLOPS, 3),
            "note": "torch device copy / zero fill / sum of 1 GiB f32 (read + write, write-only, read-only bytes) and ovo_gemm 8192^3 bf16 on random "
                    "operands (5 warm-ups + 20 launches), measured in this job"}


def shared_crops_leg(args, dev, frames, sam):
    """NOT the headline workload: the same stream with `share_identical_crops` -- a 640 x 480 frame tiles into one 336-pixel tile that IS the
    global image (textregion.py:104-143), so the reference encodes the same pixels twice; this leg encodes them once (same descriptor bits,
    test_textregion_shared_crops_equal_separate_forwards).  Off by default everywhere; reported so that the cost of the duplicate is on record."""
    from ovo_amd.pipeline import Frame, FramePipeline
    eb = max(args.encoder_batch, 1)
    steps, warm = 4 * eb, eb
    pipe = FramePipeline(dev, vit_card=args.vit, sam_card=sam, n_map=args.map_points, n_text=args.texts, dense=not args.no_dense, sam_full=args.sam_full,
                         extra_capacity=(steps + warm + 2) * 72_000, seed=0, encoder_batch=eb, share_crops=True)
    pool = frames * ((steps + warm) // len(frames) + 1)
    stream = [Frame(300_000 + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:steps + warm])]
    if eb > 1:
        pipe.prime(*stream[0].rgb.shape[:2])
    feed = Feed(stream, 1)
    feed.run(pipe, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feed.run(pipe, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del pipe
    torch.cuda.empty_cache()
    return {"encoder_batch": eb, "steps": steps, "frames_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3),
            "note": "NOT `value`: identical TextRegion crops (global image = the single tile at 640 x 480) encoded once instead of twice; same descriptors"}


def online_leg(args, dev, frames, sam):
    """The same workload with NO look-ahead (encoder_batch 1): keyframe t's encoders start when keyframe t arrives -- what an online
    mapper whose masks come from SAM2 can do (ovo.py:121-166); the headline's look-ahead needs the next keyframes of a recorded stream."""
    from ovo_amd.pipeline import Frame, FramePipeline
    steps, warm = 48, 6
    pipe = FramePipeline(dev, vit_card=args.vit, sam_card=sam, n_map=args.map_points, n_text=args.texts, dense=not args.no_dense, sam_full=args.sam_full,
                         extra_capacity=(steps + warm + 2) * 72_000, seed=0, encoder_batch=1)
    pool = frames * ((steps + warm) // len(frames) + 1)
    stream = [Frame(200_000 + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:steps + warm])]
    feed = Feed(stream, 1)
    feed.run(pipe, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feed.run(pipe, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del pipe
    torch.cuda.empty_cache()
    return {"encoder_batch": 1, "steps": steps, "frames_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3),
            "note": "same workload, no encoder look-ahead: the encoders of keyframe t are launched when it arrives (overlap with the previous keyframe's tail only)"}


def dense_merge_leg(args, dev, D):
    """north_star's literal collective, timed beside the design the keyframe path uses: every GPU holds a whole dense accumulator
    f32[map_points, D] + counts and ONE bucketed RCCL sum-reduce merges them (parallel.allreduce_dense_).  The keyframe path instead shards the
    accumulators by point and exchanges the round's descriptors (`exchange`): no reduce, bit-identical to one process.
    All ranks call this (collective); rank 0 reports."""
    from ovo_amd import parallel
    n = int(args.map_points)
    acc = torch.zeros((n, D), dtype=torch.float32, device=dev)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    acc[::4097] = 1.0
    world = parallel.world_size()
    calls = parallel.allreduce_dense_(acc, cnt)                    # warm-up (communicator channels, first-use kernels)
    torch.cuda.synchronize()
    parallel.barrier()
    reps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        parallel.allreduce_dense_(acc, cnt)
    e1.record()
    torch.cuda.synchronize()
    ms = parallel.max_over_ranks(e0.elapsed_time(e1) / reps, dev)
    nbytes = acc.numel() * 4 + cnt.numel() * 4
    del acc, cnt
    torch.cuda.empty_cache()
    return {"mode": "reduce", "bytes_per_gpu": nbytes, "collectives": calls, "ms": round(ms, 3), "algbw_gbs": round(nbytes / ms / 1e6, 1),
            "busbw_gbs": round(nbytes / ms / 1e6 * 2 * (world - 1) / world, 1),
            "note": f"ONE bucketed sum-reduce of whole per-GPU accumulators f32[{n}, {D}] + i32 counts ({parallel.DENSE_BUCKET_BYTES >> 20} MB buckets), "
                    "device-timed, max over ranks; NOT on the keyframe path (which shards by point: `exchange`), measured for comparison"}


def projection_leg(args, dev, frames, sam):
    """What ONE rank of an N-GPU job does per round, measured on this GPU (FramePipeline(emulate=(0, N))): the encoders and pooling of the one
    keyframe it owns, the replicated map / tracking chain of all N keyframes, store + re-fuse of all N keyframes' descriptors, the dense
    scatter / query of its 1/N of the rows; the round's all-gather is a local copy.  N / (round time) is what N such ranks deliver if the
    collective (512 KB per rank over xGMI) hides behind the next round's encoders -- a measured serial term, not a scaling result."""
    from ovo_amd.pipeline import Frame, FramePipeline
    N = args.projection_world
    groups = 3
    rounds = groups * max(args.encoder_batch, 1)                   # whole look-ahead groups of the emulated rank
    need = (rounds + max(args.encoder_batch, 1)) * N
    pipe = FramePipeline(dev, vit_card=args.vit, sam_card=sam, n_map=args.map_points, n_text=args.texts, dense=not args.no_dense, sam_full=args.sam_full,
                         extra_capacity=(need + 2) * 72_000, seed=0, encoder_batch=args.encoder_batch, emulate=(0, N))
    pool = frames * (need // len(frames) + 1)
    stream = [Frame(100_000 + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:need])]
    H, W = frames[0].rgb.shape[:2]
    pipe.prime(H, W)
    feed = Feed(stream, N)
    feed.run(pipe, max(args.encoder_batch, 1))                     # warm-up: one group
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feed.run(pipe, rounds - max(args.encoder_batch, 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed = rounds - max(args.encoder_batch, 1)
    del pipe
    torch.cuda.empty_cache()
    return {"world": N, "rounds": timed, "ms_per_round": round(1e3 * dt / timed, 3), "frames_per_s_if_hidden_exchange": round(N * timed / dt, 1),
            "note": f"one GPU doing the per-round work of rank 0 of {N}: 1 owned keyframe (encoders + pooling, look-ahead {args.encoder_batch}) + {N} replicated "
                    f"map / tracking chains + 1/{N} of the dense rows; all-gather replaced by a local copy; NOT a multi-GPU measurement"}


class Feed:
    """Rounds of `world` frames in order.  A step also sees the frames that follow it INSIDE the same region (warm-up / timed /
    profiled / sustained), so the encoder look-ahead never does work of a timed frame outside the timed region, nor work of later
    frames inside it."""

    def __init__(self, frames, world):
        self.frames, self.world, self.pos = frames, world, 0

    def left(self):
        return (len(self.frames) - self.pos) // self.world

    def run(self, pipe, rounds, stamp=None):
        end = self.pos + rounds * self.world
        assert end <= len(self.frames), "frame stream exhausted"
        for _ in range(rounds):
            group = self.frames[self.pos:self.pos + self.world]
            self.pos += self.world
            pipe.step_round(group, self.frames[self.pos:end])
            if stamp is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                stamp.append(ev)


def main():
    args = parse()
    from ovo_amd import _lib as L, parallel
    from ovo_amd.pipeline import Frame, FramePipeline, synthetic_frames
    rank, local_rank, world = parallel.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU (ovo_amd has no CPU path)"
    dev_index = parallel.local_device(local_rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = L.load()
    if os.environ.get("OVO_MAIN_PRIORITY"):                        # measurement knob: the keyframe's own stream at a hardware queue priority
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["OVO_MAIN_PRIORITY"])))

    if args.census:
        args.no_roofline = True
        L.check(lib.ovo_profile_start())                          # from here to the end of the process: the launches the PMC pass counts
    prof_rounds = 0 if args.no_roofline else args.profile_steps
    base_rounds = args.warmup + args.steps + 2 * prof_rounds
    # the sustained continuation re-uses the resident frames (new keyframe ids, same pixels): budget its map growth up front
    sustain_rounds = int(args.sustain_seconds * 450 / world) if args.sustain_seconds > 0 else 0
    sam = None if args.sam == "none" else args.sam
    own = {}
    if args.sam_own_masks:
        args.sam_full = True
        th = [float(x) for x in args.sam_thresholds.split(",")]
        own = {"own_masks": True, "amg_thresholds": (th[0], th[1]), "nms_score_thr": th[2], "min_own_masks": args.sam_min_own}
    pipe = FramePipeline(dev, vit_card=args.vit, sam_card=sam, n_map=args.map_points, n_text=args.texts, dense=not args.no_dense, sam_full=args.sam_full,
                         extra_capacity=(base_rounds * world + 2) * 72_000 + sustain_rounds * world * 16_000, seed=0, encoder_batch=args.encoder_batch, **own)
    # every rank holds the whole stream: the order-dependent passes run replicated (pipeline.py); rank k owns frame k of a round
    # (the two profiled passes re-use the pixels of the warm-up + timed frames under new keyframe ids, as the sustained leg does: rendering a
    # synthetic frame costs ~0.2 s of host time, and an 8-rank job would render 2 x 24 x 8 more of them on every rank)
    fresh = (args.warmup + args.steps) * world
    frames = synthetic_frames(fresh, dev, seed=int(os.environ.get("OVO_BENCH_SEED", "0")))
    for i in range(2 * prof_rounds * world):
        f = frames[i % fresh]
        frames.append(Frame(fresh + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks))
    H, W = frames[0].rgb.shape[:2]
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    map0 = pipe.slam.pcd.cpu().numpy().copy() if want_cpu else None

    pipe.prime(H, W)                                               # allocator / first-launch set-up at the full encoder-batch width
    feed = Feed(frames, world)
    feed.run(pipe, args.warmup)
    torch.cuda.synchronize()
    parallel.barrier()
    stamps = []
    first = torch.cuda.Event(enable_timing=True)
    first.record()
    L.check(lib.ovo_marker(1, L.stream()))                         # a kernel trace of this command is cut at these two markers
    t0 = time.perf_counter()
    x0, n0 = pipe.exchange_ms, pipe.exchanges
    del pipe.exchange_events[:]                                    # the timed region's collectives only (warm-up rounds dropped: the list holds 4096)
    feed.run(pipe, args.steps, stamps)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    L.check(lib.ovo_marker(2, L.stream()))
    xchg_ms = (pipe.exchange_ms - x0) / max(pipe.exchanges - n0, 1)
    xchg_dev_ms = pipe.exchange_device_ms() if world > 1 else None              # taken HERE: the profiled / sustained legs below append their own events
    cadence = sorted(a.elapsed_time(b) for a, b in zip([first] + stamps[:-1], stamps))

    roof = None
    if prof_rounds > 0:
        roof = profile_pass(pipe, feed, prof_rounds, lib)                        # as timed: three concurrent HIP streams
        torch.cuda.synchronize()
        if args.encoder_batch > 1 or world > 1:
            pipe.serial = True                                                 # one stream: every kernel has the chip to itself
            torch.cuda.synchronize()
            L.check(lib.ovo_marker(3, L.stream()))                             # markers 3 / 4 of a kernel trace: the `isolated` pass (tools/kstats_region.py)
            iso = profile_pass(pipe, feed, prof_rounds, lib)
            torch.cuda.synchronize()
            L.check(lib.ovo_marker(4, L.stream()))
            pipe.serial = False
        else:
            streams = (pipe.sam_stream, pipe.prefetch)
            pipe.sam_stream, pipe.prefetch = None, False
            iso = profile_pass(pipe, feed, prof_rounds, lib)
            pipe.sam_stream, pipe.prefetch = streams
        roof["isolated"] = {k: iso[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "gemm_tiles_tflops", "gemm_all_ms_per_frame", "gemm_all_tflops",
                                                "attention_ms_per_frame", "attention_tflops", "track_project_gbs", "scatter_accum_gbs", "scatter_accum_us")}
        roof["isolated"]["note"] = "same kernels, second profiled pass with the SAM2 / ViT streams folded into one"
        # the kernel metric as FLAT keys (a record that keeps only scalars still carries it): the dominant kernel with the chip to itself
        roof["isolated_kernel"] = iso["kernel"]
        roof["isolated_achieved"], roof["isolated_frac"], roof["isolated_avg_launch_us"] = iso["achieved"], iso["frac"], iso["avg_launch_us"]
        roof["isolated_algorithmic_bytes_per_launch"] = iso["algorithmic_bytes_per_launch"]
        roof["isolated_gemm_all_tflops"], roof["isolated_gemm_all_ms_per_frame"] = iso["gemm_all_tflops"], iso["gemm_all_ms_per_frame"]
        roof["isolated_attention_tflops"], roof["isolated_attention_ms_per_frame"] = iso["attention_tflops"], iso["attention_ms_per_frame"]
        roof["isolated_stream_gemm_tflops"] = iso["gemm_tiles_tflops"].get("stream")
        roof["isolated_scatter_accum_us"], roof["isolated_scatter_accum_gbs"] = iso["scatter_accum_us"], iso["scatter_accum_gbs"]
        torch.cuda.synchronize()

    sustained = None
    if sustain_rounds > 0:                                        # >= 2 s of the same stream: long enough for an external sampler to see
        nxt = frames[-1].index + 1
        pool = frames * (sustain_rounds * world // len(frames) + 1)
        more = [Frame(nxt + i, f.rgb[:], f.rgb_lr, f.depth, f.c2w, f.seg_map, f.masks) for i, f in enumerate(pool[:sustain_rounds * world])]   # rgb[:]: a view of its own --
        # the look-ahead keys a frame's tokens by the identity of its image object, and the pool repeats the resident frames
        sfeed = Feed(more, world)
        chunk = max(args.encoder_batch, 1) * 4
        parallel.barrier()
        ts = time.perf_counter()
        done = 0
        while sfeed.left() >= chunk and time.perf_counter() - ts < args.sustain_seconds and pipe.slam._n + chunk * world * 16_000 < pipe.slam._cap:
            sfeed.run(pipe, chunk)
            done += chunk
            if done % (chunk * 4) == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        parallel.barrier()
        dt = parallel.max_over_ranks(time.perf_counter() - ts, dev)
        sustained = {"rounds": done, "seconds": round(dt, 3), "frames_per_s": round(done * world / dt, 2) if dt > 0 else None,
                     "points_end": int(pipe.slam._n), "instances_end": len(pipe.ovo.objects),
                     "note": "the same frame stream continued (map and instance table keep growing), host-timed with a sync every 16 x encoder_batch rounds"}

    peaks = measured_peaks(dev, lib) if (rank == 0 and not args.no_roofline) else None

    projection = None
    if world == 1 and args.projection_world > 1:
        projection = projection_leg(args, dev, frames, sam)
    online = online_leg(args, dev, frames, sam) if (world == 1 and args.encoder_batch > 1 and not args.no_online) else None
    shared = shared_crops_leg(args, dev, frames, sam) if (world == 1 and not args.no_shared_crops) else None

    dense_merge = None
    if world > 1 and args.dense_merge in ("auto", "reduce"):
        dense_merge = dense_merge_leg(args, dev, pipe.D)

    cpu, parity = None, None
    if want_cpu:
        idx = [args.warmup + i for i in range(4)]
        fnp = [(f.rgb.cpu().numpy(), f.rgb_lr.cpu().numpy(), f.depth.cpu().numpy(), f.c2w, f.seg_map.cpu().numpy(), f.masks.cpu().numpy())
               for f in (frames[i] for i in idx)]
        try:
            v, cores, times, last = cpu_baseline(args, fnp, map0, pipe.texts.cpu().numpy())
            cpu = {"value": round(v, 4), "unit": "frames/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port", "seconds_per_frame": [round(t, 2) for t in times],
                   "thread_probe_seconds_per_frame": getattr(cpu_baseline, "probe", None),
                   "sample": "frames of the same workload through oracle/ (fp32 torch-CPU encoders + C geometry): 1 warm-up, one frame each at 32 / 64 / 128 "
                             f"threads (the fastest count is used: torch.set_num_threads({cores})), then the median of 3 more frames; after the GPU run"}
            parity = parity_block(pipe, frames[idx[-1]], map0, last)
        except Exception as e:                                     # the baseline must never take the bench down
            cpu = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "host_cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        fl = pipe.flops_per_frame(H, W)
        ms_step = 1e3 * elapsed / args.steps
        frame_tflops = sum(fl.values()) * world / (ms_step * 1e-3) / 1e12
        if roof is not None:
            roof["frame_frac"] = round(frame_tflops / world / MFMA_BF16_PEAK_TFLOPS, 4)
            roof["frame_frac_note"] = "(ViT + SAM2 encoder flops of one frame) / (step time per frame and GPU) / 2.5 PFLOP/s: the whole step against the MFMA peak"
            roof["measured_peaks"] = peaks
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
                       "masks": (f"SAM2's own: generator (thresholds {args.sam_thresholds}) -> mask NMS -> seg map -> tracking; "
                                 f"{sum(pipe.own_mask_counts) / max(len(pipe.own_mask_counts), 1):.1f} masks per keyframe over {len(pipe.own_mask_counts)} keyframes; "
                                 f"{pipe.own_fallbacks} keyframes kept fewer than {pipe.min_own_masks} and were tracked -- after waiting for that chain -- with the frame's "
                                 f"precomputed masks (random-init weights produce empty / whole-image masks)")
                                if pipe.own_masks else "precomputed-mask seam (mask_generator.py:94-95)",
                       "encoder_batch": pipe.encoder_batch,
                       "dense_query": "resident class map, rows touched by the keyframe re-evaluated" if (pipe.dense and pipe.incremental_query) else "all rows",
                       "parallelism": (f"rounds of {world} keyframes: rank k owns keyframe k (SAM2 + ViT + pooling), tracking / back-projection replicated in "
                                       f"keyframe order, one all-gather of descriptors per round inside the step, dense accumulators sharded by point "
                                       f"(block-cyclic, no reduce)") if world > 1 else "single GPU",
                       "gflop_per_frame": {k: round(v / 1e9, 1) for k, v in fl.items()},
                       "points_end": pipe.last.get("n_points"), "instances_end": pipe.last.get("n_instances")},
            "model_tflops_effective": round(frame_tflops, 1),
            "per_step_ms": {"median": round(cadence[len(cadence) // 2], 3), "min": round(cadence[0], 3), "max": round(cadence[-1], 3),
                            "note": "intervals between hipEvents recorded on the main stream at the end of every timed step"},
            "sustained": sustained,
            "online": online, "shared_crops": shared, "roofline": roof, "cpu_baseline": cpu, "parity": parity, "projection": projection,
        }
        if world > 1:
            dev_ms = xchg_dev_ms
            line["exchange"] = {"collectives_per_round": 1, "bytes_per_rank": int(pipe.xchg.numel() * 4), "host_ms_per_round": round(xchg_ms, 3),
                                "device_ms_per_round": None if dev_ms is None else round(dev_ms, 4),
                                "backend": torch.distributed.get_backend(),
                                "note": "all-gather of the round's descriptors, issued inside the timed step; device_ms = hipEvents around the collective "
                                        "on its stream (includes what it waits for on that stream after the first event: nothing is queued between)"}
            line["dense_merge"] = dense_merge
        print(json.dumps(line))
    if args.census and rank == 0:
        torch.cuda.synchronize()
        ms, work, n = (C.c_double * 10)(), (C.c_double * 10)(), (C.c_int64 * 10)()
        L.check(lib.ovo_profile_stop(ms, work, n, 10))
        nbytes = (C.c_double * 10)()
        L.check(lib.ovo_profile_bytes(nbytes, 10))
        tiles = {3: "256,256", 0: "256,128", 4: "128,128", 5: "128,64", 6: "64,128", 7: "64,64", 8: "stream"}
        with open(args.census, "w") as fh:
            json.dump({"note": "every launch of this process (priming, warm-up, timed steps), from the library's event profiler",
                       "tiles": {tiles[k]: {"launches": int(n[k]), "algorithmic_bytes_per_launch": (nbytes[k] / n[k] if n[k] else None),
                                            "flop_per_launch": (work[k] / n[k] if n[k] else None)} for k in tiles}}, fh, indent=1)
    parallel.barrier()


if __name__ == "__main__":
    main()
