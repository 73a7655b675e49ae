"""ONE maintainer-side run that pins what this repo had to restate from un-vendored upstream code ("parity unpinned").

The reference calls into `perception_models` (core.vision_encoder), `open_clip`, `sam2` and `torchvision`; none of them is vendored in the
reference or installed where this repo was built, so four pieces of the hot path rest on a reading of their published sources:

    rope        PE's 2-D rotary embedding (core/vision_encoder/rope.py: Rope2D)            product: ovo_amd/encoders/vit.py:rope_tables
    preprocess  the transforms the reference keeps per model card (clip_utils.py:83-84,    product: ViTSpec.resize_mode / interpolation / mean / std
                108-109: Resize / CenterCrop / Normalize of open_clip's / PE's transform)           + ovo_resize_window_normalize
    amg         SAM2AutomaticMaskGenerator's filters (sam2/automatic_mask_generator.py,    product: ovo_amg_mask_stats + sam_amg.py
                sam2/utils/amg.py, torchvision.ops.batched_nms)
    blur        torchvision gaussian_blur(7, 2.5) of depth_filter (geometry_utils.py:92-96) product: k_depth_filter
    state       parameter names / shapes of real checkpoints (clip_utils.py:77-81, 90-110;  product: encoders/*.py:random_state (the names the loaders read)
                segment_utils.py:291-295)

Two steps, in one call when both halves can run:

    collect   (needs the upstream packages; CPU is enough)   python tools/check_upstream.py --out upstream_vectors.npz [--cards ...] [--clip-ckpt CARD=PATH]
                                                             [--pe-ckpt PATH] [--sam2-ckpt PATH]
              drives UPSTREAM code on seeded probes and stores inputs + upstream outputs.  A piece whose package is missing is reported as
              SKIPPED and left out of the file.
    diff      (rope / state: any host; preprocess / amg / blur: an MI355X with libovo_hip.so)    python tools/check_upstream.py --vectors upstream_vectors.npz
              runs THIS repo's product code on the stored inputs and compares.  Exit code 1 if any compared piece differs.

Drop the file at tests/golden/upstream_vectors.npz and tests/test_upstream_vectors.py also checks the oracle against it (the oracle is test
infrastructure: this tool never imports it).  The file holds arrays and strings only -- probes, upstream outputs, parameter names -- no upstream source.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# model cards of the reference (clip_utils.py:53-63, ovo.yaml:45) -> the open_clip hub name the reference loads
OPEN_CLIP_CARDS = {
    "SigLIP": "hf-hub:timm/ViT-SO400M-14-SigLIP", "SigLIP-384": "hf-hub:timm/ViT-SO400M-14-SigLIP-384",
    "SigLIP2-384": "hf-hub:timm/ViT-SO400M-16-SigLIP2-384", "ViT-H-14": "hf-hub:laion/CLIP-ViT-H-14-laion2B-s32B-b79K",
    "ViT-B-16-qg": "hf-hub:apple/DFN2B-CLIP-ViT-B-16", "ViT-L-14-qg": "hf-hub:apple/DFN2B-CLIP-ViT-L-14-39B",
    "ViT-H-14-qg": "hf-hub:apple/DFN5B-CLIP-ViT-H-14", "ViT-H-14-378qg": "hf-hub:apple/DFN5B-CLIP-ViT-H-14-378",
    "PE-Core-L-14-336": "hf-hub:timm/PE-Core-L-14-336",
}
RESULTS = []


def report(piece: str, status: str, detail: str = "") -> None:
    RESULTS.append((piece, status))
    print(f"[{status:7s}] {piece:12s} {detail}")


def probe_frame(h: int = 480, w: int = 640) -> torch.Tensor:
    """A seeded frame [3, h, w] in [0, 1] with structure at several scales (gradients + noise): resampling filters differ visibly on it."""
    g = torch.Generator().manual_seed(7)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    base = torch.stack([xx, yy, (xx + yy) / 2])
    fine = torch.rand(3, h, w, generator=g)
    coarse = torch.nn.functional.interpolate(torch.rand(1, 3, h // 16, w // 16, generator=g), size=(h, w), mode="nearest")[0]
    return (0.4 * base + 0.3 * fine + 0.3 * coarse).clamp(0, 1).contiguous()


def probe_logits(P: int = 64, m: int = 3, h: int = 256, w: int = 256):
    """Seeded mask logits f32 [P, m, h, w] (blobs of varying sharpness: a spread of stability scores) and predicted IoUs f32 [P, m]."""
    g = torch.Generator().manual_seed(11)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    cx, cy = torch.rand(P, m, generator=g) * w, torch.rand(P, m, generator=g) * h
    r = 8 + torch.rand(P, m, generator=g) * 70
    sharp = 0.05 + torch.rand(P, m, generator=g) * 1.5
    d = ((xx[None, None] - cx[..., None, None]) ** 2 + (yy[None, None] - cy[..., None, None]) ** 2).sqrt()
    logits = (r[..., None, None] - d) * sharp[..., None, None] + 0.3 * torch.randn(P, m, h, w, generator=g)
    iou = 0.55 + 0.45 * torch.rand(P, m, generator=g)
    return logits.contiguous(), iou.contiguous()


def probe_depth(h: int = 456, w: int = 616) -> torch.Tensor:
    g = torch.Generator().manual_seed(13)
    d = 1.5 + 0.5 * torch.rand(h, w, generator=g)
    d[torch.rand(h, w, generator=g) < 0.05] = 0.0
    return d


# ------------------------------------------------------------------------------------------------ collect (upstream side)
def collect_rope(out: dict) -> None:
    try:
        from core.vision_encoder import rope as up_rope            # perception_models
    except Exception as e:
        return report("rope", "SKIPPED", f"perception_models not importable ({e!r})")
    heads, grid, hd = 16, 24, 64
    q = torch.randn(1, heads, grid * grid + 1, hd, generator=torch.Generator().manual_seed(0))
    try:
        r = up_rope.Rope2D(hd, use_cls_token=True)
        if hasattr(r, "init_tensors"):
            r.init_tensors()
        r.update_grid("cpu", grid, grid)
        q_rot, _ = r(q.clone(), q.clone())
    except Exception as e:
        return report("rope", "ERROR", f"could not drive core.vision_encoder.rope.Rope2D the way pe.py does (dim={hd}, use_cls_token=True, "
                                       f"update_grid(device, {grid}, {grid}), rope(q, k)): {e!r} -- adapt collect_rope()")
    out["rope_q"], out["rope_q_rotated"], out["rope_grid"] = q.numpy(), q_rot.float().numpy(), np.int64(grid)
    report("rope", "STORED", f"Rope2D(dim={hd}, use_cls_token=True) on a {grid} x {grid} grid, probe q {tuple(q.shape)}")


def _kept(transform):
    from torchvision.transforms import CenterCrop, Normalize, Resize
    return [t for t in transform.transforms if isinstance(t, (Resize, CenterCrop, Normalize))]       # clip_utils.py:83, 108


def collect_preprocess(out: dict, cards) -> None:
    try:
        from torchvision.transforms import Compose
    except Exception as e:
        return report("preprocess", "SKIPPED", f"torchvision not importable ({e!r})")
    frame = probe_frame()
    out["pre_frame"] = frame.numpy()
    done = []
    for card in cards:
        tf = None
        try:
            if card.startswith("PE-Core") and card not in OPEN_CLIP_CARDS:      # load_perception_encoder (clip_utils.py:90-110)
                from core.vision_encoder import transforms as pe_tf
                size = int(card.rsplit("-", 1)[-1]) if card.rsplit("-", 1)[-1].isdigit() else 336
                tf = pe_tf.get_image_transform(size)
            else:                                                                # load_clip_model (clip_utils.py:51-86)
                import open_clip
                _, tf = open_clip.create_model_from_pretrained(OPEN_CLIP_CARDS[card], precision="fp32")
        except Exception as e:
            report("preprocess", "SKIPPED", f"{card}: transform not available ({e!r})")
            continue
        kept = _kept(tf)
        y = Compose(kept)(frame.clone())
        out[f"pre_{card}"] = y.float().numpy()
        out[f"pre_{card}_repr"] = np.asarray(repr(kept))
        done.append(card)
    if done:
        out["pre_cards"] = np.asarray(done)
        report("preprocess", "STORED", f"kept Resize / CenterCrop / Normalize of {done} applied to a 480 x 640 f32 frame")


def collect_amg(out: dict) -> None:
    try:
        from sam2.utils import amg as A
        from torchvision.ops.boxes import batched_nms
    except Exception as e:
        return report("amg", "SKIPPED", f"sam2 / torchvision not importable ({e!r})")
    logits, iou = probe_logits()
    H, W = 480, 640
    pred_iou_thresh, stab_thresh, offset, thr, box_nms_thresh = 0.8, 0.95, 1.0, 0.0, 0.7          # segment_utils.py:296-301 + sam2's defaults
    # SAM2AutomaticMaskGenerator._process_batch on one full-image crop, from the decoder's outputs on
    masks = torch.nn.functional.interpolate(logits, (H, W), mode="bilinear", align_corners=False)   # SAM2Transforms.postprocess_masks
    data = A.MaskData(masks=masks.flatten(0, 1), iou_preds=iou.flatten(0, 1), index=torch.arange(iou.numel()))
    data.filter(data["iou_preds"] > pred_iou_thresh)
    data["stability_score"] = A.calculate_stability_score(data["masks"], thr, offset)
    data.filter(data["stability_score"] >= stab_thresh)
    data["masks"] = data["masks"] > thr
    data["boxes"] = A.batched_mask_to_box(data["masks"])
    keep = ~A.is_box_near_crop_edge(data["boxes"], [0, 0, W, H], [0, 0, W, H])
    if not torch.all(keep):
        data.filter(keep)
    keep = batched_nms(data["boxes"].float(), data["iou_preds"], torch.zeros_like(data["boxes"][:, 0]), iou_threshold=box_nms_thresh)   # _process_crop
    data.filter(keep)
    out.update(amg_logits=logits.numpy(), amg_iou=iou.numpy(), amg_hw=np.asarray([H, W]), amg_params=np.asarray([pred_iou_thresh, stab_thresh, offset, thr, box_nms_thresh]),
               amg_keep_index=data["index"].numpy().astype(np.int64), amg_keep_stability=data["stability_score"].float().numpy(),
               amg_keep_boxes=data["boxes"].numpy().astype(np.int64), amg_keep_masks=np.packbits(data["masks"].numpy(), axis=-1))
    report("amg", "STORED", f"{iou.numel()} candidates -> {len(data['index'])} kept (pred_iou > {pred_iou_thresh}, stability >= {stab_thresh}, box NMS {box_nms_thresh})")


def collect_blur(out: dict) -> None:
    try:
        from torchvision.transforms.functional import gaussian_blur
    except Exception as e:
        return report("blur", "SKIPPED", f"torchvision not importable ({e!r})")
    d = probe_depth()
    low = gaussian_blur(d[None], 7, 2.5)[0]                         # geometry_utils.py:93
    out["blur_depth"], out["blur_low"] = d.numpy(), low.numpy()
    out["blur_filtered"] = torch.where((d - low).abs() > 0.05, torch.tensor(-1.0), d).numpy()     # :94-95
    report("blur", "STORED", f"gaussian_blur(depth[None], 7, 2.5) on a {tuple(d.shape)} depth map with 5 % holes")


def _load_state(path: str) -> dict:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    for k in ("state_dict", "model"):
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    return {k: v for k, v in sd.items() if hasattr(v, "shape")}


def collect_state(out: dict, clip_ckpts, pe_ckpt, sam2_ckpt) -> None:
    n = 0
    for tag, path in [(f"clip:{c}", p) for c, p in clip_ckpts] + ([("pe:PE-Core-L14-336", pe_ckpt)] if pe_ckpt else []) + ([("sam2", sam2_ckpt)] if sam2_ckpt else []):
        try:
            sd = _load_state(path)
        except Exception as e:
            report("state", "ERROR", f"{tag}: cannot load {path} ({e!r})")
            continue
        out[f"state_{tag}_keys"] = np.asarray(list(sd))
        out[f"state_{tag}_shapes"] = np.asarray([",".join(str(int(d)) for d in v.shape) for v in sd.values()])
        n += 1
        report("state", "STORED", f"{tag}: {len(sd)} tensors of {path}")
    if n == 0:
        report("state", "SKIPPED", "no checkpoint given (--clip-ckpt CARD=PATH, --pe-ckpt PATH, --sam2-ckpt PATH)")


# ------------------------------------------------------------------------------------------------ diff (this repo's side)
def _have_gpu() -> bool:
    try:
        from ovo_amd import _lib as L
        return torch.cuda.is_available() and L.load() is not None
    except Exception:
        return False


def diff_rope(v) -> None:
    if "rope_q" not in v:
        return
    from ovo_amd.encoders.vit import SPECS, rope_tables
    spec = SPECS["PE-Core-L14-336"]
    q, want = torch.from_numpy(v["rope_q"]), torch.from_numpy(v["rope_q_rotated"])
    errs = {}
    for off in (1, 0):
        for order in ("xy", "yx"):
            cos, sin = rope_tables(spec, cls_offset=off, axis_order=order)
            x0, x1 = q[..., 0::2], q[..., 1::2]
            got = q * cos + torch.stack([-x1, x0], dim=-1).flatten(-2) * sin
            errs[(off, order)] = float((got - want).abs().max())
    default = (spec.rope_cls_offset, spec.rope_axis_order)
    best = min(errs, key=errs.get)
    ok = errs[default] < 1e-4
    report("rope", "PASS" if ok else "FAIL", f"default convention {default}: max |diff| {errs[default]:.2e}; all four: {errs}" +
           ("" if ok else f" -> best {best}: set ViTSpec.rope_cls_offset / rope_axis_order of the PE cards (ovo_amd/encoders/vit.py)" if errs[best] < 1e-4
            else " -> NO convention matches: the pair layout / frequency schedule of rope_tables differs from upstream (see tools/check_rope.py's list)"))


_CARD_TO_SPEC = {"PE-Core-L-14-336": "PE-Core-L14-336"}


def diff_preprocess(v) -> None:
    if "pre_cards" not in v:
        return
    if not _have_gpu():
        return report("preprocess", "SKIPPED", "needs an MI355X with libovo_hip.so (the product's resize is a HIP kernel): run --vectors there")
    from ovo_amd.encoders.vit import SPECS, HipViT
    frame = torch.from_numpy(v["pre_frame"]).cuda()
    bad = []
    for card in [str(c) for c in v["pre_cards"]]:
        name = _CARD_TO_SPEC.get(card, card)
        if name not in SPECS:
            bad.append(f"{card}: no ViTSpec")
            continue
        spec = SPECS[name]
        want = torch.from_numpy(v[f"pre_{card}"])
        enc = HipViT.__new__(HipViT)                                 # the preprocessing needs the spec only, not the weights
        enc.spec = spec
        got = enc.preprocess_clip(frame[None], scale=1.0)[0].cpu()
        if tuple(got.shape) != tuple(want.shape):
            bad.append(f"{card}: shape {tuple(got.shape)} vs upstream {tuple(want.shape)} ({v[f'pre_{card}_repr']})")
            continue
        err = float((got - want).abs().max())
        if err > 2e-3:                                              # normalised units; a different filter / crop mode is off by 1e-1 and more
            bad.append(f"{card}: max |diff| {err:.2e} with resize_mode={spec.resize_mode} interpolation={spec.interpolation}; upstream keeps {v[f'pre_{card}_repr']}")
    report("preprocess", "FAIL" if bad else "PASS", "; ".join(bad) if bad else f"{len(v['pre_cards'])} cards within 2e-3")


def diff_amg(v) -> None:
    if "amg_logits" not in v:
        return
    if not _have_gpu():
        return report("amg", "SKIPPED", "needs an MI355X with libovo_hip.so: run --vectors there")
    from ovo_amd.entities.sam_amg import HipSam2AutomaticMaskGenerator
    p = v["amg_params"]
    gen = HipSam2AutomaticMaskGenerator(None, None, pred_iou_thresh=float(p[0]), stability_score_thresh=float(p[1]))
    gen.stability_score_offset, gen.mask_threshold, gen.box_nms_thresh = float(p[2]), float(p[3]), float(p[4])
    H, W = (int(x) for x in v["amg_hw"])
    r = gen.generate_finish(gen.stats_launch(torch.from_numpy(v["amg_logits"]).cuda(), torch.from_numpy(v["amg_iou"]).cuda(), H, W))
    got, want = r["index"].astype(np.int64), v["amg_keep_index"]    # flat candidate index (click * masks-per-click + mask), in kept order
    masks = np.unpackbits(v["amg_keep_masks"], axis=-1)[..., :W].astype(bool)
    same_points = list(got) == list(want)
    got_masks = r["masks"].cpu().numpy().astype(bool)
    iou_ok = len(got_masks) == len(masks) and all(((a & b).sum() + 1) / ((a | b).sum() + 1) > 0.999 for a, b in zip(got_masks, masks))
    ok = same_points and iou_ok and np.allclose(r["stability_score"], v["amg_keep_stability"], atol=2e-3)
    report("amg", "PASS" if ok else "FAIL", f"kept {len(got)} vs upstream {len(want)}; same candidates in the same order: {same_points}; masks identical to 0.1 %: {iou_ok}")


def diff_blur(v) -> None:
    if "blur_depth" not in v:
        return
    if not _have_gpu():
        return report("blur", "SKIPPED", "needs an MI355X with libovo_hip.so: run --vectors there")
    from ovo_amd.utils import geometry_utils as G
    d = torch.from_numpy(v["blur_depth"]).cuda()
    got = G.depth_filter(d).cpu().numpy()
    want = v["blur_filtered"]
    flips = int((got != want).sum())
    near = int((np.abs(np.abs(v["blur_depth"] - v["blur_low"]) - 0.05) < 1e-5).sum())     # pixels whose high-pass sits on the threshold to an ulp
    report("blur", "PASS" if flips <= near else "FAIL", f"{flips} of {got.size} pixels filtered differently ({near} sit on the 0.05 threshold to 1e-5)")


def compare_state(expected: dict, keys, shapes, prefix: str = "") -> dict:
    """Names / shapes a loader of this repo reads (`expected`: name -> shape tuple) against a real checkpoint's (keys, 'a,b,c' shape strings).
    `prefix` ("visual.", "image_encoder."): when the checkpoint has tensors under it, ONLY those are compared, with the prefix removed (the
    other tower of a CLIP checkpoint re-uses the same names); a checkpoint without the prefix is taken whole."""
    pairs = [(str(k), str(s)) for k, s in zip(keys, shapes)]
    if prefix and any(k.startswith(prefix) for k, _ in pairs):
        pairs = [(k[len(prefix):], s) for k, s in pairs if k.startswith(prefix)]
    have = {k: (tuple(int(x) for x in s.split(",")) if s else ()) for k, s in pairs}
    missing = sorted(k for k in expected if k not in have)
    wrong = sorted(f"{k}: {have[k]} vs expected {tuple(expected[k])}" for k in expected if k in have and tuple(have[k]) != tuple(expected[k]))
    unread = sorted(k for k in have if k not in expected)
    return {"missing": missing, "wrong_shape": wrong, "unread": unread}


def diff_state(v) -> None:
    tags = sorted({k[len("state_"):-len("_keys")] for k in v.files if k.startswith("state_") and k.endswith("_keys")})
    for tag in tags:
        keys, shapes = v[f"state_{tag}_keys"], v[f"state_{tag}_shapes"]
        if tag.startswith("clip:") or tag.startswith("pe:"):
            from ovo_amd.encoders.vit import SPECS, random_state
            card = tag.split(":", 1)[1]
            name = _CARD_TO_SPEC.get(card, card)
            if name not in SPECS:
                report("state", "FAIL", f"{tag}: no ViTSpec named {name}")
                continue
            exp = {k: tuple(t.shape) for k, t in random_state(SPECS[name]).items()}
            r = compare_state(exp, keys, shapes, prefix="visual.")
        else:
            from ovo_amd.encoders import hiera, sam_decoder
            n_blocks = len({str(k).split(".")[2] for k in keys if str(k).startswith("image_encoder.trunk.blocks.") or str(k).startswith("trunk.blocks.")})
            spec = next((s for s in hiera.SPECS.values() if sum(s.stages) == n_blocks and s.image_size == 1024), None)
            if spec is None:
                report("state", "FAIL", f"{tag}: no HieraSpec with {n_blocks} blocks")
                continue
            exp = {k: tuple(t.shape) for k, t in hiera.random_state(spec).items()}
            exp.update({k: tuple(t.shape) for k, t in sam_decoder.random_state(sam_decoder.SPECS["sam2"]).items()})
            exp = {("image_encoder." + k if k.startswith(("trunk.", "neck.")) else k): s_ for k, s_ in exp.items()} if any(str(k).startswith("image_encoder.") for k in keys) else exp
            r = compare_state(exp, keys, shapes)
            r["unread"] = [k for k in r["unread"] if not k.startswith(("memory_", "obj_ptr", "maskmem", "no_mem", "no_obj", "mask_downsample", "sam_mask_decoder.pred_obj"))]
        ok = not r["missing"] and not r["wrong_shape"]
        report("state", "PASS" if ok else "FAIL", f"{tag}: {len(r['missing'])} names the loader reads are missing {r['missing'][:6]}, "
                                                  f"{len(r['wrong_shape'])} shapes differ {r['wrong_shape'][:4]}, {len(r['unread'])} checkpoint tensors unread {r['unread'][:6]}")


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=None, help="collect: write upstream vectors here (.npz)")
    ap.add_argument("--vectors", default=None, help="diff: vectors of an earlier collect (default: what this call collected)")
    ap.add_argument("--cards", default=",".join(list(OPEN_CLIP_CARDS) + ["PE-Core-L14-336"]),
                    help="model cards whose kept transform is collected (open_clip cards of clip_utils.py:53-63 + the PE card of ovo.yaml:45)")
    ap.add_argument("--clip-ckpt", action="append", default=[], metavar="CARD=PATH", help="an open_clip checkpoint (state dict) of that card")
    ap.add_argument("--pe-ckpt", default=None, help="PE-Core-L14-336.pt")
    ap.add_argument("--sam2-ckpt", default=None, help="sam2.1_hiera_*.pt")
    a = ap.parse_args()
    path = a.vectors
    if path is None:
        out: dict = {}
        collect_rope(out)
        collect_preprocess(out, [c for c in a.cards.split(",") if c])
        collect_amg(out)
        collect_blur(out)
        collect_state(out, [tuple(x.split("=", 1)) for x in a.clip_ckpt], a.pe_ckpt, a.sam2_ckpt)
        if not out:
            print("nothing collected: none of perception_models / open_clip / sam2 / torchvision is importable here and no checkpoint was given")
            return 0
        path = a.out or "upstream_vectors.npz"
        np.savez_compressed(path, **out)
        print(f"wrote {path} ({os.path.getsize(path) // 1024} KiB): copy it to tests/golden/upstream_vectors.npz for the test suite")
        RESULTS.clear()
    v = np.load(path, allow_pickle=False)
    diff_rope(v)
    diff_preprocess(v)
    diff_amg(v)
    diff_blur(v)
    diff_state(v)
    failed = [p for p, s in RESULTS if s == "FAIL"]
    print("summary:", ", ".join(f"{p} {s}" for p, s in RESULTS) or "nothing to compare")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
