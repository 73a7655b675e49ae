"""tools/check_upstream.py -- the one maintainer-side run that pins the pieces restated from un-vendored upstream packages.

Offline there is no upstream, so these tests (a) exercise the tool's DIFF half on a vector file in which the oracle's restatements stand in for
upstream (the product must agree with them: the same comparisons the maintainer's real run makes), (b) check the state-dict coverage report on
the names the loaders read, and (c) when a real `tests/golden/upstream_vectors.npz` has been dropped in, hold the ORACLE to it as well."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def _tool():
    spec = importlib.util.spec_from_file_location("check_upstream", os.path.join(ROOT, "tools", "check_upstream.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _standin_vectors(path):
    """The tool's probes through the ORACLE (in the place of upstream), in the tool's file format."""
    from oracle import geometry as OG, sam2_amg, vit as OV
    from ovo_amd.encoders.vit import SPECS
    cu = _tool()
    out = {}
    spec = SPECS["PE-Core-L14-336"]
    q = torch.randn(1, spec.heads, spec.tokens, spec.width // spec.heads, generator=torch.Generator().manual_seed(0))
    cos, sin = OV.rope_for(spec)
    out["rope_q"], out["rope_q_rotated"], out["rope_grid"] = q.numpy(), OV._rope(q, cos, sin).numpy(), np.int64(spec.grid)
    frame = cu.probe_frame()
    out["pre_frame"] = frame.numpy()
    cards = ["ViT-L-14-qg", "SigLIP-384", "PE-Core-L14-336"]
    for c in cards:
        s = SPECS[c]
        out[f"pre_{c}"] = OV.open_clip_preprocess(frame, s.image_size, s.mean, s.std, s.resize_mode, s.interpolation).numpy()
        out[f"pre_{c}_repr"] = np.asarray(f"oracle stand-in: {s.resize_mode} {s.interpolation}")
    out["pre_cards"] = np.asarray(cards)
    logits, iou = cu.probe_logits(P=24)
    r = sam2_amg.amg_postprocess(logits.numpy(), iou.numpy(), 480, 640)
    out.update(amg_logits=logits.numpy(), amg_iou=iou.numpy(), amg_hw=np.asarray([480, 640]), amg_params=np.asarray([0.8, 0.95, 1.0, 0.0, 0.7]),
               amg_keep_index=r["index"].astype(np.int64), amg_keep_stability=r["stability_score"], amg_keep_boxes=r["boxes_xyxy"].astype(np.int64),
               amg_keep_masks=np.packbits(r["masks"], axis=-1))
    d = cu.probe_depth().numpy()
    filt = OG.depth_filter(d)
    out["blur_depth"], out["blur_filtered"] = d, filt
    out["blur_low"] = np.where(filt == d, d, d + 1.0).astype(np.float32)     # (only used to count threshold-straddling pixels: none claimed here)
    np.savez_compressed(path, **out)
    return cu


def test_state_dict_coverage_report():
    """`compare_state`: every name a loader reads must exist with its shape; what the checkpoint holds beyond that is listed, not an error."""
    from ovo_amd.encoders import hiera, vit
    cu = _tool()
    sd = vit.random_state(vit.SPECS["tiny-pe"])
    exp = {k: tuple(v.shape) for k, v in sd.items()}
    keys = ["visual." + k for k in sd] + ["logit_scale", "transformer.resblocks.0.ln_1.weight"]
    shapes = [",".join(str(d) for d in v.shape) for v in sd.values()] + ["", "64"]
    r = cu.compare_state(exp, keys, shapes, prefix="visual.")
    assert r == {"missing": [], "wrong_shape": [], "unread": []}            # the text tower's same-named tensors are not compared
    del_key = next(iter(sd))
    i = list(sd).index("ln_post.weight")
    shapes2 = list(shapes)
    shapes2[i] = "7"
    r = cu.compare_state(exp, [k for k in keys if k != "visual." + del_key], [s for k, s in zip(keys, shapes2) if k != "visual." + del_key], prefix="visual.")
    assert r["missing"] == [del_key] and len(r["wrong_shape"]) == 1 and r["wrong_shape"][0].startswith("ln_post.weight")
    hs = hiera.random_state(hiera.SPECS["hiera_test"])
    r = cu.compare_state({k: tuple(v.shape) for k, v in hs.items()}, ["image_encoder." + k for k in hs], [",".join(str(d) for d in v.shape) for v in hs.values()],
                         prefix="image_encoder.")
    assert r == {"missing": [], "wrong_shape": [], "unread": []}


def test_tool_reports_nothing_to_collect_offline(capsys):
    """Without the upstream packages the collect half skips every piece and writes nothing (no crash, exit code 0)."""
    import sys
    cu = _tool()
    argv, sys.argv = sys.argv, ["check_upstream.py"]
    try:
        rc = cu.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert rc == 0 and "nothing collected" in out and out.count("SKIPPED") >= 4


def test_diff_rope_on_host(tmp_path):
    cu = _standin_vectors(str(tmp_path / "v.npz"))
    cu.RESULTS.clear()
    cu.diff_rope(np.load(str(tmp_path / "v.npz")))
    assert cu.RESULTS == [("rope", "PASS")]


@pytest.mark.gpu
def test_diff_half_against_the_product(tmp_path):
    """preprocess / amg / blur of the tool's diff half on the GPU: the product agrees with the stand-in (oracle) vectors, and a corrupted
    vector file is reported as a failure (the comparisons can fail)."""
    cu = _standin_vectors(str(tmp_path / "v.npz"))
    v = np.load(str(tmp_path / "v.npz"))
    cu.RESULTS.clear()
    cu.diff_preprocess(v)
    cu.diff_amg(v)
    cu.diff_blur(v)
    assert cu.RESULTS == [("preprocess", "PASS"), ("amg", "PASS"), ("blur", "PASS")], cu.RESULTS
    bad = {k: v[k] for k in v.files}
    bad["pre_ViT-L-14-qg"] = bad["pre_SigLIP-384"][:, :224, :224].copy()      # another card's mode
    bad["amg_keep_index"] = bad["amg_keep_index"][::-1].copy()
    bad["blur_filtered"] = np.where(bad["blur_filtered"] < 0, bad["blur_depth"], -1).astype(np.float32)
    np.savez_compressed(str(tmp_path / "bad.npz"), **bad)
    b = np.load(str(tmp_path / "bad.npz"))
    cu.RESULTS.clear()
    cu.diff_preprocess(b)
    cu.diff_amg(b)
    cu.diff_blur(b)
    assert [s for _, s in cu.RESULTS] == ["FAIL", "FAIL", "FAIL"], cu.RESULTS


REAL = os.path.join(GOLDEN, "upstream_vectors.npz")


@pytest.mark.skipif(not os.path.exists(REAL), reason="no tests/golden/upstream_vectors.npz (made by tools/check_upstream.py where the upstream packages exist)")
def test_oracle_against_real_upstream_vectors():
    """With a real vector file present the ORACLE is pinned to upstream as well (the product is pinned to the oracle by the -m gpu tests)."""
    from oracle import geometry as OG, sam2_amg, vit as OV
    from ovo_amd.encoders.vit import SPECS
    v = np.load(REAL)
    if "rope_q" in v:
        spec = SPECS["PE-Core-L14-336"]
        cos, sin = OV.rope_for(spec)
        np.testing.assert_allclose(OV._rope(torch.from_numpy(v["rope_q"]), cos, sin).numpy(), v["rope_q_rotated"], atol=1e-4)
    for c in (str(x) for x in v["pre_cards"]) if "pre_cards" in v else ():
        s = SPECS[{"PE-Core-L-14-336": "PE-Core-L14-336"}.get(c, c)]
        got = OV.open_clip_preprocess(torch.from_numpy(v["pre_frame"]), s.image_size, s.mean, s.std, s.resize_mode, s.interpolation).numpy()
        np.testing.assert_allclose(got, v[f"pre_{c}"], atol=2e-4, err_msg=c)
    if "amg_logits" in v:
        p = v["amg_params"]
        r = sam2_amg.amg_postprocess(v["amg_logits"], v["amg_iou"], int(v["amg_hw"][0]), int(v["amg_hw"][1]), float(p[0]), float(p[1]), float(p[2]), float(p[3]), float(p[4]))
        assert list(r["index"]) == list(v["amg_keep_index"])
    if "blur_depth" in v:
        got = OG.depth_filter(v["blur_depth"])
        near = np.abs(np.abs(v["blur_depth"] - v["blur_low"]) - 0.05) < 1e-5
        assert ((got != v["blur_filtered"]) & ~near).sum() == 0
