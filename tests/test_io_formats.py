"""CPU: on-disk formats byte-for-byte against files written by the reference's own io_utils
(tests/golden/io_formats.npz, tools/gen_io_golden.py)."""
import os

import numpy as np
import torch

from conftest import golden


def _masks(d):
    return np.unpackbits(d["masks"], axis=1)[:, :int(d["n_pts"])]


def test_rle_matches_reference():
    from ovo_amd.utils import io_utils as IO
    d = golden("io_formats")
    masks = _masks(d)
    for i, m in enumerate(masks):
        r = IO.rle_encode(m)
        assert r["counts"] == str(d["rle_counts"][i]) and r["length"] == int(d["rle_length"][i])
        assert np.array_equal(IO.rle_decode({"length": int(d["rle_length"][i]), "counts": str(d["rle_counts"][i])}), m)
    assert IO.rle_encode(masks[3])["counts"] == "" and IO.rle_decode(IO.rle_encode(masks[3])).sum() == 0      # empty mask


def test_instance_and_label_files_match_reference(tmp_path):
    from ovo_amd.utils import io_utils as IO
    d = golden("io_formats")
    IO.write_instances(str(tmp_path), "scene0042_00", {"masks": torch.from_numpy(_masks(d)), "classes": d["classes"], "conf": d["conf"]})
    got = {}
    for root, _, names in os.walk(tmp_path):
        for n in names:
            p = os.path.join(root, n)
            got[os.path.relpath(p, tmp_path)] = open(p).read()
    assert sorted(got) == [str(p) for p in d["inst_paths"]]
    for p, text in zip(d["inst_paths"], d["inst_texts"]):
        assert got[str(p)] == str(text), p
    out = tmp_path / "labels.txt"
    IO.write_labels(str(out), torch.from_numpy(d["labels"]))
    assert out.read_text() == str(d["labels_text"])
    assert np.array_equal(IO.read_labels(str(out)), d["labels"])


def test_checkpoint_round_trip(tmp_path):
    """ovo_map.ckpt = {"map_params", "ovo_map_params"} (ovomapping.py:81-100): written by save_dict_to_ckpt, read by torch.load."""
    from ovo_amd.utils import io_utils as IO
    ckpt = {"map_params": {"xyz": torch.randn(50, 3), "obj_ids": torch.randint(-1, 4, (50, 1), dtype=torch.int32), "max_id": 50},
            "ovo_map_params": {"ins_3d_ids": np.arange(4), "ins3d_0_clip_feature": torch.randn(1, 16)}}
    IO.save_dict_to_ckpt(ckpt, "ovo_map.ckpt", directory=tmp_path / "a" / "b")
    back = torch.load(tmp_path / "a" / "b" / "ovo_map.ckpt", map_location="cpu", weights_only=False)
    assert torch.equal(back["map_params"]["xyz"], ckpt["map_params"]["xyz"]) and back["map_params"]["max_id"] == 50
    assert np.array_equal(back["ovo_map_params"]["ins_3d_ids"], np.arange(4))
    IO.save_dict_to_yaml({"a": 1, "b": {"c": [1, 2]}}, "cfg.yaml", directory=tmp_path)
    import yaml
    assert yaml.safe_load((tmp_path / "cfg.yaml").read_text()) == {"a": 1, "b": {"c": [1, 2]}}
