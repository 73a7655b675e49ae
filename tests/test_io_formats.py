"""CPU: on-disk formats byte-for-byte against files written by the reference's own io_utils
(tests/golden/io_formats.npz, tools/gen_io_golden.py)."""
import os

import numpy as np
import pytest
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


def test_layered_config_matches_reference(tmp_path):
    """load_config / update_recursive (io_utils.py:13-61): `inherit_from` chains, default file, inherit=False -- merged dicts equal to
    the ones the reference's own loader produced from the same YAML files."""
    import json
    from ovo_amd.utils import io_utils as IO
    d = golden("io_formats")
    for name, text in zip(d["cfg_names"], d["cfg_texts"]):
        (tmp_path / str(name)).write_text(str(text).replace("{d}", str(tmp_path)))
    want = json.loads(str(d["cfg_merged_json"]).replace("{d}", str(tmp_path)))
    p = lambda n: str(tmp_path / n)
    got = {"chain": IO.load_config(p("scene.yaml")), "chain_default": IO.load_config(p("scene.yaml"), p("default.yaml")),
           "no_inherit": IO.load_config(p("scene.yaml"), p("default.yaml"), inherit=False),
           "plain_default": IO.load_config(p("plain.yaml"), p("default.yaml"))}
    assert got == want
    assert got["chain"]["semantic"]["sam"] == {"points_per_side": 32, "nms_iou_th": 0.8} and got["chain"]["semantic"]["clip"]["embed_type"] == "TextRegion"
    a = {"x": 1}
    with pytest.raises((AttributeError, TypeError)):             # a dict laid over a scalar fails upstream too
        IO.update_recursive(a, {"x": {"y": 2}})


def test_logger_files_match_reference(tmp_path):
    """Logger.write_stats (logger.py:85-96): the same file set and bytes as the reference's logger fed the same statistics."""
    from ovo_amd.entities.logger import Logger
    d = golden("io_formats")
    lg = Logger(str(tmp_path / "run"))
    for i in range(4):
        lg.log_ovo_stats({"frame_id": 10 * i, "t_sam": 0.125 * (i + 1), "t_obj": 1e-3 * i, "n_obj": [i, 2 * i], "n_matches": 3 * i,
                          "t_up": 0.5, "t_clip": 1.0 / (i + 3)})
        lg.log_fps(30.0 / (i + 1))
        lg.log_spf(0.01 * i)
        lg.stats["ram"].append(1.5 + i)
        lg.stats["vram"].append(0.25 * i)
    lg.stats["max_vram"], lg.stats["max_ram"] = [0.75], [float(np.asarray(lg.stats["ram"]).max())]
    lg.write_stats()
    folder = tmp_path / "run" / "logger"
    names = sorted(n.name for n in folder.iterdir() if n.name.endswith(".log"))
    assert names == [str(n) for n in d["log_names"]]
    for n, text in zip(d["log_names"], d["log_texts"]):
        assert (folder / str(n)).read_text() == str(text), n
    assert sorted(n.name for n in folder.iterdir() if not n.name.endswith(".log")) == [str(n) for n in d["log_dirs"]]
    with pytest.raises(KeyError):
        lg.log_ovo_stats({"not_a_stat": 1})
    lg.log_memory_usage(0)                                       # no GPU here: RAM only
    lg.log_max_memory_usage()
    assert lg.stats["max_ram"][0] >= 4.5 and len(lg.stats["vram"]) == 5
    lg.print_final_stats()
