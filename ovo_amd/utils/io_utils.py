"""On-disk formats of the reference: run-length masks, instance / label prediction files, checkpoints.

Mirror of the format half of `ovo/utils/io_utils.py` (:127-235): same function names, same bytes on disk (pinned by
tests/golden/io_formats.npz, written by the reference's own functions).  Dataset loaders (:64-125) read third-party
files through Open3D / plyfile and stay out of scope.  Pure host code: these files are written once per
scene, after the GPU work.
"""
from __future__ import annotations

import json
import math
import os
from pathlib import Path
from typing import Any, Dict, Union

import numpy as np
import torch


def rle_encode(mask: np.ndarray) -> Dict[str, Any]:
    """Reference: io_utils.py:127-141.  1-D binary mask -> {"length", "counts": "start len start len ..."} (1-based starts)."""
    mask = np.asarray(mask)
    length = mask.shape[0]
    padded = np.concatenate([[0], mask, [0]])
    runs = np.where(padded[1:] != padded[:-1])[0] + 1
    runs[1::2] -= runs[::2]
    return dict(length=length, counts=" ".join(str(x) for x in runs))


def rle_decode(rle: Dict[str, Any]) -> np.ndarray:
    """Reference: io_utils.py:143-160."""
    s = rle["counts"].split()
    starts, nums = (np.asarray(x, dtype=np.int32) for x in (s[0::2], s[1::2]))
    starts = starts - 1
    mask = np.zeros(rle["length"], dtype=np.uint8)
    for lo, hi in zip(starts, starts + nums):
        mask[lo:hi] = 1
    return mask


def write_instances(experiment_path: str, scene_name: str, instances_info: Dict[str, Any]) -> None:
    """Reference: io_utils.py:162-184 (ScanNet instance-benchmark layout): `<scene>.txt` with one
    `./predicted_masks/<scene>_<i>.json <label> <conf:.4f>` line per instance, each json an RLE of the per-vertex mask."""
    save_path = os.path.join(experiment_path, "instance_pred")
    rel_path = "./predicted_masks/"
    os.makedirs(os.path.join(save_path, rel_path), exist_ok=True)
    masks, classes, conf = (instances_info[k] for k in ("masks", "classes", "conf"))
    if isinstance(masks, torch.Tensor):
        masks = masks.cpu().numpy()
    n = len(masks)
    n_digits = math.trunc(math.log(n, 10)) + 1
    lines = []
    for i in range(n):
        mask_file = os.path.join(rel_path, f"{scene_name}_{str(i).zfill(n_digits)}.json")
        with open(os.path.join(save_path, mask_file), "w") as f:
            json.dump(rle_encode(np.asarray(masks[i])), f)
        lines.append(f"{mask_file} {int(classes[i])} {float(conf[i]):.4f}")
    with open(os.path.join(save_path, f"{scene_name}.txt"), "w") as f:
        f.write("\n".join(lines))


def write_labels(output_file: str, pcd_labels) -> None:
    """Reference: io_utils.py:186-190: one integer label per map vertex per line."""
    if isinstance(pcd_labels, torch.Tensor):
        pcd_labels = pcd_labels.cpu().numpy()
    with open(output_file, "w") as f:
        f.write("\n".join(str(int(v)) for v in np.asarray(pcd_labels).reshape(-1)))


def read_labels(output_file: str) -> np.ndarray:
    """Reference: io_utils.py:192-196."""
    with open(output_file, "r") as f:
        return np.array(f.read().splitlines()).astype(np.int64)


def save_dict_to_ckpt(dictionary: Dict[str, Any], file_name: str, *, directory: Union[str, Path]) -> None:
    """Reference: io_utils.py:198-227 (mkdir decorator + torch.save): `ovo_map.ckpt` = {"map_params", "ovo_map_params"}."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    try:
        torch.save(dictionary, directory / file_name, _use_new_zipfile_serialization=False)
    except OverflowError:                                         # > 4 GiB payloads need pickle protocol 4 (io_utils.py:224-227)
        torch.save(dictionary, directory / file_name, pickle_protocol=4)


def save_dict_to_yaml(dictionary: Dict[str, Any], file_name: str, *, directory: Union[str, Path]) -> None:
    """Reference: io_utils.py:229-235."""
    import yaml
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    with open(directory / file_name, "w") as f:
        yaml.dump(dictionary, f)


# ---- layered YAML configuration (io_utils.py:13-61; merged in run_eval.py:66-84) -------------------------------
def update_recursive(dict1: Dict[str, Any], dict2: Dict[str, Any]) -> None:
    """Reference: io_utils.py:42-61.  In-place deep merge of dict2 into dict1: nested dicts merge key by key, anything
    else overwrites.  Like the reference, a key of dict2 that dict1 lacks is created as an empty dict first, so merging a
    dict into a scalar raises (AttributeError / TypeError) instead of silently replacing it."""
    for key, value in dict2.items():
        if key not in dict1:
            dict1[key] = {}
        if isinstance(value, dict):
            update_recursive(dict1[key], value)
        else:
            dict1[key] = value


def load_config(path: str, default_path: str = None, inherit: bool = True) -> Dict[str, Any]:
    """Reference: io_utils.py:13-40.  The file at `path`, laid over its `inherit_from` chain (each link itself laid over
    `default_path`) or, without `inherit_from`, over `default_path`."""
    import yaml
    with open(path, "r") as f:
        special = yaml.full_load(f)
    base: Dict[str, Any] = {}
    parent = special.get("inherit_from")
    if parent is not None and inherit:
        base = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, "r") as f:
            base = yaml.full_load(f)
    update_recursive(base, special)
    return base
