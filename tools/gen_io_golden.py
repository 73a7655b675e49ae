"""Golden vectors for the on-disk formats made by RUNNING the reference's own writers
(ovo/utils/io_utils.py: rle_encode / rle_decode / write_instances / write_labels / read_labels):

    python tools/gen_io_golden.py [--reference /root/reference] [--out DIR]   ->  DIR/io_formats.npz (default tests/golden/; inputs + the bytes
                                                                                  the reference wrote)
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(reference="/root/reference", out_dir=None):
    import gen_golden
    gen_golden.install_stubs()
    for name in ("plyfile", "yaml"):
        try:
            __import__(name)
        except ImportError:
            gen_golden._stub(name)
    sys.path.insert(0, reference)
    from ovo.utils import io_utils as IO
    rng = np.random.default_rng(5)
    n_pts, n_inst = 5000, 12
    masks = np.zeros((n_inst, n_pts), np.uint8)
    for i in range(n_inst):
        for _ in range(rng.integers(1, 6)):
            a = rng.integers(0, n_pts - 50)
            masks[i, a:a + rng.integers(1, 400)] = 1
    masks[3] = 0                                   # an empty instance
    masks[4] = 1                                   # a full one
    masks[5, 0] = 1; masks[5, -1] = 1              # runs touching both ends
    classes = rng.integers(0, 40, n_inst)
    conf = rng.uniform(0, 1, n_inst).astype(np.float32)
    labels = rng.integers(-1, 40, n_pts)
    arrays = {"masks": np.packbits(masks, axis=1), "n_pts": np.int64(n_pts), "classes": classes, "conf": conf, "labels": labels}
    rles = [IO.rle_encode(m) for m in masks]
    arrays["rle_counts"] = np.asarray([r["counts"] for r in rles])
    arrays["rle_length"] = np.asarray([r["length"] for r in rles])
    for i, r in enumerate(rles):
        assert np.array_equal(IO.rle_decode(r), masks[i])
    with tempfile.TemporaryDirectory() as d:
        IO.write_instances(d, "scene0042_00", {"masks": masks, "classes": classes, "conf": conf})
        files = {}
        for root, _, names in os.walk(d):
            for nme in names:
                p = os.path.join(root, nme)
                files[os.path.relpath(p, d)] = open(p).read()
        arrays["inst_paths"] = np.asarray(sorted(files))
        arrays["inst_texts"] = np.asarray([files[k] for k in sorted(files)])
        IO.write_labels(os.path.join(d, "labels.txt"), labels)
        arrays["labels_text"] = np.asarray(open(os.path.join(d, "labels.txt")).read())
        assert np.array_equal(IO.read_labels(os.path.join(d, "labels.txt")), labels)
    # layered YAML configuration (io_utils.py:13-61) and the logger's text files (logger.py:85-96), by the reference's own code
    yamls = {"base.yaml": "semantic:\n  segment_every: 10\n  sam:\n    points_per_side: 16\n    nms_iou_th: 0.8\n  clip:\n    embed_type: learned\n    k_top_views: 10\ndata:\n  input_path: /data\n",
             "slam.yaml": "inherit_from: {d}/base.yaml\nslam:\n  module: vanilla\nsemantic:\n  sam:\n    points_per_side: 32\n  clip:\n    embed_type: TextRegion\n",
             "scene.yaml": "inherit_from: {d}/slam.yaml\ndata:\n  scene_name: scene0011_00\n  frame_limit: -1\nsemantic:\n  track_th: 100\ncam:\n  H: 480\n  W: 640\n",
             "default.yaml": "vis:\n  stream: false\nsemantic:\n  segment_every: 5\n  log: true\n",
             "plain.yaml": "semantic:\n  segment_every: 2\n"}
    gen_golden._stub("wandb")
    from ovo.entities.logger import Logger
    with tempfile.TemporaryDirectory() as d:
        for name, text in yamls.items():
            open(os.path.join(d, name), "w").write(text.replace("{d}", d))
        merged = {"chain": IO.load_config(os.path.join(d, "scene.yaml")),
                  "chain_default": IO.load_config(os.path.join(d, "scene.yaml"), os.path.join(d, "default.yaml")),
                  "no_inherit": IO.load_config(os.path.join(d, "scene.yaml"), os.path.join(d, "default.yaml"), inherit=False),
                  "plain_default": IO.load_config(os.path.join(d, "plain.yaml"), os.path.join(d, "default.yaml"))}
        arrays["cfg_names"] = np.asarray(sorted(yamls))
        arrays["cfg_texts"] = np.asarray([yamls[k] for k in sorted(yamls)])
        arrays["cfg_merged_json"] = np.asarray(json.dumps(merged, sort_keys=True).replace(d, "{d}"))
        lg = Logger(os.path.join(d, "run"))
        for i in range(4):
            lg.log_ovo_stats({"frame_id": 10 * i, "t_sam": 0.125 * (i + 1), "t_obj": 1e-3 * i, "n_obj": [i, 2 * i], "n_matches": 3 * i,
                              "t_up": 0.5, "t_clip": 1.0 / (i + 3)})
            lg.log_fps(30.0 / (i + 1))
            lg.log_spf(0.01 * i)
            lg.stats["ram"].append(1.5 + i)
            lg.stats["vram"].append(0.25 * i)
        lg.stats["max_vram"], lg.stats["max_ram"] = [0.75], [float(np.asarray(lg.stats["ram"]).max())]
        lg.write_stats()
        logs = {n: open(os.path.join(d, "run", "logger", n)).read() for n in sorted(os.listdir(os.path.join(d, "run", "logger"))) if n.endswith(".log")}
        arrays["log_names"], arrays["log_texts"] = np.asarray(list(logs)), np.asarray(list(logs.values()))
        arrays["log_dirs"] = np.asarray(sorted(n for n in os.listdir(os.path.join(d, "run", "logger")) if not n.endswith(".log")))
    out = os.path.join(out_dir or os.path.join(ROOT, "tests", "golden"), "io_formats.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, **arrays)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB;", len(files), "files from write_instances")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=None, help="directory for io_formats.npz (default: tests/golden/ of this checkout)")
    a = ap.parse_args()
    main(a.reference, a.out)
