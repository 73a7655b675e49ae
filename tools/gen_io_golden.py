"""Golden vectors for the on-disk formats made by RUNNING the reference's own writers
(ovo/utils/io_utils.py: rle_encode / rle_decode / write_instances / write_labels / read_labels):

    python tools/gen_io_golden.py        ->  tests/golden/io_formats.npz   (inputs + the bytes the reference wrote)
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(reference="/root/reference"):
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
    out = os.path.join(ROOT, "tests", "golden", "io_formats.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB;", len(files), "files from write_instances")


if __name__ == "__main__":
    main()
