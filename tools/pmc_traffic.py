"""Reduce rocprofv3 --pmc counter CSVs to HBM bytes per launch per kernel (the `traffic` field of bench.py's roofline).

Collected in SEPARATE passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced streaming read, so it is doubled.  Both factors are CHECKED against tools/pmc_calib.py's
known-byte kernels when --calib-* directories are given, and the measured factors are stored next to the numbers.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def read(dirname, counter):
    """-> {kernel_name: [sum_value, dispatches]}"""
    out = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                e = out[row["Kernel_Name"]]
                e[0] += float(row["Counter_Value"])
                e[1] += 1
    return out


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:120]


def calib_factor(dirname, counter, pattern, true_bytes):
    for k, (v, n) in read(dirname, counter).items():
        if re.search(pattern, k) and n > 0:
            return true_bytes / (v / n * 1024.0), k
    return None, None


def csrc_sha16():
    """sha256[:16] over the kernel sources (ovo_amd/csrc, sorted): bench.py prints it beside `traffic` and says whether the sources it runs are the
    ones the counters were taken on (the GPU box has no .git)."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ovo_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(root, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch-dir", required=True)
    ap.add_argument("--write-dir", required=True)
    ap.add_argument("--calib-fetch-dir")
    ap.add_argument("--calib-write-dir")
    ap.add_argument("--census", help="bench.py --census output of the SAME command: algorithmic bytes per launch of exactly the launches counted here")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch_fix, write_fix, notes = 2.0, 1.0, {}
    if a.calib_fetch_dir:
        f1, k1 = calib_factor(a.calib_fetch_dir, "FETCH_SIZE", r"k_similarity_mfma", float(1 << 30))
        f2, k2 = calib_factor(a.calib_fetch_dir, "FETCH_SIZE", r"copy|Copy", float(1 << 30))
        notes["fetch_calibration"] = {"ovo_similarity 1 GiB read": f1, "torch copy 1 GiB read": f2, "applied": fetch_fix}
    if a.calib_write_dir:
        w1, _ = calib_factor(a.calib_write_dir, "WRITE_SIZE", r"fill|Fill", float(1 << 30))
        w2, _ = calib_factor(a.calib_write_dir, "WRITE_SIZE", r"copy|Copy", float(1 << 30))
        notes["write_calibration"] = {"torch fill 1 GiB write": w1, "torch copy 1 GiB write": w2, "applied": write_fix}
    fetch, write = read(a.fetch_dir, "FETCH_SIZE"), read(a.write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fv, fn = fetch.get(k, [0.0, 0])
        wv, wn = write.get(k, [0.0, 0])
        fb = fv / fn * 1024.0 * fetch_fix if fn else None
        wb = wv / wn * 1024.0 * write_fix if wn else None
        kernels[short(k)] = {"launches": max(fn, wn), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                             "hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0)}
    with open(a.out, "w") as fh:
        census = None
        if a.census and os.path.exists(a.census):
            with open(a.census) as cf:
                census = json.load(cf)
        json.dump({"unit": "bytes per launch (KiB counters x 1024; FETCH_SIZE x 2 on gfx950)", "notes": notes, "csrc_sha16": csrc_sha16(), "census": census,
                   "kernels": kernels}, fh, indent=1)
    top = sorted(kernels.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))[:12]
    for k, v in top:
        print(f"{v['launches']:6d} x {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB  {k}")
    print(json.dumps(notes))
    if a.census and os.path.exists(a.census):
        with open(a.census) as cf:
            for tile, c in json.load(cf)["tiles"].items():
                hit = [v for k, v in kernels.items() if f"k_gemm8p<{tile.replace(',', ', ')}" in k]
                if hit and c["launches"]:
                    print(f"tile {tile}: PMC {hit[0]['hbm_bytes_per_launch'] / 1e6:.1f} MB over {hit[0]['launches']} launches, algorithmic {c['algorithmic_bytes_per_launch'] / 1e6:.1f} MB over "
                          f"{c['launches']} launches: ratio {hit[0]['hbm_bytes_per_launch'] / c['algorithmic_bytes_per_launch']:.3f}")


if __name__ == "__main__":
    main()
