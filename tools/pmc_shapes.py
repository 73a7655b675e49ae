"""HBM traffic per launch, shape by shape, against the algorithmic bytes of the same launch (the bench's PMC pass mixes 4- and
12-frame groups, so its per-kernel average has no single algorithmic denominator).

For every product / attention shape of the encoder groups (ViT: the bench's 14 frames = 16 156 rows; Hiera: 12 frames) this runs the one-shape tools under rocprofv3 twice -- `--pmc FETCH_SIZE`, then
`--pmc WRITE_SIZE`, each with `--kernel-trace` only (MI355X_MICROARCH.md: separate passes; KiB units; FETCH_SIZE x 2 on gfx950) -- and prints
    shape | kernel | algorithmic MB | fetched MB | written MB | (fetched + written) / algorithmic
`ROTATE=4` cycles the weights through four buffers so that a repeated launch does not find them in the last-level cache.
usage (GPU box, from /tmp):  python $R/tools/pmc_shapes.py [--out gpurun_out/pmc_shapes.json]"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (label, kind, dims, env)
CASES = [
    ("ViT QKV + rotary", "gemm", (16156, 3072, 1024), {"BIAS": "1", "ROPE": "1"}),
    ("ViT out proj (+= f32 stream)", "gemm", (16156, 1024, 1024), {"BIAS": "1", "ADD": "1", "INPLACE": "1", "OUT": "f32"}),
    ("ViT FC1 + GELU", "gemm", (16156, 4096, 1024), {"BIAS": "1", "ACT": "1"}),
    ("ViT FC2 (+= f32 stream)", "gemm", (16156, 1024, 4096), {"BIAS": "1", "ADD": "1", "INPLACE": "1", "OUT": "f32"}),
    ("Hiera s3 QKV", "gemm", (58800, 1344, 448), {"BIAS": "1"}),
    ("Hiera s3 proj (+= f32)", "gemm", (58800, 448, 448), {"BIAS": "1", "ADD": "1", "INPLACE": "1", "OUT": "f32"}),
    ("Hiera s3 FC1 + GELU", "gemm", (49152, 1792, 448), {"BIAS": "1", "ACT": "1"}),
    ("Hiera s3 FC2 (+= f32)", "gemm", (49152, 448, 1792), {"BIAS": "1", "ADD": "1", "INPLACE": "1", "OUT": "f32"}),
    ("Hiera s1 FC1 + GELU (bf16 A)", "gemm", (786432, 448, 128), {"BIAS": "1", "ACT": "1"}),
    ("Hiera s2 FC1 + GELU (bf16 A)", "gemm", (196608, 896, 256), {"BIAS": "1", "ACT": "1"}),
    ("Hiera s1 FC2 (+= f32)", "gemm", (786432, 112, 448), {"BIAS": "1", "ADD": "1", "INPLACE": "1", "OUT": "f32"}),
    ("ViT attention", "attn", (28, 16, 577, 577, 64), {}),
    ("Hiera global attention", "attn", (12, 8, 4096, 4096, 56), {}),
    ("Hiera 14x14 windows", "attn", (300, 8, 196, 196, 56), {}),
    ("Hiera 8x8 windows", "attn", (12288, 2, 64, 64, 56), {}),
]


def algorithmic(kind, dims, env):
    if kind == "attn":
        B, H, Tq, Tk, hd = dims
        return 2.0 * B * H * hd * (2 * Tq + 2 * Tk)
    m, n, k = dims
    out_b = 4 if env.get("OUT") == "f32" else 2
    return m * k * 2.0 + n * k * 2.0 + m * n * out_b + (m * n * 4.0 if env.get("ADD") else 0.0) + (n * 4.0 if env.get("BIAS") else 0.0)


def counters(dirname, counter):
    out = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    e = out[row["Kernel_Name"]]
                    e[0] += float(row["Counter_Value"]); e[1] += 1
    return out


def one_pass(counter, kind, dims, env, scratch):
    shutil.rmtree(scratch, ignore_errors=True)
    e = dict(os.environ, TMPDIR="/tmp", **env)
    if kind == "gemm":
        e.update(SHAPES=",".join(str(v) for v in dims), TILES="auto", ITERS="8", ROUNDS="1", ROTATE="4")
        cmd = [sys.executable, os.path.join(R, "tools", "gemm_bench.py")]
    else:
        cmd = [sys.executable, os.path.join(R, "tools", "attn_one.py")] + [str(v) for v in dims] + ["8"]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", scratch, "--"] + cmd, env=e, cwd="/tmp",
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:])
    pat = "k_gemm" if kind == "gemm" else "k_attention"
    best = None
    for name, (v, n) in counters(scratch, counter).items():
        if pat in name and (best is None or n > best[2]):
            best = (name, v / n * 1024.0, n)
    return best


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name)).split("(")[0]
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(k_gemm\w*?)ILi(\d+)ELi(\d+)", name)
    return f"{m.group(1)}<{m.group(2)},{m.group(3)},..>" if m else name[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(R, "gpurun_out", "pmc_shapes.json"))
    a = ap.parse_args()
    rows = []
    print("%-32s %-30s %-34s %10s %10s %10s %7s" % ("product", "shape", "kernel", "alg MB", "fetch MB", "write MB", "ratio"))
    only = os.environ.get("ONLY")                                  # substring filter on the product label
    for label, kind, dims, env in CASES:
        if only and only not in label:
            continue
        try:
            f = one_pass("FETCH_SIZE", kind, dims, env, "/tmp/pmc_shape_f")
            w = one_pass("WRITE_SIZE", kind, dims, env, "/tmp/pmc_shape_w")
        except Exception as exc:                                   # one shape failing must not lose the others
            print("%-32s FAILED: %s" % (label, str(exc)[-300:]))
            continue
        alg = algorithmic(kind, dims, env)
        fb, wb = f[1] * 2.0, w[1]                                  # gfx950: FETCH_SIZE reports half of a coalesced streaming read
        rows.append({"product": label, "shape": list(dims), "kernel": short(f[0]), "launches_counted": [f[2], w[2]], "algorithmic_bytes": alg,
                     "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "ratio": (fb + wb) / alg})
        print("%-32s %-30s %-34s %10.1f %10.1f %10.1f %7.2f" % (label, str(dims), short(f[0]), alg / 1e6, fb / 1e6, wb / 1e6, (fb + wb) / alg))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump({"unit": "bytes per launch; FETCH_SIZE KiB x 1024 x 2 (gfx950), WRITE_SIZE KiB x 1024; weights rotated through 4 buffers", "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
