"""Build libovo_hip.so (gfx950) in-tree with hipcc.  `python -m ovo_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
SO = os.path.join(OUT_DIR, "libovo_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-I" + os.path.join(os.path.dirname(HERE), "include")]
# per-file extra flags: geometry must not contract a*b+c into fma on its own (bit-exact parity); the three translation units with a LayerNorm in
# their operand load are built WITHOUT SLP vectorisation -- its packed-f32 (v_pk_*_f32) statistics chain gave wrong variances one time in ~10^5
# whenever a wave of another workgroup shared the SIMD (round 6, csrc/mlp_stream.hip: mlp_stream_launch)
# gemm8p.hip: no SLP vectorisation either, for speed -- its epilogues hold a value's table entry / bias / accumulator in registers that do not pair up, and the
# packed form pays two v_mov per v_pk_fma to line them up: FC1 + GELU (57344, 1792, 448) 135.1 -> 131.1 us, (16156, 4096, 1024) 133.6 -> 131.0, the other
# epilogues unchanged (two alternating rounds of both builds in one session, tools/gemm_bench.py); results are bit-identical (the same IEEE operations)
EXTRA = {"geometry.hip": ["-ffp-contract=off"], "mlp_stream.hip": ["-fno-slp-vectorize"], "gemm_stream.hip": ["-fno-slp-vectorize"],
         "winattn.hip": ["-fno-slp-vectorize"], "gemm8p.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libovo_hip.so cannot be built")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "ovo_hip.h"))
    deps.append(os.path.abspath(__file__))                       # the flags live here
    if force or _stale(obj, deps):
        cmd = [hipcc(), "-c", os.path.join(CSRC, src), "-o", obj] + COMMON + EXTRA.get(src, [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True, gemm_debug: bool = False, experimental: bool = False) -> str:
    """`gemm_debug` (tools/ only: python -m ovo_amd.build --force --gemm-debug): the GEMM kernels' early exits / time stamps / ablation knobs
    (OVO_8P_DEBUG, OVO_8P_STAMPS, OVO_8P_DELAY, OVO_8Q_DEBUG) are compiled in; a production build has none of them.
    `experimental` (--experimental): also compiles the two forms that lost their measurements -- the persistent 256 x 128 GEMM (gemm8q.hip,
    OVO_GEMM_TILE=256x128p) and the one-launch round chain (k_round_chain, OVO_ROUND_CHAIN=1); their parity tests skip without it."""
    if gemm_debug:
        for f in ("gemm8p.hip", "gemm8q.hip", "mlp_stream.hip"):
            EXTRA[f] = EXTRA.get(f, []) + ["-DOVO_GEMM_DEBUG"]
    if experimental:                                   # kernels that were measured and lost (persistent 256 x 128 GEMM, one-launch round chain)
        for f in ("gemm8q.hip", "geometry.hip"):
            EXTRA[f] = EXTRA.get(f, []) + ["-DOVO_EXPERIMENTAL"]
    # experiments only (tools/): OVO_HIPCC_EXTRA="mlp_stream.hip=-fno-slp-vectorize;gemm8p.hip=-mllvm,-amdgpu-..." adds flags to single translation units
    for item in filter(None, os.environ.get("OVO_HIPCC_EXTRA", "").split(";")):
        f, _, flags = item.partition("=")
        EXTRA[f] = EXTRA.get(f, []) + [x for x in flags.split(",") if x]
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(SO, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[ovo_amd.build] {SO} ({os.path.getsize(SO) / 1024:.0f} KiB, {len(srcs)} translation units)")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, gemm_debug="--gemm-debug" in sys.argv, experimental="--experimental" in sys.argv)
