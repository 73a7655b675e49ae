"""Measure every candidate tile on every product the encoder forwards launch and (re)write ovo_amd/csrc/gemm_tuned.h.

    python tools/gemm_tune.py [--write] [--batches 1,4,5,6,10,12,14] [--margin 0.02] [--report gpurun_out/gemm_tune.txt]

1. the ViT-L/14-336 and hiera_b+ forwards run once per batch size under the library's profiler (OVO_PROF_DUMP lines carry M, N, K and the launch's
   variant flags, gemm_common.h: gemm_flags) -> the set of products;
2. each product with a plain / rotary / activation / residual epilogue is timed on every tile family (OVO_GEMM_TILE) with cold weights (four
   weight buffers in rotation), three interleaved rounds, median -- and on the cost model's own choice (OVO_GEMM_NO_TUNED=1);
3. where the best tile beats the cost model by more than --margin an entry goes into the table.  Products with a window remap, a LayerNorm
   in the operand load or a fused argmax keep the cost model (their launches cannot be reproduced from M, N, K and the flags alone).
Run on the GPU box; --write replaces the header in the source tree (rebuild afterwards)."""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")
import argparse, ctypes as C, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from ovo_amd import _lib as L

TILES = ["256x256", "256x128", "128x128", "128x64", "64x128", "64x64", "stream"]
F_F32, F_ADD, F_ACT, F_ROPE, F_OTHER = 1, 2, 12, 16, ~(1 | 2 | 12 | 16)
dev = torch.device("cuda", 0)


def collect(batches):
    from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
    from ovo_amd.encoders.vit import SPECS as VS, HipViT
    lib = L.load()
    vit, sam = HipViT(VS["PE-Core-L14-336"], None, dev, 0), HipHiera(HS["hiera_b+"], None, dev, 0)
    found = {}
    os.environ["OVO_GEMM_NO_TUNED"] = "1"
    for b in batches:
        xs = torch.randn(2 * b, 3, 336, 336, device=dev), torch.randn(b, 3, 1024, 1024, device=dev)
        for fn in (lambda: vit.forward(xs[0], tokens=True), lambda: sam.forward(xs[1])):
            fn(); torch.cuda.synchronize()
            dump = tempfile.mktemp(); os.environ["OVO_PROF_DUMP"] = dump
            L.check(lib.ovo_profile_start()); fn()
            ms, work, n = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_int64 * 9)()
            L.check(lib.ovo_profile_stop(ms, work, n, 9))
            for line in open(dump):
                k, m, nn, kk, w, t, fl = line.split()
                if int(k) in (0, 3, 4, 5, 6, 7, 8):
                    e = found.setdefault((int(m), int(nn), int(kk), int(fl)), [0, 0.0])
                    e[0] += 1; e[1] += float(t)
            os.remove(dump)
    os.environ.pop("OVO_PROF_DUMP", None); os.environ.pop("OVO_GEMM_NO_TUNED", None)
    del vit, sam
    torch.cuda.empty_cache()
    return found


def make_call(m, n, k, flags):
    lib = L.load()
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16) for _ in range(4)]
    f32 = bool(flags & F_F32)
    out = torch.zeros(m, n, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    bias = torch.randn(n, device=dev)
    g = L.Gemm(); g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, g.add, g.ld_add = a.data_ptr(), k, ws[0].data_ptr(), k, bias.data_ptr(), out.data_ptr(), n, None, 0
    g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, n, k, 2, 0 if f32 else 2, (flags & F_ACT) >> 2, 1.0
    keep = [a, ws, out, bias]
    if flags & F_ADD:
        add = out if f32 else torch.zeros(m, n, device=dev)        # the encoders' residual products update the f32 stream in place
        g.add, g.ld_add = add.data_ptr(), n; keep.append(add)
    if flags & F_ROPE:
        T, hd = 577, 64
        cs, sn = torch.rand(T, hd, device=dev), torch.rand(T, hd, device=dev)
        rope = L.Rope(); rope.cos, rope.sin, rope.T, rope.hd, rope.cols, rope.t0 = cs.data_ptr(), sn.data_ptr(), T, hd, 2 * n // 3, 1
        keep += [cs, sn, rope]
        fn = lambda i: lib.ovo_gemm_rope(C.byref(g), C.byref(rope), L.stream())
    else:
        fn = lambda i: lib.ovo_gemm(C.byref(g), L.stream())
    def call(i):
        g.W = ws[i & 3].data_ptr()
        return fn(i)
    return call, keep


def time_tile(call, tile, iters):
    for k in ("OVO_GEMM_TILE", "OVO_GEMM_NO_TUNED"): os.environ.pop(k, None)
    if tile == "model": os.environ["OVO_GEMM_NO_TUNED"] = "1"
    else: os.environ["OVO_GEMM_TILE"] = tile
    for i in range(3):
        if call(i) != 0: return None                               # this tile does not take the shape
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): call(i)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,4,5,6,10,12,14")      # the encoder look-ahead groups of bench.py: 14 + 10 (default flags), 5 | 14 + 6 (--steps 20 --warmup 5)
    ap.add_argument("--margin", type=float, default=0.02)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--report", default=os.path.join(R, "gpurun_out", "gemm_tune.txt"))
    ap.add_argument("--header", default=os.path.join(R, "gpurun_out", "gemm_tuned.h"), help="where the generated header goes (also into the source tree with --write)")
    a = ap.parse_args()
    found = collect([int(v) for v in a.batches.split(",")])
    lines, entries = [], []
    hdr = "%-24s %5s %4s" % ("M, N, K", "flags", "n") + "".join("%9s" % t for t in ["model"] + TILES) + "   best"
    print(hdr); lines.append(hdr)
    for (m, n, k, fl), (cnt, tot) in sorted(found.items(), key=lambda kv: -kv[1][1]):
        if fl & F_OTHER:
            continue
        call, keep = make_call(m, n, k, fl)
        iters = 20 if 2.0 * m * n * k > 2e10 else 50
        cands = ["model"] + [t for t in TILES if t != "stream" or (m >= 16384 and k <= 256)]
        res = {t: [] for t in cands}
        for _ in range(3):
            for t in cands:
                us = time_tile(call, t, iters)
                if us is not None: res[t].append(us)
        med = {t: sorted(v)[len(v) // 2] for t, v in res.items() if v}
        best = min((t for t in med if t != "model"), key=lambda t: med[t])
        gain = med["model"] / med[best] - 1.0
        row = "%-24s %5d %4d" % (f"{m}, {n}, {k}", fl, cnt) + "".join(("%9.1f" % med[t]) if t in med else "%9s" % "-" for t in ["model"] + TILES)
        row += "   %s%s" % (best, " (+%.1f %%)" % (100 * gain) if gain > a.margin else "")
        print(row); lines.append(row)
        if gain > a.margin:
            entries.append((m, n, k, fl, best, med["model"], med[best]))
        del call, keep
        torch.cuda.empty_cache()
    for k in ("OVO_GEMM_TILE", "OVO_GEMM_NO_TUNED"): os.environ.pop(k, None)
    src = open(os.path.join(R, "ovo_amd", "csrc", "gemm_tuned.h")).read()
    head, tail = src[:src.index("static const TunedTile kTunedTiles[] = {")], "};\n}  // namespace ovo_gemm_detail\n"
    body = "static const TunedTile kTunedTiles[] = {\n"
    for m, n, k, fl, best, t0, t1 in entries:
        body += "    {%d, %d, %d, %d, \"%s\"},      // %.1f -> %.1f us\n" % (m, n, k, fl, best, t0, t1)
    body += "    {0, 0, 0, 0, nullptr},\n"
    os.makedirs(os.path.dirname(a.report), exist_ok=True)
    open(a.report, "w").write("\n".join(lines) + "\n")
    open(a.header, "w").write(head + body + tail)
    if a.write:
        open(os.path.join(R, "ovo_amd", "csrc", "gemm_tuned.h"), "w").write(head + body + tail)
    print(f"{len(entries)} entries -> {a.header}")


if __name__ == "__main__":
    main()
