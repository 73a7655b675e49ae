"""How full the GPU is inside bench.py's timed region, from a rocprofv3 --kernel-trace directory of that command (k_marker dispatches bracket the region, as in
tools/kstats_region.py): wall time, time with at least one kernel running (union of the dispatch intervals), time by number of kernels in flight, the same for the
MFMA-heavy kernels alone (GEMMs, attention, fused MLP / window attention), and the largest idle gaps with the kernels on either side.
usage: python tools/trace_occupancy.py TRACE_DIR"""
import csv, glob, sys
src = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(src)))
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
sk = next(k for k in rows[0] if k.lower().startswith("start"))
ek = next(k for k in rows[0] if k.lower().startswith("end"))
marks = sorted(int(r[sk]) for r in rows if "k_marker" in r[nk])
lo, hi = marks[0], marks[1]
ev = [(int(r[sk]), int(r[ek]), r[nk]) for r in rows if lo < int(r[sk]) < hi and "k_marker" not in r[nk]]
ev.sort()
heavy = lambda n: any(t in n for t in ("k_gemm", "k_attention", "k_mlp_stream", "k_win_attn", "k_similarity_mfma", "k_patch_embed"))


def profile(events, label):
    pts = []
    for s, e, _ in events:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    hist, cur, last = {}, 0, lo
    for t, d in pts:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d; last = t
    hist[cur] = hist.get(cur, 0) + (hi - last)
    wall = hi - lo
    print(f"{label}: wall {wall / 1e6:.3f} ms; kernels in flight -> share of wall: " + ", ".join(f"{k}: {100.0 * v / wall:.1f} %" for k, v in sorted(hist.items()) if v > 0))
    print(f"   sum of kernel durations {sum(e - s for s, e, _ in events) / 1e6:.3f} ms = {sum(e - s for s, e, _ in events) / wall:.2f} x wall")


profile(ev, "all kernels")
profile([x for x in ev if heavy(x[2])], "MFMA-heavy kernels (GEMMs, attention, fused MLP / window attention, similarity, patch embedding)")
gaps, end, prev = [], lo, "(region start)"
for s, e, n in ev:
    if s > end:
        gaps.append((s - end, prev, n))
    if e > end:
        end, prev = e, n
gaps.sort(reverse=True)
print(f"idle gaps (no kernel at all): {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.3f} ms; the largest:")
for g, a, b in gaps[:12]:
    print(f"   {g / 1e3:8.1f} us  after {a[:70]}  before {b[:70]}")
# one step in the middle of the region: every dispatch between two consecutive instance queries (k_similarity_mfma), encoder kernels summarised
sims = [e for s, e, n in ev if "k_similarity_mfma" in n]
if len(sims) > 12:
    a, b = sims[len(sims) // 2], sims[len(sims) // 2 + 1]
    print(f"one step ({(b - a) / 1e3:.1f} us between two instance queries): dispatches of the keyframe chain (encoder kernels: count only)")
    nheavy, theavy, last_end = 0, 0, a
    for s, e, n in ev:
        if s < a or s >= b:
            continue
        if heavy(n) and "k_similarity" not in n:
            nheavy += 1; theavy += e - s
            continue
        print(f"   +{(s - a) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n[:90]}")
    print(f"   (+ {nheavy} encoder dispatches, {theavy / 1e3:.1f} us of kernel time, in this window)")
