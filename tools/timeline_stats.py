"""GPU timeline of a rocprofv3 --kernel-trace CSV: busy fraction (union of kernel intervals), mean concurrency, idle gaps,
per-stream (queue) busy time -- over the steady-state window (after the first --skip fraction of the trace).  Diagnosis tool.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --no-cpu-baseline --no-roofline --steps 30
    python tools/timeline_stats.py gpurun_out/tl
"""
import csv, glob, os, sys, collections
root = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
marks = [r[0] for r in rows if "k_track_project" in r[3]]       # one launch per frame: the steady-state window is cut on it
lo, hi = marks[int(len(marks) * skip)], marks[-1]
frames = len(marks) - 1 - int(len(marks) * skip)
rows = [r for r in rows if lo <= r[0] < hi]
t0, t1 = lo, hi
print(f"{frames} frames, {(hi - lo) / 1e6 / frames:.3f} ms/frame")
wall = t1 - t0
ev = []
for s, e, q, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; conc_area = 0; depth = 0; last = t0; gaps = []
hist = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    elif t > last: gaps.append(t - last)
    conc_area += depth * (t - last); hist[min(depth, 4)] += t - last
    depth += d; last = t
print(f"window {wall / 1e6:.2f} ms, {len(rows)} kernels; busy (>=1 kernel) {100 * busy / wall:.1f}%, mean concurrency {conc_area / wall:.2f}")
print("time share by #kernels in flight:", {k: f"{100 * v / wall:.1f}%" for k, v in sorted(hist.items())})
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms ({100 * sum(gaps) / wall:.1f}%), largest {[round(g / 1e3) for g in gaps[:8]]} us")
perq = collections.defaultdict(int)
for s, e, q, n in rows: perq[q] += e - s
print("kernel time by queue:", {q: f"{v / 1e6:.1f} ms ({100 * v / wall:.0f}% of wall)" for q, v in sorted(perq.items(), key=lambda kv: -kv[1])})
