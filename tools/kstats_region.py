"""Kernel statistics of bench.py's TIMED REGION only, from a rocprofv3 --kernel-trace directory of that command: bench.py launches an empty
`k_marker` kernel right before and right after the timed steps; dispatches that start between the first two markers are aggregated into a
CSV with the columns of rocprofv3's own *_kernel_stats.csv.  A third and fourth marker bracket the `isolated` pass of the roofline leg (the same
kernels with the pipeline's streams folded into one): with a second output name that region is written too -- its average for the dominant kernel
is what `roofline.isolated.avg_launch_us` must agree with (no concurrency for the tracer to stretch).
usage: python tools/kstats_region.py TRACE_DIR OUT.csv [OUT_ISOLATED.csv]"""
import csv, glob, math, sys
from collections import defaultdict
src = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(src)))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
start_key = next(k for k in rows[0] if k.lower().startswith("start"))
end_key = next(k for k in rows[0] if k.lower().startswith("end"))
marks = sorted(int(r[start_key]) for r in rows if "k_marker" in r[name_key])
if len(marks) < 2:
    sys.exit("no two k_marker dispatches in the trace")
def region(lo, hi, out, label):
    agg = defaultdict(list)
    for r in rows:
        s = int(r[start_key])
        if lo < s < hi and "k_marker" not in r[name_key]:
            agg[r[name_key]].append(int(r[end_key]) - s)
    total = sum(sum(v) for v in agg.values())
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            n, t = len(d), sum(d)
            mean = t / n
            sd = math.sqrt(sum((x - mean) ** 2 for x in d) / n)
            w.writerow([name, n, t, round(mean, 3), round(100.0 * t / total, 4), min(d), max(d), round(sd, 3)])
    print(f"{label}: {(hi - lo) / 1e6:.3f} ms between the markers, {sum(len(v) for v in agg.values())} dispatches, {total / 1e6:.3f} ms of kernel time -> {out}")


region(marks[0], marks[1], sys.argv[2], "timed region")
if len(sys.argv) > 3:
    if len(marks) < 4:
        sys.exit("no third / fourth k_marker (the isolated pass) in the trace")
    region(marks[2], marks[3], sys.argv[3], "isolated pass")
