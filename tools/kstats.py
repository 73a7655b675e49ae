"""Per-forward kernel table from a rocprofv3 --kernel-trace --stats directory:  python tools/kstats.py DIR N_FORWARDS [ROWS]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
per = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time {tot / per / 1e6:.3f} ms per forward, {sum(int(r['Calls']) for r in rows) / per:.0f} launches")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    n = r["Name"]
    n = n[n.find("k_"):] if "k_" in n else n
    calls, avg, t = int(r["Calls"]) / per, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / per / 1e6
    print(f"{n[:64]:66s} {calls:6.1f}/fwd  avg {avg:7.1f} us  {t:6.3f} ms/fwd")
