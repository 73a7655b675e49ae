"""List the launches of kernels matching a substring inside the steady-state window of a rocprofv3 --kernel-trace CSV directory:
    python tools/find_kernel.py DIR SUBSTRING     (frame boundaries = k_track_project launches).  Diagnosis tool."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
marks = [r[0] for r in rows if "k_track_project" in r[2]]
lo, hi = marks[len(marks) // 2], marks[-1]
frame = 0
for s, e, n, g, w in rows:
    if s < lo or s >= hi: continue
    if "k_track_project" in n: frame += 1
    if sys.argv[2] in n: print(f"frame {frame:3d}  +{(s - lo) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  grid {g} wg {w}  {n[:90]}")
