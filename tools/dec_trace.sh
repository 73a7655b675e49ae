#!/bin/bash
# kernel sequence of ONE SAM2 decoder forward (256 clicks): rocprofv3 kernel trace of tools/amg_bench.py (DEC_ONLY), the last forward's launches in order
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
DEC_ONLY=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/dec_trace -- python $GRAFT_REPO_ROOT/tools/amg_bench.py 16 > $OUT/dec_trace.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("$OUT/dec_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); per = None
# the last forward: find the last k_up2_masks and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "k_up2_masks" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
with open("$OUT/dec_sequence.txt", "w") as o:
    for r in rows[a:b + 3]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:90]
        o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {name}\n")
PY
find $OUT/dec_trace -name "*.csv" -delete
