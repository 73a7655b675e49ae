#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blas -- python $R/tools/blas_ref.py > $R/gpurun_out/prof_blas.log 2>&1
f=$(find $R/gpurun_out/prof_blas -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "Cijk" in r["Name"] or "gemm" in r["Name"].lower():
        print(r["Calls"], "%.1f" % (float(r["AverageNs"])/1e3), r["Name"][:400])
PY
find $R/gpurun_out/prof_blas -name "*kernel_trace.csv" -delete
