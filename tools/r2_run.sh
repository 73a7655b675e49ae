#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_sam_decoder.py -x -q 2>&1 | tail -15
timeout 300 python tools/amg_bench.py 16 2>&1 | grep -v amdgpu
cd /tmp; export TMPDIR=/tmp
rm -rf $O/amg_prof
DEC_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/amg_prof -- python $GRAFT_REPO_ROOT/tools/amg_bench.py 16 > $O/amg_prof.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/amg_prof -name "*kernel_stats.csv" | sort | tail -1) $O/amg_kernel_stats.csv
find $O/amg_prof -name "*kernel_trace.csv" -delete
head -16 $O/amg_kernel_stats.csv | cut -c1-150
