cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print(d["value"], d["ms_per_step"], d["online"], d["projection"], d["sustained"]["frames_per_s"], d["parity"], d["cpu_baseline"]["value"])
print({k: d["roofline"][k] for k in ("achieved","frac","frame_frac")}, d["roofline"]["isolated"]["frac"])
PY
tail -3 gpurun_out/bench.err
timeout 600 python tools/vit_bench.py > gpurun_out/vit_bench.txt 2>&1; grep -v amdgpu gpurun_out/vit_bench.txt
timeout 600 python tools/query_bench.py > gpurun_out/query_bench.txt 2>&1; grep -v amdgpu gpurun_out/query_bench.txt
timeout 600 python tools/query_bench.py 1250000 >> gpurun_out/query_bench.txt 2>&1
