cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py -x -q -k "attention or hiera_vs" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
(python tools/enc_table.py sam 12 | grep -i "attn\|total") > gpurun_out/enc_attn.txt 2>&1; cat gpurun_out/enc_attn.txt
timeout 600 python bench.py --no-cpu-baseline --projection-world 0 --no-online > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; python -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['value'], d['ms_per_step'], d['sustained']['frames_per_s'], d['roofline']['isolated']['attention_ms_per_frame'], d['roofline']['attention_ms_per_frame'])"
