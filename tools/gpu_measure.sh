#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace stats and the PMC traffic passes (summaries go to profiles/ by hand).  The PMC passes run the bench's own
# encoder groups (14 + 14 + 10 frames: the roofline pass profiles 14 + 10), so that `roofline.traffic` and `algorithmic_bytes_per_launch` describe the same launches.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
OVO_VIT_LNFOLD=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_flags_nofold.json 2> $OUT/bench_driver_flags_nofold.err
timeout 600 python bench.py --sam hiera_l --no-cpu-baseline --sustain-seconds 0 --projection-world 0 --no-online > $OUT/bench_hiera_l.json 2> $OUT/bench_hiera_l.err
timeout 600 python bench.py --sam-full --no-cpu-baseline --sustain-seconds 0 --projection-world 0 --no-online > $OUT/bench_sam_full.json 2> $OUT/bench_sam_full.err
(timeout 300 python tools/query_bench.py; timeout 300 python tools/query_bench.py 1250000) > $OUT/query_bench.log 2>&1
timeout 300 python tools/amg_bench.py 16 > $OUT/amg_bench.log 2>&1
timeout 300 python tools/geom_bench.py > $OUT/geom_bench.txt 2>&1
TILES="auto,ring,256x256,256x128" SHAPES="13848,3072,1024;13848,1024,1024;13848,4096,1024;13848,1024,4096;49152,1792,448;49152,448,1792;58800,1344,448;9232,3072,1024;9232,4096,1024;4616,3072,1024;4616,4096,1024;4096,4096,4096;8192,8192,8192" timeout 600 python tools/gemm_bench.py > $OUT/gemm_sweep.txt 2>&1
(python tools/enc_table.py vit 12; python tools/enc_table.py sam 12) > $OUT/enc_tables_b12.txt 2>&1
(export TILES="tiled,stream" ROUNDS=3 ITERS=30
 echo "== bf16 out, bias"; BIAS=1 SHAPES="524288,336,128;524288,448,128;524288,672,128;131072,896,256;131072,672,256;131072,1344,256;131072,448,256;524288,32,256;131072,64,256" python tools/gemm_bench.py
 echo "== bf16 out, bias + GELU"; BIAS=1 ACT=1 SHAPES="524288,448,128;131072,896,256" python tools/gemm_bench.py
 echo "== f32 out, bias + in-place residual"; BIAS=1 ADD=1 INPLACE=1 OUT=f32 SHAPES="524288,112,128;131072,224,256;524288,256,128;524288,224,128;131072,256,256;65536,112,192" python tools/gemm_bench.py) > $OUT/gemm_stream.txt 2>&1
timeout 300 python tools/replicated_cost.py 64 8 > $OUT/replicated_cost.txt 2>&1
(for w in 1 2 4 8; do echo "world $w"; timeout 300 python tools/round_emulation.py $w; done) > $OUT/round_emulation.txt 2>&1
timeout 600 python tools/vit_bench.py > $OUT/vit_bench.txt 2>&1
timeout 300 python tools/attn_bench.py > $OUT/attn_bench.txt 2>&1
(timeout 300 python tools/fold_bench.py; timeout 300 python tools/fold_bench.py 11540) > $OUT/fold_bench.txt 2>&1
timeout 300 python tools/scatter_bench.py > $OUT/scatter_bench.txt 2>&1; timeout 300 python tools/scatter_bench.py 5000000 768 60000 >> $OUT/scatter_bench.txt 2>&1
RBS=0,1,2,3 timeout 300 python tools/mlp_bench.py > $OUT/mlp_bench.txt 2>&1
timeout 900 python tools/mlp_stress.py > $OUT/mlp_stress.txt 2>&1
(cd /tmp && hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_race $R/tools/lds_dma_race.hip && timeout 300 /tmp/lds_dma_race 4096 20) > $OUT/lds_dma_race.txt 2>&1
timeout 300 python tools/winattn_bench.py 12 > $OUT/winattn_bench.txt 2>&1
(timeout 300 python tools/patch_embed_bench.py 12; timeout 300 python tools/patch_embed_bench.py 1) > $OUT/patch_embed_bench.txt 2>&1
timeout 300 python tools/round_host.py 8 > $OUT/round_host.txt 2>&1
timeout 300 python tools/resize_batch_bench.py > $OUT/resize_batch_bench_now.txt 2>&1
# BASELINE configs[3]'s process layout on ONE GPU over gloo (8 ranks, 5 M-point map): executes rings, staging and shards at world 8; not a scaling number
(export OVO_FORCE_DEVICE=0 OVO_DIST_BACKEND=gloo; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 8 --warmup 2 --map-points 5000000 --no-cpu-baseline --no-roofline --dense-merge none 2>&1 | tail -1) > $OUT/bench_world8_one_gpu_5m.json
# the same layout with north_star's literal collective (`dense_merge reduce`: every rank a WHOLE accumulator, one bucketed sum-reduce) at a size eight whole accumulators fit one GPU
(export OVO_FORCE_DEVICE=0 OVO_DIST_BACKEND=gloo; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 8 --steps 8 --warmup 2 --map-points 400000 --no-cpu-baseline --no-roofline --dense-merge reduce 2>&1 | tail -1) > $OUT/bench_world8_one_gpu_dense_merge.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --no-cpu-baseline --steps 24 --sustain-seconds 0 --projection-world 0 --no-online > $OUT/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --no-cpu-baseline --census $OUT/pmc_census.json --no-shared-crops --steps 24 --warmup 14 --sustain-seconds 0 --projection-world 0 --no-online > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-shared-crops --steps 24 --warmup 14 --sustain-seconds 0 --projection-world 0 --no-online > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python $R/tools/pmc_calib.py > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python $R/tools/pmc_calib.py > $OUT/calib_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/wa_fetch -- python $R/tools/winattn_bench.py 12 > $OUT/wa_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wa_write -- python $R/tools/winattn_bench.py 12 > $OUT/wa_write.log 2>&1
DEC_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/amg_prof -- python $R/tools/amg_bench.py 16 > $OUT/amg_prof.log 2>&1
ITERS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/geom_stats -- python $R/tools/geom_bench.py 10000000 > $OUT/geom_stats.log 2>&1
ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/geom_fetch -- python $R/tools/geom_bench.py 10000000 > $OUT/geom_fetch.log 2>&1
ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/geom_write -- python $R/tools/geom_bench.py 10000000 > $OUT/geom_write.log 2>&1
ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/geom_sq -- python $R/tools/geom_bench.py 10000000 > $OUT/geom_sq.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch-dir $OUT/pmc_fetch --write-dir $OUT/pmc_write --calib-fetch-dir $OUT/calib_fetch --calib-write-dir $OUT/calib_write --census $OUT/pmc_census.json --out $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
python tools/pmc_traffic.py --fetch-dir $OUT/geom_fetch --write-dir $OUT/geom_write --out $OUT/geom_10m_pmc_traffic.json > $OUT/geom_pmc.log 2>&1
python tools/pmc_traffic.py --fetch-dir $OUT/wa_fetch --write-dir $OUT/wa_write --out $OUT/winattn_pmc_traffic.json > $OUT/wa_pmc.log 2>&1
python tools/sq_counters.py $OUT/geom_sq > $OUT/geom_10m_sq_counters.txt 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv
python tools/kstats_region.py $OUT/prof $OUT/bench_n1_timed_region_kernel_stats.csv $OUT/bench_n1_isolated_pass_kernel_stats.csv > $OUT/kstats_region.log 2>&1
cp $(find $OUT/geom_stats -name "*kernel_stats.csv" | head -1) $OUT/geom_10m_kernel_stats.csv
cp $(find $OUT/amg_prof -name "*kernel_stats.csv" | head -1) $OUT/sam_decoder_kernel_stats.csv
# keep the merge-back small: drop the per-dispatch traces, keep stats + reduced PMC
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/bench.json | cut -c1-1500; cut -c1-400 $OUT/bench_hiera_l.json; cut -c1-400 $OUT/bench_sam_full.json; tail -16 $OUT/pmc_traffic.log; cat $OUT/geom_bench.txt
