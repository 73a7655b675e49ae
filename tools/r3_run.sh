cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(for d in 0 1 2 3; do echo "== OVO_8Q_DEBUG=$d (1: no epilogue math, 2: no stores)"; OVO_8Q_DEBUG=$d TILES="256x128,256x128p" BIAS=1 ACT=1 ROUNDS=2 SHAPES="13848,4096,1024;49152,1792,448;8192,8192,8192" timeout 600 python tools/gemm_bench.py; done) > gpurun_out/gemm8q_dbg.txt 2>&1; grep -v amdgpu gpurun_out/gemm8q_dbg.txt
