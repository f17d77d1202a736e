set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "cost_volume" 2>&1 | tail -8 | tee gpurun_out/r2b_cv.log
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r2b_bench.err | tee gpurun_out/r2b_bench.json | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r2b_all.log
