# round 2, step A: new config tests, bench line, then the whole GPU suite
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r2a_configs.log
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r2a_bench.err | tee gpurun_out/r2a_bench.json | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py 2>&1 | tail -8 | tee gpurun_out/r2a_all.log
