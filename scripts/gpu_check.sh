set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/t3.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tee gpurun_out/bench1.log | tail -3
