mkdir -p gpurun_out/r4h
timeout 2400 python -m pytest tests/test_gpu_grad.py -x -q > gpurun_out/r4h/tests_grad.txt 2>&1
tail -3 gpurun_out/r4h/tests_grad.txt
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-220
