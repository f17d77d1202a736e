mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2 or c16" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-160
