mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "f16x2 or f4x4" -s 2>&1 | grep -E "passed|failed|f16x2 layers|F\(4x4\) layers|Error|assert" | head
