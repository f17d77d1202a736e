mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2" > gpurun_out/r4h/tests_h2.txt 2>&1
tail -2 gpurun_out/r4h/tests_h2.txt
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r4h/tests_model.txt 2>&1
tail -2 gpurun_out/r4h/tests_model.txt
