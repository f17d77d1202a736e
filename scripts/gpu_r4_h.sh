mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2" > gpurun_out/r4h/tests_h2.txt 2>&1
tail -3 gpurun_out/r4h/tests_h2.txt
timeout 600 python scripts/exp_timeline.py 8 > gpurun_out/r4h/timeline8.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q > gpurun_out/r4h/tests_model.txt 2>&1
tail -3 gpurun_out/r4h/tests_model.txt
echo done
