mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "c16 or c3_c16" > gpurun_out/r4h/tests_c16.txt 2>&1
tail -3 gpurun_out/r4h/tests_c16.txt
timeout 300 scripts/exp_c16pair.bin > gpurun_out/r4h/exp_c16pair.txt 2>&1
grep "medians\|ablations" gpurun_out/r4h/exp_c16pair.txt
