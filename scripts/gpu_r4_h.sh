mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2" > gpurun_out/r4h/tests_h2.txt 2>&1
tail -2 gpurun_out/r4h/tests_h2.txt
timeout 900 ./scripts/exp_h2.bin > gpurun_out/r4h/h2.txt 2>&1
grep -E "medians|entries off" gpurun_out/r4h/h2.txt | grep -v "F(2x2) 0, " | cut -c1-200
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q > gpurun_out/r4h/tests_model.txt 2>&1
tail -2 gpurun_out/r4h/tests_model.txt
timeout 600 python scripts/exp_ab_model.py f16x2 8 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-leg 2>/dev/null | tail -1 | cut -c1-200
