mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2" > gpurun_out/r4h/tests_h2.txt 2>&1
tail -3 gpurun_out/r4h/tests_h2.txt
timeout 600 python scripts/exp_timeline.py 8 > gpurun_out/r4h/timeline8.txt 2>&1
grep -E "fp_extractor" gpurun_out/r4h/timeline8.txt | head -20; tail -1 gpurun_out/r4h/timeline8.txt
