mkdir -p gpurun_out/r4h
timeout 1800 python -m pytest tests/test_gpu_configs.py -x -q > gpurun_out/r4h/tests_configs.txt 2>&1
tail -3 gpurun_out/r4h/tests_configs.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
