mkdir -p gpurun_out/r4h
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r4h/tests_model.txt 2>&1
tail -2 gpurun_out/r4h/tests_model.txt
timeout 300 python scripts/exp_timeline.py 8 2>/dev/null | head -3
for i in 1 2 3; do python bench.py --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
