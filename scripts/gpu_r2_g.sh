set -x
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "head or direct" 2>&1 | tail -3
timeout 300 python scripts/exp_timeline.py 8 2>/dev/null | grep -n "conv2d_5\b\|conv2d_6\|optflow_4/conv2d_5\|context/conv2d_6" | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -3
