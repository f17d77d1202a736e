set -x
for p in 0 1; do
  export PWC_WINO_PERSIST=$p
  timeout 300 python scripts/exp_timeline.py 8 2>/dev/null | sed -n 2,8p
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PERSIST=$p', round(d['value'],1), d['ms_per_step'])"
done
unset PWC_WINO_PERSIST
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wino or conv" 2>&1 | tail -3
