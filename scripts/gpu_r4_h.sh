mkdir -p gpurun_out/r4h
for rep in 1 2 3; do
for args in "--streams 2" "--streams 1"; do
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-op-leg --no-op-timing $args 2>/dev/null | tail -1 > gpurun_out/r4h/b.json
python - "$args" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r4h/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3))
PY
done; done
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
