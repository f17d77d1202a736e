set -x
mkdir -p gpurun_out
timeout 300 scripts/exp_cv2.bin > gpurun_out/r2e_exp_cv2.txt 2>&1
timeout 100 scripts/exp_valu.bin > gpurun_out/r2e_exp_valu.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2e_bench_default.json 2> gpurun_out/r2e_bench_default.err
timeout 600 python bench.py --config configs3 --steps 5 --warmup 2 --cpu-seconds 1 > gpurun_out/r2e_bench_configs3.json 2>/dev/null
timeout 600 python bench.py --config configs4 --steps 8 --warmup 2 --cpu-seconds 1 > gpurun_out/r2e_bench_configs4.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2e_bench_spawn1.json 2>/dev/null
timeout 600 python bench.py --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --no-op-leg > gpurun_out/r2e_bench_b32.json 2>/dev/null
timeout 600 python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-op-leg > gpurun_out/r2e_bench_b1.json 2>/dev/null
for f in gpurun_out/r2e_bench_*.json; do python -c "
import json,sys
d=json.load(open('$f'))
print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['config']['workload'][:70], 'frac', round(d.get('roofline',{}).get('frac',0),3), 'hbm', round(d.get('roofline_hbm',{}).get('frac',0),3))
"; done
