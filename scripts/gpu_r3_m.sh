mkdir -p gpurun_out/r3m
timeout 200 ./scripts/exp_wino4.bin > gpurun_out/r3m/w4_ldsbar.txt 2>&1
timeout 200 ./scripts/exp_wino4_drain.bin > gpurun_out/r3m/w4_drain.txt 2>&1
timeout 100 python scripts/exp_w16.py > gpurun_out/r3m/w16.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-leg > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-op-leg > gpurun_out/r3m/bench_s1.json 2> gpurun_out/r3m/bench_s1.err
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3m/t_all.log
echo done
