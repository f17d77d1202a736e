mkdir -p gpurun_out/r3l
timeout 2400 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -s 2>&1 | tail -25 > gpurun_out/r3l/t_grad.log
for k in 4 2 1; do timeout 300 python bench.py --steps 30 --warmup 8 --streams $k --no-cpu-baseline --no-op-leg --no-op-timing 2>/dev/null | tail -1 | cut -c1-200 > gpurun_out/r3l/bench_streams$k.txt; done
echo done
