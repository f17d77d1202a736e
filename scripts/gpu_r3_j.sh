mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "f4x4" 2>&1 | tail -5 > gpurun_out/r3j/t_f4.log
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3j/t_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > gpurun_out/r3j/bench_s1.json 2> gpurun_out/r3j/bench_s1.err
timeout 300 python scripts/exp_timeline.py 8 > gpurun_out/r3j/timeline.txt 2>&1
echo done
