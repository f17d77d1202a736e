mkdir -p gpurun_out/r3k
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r3k/t_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err
timeout 300 python scripts/exp_timeline.py 8 > gpurun_out/r3k/timeline.txt 2>&1
echo done
