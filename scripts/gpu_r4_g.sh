mkdir -p gpurun_out/r4g
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r4g/t_all.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
echo done
