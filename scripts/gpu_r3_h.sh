mkdir -p gpurun_out/r3h
for v in "" _st30 _st60; do
  for sh in 0 1 2; do timeout 100 ./scripts/exp_cv3$v.bin $sh > gpurun_out/r3h/cv3${v}_s$sh.txt 2>&1; done
done
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "concat or fused or cost_volume" 2>&1 | tail -8 > gpurun_out/r3h/t_ops.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r3h/t_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err
echo done
