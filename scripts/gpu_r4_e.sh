mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "side_stream or first_call or sub_batches" 2>&1 | tail -15 > gpurun_out/r4e/t_streams.log
timeout 200 python scripts/exp_host.py 1 > gpurun_out/r4e/host_b1.txt 2>&1
timeout 200 python scripts/exp_host.py 2 > gpurun_out/r4e/host_b2.txt 2>&1
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err
echo done
