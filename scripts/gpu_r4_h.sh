mkdir -p gpurun_out/r4h
timeout 600 ./scripts/exp_h2.bin > gpurun_out/r4h/h2.txt 2>&1
echo done
