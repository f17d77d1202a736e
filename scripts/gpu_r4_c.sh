mkdir -p gpurun_out/r4c
timeout 120 ./scripts/exp_pos.bin > gpurun_out/r4c/pos.txt 2>&1
echo done
