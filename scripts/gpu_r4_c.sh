mkdir -p gpurun_out/r4c
timeout 60 ./scripts/exp_m16.bin > gpurun_out/r4c/m16.txt 2>&1
timeout 120 ./scripts/exp_pos.bin > gpurun_out/r4c/pos.txt 2>&1
echo done
