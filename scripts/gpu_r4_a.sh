mkdir -p gpurun_out/r4a
timeout 120 ./scripts/exp_cross_simd.bin > gpurun_out/r4a/cross_simd.txt 2>&1
timeout 600 ./scripts/exp_wino4b.bin > gpurun_out/r4a/wino4b.txt 2>&1
echo done
