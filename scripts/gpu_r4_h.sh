mkdir -p gpurun_out/r4h
timeout 300 ./scripts/exp_c16pair.bin > gpurun_out/r4h/c16pair.txt 2>&1
grep -E "^==|fused vs|float64|medians" gpurun_out/r4h/c16pair.txt | cut -c1-200
