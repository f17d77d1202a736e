mkdir -p gpurun_out/r4h
for i in 0 4 12; do timeout 300 ./scripts/exp_h2.bin $i > gpurun_out/r4h/h2_$i.txt 2>&1; grep -E "^==|entries off|medians|s_memtime|variant 4 stream|x[0-9.]+ vs fp32" gpurun_out/r4h/h2_$i.txt | cut -c1-300; done
