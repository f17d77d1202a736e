mkdir -p gpurun_out/r4h
for i in 0 2 3 4 12; do timeout 300 ./scripts/exp_h2.bin $i > gpurun_out/r4h/h2_$i.txt 2>&1; grep -E "^==|entries off|medians|s_memtime|tap timeline|workgroup 0 wave [04]|variant" gpurun_out/r4h/h2_$i.txt | cut -c1-420; done
