mkdir -p gpurun_out/r4b
for i in 0 3; do timeout 200 ./scripts/exp_wino4b.bin $i; done > gpurun_out/r4b/wino4b.txt 2>&1
echo done
