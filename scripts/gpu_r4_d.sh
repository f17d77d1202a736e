mkdir -p gpurun_out/r4d
for i in 0 3 9 10; do timeout 200 ./scripts/exp_wino4r.bin $i; done > gpurun_out/r4d/wino4r.txt 2>&1
echo done
