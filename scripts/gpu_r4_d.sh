mkdir -p gpurun_out/r4d
for i in 0 1 2 3 4 5 6 7 9; do timeout 200 ./scripts/exp_wino4r.bin $i; done > gpurun_out/r4d/wino4r.txt 2>&1
echo done
