mkdir -p gpurun_out/r3f
timeout 100 ./scripts/exp_cv3.bin 0 0 0 stamps > gpurun_out/r3f/cv3_s0.txt 2>&1
for sh in 1 2 3 4 5 8; do timeout 100 ./scripts/exp_cv3.bin $sh > gpurun_out/r3f/cv3_s$sh.txt 2>&1; done
echo done
