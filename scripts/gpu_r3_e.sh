mkdir -p gpurun_out/r3e
for v in "" _v0 _nt _sc1 _ntnt; do
  for sh in 0 1 2; do timeout 100 ./scripts/exp_cv3$v.bin $sh > gpurun_out/r3e/cv3${v}_s$sh.txt 2>&1; done
done
echo done
