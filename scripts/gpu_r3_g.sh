mkdir -p gpurun_out/r3g
for v in _nt0 _nt16 _0nt _nt; do
  for sh in 0 1; do timeout 100 ./scripts/exp_cv3$v.bin $sh > gpurun_out/r3g/cv3${v}_s$sh.txt 2>&1; done
done
echo done
