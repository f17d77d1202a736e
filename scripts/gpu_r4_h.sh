mkdir -p gpurun_out/r4h
for k in 4 8 14; do echo "skew $k"; timeout 300 scripts/exp_c16pair_skew$k.bin 2>&1 | grep "medians" | sed -n '1p;$p'; done
echo "skew 0"; timeout 300 scripts/exp_c16pair.bin 2>&1 | grep "medians" | sed -n '1p;$p'
