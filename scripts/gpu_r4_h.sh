mkdir -p gpurun_out/r4h
for sh in 0 3 4 16 20; do for k in 0 1 0 1; do echo "shape $sh split_asm $k: $(timeout 300 scripts/exp_h2_split$k.bin $sh 2>&1 | grep -E "^  F\(2x2\)|entries off" | tail -2 | sed 's/.*F(4x4) f16x2 direct/h2/' | tr '\n' ' ')"; done; done > gpurun_out/r4h/h2_split.txt
cat gpurun_out/r4h/h2_split.txt
