mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "f16x2" > gpurun_out/r4h/tests_h2.txt 2>&1
tail -3 gpurun_out/r4h/tests_h2.txt
for sh in 0 2 3 4 20 16; do echo "shape $sh: new $(timeout 300 scripts/exp_h2_defer.bin $sh 2>&1 | grep -E "^  F\(2x2\)" | sed 's/.*F(4x4) f16x2 direct/h2/' | cut -c1-40 | tr '\n' ' ') | old $(timeout 300 scripts/exp_h2.bin $sh 2>&1 | grep -E "^  F\(2x2\)" | sed 's/.*F(4x4) f16x2 direct/h2/' | cut -c1-40 | tr '\n' ' ')"; done
