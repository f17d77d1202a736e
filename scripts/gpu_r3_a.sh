set -x
mkdir -p gpurun_out/r3a
./scripts/exp_cv2.bin 0 quick > gpurun_out/r3a/cv2_quick.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r3a/configs.log
