set -x
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r2d_all.log
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r2d_train_timing.txt
import sys, time, torch
sys.path.insert(0, '.')
from pwcnet_amd.train import Trainer
for (N, H, W) in [(8, 448, 1024), (4, 384, 512)]:
    tn = Trainer()
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    im0 = torch.rand((N, H, W, 3), generator=g, device='cuda'); im1 = torch.rand((N, H, W, 3), generator=g, device='cuda')
    gt = torch.randn((N, H, W, 2), generator=g, device='cuda') * 3
    for _ in range(2): tn.step(im0, im1, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K): tn.step(im0, im1, gt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    tf = 3 * 79.31e9 * N * (H * W) / (448 * 1024) / dt / 1e12
    print(f"train step batch {N} {H}x{W}: {dt*1e3:.1f} ms/step = {N/dt:.1f} pairs/s, ~{tf:.1f} TFLOP/s (3x forward conv flops), peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del tn
PY
