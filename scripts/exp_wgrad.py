"""Weight-gradient kernel at the training step's main shapes (batch 8, 448x1024): us and TFLOP/s per launch.
    python scripts/exp_wgrad.py"""
import sys, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from pwcnet_amd import grad_ops as G
from pwcnet_amd.modules import View

SHAPES = [  # N, H, W, Cin_phys, Cout, dil
    (8, 112, 256, 128, 128, 1), (8, 112, 256, 128, 96, 1), (8, 112, 256, 96, 64, 1), (8, 112, 256, 64, 32, 1),
    (8, 112, 256, 48, 128, 1), (8, 112, 256, 128, 128, 2), (8, 112, 256, 128, 128, 4), (8, 112, 256, 128, 96, 8),
    (8, 112, 256, 96, 64, 16), (16, 112, 256, 32, 32, 1), (8, 56, 128, 192, 128, 1), (8, 56, 128, 128, 128, 1),
    (16, 56, 128, 64, 64, 1), (8, 28, 64, 224, 128, 1), (16, 28, 64, 96, 96, 1), (16, 224, 512, 16, 16, 1),
]
import os
if os.environ.get('WG_FIRST'): SHAPES = SHAPES[:int(os.environ['WG_FIRST'])]
for (N, H, W, ci, co, d) in SHAPES:
    x = torch.randn((N, H, W, ci), device='cuda'); dy = torch.randn((N, H, W, co), device='cuda')
    dw = torch.zeros((3, 3, ci, co), device='cuda')
    vx, vd = View(x.data_ptr(), ci, N, H, W, ci), View(dy.data_ptr(), co, N, H, W, co)
    for _ in range(3): G.conv3x3_wgrad(vx, vd, dw, ci, 1, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): G.conv3x3_wgrad(vx, vd, dw, ci, 1, d)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * 9 * ci * co * N * H * W
    print(f"N{N:3d} {H:4d}x{W:<4d} {ci:4d}->{co:<4d} d{d:<3d} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s (incl. reduce)", flush=True)
