"""Winograd launches that do not fill the GPU (batch 8, 448x1024: pyramid levels 28x64 / 14x32, estimator levels 1-3):
16 vs 32 output channels per workgroup x channel split 1..4, against the library's own plan.
    python scripts/tune_wino_split.py"""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the PWC_WINO_* environment knobs exist only there
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pwcnet_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
SHAPES = [  # tag, N, H, W, Cin_phys, Cout
    ("fp3 96->96", 16, 28, 64, 96, 96), ("fp4 128->128", 16, 14, 32, 128, 128),
    ("of1 c0", 8, 14, 32, 256, 128), ("of1 c1", 8, 14, 32, 128, 128), ("of1 c2", 8, 14, 32, 128, 96),
    ("of1 c3", 8, 14, 32, 96, 64), ("of1 c4", 8, 14, 32, 64, 32),
    ("of2 c0", 8, 28, 64, 224, 128), ("of2 c1", 8, 28, 64, 128, 128), ("of2 c2", 8, 28, 64, 128, 96),
    ("of2 c3", 8, 28, 64, 96, 64), ("of2 c4", 8, 28, 64, 64, 32),
    ("of3 c3", 8, 56, 128, 96, 64), ("of3 c4", 8, 56, 128, 64, 32),
]
ws = torch.empty((64 << 20,), device="cuda")
for tag, N, H, W, ci, co in SHAPES:
    xs = [torch.randn((N, H, W, ci), device="cuda") for _ in range(4)]
    packed = torch.randn((L.pwc_conv3x3_wino_packed_floats(ci, co),), device="cuda") * 0.01
    bias = torch.zeros((co,), device="cuda")
    y = torch.empty((N, H, W, co), device="cuda")
    def run(split, x):
        if split > 1:
            return L.pwc_conv3x3_wino_split_f32(p(x), ci, p(packed), p(bias), p(y), co, N, H, W, ci, co, 1, 1, 0.1, split,
                                                p(ws), ws.numel(), None)
        return L.pwc_conv3x3_wino_f32(p(x), ci, p(packed), p(bias), p(y), co, N, H, W, ci, co, 1, 1, 0.1, None)
    def t(split):
        for x in xs: run(split, x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for x in xs: assert run(split, x) == 0
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / 20
    os.environ.pop("PWC_WINO_FORCE_BN", None)
    plan = L.pwc_conv3x3_wino_split_plan(N, H, W, ci, co, 1)
    auto = t(plan)
    res = []
    for bn in (16, 32):
        if co % bn: continue
        os.environ["PWC_WINO_FORCE_BN"] = str(bn)
        for split in (1, 2, 3, 4):
            if split > ci // 16: continue
            res.append((t(split), bn, split))
    os.environ.pop("PWC_WINO_FORCE_BN", None)
    best = min(res)
    print(f"{tag:14s} N{N:2d} {H:3d}x{W:<3d} {ci:3d}->{co:<3d} auto(split {plan}) {auto:6.1f} us | best {best[0]:6.1f} us bn{best[1]} split{best[2]} | "
          + " ".join(f"bn{b}/s{s}:{v:5.1f}" for v, b, s in res), flush=True)
