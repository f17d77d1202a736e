"""16 vs 32 output channels per workgroup, persistent or not, on the short-channel-loop layers (2-4 stages).
Measured: the plan in the library is within 1 us of the best everywhere; a persistent 32-cout form (spills 80-92 B per
lane) gives 65.9 vs 58.9 us on 32->32 at 112x256 and nothing elsewhere.   python scripts/exp_wino_bn.py"""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the PWC_WINO_* environment knobs exist only there
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pwcnet_amd import _lib
L = _lib.lib(); p = lambda t: ctypes.c_void_p(t.data_ptr())
for tag, N, H, W, ci, co in [("fp1 32->32", 16, 112, 256, 32, 32), ("fp2 64->64", 16, 56, 128, 64, 64), ("fp0 16->16", 16, 224, 512, 16, 16), ("of4 c4 64->32", 8, 112, 256, 64, 32)]:
    xs = [torch.randn((N, H, W, ci), device="cuda") for _ in range(6)]
    packed = torch.randn((L.pwc_conv3x3_wino_packed_floats(ci, co),), device="cuda") * 0.01
    bias = torch.zeros((co,), device="cuda"); y = torch.empty((N, H, W, co), device="cuda")
    def t():
        for x in xs: L.pwc_conv3x3_wino_f32(p(x), ci, p(packed), p(bias), p(y), co, N, H, W, ci, co, 1, 1, 0.1, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            for x in xs: L.pwc_conv3x3_wino_f32(p(x), ci, p(packed), p(bias), p(y), co, N, H, W, ci, co, 1, 1, 0.1, None)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / 30
    out = []
    for bn in ("", "16", "32"):
        for pers in ("0", "1"):
            if bn: os.environ["PWC_WINO_FORCE_BN"] = bn
            else: os.environ.pop("PWC_WINO_FORCE_BN", None)
            os.environ["PWC_WINO_PERSIST"] = pers
            out.append(f"bn{bn or 'auto'}/persist{pers}: {t():6.1f}")
    print(tag, " | ".join(out), flush=True)
