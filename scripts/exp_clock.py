"""Is the 128x128 conv kernel clock/power bound?  Same launch on random vs zero data."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pwcnet_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
N, H, W, ci, co = 8, 112, 256, 128, 128
ws = torch.empty(1 << 20, device="cuda")
def run(x, packed, iters=20, tile=0):
    y = torch.empty((N, H, W, co), device="cuda"); b = torch.zeros(co, device="cuda")
    f = lambda: L.pwc_conv3x3_f32(p(x), ci, p(packed), p(b), p(y), co, N, H, W, ci, co, 1, 1, 1, 0.1, tile, 1, p(ws), ws.numel(), None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    return us, 2.0 * N * H * W * 9 * ci * co / us / 1e6
nf = L.pwc_conv3x3_packed_floats(ci, co)
for name, x, w in [("random", torch.rand((N, H, W, ci), device="cuda") - 0.5, torch.rand(nf, device="cuda") - 0.5),
                   ("zeros", torch.zeros((N, H, W, ci), device="cuda"), torch.zeros(nf, device="cuda")),
                   ("random", torch.rand((N, H, W, ci), device="cuda") - 0.5, torch.rand(nf, device="cuda") - 0.5)]:
    for tile in (0, 5):
        us, tf = run(x, w, tile=tile)
        print(f"{name:7s} tile{tile}: {us:8.1f} us  {tf:6.1f} TFLOP/s")
