"""Round 6: the x2 resize pair of the non-DC estimators (flows 2 ch + features 32 ch into the next level's buffer) on the
per-output-pixel kernel against the per-source-cell kernel (resize_pair2x_kernel, so far only from 64 feature channels on), and the
final x4 flow up-sampling on resize_x4_c2_kernel against the general kernel (a channel-strided destination takes that one).
Captured chains of 24 launches, median of 7.  usage: python scripts/exp_resize_ab.py [batch]"""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: PWC_RESIZE2X_MIN_CB exists only there
import ctypes, sys, torch
sys.path.insert(0, ".")
from pwcnet_amd import _lib
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"


def chain(fn, n=24, reps=7):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[reps // 2]


print(f"# batch {B}: resize pair (flow 2 + feat 32 -> a 128-channel buffer), us per launch")
for (h, w) in [(7, 16), (14, 32), (28, 64), (56, 128)]:
    fl = torch.randn((B, h, w, 2), device=dev); ft = torch.randn((B, h, w, 32), device=dev)
    E = torch.zeros((B, 2 * h, 2 * w, 128), device=dev)
    ya = ctypes.c_void_p(E.data_ptr() + 4 * 84); yb = ctypes.c_void_p(E.data_ptr() + 4 * 88)
    res = []
    for mincb in ("64", "32"):
        os.environ["PWC_RESIZE2X_MIN_CB"] = mincb
        f = lambda: _lib.check(L.pwc_resize_bilinear_pair_f32(p(fl), 2, ya, 128, p(ft), 32, yb, 128, B, h, w, 32, 2 * h, 2 * w, _lib.current_stream()))
        res.append(chain(f))
    print(f"  {h:3d} x {w:3d}: per output pixel {res[0]:6.2f}   per source cell {res[1]:6.2f}")
os.environ.pop("PWC_RESIZE2X_MIN_CB")
print("# final x4 up-sampling of the flows (x 20)")
for (h, w) in [(112, 256), (240, 480)]:
    x = torch.randn((B, h, w, 2), device=dev)
    y = torch.empty((B, 4 * h, 4 * w, 2), device=dev); yw = torch.empty((B, 4 * h, 4 * w, 4), device=dev)
    a = chain(lambda: _lib.check(L.pwc_resize_bilinear_f32(p(x), 2, p(y), 2, B, h, w, 2, 4 * h, 4 * w, 20.0, _lib.current_stream())))
    b = chain(lambda: _lib.check(L.pwc_resize_bilinear_f32(p(x), 2, p(yw), 4, B, h, w, 2, 4 * h, 4 * w, 20.0, _lib.current_stream())))
    print(f"  {h} x {w}: per source cell {a:6.2f}   general kernel (channel-strided destination) {b:6.2f}")
