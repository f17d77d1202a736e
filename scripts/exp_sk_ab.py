"""The small-launch conv kernel (conv3x3_sk.hip) per layer shape and workgroup tile, graph replays of launch chains (round 5).
us per launch = one event pair around a captured chain of 64 launches over 4 rotating operand sets."""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the pwc_debug_* knobs exist only there (build it here first:
                                    # PWC_HARNESS=1 python -c 'from pwcnet_amd import _lib; _lib.build_library()')
import sys, torch
sys.path.insert(0, ".")
from pwcnet_amd import _lib
L = _lib.lib()
_p = lambda t: t.data_ptr()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("ext 112x256 32->64 s2", 2 * B, 112, 256, 32, 64, 2), ("ext 56x128 64->64", 2 * B, 56, 128, 64, 64, 1), ("ext 28x64 96->96", 2 * B, 28, 64, 96, 96, 1), ("ext 56x128 64->96 s2", 2 * B, 56, 128, 64, 96, 2), ("ext 28x64 96->128 s2", 2 * B, 28, 64, 96, 128, 2), ("ext 14x32 128->192 s2", 2 * B, 14, 32, 128, 192, 2), ("ext 7x16 192->192", 2 * B, 7, 16, 192, 192, 1),
          ("L0 288->128", B, 7, 16, 288, 128, 1), ("L0 128->128", B, 7, 16, 128, 128, 1), ("L0 128->96", B, 7, 16, 128, 96, 1),
          ("L0 96->64", B, 7, 16, 96, 64, 1), ("L0 64->32", B, 7, 16, 64, 32, 1),
          ("L1 256->128", B, 14, 32, 256, 128, 1), ("L1 128->128", B, 14, 32, 128, 128, 1), ("L1 128->96", B, 14, 32, 128, 96, 1),
          ("L1 96->64", B, 14, 32, 96, 64, 1), ("L1 64->32", B, 14, 32, 64, 32, 1),
          ("L2 224->128", B, 28, 64, 224, 128, 1), ("L2 128->128", B, 28, 64, 128, 128, 1), ("L2 64->32", B, 28, 64, 64, 32, 1)]
for name, N, H, W, cin, cout, stride in SHAPES:
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    k = torch.randn((3, 3, cin, cout), device="cuda") / (9 * cin) ** 0.5
    b = torch.randn((cout,), device="cuda")
    packed = torch.empty(L.pwc_conv3x3_sk_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_sk_pack_f32(_p(k), None, cin, cin, cout, _p(packed), None))
    xs = [torch.randn((N, H, W, cin), device="cuda") for _ in range(4)]
    ys = [torch.empty((N, Ho, Wo, cout), device="cuda") for _ in range(4)]
    line = f"{name:24s} M={N * Ho * Wo:6d}"
    for tile in (11, 21, 22, 31, 41, 42):
        if tile % 10 == 2 and cout % 32:
            continue
        if tile > 30 and stride != 1 and cin > 128:
            continue
        L.pwc_debug_conv3x3_sk_tile(tile)
        s = torch.cuda.current_stream().cuda_stream
        def run(i):
            _lib.check(L.pwc_conv3x3_sk_f32(_p(xs[i % 4]), cin, _p(packed), _p(b), _p(ys[i % 4]), cout, N, H, W, cin, cout, stride, 1, 1, 0.1,
                                            torch.cuda.current_stream().cuda_stream))
        run(0); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(64):
                run(i)
        g.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 64)
        line += f"   tile {tile}: {sorted(ts)[2]:6.2f}"
        del g
    print(line, flush=True)
L.pwc_debug_conv3x3_sk_tile(0)
