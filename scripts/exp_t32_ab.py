"""The thin-input conv kernel (conv3x3_t32_kernel) on the full-resolution extractor layers (round 5): graph replays of launch chains
over operand sets rotating through 400 MB; us per launch, median of 5."""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the pwc_debug_* knobs exist only there (build it here first:
                                    # PWC_HARNESS=1 python -c 'from pwcnet_amd import _lib; _lib.build_library()')
import sys, torch
sys.path.insert(0, ".")
from pwcnet_amd import _lib
L = _lib.lib()
_p = lambda t: t.data_ptr()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("conv2d_3 16->32 s2", 2 * B, 224, 512, 16, 2), ("conv2d_4 32->32", 2 * B, 112, 256, 32, 1)]
for name, N, H, W, cin, stride in SHAPES:
    cout = 32
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    k = torch.randn((3, 3, cin, cout), device="cuda") / (9 * cin) ** 0.5
    b = torch.randn((cout,), device="cuda")
    packed = torch.empty(L.pwc_conv3x3_t32_packed_floats(cin), device="cuda")
    _lib.check(L.pwc_conv3x3_t32_pack_f32(_p(k), None, cin, cin, _p(packed), None))
    nset = max(2, int(400e6 // (4 * N * (H * W * cin + Ho * Wo * cout))) + 1)
    xs = [torch.randn((N, H, W, cin), device="cuda") for _ in range(nset)]
    ys = [torch.empty((N, Ho, Wo, cout), device="cuda") for _ in range(nset)]
    def run(i):
        _lib.check(L.pwc_conv3x3_t32_f32(_p(xs[i % nset]), cin, _p(packed), _p(b), _p(ys[i % nset]), cout, N, H, W, cin, cout, stride, 1, 0.1,
                                         torch.cuda.current_stream().cuda_stream))
    line = f"{name:24s} M={N * Ho * Wo:7d}"
    for dbg in (0, 1, 2, 4, 3, 6, 5, 7, 15):
        L.pwc_debug_conv3x3_t32(dbg)
        run(0); torch.cuda.synchronize()
        n = max(8, 2 * nset)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                run(i)
        g.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        line += f"  dbg{dbg}: {sorted(ts)[2]:6.2f}"
        del g
    L.pwc_debug_conv3x3_t32(0)
    print(line + "   (1 no fetches, 2 no matrix work, 4 no stores, 8 no split pass)", flush=True)
