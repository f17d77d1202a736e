"""Round 6 experiment (harness build): the stream-K launches of conv3x3_h2 leave R CUs free (grid = CUs - R) so that the other
lanes' launch-bound kernels find a CU while a matrix-bound launch of this lane holds the rest.
PWC_HARNESS=1 python scripts/exp_pipeline_reserve.py [batch] [depth]"""
import os, sys, time, statistics
os.environ["PWC_HARNESS"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import _lib, weights as W
from pwcnet_amd.pipeline import ForwardPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = _lib.lib()
wts = W.init_weights(W.conv_specs(use_dc=False), seed=0)
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
STEPS = 60
ref = None
for R in (0, 8, 16, 32, 64, 0):
    L.pwc_debug_h2_reserve_cus(R)
    pipe = ForwardPipeline(depth=D)        # (plans record the grid: a fresh pipeline per setting)
    pipe.load_weights(wts)
    for _ in range(3 * D):
        t = pipe.submit(im0, im1)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            t = pipe.submit(im0, im1)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / STEPS * 1e3)
    out = t.result()[0]
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    print(f"batch {B}, depth {D}, {R:3d} CUs left free by conv3x3_h2: median {statistics.median(ts):.3f} ms per forward (min {min(ts):.3f}) = "
          f"{B / statistics.median(ts) * 1e3:.1f} pairs/s; max |flow - first| {float((out - ref).abs().max()):.2e}; flags {pipe.status()['flags']}")
    del pipe
