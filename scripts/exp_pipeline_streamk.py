"""Round 6 experiment: under the ForwardPipeline, the F16-pipe convolutions as one workgroup per tile (CUs free up at every tile end,
another lane's small kernels can slip in) against stream-K (one persistent workgroup per CU for the whole launch).
python scripts/exp_pipeline_streamk.py [batch] [depth]"""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import weights as W
from pwcnet_amd.pipeline import ForwardPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wts = W.init_weights(W.conv_specs(use_dc=False), seed=0)
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
STEPS = 60
for sk in (True, False, True, False):
    pipe = ForwardPipeline(depth=D)
    pipe.load_weights(wts)
    for net in pipe.nets:
        for mod in net._mods:
            mod.f16x2_stream_k = sk
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
    print(f"batch {B}, depth {D}, stream-K {'on ' if sk else 'off'}: median {statistics.median(ts):.3f} ms per forward (min {min(ts):.3f}); flags {pipe.status()['flags']}")
    del pipe
