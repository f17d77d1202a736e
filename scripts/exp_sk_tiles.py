"""Forward time with the small-launch conv kernel's workgroup tile pinned (round 5): python scripts/exp_sk_tiles.py [batch]"""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the pwc_debug_* knobs exist only there (build it here first:
                                    # PWC_HARNESS=1 python -c 'from pwcnet_amd import _lib; _lib.build_library()')
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import _lib, weights as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
nets = {}
for tile in (0, 11, 21, 22):
    L.pwc_debug_conv3x3_sk_tile(tile)
    net = pwcnet_amd.PWCDCNet(streams=1)
    net.load_weights(W.init_weights(W.conv_specs(use_dc=False), seed=0))
    for _ in range(3):
        net(im0, im1)          # the plan (a captured graph) keeps the tile it was built with
    nets[tile] = net
torch.cuda.synchronize()
L.pwc_debug_conv3x3_sk_tile(0)
times = {t: [] for t in nets}
for rnd in range(7):
    for t, net in nets.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            net(im0, im1)
        e.record(); torch.cuda.synchronize()
        times[t].append(s.elapsed_time(e) / 10)
for t in nets:
    print(f"tile {t:2d}: median {statistics.median(times[t]):.3f} ms per forward of {B} pairs (min {min(times[t]):.3f})")
