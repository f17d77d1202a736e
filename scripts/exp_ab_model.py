"""A/B timing of two model configurations in one process, interleaved (medians): python scripts/exp_ab_model.py
Toggles a class attribute of pwcnet_amd.modules._Module between forwards (plans are per net, so two nets are built)."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import modules as M, weights as W

attr = sys.argv[1] if len(sys.argv) > 1 else "f16x2_stream_k"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nets = {}
for val in (True, False):
    net = pwcnet_amd.PWCDCNet(streams=1)
    net.load_weights(W.init_weights(W.conv_specs(use_dc=False), seed=0))
    for mod in [net.fp_extractor, net.context] + net.of_estimators:
        setattr(mod, attr, val)
    if hasattr(net, attr):            # (model-level switches: two_operand, three_operand, ...)
        setattr(net, attr, val)
    nets[val] = net
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
for net in nets.values():
    for _ in range(3):
        net(im0, im1)
torch.cuda.synchronize()
times = {True: [], False: []}
for rnd in range(9):
    for val in (True, False):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            nets[val](im0, im1)
        e.record(); torch.cuda.synchronize()
        times[val].append(s.elapsed_time(e) / 10)
for val in (True, False):
    print(f"{attr}={val}: median {statistics.median(times[val]):.3f} ms per forward of {B} pairs (min {min(times[val]):.3f})")
a, b = nets[True](im0, im1)[0], nets[False](im0, im1)[0]
print("max |flow difference|", float((a - b).abs().max()), "max |flow|", float(a.abs().max()))
