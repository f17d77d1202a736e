"""Per-launch timeline of one forward (batch 8, 448x1024): every launch bracketed by HIP events,
printed in issue order with its label.  Run on the GPU box: python scripts/exp_timeline.py [batch] [dc]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import modules as M, weights as W

DC = len(sys.argv) > 2 and sys.argv[2] == "dc"
net = pwcnet_amd.PWCDCNet(use_plans=False, use_dc=DC)
net.load_weights(W.init_weights(W.conv_specs(use_dc=DC), seed=0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
for _ in range(3):
    net(im0, im1)
torch.cuda.synchronize()
recs = []
orig = M._launch
def timed_launch(fn, args, what, kname=None, flops=0.0, nbytes=0.0, exec_flops=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); rc = fn(*args); e.record()
    M._lib.check(rc, what)
    recs.append((what, kname, flops, s, e))
M._launch = timed_launch
REP = 5
for _ in range(REP):
    net(im0, im1)
torch.cuda.synchronize()
n = len(recs) // REP
tot = 0.0
for i in range(n):
    us = min(recs[r * n + i][3].elapsed_time(recs[r * n + i][4]) for r in range(REP)) * 1e3
    what, kname, flops = recs[i][0], recs[i][1], recs[i][2]
    tot += us
    print(f"{i:3d} {us:8.1f} us  cum {tot:8.1f}  {flops / us / 1e6 if us > 0 else 0:7.1f} TF  {what}")
