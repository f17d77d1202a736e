"""Host-side issue time of one forward (plan replay) vs GPU time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import weights as W
net = pwcnet_amd.PWCDCNet(streams=1); net.load_weights(W.init_weights(W.conv_specs(), seed=0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
print("batch", B)
for _ in range(3): net(im0, im1)
torch.cuda.synchronize()
plan = list(net._plans.values())[0]
print("launches per forward:", len(plan.calls))
for trial in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net(im0, im1)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0)/10:.2f} ms/forward, total {1e3*(t2-t0)/10:.2f} ms/forward")
# CUDA graph of the replay
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
try:
    with torch.cuda.stream(s):
        net2 = pwcnet_amd.PWCDCNet(streams=1, persistent_outputs=True); net2.load_weights(W.init_weights(W.conv_specs(), seed=0))
        for _ in range(2): net2(im0, im1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = net2(im0, im1)
    torch.cuda.synchronize()
    for trial in range(3):
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"graph replay total {1e3*(t2-t0)/10:.2f} ms/forward")
    ref = net(im0, im1)[0]
    print("graph output equals eager:", torch.equal(out[0], ref))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
