"""Side streams and hardware queues (as run at commit 61d1736, when BOTH sub-batches ran on side streams): which pair of torch
streams lets the two sub-batches overlap best?     python scripts/exp_queues.py <variant>"""
import sys, time, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import pwcnet_amd
variant = sys.argv[1]
N, H, W = 8, 448, 1024
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
im0 = torch.rand((N, H, W, 3), generator=g, device='cuda'); im1 = torch.rand((N, H, W, 3), generator=g, device='cuda')
net = pwcnet_amd.PWCDCNet()
keep = []
def used_stream(prio=0):
    s = torch.cuda.Stream(device=dev, priority=prio)
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device=dev) + 1)
    return s
if variant == "prio_mixed":
    net._side_streams[str(dev)] = [torch.cuda.Stream(device=dev, priority=0), torch.cuda.Stream(device=dev, priority=-1)]
elif variant.startswith("used_dummies_"):
    k = int(variant.rsplit("_", 1)[1])
    dummies = [used_stream() for _ in range(k)]
    torch.cuda.synchronize()
    net._side_streams[str(dev)] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
elif variant.startswith("pick_"):
    a, b = (int(v) for v in variant.split("_")[1:3])
    pool = [used_stream() for _ in range(8)]
    torch.cuda.synchronize()
    net._side_streams[str(dev)] = [pool[a], pool[b]]
for _ in range(10): net(im0, im1)
def bench(label, steps=120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): net(im0, im1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{label:24s} {dt*1e3:7.4f} ms/step  {N/dt:8.1f} pairs/s", flush=True)
bench(variant); bench(variant)
