"""Two half-batches on two HIP streams vs one batch on one stream (batch 8, 448x1024).
    python scripts/exp_two_streams.py [N H W]
Variants: one stream; two free-running streams; two streams joined every step, the second half released when the
first has finished its extractor (so its latency-bound coarse levels overlap the other half's MFMA-bound layers)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import pwcnet_amd
from pwcnet_amd import modules as M

N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 448, 1024)
g = torch.Generator(device='cuda'); g.manual_seed(1)
im0 = torch.rand((N, H, W, 3), generator=g, device='cuda'); im1 = torch.rand((N, H, W, 3), generator=g, device='cuda')
STEPS = 30

def bench(fn, label):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS
    print(f"{label:60s} {dt*1e3:7.3f} ms/step  {N/dt:8.1f} pairs/s", flush=True)

net = pwcnet_amd.PWCDCNet(persistent_outputs=True)
bench(lambda: net(im0, im1), "one stream, batch 8")

for parts in (2, 4):
    if N % parts: continue
    n = N // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    nets = [pwcnet_amd.PWCDCNet(persistent_outputs=True) for _ in range(parts)]
    halves = [(im0[i*n:(i+1)*n], im1[i*n:(i+1)*n]) for i in range(parts)]
    def free():
        for s, nt, (a, b) in zip(streams, nets, halves):
            with torch.cuda.stream(s):
                nt(a, b)
    bench(free, f"{parts} free-running streams, batch {n} each")
    main = torch.cuda.current_stream()
    def joined():
        ev = torch.cuda.Event(); ev.record(main)
        for s, nt, (a, b) in zip(streams, nets, halves):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                nt(a, b)
        for s in streams:
            e = torch.cuda.Event(); e.record(s); main.wait_event(e)
    bench(joined, f"{parts} streams joined per step, batch {n} each")
    # staggered: stream i+1 starts when stream i has finished its extractor
    plans = []
    for s, nt, (a, b) in zip(streams, nets, halves):
        key = [k for k in nt._plans][0]
        plans.append(nt._plans[key])
    ext_end = max(i for i, c in enumerate(plans[0].calls) if 'fp_extractor' in str(c[2])) + 1
    print("extractor launches:", ext_end, "of", len(plans[0].calls))
    def staggered():
        ev = torch.cuda.Event(); ev.record(main)
        prev = None
        for s, pl in zip(streams, plans):
            s.wait_event(ev)
            if prev is not None: s.wait_event(prev)
            with torch.cuda.stream(s):
                for i, c in enumerate(pl.calls):
                    rc = c[0](*c[1])
                    assert rc == 0
                    if i + 1 == ext_end:
                        prev = torch.cuda.Event(); prev.record(s)
        for s in streams:
            e = torch.cuda.Event(); e.record(s); main.wait_event(e)
    bench(staggered, f"{parts} streams joined per step, staggered after the extractor")
