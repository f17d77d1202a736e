"""Round 6 experiment: ForwardPipeline lanes on HIP streams of DIFFERENT priorities (the high-priority lane's launches are dispatched
first: full-chip launches of two lanes then run one after the other instead of splitting the CUs, and the lanes fall out of step by
themselves).  python scripts/exp_pipeline_priority.py [batch]"""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import weights as W, model as M
from pwcnet_amd.pipeline import ForwardPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
wts = W.init_weights(W.conv_specs(use_dc=False), seed=0)
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
probe = torch.zeros(64, device=dev)


def pick(prios):
    main = torch.cuda.current_stream(dev)
    picked = []
    for p in prios:
        for _ in range(12):
            s = torch.cuda.Stream(device=dev, priority=p)
            if not M._shares_queue(dev, main, s, probe)[0] and not any(M._shares_queue(dev, q, s, probe)[0] for q in picked):
                picked.append(s)
                break
        else:
            return None
    return picked


STEPS = 60
for prios in ([0, 0, 0], [-1, 0, 0], [-1, -1, 0], [0, 0], [-1, 0], [-1, -1, -1], [0, 0, 0]):
    pipe = ForwardPipeline(depth=len(prios))
    pipe.load_weights(wts)
    lanes = pick(prios)
    torch.cuda.synchronize()
    if lanes is None:
        print(prios, "no vetted streams"); continue
    pipe._lanes = lanes; pipe.effective_depth = len(lanes)
    for _ in range(3 * len(prios)):
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
    # the driver's form too: 20 steps between synchronisations
    t20 = []
    for rnd in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            t = pipe.submit(im0, im1)
        torch.cuda.synchronize()
        t20.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"batch {B}, lane priorities {prios} (got {[s.priority for s in lanes]}): 60 steps median {statistics.median(ts):.3f} ms per forward (min {min(ts):.3f}); "
          f"20 steps median {statistics.median(t20):.3f} (min {min(t20):.3f}); flags {pipe.status()['flags']}")
    del pipe
