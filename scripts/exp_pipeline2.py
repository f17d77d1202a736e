"""Round 6 experiment: consecutive forwards (whole batches) dealt to `depth` replicas of the model, each on a HIP stream with a
hardware queue of its own (pwcnet_amd.ForwardPipeline) -- the launch-bound coarse levels of one forward under the matrix-bound
launches of another.  python scripts/exp_pipeline2.py [batch] [depth ...]
Wall-clock over 60 forwards (host timer around issue + synchronize), medians of 5 rounds; depth 0 = the plain PWCDCNet loop."""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import weights as W
from pwcnet_amd.pipeline import ForwardPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
KS = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3, 4]
wts = W.init_weights(W.conv_specs(use_dc=False), seed=0)
im0 = torch.rand((B, 448, 1024, 3), device="cuda"); im1 = torch.rand((B, 448, 1024, 3), device="cuda")
STEPS = 60
ref = None
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for K in KS:
    if K == 0:
        pipe = pwcnet_amd.PWCDCNet(streams=1)
    else:
        pipe = ForwardPipeline(depth=K)
    pipe.load_weights(wts)
    outs = []

    def run(n):
        outs.clear()
        for i in range(n):
            outs.append(pipe(im0, im1)[0] if K == 0 else pipe.submit(im0, im1))
    run(3 * max(K, 1))
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(STEPS)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / STEPS * 1e3)
    res = [o if K == 0 else o.result()[0] for o in outs[-max(K, 1):]]
    torch.cuda.synchronize()
    st = pipe.status()
    if ref is None:
        ref = res[0].clone()
    diff = max(float((o - ref).abs().max()) for o in res)
    eff = getattr(pipe, "effective_depth", 0)
    print(f"batch {B}, depth {K} (streams vetted: {eff}): median {statistics.median(ts):.3f} ms per forward (min {min(ts):.3f}) = "
          f"{B / statistics.median(ts) * 1e3:.1f} pairs/s; max |flow - first configuration's| {diff:.2e}; flags {st['flags']}")
    del pipe, outs, res
