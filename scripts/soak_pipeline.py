"""Round 6: soak of pwcnet_amd.ForwardPipeline -- many forwards of mixed shapes over 3 lanes, every result compared bit for bit
with the one-stream forward of the same frames; any status flag (fp16 range, stream-K timeout) is a failure.
python scripts/soak_pipeline.py [forwards]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import weights as W

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
wts = W.init_weights(W.conv_specs(use_dc=False), seed=0)
net = pwcnet_amd.PWCDCNet()
net.load_weights(wts)
shapes = [(8, 448, 1024), (1, 448, 1024), (4, 448, 1024), (2, 960, 1920), (2, 256, 512), (8, 448, 1024)]
g = torch.Generator(device="cuda"); g.manual_seed(5)
cases = []
for (b, h, w) in shapes:
    a = torch.rand((b, h, w, 3), generator=g, device="cuda"); c = torch.rand((b, h, w, 3), generator=g, device="cuda")
    cases.append((a, c, net(a, c)[0].clone()))
torch.cuda.synchronize()
pipe = pwcnet_amd.ForwardPipeline(depth=3)
pipe.load_weights(wts)
bad = 0
t0 = time.perf_counter()
window = []
order = torch.randint(0, len(cases), (N,), generator=torch.Generator().manual_seed(9)).tolist()
for i, k in enumerate(order):
    a, c, ref = cases[k]
    window.append((i, k, pipe.submit(a, c)))
    if len(window) >= 48 or i == N - 1:
        rep = pipe.synchronize()
        if rep["flags"] or not rep["f16x2"]:
            print("STATUS", rep); bad += 1
        for j, kk, tk in window:
            if not torch.equal(tk.result()[0], cases[kk][2]):
                bad += 1
                print(f"forward {j} (shape {shapes[kk]}): max |diff| {float((tk.result()[0] - cases[kk][2]).abs().max()):.3e}")
        torch.cuda.synchronize()
        window = []
dt = time.perf_counter() - t0
print(f"{N} forwards of {len(shapes)} shapes in random order over {pipe.effective_depth} lanes: {bad} mismatches / flags, {dt:.1f} s "
      f"(compare step included)")
sys.exit(1 if bad else 0)
