"""Training step under rocprofv3 (kernel stats): python scripts/exp_train_profile.py [N H W [dc]]"""
import sys, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from pwcnet_amd.train import Trainer
N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 448, 1024)
tn = Trainer(use_dc=len(sys.argv) > 4 and sys.argv[4] == "dc")
g = torch.Generator(device='cuda'); g.manual_seed(1)
im0 = torch.rand((N, H, W, 3), generator=g, device='cuda'); im1 = torch.rand((N, H, W, 3), generator=g, device='cuda')
gt = torch.randn((N, H, W, 2), generator=g, device='cuda') * 3
for _ in range(4):
    tn.step(im0, im1, gt)
torch.cuda.synchronize()
