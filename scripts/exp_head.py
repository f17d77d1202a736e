import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from pwcnet_amd import _lib
from pwcnet_amd.modules import _p
L = _lib.lib()
N,H,W = 8,112,256
x = torch.randn((N,H,W,32), device='cuda'); w = torch.randn((3,3,32,2), device='cuda')*0.05; b = torch.zeros(2, device='cuda')
y = torch.empty((N,H,W,2), device='cuda'); r = torch.randn((N,H,W,2), device='cuda')
s = _lib.current_stream()
for i in range(5):
    L.pwc_conv3x3_direct_f32(_p(x.data_ptr()),32,_p(w.data_ptr()),_p(b.data_ptr()),_p(y.data_ptr()),2,_p(r.data_ptr()),2,N,H,W,32,2,1,1,0,0.0,s)
torch.cuda.synchronize()
e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    L.pwc_conv3x3_direct_f32(_p(x.data_ptr()),32,_p(w.data_ptr()),_p(b.data_ptr()),_p(y.data_ptr()),2,_p(r.data_ptr()),2,N,H,W,32,2,1,1,0,0.0,s)
e1.record(); torch.cuda.synchronize()
print("head conv 8x112x256: %.1f us" % (e0.elapsed_time(e1)*1000/20))
