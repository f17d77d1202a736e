import torch, ctypes
from pwcnet_amd import _lib
L=_lib.lib(); p=lambda t: ctypes.c_void_p(t.data_ptr())
x=torch.rand(2,8,16,2,device='cuda'); y=torch.empty(2,32,64,2,device='cuda'); st=torch.zeros(2,dtype=torch.int32,device='cuda')
print(L.pwc_resize_bilinear_status_f32(p(x),2,p(y),2,2,8,16,2,32,64,20.0,p(st),None)); torch.cuda.synchronize(); print(st, torch.isfinite(y).all())
import pwcnet_amd
from tests import util
net=pwcnet_amd.PWCDCNet(streams=1, range_check="off"); net.load_weights(util.model_weights(False))
im0,im1=util.smooth_images(2,128,192)
f,pyr=net(torch.from_numpy(im0).cuda(), torch.from_numpy(im1).cuda()); torch.cuda.synchronize()
print(torch.isfinite(f).all(), [bool(torch.isfinite(q).all()) for q in pyr])
