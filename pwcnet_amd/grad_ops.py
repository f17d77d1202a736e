"""Backward (gradient) ops of the PWC-Net forward on libpwc_hip.so -- op level of the training path
(reference train.py:66-92: tf.gradients over model.py / modules.py).

Every function takes/returns NHWC float32 CUDA tensors or Views (pointer + channel stride), like the
forward ops in modules.py; the C ABI is declared in include/pwc_hip.h ("f4: training path").
"""
import numpy as np
import torch

from . import _lib
from .modules import View, _p, _wino_pays, _h2_workspace, as_view

F16X2 = True      # training FORWARD: route the layers pwc_conv3x3_h2_supported names to conv3x3_h2 (activations of order 1)
# ... and the DATA GRADIENT (dy as the split operand): off by default.  The two-term fp16 split is exact to 22 bits
# only while both terms are normal fp16 numbers, i.e. for |x| >= ~2^-13; an upstream gradient of 1e-7 splits into subnormals with
# an absolute floor of ~1.5e-11 -- 3e-4 relative where fp32 carries 6e-8 -- and the reference is plain fp32.  Trainer(f16x2_dgrad=
# True) (or this flag) turns it on for runs whose gradients are known to stay large (-7 % step time at batch 8).
F16X2_DGRAD = False

_STATUS = {}


def status_words(dev):
    """The training path's status words on `dev` (two uint32, include/pwc_hip.h): every F16-pipe launch of conv3x3_raw ORs
    PWC_STATUS_STREAMK_TIMEOUT into word 0 when a bounded stream-K wait runs out (the launch said nothing before).
    Trainer.status() reads and clears them."""
    key = str(dev)
    w = _STATUS.get(key)
    if w is None:
        w = _STATUS[key] = torch.zeros(2, dtype=torch.int32, device=dev)
    return w


def read_status(clear=True):
    """OR of the status words of every device (synchronises); a stream-K timeout also refills the stream-K workspaces."""
    flags = 0
    for w in _STATUS.values():
        flags |= int(w[0].item())
        if clear:
            w.zero_()
    if flags & _lib.STATUS_STREAMK_TIMEOUT:
        from .modules import h2_workspaces_refill
        h2_workspaces_refill()
    return flags


def _L():
    return _lib.lib()


def _s():
    return _lib.current_stream()


def view_of(t):
    return as_view(t)[0]


def lrelu_grad_(y, dy, slope=0.1):
    """In place: dy *= (y > 0 ? 1 : slope)  (tf.nn.leaky_relu's gradient; y = the activation's output)."""
    assert (y.N, y.H, y.W, y.C) == (dy.N, dy.H, dy.W, dy.C)
    _lib.check(_L().pwc_lrelu_grad_f32(_p(y.ptr), y.cs, _p(dy.ptr), dy.cs, y.N * y.H * y.W, y.C, float(slope), _s()),
               "lrelu_grad")


def add_(src, dst, alpha=1.0, accumulate=True):
    """dst[..., :C] (+)= alpha * src[..., :C] on Views of one pixel grid."""
    assert (src.N, src.H, src.W, src.C) == (dst.N, dst.H, dst.W, dst.C), (src, dst)
    _lib.check(_L().pwc_add_f32(_p(src.ptr), src.cs, _p(dst.ptr), dst.cs, src.N * src.H * src.W, src.C, float(alpha),
                                1 if accumulate else 0, _s()), "add")


_WS = {}


def _ws(device, floats):
    """Scratch for the deterministic reductions (grown on demand, per device and stream)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    w = _WS.get(key)
    if w is None or w.numel() < floats:
        w = torch.empty((max(int(floats), 1 << 20),), dtype=torch.float32, device=device)
        _WS[key] = w
    return w


def channel_sums(dy, out, device, accumulate=False):
    """out[c] (+)= sum over pixels of dy[..., c]  (bias gradient)."""
    L = _L()
    npix = dy.N * dy.H * dy.W
    ws = _ws(device, L.pwc_channel_sums_workspace_floats(npix, dy.C))
    _lib.check(L.pwc_channel_sums_f32(_p(dy.ptr), dy.cs, npix, dy.C, _p(ws.data_ptr()), ws.numel(), _p(out.data_ptr()),
                                      1 if accumulate else 0, _s()), "channel_sums")


def lrelu_grad_channel_sums_(y, dy, out, device, slope=0.1, accumulate=False):
    """dy *= (y > 0 ? 1 : slope) in place and out[c] (+)= sum over pixels of the masked dy: one pass over dy."""
    L = _L()
    npix = dy.N * dy.H * dy.W
    ws = _ws(device, L.pwc_channel_sums_workspace_floats(npix, dy.C))
    _lib.check(L.pwc_lrelu_grad_channel_sums_f32(_p(y.ptr), y.cs, _p(dy.ptr), dy.cs, npix, dy.C, float(slope),
                                                 _p(ws.data_ptr()), ws.numel(), _p(out.data_ptr()), 1 if accumulate else 0,
                                                 _s()), "lrelu_grad_channel_sums")


def resize_grad(dy, dx, mul=1.0, accumulate=False):
    """dx (+)= mul * R^T dy for the TF-legacy bilinear resize of dx's grid to dy's (integer factor)."""
    assert dy.C == dx.C and dy.N == dx.N
    _lib.check(_L().pwc_resize_bilinear_grad_f32(_p(dy.ptr), dy.cs, _p(dx.ptr), dx.cs, dx.N, dx.H, dx.W, dx.C, dy.H, dy.W,
                                                 float(mul), 1 if accumulate else 0, _s()), "resize_grad")


_WARP_WS = {}


def warp_grad(x, flow, flow_scale, dy, dx=None, dflow=None, dflow_accumulate=False, deterministic=True):
    """Gradient of the bilinear warp: dx += corner scatter, dflow (+)= ... (both optional Views).
    deterministic (default): the scatter adds 64-bit fixed-point integers (2^-36 steps: contributions below 1.5e-11 vanish,
    |sums| up to 1.3e8) in a per-(device, stream) workspace, so the result does not depend on the order of the atomics
    (bit-reproducible training steps); a non-finite or out-of-range contribution makes the whole dx NaN instead of
    silently finite.  False: fp32 atomics (relative precision, order-dependent)."""
    L = _L()
    if deterministic and dx is not None:
        need = L.pwc_warp_bilinear_grad_workspace_bytes(x.N, x.H, x.W, x.C)
        dev = torch.cuda.current_device()        # Views carry raw pointers: the tensors live on the current device by contract
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _WARP_WS.get(key)
        if ws is None or ws.numel() * 8 < need:
            while len(_WARP_WS) >= 8:            # bounded: callers that come with a new stream every time
                _WARP_WS.pop(next(iter(_WARP_WS)))
            ws = _WARP_WS[key] = torch.empty(((need + 7) // 8,), dtype=torch.int64, device=torch.device("cuda", dev))
        _lib.check(L.pwc_warp_bilinear_grad_det_f32(
            _p(x.ptr), x.cs, _p(flow.ptr), flow.cs, float(flow_scale), _p(dy.ptr), dy.cs, _p(dx.ptr), dx.cs,
            _p(dflow.ptr) if dflow is not None else None, dflow.cs if dflow is not None else 0,
            1 if dflow_accumulate else 0, x.N, x.H, x.W, x.C, _p(ws.data_ptr()), ws.numel() * 8, _s()), "warp_grad")
        return
    _lib.check(L.pwc_warp_bilinear_grad_f32(
        _p(x.ptr), x.cs, _p(flow.ptr), flow.cs, float(flow_scale), _p(dy.ptr), dy.cs,
        _p(dx.ptr) if dx is not None else None, dx.cs if dx is not None else 0,
        _p(dflow.ptr) if dflow is not None else None, dflow.cs if dflow is not None else 0,
        1 if dflow_accumulate else 0, x.N, x.H, x.W, x.C, _s()), "warp_grad")


def cost_volume_grad(f0, f1w, cv, dcv, df0=None, df1w=None, accumulate=False, search_range=4, slope=0.1):
    _lib.check(_L().pwc_cost_volume_grad_f32(
        _p(f0.ptr), f0.cs, _p(f1w.ptr), f1w.cs, _p(cv.ptr), cv.cs, _p(dcv.ptr), dcv.cs,
        _p(df0.ptr) if df0 is not None else None, df0.cs if df0 is not None else 0,
        _p(df1w.ptr) if df1w is not None else None, df1w.cs if df1w is not None else 0,
        1 if accumulate else 0, f0.N, f0.H, f0.W, f0.C, int(search_range), float(slope), _s()), "cost_volume_grad")


def flow_norm_grad(pred, gt, dpred, gt_div=1.0, ord=2, scale=1.0, accumulate=False):
    """dpred (+)= scale * d/dpred sum_p ||pred - nearest_downsample(gt) / gt_div||_ord."""
    _lib.check(_L().pwc_flow_norm_grad_f32(_p(pred.ptr), pred.cs, _p(gt.ptr), gt.cs, pred.N, pred.H, pred.W, gt.H, gt.W,
                                           float(gt_div), int(ord), float(scale), _p(dpred.ptr), dpred.cs,
                                           1 if accumulate else 0, _s()), "flow_norm_grad")


def adam_step_(params, grads, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, l2_gamma=0.0, grad_scale=1.0):
    n = params.numel()
    assert grads.numel() == n and m.numel() == n and v.numel() == n
    _lib.check(_L().pwc_adam_step_f32(_p(params.data_ptr()), _p(grads.data_ptr()), _p(m.data_ptr()), _p(v.data_ptr()), n,
                                      float(lr_t), float(beta1), float(beta2), float(eps), float(l2_gamma),
                                      float(grad_scale), _s()), "adam_step")


def conv3x3_wgrad(x, dy, dw, cin, stride=1, dilation=1, cin_map=None, accumulate=False):
    """dw (3,3,cin,Cout) (+)= weight gradient; x: View over the PHYSICAL input channels, cin_map their
    physical->logical map (int32 tensor on the device, or None)."""
    L = _L()
    dev = dw.device
    ws = _ws(dev, L.pwc_conv3x3_wgrad_workspace_floats(x.N, x.H, x.W, x.C, dy.C, stride))
    _lib.check(L.pwc_conv3x3_wgrad_f32(_p(x.ptr), x.cs, _p(dy.ptr), dy.cs,
                                       _p(cin_map.data_ptr()) if cin_map is not None else None, int(cin), x.C, dy.C,
                                       _p(dw.data_ptr()), 1 if accumulate else 0, x.N, x.H, x.W, int(stride),
                                       int(dilation), _p(ws.data_ptr()), ws.numel(), _s()), "conv3x3_wgrad")


def conv3x3_raw(x, w_hwio, bias, y, stride=1, dilation=1, slope=None, keep=None, f16x2=None):
    """y = conv3x3_same(x, w_hwio) [+ bias] [leaky_relu] on the forward kernels; w_hwio (3,3,x.C,y.C) is given in
    the PHYSICAL channel order of x.  Packed weights are temporaries (appended to `keep`).  f16x2: may the F16-pipe kernels
    take it (None: the module's F16X2)."""
    L = _L()
    s = _s()
    dev = w_hwio.device
    cout = y.C
    if bias is None:
        bias = torch.zeros((cout,), dtype=torch.float32, device=dev)
    act, sl = (0, 0.0) if slope is None else (1, float(slope))
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    assert (y.H, y.W) == (Ho, Wo) and tuple(w_hwio.shape) == (3, 3, x.C, cout), (tuple(w_hwio.shape), x, y)
    w_hwio = w_hwio.contiguous()
    tmp = [w_hwio, bias]
    use_mfma = cout % 16 == 0 and x.C % 16 == 0 and x.cs % 4 == 0 and x.ptr % 16 == 0
    h2_ok = use_mfma and (F16X2 if f16x2 is None else f16x2) and cout % 32 == 0 and y.cs % 4 == 0 and y.ptr % 16 == 0
    if h2_ok and stride == 2 and dilation == 1 and L.pwc_conv3x3_h2_stride2_supported(x.N, x.H, x.W, x.C, cout):
        packed = torch.empty((L.pwc_conv3x3_h2_stride2_packed_floats(x.C, cout),), dtype=torch.float32, device=dev)
        _lib.check(L.pwc_conv3x3_h2_stride2_pack_f32(_p(w_hwio.data_ptr()), None, x.C, x.C, cout, _p(packed.data_ptr()), s), "h2 stride-2 pack")
        wsf = L.pwc_conv3x3_h2_stride2_workspace_floats(x.N, x.H, x.W, x.C, cout)
        ws = _h2_workspace(dev, wsf) if wsf else None
        _lib.check(L.pwc_conv3x3_h2_stride2_f32(_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.data_ptr()), _p(y.ptr), y.cs,
                                                x.N, x.H, x.W, x.C, cout, act, sl,
                                                _p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
                                                _p(status_words(dev).data_ptr()), s),
                   "conv3x3_h2 stride 2 (raw)")
        tmp.append(packed)
    elif (h2_ok and stride == 1
            and L.pwc_conv3x3_h2_supported(x.N, x.H, x.W, x.C, cout, dilation)):
        # the big stride-1 layers, forward and data gradient alike: direct convolution on the F16 matrix pipe with
        # exact-to-22-bit operand splits (more accurate than the fp32 Winograd kernel below; include/pwc_hip.h)
        packed = torch.empty((L.pwc_conv3x3_h2_packed_floats(x.C, cout),), dtype=torch.float32, device=dev)
        _lib.check(L.pwc_conv3x3_h2_pack_f32(_p(w_hwio.data_ptr()), None, x.C, x.C, cout, _p(packed.data_ptr()), s), "h2 pack")
        wsf = L.pwc_conv3x3_h2_workspace_floats(x.N, x.H, x.W, x.C, cout, dilation)
        ws = _h2_workspace(dev, wsf) if wsf else None
        _lib.check(L.pwc_conv3x3_h2_ex_f32(_p(x.ptr), x.cs, 0, None, 0, _p(packed.data_ptr()), _p(bias.data_ptr()), _p(y.ptr), y.cs,
                                           x.N, x.H, x.W, x.C, cout, dilation, act, sl,
                                           _p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
                                           _p(status_words(dev).data_ptr()), s),
                   "conv3x3_h2 (raw)")
        tmp.append(packed)
    elif use_mfma and stride == 1 and _wino_pays(L, x.N, x.H, x.W, cout, dilation):
        packed = torch.empty((L.pwc_conv3x3_wino_packed_floats(x.C, cout),), dtype=torch.float32, device=dev)
        _lib.check(L.pwc_conv3x3_wino_pack_f32(_p(w_hwio.data_ptr()), None, x.C, x.C, cout, _p(packed.data_ptr()), s), "wino pack")
        _lib.check(L.pwc_conv3x3_wino_f32(_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.data_ptr()), _p(y.ptr), y.cs,
                                          x.N, x.H, x.W, x.C, cout, dilation, act, sl, s), "conv3x3_wino (raw)")
        tmp.append(packed)
    elif use_mfma:
        packed = torch.empty((L.pwc_conv3x3_packed_floats(x.C, cout),), dtype=torch.float32, device=dev)
        _lib.check(L.pwc_conv3x3_pack_f32(_p(w_hwio.data_ptr()), None, x.C, x.C, cout, _p(packed.data_ptr()), s), "conv pack")
        ws = _ws(dev, L.pwc_conv3x3_workspace_floats(x.N * Ho * Wo, cout))
        _lib.check(L.pwc_conv3x3_f32(_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.data_ptr()), _p(y.ptr), y.cs,
                                     x.N, x.H, x.W, x.C, cout, stride, dilation, act, sl, -1, 0, _p(ws.data_ptr()),
                                     ws.numel(), s), "conv3x3 (raw)")
        tmp.append(packed)
    else:
        _lib.check(L.pwc_conv3x3_direct_f32(_p(x.ptr), x.cs, _p(w_hwio.data_ptr()), _p(bias.data_ptr()), _p(y.ptr), y.cs,
                                            None, 0, x.N, x.H, x.W, x.C, cout, stride, dilation, act, sl, s),
                   "conv3x3_direct (raw)")
    if keep is not None:
        keep.extend(tmp)
    return tmp


def conv3x3_dgrad(dy, w_hwio_phys, dx, stride=1, dilation=1, keep=None, dy_tensor=None):
    """dx = gradient of conv3x3_same w.r.t. its input.  w_hwio_phys: the forward kernel in the physical channel order
    of the input, (3,3,dx.C,dy.C) (zero rows for padding channels).  Stride 1: a 'SAME' convolution of dy with the
    flipped, transposed kernel.  Stride 2 (TF SAME on an even size pads bottom/right only: out[o] reads in[2o + t],
    t = 0..2): dy is spread onto the ODD positions of a zero map of the input's size -- dx[i] = sum_t U[i + 1 - t] w[t]
    with U[2o + 1] = dy[o] -- and that map is convolved the same way; needs dy as a dense tensor (dy_tensor)."""
    dev = w_hwio_phys.device
    wt = torch.flip(w_hwio_phys, dims=(0, 1)).permute(0, 1, 3, 2).contiguous()      # (3,3,Cout,Cin_phys)
    if stride == 1:
        return conv3x3_raw(dy, wt, None, dx, 1, dilation, None, keep, f16x2=F16X2_DGRAD)
    assert stride == 2 and dilation == 1 and dx.H == 2 * dy.H and dx.W == 2 * dy.W, "stride-2 dgrad: even input sizes"
    assert dy_tensor is not None and tuple(dy_tensor.shape) == (dy.N, dy.H, dy.W, dy.C)
    up = torch.zeros((dy.N, dx.H, dx.W, dy.C), dtype=torch.float32, device=dev)
    up.view(dy.N, dy.H, 2, dy.W, 2, dy.C)[:, :, 1, :, 1, :] = dy_tensor
    uv = View(up.data_ptr(), dy.C, dy.N, dx.H, dx.W, dy.C)
    tmp = conv3x3_raw(uv, wt, None, dx, 1, 1, None, keep, f16x2=F16X2_DGRAD)
    if keep is not None:
        keep.append(up)
    return tmp + [up]
