"""Loss functions of the reference (losses.py) on the HIP path -- forward values only.

Same names, argument order and meaning as reference losses.py:4-48.  Every function returns a
0-dim float32 tensor on the inputs' device.  The per-image sums of norms come from
pwc_flow_norm_sums_f32 (deterministic two-stage reduction); the handful of per-image scalars is
combined with torch (mean over the batch, level weights).

Note on multirobust_loss: the reference body (losses.py:45-46) computes `_l = L1loss(...)` and
then uses an undefined name `loss_level`, i.e. it raises NameError when called; what the code
evidently means -- weight * (L1 + epsilon)**q per level, Sec. 4 of the PWC-Net paper -- is what
is implemented here.
"""
import torch

from . import _lib
from .modules import _p, as_view


def _norm_sums(pred, gt, ord, gt_div=1.0):
    """Per-image sums over the pixels of ||pred - nearest_downsample(gt) / gt_div||_ord."""
    pv, pred = as_view(pred, "flows")
    gv, gt = as_view(gt, "flows_gt")
    assert pv.C == 2 and gv.C == 2 and pv.N == gv.N, "flows must be (N,h,w,2)"
    L = _lib.lib()
    ws = torch.empty((max(L.pwc_flow_norm_workspace_floats(pv.N, pv.H, pv.W), 1),), dtype=torch.float32, device=pred.device)
    out = torch.empty((pv.N,), dtype=torch.float32, device=pred.device)
    _lib.check(L.pwc_flow_norm_sums_f32(_p(pv.ptr), pv.cs, _p(gv.ptr), gv.cs, pv.N, pv.H, pv.W, gv.H, gv.W,
                                        float(gt_div), int(ord), _p(ws.data_ptr()), ws.numel(),
                                        _p(out.data_ptr()), _lib.current_stream()), "flow norm sums")
    return out, pv


def L1loss(x, y):   # shape(# batch, h, w, 2)
    """reference losses.py:4-5: mean over the batch of the per-image sum of L1 norms."""
    sums, _ = _norm_sums(y, x, 1)
    return sums.mean()


def L2loss(x, y):   # shape(# batch, h, w, 2)
    """reference losses.py:7-8."""
    sums, _ = _norm_sums(y, x, 2)
    return sums.mean()


def EPE(flows_gt, flows):
    """End point error (reference losses.py:11-13); both flows unscaled."""
    sums, v = _norm_sums(flows, flows_gt, 2)
    return sums.sum() / float(v.N * v.H * v.W)


def multiscale_loss(flows_gt, flows_pyramid, weights, name="multiscale_loss"):
    """reference losses.py:15-32: flows_gt unscaled; it is divided by 20 and
    nearest-neighbour-downsampled to every pyramid level inside."""
    loss = None
    for weight, fs in zip(weights, flows_pyramid):
        sums, _ = _norm_sums(fs, flows_gt, 2, gt_div=20.0)
        term = float(weight) * sums.mean()
        loss = term if loss is None else loss + term
    return loss


def multirobust_loss(flows_gt, flows_pyramid, weights, epsilon=0.01, q=0.4, name="multirobust_loss"):
    """reference losses.py:34-48 (see the module docstring about its undefined name)."""
    loss = None
    for weight, fs in zip(weights, flows_pyramid):
        sums, _ = _norm_sums(fs, flows_gt, 1, gt_div=20.0)
        term = float(weight) * (sums.mean() + float(epsilon)) ** float(q)
        loss = term if loss is None else loss + term
    return loss
