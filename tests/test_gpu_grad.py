"""GPU tests of the training path: every backward kernel and the assembled train step
against torch.autograd on the float64 CPU restatement of the forward (oracle/torch_ref.py).

Tolerances are relative to the largest reference gradient (fp32 kernels vs a float64 reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def go():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (torch.cuda.is_available() is False)")
    from pwcnet_amd import grad_ops
    return grad_ops


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.RandomState(seed).uniform(lo, hi, size=shape).astype(np.float32)


def gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def t64(a, grad=True):
    return torch.tensor(np.asarray(a, np.float64), dtype=torch.float64, requires_grad=grad)


def close(got, exp, rel=2e-5):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else got
    exp = exp.detach().cpu().double().numpy() if isinstance(exp, torch.Tensor) else exp
    assert got.shape == exp.shape, (got.shape, exp.shape)
    tol = rel * max(float(np.abs(exp).max()), 1e-6)
    err = float(np.abs(got - exp).max())
    assert err <= tol, f"max abs err {err:.3e} > tol {tol:.3e} (max |ref| {float(np.abs(exp).max()):.3e})"


def View(*a):
    from pwcnet_amd.modules import View as _V
    return _V(*a)


def V(t):
    from pwcnet_amd.modules import as_view
    return as_view(t)[0]


def test_lrelu_grad_and_channel_sums(go):
    y, dy = rnd((2, 5, 7, 16), 1), rnd((2, 5, 7, 16), 2)
    gy, gdy = gpu(y), gpu(dy)
    go.lrelu_grad_(V(gy), V(gdy))
    close(gdy, dy * np.where(y > 0, 1.0, 0.1))
    out = torch.zeros(16, device="cuda")
    go.channel_sums(V(gdy), out, gdy.device)
    close(out, gdy.cpu().numpy().reshape(-1, 16).sum(0), rel=1e-5)
    # fused form, awkward channel counts and pixel counts
    for (n, h, w, c) in [(2, 33, 47, 96), (1, 7, 16, 192), (3, 20, 31, 2), (1, 64, 64, 300)]:
        y2, d2 = rnd((n, h, w, c), 3 + c), rnd((n, h, w, c), 4 + c)
        g2y, g2d = gpu(y2), gpu(d2)
        o2 = torch.zeros(c, device="cuda")
        go.lrelu_grad_channel_sums_(V(g2y), V(g2d), o2, g2d.device)
        exp = d2 * np.where(y2 > 0, 1.0, 0.1)
        close(g2d, exp)
        close(o2, exp.reshape(-1, c).astype(np.float64).sum(0), rel=2e-5)


@pytest.mark.parametrize("k", [2, 4])
def test_resize_grad(go, k):
    x = rnd((2, 5, 6, 8), 3)
    dy = rnd((2, 5 * k, 6 * k, 8), 4)
    xt = t64(x)
    (tr.resize_legacy(xt, (5 * k, 6 * k)) * t64(dy, False)).sum().backward()
    dx = torch.zeros((2, 5, 6, 8), device="cuda")
    gdy = gpu(dy)                       # (Views do not own memory: keep every tensor alive in a variable)
    go.resize_grad(V(gdy), V(dx))
    close(dx, xt.grad)
    go.resize_grad(V(gdy), V(dx), mul=0.5, accumulate=True)
    close(dx, 1.5 * xt.grad)


def test_warp_grad(go):
    N, H, W, C = 2, 12, 20, 8
    x, dy = rnd((N, H, W, C), 5), rnd((N, H, W, C), 6)
    flow = util.flow_field(N, H, W, seed=7, sigma=2.0) + 0.137     # keep away from exact integers
    scale = 1.25
    xt, ft = t64(x), t64(flow)
    (tr.bilinear_warp(xt, ft * scale) * t64(dy, False)).sum().backward()
    dx = torch.zeros((N, H, W, C), device="cuda")
    dfl = torch.zeros((N, H, W, 2), device="cuda")
    gx, gfl, gdy = gpu(x), gpu(flow), gpu(dy)
    go.warp_grad(V(gx), V(gfl), scale, V(gdy), V(dx), V(dfl))
    close(dx, xt.grad, rel=1e-5)
    close(dfl, ft.grad, rel=1e-4)


def test_warp_grad_is_bit_reproducible(go):
    """Hundreds of source pixels collapse onto a few corners (a flow field that points at one spot): the default
    scatter (64-bit fixed-point integer atomics) gives the same bits on every run and matches autograd; it also
    accumulates into a non-zero dx."""
    N, H, W, C = 2, 40, 56, 32
    x, dy = rnd((N, H, W, C), 15), rnd((N, H, W, C), 16)
    gy, gx_ = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    flow = np.stack([(20.3 - gx_) * 0.97, (17.6 - gy) * 0.97], axis=2)[None].repeat(N, 0).astype(np.float32)
    flow += rnd((N, H, W, 2), 17) * 0.3
    xt, ft = t64(x), t64(flow)
    (tr.bilinear_warp(xt, ft) * t64(dy, False)).sum().backward()
    gx, gfl, gdy = gpu(x), gpu(flow), gpu(dy)
    outs = []
    for _ in range(4):
        dx = torch.zeros((N, H, W, C), device="cuda")
        go.warp_grad(V(gx), V(gfl), 1.0, V(gdy), V(dx), None)
        torch.cuda.synchronize()
        outs.append(dx.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    close(outs[0], xt.grad, rel=1e-5)
    base = gpu(rnd((N, H, W, C), 18))
    dx = base.clone()
    go.warp_grad(V(gx), V(gfl), 1.0, V(gdy), V(dx), None)
    close(dx - base, xt.grad, rel=2e-5)      # (fp32 cancellation of the base in dx - base)
    dxa = torch.zeros((N, H, W, C), device="cuda")
    go.warp_grad(V(gx), V(gfl), 1.0, V(gdy), V(dxa), None, deterministic=False)      # fp32 atomics: same sums up to order
    close(dxa, xt.grad, rel=1e-5)


def test_warp_grad_deterministic_form_does_not_hide_divergence(go):
    """the fixed-point scatter maps NaN to 0 and saturates Inf -- a non-finite (or > 2^26) upstream gradient
    must come out as NaN in dx, not as finite numbers."""
    N, H, W, C = 1, 12, 20, 8
    rs = np.random.RandomState(3)
    x, dy = rs.randn(N, H, W, C).astype(np.float32), rs.randn(N, H, W, C).astype(np.float32)
    fl = (rs.randn(N, H, W, 2) * 2).astype(np.float32)
    for bad in (np.nan, np.inf, 1e9):
        d = dy.copy(); d[0, 5, 7, 3] = bad
        gx, gfl, gdy = gpu(x), gpu(fl), gpu(d)
        dx = torch.zeros((N, H, W, C), device="cuda")
        go.warp_grad(V(gx), V(gfl), 1.0, V(gdy), V(dx), None)
        torch.cuda.synchronize()
        assert bool(torch.isnan(dx).all()), bad
    gx, gfl, gdy = gpu(x), gpu(fl), gpu(dy)
    dx = torch.zeros((N, H, W, C), device="cuda")
    go.warp_grad(V(gx), V(gfl), 1.0, V(gdy), V(dx), None)
    assert bool(torch.isfinite(dx).all())


def test_train_step_is_bit_reproducible():
    """Two trainers from the same weights on the same batch: identical losses and identical parameters after 3 steps
    (fixed-order reductions everywhere, integer atomics in the warp gradient)."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from pwcnet_amd.train import Trainer
    N, H, W = 2, 64, 128
    w = util.model_weights(False, gain=1.25)
    im0, im1 = util.smooth_images(N, H, W, seed=71, shift=(3, -2))
    gt = util.flow_field(N, H, W, seed=72, sigma=2.0, outliers=False).astype(np.float32)
    g0, g1, ggt = gpu(im0), gpu(im1), gpu(gt)
    runs = []
    for _ in range(2):
        tn = Trainer(weights=(0.32, 0.08, 0.02, 0.01, 0.005), gamma=4e-4, lr=1e-3)
        tn.load_weights(w)
        losses = [float(tn.step(g0, g1, ggt)) for _ in range(3)]
        runs.append((losses, tn.state_dict()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert np.array_equal(np.asarray(runs[0][1][k]), np.asarray(runs[1][1][k])), k


@pytest.mark.parametrize("N,H,W,C", [(1, 9, 11, 8), (2, 16, 24, 32), (1, 8, 8, 20)])
def test_cost_volume_grad(go, N, H, W, C):
    f0, f1, dcv = rnd((N, H, W, C), 8), rnd((N, H, W, C), 9), rnd((N, H, W, 81), 10)
    a, b = t64(f0), t64(f1)
    cv = tr.cost_volume(a, b)
    (cv * t64(dcv, False)).sum().backward()
    df0 = torch.zeros((N, H, W, C), device="cuda")
    df1 = torch.zeros((N, H, W, C), device="cuda")
    g0, g1, gcv, gdcv = gpu(f0), gpu(f1), gpu(cv.detach().numpy()), gpu(dcv)
    go.cost_volume_grad(V(g0), V(g1), V(gcv), V(gdcv), V(df0), V(df1))
    close(df0, a.grad)
    close(df1, b.grad)
    # cv / dcv as 16-byte aligned slices of wider buffers (the estimator layouts): the float4 path
    wcv = torch.zeros((N, H, W, 96), device="cuda"); wcv[..., 4:85] = gcv
    wd = torch.zeros((N, H, W, 100), device="cuda"); wd[..., 8:89] = gdcv
    vcv = View(wcv.data_ptr() + 16, 96, N, H, W, 81)
    vd = View(wd.data_ptr() + 32, 100, N, H, W, 81)
    df0b = torch.zeros((N, H, W, C), device="cuda")
    df1b = torch.zeros((N, H, W, C), device="cuda")
    go.cost_volume_grad(V(g0), V(g1), vcv, vd, V(df0b), V(df1b))
    assert torch.equal(df0b, df0) and torch.equal(df1b, df1)


def test_flow_norm_grad(go):
    N, H, W = 2, 16, 24
    pred, gt = rnd((N, 4, 6, 2), 11), rnd((N, H, W, 2), 12) * 20
    pt = t64(pred)
    loss = 0.32 * tr.L2loss(tr.resize_nearest(t64(gt, False) / 20.0, (4, 6)), pt)
    loss.backward()
    dp = torch.zeros((N, 4, 6, 2), device="cuda")
    gpred, ggt = gpu(pred), gpu(gt)
    go.flow_norm_grad(V(gpred), V(ggt), V(dp), gt_div=20.0, ord=2, scale=0.32 / N)
    close(dp, pt.grad)


@pytest.mark.parametrize("stride,dil,H,W,cin,cout", [
    (1, 1, 12, 16, 16, 32), (1, 1, 9, 11, 48, 16), (2, 1, 12, 16, 3, 16), (2, 1, 8, 8, 32, 64),
    (1, 4, 16, 16, 64, 64), (1, 1, 6, 10, 160, 128), (1, 1, 7, 9, 32, 2), (1, 16, 20, 36, 96, 64)])
def test_conv_wgrad_and_dgrad(go, stride, dil, H, W, cin, cout):
    N = 2
    x, k, b = rnd((N, H, W, cin), 13), rnd((3, 3, cin, cout), 14) * 0.2, rnd((cout,), 15)
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = rnd((N, Ho, Wo, cout), 16)
    xt, kt, bt = t64(x), t64(k), t64(b)
    (tr.conv3x3_same(xt, kt, bt, stride, dil) * t64(dy, False)).sum().backward()
    gx, gdy = gpu(x), gpu(dy)
    dw = torch.zeros((3, 3, cin, cout), device="cuda")
    go.conv3x3_wgrad(V(gx), V(gdy), dw, cin, stride, dil)
    close(dw, kt.grad, rel=3e-5)
    db = torch.zeros((cout,), device="cuda")
    go.channel_sums(V(gdy), db, gdy.device)
    close(db, bt.grad, rel=1e-5)
    if cin % 4 == 0:
        dx = torch.zeros((N, H, W, cin), device="cuda")
        gk = gpu(k)
        keep = []
        go.conv3x3_dgrad(V(gdy), gk, V(dx), stride, dil, keep=keep, dy_tensor=gdy)
        torch.cuda.synchronize()
        close(dx, xt.grad, rel=3e-5)


@pytest.mark.parametrize("scale", [1.0, 1e-6, 1e-8])
def test_conv_dgrad_of_small_gradients(go, scale):
    """upstream gradients of 1e-6 .. 1e-8 (what a multiscale loss averaged over a batch hands to the deep layers) at a
    shape the F16-pipe kernel would take (8 x 56 x 128, 128 -> 128).  The data gradient runs on the fp32 kernels by default
    (grad_ops.F16X2_DGRAD = False) and meets a float64 convolution at fp32 accuracy at every scale; with the split kernel
    switched on the error grows as the scale falls (the fp16 pairs leave the normal range): recorded, not asserted to be small."""
    N, H, W, cin, cout = 8, 56, 128, 128, 128
    k = rnd((3, 3, cin, cout), 24) * 0.05
    dy = rnd((N, H, W, cout), 26) * scale
    wt = np.flip(k, (0, 1)).transpose(0, 1, 3, 2).copy()
    ref = tr.conv3x3_same(t64(dy[:1], False), t64(wt, False), None, 1, 1).numpy()
    gdy, gk = gpu(dy), gpu(k)
    errs = {}
    for flag in (False, True):
        go.F16X2_DGRAD = flag
        try:
            dx = torch.zeros((N, H, W, cin), device="cuda")
            go.conv3x3_dgrad(V(gdy), gk, V(dx), 1, 1, keep=[], dy_tensor=gdy)
            torch.cuda.synchronize()
        finally:
            go.F16X2_DGRAD = False
        errs[flag] = float(np.abs(dx[:1].double().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    print(f"dgrad at |dy| ~ {scale:g}: relative error fp32 kernels {errs[False]:.2e}, F16-pipe split {errs[True]:.2e}")
    assert errs[False] <= 3e-5, errs
    if scale == 1.0:
        assert errs[True] <= 3e-5, errs


@pytest.mark.parametrize("N,dil,H,W,cin,cout", [
    (2, 1, 64, 128, 128, 128), (2, 4, 64, 128, 96, 64), (4, 1, 50, 70, 40, 36), (1, 16, 112, 256, 96, 64),
    (2, 2, 33, 47, 64, 100), (3, 1, 28, 64, 160, 32), (8, 8, 24, 20, 32, 32), (2, 1, 96, 160, 16, 16), (1, 1, 130, 77, 16, 16), (1, 1, 32, 64, 2112, 64)])
def test_conv_wgrad_lds_staged_kernel(go, N, dil, H, W, cin, cout):
    """Shapes the LDS-staged weight-gradient kernel takes (stride 1, >= 32 channels, enough 64-pixel tiles): full and
    ragged tiles, every tile shape (64x1 ... 8x8), dilation sub-lattices that do not divide the image."""
    x, dy = rnd((N, H, W, cin), 23), rnd((N, H, W, cout), 24)
    xt = t64(x, False)
    kt = torch.zeros((3, 3, cin, cout), dtype=torch.float64, requires_grad=True)
    (tr.conv3x3_same(xt, kt, torch.zeros(cout, dtype=torch.float64), 1, dil) * t64(dy, False)).sum().backward()
    gx, gdy = gpu(x), gpu(dy)
    dw = torch.full((3, 3, cin, cout), 7.0, device="cuda")
    go.conv3x3_wgrad(V(gx), V(gdy), dw, cin, 1, dil)
    close(dw, kt.grad, rel=3e-5)
    # fixed summation order (k-split partials, pixel parts): bit-identical on every run
    for _ in range(3):
        again = torch.full((3, 3, cin, cout), -3.0, device="cuda")
        go.conv3x3_wgrad(V(gx), V(gdy), again, cin, 1, dil)
        assert torch.equal(again, dw)
    # strided views (the dense-connection buffers): x and dy as channel slices of wider tensors
    if cin % 8 == 0:
        wide_x = torch.zeros((N, H, W, cin + 16), device="cuda"); wide_x[..., 8:8 + cin] = gx
        wide_d = torch.zeros((N, H, W, cout + 8), device="cuda"); wide_d[..., 4:4 + cout] = gdy
        vx = View(wide_x.data_ptr() + 32, cin + 16, N, H, W, cin)
        vd = View(wide_d.data_ptr() + 16, cout + 8, N, H, W, cout)
        dw2 = torch.zeros((3, 3, cin, cout), device="cuda")
        go.conv3x3_wgrad(vx, vd, dw2, cin, 1, dil)
        close(dw2, kt.grad, rel=3e-5)


def test_conv_wgrad_with_channel_map(go):
    """input in a padded physical layout (ChannelLayout): gradient comes back in the variable's logical order."""
    N, H, W = 1, 10, 12
    phys2log = np.array([0, 1, 2, -1, 3, 4, 5, 6, 7, 8, -1, -1, 9, 10, -1, -1], np.int32)    # 11 logical in 16 physical
    xl = rnd((N, H, W, 11), 17)
    xp = np.zeros((N, H, W, 16), np.float32)
    xp[..., phys2log >= 0] = xl[..., phys2log[phys2log >= 0]]
    k, dy = rnd((3, 3, 11, 32), 18), rnd((N, H, W, 32), 19)
    xt, kt = t64(xl), t64(k)
    (tr.conv3x3_same(xt, kt, torch.zeros(32, dtype=torch.float64)) * t64(dy, False)).sum().backward()
    dw = torch.zeros((3, 3, 11, 32), device="cuda")
    gxp, gdy, gmap = gpu(xp), gpu(dy), torch.from_numpy(phys2log).cuda()
    go.conv3x3_wgrad(V(gxp), V(gdy), dw, 11, 1, 1, cin_map=gmap)
    close(dw, kt.grad, rel=3e-5)


def test_adam_step(go):
    n = 1000
    p, g = rnd((n,), 20), rnd((n,), 21)
    m, v = np.abs(rnd((n,), 22)) * 0.1, np.abs(rnd((n,), 23)) * 0.01
    gp, gg, gm, gv = gpu(p), gpu(g), gpu(m), gpu(v)
    import math
    step, lr, gamma = 7, 1e-4, 4e-4
    lr_t = lr * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
    go.adam_step_(gp, gg, gm, gv, lr_t, l2_gamma=gamma)
    pt, mt, vt = tr.adam_step(t64(p, False), t64(g, False) + gamma * t64(p, False), t64(m, False), t64(v, False), step, lr)
    close(gp, pt, rel=1e-6)
    close(gm, mt, rel=1e-6)
    close(gv, vt, rel=3e-6)


# ------------------------------------------------------------------ assembled training step
def _ref_grads(w, im0, im1, gt, weights, use_dc=False, loss="multiscale"):
    wt = {k: t64(v) for k, v in w.items()}
    _, pyr = tr.TorchPWCDCNet(wt, use_dc=use_dc)(t64(im0, False), t64(im1, False))
    if loss == "multiscale":
        loss = tr.multiscale_loss(t64(gt, False), pyr, weights)
    else:
        loss = tr.multirobust_loss(t64(gt, False), pyr, weights, epsilon=0.02, q=0.4)
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in wt.items()}, [p.detach() for p in pyr]


@pytest.mark.parametrize("use_dc,loss", [(False, "multiscale"), (True, "multiscale"), (False, "robust"), (True, "robust")])
def test_train_step_gradients_vs_autograd(use_dc, loss):
    """Whole backward (loss gradient, context, 5 estimators -- plain and densely connected --, cost volumes, warps,
    resizes, shared-weight extractor) against torch.autograd on the float64 restatement: every variable's gradient."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from pwcnet_amd.train import Trainer
    N, H, W = 2, 64, 128
    w = util.model_weights(use_dc, gain=1.25)
    im0, im1 = util.smooth_images(N, H, W, seed=61, shift=(3, -2))
    gt = (util.flow_field(N, H, W, seed=62, sigma=2.0, outliers=False)).astype(np.float32)
    weights = (0.32, 0.08, 0.02, 0.01, 0.005)
    ref_loss, ref_g, ref_pyr = _ref_grads(w, im0, im1, gt, weights, use_dc, loss)
    tn = Trainer(weights=weights, gamma=0.0, lr=1e-4, use_dc=use_dc, loss=loss, epsilon=0.02, q=0.4)
    tn.load_weights(w)
    pyr = tn.forward(gpu(im0), gpu(im1))
    for a, b in zip(pyr, ref_pyr):
        close(a, b, rel=2e-4)
    ggt = gpu(gt)
    loss = float(tn.loss_value(ggt))
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    tn.backward(ggt)
    torch.cuda.synchronize()
    got = tn.gradients()
    worst = 0.0
    for k in sorted(ref_g):
        r = ref_g[k].numpy()
        err = float(np.abs(got[k] - r).max()) / max(float(np.abs(r).max()), 1e-12)
        worst = max(worst, err)
        assert err <= 2e-3, f"{k}: relative gradient error {err:.3e}"
    print(f"worst relative gradient error over {len(ref_g)} variables: {worst:.3e}")


def test_train_step_gradients_full_size_spot_check():
    """One 448x1024 pair (the geometry `bench.py --mode train` times: LDS-staged weight-gradient tiling with its k-split
    over pixel chunks, Winograd data gradients, full-size cost-volume / warp gradients): the gradients of the first
    conv, a level-4 128 -> 128 conv, the dilated context convs, a stride-2 extractor conv -- in fact every variable --
    against float64 autograd on the torch restatement (about a minute of CPU)."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from pwcnet_amd.train import Trainer
    N, H, W = 1, 448, 1024
    w = util.model_weights(False, gain=1.2)
    im0, im1 = util.smooth_images(N, H, W, seed=81, shift=(4, -3))
    gt = util.flow_field(N, H, W, seed=82, sigma=2.0, outliers=False).astype(np.float32)
    weights = (0.32, 0.08, 0.02, 0.01, 0.005)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref_loss, ref_g, ref_pyr = _ref_grads(w, im0, im1, gt, weights, False, "multiscale")
    tn = Trainer(weights=weights, gamma=0.0, lr=1e-4)
    tn.load_weights(w)
    pyr = tn.forward(gpu(im0), gpu(im1))
    for a, b in zip(pyr, ref_pyr):
        close(a, b, rel=2e-4)
    ggt = gpu(gt)
    loss = float(tn.loss_value(ggt))
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    tn.backward(ggt)
    torch.cuda.synchronize()
    got = tn.gradients()
    named = ["pwcdcnet/fp_extractor/conv2d/kernel", "pwcdcnet/optflow_4/conv2d_1/kernel", "pwcdcnet/context/conv2d_2/kernel",
             "pwcdcnet/context/conv2d_4/kernel", "pwcdcnet/fp_extractor/conv2d_3/kernel", "pwcdcnet/optflow_4/conv2d_5/kernel"]
    worst = 0.0
    for k in sorted(ref_g):
        r = ref_g[k].numpy()
        err = float(np.abs(got[k] - r).max()) / max(float(np.abs(r).max()), 1e-12)
        worst = max(worst, err)
        if k in named:
            print(f"  {k}: relative gradient error {err:.3e}")
        assert err <= 2e-3, f"{k}: relative gradient error {err:.3e}"
    print(f"full size: worst relative gradient error over {len(ref_g)} variables: {worst:.3e}")


def test_train_step_reduces_the_loss_and_matches_adam():
    """A few optimisation steps on one batch: the loss goes down; the first update equals tf.train.AdamOptimizer's
    formula applied to the reference gradients plus gamma * var."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from pwcnet_amd.train import Trainer
    N, H, W = 2, 64, 64
    w = util.model_weights(False, gain=1.1)
    im0, im1 = util.smooth_images(N, H, W, seed=63, shift=(2, 1))
    gt = np.zeros((N, H, W, 2), np.float32)
    gt[..., 0], gt[..., 1] = 2.0, 1.0                       # the true motion of smooth_images(shift=(2, 1))
    weights = (0.32, 0.08, 0.02, 0.01, 0.005)
    gamma, lr = 4e-4, 1e-3
    _, ref_g, _ = _ref_grads(w, im0, im1, gt, weights)
    tn = Trainer(weights=weights, gamma=gamma, lr=lr)
    tn.load_weights(w)
    g0, g1, ggt = gpu(im0), gpu(im1), gpu(gt)
    losses = [float(tn.step(g0, g1, ggt))]
    after = tn.state_dict()
    for k in ("pwcdcnet/context/conv2d_6/kernel", "pwcdcnet/optflow_2/conv2d/kernel", "pwcdcnet/fp_extractor/conv2d_4/bias"):
        g = ref_g[k] + gamma * t64(w[k], False)
        exp, _, _ = tr.adam_step(t64(w[k], False), g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        close(after[k], exp, rel=1e-4)
    for _ in range(7):
        losses.append(float(tn.step(g0, g1, ggt)))
    print("losses:", [round(x, 4) for x in losses])
    assert tn.global_step == 8 and losses[-1] < 0.9 * losses[0]


@pytest.mark.parametrize("use_dc", [False, True])
def test_trainer_forward_equals_inference_forward(use_dc):
    """The training forward (activations kept, weights from the flat buffer) and PWCDCNet.__call__ are the same
    function, with and without dense connections."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import pwcnet_amd
    from pwcnet_amd.train import Trainer, piecewise_lr
    w = util.model_weights(use_dc)
    im0, im1 = util.smooth_images(2, 64, 128, seed=64)
    tn = Trainer(use_dc=use_dc)
    tn.load_weights(w)
    pyr = tn.forward(gpu(im0), gpu(im1))
    net = pwcnet_amd.PWCDCNet(use_dc=use_dc)
    net.load_weights(w)
    _, ref = net(gpu(im0), gpu(im1))
    for a, b in zip(pyr, ref):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    with pytest.raises(ValueError, match="missing"):
        tn.load_weights({})
    # reference train.py:82-88
    assert piecewise_lr(1e-4, 0) == 1e-4 and piecewise_lr(1e-4, 200000) == 1e-4 and piecewise_lr(1e-4, 200001) == 5e-5
    assert piecewise_lr(1e-4, 360000) == 1e-4 / 16 and piecewise_lr(1e-4, 360000, scheduling=False) == 1e-4


@pytest.mark.parametrize("extra,use_dc", [((), False), (("--use-dc", "--loss", "robust"), True)])
def test_train_cli_synthetic_epoch_writes_a_restorable_bundle(tmp_path, extra, use_dc):
    """train.py (counterpart of the reference's train.py): two short epochs on synthetic translating textures;
    every epoch prints loss / validation EPE and writes a TF-format bundle that PWCDCNet can restore."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import os, subprocess, sys
    import pwcnet_amd
    from pwcnet_amd import ckpt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "train.py"), "-d", "synthetic", "--synthetic_pairs", "20",
                          "-e", "2", "-b", "4", "--crop_shape", "64", "128", "--lr", "3e-4", "--model_dir", str(tmp_path / "model"), *extra],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("epoch ")]
    assert len(lines) == 2 and "EPE/val" in lines[0] and "global_step 8" in lines[1], out.stdout[-1500:]
    w = ckpt.load_weights(str(tmp_path / "model" / "model_2.ckpt"))
    net = pwcnet_amd.PWCDCNet(use_dc=use_dc)
    net.load_weights(w)
    im0, im1 = util.smooth_images(1, 64, 128, seed=65)
    final, _ = net(gpu(im0), gpu(im1))
    assert torch.isfinite(final).all()
