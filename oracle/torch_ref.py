"""float64 torch-CPU restatement of the PWCDCNet forward, differentiable end to end.

TEST INFRASTRUCTURE ONLY (never imported from pwcnet_amd/).  PARITY UNPINNED like the rest of oracle/
(see pwc_oracle.c).  Purpose: reference GRADIENTS for the training path -- the
backward HIP kernels and the assembled train step are checked against torch.autograd on this graph,
and this graph's forward is checked against the C oracle in tests/test_oracle.py.

Every function restates the TF-1.8 op the reference calls, with the call site cited:
  conv3x3_same     tf.layers.Conv2D(...,'same')              modules.py:62-66,267,274,306-324
  cost_volume      CostVolumeLayer / get_cost                modules.py:158-204
  bilinear_warp    bilinear_warp / get_grid                  modules.py:75-81,99-137
  resize_legacy    tf.image.resize_bilinear (align_corners=False, no half-pixel)  modules.py:283-284
  losses           losses.py:4-32
"""
import math

import torch
import torch.nn.functional as F

DT = torch.float64


def _same_pads(size, stride, dilation):
    out = -(-size // stride)
    total = max((out - 1) * stride + 2 * dilation + 1 - size, 0)
    return out, total // 2, total - total // 2


def conv3x3_same(x, kernel, bias, stride=1, dilation=1):
    """x NHWC, kernel HWIO; TF 'SAME' (asymmetric for stride 2)."""
    _, H, W, _ = x.shape
    _, pt, pb = _same_pads(H, stride, dilation)
    _, pl, pr = _same_pads(W, stride, dilation)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xp, kernel.permute(3, 2, 0, 1), bias, stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


def leaky_relu(x, alpha=0.1):
    return torch.maximum(alpha * x, x)


def cost_volume(f0, f1, search_range=4, slope=0.1):
    N, H, W, C = f0.shape
    R = search_range
    f1p = F.pad(f1, (0, 0, R, R, R, R))
    maps = []
    for v in range(2 * R + 1):            # vertical outer, horizontal inner (modules.py:197-198)
        for h in range(2 * R + 1):
            maps.append((f0 * f1p[:, v:v + H, h:h + W]).mean(dim=3))
    return leaky_relu(torch.stack(maps, dim=3), slope)


def bilinear_warp(x, flow):
    N, H, W, C = x.shape
    gy, gx = torch.meshgrid(torch.arange(H, dtype=x.dtype), torch.arange(W, dtype=x.dtype), indexing="ij")
    fx, fy = flow[..., 0], flow[..., 1]
    fx0, fy0 = torch.floor(fx), torch.floor(fy)
    fx1, fy1 = fx0 + 1, fy0 + 1
    y0 = torch.clamp(gy + fy0, 0, H - 1).long()
    y1 = torch.clamp(gy + fy1, 0, H - 1).long()
    x0 = torch.clamp(gx + fx0, 0, W - 1).long()
    x1 = torch.clamp(gx + fx1, 0, W - 1).long()
    nb = torch.arange(N).view(N, 1, 1).expand(N, H, W)
    c00 = ((fy1 - fy) * (fx1 - fx)).unsqueeze(3)
    c01 = ((fy1 - fy) * (fx - fx0)).unsqueeze(3)
    c10 = ((fy - fy0) * (fx1 - fx)).unsqueeze(3)
    c11 = ((fy - fy0) * (fx - fx0)).unsqueeze(3)
    return c00 * x[nb, y0, x0] + c01 * x[nb, y0, x1] + c10 * x[nb, y1, x0] + c11 * x[nb, y1, x1]


def resize_legacy(x, out_hw):
    N, H, W, C = x.shape
    OH, OW = out_hw
    sy, sx = torch.tensor(H / OH, dtype=torch.float32), torch.tensor(W / OW, dtype=torch.float32)
    ys = (torch.arange(OH, dtype=torch.float32) * sy).to(x.dtype)
    xs = (torch.arange(OW, dtype=torch.float32) * sx).to(x.dtype)
    y0 = torch.floor(ys).long(); y1 = torch.clamp(y0 + 1, max=H - 1); yl = (ys - y0).view(1, OH, 1, 1)
    x0 = torch.floor(xs).long(); x1 = torch.clamp(x0 + 1, max=W - 1); xl = (xs - x0).view(1, 1, OW, 1)
    tl = x[:, y0][:, :, x0]; tr = x[:, y0][:, :, x1]
    bl = x[:, y1][:, :, x0]; br = x[:, y1][:, :, x1]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def resize_nearest(x, out_hw):
    """tf.image.resize_nearest_neighbor, align_corners=False (losses.py:27)."""
    N, H, W, C = x.shape
    OH, OW = out_hw
    sy, sx = torch.tensor(H / OH, dtype=torch.float32), torch.tensor(W / OW, dtype=torch.float32)
    iy = torch.clamp(torch.floor(torch.arange(OH, dtype=torch.float32) * sy).long(), max=H - 1)
    ix = torch.clamp(torch.floor(torch.arange(OW, dtype=torch.float32) * sx).long(), max=W - 1)
    return x[:, iy][:, :, ix]


def L2loss(x, y):
    return torch.linalg.vector_norm(x - y, ord=2, dim=3).sum(dim=(1, 2)).mean()


def multiscale_loss(flows_gt, flows_pyramid, weights):
    gt = flows_gt / 20.0
    loss = 0.0
    for w, fs in zip(weights, flows_pyramid):
        loss = loss + w * L2loss(resize_nearest(gt, fs.shape[1:3]), fs)
    return loss


def L1loss(x, y):
    return torch.linalg.vector_norm(x - y, ord=1, dim=3).sum(dim=(1, 2)).mean()


def multirobust_loss(flows_gt, flows_pyramid, weights, epsilon=0.01, q=0.4):
    """reference losses.py:34-48 as intended (its loop body names an undefined `loss_level`; `_l` is meant)."""
    gt = flows_gt / 20.0
    loss = 0.0
    for w, fs in zip(weights, flows_pyramid):
        loss = loss + w * (L1loss(resize_nearest(gt, fs.shape[1:3]), fs) + epsilon) ** q
    return loss


class TorchPWCDCNet:
    """reference model.py:74-134 on the functions above; weights: {name: tensor(requires_grad)}."""
    FILTERS_FP = [16, 32, 64, 96, 128, 192]
    FILTERS_OF = [128, 128, 96, 64, 32]
    CONTEXT = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1)]
    SCALES = [None, 0.625, 1.25, 2.5, 5.0, 10.0, 20.0]

    def __init__(self, weights, num_levels=6, search_range=4, use_dc=False, output_level=4, name="pwcdcnet"):
        self.w = weights
        self.num_levels, self.s_range, self.use_dc, self.output_level, self.name = num_levels, search_range, use_dc, output_level, name

    def _conv(self, scope, k, x, stride=1, dilation=1, act=True):
        n = f"{self.name}/{scope}/conv2d" + ("" if k == 0 else f"_{k}")
        y = conv3x3_same(x, self.w[n + "/kernel"], self.w[n + "/bias"], stride, dilation)
        return leaky_relu(y) if act else y

    def extractor(self, images):
        pyr, x, k = [], images, 0
        for l in range(self.num_levels):
            x = self._conv("fp_extractor", k, x, stride=2); k += 1
            x = self._conv("fp_extractor", k, x); k += 1
            x = self._conv("fp_extractor", k, x); k += 1
            pyr.append(x)
        return pyr[::-1]

    def estimator(self, l, cv, f0, flows_up, feats_up, is_output):
        scope = f"optflow_{l}"
        feats = cv
        for f in (f0, flows_up, feats_up):
            if f is not None:
                feats = torch.cat([feats, f], dim=3)
        for k in range(len(self.FILTERS_OF)):
            conv = self._conv(scope, k, feats)
            feats = torch.cat([conv, feats], dim=3) if self.use_dc else conv
        flows = self._conv(scope, 5, feats, act=False)
        if flows_up is not None:
            flows = flows + flows_up
        if is_output:
            return flows, feats
        h, w = flows.shape[1:3]
        return flows, resize_legacy(flows, (2 * h, 2 * w)), resize_legacy(feats, (2 * h, 2 * w))

    def context(self, flows, feats):
        x = torch.cat([flows, feats], dim=3)
        for k, (_, d) in enumerate(self.CONTEXT):
            x = self._conv("context", k, x, dilation=d)
        return flows + self._conv("context", 6, x, act=False)

    def __call__(self, images_0, images_1):
        pyr0, pyr1 = self.extractor(images_0), self.extractor(images_1)
        flows_pyramid, flows_up, feats_up = [], None, None
        for l, (f0, f1) in enumerate(zip(pyr0, pyr1)):
            f1w = f1 if l == 0 else bilinear_warp(f1, flows_up * self.SCALES[l])
            cv = cost_volume(f0, f1w, self.s_range)
            if l < self.output_level:
                flows, flows_up, feats_up = self.estimator(l, cv, f0, flows_up, feats_up, False)
            else:
                flows, feats = self.estimator(l, cv, f0, flows_up, feats_up, True)
                flows = self.context(flows, feats)
                flows_pyramid.append(flows)
                up = 2 ** (self.num_levels - self.output_level)
                h, w = flows.shape[1:3]
                return resize_legacy(flows, (h * up, w * up)) * 20.0, flows_pyramid
            flows_pyramid.append(flows)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (TF 1.8 adam.py): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);
    m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;  p <- p - lr_t m / (sqrt(v) + eps)."""
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    return p - lr_t * m / (torch.sqrt(v) + eps), m, v
