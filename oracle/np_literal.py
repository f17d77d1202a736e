"""Literal numpy restatement of the reference's op *sequences* (second, independent oracle).

TEST INFRASTRUCTURE ONLY (never imported from pwcnet_amd/).  PARITY UNPINNED: see
pwc_oracle.c.  Where pwc_oracle.c uses closed forms, this file replays the TF ops
the reference issues one by one (tf.pad -> multiply -> Cropping2D -> reduce_mean;
meshgrid -> clip -> gather_nd; ...) so the two can be compared in tests/test_oracle.py.
Pure numpy, float64-capable, only meant for small shapes.
"""
import numpy as np


def tf_same_pads(size, stride, dilation):
    """TF 'SAME': (out, pad_before, pad_after) for a 3-tap kernel."""
    out = -(-size // stride)
    total = max((out - 1) * stride + 2 * dilation + 1 - size, 0)
    return out, total // 2, total - total // 2


def conv3x3_same(x, kernel, bias, stride=1, dilation=1, dtype=np.float64):
    """tf.layers.Conv2D(...,'same') as explicit zero-pad + 9 shifted einsums
    (reference modules.py:62-66,267,274,306-324)."""
    x = np.asarray(x, dtype)
    k = np.asarray(kernel, dtype)
    N, H, W, _ = x.shape
    Ho, pt, pb = tf_same_pads(H, stride, dilation)
    Wo, pl, pr = tf_same_pads(W, stride, dilation)
    xp = np.pad(x, [(0, 0), (pt, pb), (pl, pr), (0, 0)])
    y = np.zeros((N, Ho, Wo, k.shape[3]), dtype)
    for ty in range(3):
        for tx in range(3):
            ys, xs = ty * dilation, tx * dilation
            patch = xp[:, ys:ys + (Ho - 1) * stride + 1:stride, xs:xs + (Wo - 1) * stride + 1:stride]
            y += np.einsum("nhwi,io->nhwo", patch, k[ty, tx])
    return y + np.asarray(bias, dtype)


def leaky_relu(x, alpha):
    return np.maximum(alpha * x, x)


def cost_for_shift(f0, f1, v, h):
    """reference modules.py:164-181 (get_cost), op for op."""
    top, bot = max(v, 0), abs(min(v, 0))
    left, right = max(h, 0), abs(min(h, 0))
    a = np.pad(f0, [(0, 0), (top, bot), (left, right), (0, 0)])
    b = np.pad(f1, [(0, 0), (bot, top), (right, left), (0, 0)])
    prod = a * b
    Hp, Wp = prod.shape[1:3]
    cropped = prod[:, top:Hp - bot, left:Wp - right]
    return cropped.mean(axis=3)


def cost_volume(f0, f1, search_range=4, slope=0.1, dtype=np.float64):
    """reference modules.py:189-204: v outer, h inner, stack on axis 3, leaky 0.1."""
    f0, f1 = np.asarray(f0, dtype), np.asarray(f1, dtype)
    maps = []
    for v in range(-search_range, search_range + 1):
        for h in range(-search_range, search_range + 1):
            maps.append(cost_for_shift(f0, f1, v, h))
    return leaky_relu(np.stack(maps, axis=3), slope)


def bilinear_warp(x, flow, dtype=np.float32):
    """reference modules.py:99-137 with gather_nd as fancy indexing."""
    x = np.asarray(x, dtype)
    flow = np.asarray(flow, dtype)
    N, H, W, _ = x.shape
    gb, gy, gx = np.meshgrid(np.arange(N), np.arange(H), np.arange(W), indexing="ij")
    gyf, gxf = gy.astype(dtype), gx.astype(dtype)
    fx, fy = flow[..., 0], flow[..., 1]
    fx0 = np.floor(fx); fx1 = fx0 + 1
    fy0 = np.floor(fy); fy1 = fy0 + 1
    y0 = np.clip(gyf + fy0, 0, H - 1).astype(np.int32)
    y1 = np.clip(gyf + fy1, 0, H - 1).astype(np.int32)
    x0 = np.clip(gxf + fx0, 0, W - 1).astype(np.int32)
    x1 = np.clip(gxf + fx1, 0, W - 1).astype(np.int32)
    c00 = ((fy1 - fy) * (fx1 - fx))[..., None]
    c01 = ((fy1 - fy) * (fx - fx0))[..., None]
    c10 = ((fy - fy0) * (fx1 - fx))[..., None]
    c11 = ((fy - fy0) * (fx - fx0))[..., None]
    return c00 * x[gb, y0, x0] + c01 * x[gb, y0, x1] + c10 * x[gb, y1, x0] + c11 * x[gb, y1, x1]


def nearest_warp(x, flow):
    """reference modules.py:83-97 (int32 cast truncates toward zero)."""
    x = np.asarray(x)
    N, H, W, _ = x.shape
    gb, gy, gx = np.meshgrid(np.arange(N), np.arange(H), np.arange(W), indexing="ij")
    fi = np.trunc(np.asarray(flow, np.float32)).astype(np.int32)
    yy = np.clip(gy + fi[..., 1], 0, H - 1)
    xx = np.clip(gx + fi[..., 0], 0, W - 1)
    return x[gb, yy, xx]


def resize_bilinear_legacy(x, out_hw, dtype=np.float32):
    """tf.image.resize_bilinear (TF 1.8, align_corners=False); reference modules.py:283-284,
    model.py:127."""
    x = np.asarray(x, dtype)
    N, H, W, C = x.shape
    OH, OW = out_hw
    sy = np.float32(H) / np.float32(OH)
    sx = np.float32(W) / np.float32(OW)
    ys = (np.arange(OH, dtype=np.float32) * sy)
    xs = (np.arange(OW, dtype=np.float32) * sx)
    y0 = np.floor(ys).astype(np.int64); y1 = np.minimum(y0 + 1, H - 1); yl = (ys - y0).astype(dtype)
    x0 = np.floor(xs).astype(np.int64); x1 = np.minimum(x0 + 1, W - 1); xl = (xs - x0).astype(dtype)
    tl = x[:, y0][:, :, x0]; tr = x[:, y0][:, :, x1]
    bl = x[:, y1][:, :, x0]; br = x[:, y1][:, :, x1]
    xl_ = xl[None, None, :, None]; yl_ = yl[None, :, None, None]
    top = tl + (tr - tl) * xl_
    bot = bl + (br - bl) * xl_
    return top + (bot - top) * yl_
