"""CPU oracle for the PWC-Net forward (ctypes front-end of pwc_oracle.c + assembly).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from pwcnet_amd/ (the product path must fail
loudly without its HIP library instead of falling back to this).

PARITY UNPINNED (see pwc_oracle.c header): the reference cannot run here (TF 1.8
absent) and ships no tests.  This file restates reference model.py:74-134
(PWCDCNet.__call__) and modules.py:42-71,227-326 on top of the C primitives.

All tensors are numpy float32 NHWC; weights are a dict
    'pwcdcnet/<scope>/conv2d[_k]/kernel' -> (3,3,Cin,Cout) HWIO
    'pwcdcnet/<scope>/conv2d[_k]/bias'   -> (Cout,)
exactly the variable names in the reference's checkpoints.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile pwc_oracle.c with gcc (oracle/Makefile).  Idempotent."""
    so = os.path.join(_HERE, "libpwc_oracle.so")
    src = os.path.join(_HERE, "pwc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpwc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        try:
            _LIB = ctypes.CDLL(build())
        except OSError:
            _LIB = ctypes.CDLL(build(force=True))
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(_f32p)


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def same_out(size, stride):
    return -(-size // stride)


def leaky_relu(x, slope=0.1):
    return np.maximum(x, np.float32(slope) * x)


def conv3x3(x, kernel, bias, stride=1, dilation=1, slope=None, residual=None, cin_slice=None):
    """tf.layers.Conv2D(Cout,(3,3),(s,s),'same',dilation_rate=d) [+ leaky_relu(slope)]
    [+ residual].  cin_slice=(lo,hi) convolves x[..., lo:hi] without copying."""
    x = _c(x)
    N, H, W, CS = x.shape
    lo, hi = (0, CS) if cin_slice is None else cin_slice
    kernel = _c(kernel)
    assert kernel.shape[:3] == (3, 3, hi - lo), (kernel.shape, lo, hi)
    Cout = kernel.shape[3]
    bias = _c(bias)
    Ho, Wo = same_out(H, stride), same_out(W, stride)
    y = np.empty((N, Ho, Wo, Cout), np.float32)
    xs = x.reshape(-1)[lo:]
    res = None
    res_cs = 0
    if residual is not None:
        res = _c(residual)
        assert res.shape[:3] == (N, Ho, Wo)
        res_cs = res.shape[3]
    rc = lib().oracle_conv3x3(
        _p(xs), N, H, W, hi - lo, CS, _p(kernel), _p(bias), Cout, int(stride), int(dilation),
        0 if slope is None else 1, ctypes.c_float(0.0 if slope is None else slope),
        _p(res) if res is not None else None, res_cs, _p(y))
    assert rc == 0
    return y


def cost_volume(f0, f1, search_range=4, slope=0.1):
    f0, f1 = _c(f0), _c(f1)
    N, H, W, C = f0.shape
    D = 2 * search_range + 1
    cv = np.empty((N, H, W, D * D), np.float32)
    lib().oracle_cost_volume(_p(f0), _p(f1), N, H, W, C, int(search_range),
                             ctypes.c_float(slope), _p(cv))
    return cv


def warp(x, flow, warp_type="bilinear", flow_scale=1.0):
    x, flow = _c(x), _c(flow)
    N, H, W, C = x.shape
    assert flow.shape == (N, H, W, 2)
    out = np.empty_like(x)
    fn = lib().oracle_warp_bilinear if warp_type == "bilinear" else lib().oracle_warp_nearest
    fn(_p(x), _p(flow), 2, ctypes.c_float(flow_scale), N, H, W, C, _p(out))
    return out


def resize_bilinear(x, out_hw, mul=1.0):
    x = _c(x)
    N, H, W, C = x.shape
    OH, OW = out_hw
    y = np.empty((N, OH, OW, C), np.float32)
    lib().oracle_resize_bilinear(_p(x), C, N, H, W, C, int(OH), int(OW), ctypes.c_float(mul), _p(y))
    return y


# --------------------------------------------------------------------------- assembly

def _vname(scope, k):
    return f"{scope}/conv2d" + ("" if k == 0 else f"_{k}")


class OraclePWCDCNet:
    """Restates reference model.py:74-134 + modules.py module bodies, eagerly."""

    FILTERS_FP = [16, 32, 64, 96, 128, 192]          # modules.py:46
    FILTERS_OF = [128, 128, 96, 64, 32]              # modules.py:235
    CONTEXT = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1)]  # modules.py:306-323
    SCALES = [None, 0.625, 1.25, 2.5, 5.0, 10.0, 20.0]   # model.py:93

    def __init__(self, weights, num_levels=6, search_range=4, warp_type="bilinear",
                 use_dc=False, output_level=4, name="pwcdcnet"):
        assert output_level < num_levels
        self.w = weights
        self.num_levels, self.s_range, self.warp_type = num_levels, search_range, warp_type
        self.use_dc, self.output_level, self.name = use_dc, output_level, name

    def _conv(self, scope, k, x, stride=1, dilation=1, slope=0.1, residual=None):
        n = _vname(f"{self.name}/{scope}", k)
        return conv3x3(x, self.w[n + "/kernel"], self.w[n + "/bias"], stride, dilation, slope, residual)

    def extractor(self, images):                       # modules.py:49-71
        pyr, x, k = [], images, 0
        for l in range(self.num_levels):
            x = self._conv("fp_extractor", k, x, stride=2); k += 1
            x = self._conv("fp_extractor", k, x); k += 1
            x = self._conv("fp_extractor", k, x); k += 1
            pyr.append(x)
        return pyr[::-1]

    def estimator(self, l, cv, f0, flows_up, feats_up, is_output):   # modules.py:239-285
        scope = f"optflow_{l}"
        feats = cv
        for f in (f0, flows_up, feats_up):
            if f is not None:
                feats = np.concatenate([feats, f], axis=3)
        for k in range(len(self.FILTERS_OF)):
            conv = self._conv(scope, k, feats)
            feats = np.concatenate([conv, feats], axis=3) if self.use_dc else conv
        flows = self._conv(scope, 5, feats, slope=None, residual=flows_up)
        if is_output:
            return flows, feats
        h, w = flows.shape[1:3]
        return flows, resize_bilinear(flows, (2 * h, 2 * w)), resize_bilinear(feats, (2 * h, 2 * w))

    def context(self, flows, feats):                   # modules.py:304-326
        x = np.concatenate([flows, feats], axis=3)
        for k, (_, d) in enumerate(self.CONTEXT):
            x = self._conv("context", k, x, dilation=d)
        return self._conv("context", 6, x, slope=None, residual=flows)

    def __call__(self, images_0, images_1, with_features=False):     # model.py:95-134
        pyr0 = self.extractor(images_0)
        pyr1 = self.extractor(images_1)
        flows_pyramid, flows_up, feats_up = [], None, None
        for l, (f0, f1) in enumerate(zip(pyr0, pyr1)):
            if l == 0:
                f1w = f1
            else:
                f1w = warp(f1, flows_up, self.warp_type, flow_scale=self.SCALES[l])
            cv = cost_volume(f0, f1w, self.s_range)
            if l < self.output_level:
                flows, flows_up, feats_up = self.estimator(l, cv, f0, flows_up, feats_up, False)
            else:
                flows, feats = self.estimator(l, cv, f0, flows_up, feats_up, True)
                flows = self.context(flows, feats)
                flows_pyramid.append(flows)
                up = 2 ** (self.num_levels - self.output_level)
                h, w = flows.shape[1:3]
                flows_final = resize_bilinear(flows, (h * up, w * up), mul=20.0)
                if with_features:
                    return flows_final, flows_pyramid, pyr0
                return flows_final, flows_pyramid
            flows_pyramid.append(flows)


def epe(flows_gt, flows):
    """losses.py:11-13: mean over all pixels of the L2 norm of the flow difference."""
    d = np.asarray(flows_gt, np.float64) - np.asarray(flows, np.float64)
    return float(np.mean(np.sqrt(np.sum(d * d, axis=3))))


# ---------------------------------------------------------------- losses (reference losses.py)
def resize_nearest(x, out_hw):
    """tf.image.resize_nearest_neighbor, TF 1.8, align_corners=False (losses.py:27,43):
    src = min(floor(dst * in/out), in - 1), scale computed in float32."""
    n, h, w, c = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    sy, sx = np.float32(h) / np.float32(oh), np.float32(w) / np.float32(ow)
    iy = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * sy).astype(np.int64), h - 1)
    ix = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * sx).astype(np.int64), w - 1)
    return x[:, iy][:, :, ix]


def L1loss(x, y):
    """losses.py:4-5: reduce_mean over the batch of reduce_sum over (h, w) of the L1 norm over channels."""
    return float(np.abs(x.astype(np.float64) - y).sum(axis=3).sum(axis=(1, 2)).mean())


def L2loss(x, y):
    """losses.py:7-8."""
    return float(np.sqrt(((x.astype(np.float64) - y) ** 2).sum(axis=3)).sum(axis=(1, 2)).mean())


def multiscale_loss(flows_gt, flows_pyramid, weights):
    """losses.py:15-32."""
    gt = flows_gt.astype(np.float32) / np.float32(20.0)
    return sum(float(wt) * L2loss(resize_nearest(gt, fs.shape[1:3]), fs) for wt, fs in zip(weights, flows_pyramid))


def multirobust_loss(flows_gt, flows_pyramid, weights, epsilon=0.01, q=0.4):
    """losses.py:34-48 as evidently intended (the reference body uses an undefined name `loss_level`
    where it means the level's L1 loss)."""
    gt = flows_gt.astype(np.float32) / np.float32(20.0)
    return sum(float(wt) * (L1loss(resize_nearest(gt, fs.shape[1:3]), fs) + epsilon) ** q
               for wt, fs in zip(weights, flows_pyramid))
