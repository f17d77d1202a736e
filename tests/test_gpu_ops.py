"""GPU parity tests, op level: every HIP kernel (called through the C ABI via the
pwcnet_amd host classes / ctypes) against the CPU oracle on the same seeded inputs.

Tolerances (floating point, fp32 everywhere): the kernels sum in a different order than
the oracle, so results are compared with atol scaled to the magnitude of the sums:
1e-5 relative to max|y| for convolutions/correlations, 1e-6 for gathers/lerps.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (torch.cuda.is_available() is False)")
    import pwcnet_amd
    from pwcnet_amd import _lib
    _lib.lib()
    return pwcnet_amd


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.RandomState(seed).uniform(lo, hi, size=shape).astype(np.float32)


def gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def close(got, exp, rel=1e-5, floor=1e-6):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    tol = max(floor, rel * float(np.abs(exp).max()))
    err = float(np.abs(got - exp).max())
    assert got.shape == exp.shape
    assert err <= tol, f"max abs err {err:.3e} > tol {tol:.3e}"


# ------------------------------------------------------------------ library surface
def test_library_loaded_and_version(pa):
    from pwcnet_amd import _lib
    L = _lib.lib()
    assert L.pwc_version() >= 100
    assert b"align" in L.pwc_error_string(-2)
    # the .so is in-tree (the driver records which native libraries the tests load)
    assert os.path.dirname(_lib.LIB_PATH).endswith(os.path.join("pwcnet_amd", "csrc"))


def test_argument_errors_are_reported_not_crashes(pa):
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = torch.zeros(1, 4, 4, 6, device="cuda")   # C = 6 is not a multiple of 4
    o = torch.zeros(1, 4, 4, 81, device="cuda")
    rc = L.pwc_cost_volume_f32(_p(x), 6, _p(x), 6, _p(o), 81, 1, 4, 4, 6, 4, 0.1, None)
    assert rc == -2
    rc = L.pwc_cost_volume_f32(None, 8, _p(x), 8, _p(o), 81, 1, 4, 4, 8, 4, 0.1, None)
    assert rc == -1
    with pytest.raises(_lib.PwcHipError):
        _lib.check(rc, "cost_volume")
    with pytest.raises(_lib.PwcHipError):
        pa.CostVolumeLayer()(torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4, 8))   # CPU tensors


# ------------------------------------------------------------------ conv (MFMA implicit GEMM)
def run_conv_mfma(x, k, b, stride, dil, slope, tile=-1, cin_map=None, cin_phys=None, y_cs=None, split=0):
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cs = x.shape
    cin, cout = k.shape[2], k.shape[3]
    cin_phys = cs if cin_phys is None else cin_phys
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_packed_floats(cin_phys, cout), device="cuda")
    cm = None if cin_map is None else torch.from_numpy(np.asarray(cin_map, np.int32)).cuda()
    _lib.check(L.pwc_conv3x3_pack_f32(_p(kg), _p(cm) if cm is not None else None, cin, cin_phys, cout,
                                      _p(packed), None))
    Ho, Wo = -(-H // stride), -(-W // stride)
    y_cs = cout if y_cs is None else y_cs
    y = torch.full((N, Ho, Wo, y_cs), -7.0, device="cuda")
    ws = torch.empty(L.pwc_conv3x3_workspace_floats(N * Ho * Wo, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_f32(_p(xg), cs, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cin_phys, cout,
                                 stride, dil, 0 if slope is None else 1, 0.0 if slope is None else slope,
                                 tile, split, _p(ws), ws.numel(), None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout,stride,dil", [
    (2, 12, 20, 16, 16, 1, 1), (2, 12, 20, 16, 32, 2, 1), (1, 24, 40, 32, 64, 1, 1),
    (1, 14, 22, 96, 96, 2, 1), (1, 28, 64, 128, 128, 1, 1), (3, 7, 16, 192, 192, 1, 1),
    (1, 40, 48, 48, 128, 1, 1), (1, 40, 48, 128, 128, 1, 2), (1, 40, 48, 128, 96, 1, 8),
    (1, 40, 48, 96, 64, 1, 16), (1, 17, 33, 64, 32, 1, 1), (1, 9, 5, 32, 16, 2, 1),
    (2, 56, 128, 160, 128, 1, 1)])
def test_conv_mfma_vs_oracle(pa, N, H, W, cin, cout, stride, dil):
    x = rnd((N, H, W, cin), 1)
    k = rnd((3, 3, cin, cout), 2) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 3) * 0.1
    y = run_conv_mfma(x, k, b, stride, dil, 0.1)
    close(y, orc.conv3x3(x, k, b, stride, dil, 0.1))


@pytest.mark.parametrize("N,H,W,c", [(1, 260, 300, 16), (2, 132, 260, 32), (1, 256, 256, 16), (3, 100, 224, 32)])
def test_conv_halo_kernel_vs_oracle(pa, N, H, W, c):
    """full-resolution 16->16 / 32->32 layers take the resident-weights halo-patch kernel
    (M >= 65536); ragged sizes exercise the zero-page halo and the partial tiles."""
    from pwcnet_amd import _lib
    assert _lib.lib().pwc_conv3x3_uses_halo_kernel(N * H * W, c, c, 1, 1) == 1
    x = rnd((N, H, W, c), 61)
    k = rnd((3, 3, c, c), 62) * float(1.0 / np.sqrt(9 * c))
    b = rnd((c,), 63) * 0.1
    exp = orc.conv3x3(x, k, b, 1, 1, 0.1)
    close(run_conv_mfma(x, k, b, 1, 1, 0.1), exp)
    close(run_conv_mfma(x, k, b, 1, 1, 0.1, tile=14 if c == 16 else 13), exp)      # generic kernel agrees
    y = run_conv_mfma(x, k, b, 1, 1, None, y_cs=c + 4)
    close(y[..., :c], orc.conv3x3(x, k, b, 1, 1, None))
    assert float(y[..., c:].min()) == -7.0


def run_conv_wino(x, k, b, slope, cin_map=None, y_cs=None, dil=1):
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cs = x.shape
    cin, cout = k.shape[2], k.shape[3]
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_wino_packed_floats(cs, cout), device="cuda")
    cm = None if cin_map is None else torch.from_numpy(np.asarray(cin_map, np.int32)).cuda()
    _lib.check(L.pwc_conv3x3_wino_pack_f32(_p(kg), _p(cm) if cm is not None else None, cin, cs, cout, _p(packed), None))
    y_cs = cout if y_cs is None else y_cs
    y = torch.full((N, H, W, y_cs), -7.0, device="cuda")
    _lib.check(L.pwc_conv3x3_wino_f32(_p(xg), cs, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cs, cout, dil,
                                      0 if slope is None else 1, 0.0 if slope is None else slope, None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout", [
    (2, 16, 16, 16, 32), (1, 32, 48, 32, 64), (2, 7, 16, 192, 192), (1, 14, 32, 128, 96), (1, 28, 64, 96, 64),
    (1, 33, 21, 64, 32), (1, 5, 3, 16, 32), (2, 56, 128, 160, 128), (2, 40, 70, 16, 16), (1, 20, 30, 32, 48),
    # 4x64-pixel block geometry (height a multiple of 4, not of 16), incl. ragged right edges
    (1, 28, 64, 96, 64), (2, 56, 128, 32, 16), (1, 20, 130, 16, 32), (1, 36, 200, 48, 48)])
def test_conv_winograd_vs_oracle(pa, N, H, W, cin, cout):
    x = rnd((N, H, W, cin), 71)
    k = rnd((3, 3, cin, cout), 72) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 73) * 0.1
    close(run_conv_wino(x, k, b, 0.1), orc.conv3x3(x, k, b, 1, 1, 0.1), rel=2e-5)
    y = run_conv_wino(x, k, b, None, y_cs=cout + 8)
    close(y[..., :cout], orc.conv3x3(x, k, b, 1, 1, None), rel=2e-5)
    assert float(y[..., cout:].min()) == -7.0


@pytest.mark.parametrize("N,H,W,cin,cout,dil", [
    (1, 40, 48, 128, 128, 2), (1, 40, 48, 128, 96, 4), (1, 56, 64, 64, 32, 8), (2, 19, 23, 32, 64, 3),
    (1, 33, 47, 16, 32, 16),
    # sub-lattices at most 8 pixels high: two of them share a workgroup (SPLIT kernels)
    (1, 112, 256, 96, 64, 16), (2, 14, 40, 32, 16, 2), (1, 16, 16, 16, 32, 2), (1, 15, 9, 48, 48, 4),
    # dilated + 4x64-pixel blocks (sub-lattices 56x128, 28x64)
    (1, 112, 256, 32, 32, 2), (1, 112, 256, 16, 32, 4)])
def test_conv_winograd_dilated_vs_oracle(pa, N, H, W, cin, cout, dil):
    x = rnd((N, H, W, cin), 77)
    k = rnd((3, 3, cin, cout), 78) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 79) * 0.1
    close(run_conv_wino(x, k, b, 0.1, dil=dil), orc.conv3x3(x, k, b, 1, dil, 0.1), rel=2e-5)


def run_conv_wino4(x, k, b, slope, cin_map=None, y_cs=None, dil=1):
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cs = x.shape
    cin, cout = k.shape[2], k.shape[3]
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_wino4_packed_floats(cs, cout), device="cuda")
    cm = None if cin_map is None else torch.from_numpy(np.asarray(cin_map, np.int32)).cuda()
    _lib.check(L.pwc_conv3x3_wino4_pack_f32(_p(kg), _p(cm) if cm is not None else None, cin, cs, cout, _p(packed), None))
    y_cs = cout if y_cs is None else y_cs
    y = torch.full((N, H, W, y_cs), -7.0, device="cuda")
    _lib.check(L.pwc_conv3x3_wino4_f32(_p(xg), cs, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cs, cout, dil,
                                       0 if slope is None else 1, 0.0 if slope is None else slope, None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout,dil", [
    (1, 32, 64, 128, 128, 1), (2, 16, 32, 64, 32, 1), (1, 33, 47, 32, 64, 1), (1, 5, 3, 16, 32, 1), (2, 50, 70, 160, 96, 1),
    (1, 56, 64, 128, 96, 2), (1, 40, 48, 64, 64, 4), (1, 112, 96, 32, 32, 8), (2, 19, 23, 48, 64, 3), (1, 20, 40, 32, 16, 1),
    (2, 30, 33, 80, 48, 1)])
def test_conv_winograd_f4x4_vs_oracle(pa, N, H, W, cin, cout, dil):
    """pwc_conv3x3_wino4_f32 (Winograd F(4x4,3x3)): ragged 16 x 32-pixel blocks, dilations (sub-lattices), short and long
    channel loops, strided output with untouched neighbours.  Tolerance: F(4x4) rounds ~10x coarser than F(2x2)
    (transform constants up to 8 and down to 1/24) -- 2e-6 of the activation scale measured, 2e-5 allowed."""
    x = rnd((N, H, W, cin), 171)
    k = rnd((3, 3, cin, cout), 172) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 173) * 0.1
    close(run_conv_wino4(x, k, b, 0.1, dil=dil), orc.conv3x3(x, k, b, 1, dil, 0.1), rel=2e-5)
    y = run_conv_wino4(x, k, b, None, y_cs=cout + 8, dil=dil)
    close(y[..., :cout], orc.conv3x3(x, k, b, 1, dil, None), rel=2e-5)
    assert float(y[..., cout:].min()) == -7.0 and float(y[..., cout:].max()) == -7.0


@pytest.mark.parametrize("cin,cout,dil", [(160, 128, 1), (128, 128, 4)])
def test_conv_winograd_f4x4_full_size_vs_oracle(pa, cin, cout, dil):
    """pwc_conv3x3_wino4_f32 at the PRODUCTION shape (8 x 112 x 256: 7168 - 8192 workgroups, XCD remap,
    two workgroups per CU) against the oracle's convolution on two sampled images of the batch (first and last), every
    entry.  The other six images are checked for finiteness and for the untouched channels behind Cout."""
    N, H, W = 8, 112, 256
    x = rnd((N, H, W, cin), 181)
    k = rnd((3, 3, cin, cout), 182) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 183) * 0.1
    y = run_conv_wino4(x, k, b, 0.1, y_cs=cout + 16, dil=dil)
    for i in (0, N - 1):
        close(y[i:i + 1, ..., :cout], orc.conv3x3(x[i:i + 1], k, b, 1, dil, 0.1), rel=2e-5)
    assert bool(torch.isfinite(y).all())
    assert float(y[..., cout:].min()) == -7.0 and float(y[..., cout:].max()) == -7.0


def test_conv_winograd_f4x4_physical_layout_and_plan(pa):
    """Padded / permuted physical input channels through cin_map (the estimator buffers), and the shapes the model routes
    to F(4x4): the big level-4 layers of a batch, not single pairs or the small levels."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(5)
    N, H, W, cin, cout, cs = 1, 32, 64, 147, 128, 160
    cmap = np.full((cs,), -1, np.int32)
    pos = np.sort(rs.choice(cs, cin, replace=False))
    cmap[pos] = np.arange(cin, dtype=np.int32)
    xl = rnd((N, H, W, cin), 181)
    xp = rnd((N, H, W, cs), 182)                       # padding channels hold garbage-free finite values
    xp[..., pos] = xl
    k = rnd((3, 3, cin, cout), 183) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 184) * 0.1
    close(run_conv_wino4(xp, k, b, 0.1, cin_map=cmap), orc.conv3x3(xl, k, b, 1, 1, 0.1), rel=2e-5)
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 160, 128, 1) == 1
    assert L.pwc_conv3x3_wino4_supported(8, 56, 128, 192, 128, 1) == 1
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 288, 128, 1) == 0     # the dense-connection inputs stay on F(2x2)
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 128, 96, 8) == 1
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 128, 96, 1) == 1      # 16-cout workgroups: 96 and 64 couts pay too
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 128, 128, 4) == 1     # ... and the d = 2, 4 layers (x1.05)
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 96, 64, 1) == 1
    assert L.pwc_conv3x3_wino4_supported(4, 56, 128, 128, 96, 1) == 1       # a side-stream sub-batch at level 3
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 32, 32, 1) == 0       # short channel loops stay on F(2x2)
    assert L.pwc_conv3x3_wino4_supported(8, 112, 256, 128, 64, 16) == 0     # 7-row sub-lattices
    assert L.pwc_conv3x3_wino4_supported(1, 56, 128, 128, 128, 1) == 0      # 128 workgroups do not fill the GPU
    assert L.pwc_conv3x3_wino4_supported(8, 14, 32, 128, 128, 1) == 0       # ... nor does a coarse level


H2_COUTS = {1: 128, 2: 64, 3: 96, 4: 32, 5: 64}


def run_conv_h2(x, k, b, slope, cin_map=None, y_cs=None, dil=1, variant=0, ws=None):
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cs = x.shape
    cin, cout = k.shape[2], k.shape[3]
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_h2_packed_floats(cs, cout), device="cuda")
    cm = None if cin_map is None else torch.from_numpy(np.asarray(cin_map, np.int32)).cuda()
    _lib.check(L.pwc_conv3x3_h2_pack_f32(_p(kg), _p(cm) if cm is not None else None, cin, cs, cout, _p(packed), None))
    y_cs = cout if y_cs is None else y_cs
    y = torch.full((N, H, W, y_cs), -7.0, device="cuda")
    act, sl = (0 if slope is None else 1), (0.0 if slope is None else slope)
    wp, wn = (None, 0) if ws is None else (_p(ws), ws.numel())
    if variant:
        _lib.check(L.pwc_conv3x3_h2_variant_f32(_p(xg), cs, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cs, cout, dil, act, sl,
                                                variant, wp, wn, None))
    else:
        _lib.check(L.pwc_conv3x3_h2_f32(_p(xg), cs, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cs, cout, dil, act, sl, wp, wn, None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout,dil", [
    (1, 32, 64, 128, 128, 1), (2, 16, 32, 64, 32, 1), (1, 33, 47, 32, 64, 1), (1, 5, 3, 16, 32, 1), (2, 50, 70, 160, 96, 1),
    (1, 56, 64, 128, 96, 2), (1, 40, 48, 64, 64, 4), (1, 112, 96, 32, 32, 8), (2, 19, 23, 48, 64, 3), (1, 20, 40, 16, 128, 1),
    (2, 30, 33, 80, 192, 1), (1, 17, 100, 48, 256, 1),
    # sub-lattices of fewer than 24 columns: two of them share a tile (taps two lattice columns apart, 36-pixel patches)
    (1, 112, 256, 96, 64, 16), (1, 50, 180, 32, 128, 8), (2, 40, 100, 48, 96, 6)])
def test_conv_f16x2_direct_vs_oracle(pa, N, H, W, cin, cout, dil):
    """pwc_conv3x3_h2_f32 (direct convolution on the F16 matrix pipe, fp32 operands as two-term fp16 splits) in EVERY tile
    variant whose couts divide Cout: ragged 8/16-row x 32-column tiles, dilations (sub-lattices), one to ten channel stages
    (the fetch pipeline's prologue and its tail), strided output with untouched neighbours.  Tolerance: the fp32 default
    (1e-5 of max |y|) -- the split arithmetic is MORE accurate than an fp32 MFMA chain, it gets no allowance."""
    x = rnd((N, H, W, cin), 271)
    k = rnd((3, 3, cin, cout), 272) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 273) * 0.1
    exp = orc.conv3x3(x, k, b, 1, dil, 0.1)
    exp_lin = orc.conv3x3(x, k, b, 1, dil, None)
    ran = 0
    paired = dil % 2 == 0 and -(-W // dil) < 24 <= -(-W // (dil // 2)) and cout % 64 == 0
    for v in (0, 1, 2, 3, 4, 5):
        if v and (cout % H2_COUTS[v] or (paired and v in (2, 4))):      # (paired sub-lattices: the 8-row tiles only)
            continue
        close(run_conv_h2(x, k, b, 0.1, dil=dil, variant=v), exp)
        y = run_conv_h2(x, k, b, None, y_cs=cout + 8, dil=dil, variant=v)
        close(y[..., :cout], exp_lin)
        assert float(y[..., cout:].min()) == -7.0 and float(y[..., cout:].max()) == -7.0
        ran += 1
    assert ran >= 2


@pytest.mark.parametrize("cin,cout,dil", [(160, 128, 1), (128, 96, 2), (64, 32, 1), (96, 64, 16)])
def test_conv_f16x2_direct_full_size_vs_oracle(pa, cin, cout, dil):
    """pwc_conv3x3_h2_f32 at the PRODUCTION shape (8 x 112 x 256: 896 - 1792 workgroups of 512 threads, XCD remap) against
    the oracle's convolution on the first and the last image of the batch, every entry; the other six are checked for
    finiteness and for the untouched channels behind Cout."""
    N, H, W = 8, 112, 256
    x = rnd((N, H, W, cin), 281)
    k = rnd((3, 3, cin, cout), 282) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 283) * 0.1
    y = run_conv_h2(x, k, b, 0.1, y_cs=cout + 16, dil=dil, ws=h2_workspace(N, H, W, cin, cout, dil))
    for i in (0, N - 1):
        close(y[i:i + 1, ..., :cout], orc.conv3x3(x[i:i + 1], k, b, 1, dil, 0.1))
    assert bool(torch.isfinite(y).all())
    assert float(y[..., cout:].min()) == -7.0 and float(y[..., cout:].max()) == -7.0


def h2_workspace(N, H, W, cs, cout, dil):
    from pwcnet_amd import _lib
    n = _lib.lib().pwc_conv3x3_h2_workspace_floats(N, H, W, cs, cout, dil)
    return None if n == 0 else torch.full((n,), -1, dtype=torch.int32, device="cuda").view(torch.float32)


@pytest.mark.parametrize("N,H,W,cin,cout,dil", [(8, 112, 256, 128, 128, 1), (8, 112, 256, 64, 32, 1), (3, 112, 250, 96, 96, 2),
                                               (5, 100, 256, 48, 64, 1), (2, 224, 512, 32, 128, 4),
                                               # fewer tiles than CUs: every tile is cut into several pieces
                                               (8, 28, 64, 192, 128, 1), (8, 28, 64, 256, 64, 1)])
def test_conv_f16x2_direct_stream_k(pa, N, H, W, cin, cout, dil):
    """More tiles than CUs (or far fewer) + a workspace: one workgroup per CU, each with an equal share of the (tile, stage) sequence; tiles
    cut in two are finished through the workspace.  Ten launches must agree BITWISE (the sum of two finished pieces does
    not depend on which workgroup is faster), match the one-workgroup-per-tile launch to fp32 rounding and the oracle on
    the first and the last image, and leave the workspace as they found it (every word 0xFFFFFFFF)."""
    x = rnd((N, H, W, cin), 291)
    k = rnd((3, 3, cin, cout), 292) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 293) * 0.1
    ws = h2_workspace(N, H, W, cin, cout, dil)
    assert ws is not None
    plain = run_conv_h2(x, k, b, 0.1, dil=dil)
    first = run_conv_h2(x, k, b, 0.1, dil=dil, ws=ws)
    for _ in range(9):
        again = run_conv_h2(x, k, b, 0.1, dil=dil, ws=ws)
        assert torch.equal(first, again)
    assert bool((ws.view(torch.int32) == -1).all())
    scale = float(plain.abs().max())
    assert float((first - plain).abs().max()) <= 2e-6 * scale
    for i in (0, N - 1):
        close(first[i:i + 1], orc.conv3x3(x[i:i + 1], k, b, 1, dil, 0.1))


@pytest.mark.parametrize("N,H,W,cin,cout,stride,dil", [
    (8, 7, 16, 288, 128, 1, 1), (8, 7, 16, 128, 128, 1, 1), (8, 7, 16, 128, 96, 1, 1), (8, 7, 16, 96, 64, 1, 1),
    (8, 7, 16, 64, 32, 1, 1), (8, 14, 32, 256, 128, 1, 1), (16, 14, 32, 128, 192, 2, 1), (16, 7, 16, 192, 192, 1, 1),
    (1, 5, 3, 32, 16, 1, 1), (2, 9, 21, 64, 48, 1, 2), (1, 13, 30, 96, 32, 2, 1), (2, 11, 7, 32, 32, 1, 4),
    (1, 1, 1, 64, 16, 1, 1), (1, 56, 128, 128, 96, 1, 1), (3, 8, 8, 352, 32, 1, 1)])
@pytest.mark.parametrize("tile", [0, 11, 21, 22, 31, 41, 42])
def test_conv_small_launch_kernel_vs_oracle(pa, N, H, W, cin, cout, stride, dil, tile):
    """pwc_conv3x3_sk_f32 (round 5, conv3x3_sk.hip: the K dimension of a tile over the eight waves of one workgroup): the
    estimator / extractor layers of BASELINE configs[1]'s two coarsest levels as they are, stride 2 (even and odd sizes),
    dilations, ragged blocks, an image smaller than a block, more K steps than one batch of requests holds (352 channels), every
    workgroup tile; strided output with untouched neighbours; padded / permuted input channels (cin_map); launches repeat
    bitwise."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    if tile % 10 == 2 and cout % 32:
        pytest.skip("the 2 x 2 tile needs C_out % 32 == 0")
    if tile > 30 and (dil != 1 or cin > (288 if stride == 1 else 128)):
        pytest.skip("the LDS-patch form takes no dilation, up to 288 (stride 2: 128) input channels")
    x = rnd((N, H, W, cin), 471)
    k = rnd((3, 3, cin, cout), 472) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 473) * 0.1
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_sk_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_sk_pack_f32(_p(kg), None, cin, cin, cout, _p(packed), None))
    exp = orc.conv3x3(x, k, b, stride, dil, 0.1)
    Ho, Wo = exp.shape[1:3]
    def sk(*a, t=tile):
        """pwc_conv3x3_sk_f32 (tile 0: the library's choice) or the same launch with the workgroup tile given -- an ARGUMENT of
        pwc_conv3x3_sk_variant_f32, not a process-wide knob."""
        if t == 0:
            return L.pwc_conv3x3_sk_f32(*a)
        return L.pwc_conv3x3_sk_variant_f32(*a[:-1], t, a[-1])
    ys = []
    for _ in range(2):
        y = torch.full((N, Ho, Wo, cout + 8), -7.0, device="cuda")
        _lib.check(sk(_p(xg), cin, _p(packed), _p(bg), _p(y), cout + 8, N, H, W, cin, cout, stride, dil, 1, 0.1, None))
        torch.cuda.synchronize()
        ys.append(y)
    close(ys[0][..., :cout], exp)
    assert float(ys[0][..., cout:].min()) == -7.0 and float(ys[0][..., cout:].max()) == -7.0
    assert torch.equal(ys[0], ys[1])
    # no activation
    y = torch.empty((N, Ho, Wo, cout), device="cuda")
    _lib.check(sk(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, cout, stride, dil, 0, 0.0, None))
    close(y, orc.conv3x3(x, k, b, stride, dil, None))
    # physical layout: channels padded / permuted, channel stride beyond them
    cs = cin + 32
    rs = np.random.RandomState(7)
    pos = np.sort(rs.choice(cs, cin, replace=False))
    cmap = np.full((cs,), -1, np.int32)
    cmap[pos] = np.arange(cin, dtype=np.int32)
    xp = rnd((N, H, W, cs + 4), 474)
    xp[..., pos] = x
    cm = torch.from_numpy(cmap).cuda()
    packed2 = torch.empty(L.pwc_conv3x3_sk_packed_floats(cs, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_sk_pack_f32(_p(kg), _p(cm), cin, cs, cout, _p(packed2), None))
    y = torch.empty((N, Ho, Wo, cout), device="cuda")
    xpg = gpu(xp)
    wide = tile > 30 and cs > (288 if stride == 1 else 128)       # (the padded input is too wide for the LDS-patch form)
    if wide:
        assert L.pwc_conv3x3_sk_variant_f32(_p(xpg), cs + 4, _p(packed2), _p(bg), _p(y), cout, N, H, W, cs, cout, stride, dil, 1, 0.1,
                                            tile, None) == -4      # PWC_EUNSUPPORTED, said -- not silently another tile
    _lib.check(sk(_p(xpg), cs + 4, _p(packed2), _p(bg), _p(y), cout, N, H, W, cs, cout, stride, dil, 1, 0.1, None, t=0 if wide else tile))
    close(y, exp)


@pytest.mark.parametrize("N,H,W,cin", [
    (16, 112, 256, 32), (8, 112, 256, 64), (8, 56, 128, 64), (1, 40, 70, 32), (3, 33, 45, 64), (1, 8, 32, 32), (1, 4, 32, 64),
    (1, 3, 5, 64), (1, 1, 1, 32), (2, 61, 100, 64), (2, 9, 100, 32), (1, 240, 480, 64)])
def test_conv_lds_resident_weights_kernel_vs_oracle(pa, N, H, W, cin):
    """pwc_conv3x3_w32_f32 (round 6, conv3x3_w32.hip: 32 output channels, the whole weight tensor resident in the LDS, patches
    through registers): fp_extractor/conv2d_4 (32 -> 32 at 16 x 112 x 256), optflow_4/conv2d_4 and context/conv2d_5 (64 -> 32 at
    8 x 112 x 256), optflow_3/conv2d_4 of BASELINE configs[1] and the 64 -> 32 layer of configs[4] at full size, ragged tiles
    (H % 8, H % 4, W % 32 != 0), images smaller than a tile, more tiles than CUs and fewer; strided output with untouched
    neighbours, no activation, padded / permuted input channels (cin_map) at a channel stride beyond them; the two K halves of
    the 64-channel form are added in a fixed order: launches repeat bitwise."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = rnd((N, H, W, cin), 481)
    k = rnd((3, 3, cin, 32), 482) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((32,), 483) * 0.1
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_w32_packed_floats(cin), device="cuda")
    _lib.check(L.pwc_conv3x3_w32_pack_f32(_p(kg), None, cin, cin, _p(packed), None))
    big = N * H * W > 200000
    exp = orc.conv3x3(x[:2] if big else x, k, b, 1, 1, 0.1)
    ys = []
    for _ in range(2):
        y = torch.full((N, H, W, 40), -7.0, device="cuda")
        _lib.check(L.pwc_conv3x3_w32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), 40, N, H, W, cin, 32, 1, 0.1, None))
        torch.cuda.synchronize()
        ys.append(y)
    close(ys[0][:exp.shape[0], ..., :32], exp)
    if big:                                                  # (the last image too: the persistent walk reaches every tile)
        close(ys[0][N - 1:, ..., :32], orc.conv3x3(x[N - 1:], k, b, 1, 1, 0.1))
    assert float(ys[0][..., 32:].min()) == -7.0 and float(ys[0][..., 32:].max()) == -7.0
    assert torch.equal(ys[0], ys[1])
    if big:
        return
    y = torch.empty((N, H, W, 32), device="cuda")
    _lib.check(L.pwc_conv3x3_w32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), 32, N, H, W, cin, 32, 0, 0.0, None))
    close(y, orc.conv3x3(x, k, b, 1, 1, None))
    # physical layout: fewer logical channels, scattered over the physical ones, at a channel stride beyond them
    clog = cin - 5
    rs = np.random.RandomState(9)
    pos = np.sort(rs.choice(cin, clog, replace=False))
    cmap = np.full((cin,), -1, np.int32)
    cmap[pos] = np.arange(clog, dtype=np.int32)
    xl = rnd((N, H, W, clog), 484)
    xp = rnd((N, H, W, cin + 4), 485)
    xp[..., pos] = xl
    k2 = rnd((3, 3, clog, 32), 486) * float(1.0 / np.sqrt(9 * clog))
    cm = torch.from_numpy(cmap).cuda()
    packed2 = torch.empty(L.pwc_conv3x3_w32_packed_floats(cin), device="cuda")
    _lib.check(L.pwc_conv3x3_w32_pack_f32(_p(gpu(k2)), _p(cm), clog, cin, _p(packed2), None))
    xpg = gpu(xp)
    _lib.check(L.pwc_conv3x3_w32_f32(_p(xpg), cin + 4, _p(packed2), _p(bg), _p(y), 32, N, H, W, cin, 32, 1, 0.1, None))
    close(y, orc.conv3x3(xl, k2, b, 1, 1, 0.1))
    # what the entry point refuses
    assert L.pwc_conv3x3_w32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), 32, N, H, W, 48, 32, 1, 0.1, None) == -4
    assert L.pwc_conv3x3_w32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), 64, N, H, W, cin, 64, 1, 0.1, None) == -4
    assert L.pwc_conv3x3_w32_f32(_p(xg), cin - 4, _p(packed), _p(bg), _p(y), 32, N, H, W, cin, 32, 1, 0.1, None) == -1


@pytest.mark.parametrize("N,H,W,cin,stride", [
    (2, 112, 256, 32, 1), (16, 112, 256, 32, 1), (2, 224, 512, 16, 2), (16, 224, 512, 16, 2), (1, 40, 70, 32, 1), (3, 33, 45, 16, 2),
    (1, 30, 64, 16, 1), (2, 9, 13, 16, 2), (1, 8, 32, 32, 1), (1, 3, 5, 32, 1), (1, 1, 1, 16, 2), (2, 61, 100, 16, 2)])
def test_conv_thin_input_kernel_vs_oracle(pa, N, H, W, cin, stride):
    """pwc_conv3x3_t32_f32 (round 5, conv3x3_t32.hip: weights stationary, patches by LDS-DMA): fp_extractor/conv2d_3 (16 -> 32,
    stride 2) and conv2d_4 (32 -> 32) of BASELINE configs[1] at full size, ragged tiles (H % 8, W % 32 != 0), odd sizes under stride
    2, images smaller than a tile, fewer tiles than workgroups; strided output with untouched neighbours; padded / permuted input
    channels; launches repeat bitwise."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    cout = 32
    x = rnd((N, H, W, cin), 571)
    k = rnd((3, 3, cin, cout), 572) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 573) * 0.1
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_t32_packed_floats(cin), device="cuda")
    _lib.check(L.pwc_conv3x3_t32_pack_f32(_p(kg), None, cin, cin, _p(packed), None))
    big = N * H * W > 200000
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    ys = []
    for _ in range(2):
        y = torch.full((N, Ho, Wo, cout + 8), -7.0, device="cuda")
        _lib.check(L.pwc_conv3x3_t32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout + 8, N, H, W, cin, cout, stride, 1, 0.1, None))
        torch.cuda.synchronize()
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    assert float(ys[0][..., cout:].min()) == -7.0 and float(ys[0][..., cout:].max()) == -7.0
    if big:
        for i in (0, N - 1):
            close(ys[0][i:i + 1, ..., :cout], orc.conv3x3(x[i:i + 1], k, b, stride, 1, 0.1))
        assert bool(torch.isfinite(ys[0]).all())
        return
    close(ys[0][..., :cout], orc.conv3x3(x, k, b, stride, 1, 0.1))
    y = torch.empty((N, Ho, Wo, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_t32_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, cout, stride, 0, 0.0, None))
    close(y, orc.conv3x3(x, k, b, stride, 1, None))
    # physical layout: 12 / 20 logical channels scattered over the 16 / 32 physical ones, channel stride beyond them
    cl = cin - 4 if cin == 16 else cin - 12
    rs = np.random.RandomState(7)
    pos = np.sort(rs.choice(cin, cl, replace=False))
    cmap = np.full((cin,), -1, np.int32)
    cmap[pos] = np.arange(cl, dtype=np.int32)
    xl = rnd((N, H, W, cl), 574)
    xp = rnd((N, H, W, cin + 4), 575)
    xp[..., pos] = xl
    kl = rnd((3, 3, cl, cout), 576) * float(1.0 / np.sqrt(9 * cl))
    cm = torch.from_numpy(cmap).cuda()
    packed2 = torch.empty(L.pwc_conv3x3_t32_packed_floats(cin), device="cuda")
    _lib.check(L.pwc_conv3x3_t32_pack_f32(_p(gpu(kl)), _p(cm), cl, cin, _p(packed2), None))
    xpg = gpu(xp)
    _lib.check(L.pwc_conv3x3_t32_f32(_p(xpg), cin + 4, _p(packed2), _p(bg), _p(y), cout, N, H, W, cin, cout, stride, 1, 0.1, None))
    close(y, orc.conv3x3(xl, kl, b, stride, 1, 0.1))


def test_conv_thin_input_kernel_rejections(pa):
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = torch.zeros((1, 8, 32, 32), device="cuda")
    y = torch.zeros((1, 8, 32, 32), device="cuda")
    w = torch.zeros((L.pwc_conv3x3_t32_packed_floats(32),), device="cuda")
    b = torch.zeros((32,), device="cuda")
    assert L.pwc_conv3x3_t32_f32(_p(x), 32, _p(w), _p(b), _p(y), 32, 1, 8, 32, 32, 32, 2, 1, 0.1, None) == -4      # 32 channels, stride 2
    assert L.pwc_conv3x3_t32_f32(_p(x), 32, _p(w), _p(b), _p(y), 32, 1, 8, 32, 32, 64, 1, 1, 0.1, None) == -4      # C_out
    assert L.pwc_conv3x3_t32_f32(_p(x), 32, _p(w), _p(b), _p(y), 32, 1, 8, 32, 48, 32, 1, 1, 0.1, None) == -4      # C_in
    assert L.pwc_conv3x3_t32_supported(16, 112, 256, 32, 32, 1) == 0 and L.pwc_conv3x3_t32_supported(16, 224, 512, 16, 32, 2) == 1
    assert L.pwc_conv3x3_t32_supported(1, 112, 256, 32, 32, 1) == 0 and L.pwc_conv3x3_t32_supported(16, 112, 256, 32, 64, 2) == 0
    assert L.pwc_conv3x3_t32_packed_floats(48) == 0


def test_conv_small_launch_kernel_error_range_and_rejections(pa):
    """The small-launch kernel against a float64 convolution: not further from it than the fp32 matrix-pipe kernel; an input
    beyond fp16's range gives NaN in the outputs that read it; unsupported shapes / alignments are refused."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cin, cout = 2, 7, 16, 128, 64
    x = rnd((N, H, W, cin), 481) * 3.0
    k = rnd((3, 3, cin, cout), 482) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 483) * 0.1
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    xp = np.zeros((N, H + 2, W + 2, cin), np.float64)
    xp[:, 1:-1, 1:-1] = x
    ref = sum(np.einsum("nhwc,co->nhwo", xp[:, dy:dy + H, dx:dx + W], k[dy, dx].astype(np.float64))
              for dy in range(3) for dx in range(3)) + b
    packed = torch.empty(L.pwc_conv3x3_sk_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_sk_pack_f32(_p(kg), None, cin, cin, cout, _p(packed), None))
    y = torch.empty((N, H, W, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_sk_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, cout, 1, 1, 0, 0.0, None))
    err_sk = float(np.abs(y.double().cpu().numpy() - ref).max())
    p32 = torch.empty(L.pwc_conv3x3_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_pack_f32(_p(kg), None, cin, cin, cout, _p(p32), None))
    y32 = torch.empty((N, H, W, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_f32(_p(xg), cin, _p(p32), _p(bg), _p(y32), cout, N, H, W, cin, cout, 1, 1, 0, 0.0, -1, 1, None, 0, None))
    err_32 = float(np.abs(y32.double().cpu().numpy() - ref).max())
    assert err_sk <= 1.5 * err_32 + 1e-7, (err_sk, err_32)
    xg[1, 3, 5, 7] = 70000.0
    _lib.check(L.pwc_conv3x3_sk_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, cout, 1, 1, 0, 0.0, None))
    bad = torch.isnan(y).any(dim=3)
    idx = torch.nonzero(bad)
    assert len(idx) == 9 and int(idx[:, 0].min()) == 1 and int((idx[:, 1] - 3).abs().max()) == 1 and int((idx[:, 2] - 5).abs().max()) == 1
    assert L.pwc_conv3x3_sk_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, 48, cout, 1, 1, 0, 0.0, None) == -4
    assert L.pwc_conv3x3_sk_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, 24, 1, 1, 0, 0.0, None) == -4
    assert L.pwc_conv3x3_sk_f32(_p(xg), cin + 2, _p(packed), _p(bg), _p(y), cout, N, H, W, cin, cout, 1, 1, 0, 0.0, None) != 0
    assert L.pwc_conv3x3_sk_supported(8, 7, 16, 288, 128, 1, 1) == 1
    assert L.pwc_conv3x3_sk_supported(8, 28, 64, 64, 32, 1, 1) == 1
    assert L.pwc_conv3x3_sk_supported(8, 28, 64, 224, 128, 1, 1) == 0
    assert L.pwc_conv3x3_sk_supported(16, 14, 32, 128, 128, 1, 1) == 1      # round 5, patch in the LDS: up to 2.4e8 multiply-adds
    assert L.pwc_conv3x3_sk_supported(8, 56, 128, 64, 32, 1, 1) == 0
    assert L.pwc_conv3x3_sk_supported(16, 28, 64, 96, 128, 2, 1) == 1
    assert L.pwc_conv3x3_sk_supported(8, 112, 256, 128, 128, 1, 1) == 0
    assert L.pwc_conv3x3_sk_supported(8, 7, 16, 48, 128, 1, 1) == 0


@pytest.mark.parametrize("N,H,W,cin,cout", [(2, 64, 96, 16, 32), (1, 34, 46, 32, 64), (2, 50, 70, 64, 96), (16, 112, 256, 32, 64),
                                            (3, 30, 64, 48, 128), (16, 224, 512, 16, 32), (1, 2, 2, 16, 32)])
def test_conv_f16x2_direct_stride2_vs_oracle(pa, N, H, W, cin, cout):
    """pwc_conv3x3_h2_stride2_f32 (round 5: the F16-pipe kernel over the input's four parity planes): TF 'SAME' stride-2 convolution
    of even sizes (pad bottom / right only: the last row / column of taps falls outside) incl. BASELINE configs[1]'s first two
    down-sampling layers at full size; ragged tiles, strided output with untouched neighbours, with and without the stream-K
    workspace, padded / permuted input channels; odd sizes are refused (the fp32 kernel takes them)."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = rnd((N, H, W, cin), 371)
    k = rnd((3, 3, cin, cout), 372) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 373) * 0.1
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_h2_stride2_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_h2_stride2_pack_f32(_p(kg), None, cin, cin, cout, _p(packed), None))
    Ho, Wo = H // 2, W // 2
    big = N * H * W > 200000
    exp = None if big else orc.conv3x3(x, k, b, 2, 1, 0.1)
    n_ws = L.pwc_conv3x3_h2_stride2_workspace_floats(N, H, W, cin, cout)
    wss = [None] + ([torch.full((n_ws,), -1, dtype=torch.int32, device="cuda").view(torch.float32)] if n_ws else [])
    outs = []
    for ws in wss:
        y = torch.full((N, Ho, Wo, cout + 8), -7.0, device="cuda")
        wp, wn = (None, 0) if ws is None else (_p(ws), ws.numel())
        _lib.check(L.pwc_conv3x3_h2_stride2_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout + 8, N, H, W, cin, cout, 1, 0.1, wp, wn, None, None))
        torch.cuda.synchronize()
        if big:
            for i in (0, N - 1):
                close(y[i:i + 1, ..., :cout], orc.conv3x3(x[i:i + 1], k, b, 2, 1, 0.1))
            assert bool(torch.isfinite(y).all())
        else:
            close(y[..., :cout], exp)
        assert float(y[..., cout:].min()) == -7.0 and float(y[..., cout:].max()) == -7.0
        outs.append(y)
    if big:
        assert len(outs) == 1 or float((outs[0] - outs[1]).abs().max()) <= 1e-5
    # physical layout: the input's channels padded / permuted (cin_map), channel stride beyond them
    if not big:
        cs = cin + 16
        rs = np.random.RandomState(7)
        pos = np.sort(rs.choice(cs, cin, replace=False))
        cmap = np.full((cs,), -1, np.int32)
        cmap[pos] = np.arange(cin, dtype=np.int32)
        xp = rnd((N, H, W, cs + 4), 374)
        xp[..., pos] = x
        cm = torch.from_numpy(cmap).cuda()
        packed2 = torch.empty(L.pwc_conv3x3_h2_stride2_packed_floats(cs, cout), device="cuda")
        _lib.check(L.pwc_conv3x3_h2_stride2_pack_f32(_p(kg), _p(cm), cin, cs, cout, _p(packed2), None))
        y = torch.full((N, Ho, Wo, cout), -7.0, device="cuda")
        xpg = gpu(xp)
        _lib.check(L.pwc_conv3x3_h2_stride2_f32(_p(xpg), cs + 4, _p(packed2), _p(bg), _p(y), cout, N, H, W, cs, cout, 1, 0.1, None, 0, None, None))
        torch.cuda.synchronize()
        close(y, exp)
    # odd sizes: not this kernel's
    y = torch.empty((N, (H + 2) // 2, Wo, cout), device="cuda")
    assert L.pwc_conv3x3_h2_stride2_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), cout, N, H - 1, W, cin, cout, 1, 0.1, None, 0, None, None) == -4
    assert L.pwc_conv3x3_h2_stride2_supported(N, H - 1, W, cin, cout) == 0
    # the extractor's layers of BASELINE configs[1] (16 images) that go to it, and the one that does not (7 x 16 outputs)
    assert L.pwc_conv3x3_h2_stride2_supported(16, 224, 512, 16, 32) == 1 and L.pwc_conv3x3_h2_stride2_supported(16, 112, 256, 32, 64) == 1
    assert L.pwc_conv3x3_h2_stride2_supported(16, 14, 32, 128, 192) == 0 and L.pwc_conv3x3_h2_stride2_supported(16, 56, 128, 64, 96) == 0


@pytest.mark.parametrize("N,H,W,xcs,ycs", [(2, 50, 70, 16, 16), (1, 16, 32, 16, 16), (3, 33, 47, 20, 24), (1, 97, 130, 16, 16),
                                          (16, 224, 512, 16, 16)])
def test_conv_c16_pair_vs_oracle(pa, N, H, W, xcs, ycs):
    """pwc_conv3x3_c16pair_f32: conv2d_1 + conv2d_2 of pyramid level 1 (16 -> 16 -> 16, leaky-relu behind each) in one launch
    with the intermediate in LDS, against the oracle's two convolutions: ragged 16 x 32-pixel tiles (the intermediate's zero
    padding at the image border is NOT a convolution result), strided input and output with untouched neighbours, the
    production shape (first and last image).  fp32 default tolerance."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = rnd((N, H, W, 16), 471)
    k1 = rnd((3, 3, 16, 16), 472) * float(1.0 / np.sqrt(9 * 16))
    k2 = rnd((3, 3, 16, 16), 473) * float(1.0 / np.sqrt(9 * 16))
    b1, b2 = rnd((16,), 474) * 0.1, rnd((16,), 475) * 0.1
    xp = np.full((N, H, W, xcs), 3.0, np.float32)
    xp[..., :16] = x
    xg = gpu(xp)
    packed = torch.empty(L.pwc_conv3x3_c16pair_packed_floats(), device="cuda")
    k1g, k2g = gpu(k1), gpu(k2)
    _lib.check(L.pwc_conv3x3_c16pair_pack_f32(_p(k1g), _p(k2g), _p(packed), None))
    y = torch.full((N, H, W, ycs), -7.0, device="cuda")
    b1g, b2g = gpu(b1), gpu(b2)
    _lib.check(L.pwc_conv3x3_c16pair_f32(_p(xg), xcs, _p(packed), _p(b1g), _p(b2g), _p(y), ycs, N, H, W, 0.1, None))
    torch.cuda.synchronize()
    for i in sorted({0, N - 1}):
        mid = orc.conv3x3(x[i:i + 1], k1, b1, 1, 1, 0.1)
        close(y[i:i + 1, ..., :16], orc.conv3x3(mid, k2, b2, 1, 1, 0.1))
    assert bool(torch.isfinite(y).all())
    if ycs > 16:
        assert float(y[..., 16:].min()) == -7.0 and float(y[..., 16:].max()) == -7.0


@pytest.mark.parametrize("Na,Nb,H0,W0,ycs", [(1, 1, 64, 96, 16), (2, 0, 33, 64, 24), (1, 2, 101, 140, 16), (1, 0, 32, 64, 16),
                                             (8, 8, 448, 1024, 16)])
def test_conv_c3_c16_pair_vs_oracle(pa, Na, Nb, H0, W0, ycs):
    """pwc_conv3x3_c3c16pair_f32: ALL of pyramid level 1 (stride-2 3 -> 16, then 16 -> 16 -> 16, leaky-relu behind each) in one
    launch from the raw images of the two frames (separate tensors), against the oracle's three convolutions: even and odd
    heights (TF 'SAME' pads the top row only for odd sizes), ragged tiles, the right / bottom zero column and row of the
    stride-2 layer, strided output with untouched neighbours, the production shape (first and last image).  fp32 default
    tolerance."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    assert L.pwc_conv3x3_c3c16pair_supported(16, 448, 1024) == 1 and L.pwc_conv3x3_c3c16pair_supported(16, 448, 1022) == 0
    xa, xb = rnd((Na, H0, W0, 3), 481), rnd((max(Nb, 1), H0, W0, 3), 482)
    k0 = rnd((3, 3, 3, 16), 483) * float(1.0 / np.sqrt(27))
    k1 = rnd((3, 3, 16, 16), 484) * float(1.0 / np.sqrt(9 * 16))
    k2 = rnd((3, 3, 16, 16), 485) * float(1.0 / np.sqrt(9 * 16))
    b0, b1, b2 = rnd((16,), 486) * 0.1, rnd((16,), 487) * 0.1, rnd((16,), 488) * 0.1
    xag, xbg, k0g, k1g, k2g, b0g, b1g, b2g = gpu(xa), gpu(xb), gpu(k0), gpu(k1), gpu(k2), gpu(b0), gpu(b1), gpu(b2)
    packed = torch.empty(L.pwc_conv3x3_c3c16pair_packed_floats(), device="cuda")
    _lib.check(L.pwc_conv3x3_c3c16pair_pack_f32(_p(k0g), _p(k1g), _p(k2g), _p(packed), None))
    H, W = -(-H0 // 2), W0 // 2
    N = Na + Nb
    y = torch.full((N, H, W, ycs), -7.0, device="cuda")
    _lib.check(L.pwc_conv3x3_c3c16pair_f32(_p(xag), Na, _p(xbg) if Nb else None, Nb, _p(packed), _p(b0g), _p(b1g), _p(b2g), _p(y), ycs,
                                           H0, W0, 0.1, None))
    torch.cuda.synchronize()
    for i in sorted({0, Na - 1, N - 1}):
        img = xa[i:i + 1] if i < Na else xb[i - Na:i - Na + 1]
        l0 = orc.conv3x3(img, k0, b0, 2, 1, 0.1)
        assert l0.shape == (1, H, W, 16)
        l1 = orc.conv3x3(l0, k1, b1, 1, 1, 0.1)
        close(y[i:i + 1, ..., :16], orc.conv3x3(l1, k2, b2, 1, 1, 0.1))
    assert bool(torch.isfinite(y).all())
    if ycs > 16:
        assert float(y[..., 16:].min()) == -7.0 and float(y[..., 16:].max()) == -7.0
    xo = torch.zeros((1, 32, 62, 3), device="cuda")
    assert L.pwc_conv3x3_c3c16pair_f32(_p(xo), 1, None, 0, _p(packed), _p(b0g), _p(b1g), _p(b2g), _p(y), ycs, 32, 62, 0.1, None) == -4


@pytest.mark.parametrize("N,H,W,ca,cb,cout,with_ws", [(2, 24, 40, 128, 32, 128, False), (1, 33, 47, 48, 16, 64, False),
                                                         (8, 112, 256, 128, 32, 128, True)])
def test_conv_f16x2_two_operand_and_status(pa, N, H, W, ca, cb, cout, with_ws):
    """pwc_conv3x3_h2_ex_f32 (round 5): the input channels given as TWO tensors (the estimator's first conv reading features_0
    from the pyramid tensor) give bit for bit what one tensor holding both gives; the third case is BASELINE configs[1]'s
    level-4 shape through the stream-K form.  The status words stay clear (they only ever carry a stream-K timeout); an operand
    of EITHER tensor beyond fp16's range gives NaN at exactly the pixels that read it, and pwc_resize_bilinear_status_f32 --
    the launch a forward ends with -- turns a NaN into PWC_STATUS_NONFINITE; pwc_absmax_f32 records the largest magnitude."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    cs_a, cs_b = ca + 16, cb + 8                      # both tensors are channel slices of wider ones
    xa = torch.randn((N, H, W, cs_a), generator=g, device="cuda")
    xb = torch.randn((N, H, W, cs_b), generator=g, device="cuda")
    one = torch.cat([xa[..., :ca], xb[..., :cb]], dim=3).contiguous()
    cin = ca + cb
    k = gpu(rnd((3, 3, cin, cout), 272) * float(1.0 / np.sqrt(9 * cin)))
    b = gpu(rnd((cout,), 273) * 0.1)
    packed = torch.empty(L.pwc_conv3x3_h2_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_h2_pack_f32(_p(k), None, cin, cin, cout, _p(packed), None))
    ws = None
    if with_ws:
        n = L.pwc_conv3x3_h2_workspace_floats(N, H, W, cin, cout, 1)
        assert n > 0
        ws = torch.full((n,), -1, dtype=torch.int32, device="cuda").view(torch.float32)
    wsa = (None, 0) if ws is None else (_p(ws), ws.numel())

    def run(two, status, xa_=xa, xb_=xb):
        y = torch.full((N, H, W, cout), -7.0, device="cuda")
        if two:
            rc = L.pwc_conv3x3_h2_ex_f32(_p(xa_), cs_a, ca, _p(xb_), cs_b, _p(packed), _p(b), _p(y), cout, N, H, W, cin, cout, 1,
                                         1, 0.1, *wsa, _p(status) if status is not None else None, None)
        else:
            rc = L.pwc_conv3x3_h2_ex_f32(_p(one), cin, 0, None, 0, _p(packed), _p(b), _p(y), cout, N, H, W, cin, cout, 1,
                                         1, 0.1, *wsa, _p(status) if status is not None else None, None)
        _lib.check(rc)
        torch.cuda.synchronize()
        return y
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    y1, y2 = run(False, None), run(True, status)
    assert torch.equal(y1, y2)
    if N * H * W <= 4096:
        close(y2, orc.conv3x3(one.cpu().numpy(), k.cpu().numpy(), b.cpu().numpy(), 1, 1, 0.1))
    assert int(status[0].item()) == 0
    _lib.check(L.pwc_absmax_f32(_p(xa), cs_a, N * H * W, ca, _p(status), None))
    _lib.check(L.pwc_absmax_f32(_p(xb), cs_b, N * H * W, cb, _p(status), None))
    torch.cuda.synchronize()
    assert float(status[1:2].view(torch.float32).item()) == float(one.abs().max())
    if with_ws:
        assert bool((ws.view(torch.int32) == -1).all())           # the launch leaves the workspace clean
    # a value beyond fp16's range in the SECOND tensor
    xb2 = xb.clone()
    xb2[N - 1, H // 2, W // 3, 1] = 7.0e4
    y3 = run(True, status, xb_=xb2)
    assert bool(torch.isnan(y3[N - 1, H // 2, W // 3]).all()) and not bool(torch.isnan(y3[0, 0, 0]).any())
    up = torch.empty((N, 2 * H, 2 * W, cout), device="cuda")
    for src, want in ((y2, 0), (y3, _lib.STATUS_NONFINITE)):
        _lib.check(L.pwc_resize_bilinear_status_f32(_p(src), cout, _p(up), cout, N, H, W, cout, 2 * H, 2 * W, 20.0, _p(status), None))
        torch.cuda.synchronize()
        assert int(status[0].item()) == want
    ref_up = torch.empty_like(up)
    _lib.check(L.pwc_resize_bilinear_f32(_p(y3), cout, _p(ref_up), cout, N, H, W, cout, 2 * H, 2 * W, 20.0, None))
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(up, nan=-1.0), torch.nan_to_num(ref_up, nan=-1.0))
    # argument checks: the first tensor's share must be whole stages
    assert L.pwc_conv3x3_h2_ex_f32(_p(xa), cs_a, ca - 8, _p(xb), cs_b, _p(packed), _p(b), _p(y1), cout, N, H, W, cin, cout, 1,
                                   1, 0.1, None, 0, None, None) == -1
    assert L.pwc_conv3x3_h2_ex_f32(_p(xa), cs_a, ca, _p(xb), cb - 4, _p(packed), _p(b), _p(y1), cout, N, H, W, cin, cout, 1,
                                   1, 0.1, None, 0, None, None) == -1


@pytest.mark.parametrize("N,H,W,c0,cout,with_ws", [(2, 24, 40, 32, 128, False), (1, 33, 47, 64, 64, False), (2, 28, 64, 96, 128, False),
                                                      (8, 112, 256, 32, 128, True)])
def test_conv_f16x2_three_operands_vs_oracle(pa, N, H, W, c0, cout, with_ws):
    """pwc_conv3x3_h2_ex3_f32 (round 6): the estimator's first conv on THREE dense tensors -- [cv 81 | flows_up_prev 2 | 0] in
    84-channel records (six 16-channel stages: the sixth runs 12 channels into the NEXT pixel's record, zero weights there),
    features_0, features_up_prev -- against the oracle's convolution of tf.concat([cv, features_0, flows_up_prev, features_up_prev])
    (reference modules.py:261-267) with the logical kernel, and bit for bit against the one-tensor launch of the same physical
    channel order with zeros in the padding channels; last case: BASELINE configs[1]'s level-4 shape through stream-K.  The
    garbage behind a record (the next record's first channels) and behind the tensor's end must not reach the result."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda"); g.manual_seed(21)
    cvx = torch.randn((N, H, W, 84), generator=g, device="cuda")
    cvx[..., 83] = 0.0
    f0 = torch.randn((N, H, W, c0 + 8), generator=g, device="cuda")       # a channel slice of a wider tensor
    fu = torch.randn((N, H, W, 32), generator=g, device="cuda")
    est = pa.OpticalFlowEstimator_custom(name="optflow_9")
    cm, cl = est.three_operand_map(81, c0, 32)
    assert cl == 81 + c0 + 2 + 32 and len(cm) == 96 + c0 + 32 and sorted(cm[cm >= 0].tolist()) == list(range(cl))
    cin = len(cm)
    k = gpu(rnd((3, 3, cl, cout), 372) * float(1.0 / np.sqrt(9 * cl)))
    b = gpu(rnd((cout,), 373) * 0.1)
    packed = torch.empty(L.pwc_conv3x3_h2_packed_floats(cin, cout), device="cuda")
    cmg = torch.from_numpy(cm).cuda()
    _lib.check(L.pwc_conv3x3_h2_pack_f32(_p(k), _p(cmg), cl, cin, cout, _p(packed), None))
    ws = None
    if with_ws:
        n = L.pwc_conv3x3_h2_workspace_floats(N, H, W, cin, cout, 1)
        assert n > 0
        ws = torch.full((n,), -1, dtype=torch.int32, device="cuda").view(torch.float32)
    wsa = (None, 0) if ws is None else (_p(ws), ws.numel())
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    y3 = torch.full((N, H, W, cout), -7.0, device="cuda")
    _lib.check(L.pwc_conv3x3_h2_ex3_f32(_p(cvx), 84, 96, _p(f0), c0 + 8, c0, _p(fu), 32, _p(packed), _p(b), _p(y3), cout,
                                        N, H, W, cin, cout, 1, 0.1, *wsa, _p(status), None))
    # the same physical channels as ONE tensor (padding channels zero)
    one = torch.zeros((N, H, W, cin), device="cuda")
    one[..., :84] = cvx
    one[..., 96:96 + c0] = f0[..., :c0]
    one[..., 96 + c0:] = fu
    y1 = torch.full((N, H, W, cout), -7.0, device="cuda")
    _lib.check(L.pwc_conv3x3_h2_ex_f32(_p(one), cin, 0, None, 0, _p(packed), _p(b), _p(y1), cout, N, H, W, cin, cout, 1,
                                       1, 0.1, *wsa, None, None))
    torch.cuda.synchronize()
    assert torch.equal(y1, y3)
    assert int(status[0].item()) == 0
    if with_ws:
        assert bool((ws.view(torch.int32) == -1).all())
    if N * H * W <= 4096:
        logical = torch.cat([cvx[..., :81], f0[..., :c0], cvx[..., 81:83], fu], dim=3).cpu().numpy()
        close(y3, orc.conv3x3(logical, k.cpu().numpy(), b.cpu().numpy(), 1, 1, 0.1))
    # argument checks: whole stages, strides that cover the operand (up to the 12 channels of slack), a third tensor needs a second
    bad = lambda *a: L.pwc_conv3x3_h2_ex3_f32(*a)
    args = [_p(cvx), 84, 96, _p(f0), c0 + 8, c0, _p(fu), 32, _p(packed), _p(b), _p(y1), cout, N, H, W, cin, cout, 1, 0.1, None, 0, None, None]
    for pos, val in ((2, 88), (1, 80), (5, c0 - 8), (7, 16), (3, None), (6, None)):
        a2 = list(args); a2[pos] = val
        assert bad(*a2) in (-1, -2, -3), (pos, val)


def test_conv_f16x2_direct_physical_layout_range_and_plan(pa):
    """Padded / permuted physical input channels through cin_map (the estimator buffers); operands of very different
    magnitudes (the split is relative, not absolute); an input beyond fp16's range poisons exactly the outputs that
    read it (NaN, never a silently wrong number); and the shapes the model routes to this kernel."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(5)
    N, H, W, cin, cout, cs = 1, 32, 64, 147, 128, 160
    cmap = np.full((cs,), -1, np.int32)
    pos = np.sort(rs.choice(cs, cin, replace=False))
    cmap[pos] = np.arange(cin, dtype=np.int32)
    xl = rnd((N, H, W, cin), 181)
    xp = rnd((N, H, W, cs), 182)                       # padding channels hold finite values the zero weights must erase
    xp[..., pos] = xl
    k = rnd((3, 3, cin, cout), 183) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 184) * 0.1
    close(run_conv_h2(xp, k, b, 0.1, cin_map=cmap), orc.conv3x3(xl, k, b, 1, 1, 0.1))
    # magnitudes: activations x 1e3 and x 1e-4 (the second product sits in fp16's normal range only because m' is scaled)
    for scale in (1e3, 1e-4):
        bs = (b * scale).astype(np.float32)
        close(run_conv_h2((xp * scale).astype(np.float32), k, bs, 0.1, cin_map=cmap),
              orc.conv3x3((xl * scale).astype(np.float32), k, bs, 1, 1, 0.1))
    # range: one input of 1e5 at (10, 20), logical channel 3
    xb = xp.copy()
    xb[0, 10, 20, pos[3]] = 1.0e5
    yb = run_conv_h2(xb, k, b, 0.1, cin_map=cmap).cpu().numpy()
    good = orc.conv3x3(xl, k, b, 1, 1, 0.1)
    nan = np.isnan(yb)
    assert nan[0, 9:12, 19:22, :].all() and int(nan.sum()) == 9 * cout
    assert float(np.abs(np.where(nan, 0.0, yb - good)).max()) <= 1e-5 * float(np.abs(good).max())
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 160, 128, 1) == 1
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 128, 96, 8) == 1
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 96, 64, 16) == 1            # 16 x 8 lattice, 7 x 32 pixels each
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 64, 32, 1) == 1
    assert L.pwc_conv3x3_h2_supported(8, 56, 128, 192, 128, 1) == 1
    assert L.pwc_conv3x3_h2_supported(8, 96, 256, 128, 64, 16) == 0         # 6-row sub-lattices
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 96, 32, 16) == 0         # 16-column sub-lattices, no 8-row tile for 32 couts
    assert L.pwc_conv3x3_h2_supported(8, 14, 32, 128, 128, 1) == 0          # a coarse level does not fill the GPU
    assert L.pwc_conv3x3_h2_supported(8, 112, 256, 128, 48, 1) == 0         # Cout % 32
    assert L.pwc_conv3x3_h2_plan(8, 112, 256, 128, 96, 1) == 3
    assert L.pwc_conv3x3_h2_plan(8, 112, 256, 64, 32, 1) == 4
    assert L.pwc_conv3x3_h2_plan(8, 112, 256, 128, 128, 1) in (1, 2)


@pytest.mark.parametrize("N,H,W,cin,cout,dil,csplit", [
    (2, 14, 32, 256, 128, 1, 4), (1, 28, 64, 224, 96, 1, 2), (2, 14, 32, 112, 48, 1, 3), (1, 28, 64, 64, 32, 1, 2),
    (1, 20, 36, 128, 64, 2, 4), (8, 14, 32, 128, 128, 1, 0), (1, 16, 16, 80, 16, 1, 5)])
def test_conv_winograd_channel_split(pa, N, H, W, cin, cout, dil, csplit):
    """The input-channel stages dealt to csplit workgroups per tile (under-filled launches): partial slabs + the
    fixed-order reduce with bias and activation; uneven stage counts (7 stages over 3, 5 over 5), strided and
    unaligned outputs, csplit 0 = whatever pwc_conv3x3_wino_split_plan says for the shape."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = rnd((N, H, W, cin), 171)
    k = rnd((3, 3, cin, cout), 172) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 173) * 0.1
    if csplit == 0:
        csplit = L.pwc_conv3x3_wino_split_plan(N, H, W, cin, cout, dil)
        assert csplit > 1, "this shape is meant to be split by the planner"
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    packed = torch.empty(L.pwc_conv3x3_wino_packed_floats(cin, cout), device="cuda")
    _lib.check(L.pwc_conv3x3_wino_pack_f32(_p(kg), None, cin, cin, cout, _p(packed), None))
    ws = torch.full((L.pwc_conv3x3_wino_split_workspace_floats(N, H, W, cout, csplit) + 8,), float("nan"), device="cuda")
    for y_cs, slope in ((cout, 0.1), (cout + 3, None)):
        y = torch.full((N, H, W, y_cs), -7.0, device="cuda")
        _lib.check(L.pwc_conv3x3_wino_split_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cin, cout, dil,
                                                0 if slope is None else 1, 0.0 if slope is None else slope, csplit,
                                                _p(ws), ws.numel(), None))
        torch.cuda.synchronize()
        close(y[..., :cout], orc.conv3x3(x, k, b, 1, dil, slope), rel=2e-5)
        if y_cs > cout:
            assert float(y[..., cout:].min()) == -7.0
        again = torch.full((N, H, W, y_cs), -7.0, device="cuda")
        _lib.check(L.pwc_conv3x3_wino_split_f32(_p(xg), cin, _p(packed), _p(bg), _p(again), y_cs, N, H, W, cin, cout, dil,
                                                0 if slope is None else 1, 0.0 if slope is None else slope, csplit,
                                                _p(ws), ws.numel(), None))
        assert torch.equal(again, y)
    # a workspace that is too small, or more parts than stages, is refused
    assert L.pwc_conv3x3_wino_split_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cin, cout, dil, 0, 0.0,
                                        csplit, _p(ws), 16, None) != 0
    assert L.pwc_conv3x3_wino_split_f32(_p(xg), cin, _p(packed), _p(bg), _p(y), y_cs, N, H, W, cin, cout, dil, 0, 0.0,
                                        cin // 16 + 1, _p(ws), ws.numel(), None) != 0


def test_conv_winograd_unaligned_output_stride_and_slopes(pa):
    """channel stride not a multiple of 4 (scalar stores through the buffer resource); slopes > 1
    and < 0 (tf.nn.leaky_relu is max(v, slope*v) whatever the slope)"""
    x = rnd((2, 21, 37, 32), 91)
    k = rnd((3, 3, 32, 48), 92) * float(1.0 / np.sqrt(9 * 32))
    b = rnd((48,), 93) * 0.1
    y = run_conv_wino(x, k, b, 0.1, y_cs=48 + 3)
    close(y[..., :48], orc.conv3x3(x, k, b, 1, 1, 0.1), rel=2e-5)
    assert float(y[..., 48:].min()) == -7.0 and float(y[..., 48:].max()) == -7.0
    for slope in (1.5, -0.25):
        close(run_conv_wino(x, k, b, slope), orc.conv3x3(x, k, b, 1, 1, slope), rel=2e-5)


def test_conv_winograd_physical_layout(pa):
    from pwcnet_amd.weights import estimator_layout
    lay = estimator_layout(4, False)
    xl = rnd((1, 20, 24, lay.n_logical), 74)
    p2l = np.asarray(lay.phys2log)
    xp = np.zeros((1, 20, 24, lay.n_phys), np.float32)
    xp[..., p2l >= 0] = xl[..., p2l[p2l >= 0]]
    k = rnd((3, 3, 147, 128), 75) * 0.03
    b = rnd((128,), 76) * 0.1
    close(run_conv_wino(xp, k, b, 0.1, cin_map=lay.cin_map()), orc.conv3x3(xl, k, b, 1, 1, 0.1), rel=2e-5)


@pytest.mark.parametrize("tile", list(range(15)))
def test_conv_mfma_every_tile_config(pa, tile):
    bn = [128, 96, 64, 32, 16, 128, 96, 64, 32, 16, 128, 96, 64, 32, 16][tile]
    cout = {128: 128, 96: 192, 64: 64, 32: 32, 16: 48}[bn]
    for cin in (16, 64):            # KC = 16 and KC = 32 instantiations
        x = rnd((2, 19, 37, cin), 4 + tile)
        k = rnd((3, 3, cin, cout), 5) * float(1.0 / np.sqrt(9 * cin))
        b = rnd((cout,), 6) * 0.1
        y = run_conv_mfma(x, k, b, 1, 1, 0.1, tile=tile)
        close(y, orc.conv3x3(x, k, b, 1, 1, 0.1))


@pytest.mark.parametrize("split", [3, 9])
@pytest.mark.parametrize("N,H,W,cin,cout,stride,dil", [
    (2, 7, 16, 192, 192, 1, 1), (1, 14, 32, 256, 128, 1, 1), (1, 28, 64, 96, 64, 1, 1), (1, 9, 13, 64, 32, 2, 1),
    (1, 20, 24, 128, 96, 1, 4)])
def test_conv_mfma_tap_split_vs_oracle(pa, split, N, H, W, cin, cout, stride, dil):
    x = rnd((N, H, W, cin), 51)
    k = rnd((3, 3, cin, cout), 52) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 53) * 0.1
    exp = orc.conv3x3(x, k, b, stride, dil, 0.1)
    y = run_conv_mfma(x, k, b, stride, dil, 0.1, split=split)
    close(y, exp)
    y2 = run_conv_mfma(x, k, b, stride, dil, 0.1, split=split, y_cs=cout + 4)
    close(y2[..., :cout], exp)
    assert torch.equal(y, run_conv_mfma(x, k, b, stride, dil, 0.1, split=split))     # deterministic


def test_conv_mfma_auto_plan_large_m(pa):
    """large M, pixel count not a multiple of any tile: 76800 = 600 tiles of 128 px."""
    N, H, W, cin, cout = 1, 240, 320, 32, 128
    x = rnd((N, H, W, cin), 54)
    k = rnd((3, 3, cin, cout), 55) * 0.05
    b = rnd((cout,), 56) * 0.1
    close(run_conv_mfma(x, k, b, 1, 1, 0.1), orc.conv3x3(x, k, b, 1, 1, 0.1))


def test_conv_mfma_physical_layout_padding_and_strided_output(pa):
    """estimator-style input: logical 147 channels scattered in a 160-channel buffer
    (segments aligned to 4, zero padding channels), output into a channel slice."""
    from pwcnet_amd.weights import estimator_layout
    lay = estimator_layout(4, False)
    assert (lay.n_phys, lay.n_logical) == (160, 147)
    xl = rnd((1, 20, 24, lay.n_logical), 7)
    p2l = np.asarray(lay.phys2log)
    xp = np.zeros((1, 20, 24, lay.n_phys), np.float32)
    xp[..., p2l >= 0] = xl[..., p2l[p2l >= 0]]
    k = rnd((3, 3, 147, 128), 8) * 0.03
    b = rnd((128,), 9) * 0.1
    y = run_conv_mfma(xp, k, b, 1, 1, 0.1, cin_map=lay.cin_map(), y_cs=144)
    exp = orc.conv3x3(xl, k, b, 1, 1, 0.1)
    close(y[..., :128], exp)
    assert float(y[..., 128:].min()) == -7.0 and float(y[..., 128:].max()) == -7.0   # untouched
    # linear (no activation) variant
    y2 = run_conv_mfma(xp, k, b, 1, 1, None, cin_map=lay.cin_map())
    close(y2, orc.conv3x3(xl, k, b, 1, 1, None))


def test_conv_mfma_linearity_full_size(pa):
    """BASELINE-size property (oracle too slow to brute-force every layer at 8x448x1024):
    conv(a*x1 + x2) - conv(0) is linear; checked on the 147->128 layer at 112x256, N=8."""
    N, H, W, cin, cout = 8, 112, 256, 160, 128
    k = rnd((3, 3, cin, cout), 10) * 0.03
    b = rnd((cout,), 11) * 0.1
    x1, x2 = rnd((N, H, W, cin), 12), rnd((N, H, W, cin), 13)
    y0 = run_conv_mfma(np.zeros_like(x1), k, b, 1, 1, None)
    y1 = run_conv_mfma(x1, k, b, 1, 1, None) - y0
    y2 = run_conv_mfma(x2, k, b, 1, 1, None) - y0
    y12 = run_conv_mfma(2.0 * x1 + x2, k, b, 1, 1, None) - y0
    err = float((y12 - (2.0 * y1 + y2)).abs().max())
    assert err <= 2e-5 * float(y12.abs().max()) + 1e-6
    # and one image row band against the oracle
    exp = orc.conv3x3(x1[:1], k, b, 1, 1, None)
    close(run_conv_mfma(x1[:1], k, b, 1, 1, None), exp)


# ------------------------------------------------------------------ conv (direct)
def run_conv_direct(x, k, b, stride, dil, slope, residual=None):
    from pwcnet_amd import _lib
    L = _lib.lib()
    N, H, W, cs = x.shape
    cin, cout = k.shape[2], k.shape[3]
    xg, kg, bg = gpu(x), gpu(k), gpu(b)
    Ho, Wo = -(-H // stride), -(-W // stride)
    y = torch.empty((N, Ho, Wo, cout), device="cuda")
    rg = None if residual is None else gpu(residual)
    _lib.check(L.pwc_conv3x3_direct_f32(_p(xg), cs, _p(kg), _p(bg), _p(y), cout,
                                        _p(rg) if rg is not None else None,
                                        0 if rg is None else residual.shape[3], N, H, W, cin, cout, stride, dil,
                                        0 if slope is None else 1, 0.0 if slope is None else slope, None))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout,stride,dil,slope", [
    (2, 32, 64, 3, 16, 2, 1, 0.1), (1, 17, 23, 3, 16, 2, 1, 0.1), (2, 16, 24, 32, 2, 1, 1, None),
    (1, 9, 11, 5, 7, 1, 2, 0.1), (1, 12, 12, 8, 13, 2, 1, None), (1, 10, 14, 36, 4, 1, 1, 0.1)])
def test_conv_direct_vs_oracle(pa, N, H, W, cin, cout, stride, dil, slope):
    x = rnd((N, H, W, cin), 14)
    k = rnd((3, 3, cin, cout), 15) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 16) * 0.1
    close(run_conv_direct(x, k, b, stride, dil, slope), orc.conv3x3(x, k, b, stride, dil, slope))


@pytest.mark.parametrize("N,H,W,stride,dil,slope", [
    (2, 448, 1024, 2, 1, 0.1), (1, 33, 47, 2, 1, 0.1), (2, 16, 16, 1, 1, 0.1), (1, 21, 35, 1, 2, None), (3, 7, 5, 2, 1, 0.1),
    (1, 64, 31, 2, 2, 0.1), (1, 1, 1, 2, 1, 0.1)])
def test_conv_first_layer_mfma_kernel_vs_oracle(pa, N, H, W, stride, dil, slope):
    """3 -> 16 channels (reference modules.py:64, fp_extractor/conv2d) runs conv3x3_cin3_mfma_kernel: interior groups,
    groups on every SAME-padding border, ragged rows (Wo % 16 != 0), every stride / dilation the entry point accepts;
    the first case is the bench layer at full size."""
    x = rnd((N, H, W, 3), 140)
    k = rnd((3, 3, 3, 16), 141) * float(1.0 / np.sqrt(27))
    b = rnd((16,), 142) * 0.1
    got = run_conv_direct(x, k, b, stride, dil, slope)
    close(got, orc.conv3x3(x, k, b, stride, dil, slope))


def test_conv_direct_residual_flow_head(pa):
    x = rnd((2, 14, 18, 32), 17)
    k = rnd((3, 3, 32, 2), 18) * 0.05
    b = rnd((2,), 19) * 0.1
    res = rnd((2, 14, 18, 2), 20)
    close(run_conv_direct(x, k, b, 1, 1, None, residual=res), orc.conv3x3(x, k, b, 1, 1, None, residual=res))


@pytest.mark.parametrize("N,H,W,res", [(1, 128, 128, True), (2, 131, 150, False), (1, 112, 256, True), (8, 112, 256, True),
                                       (1, 8, 2050, False), (2, 64, 64, True), (1, 65, 67, False), (3, 56, 128, True)])
def test_conv_flow_head_gemm_kernel(pa, N, H, W, res):
    """32 -> 2 heads on maps of >= 4096 pixels take conv3x3_head2_mfma_kernel (a 1x1 GEMM to 9 x 2 partial outputs per input
    pixel on the matrix pipe + 9 shifted adds from LDS): 8 x 32-pixel tiles, ragged edges, one-tile maps; (8, 112, 256) is
    the bench's level-4 head at full size."""
    x = rnd((N, H, W, 32), 117)
    k = rnd((3, 3, 32, 2), 118) * 0.05
    b = rnd((2,), 119) * 0.1
    r = rnd((N, H, W, 2), 120) if res else None
    close(run_conv_direct(x, k, b, 1, 1, None, residual=r), orc.conv3x3(x, k, b, 1, 1, None, residual=r))
    close(run_conv_direct(x, k, b, 1, 1, 0.1), orc.conv3x3(x, k, b, 1, 1, 0.1))


@pytest.mark.parametrize("N,H,W,cin,res", [(2, 7, 16, 724, False), (1, 30, 45, 100, True), (1, 64, 96, 1384, True),
                                           (2, 9, 70, 64, False), (1, 17, 33, 92, True),
                                           # Cin % 16 == 0 (the physical layouts of the model): the matrix-pipe form of round 5, incl. a
                                           # last chunk of 16 channels, ragged tiles, an image smaller than a tile, BASELINE configs[3]'s level 4
                                           (2, 7, 16, 720, True), (1, 30, 45, 112, True), (1, 64, 96, 1392, False), (1, 3, 5, 80, True),
                                           (1, 112, 256, 576, True)])
def test_conv_flow_head_wide_kernel(pa, N, H, W, cin, res):
    """Cin >= 64 -> 2 heads (the dense-connection estimators' flow heads: Cin = 725 ... 3169 logical channels): tiles with
    a loop over 32-channel chunks, ragged tiles and a last chunk of fewer than 32 channels."""
    x = rnd((N, H, W, cin), 121)
    k = rnd((3, 3, cin, 2), 122) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((2,), 123) * 0.1
    r = rnd((N, H, W, 2), 124) if res else None
    close(run_conv_direct(x, k, b, 1, 1, None, residual=r), orc.conv3x3(x, k, b, 1, 1, None, residual=r))
    close(run_conv_direct(x, k, b, 1, 1, 0.1), orc.conv3x3(x, k, b, 1, 1, 0.1))


def test_conv_direct_equals_mfma(pa):
    x = rnd((1, 21, 30, 64), 21)
    k = rnd((3, 3, 64, 32), 22) * 0.05
    b = rnd((32,), 23) * 0.1
    a = run_conv_direct(x, k, b, 1, 1, 0.1)
    m = run_conv_mfma(x, k, b, 1, 1, 0.1)
    close(m, a.cpu().numpy())


# ------------------------------------------------------------------ cost volume
@pytest.mark.parametrize("N,H,W,C,R", [
    (2, 7, 16, 192, 4), (2, 14, 32, 128, 4), (1, 28, 64, 96, 4), (1, 56, 128, 64, 4), (2, 20, 70, 32, 4),
    (1, 5, 3, 8, 4), (1, 9, 130, 16, 4), (1, 12, 20, 4, 4), (1, 11, 37, 32, 2), (1, 8, 66, 16, 1),
    (1, 8, 66, 24, 3)])
def test_cost_volume_vs_oracle(pa, N, H, W, C, R):
    f0, f1 = rnd((N, H, W, C), 24), rnd((N, H, W, C), 25)
    cv = pa.CostVolumeLayer(R)(gpu(f0), gpu(f1))
    close(cv, orc.cost_volume(f0, f1, R), rel=2e-6, floor=2e-7)


def test_cost_volume_golden_and_strided_io(pa, golden_dir):
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    close(pa.CostVolumeLayer(4)(gpu(g["cv_f0"]), gpu(g["cv_f1"])), g["cv_out"], rel=2e-6, floor=2e-7)
    # inputs as channel slices of wider buffers, output into a slice (out_cs = 84)
    from pwcnet_amd.modules import View, sub_view
    N, H, W, C = g["cv_f0"].shape
    big0 = torch.zeros(N, H, W, C + 8, device="cuda"); big0[..., 4:4 + C] = gpu(g["cv_f0"])
    big1 = torch.zeros(N, H, W, C + 4, device="cuda"); big1[..., :C] = gpu(g["cv_f1"])
    out = torch.full((N, H, W, 84), 5.0, device="cuda")
    v0 = sub_view(View(big0.data_ptr(), C + 8, N, H, W, C + 8), 4, C)
    v1 = View(big1.data_ptr(), C + 4, N, H, W, C)
    pa.CostVolumeLayer(4)._run(v0, v1, View(out.data_ptr(), 84, N, H, W, 81))
    close(out[..., :81], g["cv_out"], rel=2e-6, floor=2e-7)
    assert float(out[..., 81:].min()) == 5.0


@pytest.mark.parametrize("N,H,W", [(1, 64, 64), (2, 70, 75), (3, 100, 75), (1, 129, 33), (9, 64, 96), (1, 17, 256)])
def test_cost_volume_rolling_kernel_vs_oracle(pa, N, H, W):
    """C = 32, search range 4, >= 4096 pixels: pwc_cost_volume_f32 takes the rolling-window kernel
    (cost_volume_roll.hip).  Ragged cases on purpose: W not a multiple of the 32-column strip, H not a
    multiple of the 4-row step, segments of unequal height, more work items than workgroups is covered
    by the full-size test; out_cs = 84 / 160 (the model's buffers) and strided inputs."""
    from pwcnet_amd import _lib
    from pwcnet_amd.modules import View, sub_view
    L = _lib.lib()
    C = 32
    assert L.pwc_cost_volume_uses_rolling_kernel(H, W, C, 4, C, C, 84) == 1
    assert L.pwc_cost_volume_uses_rolling_kernel(H, W, C, 4, C, C, 81) == 0      # out_cs % 4 != 0: tile kernel
    assert L.pwc_cost_volume_uses_rolling_kernel(H, W, 64, 4, 64, 64, 84) == 0
    f0, f1 = rnd((N, H, W, C), 60 + H), rnd((N, H, W, C), 61 + W)
    exp = orc.cost_volume(f0, f1, 4)
    big0 = torch.zeros(N, H, W, C + 8, device="cuda"); big0[..., 4:4 + C] = gpu(f0)
    big1 = torch.zeros(N, H, W, C + 4, device="cuda"); big1[..., :C] = gpu(f1)
    for ocs in (84, 160):
        out = torch.full((N, H, W, ocs), 7.0, device="cuda")
        v0 = sub_view(View(big0.data_ptr(), C + 8, N, H, W, C + 8), 4, C)
        v1 = View(big1.data_ptr(), C + 4, N, H, W, C)
        pa.CostVolumeLayer(4)._run(v0, v1, View(out.data_ptr(), ocs, N, H, W, 81))
        close(out[..., :81], exp, rel=2e-6, floor=2e-7)
        assert float(out[..., 81:].min()) == 7.0 and float(out[..., 81:].max()) == 7.0   # nothing beyond channel 80 touched
    # the tile kernel (dense out_cs = 81) gives the same numbers up to summation order
    cv = pa.CostVolumeLayer(4)(gpu(f0), gpu(f1))
    close(cv, exp, rel=2e-6, floor=2e-7)


def test_cost_volume_known_answers_gpu(pa):
    f = rnd((1, 16, 16, 32), 15)
    v, h = 2, -3
    shifted = np.zeros_like(f)
    shifted[:, v:, : 16 + h] = f[:, : 16 - v, -h:]
    cv = pa.CostVolumeLayer(4)(gpu(f), gpu(shifted)).cpu().numpy()
    assert int(np.argmax(cv[0, 6, 8])) == (v + 4) * 9 + (h + 4)
    c = (f * shifted).mean(axis=3)
    np.testing.assert_allclose(cv[..., 40], np.maximum(c, 0.1 * c), atol=1e-6)
    assert cv[0, 0, 0, 0] == 0.0


def test_cost_volume_full_size_properties(pa):
    """BASELINE size (N=8, 112x256x32): symmetry property cv(f0,f1)[y,x,(v,h)] ==
    cv(f1,f0)[y+v,x+h,(-v,-h)] and a row band against the oracle."""
    N, H, W, C = 8, 112, 256, 32
    f0, f1 = rnd((N, H, W, C), 26), rnd((N, H, W, C), 27)
    a = pa.CostVolumeLayer(4)(gpu(f0), gpu(f1)).cpu().numpy()
    b = pa.CostVolumeLayer(4)(gpu(f1), gpu(f0)).cpu().numpy()
    for (v, h) in [(0, 0), (4, -4), (-3, 2), (1, 4)]:
        d, dm = (v + 4) * 9 + (h + 4), (-v + 4) * 9 + (-h + 4)
        ys, xs = slice(max(0, -v), H - max(0, v)), slice(max(0, -h), W - max(0, h))
        yd, xd = slice(max(0, v), H - max(0, -v)), slice(max(0, h), W - max(0, -h))
        np.testing.assert_allclose(a[:, ys, xs, d], b[:, yd, xd, dm], atol=1e-6)
    exp = orc.cost_volume(f0[3:4], f1[3:4], 4)
    np.testing.assert_allclose(a[3:4], exp, atol=1e-6)


def test_cost_volume_rolling_kernel_full_size(pa):
    """BASELINE level-4 size through the rolling kernel: batch 16 (512 work items on 256 persistent
    workgroups: the item loop), output into a 160-channel estimator buffer; the whole result against the
    tile kernel, two images against the oracle."""
    from pwcnet_amd.modules import View
    N, H, W, C = 16, 112, 256, 32
    f0, f1 = rnd((N, H, W, C), 28), rnd((N, H, W, C), 29)
    g0, g1 = gpu(f0), gpu(f1)
    out = torch.zeros((N, H, W, 160), device="cuda")
    pa.CostVolumeLayer(4)._run(View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C),
                               View(out.data_ptr(), 160, N, H, W, 81))
    tile = pa.CostVolumeLayer(4)(g0, g1)                   # dense output: tile kernel
    assert float((out[..., :81] - tile).abs().max()) <= 1e-6
    assert float(out[..., 81:].abs().max()) == 0.0
    for n in (0, 15):
        np.testing.assert_allclose(out[n:n + 1, ..., :81].cpu().numpy(), orc.cost_volume(f0[n:n + 1], f1[n:n + 1], 4), atol=1e-6)


def test_cost_volume_rolling_kernel_stress_no_stale_lds(pa):
    """The rolling kernel never drains its DMA or its stores: a step's prefetch is waited for with
    `s_waitcnt vmcnt(#stores issued after it)`, which relies on a wave's memory operations retiring in issue order.
    If that ever failed, a step would read ring rows of an EARLIER launch or step.  40 launches on alternating,
    unrelated operand sets (so that stale rows would be plainly wrong), every result against the tile kernel."""
    from pwcnet_amd.modules import View
    N, H, W, C = 8, 112, 256, 32
    sets = [(gpu(rnd((N, H, W, C), 70 + i)), gpu(rnd((N, H, W, C), 80 + i) * (i + 1))) for i in range(3)]
    refs = [pa.CostVolumeLayer(4)(a, b) for a, b in sets]            # dense output: tile kernel
    out = torch.zeros((N, H, W, 160), device="cuda")
    worst = 0.0
    for it in range(40):
        a, b = sets[it % 3]
        pa.CostVolumeLayer(4)._run(View(a.data_ptr(), C, N, H, W, C), View(b.data_ptr(), C, N, H, W, C),
                                   View(out.data_ptr(), 160, N, H, W, 81))
        worst = max(worst, float((out[..., :81] - refs[it % 3]).abs().max()) / float(refs[it % 3].abs().max()))
    assert worst <= 2e-6, worst


# ------------------------------------------------------------------ warp
@pytest.mark.parametrize("wt", ["bilinear", "nearest"])
@pytest.mark.parametrize("N,H,W,C", [(2, 14, 32, 128), (1, 56, 128, 64), (2, 9, 11, 4), (1, 28, 64, 96)])
def test_warp_vs_oracle(pa, wt, N, H, W, C):
    x = rnd((N, H, W, C), 28)
    flow = util.flow_field(N, H, W, seed=29)
    out = pa.WarpingLayer(wt)(gpu(x), gpu(flow))
    exp = orc.warp(x, flow, wt)
    if wt == "nearest":
        np.testing.assert_array_equal(out.cpu().numpy(), exp)
    else:
        close(out, exp, rel=2e-6, floor=2e-6)


def test_warp_golden_and_known_answers(pa, golden_dir):
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    close(pa.WarpingLayer("bilinear")(gpu(g["warp_x"]), gpu(g["warp_flow"])), g["warp_bilinear"], rel=2e-6, floor=2e-6)
    np.testing.assert_array_equal(pa.WarpingLayer("nearest")(gpu(g["warp_x"]), gpu(g["warp_flow"])).cpu().numpy(),
                                  g["warp_nearest"])
    x = rnd((1, 8, 12, 8), 30)
    zero = np.zeros((1, 8, 12, 2), np.float32)
    np.testing.assert_array_equal(pa.WarpingLayer("bilinear")(gpu(x), gpu(zero)).cpu().numpy(), x)
    flow = zero.copy(); flow[..., 0], flow[..., 1] = 2.0, -3.0
    exp = x[:, np.clip(np.arange(8) - 3, 0, 7)][:, :, np.clip(np.arange(12) + 2, 0, 11)]
    np.testing.assert_array_equal(pa.WarpingLayer("bilinear")(gpu(x), gpu(flow)).cpu().numpy(), exp)
    with pytest.raises(AssertionError):
        pa.WarpingLayer("cubic")(gpu(x), gpu(flow))   # reference modules.py:149


@pytest.mark.parametrize("N,H,W,C", [(2, 14, 32, 128), (1, 28, 70, 32), (1, 56, 128, 64), (1, 6, 9, 16)])
def test_fused_warp_cost_volume_vs_oracle(pa, N, H, W, C):
    from pwcnet_amd.modules import View
    f0, f1 = rnd((N, H, W, C), 31), rnd((N, H, W, C), 32)
    flow = util.flow_field(N, H, W, seed=33) / 5.0
    exp = orc.cost_volume(f0, orc.warp(f1, flow, "bilinear", flow_scale=5.0), 4)
    g0, g1, gf = gpu(f0), gpu(f1), gpu(flow)
    out = torch.empty((N, H, W, 81), device="cuda")
    pa.CostVolumeLayer(4)._run(View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C),
                               View(out.data_ptr(), 81, N, H, W, 81),
                               flow=View(gf.data_ptr(), 2, N, H, W, 2), flow_scale=5.0)
    close(out, exp, rel=4e-6, floor=4e-7)


@pytest.mark.parametrize("kind", ["bilinear", "nearest"])
def test_warp_with_fused_copy(pa, kind):
    """pwc_warp_copy_f32: the warp plus a channel-slice copy (the concat of features_0) in one launch."""
    from pwcnet_amd.modules import View, sub_view
    N, H, W, C = 2, 9, 21, 24
    x, f0 = rnd((N, H, W, C), 51), rnd((N, H, W, C), 52)
    flow = util.flow_field(N, H, W, seed=53) / 2.5
    gx, g0, gf = gpu(x), gpu(f0), gpu(flow)
    out = torch.empty((N, H, W, C), device="cuda")
    E = torch.full((N, H, W, C + 12), -4.0, device="cuda")
    Ev = View(E.data_ptr(), C + 12, N, H, W, C + 12)
    pa.WarpingLayer(kind)._run(View(gx.data_ptr(), C, N, H, W, C), View(gf.data_ptr(), 2, N, H, W, 2),
                               View(out.data_ptr(), C, N, H, W, C), flow_scale=2.5,
                               copy=(View(g0.data_ptr(), C, N, H, W, C), sub_view(Ev, 8, C)))
    torch.cuda.synchronize()
    close(out, orc.warp(x, flow, kind, flow_scale=2.5), rel=2e-6, floor=1e-6)
    assert torch.equal(E[..., 8:8 + C], g0)
    assert float(E[..., :8].max()) == -4.0 and float(E[..., 8 + C:].max()) == -4.0


@pytest.mark.parametrize("N,H,W,C,with_flow", [
    (2, 7, 16, 192, False), (2, 14, 32, 128, True), (1, 28, 64, 96, True), (3, 9, 21, 32, True),
    (1, 5, 3, 8, True), (1, 17, 10, 48, False), (2, 8, 8, 20, True)])
def test_coarse_cost_volume_kernel_vs_oracle(pa, N, H, W, C, with_flow):
    """pwc_cost_volume_coarse_f32: warp + cost volume + f0 copy in one launch (coarse levels),
    written into channel slices of wider buffers; ragged tiles, C not a multiple of 32."""
    from pwcnet_amd.modules import View
    f0, f1 = rnd((N, H, W, C), 41), rnd((N, H, W, C), 42)
    flow = util.flow_field(N, H, W, seed=43) / 5.0
    f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0) if with_flow else f1
    exp = orc.cost_volume(f0, f1w, 4)
    g0, g1 = gpu(f0), gpu(f1)
    fl = torch.full((N, H, W, 4), 9.0, device="cuda")
    fl[..., :2] = gpu(flow)
    E = torch.full((N, H, W, 84 + C + 4), -3.0, device="cuda")            # [cv 81 | pad 3 | f0 C | pad 4]
    Ev = View(E.data_ptr(), 84 + C + 4, N, H, W, 84 + C + 4)
    from pwcnet_amd.modules import sub_view
    pa.CostVolumeLayer(4)._run(View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C),
                               sub_view(Ev, 0, 81), flow=View(fl.data_ptr(), 4, N, H, W, 2) if with_flow else None,
                               flow_scale=5.0, f0_copy=sub_view(Ev, 84, C), coarse=True)
    torch.cuda.synchronize()
    close(E[..., :81], exp, rel=4e-6, floor=4e-7)
    assert torch.equal(E[..., 84:84 + C], g0)
    assert float(E[..., 81:84].min()) == -3.0 and float(E[..., 84 + C:].max()) == -3.0


# ------------------------------------------------------------------ matrix-pipe fused kernel (cost_volume_mfma.hip)
def _run_concat(pa, f0, f1, flow, flow_scale, ecs, copy, pad, fill=-3.0, flow_cs=4, f16x2=True, blk=False):
    """pwc_warp_cost_volume_concat_f32 (f16x2=False) / pwc_warp_cost_volume_concat_h2_f32 (the F16-pipe kernel of round 5,
    cost_volume_h2.hip) into an estimator-style buffer [cv 81 | pad 3 | f0 C | rest]; returns E."""
    from pwcnet_amd.modules import View, sub_view
    N, H, W, C = f0.shape
    g0, g1 = gpu(f0), gpu(f1)
    fl = None
    if flow is not None:
        fl = torch.full((N, H, W, flow_cs), 9.0, device="cuda")
        fl[..., :2] = gpu(flow)
    E = torch.full((N, H, W, ecs), fill, device="cuda")
    Ev = View(E.data_ptr(), ecs, N, H, W, ecs)
    layer = pa.CostVolumeLayer(4)
    layer.f16x2 = f16x2
    v0, v1 = View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C)
    fv = View(fl.data_ptr(), flow_cs, N, H, W, 2) if fl is not None else None
    cpy = sub_view(Ev, 84, C) if copy else None
    if blk:
        layer.BLK_MAX_PIXELS = 1 << 30
        assert layer.blk_ok(v0, v1, sub_view(Ev, 0, 81), flow=fv, f0_copy=cpy)
    else:
        assert layer.concat_ok(v0, v1, sub_view(Ev, 0, 81), flow=fv, f0_copy=cpy)
    layer._run(v0, v1, sub_view(Ev, 0, 81), flow=fv, flow_scale=flow_scale, f0_copy=cpy, concat=True,
               out_pad_writable=pad, blk=blk)
    torch.cuda.synchronize()
    return E, g0


@pytest.mark.parametrize("N,H,W,C,with_flow,copy,pad", [
    (8, 7, 16, 192, False, True, True), (8, 14, 32, 128, True, True, True), (2, 28, 64, 96, True, True, True),
    (2, 14, 32, 128, True, False, False), (3, 9, 21, 96, True, True, False), (1, 5, 3, 192, True, False, True),
    (1, 17, 10, 128, False, True, False), (2, 8, 8, 192, True, True, True), (1, 1, 1, 128, True, True, True),
    (1, 4, 4, 96, False, False, True), (1, 13, 30, 192, True, True, True), (1, 56, 128, 64, True, True, True),
    (2, 6, 9, 64, True, False, False), (1, 28, 64, 96, True, True, True), (8, 28, 64, 96, True, True, True)])
def test_block_cost_volume_kernel_vs_oracle(pa, N, H, W, C, with_flow, copy, pad):
    """pwc_warp_cost_volume_concat_blk_f32 (cost_volume_blk.hip, the small pyramid levels on the F16 matrix pipe): the three
    coarsest levels of BASELINE configs[1] as they are, ragged blocks (H % 4, W % 4 != 0), images smaller than a block and
    than the search window, every C, flows with far outliers, every combination of the optional parts.  Nothing outside the
    declared slices may change."""
    f0, f1 = rnd((N, H, W, C), 61), rnd((N, H, W, C), 62)
    flow = util.flow_field(N, H, W, seed=63) / 5.0
    f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0) if with_flow else f1
    exp = orc.cost_volume(f0, f1w, 4)
    ecs = 84 + C + 8
    E, g0 = _run_concat(pa, f0, f1, flow if with_flow else None, 5.0, ecs, copy, pad, blk=True)
    close(E[..., :81], exp, rel=4e-6, floor=4e-7)
    if copy:
        assert torch.equal(E[..., 84:84 + C], g0)
    else:
        assert float(E[..., 84:84 + C].max()) == -3.0 and float(E[..., 84:84 + C].min()) == -3.0
    if pad:
        assert float(E[..., 81:84].abs().max()) == 0.0
    else:
        assert float(E[..., 81:84].min()) == -3.0 and float(E[..., 81:84].max()) == -3.0
    assert float(E[..., 84 + C:].min()) == -3.0 and float(E[..., 84 + C:].max()) == -3.0


def test_block_cost_volume_error_and_range(pa):
    """The block kernel against a float64 cost volume: not further from it than the fp32 coarse kernel is (features of
    1e-3 .. 300 in magnitude); a feature beyond fp16's range gives NaN where it is read."""
    from pwcnet_amd.modules import View
    N, H, W, C = 2, 14, 32, 128
    for scale in (1e-3, 1.0, 300.0):
        f0, f1 = rnd((N, H, W, C), 201) * scale, rnd((N, H, W, C), 202) * scale
        flow = util.flow_field(N, H, W, seed=203) / 5.0
        f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0).astype(np.float64)
        pad1 = np.zeros((N, H + 8, W + 8, C))
        pad1[:, 4:-4, 4:-4] = f1w
        ref = np.stack([(f0.astype(np.float64) * pad1[:, 4 + v:4 + v + H, 4 + h:4 + h + W]).mean(axis=3)
                        for v in range(-4, 5) for h in range(-4, 5)], axis=3)
        ref = np.maximum(ref, 0.1 * ref)
        E, _ = _run_concat(pa, f0, f1, flow, 5.0, 224, False, True, fill=0.0, blk=True)
        err_blk = float(np.abs(E[..., :81].double().cpu().numpy() - ref).max())
        g0, g1, fl = gpu(f0), gpu(f1), gpu(flow)
        out = torch.zeros((N, H, W, 81), device="cuda")
        layer = pa.CostVolumeLayer(4)
        layer._run(View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C),
                   View(out.data_ptr(), 81, N, H, W, 81), flow=View(fl.data_ptr(), 2, N, H, W, 2), flow_scale=5.0, coarse=True)
        err_f32 = float(np.abs(out.double().cpu().numpy() - ref).max())
        assert err_blk <= 1.25 * err_f32 + 1e-12 * scale * scale, (scale, err_blk, err_f32)
    f0, f1 = rnd((1, 8, 16, 96), 211), rnd((1, 8, 16, 96), 212)
    f1[0, 3, 5, 7] = 70000.0
    E, _ = _run_concat(pa, f0, f1, None, 1.0, 192, False, True, fill=0.0, blk=True)
    cv = E[0, ..., :81]
    assert bool(torch.isnan(cv).any())
    # only entries that read pixel (3, 5) of f1: P pixel (y, x) with |y - 3| <= 4, |x - 5| <= 4
    bad = torch.isnan(cv).any(dim=2)
    ys, xs = torch.nonzero(bad, as_tuple=True)
    assert int((ys - 3).abs().max()) <= 4 and int((xs - 5).abs().max()) <= 4


@pytest.mark.parametrize("N,H,W,C,with_flow,copy,pad", [
    (2, 28, 64, 96, True, True, True), (1, 56, 128, 64, True, True, True), (2, 40, 48, 32, True, True, True),
    (3, 9, 21, 32, True, True, False), (1, 5, 3, 32, True, False, False), (1, 17, 10, 64, False, True, True),
    (2, 8, 8, 96, True, False, True), (1, 30, 60, 64, True, True, True), (1, 15, 30, 96, False, False, False),
    (1, 4, 16, 32, True, True, True), (2, 33, 17, 32, False, True, False)])
@pytest.mark.parametrize("f16x2", [True, False])
def test_concat_cost_volume_kernel_vs_oracle(pa, N, H, W, C, with_flow, copy, pad, f16x2):
    """pwc_warp_cost_volume_concat_f32 (correlation on the matrix pipe): warp + cost volume + f0 copy in one launch
    into channel slices of a wider buffer; ragged block rows / strips (H % 4, W % 16 != 0), single-block images,
    every supported C, flows with far outliers, every combination of the optional parts.  Nothing outside the
    declared slices may change."""
    f0, f1 = rnd((N, H, W, C), 61), rnd((N, H, W, C), 62)
    flow = util.flow_field(N, H, W, seed=63) / 5.0
    f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0) if with_flow else f1
    exp = orc.cost_volume(f0, f1w, 4)
    ecs = 84 + C + 8
    E, g0 = _run_concat(pa, f0, f1, flow if with_flow else None, 5.0, ecs, copy, pad, f16x2=f16x2)
    close(E[..., :81], exp, rel=4e-6, floor=4e-7)
    if copy:
        assert torch.equal(E[..., 84:84 + C], g0)
    else:
        assert float(E[..., 84:84 + C].max()) == -3.0 and float(E[..., 84:84 + C].min()) == -3.0
    if pad:
        assert float(E[..., 81:84].abs().max()) == 0.0
    else:
        assert float(E[..., 81:84].min()) == -3.0 and float(E[..., 81:84].max()) == -3.0
    assert float(E[..., 84 + C:].min()) == -3.0 and float(E[..., 84 + C:].max()) == -3.0


@pytest.mark.parametrize("N,H,W,C,ecs", [(2, 28, 64, 96, 84), (1, 56, 128, 64, 84), (2, 40, 48, 32, 84), (3, 9, 21, 32, 84),
                                         (1, 5, 3, 32, 100), (2, 33, 17, 64, 84), (1, 4, 16, 32, 84)])
def test_concat_cost_volume_flow_in_record(pa, N, H, W, C, ecs):
    """pwc_warp_cost_volume_concat_h2_f32 with out_pad_writable = 2 (round 6): the record of a pixel is [cv 81 | flow x, y | 0] --
    the [cv, ..., flows_up_prev] parts of the estimator's concat (reference modules.py:261-264) in one dense tensor of 84-channel
    records (ecs = 84: what the model runs; ragged block rows / strips; far outliers in the flow).  The cost volume is the
    oracle's, the flow is the bits that were read, nothing behind channel 83 changes; the other two kernels refuse the mode."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    f0, f1 = rnd((N, H, W, C), 61), rnd((N, H, W, C), 62)
    flow = util.flow_field(N, H, W, seed=63) / 5.0
    exp = orc.cost_volume(f0, orc.warp(f1, flow, "bilinear", flow_scale=5.0), 4)
    for flow_cs in (2, 4):
        E, _ = _run_concat(pa, f0, f1, flow, 5.0, ecs, False, 2, flow_cs=flow_cs)
        close(E[..., :81], exp, rel=4e-6, floor=4e-7)
        assert torch.equal(E[..., 81:83], gpu(flow))
        assert float(E[..., 83].abs().max()) == 0.0
        if ecs > 84:
            assert float(E[..., 84:].min()) == -3.0 and float(E[..., 84:].max()) == -3.0
    g0, g1, fl = gpu(f0), gpu(f1), gpu(flow)
    out = torch.zeros((N, H, W, 84), device="cuda")
    args = (_p(g0), C, _p(g1), C, _p(fl), 2, 5.0, _p(out), 84, 2, None, 0, N, H, W, C, 4, 0.1, None)
    assert L.pwc_warp_cost_volume_concat_f32(*args) == -4
    assert L.pwc_warp_cost_volume_concat_blk_f32(*args) == -4
    noflow = (_p(g0), C, _p(g1), C, None, 0, 1.0, _p(out), 84, 2, None, 0, N, H, W, C, 4, 0.1, None)
    assert L.pwc_warp_cost_volume_concat_h2_f32(*noflow) == -1


@pytest.mark.parametrize("N,H,W,with_flow,copy,pad", [
    (9, 110, 250, True, True, True), (9, 110, 250, True, False, False), (10, 106, 244, False, True, True),
    (10, 106, 244, False, False, False), (9, 110, 250, True, False, 2), (8, 112, 256, True, False, 2), (16, 57, 255, True, True, 1)])
def test_big_ragged_c32_cost_volume_vs_oracle(pa, N, H, W, with_flow, copy, pad):
    """The 1/4-resolution level's launch (C = 32) at full-size pixel counts with ragged right / bottom edges (W % 16, H % 4, an odd
    number of block rows), every combination of warp / f0 copy / padding / the flow in the record (pad = 2), flows with far
    outliers -- the shapes round 6's tile-form experiment (scripts/experiments/cost_volume_tile.hip, profiles/
    r06_exp_cost_volume_tile.txt) was checked on, kept as tests of the entry point.  Same contract as
    test_concat_cost_volume_kernel_vs_oracle: the oracle's cost volume, the copy bit for bit, nothing outside the declared slices
    changes."""
    C = 32
    f0, f1 = rnd((N, H, W, C), 71), rnd((N, H, W, C), 72)
    flow = util.flow_field(N, H, W, seed=73) / 5.0
    f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0) if with_flow else f1
    exp = orc.cost_volume(f0, f1w, 4)
    ecs = 84 if pad == 2 else 84 + C + 8
    E, g0 = _run_concat(pa, f0, f1, flow if with_flow else None, 5.0, ecs, copy, pad)
    close(E[..., :81], exp, rel=4e-6, floor=4e-7)
    if pad == 2:
        assert torch.equal(E[..., 81:83], gpu(flow)) and float(E[..., 83].abs().max()) == 0.0
        return
    if copy:
        assert torch.equal(E[..., 84:84 + C], g0)
    else:
        assert float(E[..., 84:84 + C].max()) == -3.0 and float(E[..., 84:84 + C].min()) == -3.0
    if pad:
        assert float(E[..., 81:84].abs().max()) == 0.0
    else:
        assert float(E[..., 81:84].min()) == -3.0 and float(E[..., 81:84].max()) == -3.0
    assert float(E[..., 84 + C:].min()) == -3.0 and float(E[..., 84 + C:].max()) == -3.0


@pytest.mark.parametrize("f16x2", [True, False])
def test_concat_cost_volume_known_answers(pa, f16x2):
    """Zero flow = plain cost volume; a constant integer flow = the cost volume of the shifted map (edge
    replication from the warp's clipping, zeros outside from the cost volume's padding); the centre channel is
    lrelu(mean_c f0 * f1w)."""
    N, H, W, C = 1, 24, 40, 32
    f0, f1 = rnd((N, H, W, C), 71), rnd((N, H, W, C), 72)
    zero = np.zeros((N, H, W, 2), np.float32)
    E, _ = _run_concat(pa, f0, f1, zero, 5.0, 128, False, True, f16x2=f16x2)
    close(E[..., :81], orc.cost_volume(f0, f1, 4), rel=4e-6, floor=4e-7)
    flow = zero.copy(); flow[..., 0], flow[..., 1] = 2.0 / 5.0, -3.0 / 5.0
    shifted = f1[:, np.clip(np.arange(H) - 3, 0, H - 1)][:, :, np.clip(np.arange(W) + 2, 0, W - 1)]
    E, _ = _run_concat(pa, f0, f1, flow, 5.0, 128, False, True, f16x2=f16x2)
    close(E[..., :81], orc.cost_volume(f0, shifted, 4), rel=4e-6, floor=4e-7)
    centre = (f0 * shifted).mean(axis=3)
    centre = np.maximum(centre, 0.1 * centre)
    close(E[..., 40], centre, rel=4e-6, floor=4e-7)


@pytest.mark.parametrize("f16x2", [True, False])
def test_concat_cost_volume_full_size_vs_separate_launches(pa, f16x2):
    """BASELINE configs[1] level-4 geometry (8 x 112 x 256 x 32, estimator channel stride 160), flows ~ N(0, 3^2)
    px with outliers: the one-launch kernel against warp + cost volume as separate launches, every entry."""
    from pwcnet_amd.modules import View, sub_view
    N, H, W, C = 8, 112, 256, 32
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    f0 = torch.randn((N, H, W, C), generator=g, device="cuda")
    f1 = torch.randn((N, H, W, C), generator=g, device="cuda")
    fl = torch.randn((N, H, W, 2), generator=g, device="cuda") * (3.0 / 5.0)
    fl[0, 0, 0, 0], fl[0, 0, 0, 1], fl[7, 111, 255, 0] = 60.0, -60.0, -45.0
    E = torch.zeros((N, H, W, 160), device="cuda")
    Ev = View(E.data_ptr(), 160, N, H, W, 160)
    v0, v1 = View(f0.data_ptr(), C, N, H, W, C), View(f1.data_ptr(), C, N, H, W, C)
    fv = View(fl.data_ptr(), 2, N, H, W, 2)
    layer = pa.CostVolumeLayer(4)
    layer.f16x2 = f16x2
    layer._run(v0, v1, sub_view(Ev, 0, 81), flow=fv, flow_scale=5.0, f0_copy=sub_view(Ev, 84, C), concat=True,
               out_pad_writable=True)
    f1w = pa.WarpingLayer("bilinear")(f1, fl * 5.0)
    ref = layer(f0, f1w)
    assert float((E[..., :81] - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert torch.equal(E[..., 84:116], f0) and float(E[..., 81:84].abs().max()) == 0.0
    assert float(E[..., 116:].abs().max()) == 0.0


@pytest.mark.parametrize("f16x2", [True, False])
def test_concat_cost_volume_full_size_vs_oracle(pa, f16x2):
    """the one-launch kernel at BASELINE configs[1]'s level-4 geometry (8 x 112 x 256 x 32) against
    orc.cost_volume(orc.warp(...)) DIRECTLY (not against other HIP launches), images 0 and 7, every entry; flows ~ N(0, 3^2)
    px with far outliers."""
    N, H, W, C = 8, 112, 256, 32
    f0, f1 = rnd((N, H, W, C), 91), rnd((N, H, W, C), 92)
    flow = (np.random.RandomState(93).randn(N, H, W, 2) * (3.0 / 5.0)).astype(np.float32)
    flow[0, 0, 0], flow[7, 111, 255, 0], flow[7, 50, 100] = (60.0, -60.0), -45.0, (11.3, 7.7)
    E, g0 = _run_concat(pa, f0, f1, flow, 5.0, 160, True, True, fill=0.0, f16x2=f16x2)
    for i in (0, N - 1):
        f1w = orc.warp(f1[i:i + 1], flow[i:i + 1], "bilinear", flow_scale=5.0)
        close(E[i:i + 1, ..., :81], orc.cost_volume(f0[i:i + 1], f1w, 4), rel=4e-6, floor=4e-7)
    assert torch.equal(E[..., 84:84 + C], g0) and float(E[..., 81:84].abs().max()) == 0.0
    assert float(E[..., 84 + C:].abs().max()) == 0.0 and bool(torch.isfinite(E).all())


def test_concat_cost_volume_rejects_what_it_does_not_support(pa):
    from pwcnet_amd import _lib
    L = _lib.lib()
    assert L.pwc_warp_cost_volume_concat_supported(28, 64, 96, 4, 96, 96, 2, 224, 224) == 1
    assert L.pwc_warp_cost_volume_concat_supported(28, 64, 128, 4, 128, 128, 2, 224, 224) == 0     # C
    assert L.pwc_warp_cost_volume_concat_supported(28, 64, 32, 3, 32, 32, 2, 160, 160) == 0        # search range
    assert L.pwc_warp_cost_volume_concat_supported(28, 64, 32, 4, 32, 32, 2, 81, 0) == 0           # out_cs % 4
    x = torch.zeros((1, 8, 8, 128), device="cuda")
    out = torch.zeros((1, 8, 8, 84), device="cuda")
    rc = L.pwc_warp_cost_volume_concat_f32(_p(x), 128, _p(x), 128, None, 0, 1.0, _p(out), 84, 1, None, 0,
                                           1, 8, 8, 128, 4, 0.1, None)
    assert rc == -4
    y = torch.zeros((1, 8, 8, 32), device="cuda")
    rc = L.pwc_warp_cost_volume_concat_f32(_p(y), 32, _p(y), 32, None, 0, 1.0, _p(out), 84, 1, None, 0,
                                           1, 8, 8, 32, 2, 0.1, None)
    assert rc == -4


def test_concat_cost_volume_f16x2_error_and_status(pa):
    """The F16-pipe correlation against a float64 cost volume: not further from it than the fp32 matrix-pipe kernel is
    (features of 1e-3 .. 300 in magnitude); a feature beyond fp16's range gives NaN where it is read (never a wrong number)."""
    from pwcnet_amd import _lib
    from pwcnet_amd.modules import View, sub_view
    N, H, W, C = 2, 24, 48, 64
    for scale in (1e-3, 1.0, 300.0):
        f0, f1 = rnd((N, H, W, C), 201) * scale, rnd((N, H, W, C), 202) * scale
        flow = util.flow_field(N, H, W, seed=203) / 5.0
        f1w = orc.warp(f1, flow, "bilinear", flow_scale=5.0).astype(np.float64)
        pad1 = np.zeros((N, H + 8, W + 8, C))
        pad1[:, 4:-4, 4:-4] = f1w
        ref = np.stack([(f0.astype(np.float64) * pad1[:, 4 + v:4 + v + H, 4 + h:4 + h + W]).mean(axis=3)
                        for v in range(-4, 5) for h in range(-4, 5)], axis=3)
        ref = np.maximum(ref, 0.1 * ref)
        errs = {}
        for f16x2 in (True, False):
            E, _ = _run_concat(pa, f0, f1, flow, 5.0, 160, False, True, fill=0.0, f16x2=f16x2)
            errs[f16x2] = float(np.abs(E[..., :81].double().cpu().numpy() - ref).max())
        assert errs[True] <= 1.25 * errs[False] + 1e-12 * scale * scale, (scale, errs)
    # range
    f0, f1 = rnd((1, 16, 32, 32), 211), rnd((1, 16, 32, 32), 212)
    layer = pa.CostVolumeLayer(4)
    for big in (None, 70000.0):
        g0, g1 = gpu(f0), gpu(f1)
        if big is not None:
            g1[0, 7, 9, 3] = big
        E = torch.zeros((1, 16, 32, 128), device="cuda")
        Ev = View(E.data_ptr(), 128, 1, 16, 32, 128)
        layer._run(View(g0.data_ptr(), 32, 1, 16, 32, 32), View(g1.data_ptr(), 32, 1, 16, 32, 32), sub_view(Ev, 0, 81),
                   concat=True, out_pad_writable=True)
        torch.cuda.synchronize()
        nan = torch.isnan(E[..., :81])
        assert bool(torch.isfinite(E[..., :81]).all()) == (big is None)
        if big is not None:       # exactly the 81 (pixel, displacement) pairs that meet pixel (7, 9) of f1
            assert int(nan.sum()) == 81 and bool(nan[0, 3:12, 5:14].any(dim=2).all())


def test_coarse_cost_volume_rejects_other_search_ranges(pa):
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = torch.zeros((1, 8, 8, 32), device="cuda")
    out = torch.zeros((1, 8, 8, 81), device="cuda")
    rc = L.pwc_cost_volume_coarse_f32(_p(x), 32, _p(x), 32, None, 0, 1.0, _p(out), 81, None, 0, 1, 8, 8, 32, 3, 0.1, None)
    assert rc == -4


# ------------------------------------------------------------------ resize / copy
@pytest.mark.parametrize("N,H,W,C,OH,OW,mul", [
    (2, 7, 16, 2, 14, 32, 1.0), (2, 7, 16, 32, 14, 32, 1.0), (1, 112, 256, 2, 448, 1024, 20.0),
    (1, 5, 6, 3, 10, 12, 1.0), (1, 4, 4, 8, 4, 4, 1.0), (1, 6, 10, 736, 12, 20, 1.0)])
def test_resize_vs_oracle(pa, N, H, W, C, OH, OW, mul):
    x = rnd((N, H, W, C), 34)
    close(pa.resize_bilinear(gpu(x), (OH, OW), mul), orc.resize_bilinear(x, (OH, OW), mul), rel=1e-6, floor=1e-6)


@pytest.mark.parametrize("N,H,W,x_cs", [(8, 112, 256, 2), (3, 5, 7, 2), (1, 1, 1, 2), (2, 9, 13, 6), (1, 240, 480, 2)])
def test_final_flow_upsampling_x4_vs_oracle(pa, N, H, W, x_cs):
    """Round 6: the forward's last launch -- x4 up-sampling of the 2-channel flows, times 20 (reference model.py:125-127) -- on
    the one-thread-per-source-cell kernel (resize_x4_c2_kernel): production shapes of configs[1] / configs[4], odd sizes, a
    single pixel, a source that is a channel slice of a wider tensor; the status variant flags a non-finite output and nothing
    else; the same values as the general kernel (a destination with a channel stride takes that one)."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    x = rnd((N, H, W, x_cs), 341) * 3.0
    xg = gpu(x)
    exp = orc.resize_bilinear(np.ascontiguousarray(x[..., :2]), (4 * H, 4 * W), 20.0)
    y = torch.full((N, 4 * H, 4 * W, 2), -5.0, device="cuda")
    st = torch.zeros(2, dtype=torch.int32, device="cuda")
    _lib.check(L.pwc_resize_bilinear_status_f32(_p(xg), x_cs, _p(y), 2, N, H, W, 2, 4 * H, 4 * W, 20.0, _p(st), None))
    torch.cuda.synchronize()
    close(y, exp, rel=1e-6, floor=1e-6)
    assert int(st[0]) == 0
    y2 = torch.empty_like(y)
    _lib.check(L.pwc_resize_bilinear_f32(_p(xg), x_cs, _p(y2), 2, N, H, W, 2, 4 * H, 4 * W, 20.0, None))
    wide = torch.full((N, 4 * H, 4 * W, 4), -5.0, device="cuda")          # (channel stride 4: the general kernel)
    _lib.check(L.pwc_resize_bilinear_f32(_p(xg), x_cs, _p(wide), 4, N, H, W, 2, 4 * H, 4 * W, 20.0, None))
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(wide[..., :2], y) and float(wide[..., 2:].max()) == -5.0
    xg[N - 1, H - 1, W // 2, 1] = float("inf")
    _lib.check(L.pwc_resize_bilinear_status_f32(_p(xg), x_cs, _p(y), 2, N, H, W, 2, 4 * H, 4 * W, 20.0, _p(st), None))
    torch.cuda.synchronize()
    assert int(st[0]) & _lib.STATUS_NONFINITE


@pytest.mark.parametrize("N,H,W,C", [(2, 7, 16, 32), (1, 14, 32, 32), (1, 5, 9, 8), (2, 6, 10, 288), (1, 1, 1, 64), (2, 9, 13, 736),
                                     (1, 56, 128, 128)])
def test_resize_pair_vs_oracle(pa, N, H, W, C):
    """flows (2 ch) + features (C ch) of one level resized x2 in ONE launch into channel slices of
    the next level's buffer (modules.py:283-284)."""
    from pwcnet_amd.modules import View, _resize_pair, sub_view
    fl, ft = rnd((N, H, W, 2), 36), rnd((N, H, W, C), 37)
    gfl, gft = gpu(fl), gpu(ft)
    E = torch.full((N, 2 * H, 2 * W, 8 + C + 4), -2.0, device="cuda")       # [pad 4 | flow 2 pad 2 | feat C | pad 4]
    Ev = View(E.data_ptr(), 8 + C + 4, N, 2 * H, 2 * W, 8 + C + 4)
    _resize_pair(View(gfl.data_ptr(), 2, N, H, W, 2), sub_view(Ev, 4, 2), View(gft.data_ptr(), C, N, H, W, C), sub_view(Ev, 8, C))
    torch.cuda.synchronize()
    close(E[..., 4:6], orc.resize_bilinear(fl, (2 * H, 2 * W)), rel=1e-6, floor=1e-6)
    close(E[..., 8:8 + C], orc.resize_bilinear(ft, (2 * H, 2 * W)), rel=1e-6, floor=1e-6)
    assert float(E[..., :4].max()) == -2.0 and float(E[..., 6:8].max()) == -2.0 and float(E[..., 8 + C:].max()) == -2.0


def test_resize_known_answer_and_golden(pa, golden_dir):
    x = np.arange(4, dtype=np.float32).reshape(1, 1, 4, 1)
    y = pa.resize_bilinear(gpu(x), (1, 8)).cpu().numpy().ravel()
    np.testing.assert_array_equal(y, [0, .5, 1, 1.5, 2, 2.5, 3, 3])
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    close(pa.resize_bilinear(gpu(g["rs_x"]), (12, 20)), g["rs_x2"], rel=1e-6, floor=1e-6)


def test_copy_channels(pa):
    from pwcnet_amd.modules import View, _copy_channels, sub_view
    src = gpu(rnd((2, 5, 7, 33), 35))
    dst = torch.zeros(2, 5, 7, 48, device="cuda")
    _copy_channels(View(src.data_ptr(), 33, 2, 5, 7, 33), sub_view(View(dst.data_ptr(), 48, 2, 5, 7, 48), 3, 33), 33)
    torch.cuda.synchronize()
    assert torch.equal(dst[..., 3:36], src) and float(dst[..., :3].abs().max()) == 0 and float(dst[..., 36:].abs().max()) == 0
    src4 = gpu(rnd((2, 5, 7, 32), 36))
    _copy_channels(View(src4.data_ptr(), 32, 2, 5, 7, 32), sub_view(View(dst.data_ptr(), 48, 2, 5, 7, 48), 8, 32), 32)
    torch.cuda.synchronize()
    assert torch.equal(dst[..., 8:40], src4)


# ------------------------------------------------------------------ losses (forward)
@pytest.mark.parametrize("N,H,W", [(2, 7, 16), (3, 64, 128), (1, 448, 1024), (4, 5, 3)])
def test_losses_vs_oracle(pa, N, H, W):
    """pwcnet_amd.losses (reference losses.py:4-48, forward values) against the oracle."""
    from pwcnet_amd import losses
    a, b = rnd((N, H, W, 2), 61) * 3.0, rnd((N, H, W, 2), 62) * 3.0
    ga, gb = gpu(a), gpu(b)
    assert float(losses.L1loss(ga, gb)) == pytest.approx(orc.L1loss(a, b), rel=2e-5)
    assert float(losses.L2loss(ga, gb)) == pytest.approx(orc.L2loss(a, b), rel=2e-5)
    assert float(losses.EPE(ga, gb)) == pytest.approx(orc.epe(a, b), rel=2e-5)
    # strided views (a flow stored in a wider buffer)
    wide = torch.full((N, H, W, 6), 7.0, device="cuda")
    wide[..., 2:4] = gb
    assert float(losses.EPE(ga, wide[..., 2:4])) == pytest.approx(orc.epe(a, b), rel=2e-5)


def test_multiscale_losses_vs_oracle(pa):
    from pwcnet_amd import losses
    gt = rnd((2, 64, 128, 2), 63) * 40.0
    pyr = [rnd((2, 64 // 2 ** (6 - l), 128 // 2 ** (6 - l), 2), 64 + l) for l in range(5)]
    wts = [0.32, 0.08, 0.02, 0.01, 0.005]
    g_gt, g_pyr = gpu(gt), [gpu(p) for p in pyr]
    assert float(losses.multiscale_loss(g_gt, g_pyr, wts)) == pytest.approx(orc.multiscale_loss(gt, pyr, wts), rel=2e-5)
    assert float(losses.multirobust_loss(g_gt, g_pyr, wts, 0.01, 0.4)) == pytest.approx(
        orc.multirobust_loss(gt, pyr, wts, 0.01, 0.4), rel=2e-5)
    # a non-integer downsampling ratio exercises floor(dst * in/out)
    odd = [rnd((2, 3, 5, 2), 70)]
    assert float(losses.multiscale_loss(g_gt, [gpu(odd[0])], [1.0])) == pytest.approx(orc.multiscale_loss(gt, odd, [1.0]), rel=2e-5)


# ------------------------------------------------------------------ seeded random sweeps
def _sweep_cases(seed, n):
    rs = np.random.RandomState(seed)
    return [tuple(int(v) for v in (rs.randint(1, 4), rs.randint(1, 70), rs.randint(1, 150))) for _ in range(n)]


@pytest.mark.parametrize("case", range(24))
def test_conv_winograd_random_sweep(pa, case):
    """Seeded random shapes through every geometry of the Winograd kernel (16x16 / split / 4x64
    blocks, 16 or 32 couts per workgroup, dilation 1..16, ragged edges) against the oracle."""
    rs = np.random.RandomState(1000 + case)
    N = int(rs.randint(1, 4))
    dil = int(rs.choice([1, 1, 1, 2, 2, 3, 4, 8, 16]))
    H, W = int(rs.randint(1, 40) * (1 if dil < 4 else 3)), int(rs.randint(1, 90) * (1 if dil < 4 else 2))
    cin, cout = int(rs.choice([16, 32, 48, 64, 96])), int(rs.choice([16, 32, 48, 64, 96, 128]))
    x = rnd((N, H, W, cin), 2000 + case)
    k = rnd((3, 3, cin, cout), 3000 + case) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 4000 + case) * 0.1
    slope = None if case % 5 == 0 else 0.1
    close(run_conv_wino(x, k, b, slope, dil=dil), orc.conv3x3(x, k, b, 1, dil, slope), rel=2e-5)


@pytest.mark.parametrize("case", range(16))
def test_conv_mfma_random_sweep(pa, case):
    """Seeded random shapes / strides / dilations through the automatic plan of the direct MFMA kernel."""
    rs = np.random.RandomState(5000 + case)
    N, H, W = int(rs.randint(1, 4)), int(rs.randint(2, 60)), int(rs.randint(2, 120))
    stride = int(rs.choice([1, 2]))
    if stride == 2:
        H, W = 2 * (H // 2 + 1), 2 * (W // 2 + 1)           # the reference only sees even sizes under stride 2
    dil = 1 if stride == 2 else int(rs.choice([1, 2, 4]))
    cin, cout = int(rs.choice([16, 32, 64, 96, 160])), int(rs.choice([16, 32, 64, 96, 128, 192]))
    x = rnd((N, H, W, cin), 6000 + case)
    k = rnd((3, 3, cin, cout), 7000 + case) * float(1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 8000 + case) * 0.1
    close(run_conv_mfma(x, k, b, stride, dil, 0.1), orc.conv3x3(x, k, b, stride, dil, 0.1), rel=2e-5)


@pytest.mark.parametrize("case", range(12))
def test_cost_volume_warp_resize_random_sweep(pa, case):
    """Seeded random shapes through the cost-volume (streaming and coarse), warp and resize kernels."""
    rs = np.random.RandomState(9000 + case)
    N, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 40)), int(rs.randint(1, 100))
    C = int(rs.choice([4, 16, 32, 64, 96]))
    f0, f1 = rnd((N, H, W, C), 9100 + case), rnd((N, H, W, C), 9200 + case)
    flow = util.flow_field(N, H, W, seed=9300 + case) / 5.0
    close(pa.CostVolumeLayer(4)(gpu(f0), gpu(f1)), orc.cost_volume(f0, f1, 4), rel=4e-6, floor=4e-7)
    for wt in ("bilinear", "nearest"):
        close(pa.WarpingLayer(wt)(gpu(f1), gpu(flow * 5.0)), orc.warp(f1, flow * 5.0, wt), rel=2e-6, floor=1e-6)
    oh, ow = int(rs.randint(1, 3 * H + 1)), int(rs.randint(1, 3 * W + 1))
    close(pa.resize_bilinear(gpu(f0), (oh, ow)), orc.resize_bilinear(f0, (oh, ow)), rel=1e-6, floor=1e-6)
    if C % 32 == 0 or True:
        from pwcnet_amd.modules import View
        out = torch.empty((N, H, W, 81), device="cuda")
        g0, g1, gf = gpu(f0), gpu(f1), gpu(flow)
        pa.CostVolumeLayer(4)._run(View(g0.data_ptr(), C, N, H, W, C), View(g1.data_ptr(), C, N, H, W, C),
                                   View(out.data_ptr(), 81, N, H, W, 81), flow=View(gf.data_ptr(), 2, N, H, W, 2),
                                   flow_scale=5.0, coarse=N * H * W <= 4096)
        torch.cuda.synchronize()
        close(out, orc.cost_volume(f0, orc.warp(f1, flow, "bilinear", flow_scale=5.0), 4), rel=4e-6, floor=4e-7)
