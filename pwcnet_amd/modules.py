"""Host-side mirror of the reference's modules.py op classes on top of libpwc_hip.so.

Same class names, constructor kwargs, call signatures and NHWC float32 tensor
conventions as reference modules.py:42-326 (FeaturePyramidExtractor_custom,
WarpingLayer, CostVolumeLayer, OpticalFlowEstimator_custom, ContextNetwork); torch-ROCm
tensors are only memory handles (data_ptr + strides) for the HIP kernels.

Variables follow TensorFlow's lazy-creation model: a module creates (or re-uses, by
name) its ``<scope>/conv2d[_k]/{kernel,bias}`` variables in a VariableStore on first
call, numbered per call like tf.layers does inside a re-entered variable scope -- this
is what makes both pyramid extractions share one weight set (reference model.py:97-98).

Every public ``__call__`` accepts/returns ordinary logical tensors.  The ``_run*``
methods are the zero-copy forms PWCDCNet composes: producers write straight into the
channel slices of the consumer's buffer (ChannelLayout), so tf.concat costs nothing.
"""
import collections
import contextlib
import math

import numpy as np
import torch

from . import _lib
from .profiler import active as active_timer
from .profiler import timed
from .weights import FILTERS_FP, FILTERS_OF, CONTEXT, ChannelLayout

View = collections.namedtuple("View", "ptr cs N H W C")


# ---------------------------------------------------------------- tensor plumbing

def _check_tensor(t, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.PwcHipError(f"{what}: tensor is on {t.device}; pwcnet_amd runs on the GPU only")
    if t.dtype != torch.float32 or t.dim() != 4:
        raise ValueError(f"{what}: expected a 4-D float32 NHWC tensor, got {t.dtype} {tuple(t.shape)}")


def _pixel_dense(t):
    N, H, W, C = t.shape
    s = t.stride()
    cs = s[2]
    return (s[3] == 1 or C == 1) and cs >= C and s[1] == W * cs and s[0] == H * W * cs


def as_view(t, what="tensor"):
    """(ptr, channel stride) of an NHWC tensor whose pixels are laid out densely
    (a channel slice of a wider buffer qualifies); anything else is made contiguous."""
    _check_tensor(t, what)
    if not _pixel_dense(t):
        t = t.contiguous()
    N, H, W, C = t.shape
    return View(t.data_ptr(), t.stride(2), N, H, W, C), t


def sub_view(v, off, C):
    return View(v.ptr + 4 * off, v.cs, v.N, v.H, v.W, C)


def _same_out(size, stride):
    return -(-size // stride)


def _p(ptr):
    return _lib.ctypes.c_void_p(ptr)


# ---------------------------------------------------------------- launches / launch plans
class LaunchPlan:
    """A recorded forward: the exact sequence of C-ABI calls (function + prepared
    arguments) plus every tensor they touch.  Replaying it costs one ctypes call per
    kernel -- the Python bookkeeping of the eager path (views, layouts, caches,
    allocations) runs once per input shape."""

    def __init__(self):
        self.calls = []      # [fn, args(list), what, kernel name, flops, bytes, executed flops]
        self.keep = []       # tensors whose memory the recorded pointers reference
        self.patch = {}      # input name -> [(call index, arg index)]
        self.outputs = None

    def replay(self, inputs=None):
        for name, ptr in (inputs or {}).items():
            for ci, ai in self.patch.get(name, ()):
                self.calls[ci][1][ai] = _p(ptr)
        timer = active_timer()
        for fn, args, what, kname, flops, nbytes, xflops in self.calls:
            if timer is not None and kname is not None and timer.wants(kname):
                with timer.launch(kname, flops, nbytes, xflops):
                    rc = fn(*args)
            else:
                rc = fn(*args)
            if rc:
                _lib.check(rc, what)


_RECORDER = None


def _launch(fn, args, what, kname=None, flops=0.0, nbytes=0.0, exec_flops=None):
    """Enqueue one library call on torch's current stream (timed when an OpTimer is
    active, recorded when a LaunchPlan is being built).  flops / nbytes: ALGORITHMIC work
    of the launch; exec_flops: what the MFMA units execute when that
    differs (Winograd)."""
    with timed(kname, flops, nbytes, exec_flops):
        rc = fn(*args)
    _lib.check(rc, what)
    if _RECORDER is not None:
        _RECORDER.calls.append([fn, list(args), what, kname, flops, nbytes, exec_flops])


def _keep(*tensors):
    if _RECORDER is not None:
        _RECORDER.keep.extend(t for t in tensors if t is not None)


# ---------------------------------------------------------------- variables

class Variable:
    """A named model variable (mirrors the bits of tf.Variable the reference touches:
    `.name`, `.shape`)."""

    def __init__(self, name, value):
        self.name = name + ":0"
        self.key = name
        self.value = value

    @property
    def shape(self):
        return tuple(self.value.shape)

    def __repr__(self):
        return f"<Variable {self.name} shape={self.shape}>"


class VariableStore:
    """name -> Variable.  Creation follows tf.layers.Conv2D defaults: kernel
    glorot_uniform, bias zeros."""

    def __init__(self, seed=0, device="cuda"):
        self.vars = collections.OrderedDict()
        self.device = device
        self._rng = np.random.RandomState(seed)
        self.version = 0

    def get(self, name, shape, kind):
        v = self.vars.get(name)
        if v is not None:
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"variable {name} has shape {v.shape}, requested {tuple(shape)}")
            return v
        if kind == "kernel":
            limit = math.sqrt(6.0 / (9 * shape[2] + 9 * shape[3]))
            val = self._rng.uniform(-limit, limit, size=shape).astype(np.float32)
        else:
            val = np.zeros(shape, np.float32)
        v = Variable(name, torch.from_numpy(val).to(self.device))
        self.vars[name] = v
        return v

    def assign(self, name, value):
        t = torch.as_tensor(np.ascontiguousarray(value, dtype=np.float32)).to(self.device)
        if name in self.vars:
            if tuple(self.vars[name].shape) != tuple(t.shape):
                raise ValueError(f"assign {name}: shape {tuple(t.shape)} != {self.vars[name].shape}")
            self.vars[name].value = t
        else:
            self.vars[name] = Variable(name, t)
        self.version += 1


_DEFAULT_STORE = None
_SCOPE = []
_STORE_STACK = []


def default_store():
    global _DEFAULT_STORE
    if _STORE_STACK:
        return _STORE_STACK[-1]
    if _DEFAULT_STORE is None:
        _DEFAULT_STORE = VariableStore()
    return _DEFAULT_STORE


@contextlib.contextmanager
def variable_scope(name, store=None):
    """Minimal counterpart of tf.variable_scope: prefixes variable names, optionally
    selects the VariableStore used inside."""
    _SCOPE.append(name)
    if store is not None:
        _STORE_STACK.append(store)
    try:
        yield
    finally:
        _SCOPE.pop()
        if store is not None:
            _STORE_STACK.pop()


def _scoped(name):
    return "/".join(_SCOPE + [name])


# ---------------------------------------------------------------- conv layer

class _ConvRunner:
    """Runs the convs of one module call; numbers them conv2d, conv2d_1, ... like
    tf.layers does inside a (re-entered) variable scope."""

    def __init__(self, owner):
        self.owner = owner
        self.k = 0
        self.scope = _scoped(owner.name)
        self.store = default_store()

    def conv_pair(self, x, cout, slope=0.1):
        """Two consecutive stride-1 convs x -> cout -> cout (conv2d_k, conv2d_{k+1}), both + leaky_relu.  16 -> 16 -> 16 on
        a shape pwc_conv3x3_c16pair_supported names: ONE launch with the intermediate in LDS; otherwise two conv() calls.
        Returns (View y, tensor)."""
        L = _lib.lib()
        fused = (x.C == 16 and cout == 16 and slope is not None and getattr(self.owner, "f16x2", True)
                 and x.cs % 4 == 0 and x.ptr % 16 == 0 and L.pwc_conv3x3_c16pair_supported(x.N, x.H, x.W))
        if not fused:
            x, _ = self.conv(x, cout, slope=slope)
            return self.conv(x, cout, slope=slope)
        names = []
        for _ in range(2):
            names.append(self.scope + "/conv2d" + ("" if self.k == 0 else f"_{self.k}"))
            self.k += 1
        k1 = self.store.get(names[0] + "/kernel", (3, 3, 16, 16), "kernel")
        b1 = self.store.get(names[0] + "/bias", (16,), "bias")
        k2 = self.store.get(names[1] + "/kernel", (3, 3, 16, 16), "kernel")
        b2 = self.store.get(names[1] + "/bias", (16,), "bias")
        dev = k1.value.device
        s = _lib.current_stream()
        y_t = torch.empty((x.N, x.H, x.W, 16), dtype=torch.float32, device=dev)
        y = View(y_t.data_ptr(), 16, x.N, x.H, x.W, 16)
        cache = self.owner._cache
        key = (names[0], "c16pair", self.store.version)
        packed = cache.get(key)
        if packed is None:
            packed = torch.empty((L.pwc_conv3x3_c16pair_packed_floats(),), dtype=torch.float32, device=dev)
            _lib.check(L.pwc_conv3x3_c16pair_pack_f32(_p(k1.value.data_ptr()), _p(k2.value.data_ptr()), _p(packed.data_ptr()), s),
                       "conv3x3 c16pair pack")
            cache[key] = packed
        _keep(packed, y_t)
        M = x.N * x.H * x.W
        _track_max(self.owner, x)
        _launch(L.pwc_conv3x3_c16pair_f32,
                (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(b1.value.data_ptr()), _p(b2.value.data_ptr()), _p(y.ptr), y.cs,
                 x.N, x.H, x.W, float(slope), s),
                f"conv3x3_c16pair {names[0]}+{names[1]}", "conv3x3_c16pair_kernel",
                2.0 * 2.0 * M * 9 * 16 * 16, 4.0 * (M * 16 + M * 16),
                # executed: 14 MFMAs of 16x16x32 per 16 pixels and layer on 1.27 / 1.13 x the tile's pixels (halo, row wrap)
                exec_flops=2.0 * 14 * 16 * 16 * 32 * (41 + 36) * (M / 512.0))
        return y, y_t

    def conv_level1(self, image_views, slope=0.1):
        """ALL of pyramid level 1 from the raw frames: conv2d (3 -> 16, stride 2), conv2d_1, conv2d_2 (16 -> 16), leaky_relu
        behind each (reference modules.py:57-67), in ONE launch for one or two image batches that share the weights.
        Returns (View y over the stacked batch, tensor), or None where pwc_conv3x3_c3c16pair_supported says no (the caller
        then runs the three layers one by one)."""
        L = _lib.lib()
        v0 = image_views[0]
        n_tot = sum(v.N for v in image_views)
        ok = (self.k == 0 and len(image_views) <= 2 and slope is not None and getattr(self.owner, "f16x2", True)
              and all(v.C == 3 and v.cs == 3 and v.ptr % 16 == 0 and v.H == v0.H and v.W == v0.W for v in image_views)
              and L.pwc_conv3x3_c3c16pair_supported(n_tot, v0.H, v0.W))
        if not ok:
            return None
        names = [self.scope + "/conv2d" + ("" if k == 0 else f"_{k}") for k in range(3)]
        self.k = 3
        shapes = [(3, 3, 3, 16), (3, 3, 16, 16), (3, 3, 16, 16)]
        ks = [self.store.get(nm + "/kernel", sh, "kernel") for nm, sh in zip(names, shapes)]
        bs = [self.store.get(nm + "/bias", (16,), "bias") for nm in names]
        dev = ks[0].value.device
        s = _lib.current_stream()
        Ho, Wo = _same_out(v0.H, 2), _same_out(v0.W, 2)
        y_t = torch.empty((n_tot, Ho, Wo, 16), dtype=torch.float32, device=dev)
        y = View(y_t.data_ptr(), 16, n_tot, Ho, Wo, 16)
        cache = self.owner._cache
        key = (names[0], "c3c16pair", self.store.version)
        packed = cache.get(key)
        if packed is None:
            packed = torch.empty((L.pwc_conv3x3_c3c16pair_packed_floats(),), dtype=torch.float32, device=dev)
            _lib.check(L.pwc_conv3x3_c3c16pair_pack_f32(_p(ks[0].value.data_ptr()), _p(ks[1].value.data_ptr()),
                                                        _p(ks[2].value.data_ptr()), _p(packed.data_ptr()), s),
                       "conv3x3 c3c16pair pack")
            cache[key] = packed
        _keep(packed, y_t)
        va = image_views[0]
        vb = image_views[1] if len(image_views) == 2 else None
        M = n_tot * Ho * Wo
        for v in image_views:
            _track_max(self.owner, v)
        _launch(L.pwc_conv3x3_c3c16pair_f32,
                (_p(va.ptr), va.N, _p(vb.ptr) if vb is not None else None, vb.N if vb is not None else 0, _p(packed.data_ptr()),
                 _p(bs[0].value.data_ptr()), _p(bs[1].value.data_ptr()), _p(bs[2].value.data_ptr()), _p(y.ptr), y.cs,
                 v0.H, v0.W, float(slope), s),
                f"conv3x3_c3c16pair {names[0]}+{names[1]}+{names[2]}", "conv3x3_c16pair_kernel",
                2.0 * M * 9 * (3 * 16 + 2 * 16 * 16), 4.0 * (n_tot * v0.H * v0.W * 3 + M * 16),
                # executed: 3 MFMAs of 16x16x32 per 16 patch pixels (45 tiles) + 14 per 16 pixels and layer (41 + 36 tiles)
                exec_flops=2.0 * 16 * 16 * 32 * (3 * 45 + 14 * (41 + 36)) * (M / 512.0))
        return y, y_t

    def conv(self, x, cout, y=None, stride=1, dilation=1, slope=0.1, cin_map=None,
             cin_logical=None, residual=None, tile=-1, split=0, x2=None, x3=None):
        """x: View over the PHYSICAL input channels.  cin_map: physical->logical map (or
        None = identity).  x2 (round 5): a second View over the same pixels whose channels follow x's in the physical order
        (cin_map covers both) -- only the F16-pipe kernel takes it: the caller asks h2_two_operand_ok first.  x3 (round 6): a
        third one (pwc_conv3x3_h2_ex3_f32); a View's C is then its 16-channel stage count x 16 and may exceed its channel
        stride by up to 12 (the last stage runs into the next pixel's record; cin_map = -1 there).
        Returns (View y, tensor or None)."""
        name = self.scope + "/conv2d" + ("" if self.k == 0 else f"_{self.k}")
        self.k += 1
        cin = x.C if cin_logical is None else cin_logical
        kern = self.store.get(name + "/kernel", (3, 3, cin, cout), "kernel")
        bias = self.store.get(name + "/bias", (cout,), "bias")
        Ho, Wo = _same_out(x.H, stride), _same_out(x.W, stride)
        y_t = None
        if y is None:
            y_t = torch.empty((x.N, Ho, Wo, cout), dtype=torch.float32, device=kern.value.device)
            y = View(y_t.data_ptr(), cout, x.N, Ho, Wo, cout)
        assert (y.N, y.H, y.W, y.C) == (x.N, Ho, Wo, cout), (y, x, stride)
        L = _lib.lib()
        s = _lib.current_stream()
        act = 0 if slope is None else 1
        sl = 0.0 if slope is None else float(slope)
        assert x3 is None or x2 is not None
        c_phys = x.C + (x2.C if x2 is not None else 0) + (x3.C if x3 is not None else 0)
        use_mfma = (cout % 16 == 0 and x.C % 16 == 0 and x.cs % 4 == 0 and x.ptr % 16 == 0
                    and residual is None)
        cache = self.owner._cache
        use_wino = (use_mfma and getattr(self.owner, "winograd", False) and stride == 1
                    and tile < 0 and split == 0 and _wino_pays(L, x.N, x.H, x.W, cout, dilation))
        use_wino4 = (use_wino and getattr(self.owner, "winograd4", True) and y.cs % 4 == 0 and y.ptr % 16 == 0
                     and L.pwc_conv3x3_wino4_supported(x.N, x.H, x.W, x.C, cout, dilation))
        use_h2 = (use_mfma and getattr(self.owner, "f16x2", True) and stride == 1 and tile < 0 and split == 0
                  and (x3 is None or dilation == 1)
                  and cout % 32 == 0 and y.cs % 4 == 0 and y.ptr % 16 == 0
                  and L.pwc_conv3x3_h2_supported(x.N, x.H, x.W, c_phys, cout, dilation))
        # (experiment switch, scripts/exp_ab_model.py h2_relaxed: the F16-pipe kernel wherever it HAS a plan, also below the
        # tile count its _supported asks for)
        if (not use_h2 and getattr(self.owner, "h2_relaxed", False) and use_mfma and getattr(self.owner, "f16x2", True)
                and stride == 1 and dilation == 1 and tile < 0 and split == 0 and cout % 32 == 0 and c_phys >= 32
                and y.cs % 4 == 0 and y.ptr % 16 == 0 and x.H >= 7 and x.W >= 24
                and L.pwc_conv3x3_h2_plan(x.N, x.H, x.W, c_phys, cout, 1) != 0):
            use_h2 = True
        # small launches (round 5): the K dimension dealt to the waves of one workgroup, ONE dispatch (conv3x3_sk.hip)
        use_sk = (use_mfma and getattr(self.owner, "f16x2", True) and getattr(self.owner, "small_conv", True) and x2 is None
                  and tile < 0 and split == 0 and x.C % 32 == 0 and y.cs % 4 == 0 and y.ptr % 16 == 0
                  and L.pwc_conv3x3_sk_supported(x.N, x.H, x.W, x.C, cout, stride, dilation))
        # thin inputs to 32 channels (round 5): weights stationary in registers (conv3x3_t32.hip)
        use_t32 = (use_mfma and getattr(self.owner, "f16x2", True) and getattr(self.owner, "thin_conv", True) and x2 is None
                   and tile < 0 and split == 0 and dilation == 1 and cout == 32 and y.cs % 4 == 0 and y.ptr % 16 == 0
                   and (x.C == 16 or getattr(self.owner, "thin_conv32", True))
                   and x.N * Ho * Wo * y.cs * 4 < (1 << 31)      # (32-bit store offsets: ADVICE r5)
                   and L.pwc_conv3x3_t32_supported(x.N, x.H, x.W, x.C, cout, stride))
        if x2 is not None and not use_h2:
            raise _lib.PwcHipError(f"{name}: a two-operand input needs the F16-pipe kernel (h2_two_operand_ok)")
        # 32 output channels from 32 / 64 input channels (round 6): the weights resident in the LDS (conv3x3_w32.hip)
        use_w32 = (use_mfma and getattr(self.owner, "f16x2", True) and getattr(self.owner, "w32_conv", True) and x2 is None
                   and tile < 0 and split == 0 and stride == 1 and dilation == 1 and cout == 32 and x.C in (32, 64)
                   and y.cs % 4 == 0 and y.ptr % 16 == 0 and x.N * Ho * Wo * y.cs * 4 < (1 << 31)
                   and L.pwc_conv3x3_w32_supported(x.N, x.H, x.W, x.C, cout, stride, dilation))
        if use_w32:
            key = (name, "w32", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                packed = torch.empty((L.pwc_conv3x3_w32_packed_floats(x.C),), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_w32_pack_f32(_p(kern.value.data_ptr()), _p(cm.data_ptr()) if cm is not None else None,
                                                      cin, x.C, _p(packed.data_ptr()), s), "conv3x3 w32 pack")
                cache[key] = packed
            _keep(packed, y_t)
            _track_max(self.owner, x)
            _launch(L.pwc_conv3x3_w32_f32,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, act, sl, s),
                    f"conv3x3_w32 {name}", "conv3x3_w32_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    exec_flops=3.0 * 2.0 * x.N * Ho * Wo * 9 * x.C * cout)
            return y, y_t
        if use_t32:
            key = (name, "t32", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                packed = torch.empty((L.pwc_conv3x3_t32_packed_floats(x.C),), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_t32_pack_f32(_p(kern.value.data_ptr()), _p(cm.data_ptr()) if cm is not None else None,
                                                      cin, x.C, _p(packed.data_ptr()), s), "conv3x3 t32 pack")
                cache[key] = packed
            _keep(packed, y_t)
            _track_max(self.owner, x)
            _launch(L.pwc_conv3x3_t32_f32,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, stride, act, sl, s),
                    f"conv3x3_t32 {name}", "conv3x3_t32_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    exec_flops=3.0 * 2.0 * x.N * Ho * Wo * 9 * x.C * cout)
            return y, y_t
        if use_sk:
            key = (name, "sk", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                packed = torch.empty((L.pwc_conv3x3_sk_packed_floats(x.C, cout),), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_sk_pack_f32(_p(kern.value.data_ptr()), _p(cm.data_ptr()) if cm is not None else None,
                                                     cin, x.C, cout, _p(packed.data_ptr()), s), "conv3x3 sk pack")
                cache[key] = packed
            _keep(packed, y_t)
            _track_max(self.owner, x)
            _launch(L.pwc_conv3x3_sk_f32,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, stride, dilation, act, sl, s),
                    f"conv3x3_sk {name}", "conv3x3_sk_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    exec_flops=3.0 * 2.0 * x.N * Ho * Wo * 9 * x.C * cout)
            return y, y_t
        use_h2s2 = (use_mfma and getattr(self.owner, "f16x2", True) and stride == 2 and dilation == 1 and tile < 0 and split == 0
                    and cout % 32 == 0 and y.cs % 4 == 0 and y.ptr % 16 == 0
                    and L.pwc_conv3x3_h2_stride2_supported(x.N, x.H, x.W, x.C, cout))
        if use_h2 or use_h2s2:
            # direct convolution on the F16 matrix pipe, fp32 operands as two-term fp16 splits (conv3x3_h2.hip); stride 2 = the
            # same kernel over the input's four parity planes (round 5)
            key = (name, "h2s2" if use_h2s2 else "h2", c_phys, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                nfl = (L.pwc_conv3x3_h2_stride2_packed_floats if use_h2s2 else L.pwc_conv3x3_h2_packed_floats)(c_phys, cout)
                packed = torch.empty((nfl,), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == c_phys
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check((L.pwc_conv3x3_h2_stride2_pack_f32 if use_h2s2 else L.pwc_conv3x3_h2_pack_f32)(
                    _p(kern.value.data_ptr()), _p(cm.data_ptr()) if cm is not None else None,
                    cin, c_phys, cout, _p(packed.data_ptr()), s), "conv3x3 h2 pack")
                cache[key] = packed
            _keep(packed, y_t)
            # more tiles than CUs: one workgroup per CU with an equal share of the (tile, stage) sequence (stream-K)
            wsf = (L.pwc_conv3x3_h2_stride2_workspace_floats(x.N, x.H, x.W, c_phys, cout) if use_h2s2 else
                   L.pwc_conv3x3_h2_workspace_floats(x.N, x.H, x.W, c_phys, cout, dilation))
            ws = _h2_workspace(kern.value.device, wsf) if wsf and getattr(self.owner, "f16x2_stream_k", True) else None
            if ws is not None:
                _keep(ws)
            wsa = (_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0)
            status = getattr(self.owner, "status", None)      # the model's status words (stream-K timeout)
            if status is not None:
                _keep(status)
            for v in (x, x2, x3):
                if v is not None:
                    _track_max(self.owner, v._replace(C=min(v.C, v.cs)))
            if use_h2 and x3 is not None:
                fn = L.pwc_conv3x3_h2_ex3_f32
                args = (_p(x.ptr), x.cs, x.C, _p(x2.ptr), x2.cs, x2.C, _p(x3.ptr), x3.cs,
                        _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                        x.N, x.H, x.W, c_phys, cout, act, sl) + wsa + (_p(status.data_ptr()) if status is not None else None, s)
            elif use_h2:
                fn = L.pwc_conv3x3_h2_ex_f32
                args = (_p(x.ptr), x.cs, x.C if x2 is not None else 0, _p(x2.ptr) if x2 is not None else None,
                        x2.cs if x2 is not None else 0, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                        x.N, x.H, x.W, c_phys, cout, dilation, act, sl) + wsa + (_p(status.data_ptr()) if status is not None else None, s)
            else:
                fn = L.pwc_conv3x3_h2_stride2_f32
                args = (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                        x.N, x.H, x.W, x.C, cout, act, sl) + wsa + (_p(status.data_ptr()) if status is not None else None, s)
            _launch(fn, args, f"conv3x3_h2 {name}", "conv3x3_h2_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    # executed: three fp16 products per multiply-add, per physical input channel (stride 2: 16 products per
                    # output and channel -- four taps on each of the four parity planes)
                    exec_flops=3.0 * 2.0 * x.N * Ho * Wo * (16 if use_h2s2 else 9) * c_phys * cout)
        elif use_wino4:
            # F(4x4,3x3): 36 multiplies per 4x4 outputs (the big full-resolution layers)
            key = (name, "wino4", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                nfl = L.pwc_conv3x3_wino4_packed_floats(x.C, cout)
                packed = torch.empty((nfl,), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_wino4_pack_f32(_p(kern.value.data_ptr()),
                                                        _p(cm.data_ptr()) if cm is not None else None,
                                                        cin, x.C, cout, _p(packed.data_ptr()), s), "conv3x3 wino4 pack")
                cache[key] = packed
            _keep(packed, y_t)
            _launch(L.pwc_conv3x3_wino4_f32,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, dilation, act, sl, s),
                    f"conv3x3_wino4 {name}", "conv3x3_wino4_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    # executed: 36 multiplies per 4x4 output tile (per sub-lattice of a dilated conv), per physical input channel
                    exec_flops=2.0 * x.N * dilation * dilation * (-(-(-(-Ho // dilation)) // 4)) * (-(-(-(-Wo // dilation)) // 4))
                    * 36 * x.C * cout)
        elif use_wino:
            key = (name, "wino", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                nfl = L.pwc_conv3x3_wino_packed_floats(x.C, cout)
                packed = torch.empty((nfl,), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_wino_pack_f32(_p(kern.value.data_ptr()),
                                                       _p(cm.data_ptr()) if cm is not None else None,
                                                       cin, x.C, cout, _p(packed.data_ptr()), s), "conv3x3 wino pack")
                cache[key] = packed
            _keep(packed, y_t)
            # launches that would leave most workgroup slots empty deal their channel stages to several workgroups
            csplit = (L.pwc_conv3x3_wino_split_plan(x.N, x.H, x.W, x.C, cout, dilation)
                      if getattr(self.owner, "wino_channel_split", True) else 1)
            if csplit > 1:
                ws = _workspace(kern.value.device, L.pwc_conv3x3_wino_split_workspace_floats(x.N, x.H, x.W, cout, csplit))
                _keep(ws)
                fn, extra = L.pwc_conv3x3_wino_split_f32, (csplit, _p(ws.data_ptr()), ws.numel(), s)
            else:
                fn, extra = L.pwc_conv3x3_wino_f32, (s,)
            _launch(fn,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, dilation, act, sl) + extra,
                    f"conv3x3_wino {name}", "conv3x3_wino_kernel",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout),
                    # executed: 16 multiplies per 2x2 output tile, per physical input channel
                    exec_flops=2.0 * x.N * (-(-Ho // 2)) * (-(-Wo // 2)) * 16 * x.C * cout)
        elif use_mfma:
            key = (name, "mfma", x.C, None if cin_map is None else cin_map.tobytes(), self.store.version)
            packed = cache.get(key)
            if packed is None:
                nfl = L.pwc_conv3x3_packed_floats(x.C, cout)
                packed = torch.empty((nfl,), dtype=torch.float32, device=kern.value.device)
                cm = None
                if cin_map is not None:
                    assert len(cin_map) == x.C
                    cm = torch.from_numpy(np.ascontiguousarray(cin_map, np.int32)).to(kern.value.device)
                _lib.check(L.pwc_conv3x3_pack_f32(_p(kern.value.data_ptr()),
                                                  _p(cm.data_ptr()) if cm is not None else None,
                                                  cin, x.C, cout, _p(packed.data_ptr()), s), "conv3x3 pack")
                cache[key] = packed
                cache[key + ("cm",)] = cm
            flops = 2.0 * x.N * Ho * Wo * 9 * cin * cout     # algorithmic: logical Cin, no padding
            ws = _workspace(kern.value.device, L.pwc_conv3x3_workspace_floats(x.N * Ho * Wo, cout))
            _keep(ws, packed, y_t)
            _launch(L.pwc_conv3x3_f32,
                    (_p(x.ptr), x.cs, _p(packed.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs,
                     x.N, x.H, x.W, x.C, cout, stride, dilation, act, sl, tile, split, _p(ws.data_ptr()),
                     ws.numel(), s),
                    f"conv3x3 {name}",
                    (f"conv3x3_halo_kernel<{x.C},{cout}>"
                     if tile < 0 and split == 0 and L.pwc_conv3x3_uses_halo_kernel(x.N * Ho * Wo, x.C, cout, stride, dilation)
                     else _mfma_kernel_name(L, x.N * Ho * Wo, cout, x.C, tile, split)), flops,
                    4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout))
        else:
            w = kern.value
            if cin_map is not None:
                key = (name, "direct", x.C, cin_map.tobytes(), self.store.version)
                w = cache.get(key)
                if w is None:
                    idx = torch.from_numpy(np.ascontiguousarray(cin_map, np.int64)).to(kern.value.device)
                    w = torch.zeros((3, 3, x.C, cout), dtype=torch.float32, device=kern.value.device)
                    valid = idx >= 0
                    w[:, :, valid, :] = kern.value[:, :, idx[valid], :]
                    w = w.contiguous()
                    cache[key] = w
            elif x.C != cin:
                raise ValueError(f"{name}: input has {x.C} channels, kernel expects {cin}")
            r_ptr, r_cs = (None, 0) if residual is None else (_p(residual.ptr), residual.cs)
            _keep(w, y_t)
            _launch(L.pwc_conv3x3_direct_f32,
                    (_p(x.ptr), x.cs, _p(w.data_ptr()), _p(bias.value.data_ptr()), _p(y.ptr), y.cs, r_ptr, r_cs,
                     x.N, x.H, x.W, x.C, cout, stride, dilation, act, sl, s),
                    f"conv3x3_direct {name}", f"conv3x3_direct_kernel<cout={cout}>",
                    2.0 * x.N * Ho * Wo * 9 * cin * cout, 4.0 * (x.N * x.H * x.W * cin + x.N * Ho * Wo * cout))
        return y, y_t


def _track_max(owner, v):
    """PWCDCNet(track_max=True): the largest |value| of an operand of an F16-pipe kernel goes into the model's status words
    (pwc_absmax_f32) -- a debugging aid, one small launch per operand."""
    if getattr(owner, "track_max", False) and getattr(owner, "status", None) is not None:
        _keep(owner.status)
        _launch(_lib.lib().pwc_absmax_f32, (_p(v.ptr), v.cs, v.N * v.H * v.W, v.C, _p(owner.status.data_ptr()), _lib.current_stream()),
                "absmax", "absmax_kernel", 0.0, 4.0 * v.N * v.H * v.W * v.C)


_WS = {}
# tap-split scratch is only ever used for small outputs; cap what is kept around
_WS_CAP_FLOATS = 64 << 20


def _workspace(device, want_floats):
    """Caller-owned scratch for the conv tap split (grown on demand, one per device AND
    stream: micro-batches running on different streams must not share it)."""
    want = int(min(want_floats, _WS_CAP_FLOATS))
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < want:
        ws = torch.empty((max(want, 1 << 20),), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


_H2_WS = collections.OrderedDict()
_H2_WS_MAX = 8


def _h2_workspace(device, want_floats):
    """Caller-owned workspace of pwc_conv3x3_h2_f32's stream-K form: one per device AND stream (launches that may run
    concurrently must not share it), every byte 0xFF when created (= "nothing published"; the kernel leaves it so)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _H2_WS.pop(key, None)
    if ws is None or ws.numel() < want_floats:
        ws = torch.full((int(want_floats),), -1, dtype=torch.int32, device=device).view(torch.float32)
    _H2_WS[key] = ws                                  # (most recently used last)
    while len(_H2_WS) > _H2_WS_MAX:                   # callers that come with a new stream every time: up to 32 MB each
        _H2_WS.pop(next(iter(_H2_WS)))
    return ws


def h2_workspaces_refill():
    """Every stream-K workspace back to "nothing published" (0xFF bytes) -- after a launch reported
    PWC_STATUS_STREAMK_TIMEOUT a late publisher may have left sums in a slot.  The workspaces belong to
    different streams: every device that owns one is synchronised first, so that no launch in flight on another stream
    is using a workspace while it is refilled (a rare path: a timeout means a fault)."""
    for dev in {ws.device for ws in _H2_WS.values()}:
        torch.cuda.synchronize(dev)
    for ws in _H2_WS.values():
        ws.view(torch.int32).fill_(-1)
    for dev in {ws.device for ws in _H2_WS.values()}:
        torch.cuda.synchronize(dev)


def _wino_pays(L, N, H, W, cout, dilation):
    """Measured on MI355X (scripts/tune_conv.py --wino): the Winograd kernel wins (1.1-2.2x)
    from the 14x32 pyramid level upwards, i.e. whenever its 16x16-pixel blocks are reasonably
    filled (the 7x16 level fills 44 % and loses to the split-K direct kernel); a dilation-d
    launch works on (H/d) x (W/d) sub-lattices."""
    hs, wsub = -(-H // dilation), -(-W // dilation)
    # block geometries of conv3x3_wino.hip (wino_geo): two short sub-lattices per 16x16 block,
    # 4x64-pixel blocks, 16x16-pixel blocks
    if hs <= 8 and (dilation * dilation) % 2 == 0:
        area = 8 * (-(-wsub // 16) * 16)
    else:
        sq = (-(-hs // 16) * 16) * (-(-wsub // 16) * 16)
        wd = (-(-hs // 4) * 4) * (-(-wsub // 64) * 64)
        area = wd if wd < 0.95 * sq else sq
    fill = (hs * wsub) / float(area)
    return L.pwc_conv3x3_wino_workgroups(N, H, W, cout, dilation) >= 24 and fill >= 0.6


def _mfma_kernel_name(L, M, cout, cin_phys, tile, split):
    plan = (_lib.ctypes.c_int * 4)()
    if tile >= 0:
        plan[0], plan[1], plan[3] = tile, -1, max(split, 1)
    else:
        L.pwc_conv3x3_plan(M, cout, cin_phys, plan)
    bm, bn = _lib.ctypes.c_int(), _lib.ctypes.c_int()
    L.pwc_conv3x3_tile_shape(plan[0], bm, bn)
    name = f"conv3x3_mfma_kernel<{bm.value}x{bn.value},KC{32 if cin_phys % 32 == 0 else 16}>"
    if plan[1] >= 0:
        L.pwc_conv3x3_tile_shape(plan[1], bm, bn)
        name += f"+tail<{bm.value}x{bn.value}>"
    if plan[3] > 1:
        name += f"+split{plan[3]}"
    return name


class _Module:
    winograd = False     # route eligible convs (stride 1, dilation 1, Cout % 32 == 0) to the Winograd kernel
    winograd4 = True     # ... and the big ones among them to the F(4x4,3x3) kernel (pwc_conv3x3_wino4_supported)
    wino_channel_split = True    # F(2x2) launches that leave most workgroup slots empty deal their channel stages to several workgroups
    f16x2_stream_k = True    # ... with one workgroup per CU and an equal share of the work each where a launch has more tiles than CUs
    f16x2 = True         # the layers pwc_conv3x3_h2_supported names go to the direct F16-matrix-pipe kernel (fp32 operands as
                         # exact-to-22-bit fp16 pairs, fp32 accumulation; inputs must stay below 65504)

    def __init__(self, name):
        self.name = name
        self._cache = {}


def _copy_channels(src, dst, C):
    """dst[..., 0:C] = src[..., 0:C] (both Views over the same pixel grid)."""
    assert (src.N, src.H, src.W) == (dst.N, dst.H, dst.W)
    npix = src.N * src.H * src.W
    _launch(_lib.lib().pwc_copy_channels_f32,
            (_p(src.ptr), src.cs, _p(dst.ptr), dst.cs, npix, C, _lib.current_stream()),
            "copy_channels", "copy_channels_kernel", 0.0, 8.0 * npix * C)


def _resize(src, dst, mul=1.0, status=None):
    """status: the model's status words -- the launch then reports a non-finite output (pwc_resize_bilinear_status_f32)."""
    if status is not None:
        _keep(status)
        _launch(_lib.lib().pwc_resize_bilinear_status_f32,
                (_p(src.ptr), src.cs, _p(dst.ptr), dst.cs, src.N, src.H, src.W, src.C, dst.H, dst.W, float(mul),
                 _p(status.data_ptr()), _lib.current_stream()),
                "resize_bilinear", "resize_kernel", 0.0, 4.0 * src.C * src.N * (src.H * src.W + dst.H * dst.W))
        return
    _launch(_lib.lib().pwc_resize_bilinear_f32,
            (_p(src.ptr), src.cs, _p(dst.ptr), dst.cs, src.N, src.H, src.W, src.C, dst.H, dst.W, float(mul),
             _lib.current_stream()),
            "resize_bilinear", "resize_kernel", 0.0, 4.0 * src.C * src.N * (src.H * src.W + dst.H * dst.W))


def _resize_pair(src_a, dst_a, src_b, dst_b):
    """x2-style resize of a 2-channel map and a C-channel map of the same geometry in one launch."""
    assert src_a.C == 2 and (src_a.N, src_a.H, src_a.W) == (src_b.N, src_b.H, src_b.W)
    _launch(_lib.lib().pwc_resize_bilinear_pair_f32,
            (_p(src_a.ptr), src_a.cs, _p(dst_a.ptr), dst_a.cs, _p(src_b.ptr), src_b.cs, _p(dst_b.ptr), dst_b.cs,
             src_a.N, src_a.H, src_a.W, src_b.C, dst_a.H, dst_a.W, _lib.current_stream()),
            "resize_bilinear_pair", "resize_kernel", 0.0,
            4.0 * (2 + src_b.C) * src_a.N * (src_a.H * src_a.W + dst_a.H * dst_a.W))


def resize_bilinear(x, size, mul=1.0):
    """tf.image.resize_bilinear(x, size) * mul with TF-1.8 legacy sampling (reference
    modules.py:283-284, model.py:127)."""
    xv, x = as_view(x, "resize_bilinear input")
    out = torch.empty((xv.N, int(size[0]), int(size[1]), xv.C), dtype=torch.float32, device=x.device)
    _resize(xv, View(out.data_ptr(), xv.C, xv.N, int(size[0]), int(size[1]), xv.C), mul)
    return out


# ---------------------------------------------------------------- extractor (a6)

class FeaturePyramidExtractor_custom(_Module):
    """Feature pyramid extractor (reference modules.py:42-71): per level a stride-2 conv
    and two stride-1 convs, all 3x3 + leaky_relu(0.1); returns deep -> shallow."""

    def __init__(self, num_levels=6, name="fp_extractor"):
        super().__init__(name)
        self.num_levels = num_levels
        self.filters = list(FILTERS_FP)

    def _run(self, image_views, device):
        """image_views: Views of one or two image batches that share the weights; they
        are convolved into ONE stacked batch (first layer per input, the remaining 17
        layers once on the stack).  Returns the list of stacked feature tensors,
        shallow -> deep."""
        run = _ConvRunner(self)
        v0 = image_views[0]
        n_tot = sum(v.N for v in image_views)
        feats = []
        x = None
        for l in range(self.num_levels):
            f = self.filters[l]
            if l == 0 and f == 16:
                fused = run.conv_level1(image_views)
                if fused is not None:
                    x, t2 = fused
                    feats.append(t2)
                    continue
            if l == 0:
                Ho, Wo = _same_out(v0.H, 2), _same_out(v0.W, 2)
                y_t = torch.empty((n_tot, Ho, Wo, f), dtype=torch.float32, device=device)
                _keep(y_t)
                n_off = 0
                for iv in image_views:
                    yv = View(y_t.data_ptr() + 4 * n_off * Ho * Wo * f, f, iv.N, Ho, Wo, f)
                    run.k = 0
                    run.conv(iv, f, y=yv, stride=2)
                    n_off += iv.N
                x = View(y_t.data_ptr(), f, n_tot, Ho, Wo, f)
            else:
                x, y_t = run.conv(x, f, stride=2)
            x, t2 = run.conv_pair(x, f)
            feats.append(t2)
        return feats

    def __call__(self, images, reuse=True):
        iv, images = as_view(images, "images")
        feats = self._run([iv], images.device)
        return feats[::-1]


# ---------------------------------------------------------------- warping (a2 / a3)

class WarpingLayer(_Module):
    """Backward warping by a per-pixel flow (reference modules.py:140-154)."""

    def __init__(self, warp_type="nearest", name="warping"):
        super().__init__(name)
        self.warp = warp_type

    def _run(self, x, flow, out, flow_scale=1.0, copy=None):
        """copy = (src view, dst view): also copies src's channels into dst in the same launch (the
        features_0 part of the estimator input's concat, modules.py:264)."""
        assert self.warp in ["nearest", "bilinear"]
        L = _lib.lib()
        nbytes = 4.0 * x.N * x.H * x.W * (2 * x.C + 2)
        if copy is not None:
            src, dst = copy
            assert (src.N, src.H, src.W) == (x.N, x.H, x.W) and dst.C == src.C
            _launch(L.pwc_warp_copy_f32,
                    (1 if self.warp == "bilinear" else 0, _p(x.ptr), x.cs, _p(flow.ptr), flow.cs, float(flow_scale),
                     _p(out.ptr), out.cs, x.N, x.H, x.W, x.C, _p(src.ptr), src.cs, _p(dst.ptr), dst.cs, src.C,
                     _lib.current_stream()),
                    # algorithmic bytes of the WARP only; the concat copy riding along is not credited
                    f"warp_{self.warp}+copy", f"warp_kernel<{self.warp}>", 0.0, nbytes)
            return
        fn = L.pwc_warp_bilinear_f32 if self.warp == "bilinear" else L.pwc_warp_nearest_f32
        _launch(fn, (_p(x.ptr), x.cs, _p(flow.ptr), flow.cs, float(flow_scale), _p(out.ptr), out.cs,
                     x.N, x.H, x.W, x.C, _lib.current_stream()),
                f"warp_{self.warp}", f"warp_kernel<{self.warp}>", 0.0, nbytes)

    def __call__(self, x, flow):
        assert self.warp in ["nearest", "bilinear"]
        xv, x = as_view(x, "x")
        fv, flow = as_view(flow, "flow")
        assert (fv.N, fv.H, fv.W, fv.C) == (xv.N, xv.H, xv.W, 2), "flow must be (N,H,W,2)"
        out = torch.empty((xv.N, xv.H, xv.W, xv.C), dtype=torch.float32, device=x.device)
        self._run(xv, fv, View(out.data_ptr(), xv.C, xv.N, xv.H, xv.W, xv.C))
        return out


# ---------------------------------------------------------------- cost volume (a1)

class CostVolumeLayer(_Module):
    """Cost volume (reference modules.py:185-204): (2R+1)^2 shifted channel-mean dot
    products, vertical shift outer / horizontal inner, leaky_relu(0.1)."""

    def __init__(self, search_range=4, name="cost_volume"):
        super().__init__(name)
        self.s_range = search_range

    # pixels per batch up to which the coarse-level kernel (warp + cost volume + f0 copy in one
    # launch) is used.  Measured at batch 8: 7x16 level 11.5 us vs 48 us for the three separate
    # launches, 14x32 level 21 vs 44 us, 28x64 level 41 vs 41 us (no gain: the streaming kernels stay)
    COARSE_MAX_PIXELS = 4096

    def coarse_ok(self, f0):
        return self.s_range == 4 and f0.N * f0.H * f0.W <= self.COARSE_MAX_PIXELS

    def concat_ok(self, f0, f1, out, flow=None, f0_copy=None):
        """True if pwc_warp_cost_volume_concat_f32 (matrix-pipe kernel: warp + cost volume + f0 concat copy in one
        launch) takes this geometry: search range 4, C in {32, 64, 96}, 16-byte aligned operands."""
        L = _lib.lib()
        if any(v.ptr % 16 for v in (f0, f1, out)) or (f0_copy is not None and f0_copy.ptr % 16) or \
                (flow is not None and flow.ptr % 4):
            return False
        return bool(L.pwc_warp_cost_volume_concat_supported(f0.H, f0.W, f0.C, self.s_range, f0.cs, f1.cs,
                                                            flow.cs if flow is not None else 0, out.cs,
                                                            f0_copy.cs if f0_copy is not None else 0))

    # pixels per batch up to which a C = 64 / 96 level goes to the block-per-workgroup kernel (its window is re-gathered by up
    # to nine workgroups: past this the row-walking kernel, which reads less, is faster.  Measured, graph replays, us per launch,
    # blk / row-walking: 28x64x96 batch 1: 6.8 / 15.8, batch 4: 10.9 / 16.1, batch 8: 19.5 / 16.4; 56x128x64 batch 1: 8.2 / 11.7,
    # batch 2: 13.6 / 11.9 -- profiles/r05_exp_blk_ab.txt)
    BLK_MAX_PIXELS = 8192

    def blk_ok(self, f0, f1, out, flow=None, f0_copy=None):
        """True if pwc_warp_cost_volume_concat_blk_f32 (F16 matrix pipe, one 4 x 4 block per workgroup: the small pyramid
        levels) takes this geometry and is the faster launch for it."""
        if not getattr(self, "f16x2", True) or f0.C not in (64, 96, 128, 192):
            return False
        if f0.C <= 96 and f0.N * f0.H * f0.W > self.BLK_MAX_PIXELS:
            return False
        if any(v.ptr % 16 for v in (f0, f1, out)) or (f0_copy is not None and f0_copy.ptr % 16) or \
                (flow is not None and flow.ptr % 4):
            return False
        return bool(_lib.lib().pwc_warp_cost_volume_concat_blk_supported(
            f0.H, f0.W, f0.C, self.s_range, f0.cs, f1.cs, flow.cs if flow is not None else 0, out.cs,
            f0_copy.cs if f0_copy is not None else 0))

    def _run(self, f0, f1, out, flow=None, flow_scale=1.0, f0_copy=None, coarse=False, concat=False, out_pad_writable=False,
             blk=False):
        """flow given: f1 is the UN-warped map and the bilinear warp is fused in.
        coarse: one launch of pwc_cost_volume_coarse_f32 (optionally also copying f0 into
        the `f0_copy` view, the features_0 slice of the estimator input).
        concat: one launch of pwc_warp_cost_volume_concat_f32 (same operands; out_pad_writable: channels 81..83 of
        `out` are padding the kernel may zero).  blk (with concat): pwc_warp_cost_volume_concat_blk_f32, the F16-pipe
        kernel of the small levels."""
        L = _lib.lib()
        s = _lib.current_stream()
        D = (2 * self.s_range + 1) ** 2
        npix = f0.N * f0.H * f0.W
        flops = 2.0 * npix * D * f0.C
        if concat:
            h2 = getattr(self, "f16x2", True)        # correlation on the F16 matrix pipe (two-term operand splits), round 5
            if h2:
                _track_max(self, f0)
                _track_max(self, f1)
            args = (_p(f0.ptr), f0.cs, _p(f1.ptr), f1.cs, _p(flow.ptr) if flow is not None else None,
                    flow.cs if flow is not None else 0, float(flow_scale), _p(out.ptr), out.cs,
                    int(out_pad_writable),      # (2, round 6: the flow rides in channels 81, 82 of the record)
                    _p(f0_copy.ptr) if f0_copy is not None else None, f0_copy.cs if f0_copy is not None else 0,
                    f0.N, f0.H, f0.W, f0.C, self.s_range, 0.1)
            assert h2 or not blk
            fn = L.pwc_warp_cost_volume_concat_blk_f32 if blk else \
                L.pwc_warp_cost_volume_concat_h2_f32 if h2 else L.pwc_warp_cost_volume_concat_f32
            _launch(fn, args + (s,),
                    "warp_cost_volume_concat" if flow is not None else "cost_volume_concat",
                    f"cost_volume_{'blk' if blk else 'h2' if h2 else 'mfma'}_kernel<C{f0.C}{',warp' if flow is not None else ''}>", flops,
                    # (2C+81) or fused (2C+2+81) bytes per pixel; the f0 concat copy is not credited
                    4.0 * npix * (2 * f0.C + D + (2 if flow is not None else 0)))
            return
        if coarse:
            assert self.coarse_ok(f0)
            _launch(L.pwc_cost_volume_coarse_f32,
                    (_p(f0.ptr), f0.cs, _p(f1.ptr), f1.cs, _p(flow.ptr) if flow is not None else None,
                     flow.cs if flow is not None else 0, float(flow_scale), _p(out.ptr), out.cs,
                     _p(f0_copy.ptr) if f0_copy is not None else None, f0_copy.cs if f0_copy is not None else 0,
                     f0.N, f0.H, f0.W, f0.C, self.s_range, 0.1, s),
                    "cost_volume_coarse", "cost_volume_coarse_kernel", flops,
                    # (2C+81) or fused (2C+2+81) bytes per pixel; the f0 concat copy is not credited
                    4.0 * npix * (2 * f0.C + D + (2 if flow is not None else 0)))
            return
        assert f0_copy is None
        if flow is None:
            _launch(L.pwc_cost_volume_f32,
                    (_p(f0.ptr), f0.cs, _p(f1.ptr), f1.cs, _p(out.ptr), out.cs, f0.N, f0.H, f0.W, f0.C,
                     self.s_range, 0.1, s),
                    "cost_volume",
                    ("cost_volume_roll_kernel" if L.pwc_cost_volume_uses_rolling_kernel(f0.H, f0.W, f0.C, self.s_range, f0.cs, f1.cs,
                                                                                      out.cs) and out.ptr % 16 == 0
                     else f"cost_volume_dma_kernel<R{self.s_range}>"), flops, 4.0 * npix * (2 * f0.C + D))
        else:
            _launch(L.pwc_warp_cost_volume_f32,
                    (_p(f0.ptr), f0.cs, _p(f1.ptr), f1.cs, _p(flow.ptr), flow.cs, float(flow_scale), _p(out.ptr),
                     out.cs, f0.N, f0.H, f0.W, f0.C, self.s_range, 0.1, s),
                    "warp_cost_volume", f"cost_volume_kernel<R{self.s_range},fused_warp>", flops,
                    4.0 * npix * (2 * f0.C + 2 + D))

    def __call__(self, features_0, features_0from1):
        f0, features_0 = as_view(features_0, "features_0")
        f1, features_0from1 = as_view(features_0from1, "features_0from1")
        assert f0[2:] == f1[2:], "feature maps must have equal shapes"
        D = (2 * self.s_range + 1) ** 2
        out = torch.empty((f0.N, f0.H, f0.W, D), dtype=torch.float32, device=features_0.device)
        self._run(f0, f1, View(out.data_ptr(), D, f0.N, f0.H, f0.W, D))
        return out


# ---------------------------------------------------------------- flow estimator (a4)

class OpticalFlowEstimator_custom(_Module):
    """Optical flow estimator (reference modules.py:227-285)."""

    def __init__(self, use_dc=False, name="of_estimator"):
        super().__init__(name)
        self.filters = list(FILTERS_OF)
        self.use_dc = use_dc

    # -- layout of the buffer holding the (growing) `features` tensor
    def _layout(self, c_cv, c_f0, has_flow, fu_map, f0_external=False):
        """f0_external (round 5, non-DC): features_0 keeps its place in the concat order (reference modules.py:261-264) but
        stays in the pyramid tensor -- the first conv reads it through a second operand pointer (two_operand_ok)."""
        lay = ChannelLayout()
        if self.use_dc:
            for k in reversed(range(len(self.filters))):
                lay.add(f"conv{k}", self.filters[k])
        lay.add("cv", c_cv)
        if c_f0 and f0_external:
            assert not self.use_dc
            lay.reserve("f0", c_f0)
        elif c_f0:
            lay.add("f0", c_f0)
        if has_flow:
            lay.add("flow", 2)
        if fu_map is not None:
            lay.add("feat_up", len(fu_map), log_map=list(fu_map))
        return lay.finish(16)

    def two_operand_ok(self, N, h, w, c_cv, c_f0, has_flow, fu_map):
        """True where the first conv of this (non-DC) estimator can read features_0 from the pyramid tensor: its launch goes
        to the F16-pipe kernel (the one with a second operand pointer) at the channel count the two-operand layout gives."""
        if self.use_dc or not getattr(self, "f16x2", True) or not c_f0 or c_f0 % 16:
            return False
        lay = self._layout(c_cv, c_f0, has_flow, fu_map, f0_external=True)
        return bool(_lib.lib().pwc_conv3x3_h2_supported(N, h, w, lay.n_phys + c_f0, self.filters[0], 1))

    CVX_CS = 84          # channel stride of the dense [cv 81 | flow 2 | 0] tensor (336-byte records)

    def three_operand_map(self, c_cv, c_f0, c_fu):
        """Round 6: the first conv's input as THREE dense tensors -- [cv | flows_up_prev | 0] (84-channel records, read as six
        16-channel stages: the last one runs 12 channels into the next record, zero weights), features_0 (the pyramid tensor),
        features_up_prev.  Returns (physical -> logical channel map, logical channel count) for the weight packer: the concat
        order of reference modules.py:261-264, [cv, features_0, flows_up_prev, features_up_prev], is the LOGICAL one."""
        assert not self.use_dc and c_cv + 2 <= self.CVX_CS and c_f0 % 16 == 0 and c_fu % 16 == 0
        a_phys = -(-(c_cv + 2) // 16) * 16
        m = np.full((a_phys + c_f0 + c_fu,), -1, np.int32)
        m[:c_cv] = np.arange(c_cv)
        m[c_cv:c_cv + 2] = c_cv + c_f0 + np.arange(2)
        m[a_phys:a_phys + c_f0] = c_cv + np.arange(c_f0)
        m[a_phys + c_f0:] = c_cv + c_f0 + 2 + np.arange(c_fu)
        return m, c_cv + c_f0 + 2 + c_fu

    def three_operand_ok(self, N, h, w, c_cv, c_f0, c_fu):
        """True where the first conv of this (non-DC) estimator takes the three-tensor input (the F16-pipe kernel, at the stage
        count that input has)."""
        if self.use_dc or not getattr(self, "f16x2", True) or not c_f0 or c_f0 % 16 or not c_fu or c_fu % 16 or c_cv != 81:
            return False
        return bool(_lib.lib().pwc_conv3x3_h2_supported(N, h, w, 96 + c_f0 + c_fu, self.filters[0], 1))

    def _run3(self, cvx, f0, feat_up, flow_res, flows_out, feat_out=None):
        """The estimator on the three-tensor input: cvx = View (C = 96 physical, cs = 84) of the [cv | flow | 0] records, f0 and
        feat_up Views of features_0 / features_up_prev, flow_res the flows_up_prev View the head's residual adds."""
        run = _ConvRunner(self)
        cm, cl = self.three_operand_map(81, f0.C, feat_up.C)
        x, keep = None, []
        for k, f in enumerate(self.filters):
            y = feat_out if (k == len(self.filters) - 1 and feat_out is not None) else None
            if k == 0:
                x, t = run.conv(cvx, f, y=y, cin_map=cm, cin_logical=cl, x2=f0, x3=feat_up)
            else:
                x, t = run.conv(x, f, y=y)
            keep.append(t)
        run.conv(x, 2, y=flows_out, slope=None, residual=flow_res)
        return x, keep[-1]

    def _run(self, buf, lay, flows_out, feat_out=None, f0_ext=None):
        """buf: View of the (N,h,w,lay.n_phys) buffer whose cv/f0/flow/feat_up segments
        are already filled (padding channels zero).  Writes `flows` (2 ch) to flows_out.
        non-DC: the 32-channel features go to feat_out (a View) -- DC: they are `buf`.
        f0_ext: the features_0 View where the layout keeps it external (_layout(f0_external=True))."""
        run = _ConvRunner(self)
        res = sub_view(buf, lay.offset("flow"), 2) if "flow" in lay.segments else None
        if self.use_dc:
            n_conv = sum(self.filters)
            start = lay.offset("cv")
            done = 0
            for k, f in enumerate(self.filters):
                off = lay.offset(f"conv{k}")
                x = sub_view(buf, start, lay.n_phys - start)
                cm = lay.cin_map(start, logical_base=n_conv - done)
                run.conv(x, f, y=sub_view(buf, off, f), cin_map=cm, cin_logical=lay.n_logical - (n_conv - done))
                start, done = off, done + f
            head_in = buf
            run.conv(View(buf.ptr, buf.cs, buf.N, buf.H, buf.W, lay.n_phys), 2, y=flows_out, slope=None,
                     cin_map=lay.cin_map(0, 0), cin_logical=lay.n_logical, residual=res)
            return head_in
        x = View(buf.ptr, buf.cs, buf.N, buf.H, buf.W, lay.n_phys)
        keep = []
        cm, cl = lay.cin_map(0, 0), lay.n_logical
        x2 = None
        if "f0" in lay.external:
            assert f0_ext is not None and f0_ext.C == lay.external["f0"][1]
            cm, x2 = lay.cin_map_with("f0"), f0_ext
        for k, f in enumerate(self.filters):
            y = feat_out if (k == len(self.filters) - 1 and feat_out is not None) else None
            x, t = run.conv(x, f, y=y, cin_map=cm, cin_logical=cl, x2=x2)
            keep.append(t)
            cm, cl, x2 = None, None, None
        run.conv(x, 2, y=flows_out, slope=None, residual=res)
        return x, keep[-1]

    def __call__(self, cv, features_0=None, flows_up_prev=None, features_up_prev=None, is_output=False):
        cvv, cv = as_view(cv, "cv")
        dev = cv.device
        ins = [("cv", cvv)]
        for nm, t in (("f0", features_0), ("flow", flows_up_prev), ("feat_up", features_up_prev)):
            if t is not None:
                v, _t = as_view(t, nm)
                assert (v.N, v.H, v.W) == (cvv.N, cvv.H, cvv.W)
                ins.append((nm, v))
        d = dict(ins)
        lay = self._layout(cvv.C, d["f0"].C if "f0" in d else 0, "flow" in d,
                           list(range(d["feat_up"].C)) if "feat_up" in d else None)
        buf_t = torch.zeros((cvv.N, cvv.H, cvv.W, lay.n_phys), dtype=torch.float32, device=dev)
        buf = View(buf_t.data_ptr(), lay.n_phys, cvv.N, cvv.H, cvv.W, lay.n_phys)
        for nm, v in ins:
            _copy_channels(v, sub_view(buf, lay.offset(nm), v.C), v.C)
        flows = torch.empty((cvv.N, cvv.H, cvv.W, 2), dtype=torch.float32, device=dev)
        fv = View(flows.data_ptr(), 2, cvv.N, cvv.H, cvv.W, 2)
        if self.use_dc:
            self._run(buf, lay, fv)
            features = _gather_logical(buf_t, lay)
        else:
            _, features = self._run(buf, lay, fv)
        if is_output:
            return flows, features
        h, w = cvv.H, cvv.W
        return flows, resize_bilinear(flows, (2 * h, 2 * w)), resize_bilinear(features, (2 * h, 2 * w))


def _gather_logical(buf_t, lay):
    """Logical (TF channel order) copy of a physical buffer -- index_select on the
    channel axis; only used by the public, non-fused call paths."""
    p2l = np.asarray(lay.phys2log)
    phys = np.nonzero(p2l >= 0)[0]
    order = phys[np.argsort(p2l[phys], kind="stable")]
    idx = torch.from_numpy(order.astype(np.int64)).to(buf_t.device)
    return buf_t.index_select(3, idx).contiguous()


# ---------------------------------------------------------------- context network (a5)

class ContextNetwork(_Module):
    """Context module (reference modules.py:290-326): 7 dilated 3x3 convs on
    concat[flows, features], residual add of `flows`."""

    def __init__(self, name="context"):
        super().__init__(name)

    def _run(self, buf, lay, out):
        """buf: View of the (N,h,w,lay.n_phys) buffer with `flow` and `features` filled."""
        run = _ConvRunner(self)
        x = View(buf.ptr, buf.cs, buf.N, buf.H, buf.W, lay.n_phys)
        cm, cl = lay.cin_map(0, 0), lay.n_logical
        keep = []
        for f, d in CONTEXT[:-1]:
            x, t = run.conv(x, f, dilation=d, cin_map=cm, cin_logical=cl)
            keep.append(t)
            cm, cl = None, None
        run.conv(x, 2, y=out, slope=None, residual=sub_view(buf, lay.offset("flow"), 2))

    def __call__(self, flows, features):
        fl, flows = as_view(flows, "flows")
        ft, features = as_view(features, "features")
        lay = ChannelLayout()
        lay.add("flow", 2)
        lay.add("features", ft.C)
        lay.finish(16)
        buf_t = torch.zeros((fl.N, fl.H, fl.W, lay.n_phys), dtype=torch.float32, device=flows.device)
        buf = View(buf_t.data_ptr(), lay.n_phys, fl.N, fl.H, fl.W, lay.n_phys)
        _copy_channels(fl, sub_view(buf, 0, 2), 2)
        _copy_channels(ft, sub_view(buf, lay.offset("features"), ft.C), ft.C)
        out = torch.empty((fl.N, fl.H, fl.W, 2), dtype=torch.float32, device=flows.device)
        self._run(buf, lay, View(out.data_ptr(), 2, fl.N, fl.H, fl.W, 2))
        return out
