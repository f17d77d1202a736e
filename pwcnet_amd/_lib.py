"""ctypes binding of libpwc_hip.so (the C ABI declared in include/pwc_hip.h).

The library is the ONLY compute path of this package: if it is missing or does not
export every declared symbol, importing the ops fails loudly -- there is no CPU or
eager-PyTorch fallback.  torch must be imported first so that the HIP runtime the
library binds to (libamdhip64.so.7) is the one PyTorch already loaded.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (loads libamdhip64 before libpwc_hip.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# PWC_HARNESS=1 (the A/B scripts under scripts/ set it before importing this package; nothing else does): load
# libpwc_hip_harness.so instead -- the same sources compiled with -DPWC_HARNESS, which adds the process-wide pwc_debug_* knobs
# (tile pinning, ablations).  The production library exports none of them (tests/test_host.py checks).
HARNESS = os.environ.get("PWC_HARNESS", "") == "1"
LIB_PATH = os.path.join(CSRC, "libpwc_hip_harness.so" if HARNESS else "libpwc_hip.so")
HARNESS_SIGNATURES_NAMES = ("pwc_debug_cost_volume_blk_rows", "pwc_debug_conv3x3_sk_tile", "pwc_debug_conv3x3_t32",
                            "pwc_debug_h2_reserve_cus")
SOURCES = ["conv3x3_mfma.hip", "conv3x3_wino.hip", "conv3x3_direct.hip", "cost_volume.hip", "pwc_ops.hip",
           "pwc_backward.hip", "conv3x3_wgrad.hip", "conv3x3_h2.hip", "conv3x3_c16pair.hip", "conv3x3_sk.hip", "conv3x3_t32.hip", "conv3x3_w32.hip"]
HEADERS = ["pwc_common.h", "cost_volume_roll.hip", "cost_volume_mfma.hip", "cost_volume_h2.hip", "cost_volume_blk.hip", "conv3x3_wino4.hip", os.path.join("..", "..", "include", "pwc_hip.h")]

_vp, _i, _f, _l, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/pwc_hip.h one to one
SIGNATURES = {
    "pwc_version": (_i, []),
    "pwc_error_string": (ctypes.c_char_p, [_i]),
    "pwc_cost_volume_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_cost_volume_uses_rolling_kernel": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "pwc_warp_bilinear_f32": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "pwc_warp_nearest_f32": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "pwc_cost_volume_coarse_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_resize_bilinear_pair_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pwc_warp_copy_f32": (_i, [_i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "pwc_warp_cost_volume_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_warp_cost_volume_concat_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_warp_cost_volume_concat_h2_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_warp_cost_volume_concat_blk_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _f, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_warp_cost_volume_concat_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "pwc_warp_cost_volume_concat_blk_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "pwc_conv3x3_workspace_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_plan": (_i, [_i, _i, _i, ctypes.POINTER(_i)]),
    "pwc_conv3x3_uses_halo_kernel": (_i, [_i, _i, _i, _i, _i]),
    "pwc_conv3x3_tile_shape": (_i, [_i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "pwc_conv3x3_wino_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_wino_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_wino_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_wino_workgroups": (_l, [_i, _i, _i, _i, _i]),
    "pwc_conv3x3_wino4_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_wino4_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_wino4_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_wino4_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_h2_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_h2_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_h2_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "pwc_conv3x3_h2_ex_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp, _vp]),
    "pwc_conv3x3_h2_ex3_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp, _vp]),
    "pwc_conv3x3_h2_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_h2_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_h2_plan": (_i, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_h2_stride2_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp, _vp]),
    "pwc_conv3x3_h2_stride2_supported": (_i, [_i, _i, _i, _i, _i]),
    "pwc_conv3x3_sk_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_sk_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_sk_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_sk_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_sk_variant_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "pwc_conv3x3_t32_packed_floats": (_sz, [_i]),
    "pwc_conv3x3_t32_pack_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pwc_conv3x3_t32_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_t32_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_w32_packed_floats": (_sz, [_i]),
    "pwc_conv3x3_w32_pack_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pwc_conv3x3_w32_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_w32_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_h2_stride2_packed_floats": (_sz, [_i, _i]),
    "pwc_conv3x3_h2_stride2_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pwc_conv3x3_h2_stride2_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "pwc_conv3x3_c16pair_packed_floats": (_sz, []),
    "pwc_conv3x3_c16pair_pack_f32": (_i, [_vp, _vp, _vp, _vp]),
    "pwc_conv3x3_c16pair_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_c16pair_supported": (_i, [_i, _i, _i]),
    "pwc_conv3x3_c3c16pair_packed_floats": (_sz, []),
    "pwc_conv3x3_c3c16pair_pack_f32": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "pwc_conv3x3_c3c16pair_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pwc_conv3x3_c3c16pair_supported": (_i, [_i, _i, _i]),
    "pwc_conv3x3_h2_variant_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "pwc_conv3x3_wino_split_plan": (_i, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_wino_split_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "pwc_conv3x3_wino_split_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "pwc_conv3x3_direct_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_resize_bilinear_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_resize_bilinear_status_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "pwc_absmax_f32": (_i, [_vp, _i, _l, _i, _vp, _vp]),
    "pwc_copy_channels_f32": (_i, [_vp, _i, _vp, _i, _l, _i, _vp]),
    "pwc_device_spin": (_i, [ctypes.c_longlong, _vp]),
    "pwc_device_touch": (_i, [_vp, _vp]),
    "pwc_lrelu_grad_f32": (_i, [_vp, _i, _vp, _i, _l, _i, _f, _vp]),
    "pwc_add_f32": (_i, [_vp, _i, _vp, _i, _l, _i, _f, _i, _vp]),
    "pwc_channel_sums_workspace_floats": (_sz, [_l, _i]),
    "pwc_channel_sums_f32": (_i, [_vp, _i, _l, _i, _vp, _sz, _vp, _i, _vp]),
    "pwc_lrelu_grad_channel_sums_f32": (_i, [_vp, _i, _vp, _i, _l, _i, _f, _vp, _sz, _vp, _i, _vp]),
    "pwc_resize_bilinear_grad_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "pwc_warp_bilinear_grad_f32": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pwc_warp_bilinear_grad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pwc_warp_bilinear_grad_det_f32": (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pwc_cost_volume_grad_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "pwc_flow_norm_grad_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _i, _i, _vp]),
    "pwc_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _f, _vp]),
    "pwc_conv3x3_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "pwc_conv3x3_wgrad_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pwc_flow_norm_workspace_floats": (_sz, [_i, _i, _i]),
    "pwc_flow_norm_sums_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp]),
}

if HARNESS:
    for _n in HARNESS_SIGNATURES_NAMES:
        SIGNATURES[_n] = (_i, [_i])

# status word bits of the F16-matrix-pipe kernels (include/pwc_hip.h)
STATUS_NONFINITE = 1
STATUS_STREAMK_TIMEOUT = 2

_lib = None


class PwcHipError(RuntimeError):
    pass


def _includes(path, seen=None):
    """The quoted #include closure of a source (per-object staleness check)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, "r", errors="replace") as f:
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', f.read(), flags=re.M):
            _includes(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
    return seen


def build_library(force=False, verbose=False):
    """hipcc-compile the HIP sources for gfx950 into csrc/libpwc_hip.so (in-tree): one object per source, compiled in
    parallel and only when the source or something it includes is newer, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build_harness" if HARNESS else "build")
    os.makedirs(objdir, exist_ok=True)
    # -fno-slp-vectorize: hipcc otherwise packs the scalar fp32 FMA chains of the
    # correlation kernel into v_pk_fma_f32 pairs (hundreds of v_mov shuffles, VGPR spills)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-pass-failed"]
    if HARNESS:
        flags.append("-DPWC_HARNESS")
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        newest = max(os.path.getmtime(d) for d in _includes(src))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            jobs.append([hipcc, *flags, "-c", src, "-o", obj])
    if not jobs and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH])
    return LIB_PATH


def lib():
    """The loaded library; raises if it is absent (never falls back to another path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PwcHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). pwcnet_amd has no fallback compute path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise PwcHipError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pwc_error_string(rc).decode()
        raise PwcHipError(f"{what or 'pwc call'} failed: {msg} (code {rc})")


def current_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
