"""Per-launch HIP-event timing of the ops (used by bench.py for the roofline figures).

When a timer is installed (``with OpTimer() as t:``) every kernel launch issued by
pwcnet_amd.modules is bracketed by two events recorded on torch's current stream -- the
stream the launch itself goes to -- and tagged with its kernel name and its ALGORITHMIC
work (flops and bytes as DESIGN.md defines them).  No synchronisation happens until
``summary()``.
"""
import collections
import contextlib

import torch

_ACTIVE = None


def source_stamp():
    """sha256 over the kernel sources and the C header (pwcnet_amd/csrc/*.hip, *.h, include/pwc_hip.h; names and bytes, sorted):
    what identifies THE BUILD a set of hardware-counter passes was taken from.  scripts/pmc_*_table.py write it into
    profiles/pmc_traffic.json, bench.py recomputes it and refuses to print counter traffic taken from other sources
    (`traffic_stale`).  Computable on the GPU box (no .git there)."""
    import glob
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "pwcnet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "pwcnet_amd", "csrc", "*.h"))
                   + [os.path.join(root, "include", "pwc_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def active():
    return _ACTIVE


class OpTimer:
    def __init__(self, only=None):
        """only: optional set of kernel-name prefixes to time (others run un-instrumented:
        an event pair costs a few microseconds and a pipeline bubble per launch)."""
        self.records = []   # (kernel, flops, bytes, start event, end event)
        self.only = None if only is None else tuple(only)
        # callers may switch the timer off for some steps (bench.py samples every 8th step of its
        # timed region: an event pair serialises the launches around it, ~4 us each on MI355X)
        self.enabled = True

    def wants(self, kernel):
        return self.enabled and (self.only is None or kernel.startswith(self.only))

    def __enter__(self):
        global _ACTIVE
        self._prev = _ACTIVE
        _ACTIVE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._prev
        return False

    @contextlib.contextmanager
    def launch(self, kernel, flops=0.0, nbytes=0.0, exec_flops=None):
        """exec_flops: multiply-add flops the launch EXECUTES when that differs from the
        algorithmic `flops` (Winograd kernels); defaults to `flops`."""
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        try:
            yield
        finally:
            e.record()
            self.records.append((kernel, float(flops), float(nbytes), s, e,
                                 float(flops if exec_flops is None else exec_flops)))

    def summary(self):
        """{kernel: dict(launches, ms, flops, bytes, exec_flops)} -- synchronises the device."""
        torch.cuda.synchronize()
        out = collections.OrderedDict()
        for k, fl, by, s, e, xf in self.records:
            d = out.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, exec_flops=0.0))
            d["exec_flops"] += xf
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


@contextlib.contextmanager
def timed(kernel, flops=0.0, nbytes=0.0, exec_flops=None):
    t = _ACTIVE
    if t is None or kernel is None or not t.wants(kernel):
        yield
    else:
        with t.launch(kernel, flops, nbytes, exec_flops):
            yield
