#!/usr/bin/env python
"""Winograd F(4x4,3x3) in fp32: does its rounding error fit the 1e-3 px flow bound?  (decide
with data.)  Runs on the CPU, no GPU needed.

The float64 torch restatement of the forward (oracle/torch_ref.py) is the reference.  The same graph is evaluated in
float32 with the 3x3 stride-1 convolutions computed three ways:
  direct    F.conv2d in fp32 (what a direct MFMA kernel computes, up to summation order)
  F(2x2)    Winograd F(2x2,3x3) in fp32 on every eligible layer -- the arithmetic conv3x3_wino.hip ships
  F(4x4)    Winograd F(4x4,3x3) in fp32 on a chosen set of layers, F(2x2) on the others
All transforms are carried out in fp32 exactly as a kernel would (input transform B^T d B per 16-channel group in
registers, products accumulated in fp32, output transform A^T M A in fp32); the transformed weights U = G g G^T are
computed in float64 and rounded once (the best a weight packer can do).  Cases = the parity tests that bound the
shipped path: tests/test_gpu_model.py::test_e2e_large_flows_vs_oracle (gain 1.35 / 1.6, 2 x 128 x 192) and gain 1.0.

Output: a table of max |flow error| (px, flows_final) per configuration; profiles/r03_f4x4_numerics.txt keeps it.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_ref as TR  # noqa: E402
from tests import util  # noqa: E402

torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))

# ---- Winograd matrices (Lavin & Gray), float64 masters
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
               [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def wino_conv(x, kernel, bias, m, dilation=1):
    """3x3 stride-1 SAME conv of NHWC fp32 `x` by Winograd F(m x m, 3x3), all arithmetic in fp32.
    Dilation d = d*d ordinary convs on the pixel sub-lattices (as conv3x3_wino.hip)."""
    if dilation > 1:
        N, H, W, C = x.shape
        y = torch.empty((N, H, W, kernel.shape[3]), dtype=x.dtype)
        for ry in range(dilation):
            for rx in range(dilation):
                sub = x[:, ry::dilation, rx::dilation]
                if sub.shape[1] == 0 or sub.shape[2] == 0:
                    continue
                y[:, ry::dilation, rx::dilation] = wino_conv(sub, kernel, bias, m, 1)
        return y
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    t = m + 2
    N, H, W, C = x.shape
    Cout = kernel.shape[3]
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x.permute(0, 3, 1, 2), (1, tw * m + 1 - W, 1, th * m + 1 - H))          # N C H+ W+
    tiles = xp.unfold(2, t, m).unfold(3, t, m)                                         # N C th tw t t
    bt = torch.from_numpy(BT).to(torch.float32)
    at = torch.from_numpy(AT).to(torch.float32)
    # input transform in fp32: rows, then columns (two passes of adds / small-integer multiplies)
    v = torch.einsum("ai,ncyxij->ncyxaj", bt, tiles)
    v = torch.einsum("bj,ncyxaj->ncyxab", bt, v)
    # transformed weights: float64, rounded once
    g64 = kernel.to(torch.float64).permute(2, 3, 0, 1)                                 # C Cout 3 3
    u = torch.einsum("ap,copq,bq->coab", torch.from_numpy(G), g64, torch.from_numpy(G)).to(torch.float32)
    # 16 / 36 independent GEMMs over the channels, fp32 accumulation
    mm = torch.einsum("coab,ncyxab->noyxab", u, v)
    y = torch.einsum("ia,noyxab->noyxib", at, mm)
    y = torch.einsum("jb,noyxib->noyxij", at, y)                                       # N Cout th tw m m
    y = y.permute(0, 2, 4, 3, 5, 1).reshape(N, th * m, tw * m, Cout)[:, :H, :W]
    return y + bias


class Net32(TR.TorchPWCDCNet):
    """fp32 forward with a per-layer choice of the stride-1 conv arithmetic: mode[name] in {0: direct, 2, 4}."""

    def __init__(self, weights, mode, default=0, **kw):
        super().__init__(weights, **kw)
        self.mode, self.default = mode, default

    def _conv(self, scope, k, x, stride=1, dilation=1, act=True):
        n = f"{self.name}/{scope}/conv2d" + ("" if k == 0 else f"_{k}")
        kern, bias = self.w[n + "/kernel"], self.w[n + "/bias"]
        m = self.mode.get(f"{scope}/{k}", self.default)
        eligible = stride == 1 and kern.shape[3] % 16 == 0 and x.shape[3] >= 16 and x.shape[1] >= 14
        if m and eligible:
            y = wino_conv(x, kern, bias, m, dilation)
        else:
            y = TR.conv3x3_same(x, kern, bias, stride, dilation)
        return TR.leaky_relu(y) if act else y


BIG8 = ["optflow_4/0", "optflow_4/1", "optflow_4/2", "optflow_4/3", "context/1", "context/2", "context/3", "context/4"]


def run_case(gain, shape=(2, 128, 192), seed=21, shift=(3, -5)):
    w = util.model_weights(False, gain=gain)
    im0, im1 = util.smooth_images(*shape, seed=seed, shift=shift)
    w64 = {k: torch.from_numpy(v).to(torch.float64) for k, v in w.items()}
    w32 = {k: torch.from_numpy(v) for k, v in w.items()}
    a64, b64 = torch.from_numpy(im0).to(torch.float64), torch.from_numpy(im1).to(torch.float64)
    a32, b32 = torch.from_numpy(im0), torch.from_numpy(im1)
    with torch.no_grad():
        ref = TR.TorchPWCDCNet(w64)(a64, b64)[0]
        mag = float(ref.abs().max())
        rows = []

        def err(mode, default):
            out = Net32(w32, mode, default)(a32, b32)[0]
            return float((out.to(torch.float64) - ref).abs().max())

        rows.append(("direct fp32 everywhere", err({}, 0)))
        rows.append(("F(2x2) everywhere eligible (shipped arithmetic)", err({}, 2)))
        rows.append(("F(4x4) on the 8 big level-4 layers, F(2x2) elsewhere", err({n: 4 for n in BIG8}, 2)))
        rows.append(("F(4x4) everywhere eligible", err({}, 4)))
        for n in BIG8:
            rows.append((f"F(4x4) on {n} only, F(2x2) elsewhere", err({n: 4}, 2)))
    return mag, rows


def main():
    out = []
    for gain in (1.0, 1.35, 1.6):
        mag, rows = run_case(gain)
        out.append(f"== kernel gain {gain}: 2 x 128 x 192 pair, max |flow| {mag:.2f} px (float64 reference), bound 1e-3 px")
        base = rows[1][1]
        for name, e in rows:
            out.append(f"   {name:58s} max abs err {e:.3e} px   x{e / base:5.1f} of F(2x2)   {'OK' if e * 3 <= 1e-3 else ('within bound, margin < 3x' if e <= 1e-3 else 'EXCEEDS 1e-3')}")
        print("\n".join(out[-(len(rows) + 1):]), flush=True)
    path = os.path.join(ROOT, "profiles", "r03_f4x4_numerics.txt")
    with open(path, "w") as f:
        f.write("# scripts/exp_f4x4_numerics.py -- fp32 Winograd F(4x4,3x3) vs the 1e-3 px bound (CPU emulation of the kernel arithmetic)\n")
        f.write("\n".join(out) + "\n")
    print("written:", path)


if __name__ == "__main__":
    main()
