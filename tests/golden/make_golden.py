"""Generates the committed golden vectors with the repo's own CPU oracle.

The reference (TF 1.8) cannot run in this project, so these are regression pins of the
oracle (PARITY UNPINNED, see oracle/pwc_oracle.c), not reference outputs.  Re-run:
    python tests/golden/make_golden.py
Weights are NOT stored: they are regenerated from the seeded numpy stream in
pwcnet_amd.weights.init_weights / randomize_biases (tests/util.model_weights).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from tests import util  # noqa: E402


def rnd(shape, seed):
    return np.random.RandomState(seed).uniform(-1, 1, size=shape).astype(np.float32)


def main():
    # per-op vectors
    g = {}
    g["cv_f0"], g["cv_f1"] = rnd((2, 10, 20, 8), 101), rnd((2, 10, 20, 8), 102)
    g["cv_out"] = orc.cost_volume(g["cv_f0"], g["cv_f1"], 4)
    g["warp_x"] = rnd((2, 12, 20, 8), 103)
    g["warp_flow"] = util.flow_field(2, 12, 20, seed=104)
    g["warp_bilinear"] = orc.warp(g["warp_x"], g["warp_flow"], "bilinear")
    g["warp_nearest"] = orc.warp(g["warp_x"], g["warp_flow"], "nearest")
    g["rs_x"] = rnd((2, 6, 10, 4), 105)
    g["rs_x2"] = orc.resize_bilinear(g["rs_x"], (12, 20))
    g["conv_x"] = rnd((2, 12, 16, 16), 106)
    g["conv_k"] = rnd((3, 3, 16, 32), 107) * 0.2
    g["conv_b"] = rnd((32,), 108) * 0.1
    g["conv_s2"] = orc.conv3x3(g["conv_x"], g["conv_k"], g["conv_b"], 2, 1, 0.1)
    g["conv_d4"] = orc.conv3x3(g["conv_x"], g["conv_k"], g["conv_b"], 1, 4, 0.1)
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **g)

    # end-to-end 64x128 pair, both estimator variants
    im0, im1 = util.smooth_images(1, 64, 128)
    for use_dc in (False, True):
        w = util.model_weights(use_dc)
        net = orc.OraclePWCDCNet(w, use_dc=use_dc)
        final, pyr = net(im0, im1)
        e = {"images_0": im0, "images_1": im1, "flows_final": final}
        for l, p in enumerate(pyr):
            e[f"flows_{l}"] = p
        np.savez_compressed(os.path.join(HERE, f"e2e_64x128_dc{int(use_dc)}.npz"), **e)
        print("dc", use_dc, "max |flows_final|", float(np.abs(final).max()),
              [float(np.abs(p).max()) for p in pyr])


if __name__ == "__main__":
    main()
