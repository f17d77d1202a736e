"""Shared helpers for the tests (seeded inputs, weight sets)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pwcnet_amd import weights as W  # noqa: E402  (host-side numpy only)


def images(n, h, w, seed=1234):
    rng = np.random.RandomState(seed)
    return (rng.uniform(0, 1, size=(n, h, w, 3)).astype(np.float32),
            rng.uniform(0, 1, size=(n, h, w, 3)).astype(np.float32))


def smooth_images(n, h, w, seed=7, shift=(3, -2)):
    """Second image = first one translated (plus noise): gives the net a real motion."""
    rng = np.random.RandomState(seed)
    base = rng.uniform(0, 1, size=(n, h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    big = np.kron(base, np.ones((1, 8, 8, 1), np.float32))[:, : h + 16, : w + 16]
    im0 = big[:, 8:8 + h, 8:8 + w]
    im1 = big[:, 8 + shift[1]:8 + shift[1] + h, 8 + shift[0]:8 + shift[0] + w]
    return np.ascontiguousarray(im0), np.ascontiguousarray(im1)


def model_weights(use_dc=False, seed=0, bias_seed=1, gain=1.0):
    specs = W.conv_specs(use_dc=use_dc)
    w = W.randomize_biases(W.init_weights(specs, seed=seed), seed=bias_seed)
    if gain != 1.0:
        for k in w:
            if k.endswith("/kernel"):
                w[k] = (w[k] * gain).astype(np.float32)
    return w


def flow_field(n, h, w, seed=3, sigma=3.0, outliers=True):
    """flows ~ N(0, sigma^2) px with a few +-50 px outliers and exact integers."""
    rng = np.random.RandomState(seed)
    f = rng.normal(0, sigma, size=(n, h, w, 2)).astype(np.float32)
    if outliers:
        m = rng.uniform(size=(n, h, w)) < 0.02
        f[m] = rng.choice([-50.0, 50.0, -7.0, 3.0, 0.0], size=(int(m.sum()), 2)).astype(np.float32)
    return f
