"""Optical-flow file IO and colour coding (host-side numpy).

`.flo` (Middlebury): float32 magic 202021.25, int32 width, int32 height, then
height*width*2 float32 (u, v interleaved) -- what reference flow_utils.py:13-29 reads/writes.
`flow_to_color`: the Middlebury colour wheel (55 hues: RY 15, YG 6, GC 4, CB 11, BM 13,
MR 6; reference flow_utils.py:32-153): hue = direction, saturation = magnitude / max.
"""
import numpy as np

FLO_MAGIC = 202021.25


def write_flo(path, flow):
    flow = np.ascontiguousarray(flow, dtype="<f4")
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError(f"flow must be (H, W, 2), got {flow.shape}")
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        np.array([FLO_MAGIC], "<f4").tofile(f)
        np.array([w, h], "<i4").tofile(f)
        flow.tofile(f)


def read_flo(path):
    with open(path, "rb") as f:
        magic = np.fromfile(f, "<f4", count=1)
        if magic.size != 1 or float(magic[0]) != FLO_MAGIC:
            raise ValueError(f"{path}: not a .flo file (bad magic)")
        w, h = (int(v) for v in np.fromfile(f, "<i4", count=2))
        data = np.fromfile(f, "<f4", count=h * w * 2)
        if data.size != h * w * 2:
            raise ValueError(f"{path}: truncated ({data.size} of {h * w * 2} values)")
    return data.reshape(h, w, 2)


def color_wheel():
    """(55, 3) RGB in 0..255."""
    segments = [(15, (255, 0, 0), (0, 1, 0)), (6, (255, 255, 0), (-1, 0, 0)), (4, (0, 255, 0), (0, 0, 1)),
                (11, (0, 255, 255), (0, -1, 0)), (13, (0, 0, 255), (1, 0, 0)), (6, (255, 0, 255), (0, 0, -1))]
    rows = []
    for n, start, ramp in segments:
        t = np.floor(255.0 * np.arange(n) / n)
        rows.append(np.stack([np.full(n, start[c], np.float64) + ramp[c] * t for c in range(3)], axis=1))
    return np.concatenate(rows, axis=0)


def flow_to_color(flow, max_rad=None):
    """(H, W, 2) flow -> (H, W, 3) uint8 RGB; magnitudes are normalised by `max_rad`
    (default: the largest magnitude in the field, as reference vis_flow does)."""
    flow = np.asarray(flow, np.float64)
    u, v = flow[..., 0].copy(), flow[..., 1].copy()
    bad = ~np.isfinite(u) | ~np.isfinite(v) | (np.abs(u) > 1e9) | (np.abs(v) > 1e9)
    u[bad] = 0
    v[bad] = 0
    rad = np.sqrt(u * u + v * v)
    if max_rad is None:
        max_rad = float(rad.max())
    scale = max_rad + np.finfo(np.float64).eps
    u, v, rad = u / scale, v / scale, rad / scale
    wheel = color_wheel()
    ncols = wheel.shape[0]
    fk = (np.arctan2(-v, -u) / np.pi + 1.0) / 2.0 * (ncols - 1)
    k0 = np.floor(fk).astype(np.int64)
    k1 = (k0 + 1) % ncols
    f = (fk - k0)[..., None]
    col = (1 - f) * wheel[k0] / 255.0 + f * wheel[k1] / 255.0
    inside = (rad <= 1)[..., None]
    col = np.where(inside, 1 - rad[..., None] * (1 - col), col * 0.75)
    return np.floor(255.0 * col).astype(np.uint8)


def factor_crop(image, factor=64):
    """Crop H, W down to multiples of `factor` (reference test.py:13-17)."""
    assert image.ndim == 3
    h, w = image.shape[:2]
    return image[: factor * (h // factor), : factor * (w // factor)]
