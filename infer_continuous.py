#!/usr/bin/env python
"""Counterpart of the reference's test_continuous.py on the HIP path: flows between the
consecutive frames of an image sequence.

    python infer_continuous.py --input_images f0.png f1.png f2.png ... [--resume model_600.ckpt]

Follows reference test_continuous.py:45-64: frames are cropped to multiples of 64 and scaled to
[0,1]; every consecutive pair (i, i+1) goes through PWCDCNet; the 5-level flow pyramid is rescaled
to pixels per level (x 20 / 2^(6-l)).  Differences: consecutive pairs of equal size are run as ONE
batch (the reference feeds them one by one), and instead of a matplotlib figure each pair gets
<out>/<dname>/<fname>.png -- first frame and the colour-coded pyramid side by side -- plus
<out>/<dname>/<fname>.flo with the full-resolution flow.
"""
import argparse
import os
import re
from glob import glob

import numpy as np
import torch


def pyramid_montage(image, flows):
    """First frame followed by the colour-coded flows (coarse -> fine), all resized (nearest) to the
    frame's height, concatenated horizontally."""
    from pwcnet_amd import flow_io
    h = image.shape[0]
    tiles = [image]
    for f in flows:
        c = flow_io.flow_to_color(f)
        ry = np.arange(h) * c.shape[0] // h
        rx = np.arange(h * c.shape[1] // c.shape[0]) * c.shape[1] // (h * c.shape[1] // c.shape[0])
        tiles.append(c[ry][:, rx])
    return np.concatenate(tiles, axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--input_images", type=str, nargs="+", required=True, help="Target images (required)")
    ap.add_argument("-r", "--resume", type=str, default=None, help="Learned parameter checkpoint prefix [None]")
    ap.add_argument("--out", type=str, default="./test_figure")
    ap.add_argument("--batch", type=int, default=8, help="pairs per forward")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=3, help="forwards in flight (pwcnet_amd.ForwardPipeline depth; 1 = one at a time)")
    args = ap.parse_args()
    paths = []
    for p in args.input_images:                      # expand wild-cards (test_continuous.py:75-78)
        paths.extend(sorted(glob(p)) if "*" in p else [p])
    if len(paths) < 2:
        raise ValueError("# of input images must be >= 2")

    from PIL import Image
    import pwcnet_amd
    from pwcnet_amd import ckpt, flow_io

    torch.cuda.set_device(args.gpu)
    # Round 6: the batches of a sequence are independent forwards -- dealt to the replicas of a ForwardPipeline (HIP streams with
    # hardware queues of their own) the launch-bound coarse levels of one batch run under the matrix-bound launches of another.
    # Results are read behind pipeline.synchronize(): every forward has then passed its fp16-range check (a flagged one has been
    # repeated on the fp32 kernels), which is what range_check="sync" gave the one-forward-at-a-time loop.
    model = pwcnet_amd.ForwardPipeline(depth=args.pipeline)
    if args.resume is not None:
        print(f"Loading learned model from checkpoint {args.resume}")
        model.load_weights(ckpt.load_weights(args.resume))
    else:
        print("!!! Test with un-learned model !!!")

    frames = [flow_io.factor_crop(np.asarray(Image.open(p).convert("RGB"))) for p in paths]
    # runs of consecutive pairs whose frames all have one size: one batch each
    jobs = []
    i = 0
    while i < len(frames) - 1:
        j = i
        while j < len(frames) - 1 and j - i < args.batch and frames[j + 1].shape == frames[i].shape:
            j += 1
        if j == i:
            raise ValueError(f"{paths[i]} and {paths[i + 1]} differ in size after cropping")
        jobs.append((i, j))
        i = j
    num_levels = model.nets[0].num_levels
    window = 2 * max(1, args.pipeline)               # batches submitted before the first one is read back
    for w0 in range(0, len(jobs), window):
        tickets = []
        for i, j in jobs[w0:w0 + window]:
            seq = torch.from_numpy(np.stack(frames[i:j + 1]).astype(np.float32) / 255.0).cuda()
            tickets.append((i, j, model.submit(seq[:-1].contiguous(), seq[1:].contiguous())))
        model.synchronize()
        for i, j, ticket in tickets:
            flow_final, flows = ticket.result()
            flow_final = flow_final.cpu().numpy()
            flows = [f.cpu().numpy() for f in flows]
            for k in range(j - i):
                parts = re.split("[/.]", paths[i + k])
                dname, fname = (parts[-3:-1] if len(parts) >= 3 else ("", parts[-2]))
                os.makedirs(os.path.join(args.out, dname), exist_ok=True)
                pyr = [f[k] * (20.0 / 2 ** (num_levels - l)) for l, f in enumerate(flows)]
                Image.fromarray(pyramid_montage(frames[i + k], pyr)).save(os.path.join(args.out, dname, fname + ".png"))
                flow_io.write_flo(os.path.join(args.out, dname, fname + ".flo"), flow_final[k])
    print("Figure saved")


if __name__ == "__main__":
    main()
