#!/usr/bin/env python
"""Counterpart of the reference's test.py (inference + timing CLI) on the HIP path.

    python infer.py --input_images frame_0.png frame_1.png [--resume model_600.ckpt] [--time]

Follows reference test.py:27-65: crop both images to multiples of 64, scale to [0,1],
batch of one pair, default PWCDCNet(), optional checkpoint restore (TF V2 bundle, read
without TensorFlow), optional timing loop over repeated forwards of the same pair, then
the 5-level flow pyramid rescaled to pixels at each level (x 20 / 2^(6-l), test.py:57-60).
Writes <out>/flow_final.flo plus a colour-coded PNG per pyramid level instead of the
reference's matplotlib PDF.  No interactive GPU prompt: the device is cuda:<--gpu>.
"""
import argparse
import os
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input_images", type=str, nargs=2, required=True, help="Target images (required)")
    ap.add_argument("--resume", type=str, default=None, help="Learned parameter checkpoint prefix [None]")
    ap.add_argument("--time", "-t", action="store_true", help="measure inference speed")
    ap.add_argument("--iters", type=int, default=1000, help="timed iterations with --time [1000]")
    ap.add_argument("--out", type=str, default="./test_figure")
    ap.add_argument("--gpu", type=int, default=0)
    args = ap.parse_args()

    from PIL import Image
    import pwcnet_amd
    from pwcnet_amd import ckpt, flow_io

    torch.cuda.set_device(args.gpu)
    imgs = [flow_io.factor_crop(np.asarray(Image.open(p).convert("RGB"))) for p in args.input_images]
    images = np.array(imgs, dtype=np.float32) / 255.0                      # (2, h, w, 3)
    x = torch.from_numpy(images).cuda()

    model = pwcnet_amd.PWCDCNet(range_check="sync")      # results are final when a call returns (fp16-range check + fp32 repeat)
    if args.resume is not None:
        print(f"Loading learned model from checkpoint {args.resume}")
        model.load_weights(ckpt.load_weights(args.resume))
    else:
        print("!!! Test with un-learned model !!!")

    flow_final, flows = model(x[0:1], x[1:2])
    if args.time:
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            flow_final, flows = model(x[0:1], x[1:2])
        torch.cuda.synchronize()
        print(f"Inference time: {(time.time() - t0) / args.iters} sec (averaged over {args.iters} iterations)")

    os.makedirs(args.out, exist_ok=True)
    flow_io.write_flo(os.path.join(args.out, "flow_final.flo"), flow_final[0].cpu().numpy())
    for l, flow in enumerate(flows):
        upscale = 20.0 / 2 ** (model.num_levels - l)
        Image.fromarray(flow_io.flow_to_color(flow[0].cpu().numpy() * upscale)).save(
            os.path.join(args.out, f"flow_level{l}.png"))
    Image.fromarray(flow_io.flow_to_color(flow_final[0].cpu().numpy())).save(os.path.join(args.out, "flow_final.png"))
    print("Figure saved")


if __name__ == "__main__":
    main()
