#!/usr/bin/env python
"""Sharded EPE evaluation on the HIP path (the validation step of reference
train.py:124-131 as a stand-alone tool).

    python evaluate.py --list pairs.txt [--resume model_600.ckpt] [--batch 8] [--save_dir out]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --list ...

pairs.txt: one pair per line, `image_0 image_1 flow_gt.flo`.  Images are cropped to multiples
of 64 (reference test.py:13-17), scaled to [0,1]; the ground truth is cropped the same way.
Pairs are sharded contiguously over the ranks (one process per GPU, RCCL only for the final
all-gather of the statistics / flows); rank 0 prints one JSON line.
"""
import argparse
import json
import os
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", required=True)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--save_dir", default=None, help="write every predicted flow as <index>.flo (rank 0, after the gather)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from PIL import Image
    import pwcnet_amd
    from pwcnet_amd import ckpt, flow_io, sharding

    pairs = [ln.split() for ln in open(args.list) if ln.strip() and not ln.startswith("#")]
    model = pwcnet_amd.PWCDCNet(range_check="sync")      # results are final when a call returns (fp16-range check + fp32 repeat)
    if args.resume:
        model.load_weights(ckpt.load_weights(args.resume))

    def load_pair(i):
        p0, p1, pf = pairs[i]
        im0 = flow_io.factor_crop(np.asarray(Image.open(p0).convert("RGB")))
        im1 = flow_io.factor_crop(np.asarray(Image.open(p1).convert("RGB")))
        gt = flow_io.factor_crop(flow_io.read_flo(pf))
        return (torch.from_numpy(np.ascontiguousarray(im0, np.float32) / 255.0),
                torch.from_numpy(np.ascontiguousarray(im1, np.float32) / 255.0),
                torch.from_numpy(np.ascontiguousarray(gt, np.float32)))

    def forward(im0, im1):
        return model(im0, im1)[0]

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sharding.evaluate_pairs(forward, load_pair, len(pairs), args.batch, dist, dev, gather=args.save_dir is not None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is None or dist.get_rank() == 0:
        if args.save_dir:
            os.makedirs(args.save_dir, exist_ok=True)
            for i, f in enumerate(res["flows"].cpu().numpy()):
                flow_io.write_flo(os.path.join(args.save_dir, f"{i:06d}.flo"), f)
        print(json.dumps({"epe": res["epe"], "pairs": res["pairs"], "seconds": dt, "n_gpus": world,
                          "per_pair_epe": res["per_pair_epe"]}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
