#!/usr/bin/env python
"""Counterpart of the reference's train.py on the HIP path (pwcnet_amd.train.Trainer).

    python train.py -dd <dataset_dir> [-e 100] [-b 4] [--crop_shape 384 448] [--lr 1e-4] [--gamma 4e-4] ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -dd <dir> ...

Follows reference train.py:110-170: per epoch a pass over the training pairs (images / 255, un-scaled ground-truth
flow), one Adam step per batch, then validation (EPE of flows_final, train.py:77), then `./model/model_<epoch>.ckpt`
-- written as a TensorFlow V2 bundle (pwcnet_amd.ckpt, no TensorFlow needed) that the reference's Saver and this
repo's infer.py can restore.  Same argument names and defaults as the reference where they apply; no interactive
GPU prompt (one process per GPU, LOCAL_RANK picks the device; gradients are averaged with one RCCL all-reduce).

Dataset: the reference's loaders live in its empty `datahandler` submodule; here a directory is scanned for
MPI-Sintel-style pairs  <dir>/<pass>/<seq>/frame_NNNN.png  with  <dir>/flow/<seq>/frame_NNNN.flo , or, with
`--dataset synthetic`, random translating textures with known flow are generated (no files needed).
Both use_dc settings and both losses (multiscale, robust) are implemented (Trainer docstring).
"""
import argparse
import glob
import os
import time

import numpy as np
import torch


def sintel_pairs(root, render="clean"):
    pairs = []
    for seq in sorted(glob.glob(os.path.join(root, render, "*"))):
        frames = sorted(glob.glob(os.path.join(seq, "frame_*.png")))
        for a, b in zip(frames[:-1], frames[1:]):
            flo = os.path.join(root, "flow", os.path.basename(seq), os.path.basename(a).replace(".png", ".flo"))
            if os.path.exists(flo):
                pairs.append((a, b, flo))
    return pairs


class SyntheticPairs:
    """Random smooth textures translated by a per-pair integer shift; ground truth = that shift."""

    def __init__(self, n, shape, seed=0):
        self.n, self.shape, self.seed = n, shape, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.RandomState(self.seed + i)
        h, w = self.shape
        sx, sy = rng.randint(-6, 7), rng.randint(-6, 7)
        base = rng.uniform(0, 255, size=(h // 8 + 4, w // 8 + 4, 3)).astype(np.float32)
        big = np.kron(base, np.ones((8, 8, 1), np.float32))
        im0 = big[16:16 + h, 16:16 + w]
        im1 = big[16 - sy:16 - sy + h, 16 - sx:16 - sx + w]       # im1(p) = im0(p - s): content moves by +s
        flow = np.empty((h, w, 2), np.float32)
        flow[..., 0], flow[..., 1] = sx, sy
        return np.ascontiguousarray(im0), np.ascontiguousarray(im1), flow


class FilePairs:
    def __init__(self, pairs, crop_shape, crop_type="random", seed=0):
        self.pairs, self.crop, self.crop_type = pairs, crop_shape, crop_type
        self.rng = np.random.RandomState(seed)

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        from PIL import Image
        from pwcnet_amd import flow_io
        a, b, f = self.pairs[i]
        im0, im1 = (np.asarray(Image.open(p).convert("RGB"), np.float32) for p in (a, b))
        flow = flow_io.read_flo(f)
        ch, cw = self.crop
        H, W = im0.shape[:2]
        y0 = self.rng.randint(0, H - ch + 1) if self.crop_type == "random" else (H - ch) // 2
        x0 = self.rng.randint(0, W - cw + 1) if self.crop_type == "random" else (W - cw) // 2
        sl = (slice(y0, y0 + ch), slice(x0, x0 + cw))
        return np.ascontiguousarray(im0[sl]), np.ascontiguousarray(im1[sl]), np.ascontiguousarray(flow[sl])


def batches(ds, idx, bs):
    for i in range(0, len(idx) - bs + 1, bs):                       # drop_last, like the reference's loader
        items = [ds[j] for j in idx[i:i + bs]]
        yield tuple(torch.from_numpy(np.stack([it[k] for it in items])) for k in range(3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--dataset", type=str, default="SintelClean", help="SintelClean | SintelFinal | synthetic")
    ap.add_argument("-dd", "--dataset_dir", type=str, default=None, help="Directory containing target dataset")
    ap.add_argument("-e", "--num_epochs", type=int, default=100, help="# of epochs [100]")
    ap.add_argument("-b", "--batch_size", type=int, default=4, help="Batch size per GPU [4]")
    ap.add_argument("--crop_type", type=str, default="random", help="Crop type for raw data [random]")
    ap.add_argument("--crop_shape", nargs=2, type=int, default=[384, 448], help="Crop shape for raw data [384, 448]")
    ap.add_argument("--num_levels", type=int, default=6)
    ap.add_argument("--search_range", type=int, default=4)
    ap.add_argument("--warp_type", default="bilinear", choices=["bilinear", "nearest"])
    ap.add_argument("--use-dc", dest="use_dc", action="store_true")
    ap.add_argument("--no-dc", dest="use_dc", action="store_false")
    ap.set_defaults(use_dc=False)
    ap.add_argument("--output_level", type=int, default=4)
    ap.add_argument("--loss", default="multiscale", choices=["multiscale", "robust"])
    ap.add_argument("--lr", type=float, default=1e-4, help="Learning rate [1e-4]")
    ap.add_argument("--lr_scheduling", dest="lr_scheduling", action="store_true")
    ap.add_argument("--no-lr_scheduling", dest="lr_scheduling", action="store_false")
    ap.set_defaults(lr_scheduling=True)
    ap.add_argument("--epsilon", type=float, default=0.02, help="robust loss epsilon [0.02]")
    ap.add_argument("--q", type=float, default=0.4, help="robust loss exponent [0.4]")
    ap.add_argument("--weights", nargs="+", type=float, default=[0.32, 0.08, 0.02, 0.01, 0.005])
    ap.add_argument("--gamma", type=float, default=0.0004, help="Coefficient for weight decay [4e-4]")
    ap.add_argument("-r", "--resume", type=str, default=None, help="Learned parameter checkpoint prefix [None]")
    ap.add_argument("--synthetic_pairs", type=int, default=64, help="pairs per epoch with --dataset synthetic")
    ap.add_argument("--val_fraction", type=float, default=0.1)
    ap.add_argument("--model_dir", type=str, default="./model")
    args = ap.parse_args()
    # what the training path does not build is refused here, not by an assert after the data set has been read
    if args.warp_type != "bilinear":
        ap.error("--warp_type nearest has no gradient path here (the reference trains with its model default, bilinear)")
    if args.num_levels != 6 or args.search_range != 4:
        ap.error("training supports --num_levels 6 --search_range 4 (the reference's scales and *20 hard-code 6 levels)")
    if not 0 <= args.output_level < args.num_levels:
        ap.error("--output_level must be in [0, num_levels)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        for key, item in vars(args).items():
            print(f"{key} : {item}")

    from pwcnet_amd import ckpt, sharding
    from pwcnet_amd.train import Trainer

    if args.dataset == "synthetic":
        ds = SyntheticPairs(args.synthetic_pairs, tuple(args.crop_shape))
    else:
        if not args.dataset_dir:
            raise SystemExit("train.py: --dataset_dir is required for file datasets")
        pairs = sintel_pairs(args.dataset_dir, "final" if args.dataset == "SintelFinal" else "clean")
        if not pairs:
            raise SystemExit(f"train.py: no frame pairs with flow found under {args.dataset_dir}")
        ds = FilePairs(pairs, tuple(args.crop_shape), args.crop_type)
    n_val = max(1, int(len(ds) * args.val_fraction))
    perm = np.random.RandomState(0).permutation(len(ds))
    val_idx, train_idx = perm[:n_val], perm[n_val:]

    trainer = Trainer(num_levels=args.num_levels, search_range=args.search_range, warp_type=args.warp_type,
                      use_dc=args.use_dc, output_level=args.output_level, weights=args.weights, gamma=args.gamma,
                      lr=args.lr, lr_scheduling=args.lr_scheduling, device=f"cuda:{local_rank}", dist=dist,
                      loss=args.loss, epsilon=args.epsilon, q=args.q)
    if args.resume is not None:
        print(f"Loading learned model from checkpoint {args.resume}")
        trainer.load_weights(ckpt.load_weights(args.resume))

    for e in range(args.num_epochs):
        order = np.random.RandomState(1000 + e).permutation(train_idx)
        lo, hi = sharding.shard_range(len(order), world, rank)       # pairs shard across the ranks
        steps = (hi - lo) // args.batch_size
        if dist is not None:                                          # every rank must take the same number of steps
            st = torch.tensor([steps], device="cuda")
            dist.all_reduce(st, op=dist.ReduceOp.MIN)
            steps = int(st.item())
        t0, loss_sum, n_steps = time.time(), 0.0, 0
        for images_0, images_1, flows_gt in batches(ds, order[lo:lo + steps * args.batch_size], args.batch_size):
            loss = trainer.step((images_0 / 255.0).cuda(), (images_1 / 255.0).cuda(), flows_gt.cuda())
            loss_sum += float(loss)
            n_steps += 1
        # validation: EPE of flows_final (reference train.py:77,124-131), sharded over the ranks
        from pwcnet_amd import PWCDCNet
        net = PWCDCNet(num_levels=args.num_levels, search_range=args.search_range, warp_type=args.warp_type,
                       use_dc=args.use_dc, output_level=args.output_level)
        net.load_weights(trainer.state_dict())
        res = sharding.evaluate_pairs(lambda a, b: net(a / 255.0, b / 255.0)[0],
                                      lambda i: tuple(torch.from_numpy(x) for x in ds[val_idx[i]]),
                                      len(val_idx), batch=args.batch_size, dist=dist, device="cuda")
        if rank == 0:
            dt = time.time() - t0
            print(f"epoch {e + 1}: loss/pwc {loss_sum / max(n_steps, 1):.4f}  EPE/val {res['epe']:.4f}  "
                  f"global_step {trainer.global_step}  {n_steps * args.batch_size * world / max(dt, 1e-9):.1f} pairs/s")
            os.makedirs(args.model_dir, exist_ok=True)
            ckpt.save_weights(os.path.join(args.model_dir, f"model_{e + 1}.ckpt"), trainer.state_dict())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
