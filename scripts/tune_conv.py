"""Times every tile configuration (and tap split) of pwc_conv3x3_f32 on the layer shapes of
the PWCDCNet forward (batch 8, 448x1024) and compares with the library's automatic plan.
Run on the GPU box:  python scripts/tune_conv.py [--quick]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pwcnet_amd import _lib  # noqa: E402

L = _lib.lib()


def p(t):
    return ctypes.c_void_p(t.data_ptr())


# (tag, images, H, W, Cin_phys, Cin_logical, Cout, stride, dilation)
def layers(batch=8):
    B2 = 2 * batch
    out = []
    fp = [16, 32, 64, 96, 128, 192]
    H, W, cin = 448, 1024, 3
    for l, f in enumerate(fp):
        if l > 0:
            out.append((f"fp{l}_s2", B2, H, W, cin, cin, f, 2, 1))
        H, W = H // 2, W // 2
        out.append((f"fp{l}_s1", B2, H, W, f, f, f, 1, 1))
        cin = f
    # estimators, level 0..4 (h,w of level l: 7*2^l x 16*2^l)
    phys = [288, 256, 224, 192, 160]
    logi = [273, 243, 211, 179, 147]
    for l in range(5):
        h, w = 7 * 2 ** l, 16 * 2 ** l
        out.append((f"of{l}_c0", batch, h, w, phys[l], logi[l], 128, 1, 1))
        out.append((f"of{l}_c1", batch, h, w, 128, 128, 128, 1, 1))
        out.append((f"of{l}_c2", batch, h, w, 128, 128, 96, 1, 1))
        out.append((f"of{l}_c3", batch, h, w, 96, 96, 64, 1, 1))
        out.append((f"of{l}_c4", batch, h, w, 64, 64, 32, 1, 1))
    h, w = 112, 256
    for k, (ci, cil, co, d) in enumerate([(48, 34, 128, 1), (128, 128, 128, 2), (128, 128, 128, 4),
                                          (128, 128, 96, 8), (96, 96, 64, 16), (64, 64, 32, 1)]):
        out.append((f"ctx_c{k}", batch, h, w, ci, cil, co, 1, d))
    return out


def time_conv(x, packed, bias, y, ws, N, H, W, cin_phys, cout, stride, dil, tile, split, iters=5):
    def run():
        rc = L.pwc_conv3x3_f32(p(x), cin_phys, p(packed), p(bias), p(y), cout, N, H, W, cin_phys, cout, stride, dil,
                               1, 0.1, tile, split, p(ws), ws.numel(), None)
        return rc
    rc = run()
    if rc != 0:
        return None
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def time_wino(x, packed_u, bias, y, N, H, W, cin_phys, cout, dil=1, iters=5):
    def run():
        return L.pwc_conv3x3_wino_f32(p(x), cin_phys, p(packed_u), p(bias), p(y), cout, N, H, W, cin_phys, cout, dil, 1, 0.1, None)
    if run() != 0:
        return None
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def wino_main():
    tot_a = tot_w = tot_best = 0.0
    ws = torch.empty(64 << 20, device="cuda")
    print(f"{'layer':9s} {'M':>8s} {'cin':>4s} {'cout':>4s} {'GF':>6s} | {'auto us':>8s} {'TF':>6s} | {'wino us':>8s} {'eff TF':>7s} speedup")
    for tag, N, H, W, cp, cl, co, st, dl in layers():
        if st != 1 or co % 16:
            continue
        M = N * H * W
        x = torch.rand((N, H, W, cp), device="cuda") - 0.5
        packed = torch.rand((L.pwc_conv3x3_packed_floats(cp, co),), device="cuda") - 0.5
        pu = torch.rand((L.pwc_conv3x3_wino_packed_floats(cp, co),), device="cuda") - 0.5
        bias = torch.zeros(co, device="cuda")
        y = torch.empty((N, H, W, co), device="cuda")
        gf = 2.0 * M * 9 * cl * co / 1e9
        ta = min(time_conv(x, packed, bias, y, ws, N, H, W, cp, co, 1, dl, -1, 0) for _ in range(2))
        tw = min(time_wino(x, pu, bias, y, N, H, W, cp, co, dl) for _ in range(2))
        tot_a += ta; tot_w += tw; tot_best += min(ta, tw)
        print(f"{tag:9s} {M:8d} {cp:4d} {co:4d} {gf:6.2f} | {ta:8.1f} {gf / ta * 1e3:6.1f} | {tw:8.1f} {gf / tw * 1e3:7.1f} {ta / tw:5.2f}x")
    print(f"TOTAL eligible layers: auto {tot_a:.0f} us, winograd {tot_w:.0f} us, per-layer best {tot_best:.0f} us")


def main():
    if "--wino" in sys.argv:
        return wino_main()
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    ws = torch.empty(64 << 20, device="cuda")
    tot_auto, tot_best = 0.0, 0.0
    print(f"{'layer':9s} {'M':>8s} {'cin':>4s} {'cout':>4s} {'GF':>6s} | {'auto us':>8s} {'TF':>6s} plan | best-single-tile us (tile,split) | all")
    for tag, N, H, W, cp, cl, co, st, dl in layers():
        Ho, Wo = -(-H // st), -(-W // st)
        M = N * Ho * Wo
        x = torch.rand((N, H, W, cp), device="cuda") - 0.5
        packed = torch.rand((L.pwc_conv3x3_packed_floats(cp, co),), device="cuda") - 0.5
        bias = torch.zeros(co, device="cuda")
        y = torch.empty((N, Ho, Wo, co), device="cuda")
        gf = 2.0 * M * 9 * cl * co / 1e9
        plan = (ctypes.c_int * 4)()
        L.pwc_conv3x3_plan(M, co, cp, plan)
        t_auto = time_conv(x, packed, bias, y, ws, N, H, W, cp, co, st, dl, -1, 0)
        res = []
        for tile in range(15):
            bm, bn = ctypes.c_int(), ctypes.c_int()
            L.pwc_conv3x3_tile_shape(tile, bm, bn)
            if co % bn.value:
                continue
            for split in ((1,) if (quick or M > 60000) else (1, 3, 9)):
                if split > 1 and split * M * co > ws.numel():
                    continue
                t = time_conv(x, packed, bias, y, ws, N, H, W, cp, co, st, dl, tile, split)
                if t is not None:
                    res.append((t, tile, split, bm.value, bn.value))
        res.sort()
        best = res[0]
        tot_auto += t_auto
        tot_best += best[0]
        alls = " ".join(f"{bm}x{bn}/s{sp}:{t:.0f}" for t, tl, sp, bm, bn in res[:6])
        print(f"{tag:9s} {M:8d} {cp:4d} {co:4d} {gf:6.2f} | {t_auto:8.1f} {gf / t_auto * 1e3:6.1f} "
              f"[{plan[0]},{plan[1]},{plan[2]},{plan[3]}] | {best[0]:8.1f} ({best[3]}x{best[4]},s{best[2]}) {gf / best[0] * 1e3:6.1f} TF | {alls}")
    print(f"TOTAL auto {tot_auto:.0f} us, best-single {tot_best:.0f} us")


if __name__ == "__main__":
    main()
