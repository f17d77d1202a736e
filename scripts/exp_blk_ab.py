"""A/B of the small-level correlation launches in ONE process (round 5): coarse fp32 kernel / row-walking F16 kernel / block
kernel with 1 or 3 window block rows per workgroup, on rotating operand sets (cold: > 256 MB in rotation; warm: one set).
Prints the average time of a launch from one pair of events around a captured chain of launches (graph replay: no host in the loop)."""
import os
os.environ["PWC_HARNESS"] = "1"   # libpwc_hip_harness.so: the pwc_debug_* knobs exist only there (build it here first:
                                    # PWC_HARNESS=1 python -c 'from pwcnet_amd import _lib; _lib.build_library()')
import sys, ctypes, torch
sys.path.insert(0, ".")
import pwcnet_amd as pa
from pwcnet_amd import _lib
from pwcnet_amd.modules import View, sub_view

L = _lib.lib()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LEVELS = [(NB, 7, 16, 192, False, 288), (NB, 14, 32, 128, True, 256), (NB, 28, 64, 96, True, 128), (NB, 56, 128, 64, True, 128)]


def sets(N, H, W, C, ecs, nsets):
    out = []
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for _ in range(nsets):
        f0 = torch.randn((N, H, W, C), generator=g, device="cuda")
        f1 = torch.randn((N, H, W, C), generator=g, device="cuda")
        fl = torch.randn((N, H, W, 2), generator=g, device="cuda") * 0.6
        E = torch.zeros((N, H, W, ecs), device="cuda")
        out.append((f0, f1, fl, E))
    return out


def run(kind, layer, s, with_flow, copy):
    f0, f1, fl, E = s
    N, H, W, C = f0.shape
    ecs = E.shape[3]
    Ev = View(E.data_ptr(), ecs, N, H, W, ecs)
    v0, v1 = View(f0.data_ptr(), C, N, H, W, C), View(f1.data_ptr(), C, N, H, W, C)
    fv = View(fl.data_ptr(), 2, N, H, W, 2) if with_flow else None
    cpy = sub_view(Ev, 84, C) if copy else None
    if kind == "coarse":
        layer._run(v0, v1, sub_view(Ev, 0, 81), flow=fv, flow_scale=5.0, f0_copy=cpy, coarse=True)
    elif kind == "h2":
        layer._run(v0, v1, sub_view(Ev, 0, 81), flow=fv, flow_scale=5.0, f0_copy=cpy, concat=True, out_pad_writable=True)
    else:
        layer._run(v0, v1, sub_view(Ev, 0, 81), flow=fv, flow_scale=5.0, f0_copy=cpy, concat=True, out_pad_writable=True, blk=True)


def main():
    layer = pa.CostVolumeLayer(4)
    layer.f16x2 = True
    layer.COARSE_MAX_PIXELS = 1 << 30
    for (N, H, W, C, with_flow, ecs) in LEVELS:
        per_set = N * H * W * (2 * C + 2 + ecs) * 4
        for mode, nsets in (("cold", max(2, (300 << 20) // per_set + 1)), ("warm", 1)):
            ss = sets(N, H, W, C, ecs, min(nsets, 400))
            copy = ecs >= 84 + C
            line = f"{N}x{H}x{W}x{C} {mode:4s} sets={len(ss):3d}"
            for kind in (["coarse"] if C >= 128 else []) + (["h2"] if C <= 96 else []) + ["blk1", "blk3"]:
                if kind.startswith("blk"):
                    L.pwc_debug_cost_volume_blk_rows(int(kind[3]))
                best = []
                reps = max(1, 200 // len(ss))
                run(kind, layer, ss[0], with_flow, copy)          # attributes, lazy init outside the capture
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(reps):
                        for s in ss:
                            run(kind, layer, s, with_flow, copy)
                n = reps * len(ss)
                graph.replay()
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best.append(e0.elapsed_time(e1) * 1e3 / n)
                del graph
                best.sort()
                line += f"  {kind} {best[0]:6.2f}/{best[2]:6.2f}"
            print(line, flush=True)
            del ss
            torch.cuda.empty_cache()
    L.pwc_debug_cost_volume_blk_rows(0)


main()
