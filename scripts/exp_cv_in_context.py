"""Why do the correlation launches take 15-20 % longer inside the forward than in bench.py's op-level leg?  

Suspect: the clock.  Inside the forward a correlation launch follows a chain of conv3x3_h2 launches that hold the chip at its
power cap (sclk ~1700 MHz, profiles/r05_exp_h2_micro_clock_power.txt); in the op leg it runs among its own kind (memory-bound,
far below the cap, sclk ~2400 MHz).  A latency-bound kernel's time scales with 1 / sclk.

Per pyramid level 4 / 3 / 2 of the batch-8 workload, three captured chains (graph replays, one event pair around each):
    A = n x [cost volume]                    (the op leg's figure)
    C = n x [conv3x3_h2 128 -> 128 at 8 x 112 x 256]
    B = n x [conv3x3_h2 ; cost volume]
"in context" = (B - C) / n: what a correlation launch costs right behind a matrix-bound launch.  Each with iid N(0, 3^2) px
flows (op leg) and with zero flows (the random-init forward).  Run under scripts/gpu_clock_log.sh for the sclk / power trace.
usage: python scripts/exp_cv_in_context.py [batch]"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwcnet_amd
from pwcnet_amd import _lib, modules as M
from pwcnet_amd.weights import pyramid_channels

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 448, 1024
dev = torch.device("cuda:0")
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
net = pwcnet_amd.PWCDCNet()
g = torch.Generator(device=dev); g.manual_seed(5)

# the matrix-bound neighbour: 128 -> 128 at 8 x 112 x 256 on conv3x3_h2 (stream-K workspace)
ci = co = 128
hx, wx = 112, 256
xin = torch.randn((B, hx, wx, ci), generator=g, device=dev)
yout = torch.empty((B, hx, wx, co), device=dev)
wk = torch.randn((3, 3, ci, co), generator=g, device=dev) * 0.03
bias = torch.zeros(co, device=dev)
packed = torch.empty(L.pwc_conv3x3_h2_packed_floats(ci, co), device=dev)
_lib.check(L.pwc_conv3x3_h2_pack_f32(p(wk), None, ci, ci, co, p(packed), None))
wsf = L.pwc_conv3x3_h2_workspace_floats(B, hx, wx, ci, co, 1)
ws = torch.full((max(int(wsf), 1),), -1, dtype=torch.int32, device=dev).view(torch.float32)


def conv():
    _lib.check(L.pwc_conv3x3_h2_f32(p(xin), ci, p(packed), p(bias), p(yout), co, B, hx, wx, ci, co, 1, 1, 0.1,
                                    p(ws) if wsf else None, ws.numel() if wsf else 0, _lib.current_stream()))


def chain_us(fn, n, reps=7):
    for _ in range(2):
        fn(0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for r in range(n):
            fn(r)
    graph.replay()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); graph.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


chans = pyramid_channels(net.num_levels)
n = 24
print(f"# batch {B}, {H}x{W}; chains of {n} launches, median of 7 graph replays; us per launch")
print(f"{'level':>5s} {'flows':>10s} {'A alone':>9s} {'C conv':>9s} {'B conv+cv':>10s} {'in context (B-C)':>17s} {'ratio':>6s}")
for l in (4, 3, 2):
    h, w, C = H >> (net.num_levels - l), W >> (net.num_levels - l), chans[l]
    lay = net._est_layout(l, B, h, w, C, True, list(range(32)))
    est_cs = lay.n_phys
    nsets = min(64, max(2, int(300e6 // (4 * B * h * w * (3 * C + 2 + est_cs))) + 1))
    for kind in ("N(0,3^2)", "zero"):
        sets = []
        for _ in range(nsets):
            f0 = torch.randn((B, h, w, C), generator=g, device=dev)
            f1 = torch.randn((B, h, w, C), generator=g, device=dev)
            fl = (torch.randn((B, h, w, 2), generator=g, device=dev) * (3.0 / net.scales[l])) if kind != "zero" else torch.zeros((B, h, w, 2), device=dev)
            E = torch.zeros((B, h, w, est_cs), device=dev)
            sets.append((f0, f1, fl, E))

        def cv(r):
            f0, f1, fl, E = sets[r % nsets]
            v0 = M.View(f0.data_ptr(), C, B, h, w, C)
            v1 = M.View(f1.data_ptr(), C, B, h, w, C)
            Ev = M.View(E.data_ptr(), est_cs, B, h, w, est_cs)
            cv_out = M.sub_view(Ev, lay.offset("cv"), 81)
            f0_dst = M.sub_view(Ev, lay.offset("f0"), C) if "f0" in lay.segments else None
            net._corr_level(l, v0, v1, M.View(fl.data_ptr(), 2, B, h, w, 2), cv_out, f0_dst, Ev, dev)

        a = chain_us(lambda r: cv(r), n) / n
        c = chain_us(lambda r: conv(), n) / n
        b = chain_us(lambda r: (conv(), cv(r)), n) / n
        print(f"{l:5d} {kind:>10s} {a:9.2f} {c:9.2f} {b:10.2f} {b - c:17.2f} {(b - c) / a:6.2f}")
        del sets
