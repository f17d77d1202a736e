# Round 6: cost_volume_tile.hip -- cost-volume tests, the op-level leg, whole forwards with / without it (harness knob)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6m
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "cost_volume" > $O/tests_cv.txt 2>&1
tail -5 $O/tests_cv.txt
timeout 300 python bench.py --op-leg-only 2>$O/op_leg_err.txt | tail -1 > $O/op_leg.json
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6m/op_leg.json"))["roofline_hbm"]
print("op leg: frac", round(d["frac"],4), "us", round(d["us_per_forward"],1))
for k,v in d["per_kernel"].items(): print("  ", k, round(v["avg_us"],2))
PY
PWC_HARNESS=1 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exp_ab_tile.txt
import os, sys, statistics, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pwcnet_amd
from pwcnet_amd import _lib, weights as W
L = _lib.lib()
nets = {}
for mode in (0, 1):
    L.pwc_debug_cost_volume_tile(mode)
    net = pwcnet_amd.PWCDCNet(streams=1, use_plans=False)      # eager: the knob is read at launch time
    net.load_weights(W.init_weights(W.conv_specs(use_dc=False), seed=0))
    nets[mode] = net
im0 = torch.rand((8, 448, 1024, 3), device="cuda"); im1 = torch.rand((8, 448, 1024, 3), device="cuda")
times = {0: [], 1: []}
for rnd in range(9):
    for mode in (0, 1):
        L.pwc_debug_cost_volume_tile(mode)
        for _ in range(2): nets[mode](im0, im1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): nets[mode](im0, im1)
        e.record(); torch.cuda.synchronize()
        times[mode].append(s.elapsed_time(e) / 10)
for mode in (0, 1):
    print(f"tile kernel {'on ' if mode == 0 else 'off'}: median {statistics.median(times[mode]):.3f} ms per eager forward of 8 pairs (min {min(times[mode]):.3f})")
L.pwc_debug_cost_volume_tile(0); a = nets[0](im0, im1)[0]
L.pwc_debug_cost_volume_tile(1); b = nets[1](im0, im1)[0]
print("max |flow difference|", float((a - b).abs().max()))
PY
