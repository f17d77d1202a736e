# Round 6: shared frames of a sequence through the extractor once -- tests, timing of the sequence call against the two-tensor call
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6n
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "sequence or infer_continuous or pipeline or golden or batch8" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/exp_sequence.txt
import os, sys, statistics, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pwcnet_amd
from pwcnet_amd import weights as W
net = pwcnet_amd.PWCDCNet()
net.load_weights(W.init_weights(W.conv_specs(use_dc=False), seed=0))
for N in (8, 1):
    frames = torch.rand((N + 1, 448, 1024, 3), device="cuda")
    a0, a1 = frames[:-1], frames[1:]
    b0, b1 = a0.clone(), a1.clone()
    res = {}
    for name, (x, y) in (("sequence (N + 1 frames, one tensor)", (a0, a1)), ("two tensors (2 N images)", (b0, b1))):
        for _ in range(3): net(x, y)
        torch.cuda.synchronize()
        ts = []
        for rnd in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): out = net(x, y)[0]
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10)
        res[name] = out
        print(f"{N} pairs at 448 x 1024, {name}: median {statistics.median(ts):.3f} ms per forward (min {min(ts):.3f})")
    v = list(res.values())
    print("   max |flow difference|", float((v[0] - v[1]).abs().max()))
PY
