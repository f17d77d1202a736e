"""GPU parity tests, module and model level: the host classes that mirror the reference's
model.py / modules.py API against the CPU oracle (same seeded inputs and weights).

Tolerance for the end-to-end forward is the one BASELINE.json's north_star states:
max-abs 1e-3 per flow component on `flows_final` (pixels), i.e. 5e-5 on the px/20
pyramid flows.  Module-level tolerances are tighter (1e-5 relative).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (torch.cuda.is_available() is False)")
    import pwcnet_amd
    return pwcnet_amd


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return np.random.RandomState(seed).uniform(lo, hi, size=shape).astype(np.float32)


def gpu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, exp, rel=1e-5, floor=1e-6):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == exp.shape, (got.shape, exp.shape)
    tol = max(floor, rel * float(np.abs(exp).max()))
    err = float(np.abs(got - exp).max())
    assert err <= tol, f"max abs err {err:.3e} > tol {tol:.3e}"


def make_net(pa, use_dc=False, **kw):
    w = util.model_weights(use_dc)
    net = pa.PWCDCNet(use_dc=use_dc, **kw)
    net.load_weights(w)
    return net, w


# ------------------------------------------------------------------ modules (public API)
def test_extractor_module(pa):
    from pwcnet_amd.modules import VariableStore, variable_scope
    w = util.model_weights(False)
    store = VariableStore()
    for k, v in w.items():
        store.assign(k, v)
    im0, _ = util.images(2, 64, 128)
    with variable_scope("pwcdcnet", store=store):
        pyr = pa.FeaturePyramidExtractor_custom()(gpu(im0), reuse=False)
    exp = orc.OraclePWCDCNet(w).extractor(im0)
    assert [tuple(p.shape) for p in pyr] == [e.shape for e in exp]
    for p, e in zip(pyr, exp):
        close(p, e)


@pytest.mark.parametrize("use_dc", [False, True])
@pytest.mark.parametrize("level", [0, 2])
def test_estimator_module(pa, use_dc, level):
    from pwcnet_amd.modules import VariableStore, variable_scope
    from pwcnet_amd import weights as W
    w = util.model_weights(use_dc)
    store = VariableStore()
    for k, v in w.items():
        store.assign(k, v)
    C = W.pyramid_channels()[level]
    N, h, wd = 2, 6 * 2 ** level, 10 * 2 ** level
    cv, f0 = rnd((N, h, wd, 81), 40), rnd((N, h, wd, C), 41)
    fl = fu = None
    if level > 0:
        fl = rnd((N, h, wd, 2), 42)
        fu = rnd((N, h, wd, W.estimator_feature_channels(level - 1, use_dc)), 43)
    onet = orc.OraclePWCDCNet(w, use_dc=use_dc)
    e_flows, e_fup, e_featup = onet.estimator(level, cv, f0, fl, fu, False)
    g = [None if a is None else gpu(a) for a in (cv, f0, fl, fu)]
    with variable_scope("pwcdcnet", store=store):
        est = pa.OpticalFlowEstimator_custom(use_dc=use_dc, name=f"optflow_{level}")
        flows, fup, featup = est(*g)
        flows2, feats2 = est(*g, is_output=True)
    close(flows, e_flows)
    close(fup, e_fup)
    close(featup, e_featup)
    close(flows2, e_flows)
    _, e_feats = onet.estimator(level, cv, f0, fl, fu, True)
    close(feats2, e_feats)


def test_context_module(pa):
    from pwcnet_amd.modules import VariableStore, variable_scope
    w = util.model_weights(False)
    store = VariableStore()
    for k, v in w.items():
        store.assign(k, v)
    flows, feats = rnd((1, 40, 56, 2), 44), rnd((1, 40, 56, 32), 45)
    with variable_scope("pwcdcnet", store=store):
        out = pa.ContextNetwork(name="context")(gpu(flows), gpu(feats))
    close(out, orc.OraclePWCDCNet(w).context(flows, feats))


def test_lazy_variable_creation_names_and_count(pa):
    net = pa.PWCDCNet()
    im0, im1 = util.images(1, 64, 64)
    net(gpu(im0), gpu(im1))
    names = [v.name for v in net.vars]
    assert len(names) == 110 and sum(int(np.prod(v.shape)) for v in net.vars) == 5029868
    assert "pwcdcnet/fp_extractor/conv2d/kernel:0" in names and "pwcdcnet/fp_extractor/conv2d_17/bias:0" in names
    assert "pwcdcnet/optflow_4/conv2d_5/kernel:0" in names and "pwcdcnet/context/conv2d_6/kernel:0" in names
    assert not any("optflow_5" in n for n in names)          # never called at output_level 4
    with pytest.raises(AssertionError):
        pa.PWCDCNet(num_levels=4, output_level=4)            # reference model.py:82


# ------------------------------------------------------------------ end to end
@pytest.mark.parametrize("use_dc", [False, True])
def test_e2e_golden_64x128(pa, use_dc, golden_dir):
    g = np.load(os.path.join(golden_dir, f"e2e_64x128_dc{int(use_dc)}.npz"))
    net, _ = make_net(pa, use_dc)
    final, pyr = net(gpu(g["images_0"]), gpu(g["images_1"]))
    assert final.shape == (1, 64, 128, 2) and len(pyr) == 5
    err = float(np.abs(final.cpu().numpy() - g["flows_final"]).max())
    assert err <= 1e-3, err
    for l, p in enumerate(pyr):
        assert float(np.abs(p.cpu().numpy() - g[f"flows_{l}"]).max()) <= 5e-5
    assert float(np.abs(g["flows_final"]).max()) > 0.5      # the comparison is not vacuous


def test_e2e_batch_and_features_vs_oracle(pa):
    net, w = make_net(pa, False)
    im0, im1 = util.smooth_images(2, 128, 192, seed=11, shift=(-4, 2))
    final, pyr, feats = net(gpu(im0), gpu(im1), with_features=True)
    e_final, e_pyr, e_feats = orc.OraclePWCDCNet(w)(im0, im1, with_features=True)
    assert float(np.abs(final.cpu().numpy() - e_final).max()) <= 1e-3
    for p, e in zip(pyr, e_pyr):
        assert float(np.abs(p.cpu().numpy() - e).max()) <= 5e-5
    for f, e in zip(feats, e_feats):
        close(f, e)
    # EPE (reference losses.py:11-13) between HIP and oracle flows
    assert orc.epe(e_final, final.cpu().numpy()) <= 1e-4


def test_e2e_unfused_warp_and_nearest_variants(pa):
    im0, im1 = util.smooth_images(1, 64, 128, seed=12)
    net_f, w = make_net(pa, False, fuse_warp=True)
    net_u, _ = make_net(pa, False, fuse_warp=False)
    a, _ = net_f(gpu(im0), gpu(im1))
    b, _ = net_u(gpu(im0), gpu(im1))
    assert float((a - b).abs().max()) <= 2e-4
    net_n, _ = make_net(pa, False, warp_type="nearest")
    c, _ = net_n(gpu(im0), gpu(im1))
    e, _ = orc.OraclePWCDCNet(w, warp_type="nearest")(im0, im1)
    # nearest warping is discontinuous in the flow: allow isolated pixels to flip
    d = np.abs(c.cpu().numpy() - e)
    assert float(np.median(d)) <= 1e-4 and float((d > 1e-3).mean()) < 0.02


def test_e2e_deterministic_and_buffer_reuse(pa):
    net, _ = make_net(pa, False)
    im0, im1 = util.smooth_images(1, 64, 128, seed=13)
    a, pa_ = net(gpu(im0), gpu(im1))
    a = a.clone()
    other0, other1 = util.images(1, 64, 128, seed=99)
    net(gpu(other0), gpu(other1))
    b, _ = net(gpu(im0), gpu(im1))
    assert torch.equal(a, b)


def test_e2e_full_size_448x1024_vs_oracle(pa):
    """BASELINE config shape (one 448x1024 Sintel-shaped pair) against the oracle."""
    net, w = make_net(pa, False)
    im0, im1 = util.smooth_images(1, 448, 1024, seed=14, shift=(5, -3))
    final, pyr = net(gpu(im0), gpu(im1))
    e_final, e_pyr = orc.OraclePWCDCNet(w)(im0, im1)
    err = float(np.abs(final.cpu().numpy() - e_final).max())
    assert err <= 1e-3, err
    assert final.shape == (1, 448, 1024, 2)
    assert [tuple(p.shape[1:3]) for p in pyr] == [(7, 16), (14, 32), (28, 64), (56, 128), (112, 256)]


def test_e2e_batch8_matches_single_pair(pa):
    """size-independent property at the bench batch: pair i of a batch of 8 equals the
    same pair run alone (pairs are independent)."""
    net, _ = make_net(pa, False)
    im0, im1 = util.smooth_images(8, 128, 256, seed=15)
    f8, _ = net(gpu(im0), gpu(im1))
    f8 = f8.clone()
    for i in (0, 5, 7):
        f1, _ = net(gpu(im0[i:i + 1]), gpu(im1[i:i + 1]))
        assert float((f8[i:i + 1] - f1).abs().max()) <= 1e-5


def test_launch_plan_replay_matches_eager(pa):
    """the recorded launch plan (replayed from the 2nd call on, inputs patched by pointer)
    gives bit-identical results to the eager path, for fresh input tensors each call."""
    net_p, _ = make_net(pa, False)
    net_e, _ = make_net(pa, False, use_plans=False)
    for seed in (21, 22, 23):
        im0, im1 = util.smooth_images(2, 64, 128, seed=seed)
        a, pyr_a = net_p(gpu(im0), gpu(im1))
        b, pyr_b = net_e(gpu(im0), gpu(im1))
        assert torch.equal(a, b)
        for x, y in zip(pyr_a, pyr_b):
            assert torch.equal(x, y)
    assert len(net_p._plans) == 1 and len(net_e._plans) == 0
    # new weights invalidate the plan
    w2 = util.model_weights(False, seed=5)
    net_p.load_weights(w2)
    net_e.load_weights(w2)
    im0, im1 = util.smooth_images(2, 64, 128, seed=24)
    assert torch.equal(net_p(gpu(im0), gpu(im1))[0], net_e(gpu(im0), gpu(im1))[0])


def test_infer_cli_end_to_end(pa, tmp_path):
    """infer.py (counterpart of reference test.py): PNG pair -> crop to x64 -> /255 -> forward
    with weights restored from a TF-format bundle -> .flo + colour PNGs; checked against the
    oracle on the same cropped images."""
    import subprocess, sys
    from PIL import Image
    from pwcnet_amd import ckpt, flow_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(5)
    a = (rng.uniform(0, 255, size=(70, 140, 3))).astype(np.uint8)
    b = np.roll(a, 2, axis=1)
    Image.fromarray(a).save(tmp_path / "a.png")
    Image.fromarray(b).save(tmp_path / "b.png")
    w = util.model_weights(False)
    ckpt.save_weights(str(tmp_path / "m.ckpt"), w)
    out = subprocess.run([sys.executable, os.path.join(root, "infer.py"), "--input_images", str(tmp_path / "a.png"),
                          str(tmp_path / "b.png"), "--resume", str(tmp_path / "m.ckpt"), "--out", str(tmp_path / "o"),
                          "--time", "--iters", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Inference time" in out.stdout and "Figure saved" in out.stdout
    flow = flow_io.read_flo(str(tmp_path / "o" / "flow_final.flo"))
    assert flow.shape == (64, 128, 2)
    im = np.stack([a[:64, :128], b[:64, :128]]).astype(np.float32) / 255.0
    e_final, _ = orc.OraclePWCDCNet(w)(im[0:1], im[1:2])
    assert float(np.abs(flow - e_final[0]).max()) <= 1e-3
    assert all(os.path.exists(tmp_path / "o" / f"flow_level{l}.png") for l in range(5))


def test_infer_continuous_cli(pa, tmp_path):
    """infer_continuous.py (counterpart of reference test_continuous.py): a 4-frame sequence ->
    3 consecutive pairs in one batch -> per-pair .flo + montage; checked against the oracle."""
    import subprocess, sys
    from PIL import Image
    from pwcnet_amd import ckpt, flow_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(7)
    base = rng.uniform(0, 255, size=(66, 130, 3)).astype(np.uint8)
    seq = tmp_path / "seq"
    seq.mkdir()
    frames = [np.roll(base, 2 * i, axis=1) for i in range(4)]
    for i, f in enumerate(frames):
        Image.fromarray(f).save(seq / f"frame_{i:02d}.png")
    w = util.model_weights(False)
    ckpt.save_weights(str(tmp_path / "m.ckpt"), w)
    out = subprocess.run([sys.executable, os.path.join(root, "infer_continuous.py"), "-i", str(seq / "frame_*.png"),
                          "-r", str(tmp_path / "m.ckpt"), "--out", str(tmp_path / "o")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Figure saved" in out.stdout
    net = orc.OraclePWCDCNet(w)
    for i in range(3):
        flow = flow_io.read_flo(str(tmp_path / "o" / "seq" / f"frame_{i:02d}.flo"))
        im = np.stack([frames[i][:64, :128], frames[i + 1][:64, :128]]).astype(np.float32) / 255.0
        assert flow.shape == (64, 128, 2)
        assert float(np.abs(flow - net(im[0:1], im[1:2])[0][0]).max()) <= 1e-3
        assert os.path.exists(tmp_path / "o" / "seq" / f"frame_{i:02d}.png")
    assert not os.path.exists(tmp_path / "o" / "seq" / "frame_03.flo")


def test_evaluate_cli_end_to_end(pa, tmp_path):
    """evaluate.py: list of PNG pairs + ground-truth .flo -> sharded forward ->
    EPE; with the oracle's own flows as ground truth the EPE must vanish, and the saved flows
    must equal the oracle's."""
    import json, subprocess, sys
    from PIL import Image
    from pwcnet_amd import ckpt, flow_io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(11)
    w = util.model_weights(False)
    ckpt.save_weights(str(tmp_path / "m.ckpt"), w)
    net = orc.OraclePWCDCNet(w)
    lines, gts = [], []
    for i in range(3):
        a = rng.uniform(0, 255, size=(64, 128, 3)).astype(np.uint8)
        b = np.roll(a, i + 1, axis=1)
        Image.fromarray(a).save(tmp_path / f"a{i}.png")
        Image.fromarray(b).save(tmp_path / f"b{i}.png")
        im = np.stack([a, b]).astype(np.float32) / 255.0
        gt = net(im[0:1], im[1:2])[0][0]
        if i == 2:
            gt = gt + np.float32(0.5)                      # a known error on the last pair: |(.5,.5)| = 0.7071
        flow_io.write_flo(str(tmp_path / f"gt{i}.flo"), gt)
        gts.append(gt)
        lines.append(f"{tmp_path / f'a{i}.png'} {tmp_path / f'b{i}.png'} {tmp_path / f'gt{i}.flo'}")
    (tmp_path / "pairs.txt").write_text("\n".join(lines) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "evaluate.py"), "--list", str(tmp_path / "pairs.txt"),
                          "--resume", str(tmp_path / "m.ckpt"), "--batch", "2", "--save_dir", str(tmp_path / "o")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["pairs"] == 3 and res["n_gpus"] == 1
    assert max(res["per_pair_epe"][:2]) <= 1e-3 and abs(res["per_pair_epe"][2] - 0.5 * np.sqrt(2)) <= 1e-3
    assert abs(res["epe"] - 0.5 * np.sqrt(2) / 3) <= 1e-3
    pred = flow_io.read_flo(str(tmp_path / "o" / "000001.flo"))
    assert float(np.abs(pred - gts[1]).max()) <= 1e-3


@pytest.mark.parametrize("kw", [dict(search_range=2), dict(output_level=3), dict(output_level=2, search_range=3)])
def test_e2e_constructor_variants_vs_oracle(pa, kw):
    """non-default constructor kwargs of reference model.py:75-77: search_range changes the
    cost-volume depth (and every estimator's Cin), output_level where the context network
    and the final x2^(6-l) upsampling happen."""
    from pwcnet_amd import weights as W
    specs = W.conv_specs(search_range=kw.get("search_range", 4), output_level=kw.get("output_level", 4))
    w = W.randomize_biases(W.init_weights(specs, seed=2), seed=3)
    net = pa.PWCDCNet(**kw)
    net.load_weights(w)
    im0, im1 = util.smooth_images(1, 128, 128, seed=41)
    final, pyr = net(gpu(im0), gpu(im1))
    e_final, e_pyr = orc.OraclePWCDCNet(w, **kw)(im0, im1)
    assert final.shape == e_final.shape == (1, 128, 128, 2) and len(pyr) == len(e_pyr) == kw.get("output_level", 4) + 1
    assert float(np.abs(final.cpu().numpy() - e_final).max()) <= 1e-3
    assert len(net.vars) == 2 * len(specs)


def test_sizes_not_multiple_of_64_are_rejected_like_the_reference(pa):
    """reference test.py:13-17 crops inputs to multiples of 64 because the x2 upsampling of
    one level must match the next pyramid level (tf.concat would fail otherwise)."""
    net, _ = make_net(pa, False)
    im0, im1 = util.images(1, 100, 136)
    with pytest.raises(AssertionError, match="double in size"):
        net(gpu(im0), gpu(im1))


def test_winograd_and_direct_convs_agree_end_to_end(pa):
    """PWCDCNet routes its stride-1 convs to the Winograd kernel by default; the direct
    implicit-GEMM path (winograd=False) must give the same flow within fp32 noise."""
    net_w, w = make_net(pa, False)
    net_d, _ = make_net(pa, False, winograd=False)
    im0, im1 = util.smooth_images(2, 192, 256, seed=51, shift=(3, 1))
    a, pyr_a = net_w(gpu(im0), gpu(im1))
    b, pyr_b = net_d(gpu(im0), gpu(im1))
    assert float((a - b).abs().max()) <= 2e-4
    e, _ = orc.OraclePWCDCNet(w)(im0, im1)
    assert float(np.abs(a.cpu().numpy() - e).max()) <= 1e-3 and float(np.abs(b.cpu().numpy() - e).max()) <= 1e-3


def test_bench_json_contract(pa):
    """bench.py prints ONE JSON line with the driver's keys plus `roofline` and `cpu_baseline`
    (small configuration so that the test takes seconds)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--batch", "2", "--height", "128", "--width", "192", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "pairs/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32") and "fp16x2" in d["dtype"]          # the arithmetic is named, not just the tensor type
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # round 6: the timed region deals its K steps to the replicas of a ForwardPipeline (whole forwards overlap); the plain loop's
    # figure and the loop the per-kernel events were taken in stand beside it
    pl = d["config"]["pipeline"]
    assert pl["asked"] == 3 and 2 <= pl["depth"] <= 3 and pl["streams"] == "vetted"
    assert d["value_one_stream"] > 0 and abs(d["value_one_stream"] - 2 * 1e3 / d["ms_per_step_one_stream"]) <= 1e-6 * d["value_one_stream"]
    assert "ONE-STREAM loop" in d["roofline"]["measured"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_algorithmic", "frac_executed"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert r["frac"] == r["frac_algorithmic"] and abs(r["achieved"] - r["algorithmic_tflops"]) <= 1e-9
    for k in ("timed_region_ms", "per_rank_ms_per_step", "value_fp32_only", "range_status", "roofline_hbm"):
        assert k in d, k
    assert d["range_status"]["flags"] == 0 and len(d["per_rank_ms_per_step"]) == 1
    assert abs(d["timed_region_ms"] - 3 * d["ms_per_step"]) <= 1e-6 * d["timed_region_ms"]
    assert "traffic" in d["roofline_hbm"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0
    # the shared host makes the CPU figure swing -- repeat count, median, best and worst are in the line
    assert c["repeats"] >= 3 and len(c["seconds_per_pair"]) == c["repeats"]
    assert c["value_worst"] <= c["value_median"] <= c["value_best"] and c["value_worst"] <= c["value"] <= c["value_best"]
    assert d["parity"]["max_abs_flows_final"] <= d["parity"]["tolerance"]
    pm = d["parity_real_motion"]
    assert pm["f16x2_kept"] is True and pm["max_abs_flows_final"] <= pm["tolerance"]
    # counter traffic is printed only when its stamp matches the build and launch pattern of THIS run (the
    # committed passes are of the batch-8 448x1024 workload: this small configuration gets none and says nothing stale either)
    assert r["traffic"] is None and "traffic_stale" not in r


@pytest.mark.parametrize("gain,use_dc", [(1.35, False), (1.6, False), (1.25, True)])
def test_e2e_large_flows_vs_oracle(pa, gain, use_dc):
    """Kernels scaled up so that the network's flows reach several pixels (glorot weights give
    ~1 px): the warps then move features by whole pixels at every level and the 1e-3 px bound
    is tested at realistic magnitudes."""
    w = util.model_weights(use_dc, gain=gain)
    net = pa.PWCDCNet(use_dc=use_dc)
    net.load_weights(w)
    im0, im1 = util.smooth_images(2, 128, 192, seed=21, shift=(3, -5))
    final, pyr = net(gpu(im0), gpu(im1))
    e_final, e_pyr = orc.OraclePWCDCNet(w, use_dc=use_dc)(im0, im1)
    mag = float(np.abs(e_final).max())
    err = float(np.abs(final.cpu().numpy() - e_final).max())
    print(f"gain {gain} use_dc {use_dc}: max |flow| {mag:.3f} px, max abs err {err:.3e}")
    assert np.isfinite(mag) and mag >= 2.0, mag
    assert err <= 1e-3


def test_f4x4_layers_leave_the_flows_where_f2x2_puts_them(pa):
    """BASELINE configs[1] batch: the big level-4 convs run on Winograd F(4x4,3x3) (pwc_conv3x3_wino4_f32); with
    winograd4=False they run on F(2x2).  Flows of a few pixels (kernel gain 1.3): the two forwards must agree far
    inside the 1e-3 px bound, and pair 0 must meet the oracle."""
    w = util.model_weights(False, gain=1.3)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=95, shift=(-4, 3))
    net4 = pa.PWCDCNet(streams=1)
    net4.load_weights(w)
    net2 = pa.PWCDCNet(streams=1, winograd4=False)
    net2.load_weights(w)
    from pwcnet_amd.profiler import OpTimer
    t = OpTimer()
    with t:
        a, pyr_a = net4(gpu(im0), gpu(im1))
    assert any(k.startswith("conv3x3_wino4") for k in t.summary()), sorted(t.summary())
    b, pyr_b = net2(gpu(im0), gpu(im1))
    mag = float(b.abs().max())
    assert mag >= 1.0, mag
    assert float((a - b).abs().max()) <= 1e-4, float((a - b).abs().max())
    e_final, _ = orc.OraclePWCDCNet(w)(im0[:1], im1[:1])
    err = float(np.abs(a[:1].cpu().numpy() - e_final).max())
    print(f"F(4x4) layers: max |flow| {mag:.2f} px, vs F(2x2) {float((a - b).abs().max()):.2e}, vs oracle {err:.2e}")
    assert err <= 1e-3 / 3, err


def test_f16x2_layers_leave_the_flows_where_fp32_puts_them(pa):
    """BASELINE configs[1] batch: the big stride-1 convs run on conv3x3_h2 (fp32 operands as two-term fp16 splits on the F16
    matrix pipe, stream-K through a workspace); with f16x2=False they run on the fp32 Winograd kernels.  Flows of a few
    pixels (kernel gain 1.3): the two forwards must agree far inside the 1e-3 px bound, pair 0 must meet the oracle at least
    as well as the fp32 forward does, two f16x2 forwards must agree bitwise (the cut tiles of stream-K are summed in a fixed
    order), and the estimator input (cost volume ++ features ++ flow ++ upfeat) must be what the kernel's range expects."""
    w = util.model_weights(False, gain=1.3)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=95, shift=(-4, 3))
    net_h = pa.PWCDCNet(streams=1)
    net_h.load_weights(w)
    net_f = pa.PWCDCNet(streams=1, f16x2=False)
    net_f.load_weights(w)
    from pwcnet_amd.profiler import OpTimer
    t = OpTimer()
    with t:
        a, _ = net_h(gpu(im0), gpu(im1))
    names = sorted(t.summary())
    assert any(k.startswith("conv3x3_h2") for k in names), names
    a = a.clone()
    a2, _ = net_h(gpu(im0), gpu(im1))
    assert torch.equal(a, a2)
    t2 = OpTimer()
    with t2:
        b, _ = net_f(gpu(im0), gpu(im1))
    assert not any(k.startswith("conv3x3_h2") for k in t2.summary()), sorted(t2.summary())
    mag = float(b.abs().max())
    assert mag >= 1.0 and bool(torch.isfinite(a).all())
    assert float((a - b).abs().max()) <= 1e-4, float((a - b).abs().max())
    e_final, _ = orc.OraclePWCDCNet(w)(im0[:1], im1[:1])
    err_h = float(np.abs(a[:1].cpu().numpy() - e_final).max())
    err_f = float(np.abs(b[:1].cpu().numpy() - e_final).max())
    print(f"f16x2 layers: max |flow| {mag:.2f} px, vs fp32 kernels {float((a - b).abs().max()):.2e}, vs oracle {err_h:.2e} (fp32 kernels: {err_f:.2e})")
    assert err_h <= 1e-3 / 3 and err_h <= 2.0 * err_f + 1e-6, (err_h, err_f)


@pytest.mark.parametrize("mode,streams", [("sync", 1), ("lazy", 1), ("lazy", None)])
def test_out_of_range_frames_fall_back_to_fp32_kernels(pa, mode, streams):
    """the F16-pipe kernels split their operands into fp16 pairs and turn |x| >= 65504 into NaN; the reference
    is plain fp32 and nothing bounds its inputs (test.py:31 only divides by 255).  Frames scaled by 1e7: the model notices
    (status words of the kernels), warns, repeats the forward on the fp32 kernels INTO THE TENSORS IT RETURNED and stays on
    them -- range_check="sync" before the call returns, "lazy" at status() / the next call.  The result equals what a
    f16x2=False model gives, bit for bit; in-range frames never trip it and never leave the fast kernels."""
    import warnings
    w = util.model_weights(False)
    im0, im1 = util.smooth_images(4, 192, 256, seed=31, shift=(2, -1))     # (sub-batches of 2 still reach the F16-pipe correlation)
    ref = pa.PWCDCNet(f16x2=False, streams=streams)       # (the same sub-batches: the same tile plans, bit for bit)
    ref.load_weights(w)
    net = pa.PWCDCNet(range_check=mode, streams=streams, track_max=True)
    net.load_weights(w)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a, pyr = net(gpu(im0), gpu(im1))
        rep = net.status()
    assert rep["flags"] == 0 and rep["f16x2"] is True and rep["fallback_reason"] is None
    assert 0.05 <= rep["max_abs"] < 65504.0, rep
    b, _ = ref(gpu(im0), gpu(im1))
    assert float((a - b).abs().max()) <= 1e-4
    big0, big1 = gpu(im0 * 1e7), gpu(im1 * 1e7)        # (at this size the frames enter through fp32 kernels: the activations must overflow)
    want, want_pyr = ref(big0, big1)
    assert bool(torch.isfinite(want).all())
    with pytest.warns(RuntimeWarning, match="fp16's range"):       # (the flows came out non-finite)
        got, got_pyr = net(big0, big1)
        rep = net.status()
    assert rep["f16x2"] is False and "65504" in rep["fallback_reason"]
    torch.cuda.synchronize()

    def same(x, y):
        return torch.equal(x, y)
    assert same(got, want) and all(same(g, e) for g, e in zip(got_pyr, want_pyr))
    with warnings.catch_warnings():                              # the model stays on the fp32 kernels: no second warning
        warnings.simplefilter("error")
        again, _ = net(big0, big1)
        assert net.status()["flags"] == 0
    assert same(again, want)


def test_misaligned_frames_after_an_aligned_plan(pa):
    """the extractor's fused level-1 launch takes 16-byte aligned frames only.  A forward of aligned frames records a
    launch plan with it; the same shape arriving as a view that starts 4 bytes into an allocation must get a plan of its own
    (the per-layer path) and the same flows, not PWC_EALIGN out of a replayed launch."""
    w = util.model_weights(False)
    net = pa.PWCDCNet(streams=1)
    net.load_weights(w)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=41)
    a0, a1 = gpu(im0), gpu(im1)
    ref, _ = net(a0, a1)
    ref = ref.clone()

    def shifted(t):
        buf = torch.empty(t.numel() + 1, device="cuda")
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4
        return v
    got, _ = net(shifted(a0), shifted(a1))
    assert float((got - ref).abs().max()) <= 1e-4
    again, _ = net(a0, a1)
    assert torch.equal(again, ref)


def test_two_operand_first_conv_matches_the_concat_copy(pa):
    """Round 5: the estimator's first conv reads features_0 from the pyramid tensor (second operand pointer of the F16-pipe
    kernel) at the levels that run on it, instead of a copy in the estimator buffer.  Same channels in another order of
    16-channel stages: the flows agree to fp32 summation order with the copy form, and with the oracle."""
    w = util.model_weights(False, gain=1.3)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=96, shift=(3, -2))
    net_a = pa.PWCDCNet(streams=1)
    net_a.load_weights(w)
    net_b = pa.PWCDCNet(streams=1)
    net_b.load_weights(w)
    net_b.two_operand = False
    from pwcnet_amd.profiler import OpTimer
    a, _ = net_a(gpu(im0), gpu(im1))
    b, _ = net_b(gpu(im0), gpu(im1))
    lay4 = net_a._est_layout(4, 8, 112, 256, 32, True, list(range(32)))
    lay0 = net_a._est_layout(0, 8, 7, 16, 192, False, None)
    assert "f0" in lay4.external and lay4.n_phys == 128 and "f0" in lay0.segments
    assert "f0" in net_b._est_layout(4, 8, 112, 256, 32, True, list(range(32))).segments
    assert float((a - b).abs().max()) <= 2e-5, float((a - b).abs().max())
    e_final, _ = orc.OraclePWCDCNet(w)(im0[:1], im1[:1])
    assert float(np.abs(a[:1].cpu().numpy() - e_final).max()) <= 1e-3 / 3


def test_three_operand_first_conv_matches_the_estimator_buffer(pa):
    """Round 6: at the levels whose correlation runs on the row-walking F16-pipe kernel, the estimator input is three DENSE tensors
    -- [cv | flows_up_prev | 0] in 84-channel records (the correlation launch also writes the flow it read), features_0 in the
    pyramid tensor, features_up_prev -- and the first conv takes all three (pwc_conv3x3_h2_ex3_f32): `tf.concat` (reference
    modules.py:261-264) moves nothing and no producer writes a slice of a wider record.  Same channels in another order of
    16-channel stages: the flows agree to fp32 summation order with the one-buffer form, and with the oracle; which levels take
    the form is checked too (batch 8: levels 2-4; a single pair: level 4 only; never with dense connections)."""
    w = util.model_weights(False, gain=1.3)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=96, shift=(3, -2))
    net_a = pa.PWCDCNet(streams=1)
    net_a.load_weights(w)
    net_b = pa.PWCDCNet(streams=1)
    net_b.load_weights(w)
    net_b.three_operand = False
    fu = list(range(32))
    assert [net_a._three_operand_level(l, 8, 7 << l, 16 << l, c, fu) for l, c in ((1, 128), (2, 96), (3, 64), (4, 32))] == \
        [False, True, True, True]
    assert [net_a._three_operand_level(l, 1, 7 << l, 16 << l, c, fu) for l, c in ((2, 96), (3, 64), (4, 32))] == [False, False, True]
    assert not net_b._three_operand_level(4, 8, 112, 256, 32, fu)
    assert not pa.PWCDCNet(use_dc=True)._three_operand_level(4, 8, 112, 256, 32, fu)
    a, pa_ = net_a(gpu(im0), gpu(im1))
    b, pb_ = net_b(gpu(im0), gpu(im1))
    assert float((a - b).abs().max()) <= 2e-5, float((a - b).abs().max())
    for x, y in zip(pa_, pb_):
        assert float((x - y).abs().max()) <= 2e-6
    e_final, _ = orc.OraclePWCDCNet(w)(im0[:1], im1[:1])
    assert float(np.abs(a[:1].cpu().numpy() - e_final).max()) <= 1e-3 / 3
    # a single pair (level 4 only) and the replayed plan
    a1, _ = net_a(gpu(im0[:1]), gpu(im1[:1]))
    a1r, _ = net_a(gpu(im0[:1]), gpu(im1[:1]))
    assert torch.equal(a1, a1r)
    assert float(np.abs(a1.cpu().numpy() - e_final).max()) <= 1e-3 / 3


def test_channel_split_launches_do_not_change_the_flows(pa, monkeypatch):
    """The coarse estimator levels run their Winograd convs with the channel loop dealt to several workgroups
    (pwc_conv3x3_wino_split_f32); with the split disabled the forward must give the same flows up to fp32 summation order."""
    from pwcnet_amd import _lib
    L = _lib.lib()
    assert L.pwc_conv3x3_wino_split_plan(8, 14, 32, 256, 128, 1) > 1          # the 14x32 level of a batch of 8
    assert L.pwc_conv3x3_wino_split_plan(8, 112, 256, 160, 128, 1) == 1       # full launches are left alone
    w = util.model_weights(False, gain=1.2)
    im0, im1 = util.smooth_images(8, 448, 1024, seed=91, shift=(2, -3))
    net = pa.PWCDCNet()
    net.load_weights(w)
    final, pyr = net(gpu(im0), gpu(im1))
    ref = pa.PWCDCNet()
    ref.load_weights(w)
    for mod in ref._mods:                    # (a host-side routing attribute: the library reads no environment variable)
        mod.wino_channel_split = False
    rfinal, rpyr = ref(gpu(im0), gpu(im1))
    assert float((final - rfinal).abs().max()) <= 2e-5 * max(1.0, float(rfinal.abs().max()))
    for a, b in zip(pyr, rpyr):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


# ------------------------------------------------------------------ real motion
def motion_weights(use_dc, head_bias, gain=1.3):
    """Glorot weights (kernel gain 1.3: flows that vary by tens of pixels over the frame) with the bias of the coarsest flow
    head set to `head_bias` (px / 20): every level adds its residual to the upsampled flow of the level below
    (reference modules.py:275-277, :283 -- no x2 on the values), so the whole pyramid carries that translation and flows_final
    = 20 x it (model.py:127): Sintel-scale motion out of random weights, with warps of up to 20 / 2^(6-l) x it at level l."""
    w = util.model_weights(use_dc, gain=gain)
    w["pwcdcnet/optflow_0/conv2d_5/bias"] = np.asarray(head_bias, np.float32)
    return w


def float64_forward(w, use_dc, im0, im1):
    """flows_final of the float64 restatement (oracle/torch_ref.py, pinned against the C oracle in tests/test_oracle.py): the
    yardstick for what an fp32 forward of this depth can hold at a given flow magnitude."""
    from oracle import torch_ref as TR
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    w64 = {k: torch.from_numpy(v).to(torch.float64) for k, v in w.items()}
    with torch.no_grad():
        out = TR.TorchPWCDCNet(w64, use_dc=use_dc)(torch.from_numpy(im0).double(), torch.from_numpy(im1).double())
    return out[0].numpy()


@pytest.mark.parametrize("head_bias,floor_px,use_dc", [((5.2, -3.1), 100.0, False), ((15.5, -9.0), 300.0, False),
                                                        ((5.2, -3.1), 100.0, True), ((15.5, -9.0), 300.0, True)])
def test_e2e_real_motion_vs_oracle(pa, head_bias, floor_px, use_dc):
    """BASELINE configs[1] / configs[3] frame size, DEFAULT routing (F16-pipe kernels, the F(4x4) layer, stream-K), flows of
    >= 100 px and >= 300 px.  The oracle's flows_final must peak above `floor_px`; at 100 px the HIP forward must meet the fp32
    oracle within the north_star's 1e-3 px.  At 300-400 px two correct fp32 implementations of a 60-layer network need not agree
    to 1e-3 px (an ulp of a 400 px flow is 3e-5 px): there the float64 restatement is the yardstick -- the HIP forward must be
    within 1e-3 px of the TRUE flows or, where the fp32 reference arithmetic itself is not (its own distance e_o), within 1.5 e_o;
    and never more than 2.5e-3 px from the fp32 oracle.  The margins are printed (DESIGN.md section 4 quotes them); the same
    forward on the fp32 kernels is run beside it so that the split arithmetic's share of the error is visible."""
    w = motion_weights(use_dc, head_bias)
    im0, im1 = util.smooth_images(1, 448, 1024, seed=95, shift=(-4, 3))
    net = pa.PWCDCNet(use_dc=use_dc, range_check="sync")
    net.load_weights(w)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # no fallback: the F16-pipe kernels must carry this
        final, pyr = net(gpu(im0), gpu(im1))
    assert net.status()["f16x2"] is True
    net32 = pa.PWCDCNet(use_dc=use_dc, f16x2=False)
    net32.load_weights(w)
    final32, _ = net32(gpu(im0), gpu(im1))
    e_final, e_pyr = orc.OraclePWCDCNet(w, use_dc=use_dc)(im0, im1)
    mag = float(np.abs(e_final).max())
    got = final.cpu().numpy()
    err = float(np.abs(got - e_final).max())
    err32 = float(np.abs(final32.cpu().numpy() - e_final).max())
    msg = (f"real motion use_dc={use_dc}: oracle max |flow| {mag:.1f} px; HIP vs fp32 oracle {err:.3e} px = {err / 1e-3:.2f} of the "
           f"1e-3 budget (the same forward on the fp32 kernels: {err32:.3e})")
    assert np.isfinite(mag) and mag >= floor_px, mag
    if floor_px <= 100.0:
        print(msg)
        assert err <= 1e-3, err
        for g, e in zip(pyr, e_pyr):
            assert float(np.abs(g.cpu().numpy() - e).max()) <= 1e-3 / 20.0
        return
    truth = float64_forward(w, use_dc, im0, im1)
    e_h = float(np.abs(got - truth).max())
    e_o = float(np.abs(e_final - truth).max())
    e_32 = float(np.abs(final32.cpu().numpy() - truth).max())
    print(msg + f"; vs float64: HIP {e_h:.3e}, fp32 oracle {e_o:.3e}, HIP on fp32 kernels {e_32:.3e}")
    assert e_h <= max(1e-3, 1.5 * e_o), (e_h, e_o)
    assert err <= 2.5e-3, err


def test_activations_near_the_fp16_range_stay_on_the_fast_kernels(pa):
    """walk the range guard of the split arithmetic (|x| < 65504) without crossing it.  Frames scaled by 1.5e4:
    the operands of the F16-pipe kernels reach 1e4 ... 3e4 (frames 1.5e4, features 2e4, the cost volume 1e4) and the random-init
    network is in its chaotic regime (flows of 1e4 px and more -- two fp32 forwards differ by 1e-4 of that).  The F16-pipe forward
    must (a) raise no flag and stay on the fast kernels, (b) report a largest operand inside [1e4, 65504), (c) be as close to the
    oracle as the fp32-kernel forward is, relative to the flow magnitude (an absolute 1e-3 px means nothing there)."""
    import warnings
    w = util.model_weights(False)
    im0, im1 = util.smooth_images(1, 448, 1024, seed=95, shift=(-4, 3))
    im0, im1 = im0 * 1.5e4, im1 * 1.5e4
    net = pa.PWCDCNet(range_check="sync", track_max=True)
    net.load_weights(w)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        final, _ = net(gpu(im0), gpu(im1))
        rep = net.status()
    assert rep["flags"] == 0 and rep["f16x2"] is True, rep
    assert 1e4 <= rep["max_abs"] < 65504.0, rep
    net32 = pa.PWCDCNet(f16x2=False)
    net32.load_weights(w)
    final32, _ = net32(gpu(im0), gpu(im1))
    e_final, _ = orc.OraclePWCDCNet(w)(im0, im1)
    mag = float(np.abs(e_final).max())
    err = float(np.abs(final.cpu().numpy() - e_final).max())
    err32 = float(np.abs(final32.cpu().numpy() - e_final).max())
    print(f"near the fp16 range: largest F16-pipe operand {rep['max_abs']:.0f} ({rep['max_abs'] / 65504:.2f} of the range), "
          f"max |flow| {mag:.0f} px, err {err:.3e} ({err / mag:.1e} relative; fp32 kernels: {err32:.3e})")
    assert np.isfinite(mag) and bool(torch.isfinite(final).all())
    assert err <= 3.0 * err32 + 1e-6 * mag and err <= 2e-3 * mag, (err, err32, mag)


# ------------------------------------------------------------------ the range check in a pipelined loop
def test_lazy_range_check_maps_each_flag_to_its_forward(pa):
    """range_check="lazy" in a pipelined loop: forwards 0-1 in range, forward 2 overflows, forwards 3-4 are issued behind it
    before anybody looks.  Every forward has a status slot of its own; the words are sticky, so the first slot that shows the
    flag names the culprit: forwards 2, 3, 4 are repeated on the fp32 kernels into the tensors they returned, forwards 0-1 are
    left alone.  A forward whose INPUT tensors were refilled in place since (torch's version counter) is not repeated from
    them: its outputs become NaN, never flows of other frames."""
    w = util.model_weights(False)
    im0, im1 = util.smooth_images(2, 192, 256, seed=31, shift=(2, -1))
    ref = pa.PWCDCNet(f16x2=False)
    ref.load_weights(w)
    net = pa.PWCDCNet(range_check="lazy")
    net.load_weights(w)
    frames = [(gpu(im0 * s), gpu(im1 * s)) for s in (1.0, 0.5, 1e7, 2.0, 0.25)]
    want = [ref(a, b)[0].clone() for a, b in frames]
    torch.cuda.synchronize()
    from pwcnet_amd import _lib
    with pytest.warns(RuntimeWarning, match="issued behind it"):
        outs = [net(a, b)[0] for a, b in frames[:2]]
        # a device-side spin (~50 ms) in front of forward 2: none of the status copies behind it can arrive before status()
        _lib.check(_lib.lib().pwc_device_spin(5_000_000, _lib.current_stream()))
        outs.append(net(*frames[2])[0])
        outs.append(net(*frames[3])[0])
        # forward 4's frames are refilled in place before anybody has looked at the status words
        a4, b4 = frames[4][0].clone(), frames[4][1].clone()
        outs.append(net(a4, b4)[0])
        a4.mul_(3.0)
        rep = net.status()
    assert rep["f16x2"] is False
    torch.cuda.synchronize()
    for i in (0, 1):
        assert float((outs[i] - want[i]).abs().max()) <= 1e-4          # (F16-pipe results, untouched)
    for i in (2, 3):
        assert torch.equal(outs[i], want[i]), i                        # repeated on the fp32 kernels
    assert bool(torch.isnan(outs[4]).all())                            # inputs modified since: invalidated, not recomputed


@pytest.mark.parametrize("streams", [1, 2])
def test_fallback_refreshes_the_returned_pyramid_too(pa, streams):
    """with_features=True hands out pyramid_0, which came from the overflowing F16-pipe extractor -- the fp32 repeat
    must refresh it along with the flows; and the repeat must also work when the batch runs as sub-batches on side streams."""
    w = util.model_weights(False)
    im0, im1 = util.smooth_images(4, 192, 256, seed=33, shift=(1, 2))
    big0, big1 = gpu(im0 * 1e7), gpu(im1 * 1e7)
    ref = pa.PWCDCNet(f16x2=False, streams=streams)
    ref.load_weights(w)
    net = pa.PWCDCNet(range_check="sync", streams=streams)
    net.load_weights(w)
    if streams == 1:
        want = ref(big0, big1, with_features=True)
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            got = net(big0, big1, with_features=True)
        torch.cuda.synchronize()
        assert len(got) == 3 and torch.equal(got[0], want[0])
        for g, e in zip(got[2], want[2]):
            assert bool(torch.isfinite(g).all()) and torch.equal(g, e)
    else:
        want = ref(big0, big1)
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            got = net(big0, big1)
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and all(torch.equal(g, e) for g, e in zip(got[1], want[1]))


# ------------------------------------------------------------------ round 6: whole forwards in flight (pwcnet_amd.ForwardPipeline)
@pytest.mark.parametrize("depth", [2, 3])
def test_pipeline_matches_single_stream(pa, depth):
    """Consecutive forwards dealt to `depth` replicas on HIP streams of their own (the launch-bound coarse levels of one forward
    under the matrix-bound launches of another) give, ticket by ticket, the tensors the plain PWCDCNet loop gives -- bit for bit:
    same kernels, same launch plans, nothing shared between replicas but the weights.  Seven DIFFERENT batches, so that a ticket
    handed the wrong replica's tensors, or a replica reading frames of another submission, cannot pass; the frames of a
    submission are dropped by the caller right behind submit() (the allocator may not recycle them under the lane)."""
    from pwcnet_amd.pipeline import ForwardPipeline
    w = util.model_weights(False)
    net = pa.PWCDCNet()
    net.load_weights(w)
    pipe = ForwardPipeline(depth=depth)
    pipe.load_weights(w)
    frames = [util.smooth_images(2, 128, 192, seed=100 + i, shift=(1 + i % 3, -(i % 2))) for i in range(7)]
    want = []
    for im0, im1 in frames:
        f, pyr = net(gpu(im0), gpu(im1))
        want.append((f.clone(), [p.clone() for p in pyr]))
    torch.cuda.synchronize()
    tickets = []
    for im0, im1 in frames:
        a, b = gpu(im0), gpu(im1)
        tickets.append(pipe.submit(a, b))
        del a, b
        poison = [torch.full((2, 128, 192, 3), float("nan"), device="cuda") for _ in range(4)]      # (a recycled block would be poisoned here)
        del poison
    assert pipe.effective_depth >= 1
    for tk, (f, pyr) in zip(tickets, want):
        got_f, got_pyr = tk.result()
        torch.cuda.current_stream().synchronize()
        assert torch.equal(got_f, f)
        assert len(got_pyr) == len(pyr) and all(torch.equal(g, e) for g, e in zip(got_pyr, pyr))
    rep = pipe.synchronize()
    assert rep["flags"] == 0 and rep["f16x2"] is True and len(rep["replicas"]) == depth
    # the drop-in form: submit + result
    f, _ = pipe(gpu(frames[0][0]), gpu(frames[0][1]))
    torch.cuda.synchronize()
    assert torch.equal(f, want[0][0])
    # against the oracle too (the tolerance of the forward)
    e, _ = orc.OraclePWCDCNet(w)(*frames[0])
    assert float(np.abs(f.cpu().numpy() - e).max()) <= 1e-3


def test_pipeline_falls_back_to_fp32_like_the_model(pa):
    """A replica whose forward leaves fp16's range repeats it on the fp32 kernels into the tensors its ticket holds, exactly as
    PWCDCNet does (every replica keeps its own status words)."""
    import warnings
    from pwcnet_amd.pipeline import ForwardPipeline
    w = util.model_weights(False)
    im0, im1 = util.smooth_images(4, 192, 256, seed=31, shift=(2, -1))
    ref = pa.PWCDCNet(f16x2=False)
    ref.load_weights(w)
    pipe = ForwardPipeline(depth=2)
    pipe.load_weights(w)
    big0, big1 = gpu(im0 * 1e7), gpu(im1 * 1e7)
    want, _ = ref(big0, big1)
    ok_want, _ = ref(gpu(im0), gpu(im1))
    t_ok = pipe.submit(gpu(im0), gpu(im1))
    t_big = pipe.submit(big0, big1)
    with pytest.warns(RuntimeWarning, match="fp16's range"):
        rep = pipe.synchronize()
    assert rep["f16x2"] is False and sum(1 for r in rep["replicas"] if not r["f16x2"]) == 1
    torch.cuda.synchronize()
    assert torch.equal(t_big.result()[0], want)
    assert float((t_ok.result()[0] - ok_want).abs().max()) <= 1e-4
    torch.cuda.synchronize()


def test_pipeline_at_bench_size_with_stream_k_launches_in_flight(pa):
    """The bench shape (8 x 448 x 1024): here the big layers run as stream-K launches (one workgroup per CU, partial sums exchanged
    through a workspace per stream) and three of them can be in flight at once.  24 forwards over 3 lanes, two different batches
    alternating: every ticket equals the one-stream result of ITS batch bit for bit, no status flag (a stream-K wait that ran out
    would raise PWC_STATUS_STREAMK_TIMEOUT and move the replica to the fp32 kernels)."""
    import warnings
    from pwcnet_amd.pipeline import ForwardPipeline
    w = util.model_weights(False)
    net = pa.PWCDCNet()
    net.load_weights(w)
    pairs = [tuple(gpu(x) for x in util.smooth_images(8, 448, 1024, seed=300 + i, shift=(2 + i, -1 - i))) for i in range(2)]
    want = [net(a, b)[0].clone() for a, b in pairs]
    torch.cuda.synchronize()
    pipe = ForwardPipeline(depth=3)
    pipe.load_weights(w)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tickets = [pipe.submit(*pairs[i % 2]) for i in range(24)]
        rep = pipe.synchronize()
    assert rep["flags"] == 0 and rep["f16x2"] is True
    for i, tk in enumerate(tickets):
        assert torch.equal(tk.result()[0], want[i % 2]), i
    torch.cuda.synchronize()


# ------------------------------------------------------------------ round 6: a sequence's shared frames go through the extractor once
@pytest.mark.parametrize("use_dc", [False, True])
def test_sequence_frames_share_the_extractor(pa, use_dc):
    """net(frames[:-1], frames[1:]) on ONE tensor of N + 1 frames (what infer_continuous.py feeds; reference test_continuous.py:55-62
    runs the pairs one by one): the model notices that images_1 is images_0 one frame on and runs the extractor over N + 1 frames
    instead of 2 N.  Same flows as the two-tensor call (the extractor's tile plans see another image count: equal to rounding, not
    bit for bit) and as the oracle pair by pair; a launch plan recorded on a sequence is not replayed on two separate tensors of the
    same shape (and vice versa); with_features returns the first frames' pyramid."""
    w = util.model_weights(use_dc)
    net = pa.PWCDCNet(use_dc=use_dc)
    net.load_weights(w)
    ims = [util.smooth_images(1, 128, 192, seed=400 + i, shift=(2, -1))[0][0] for i in range(5)]
    frames = gpu(np.stack(ims))
    a0, a1 = frames[:-1], frames[1:]
    from pwcnet_amd.modules import as_view
    assert net._frames_shared(as_view(a0)[0], as_view(a1)[0]) and not net._frames_shared(as_view(a0)[0], as_view(a0.clone())[0])
    seq_f, seq_pyr = net(a0, a1)                                   # records the sequence plan
    seq_f2, _ = net(a0, a1)                                        # ... and replays it
    b0, b1 = a0.clone(), a1.clone()
    two_f, two_pyr = net(b0, b1)                                   # separate tensors: the 2 N path, its own plan
    seq_f3, _ = net(a0, a1)
    torch.cuda.synchronize()
    assert torch.equal(seq_f, seq_f2) and torch.equal(seq_f, seq_f3)
    assert float((seq_f - two_f).abs().max()) <= 2e-5
    for s_, t_ in zip(seq_pyr, two_pyr):
        assert float((s_ - t_).abs().max()) <= 2e-6
    e, _ = orc.OraclePWCDCNet(w, use_dc=use_dc)(np.stack(ims[:-1]), np.stack(ims[1:]))
    assert float(np.abs(seq_f.cpu().numpy() - e).max()) <= 1e-3
    # other frames in the same storage: the replayed sequence plan follows the pointers
    frames2 = gpu(np.stack(ims[::-1]))
    r_f, _ = net(frames2[:-1], frames2[1:])
    e2, _ = orc.OraclePWCDCNet(w, use_dc=use_dc)(np.stack(ims[::-1][:-1]), np.stack(ims[::-1][1:]))
    assert float(np.abs(r_f.cpu().numpy() - e2).max()) <= 1e-3
    _, _, feats = net(a0, a1, with_features=True)
    _, _, feats2 = net(b0, b1, with_features=True)
    for x, y in zip(feats, feats2):
        assert x.shape == y.shape and float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))
    assert net.status()["flags"] == 0
