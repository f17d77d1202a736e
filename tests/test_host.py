"""CPU tests of the host logic: C-ABI surface, variable inventory, channel layouts,
checkpoint bundle reader/writer, pair sharding (incl. a world-size-2 gloo run)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from pwcnet_amd import _lib, ckpt, sharding
from pwcnet_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """Every function include/pwc_hip.h declares is exported by the built .so and bound
    in _lib.SIGNATURES (no compute call is made: there is no GPU here)."""
    header = open(os.path.join(ROOT, "include", "pwc_hip.h")).read()
    declared = set(re.findall(r"\b(pwc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name)
    assert L.pwc_version() >= 100
    assert L.pwc_error_string(-1).decode().startswith("invalid argument")
    # pure host-side helpers are callable without a GPU
    assert L.pwc_conv3x3_packed_floats(160, 128) == 9 * 160 * 128
    plan = (ctypes.c_int * 4)()
    assert L.pwc_conv3x3_plan(8 * 112 * 256, 128, 160, plan) == 0
    bm, bn = ctypes.c_int(), ctypes.c_int()
    assert L.pwc_conv3x3_tile_shape(plan[0], bm, bn) == 0 and (bm.value, bn.value) == (128, 128)
    assert plan[1] == -1 and plan[2] == 8 * 112 * 256 and plan[3] == 1
    # mid-size M: the largest tile that still gives >= 384 workgroups (measured rule)
    assert L.pwc_conv3x3_plan(8 * 28 * 64, 128, 224, plan) == 0
    assert L.pwc_conv3x3_tile_shape(plan[0], bm, bn) == 0 and (bm.value, bn.value, plan[3]) == (32, 128, 1)
    assert L.pwc_conv3x3_plan(8 * 7 * 16, 128, 288, plan) == 0 and plan[3] in (3, 9)   # tiny M: tap split
    assert L.pwc_conv3x3_workspace_floats(100, 20) == 9 * 100 * 32
    # argument validation happens before any launch
    assert L.pwc_cost_volume_f32(None, 4, None, 4, None, 81, 1, 4, 4, 4, 4, 0.1, None) == -1
    assert L.pwc_conv3x3_f32(None, 16, None, None, None, 16, 1, 4, 4, 16, 16, 1, 1, 1, 0.1, -1, 0, None, 0, None) == -1


def test_hot_kernels_do_not_spill_to_scratch(tmp_path):
    """A regression guard that needs no GPU (round 6: a third buffer resource in conv3x3_h2_kernel took six more SGPRs, the
    128-cout variants went over 256 VGPRs into scratch and every launch ran 2 x slower -- green tests, half the throughput).
    The code objects inside the built libpwc_hip.so say what each kernel takes: the matrix-pipe kernels that carry the forward
    must have NO private segment (scratch); the C = 96 correlation variants are allowed their few spilled registers."""
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not found")
    _lib.lib()
    import shutil
    so = shutil.copy(_lib.LIB_PATH, str(tmp_path / "lib.so"))
    subprocess.check_call([objdump, "--offloading", so], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    scratch = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", str(tmp_path / f)], capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".private_segment_fixed_size:") and name:
                scratch[name] = int(line.split(":", 1)[1])
                name = None
    hot = [k for k in scratch if re.search(r"conv3x3_(h2|w32|t32|sk|skp|c16pair)_kernel|cost_volume_(h2|blk)_kernel", k)]
    assert len(hot) >= 30, sorted(scratch)
    bad = {k: v for k, v in scratch.items() if k in hot and v > (48 if "cost_volume_h2_kernelILi6" in k else 0)}
    assert not bad, bad


def test_production_library_has_no_debug_knobs():
    """VERDICT r5 item 5 / SURVEY 8(b) "no global mutable state": the production libpwc_hip.so exports no pwc_debug_* symbol
    (the tile-pinning / ablation knobs of the A/B scripts live in libpwc_hip_harness.so, built with -DPWC_HARNESS only when a
    script asks, PWC_HARNESS=1), the header declares none, and the tile of the small-launch conv is an ARGUMENT
    (pwc_conv3x3_sk_variant_f32) validated before any launch."""
    assert not _lib.HARNESS and _lib.LIB_PATH.endswith("libpwc_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert any(n == "pwc_version" for n in exported)
    assert [n for n in exported if "debug" in n.lower()] == []
    header = open(os.path.join(ROOT, "include", "pwc_hip.h")).read()
    assert not re.search(r"\bpwc_debug_[a-z0-9_]+\s*\(", header)
    assert not any("debug" in n for n in _lib.SIGNATURES)
    L = _lib.lib()
    for name in _lib.HARNESS_SIGNATURES_NAMES:
        assert not hasattr(L, name), name
    al = ctypes.c_void_p(4096)      # (an aligned non-null address: argument checks only, nothing is launched)
    v = L.pwc_conv3x3_sk_variant_f32
    assert v(al, 64, al, al, al, 32, 1, 8, 8, 64, 32, 1, 1, 1, 0.1, 0, None) == -1       # tile 0 is pwc_conv3x3_sk_f32's business
    assert v(al, 64, al, al, al, 32, 1, 8, 8, 64, 32, 1, 1, 1, 0.1, 13, None) == -1      # no such tile
    assert v(al, 64, al, al, al, 48, 1, 8, 8, 64, 48, 1, 1, 1, 0.1, 22, None) == -4      # x2 tiles need Cout % 32 == 0
    assert v(al, 64, al, al, al, 32, 1, 8, 8, 64, 32, 1, 2, 1, 0.1, 31, None) == -4      # the LDS-patch form takes no dilation
    assert v(al, 320, al, al, al, 32, 1, 8, 8, 320, 32, 1, 1, 1, 0.1, 41, None) == -4    # ... and at most 288 input channels


def test_round5_routing_rules_are_host_logic():
    """Which kernel takes which launch is decided by pure host functions of the library (no GPU needed): the small-launch conv
    (up to 1e8 multiply-adds and 4096 output pixels; stride-2 / thin layers beyond), the weights-stationary thin-input conv
    (16 input channels, 32 output channels, at least 256 tiles), the block correlation (alignment only: the pixel limit is the
    caller's), packed sizes, and argument validation before any launch."""
    L = _lib.lib()
    sk = L.pwc_conv3x3_sk_supported
    # BASELINE configs[1], batch 8: estimator levels 0-1, the extractor's last levels (both frames stacked: N = 16)
    assert sk(8, 7, 16, 288, 128, 1, 1) == 1 and sk(8, 14, 32, 128, 128, 1, 1) == 1 and sk(16, 14, 32, 128, 192, 2, 1) == 1
    assert sk(8, 14, 32, 256, 128, 1, 1) == 1            # patch in the LDS (stride 1, 96 ... 288 channels): up to 2.4e8 multiply-adds
    assert sk(16, 14, 32, 128, 128, 1, 1) == 1 and sk(8, 28, 64, 128, 128, 1, 1) == 0     # up to 8 K output pixels
    assert sk(8, 28, 64, 224, 128, 1, 1) == 0            # 4.1e8: conv3x3_h2_kernel is faster (measured)
    assert sk(8, 56, 128, 128, 96, 1, 1) == 0            # 57344 output pixels
    assert sk(16, 14, 32, 128, 128, 1, 2) == 0           # dilated: fragments from global memory, 7168 pixels, not thin
    assert sk(16, 28, 64, 96, 128, 2, 1) == 1            # stride 2 beyond 4096 pixels
    assert sk(16, 56, 128, 64, 96, 2, 1) == 1            # 1.76e8 multiply-adds at stride 2: against the fp32-pipe kernel (measured)
    assert sk(16, 112, 256, 32, 64, 2, 1) == 0
    assert sk(8, 28, 64, 64, 32, 1, 1) == 1              # thin layer beyond 4096 pixels
    assert sk(8, 112, 256, 128, 128, 1, 1) == 0 and sk(8, 7, 16, 48, 128, 1, 1) == 0 and sk(8, 7, 16, 64, 24, 1, 1) == 0
    assert sk(1, 112, 256, 32, 32, 1, 1) == 1            # a single pair: 28672 pixels but 2.9e7 multiply-adds
    assert L.pwc_conv3x3_sk_packed_floats(288, 128) == 9 * 288 * 128 and L.pwc_conv3x3_sk_packed_floats(48, 128) == 0
    t32 = L.pwc_conv3x3_t32_supported
    assert t32(16, 224, 512, 16, 32, 2) == 1 and t32(16, 112, 256, 32, 32, 1) == 0 and t32(16, 224, 512, 16, 16, 2) == 0
    assert t32(1, 64, 128, 16, 32, 2) == 0               # 16 tiles: not worth a persistent launch
    assert L.pwc_conv3x3_t32_packed_floats(16) == 9 * 512 and L.pwc_conv3x3_t32_packed_floats(32) == 18 * 512
    blk = L.pwc_warp_cost_volume_concat_blk_supported
    assert blk(7, 16, 192, 4, 192, 192, 0, 288, 288) == 1 and blk(14, 32, 128, 4, 128, 128, 2, 256, 256) == 1
    assert blk(14, 32, 32, 4, 32, 32, 2, 128, 0) == 0 and blk(14, 32, 128, 3, 128, 128, 2, 256, 256) == 0
    assert blk(14, 32, 128, 4, 130, 128, 2, 256, 256) == 0      # channel stride % 4
    assert L.pwc_conv3x3_h2_stride2_supported(16, 224, 512, 16, 32) in (0, 1)
    # validation before any launch
    assert L.pwc_conv3x3_sk_f32(None, 32, None, None, None, 32, 1, 4, 4, 32, 32, 1, 1, 1, 0.1, None) == -1
    assert L.pwc_conv3x3_t32_f32(None, 16, None, None, None, 32, 1, 4, 4, 16, 32, 1, 1, 0.1, None) == -1
    assert L.pwc_warp_cost_volume_concat_blk_f32(None, 128, None, 128, None, 0, 1.0, None, 84, 1, None, 0, 1, 4, 4, 128, 4, 0.1, None) == -1


def test_effective_streams_rule():
    """PWCDCNet.effective_streams: the one place that decides into how many sub-batches a batch is cut (bench.py asks the model
    instead of restating the rule).  Constructing the model needs the library, not a GPU... but its VariableStore allocates
    on the device, so the rule is exercised on the unbound method with a stand-in object."""
    import types
    from pwcnet_amd.model import PWCDCNet
    rule = PWCDCNet.effective_streams
    auto = types.SimpleNamespace(streams=None)
    assert rule(auto, (8, 448, 1024, 3)) == 1 and rule(auto, (4, 64, 128, 3)) == 1       # round 5: one stream (measured)
    assert rule(auto, (2, 448, 1024, 3)) == 1 and rule(auto, (2, 128, 192, 3)) == 1
    dc = types.SimpleNamespace(streams=None, use_dc=True)
    assert rule(dc, (8, 448, 1024, 3)) == 1
    assert rule(auto, (1, 448, 1024, 3)) == 1 and rule(auto, (3, 448, 1024, 3)) == 1 and rule(auto, (7, 448, 1024, 3)) == 1
    two, one, four = (types.SimpleNamespace(streams=k) for k in (2, 1, 4))
    assert rule(two, (6, 64, 64, 3)) == 2 and rule(two, (5, 64, 64, 3)) == 1 and rule(one, (8, 448, 1024, 3)) == 1
    assert rule(four, (8, 64, 64, 3)) == 4 and rule(four, (6, 64, 64, 3)) == 1 and rule(four, (2, 64, 64, 3)) == 1


def test_side_stream_probe_inconclusive_falls_back_to_one_stream(monkeypatch):
    """eight ranks of a node run the side-stream probe at the same time (one per GPU, sharing the host).  When the
    probe cannot tell -- the spin kernel did not spin, or every candidate looks as if it shared the caller's queue -- the decision
    must be "single stream" with a report that says why, never an un-vetted side stream and never an exception.  No GPU here: the
    streams, events and the probe kernels are stand-ins; eight threads run the decision concurrently."""
    import threading
    import types
    import torch
    from pwcnet_amd import model as M

    class FakeStream:
        n = 0
        lock = threading.Lock()

        def __init__(self, device=None):
            with FakeStream.lock:
                FakeStream.n += 1
                self.cuda_stream = 1000 + FakeStream.n
    fake_cuda = types.SimpleNamespace(Stream=FakeStream, synchronize=lambda *a, **k: None, is_current_stream_capturing=lambda: False)
    fake_torch = types.SimpleNamespace(cuda=fake_cuda, zeros=lambda n, device=None: torch.zeros(n))
    monkeypatch.setattr(M, "torch", fake_torch)
    verdicts = {"stuck": (True, 0.001, 0.001),        # the spin did not spin: no verdict possible
                "shared": (True, 1.0, 1.0)}           # every candidate finishes behind the spin: all share the caller's queue
    for kind, answer in verdicts.items():
        monkeypatch.setattr(M, "_shares_queue", lambda dev, a, b, probe, spin_ticks=0, _a=answer: _a)
        out = [None] * 8
        def run(i):
            out[i] = M._pick_side_streams("cuda:%d" % i, FakeStream(), 1)
        th = [threading.Thread(target=run, args=(i,)) for i in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        for i, (picked, report) in enumerate(out):
            assert picked is None, (kind, i)
            assert report["verdict"] == "only 0 of 1 vetted: single stream" and report["picked"] == []
            assert report["rejected"] >= 8 and report["device"] == "cuda:%d" % i and len(report["probes"]) >= 8
    # ... and a clean probe picks the first candidate
    monkeypatch.setattr(M, "_shares_queue", lambda dev, a, b, probe, spin_ticks=0: (False, 1.0, 0.01))
    picked, report = M._pick_side_streams("cuda:0", FakeStream(), 1)
    assert picked is not None and len(picked) == 1 and report["verdict"] == "vetted"
    # the model then runs the batch on ONE stream: the forward with no vetted side stream returns None from the side-stream path
    net = types.SimpleNamespace(_side_streams={}, side_stream_report=None)
    monkeypatch.setattr(M, "_shares_queue", lambda dev, a, b, probe, spin_ticks=0: (True, 0.001, 0.001))
    streams, rep = M._pick_side_streams("cuda:0", FakeStream(), 1)
    assert streams is None and "single stream" in rep["verdict"]


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pwcnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("SURVEY", ""), f"{f} mentions the oracle"


def test_cpu_tensors_are_rejected_not_computed_elsewhere():
    import torch
    import pwcnet_amd
    with pytest.raises(_lib.PwcHipError):
        pwcnet_amd.WarpingLayer("bilinear")(torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 2))


# ------------------------------------------------------------------ inventory / layouts
def test_conv_specs_match_checkpoint_index(golden_dir):
    _, entries = ckpt.read_index(os.path.join(golden_dir, "model_600.ckpt.index"))
    assert len(entries) == 333
    mv = ckpt.model_variables(entries)
    specs = W.conv_specs()
    assert len(mv) == 2 * len(specs) == 110
    for name, cin, cout in specs:
        assert mv[name + "/kernel"].shape == (3, 3, cin, cout) and mv[name + "/kernel"].dtype == ckpt.DT_FLOAT
        assert mv[name + "/bias"].shape == (cout,)
    assert sum(int(np.prod(e.shape)) for e in mv.values()) == W.num_parameters(specs) == 5029868
    assert max(e.offset + e.size for e in entries.values()) == 60358428
    assert entries["Variable"].dtype == ckpt.DT_INT32 and entries["Variable"].shape == ()
    assert "pwcdcnet/optflow_5/conv2d/kernel" not in entries


def test_estimator_channel_counts():
    assert [W.estimator_in_channels(l, False) for l in range(5)] == [273, 243, 211, 179, 147]
    assert [W.estimator_in_channels(l, True) for l in range(5)] == [273, 932, 1559, 2154, 2717]
    assert [W.estimator_feature_channels(l, True) for l in range(5)] == [721, 1380, 2007, 2602, 3165]
    assert W.num_parameters(W.conv_specs(use_dc=True)) == 40182338


@pytest.mark.parametrize("use_dc", [False, True])
def test_layouts_are_aligned_bijective_and_zero_padded(use_dc):
    for l in range(5):
        lay = W.estimator_layout(l, use_dc)
        p2l = np.asarray(lay.phys2log)
        assert lay.n_phys % 16 == 0 and lay.n_logical == W.estimator_feature_channels(l, use_dc) if use_dc \
            else lay.n_logical == W.estimator_in_channels(l, use_dc)
        assert sorted(p2l[p2l >= 0].tolist()) == list(range(lay.n_logical))      # bijection onto logical
        for name, (off, ln) in lay.segments.items():
            assert off % 4 == 0
        if use_dc:
            # conv k reads the suffix that starts where conv k-1 wrote / where cv starts
            starts = [lay.offset("cv")] + [lay.offset(f"conv{k}") for k in range(5)]
            assert all(s % 16 == 0 for s in starts)
            done = 0
            for k in range(5):
                m = lay.cin_map(starts[k], logical_base=448 - done)
                cin = lay.n_logical - (448 - done)
                assert sorted(m[m >= 0].tolist()) == list(range(cin))
                done += W.FILTERS_OF[k]
    cx = W.context_layout(use_dc)
    assert cx.n_logical == (3167 if use_dc else 34) and cx.offset("features") == 4


def test_layout_reorders_like_tf_concat():
    lay = W.estimator_layout(2, False)                      # [cv81 | f0 96 | flow 2 | feat_up 32]
    p2l = np.asarray(lay.phys2log)
    assert list(p2l[:81]) == list(range(81)) and list(p2l[81:84]) == [-1] * 3
    off = lay.offset("f0")
    assert off == 84 and list(p2l[off:off + 96]) == list(range(81, 177))
    assert list(p2l[lay.offset("flow"):lay.offset("flow") + 4]) == [177, 178, -1, -1]
    assert list(p2l[lay.offset("feat_up"):lay.offset("feat_up") + 32]) == list(range(179, 211))


def test_glorot_init_is_seeded_and_bounded():
    specs = W.conv_specs()
    a, b = W.init_weights(specs, seed=0), W.init_weights(specs, seed=0)
    k = "pwcdcnet/optflow_4/conv2d/kernel"
    assert np.array_equal(a[k], b[k]) and a[k].shape == (3, 3, 147, 128)
    assert np.abs(a[k]).max() <= np.sqrt(6.0 / (9 * 147 + 9 * 128)) and not np.any(a["pwcdcnet/context/conv2d/bias"])
    assert not np.array_equal(a[k], W.init_weights(specs, seed=1)[k])


# ------------------------------------------------------------------ checkpoint bundle
def test_bundle_round_trip_and_crc(tmp_path):
    specs = W.conv_specs()[:4] + W.conv_specs()[-2:]
    w = W.randomize_biases(W.init_weights(specs, seed=3))
    prefix = str(tmp_path / "model_1.ckpt")
    ckpt.save_weights(prefix, w)
    back = ckpt.load_weights(prefix)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])
    # corrupt one byte of the data file -> CRC failure
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(100); b = f.read(1); f.seek(100); f.write(bytes([b[0] ^ 0xFF]))
    with pytest.raises(ValueError, match="CRC32C"):
        ckpt.load_weights(prefix)
    # corrupt the index -> block CRC failure
    raw = bytearray(open(prefix + ".index", "rb").read()); raw[10] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        ckpt.read_index(prefix + ".index")


def test_reference_index_without_data_blob_reports_missing(golden_dir):
    with pytest.raises(FileNotFoundError, match="data-00000-of-00001"):
        ckpt.load_weights(os.path.join(golden_dir, "model_600.ckpt"))
    assert ckpt.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value


def test_load_weights_strict_rejects_wrong_variable_sets():
    """tf.train.Saver.restore raises on a missing / mis-shaped tensor; so does load_weights
    (checked before anything is copied to a device, so this runs without a GPU)."""
    import pwcnet_amd
    from pwcnet_amd import weights as W
    w = W.init_weights(W.conv_specs())
    net = pwcnet_amd.PWCDCNet()
    with pytest.raises(ValueError, match="missing"):
        net.load_weights({})
    with pytest.raises(ValueError, match="missing"):
        net.load_weights({k.replace("pwcdcnet/", "other/"): v for k, v in w.items()})
    with pytest.raises(ValueError):
        pwcnet_amd.PWCDCNet(use_dc=True).load_weights(w)
    bad = dict(w)
    bad["pwcdcnet/context/conv2d/bias"] = np.zeros((64,), np.float32)
    with pytest.raises(ValueError, match="shape"):
        net.load_weights(bad)


def test_forward_pipeline_host_side():
    """pwcnet_amd.ForwardPipeline (round 6) without a GPU: `depth` replicas of PWCDCNet that keep their status copy in line,
    the model's kwargs passed through, the pipeline's own business (`streams`, `persistent_outputs`) refused, load_weights
    fanned out with the model's strictness, nothing of the oracle imported."""
    import pwcnet_amd
    from pwcnet_amd import pipeline
    pipe = pwcnet_amd.ForwardPipeline(depth=3, device="cuda:0", use_dc=True, output_level=3)
    assert len(pipe.nets) == 3 and pipe.depth == 3 and pipe.effective_depth == 0
    assert all(n.use_dc and n.output_level == 3 and n.streams == 1 and n.status_copy_on_caller_stream for n in pipe.nets)
    assert pwcnet_amd.PWCDCNet().status_copy_on_caller_stream is False
    assert len({id(n.store) for n in pipe.nets}) == 3            # replicas share no state
    for bad in ({"streams": 2}, {"persistent_outputs": True}):
        with pytest.raises(AssertionError):
            pwcnet_amd.ForwardPipeline(depth=2, device="cuda:0", **bad)
    with pytest.raises(ValueError, match="missing"):
        pipe.load_weights({})
    import torch
    with pytest.raises(ValueError, match="CUDA tensors"):      # (CPU tensors are refused like PWCDCNet refuses them: no fallback path)
        pipe.submit(torch.zeros((1, 64, 64, 3)), torch.zeros((1, 64, 64, 3)))
    src = open(pipeline.__file__).read()
    assert "oracle" not in src and "subprocess" not in src


# ------------------------------------------------------------------ sharding
def test_shard_range_partitions():
    for n, world in [(64, 8), (8, 8), (10, 4), (3, 8), (0, 2)]:
        parts = [sharding.shard_range(n, world, r) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)
    assert sharding.gather_stats({"pairs": 8.0, "seconds": 1.5}) == [{"pairs": 8.0, "seconds": 1.5}]


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pwcnet_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
lo, hi = sharding.shard_range(10, w, r)
stats = sharding.gather_stats({"pairs": float(hi - lo), "seconds": 1.0 + r, "lo": float(lo)}, dist, "cpu")
assert len(stats) == w and [s["seconds"] for s in stats] == [1.0 + i for i in range(w)]
assert sum(s["pairs"] for s in stats) == 10 and stats[r]["lo"] == lo
if r == 0:
    print("GATHER_OK", max(s["seconds"] for s in stats), sum(s["pairs"] for s in stats))
dist.destroy_process_group()
"""


def test_gather_stats_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script), ROOT],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GATHER_OK 2.0 10.0" in out.stdout


_WORLD8_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pwcnet_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
steps, per_gpu_batch = 20, 8
# what bench.py hands to the gather on every rank: its pairs and ITS elapsed seconds (rank 5 is the slowest)
mine = {"pairs": float(per_gpu_batch * steps), "seconds": 0.070 + 0.001 * r + (0.010 if r == 5 else 0.0),
        "issue_seconds": 0.010 + 0.001 * r}
stats = sharding.gather_stats(mine, dist, "cpu")
# (bench.py prints every rank's ms per step beside the job's: the straggler is rank 5, visibly)
per_rank_ms = [1e3 * st["seconds"] / steps for st in stats]
assert len(per_rank_ms) == 8 and max(range(8), key=lambda i: per_rank_ms[i]) == 5
value, ms_per_step, total, n = sharding.aggregate_throughput(stats, steps)
assert n == w == 8 and total == 8 * per_gpu_batch * steps
slowest = 0.070 + 0.005 + 0.010
assert abs(ms_per_step - 1e3 * slowest / steps) < 1e-9 and abs(value - total / slowest) < 1e-6
# every rank computes the same aggregate (the gather is an ALL-gather)
agg = sharding.gather_stats({"value": value}, dist, "cpu")
assert all(abs(a["value"] - value) < 1e-9 for a in agg)
if r == 0:
    print("WORLD8_OK", n, int(total), round(value, 3))
dist.destroy_process_group()
"""


def test_bench_aggregation_world8_gloo(tmp_path):
    """The host logic of the 8-GPU bench line (BASELINE configs[2]) on 8 gloo ranks: value = sum of the ranks' pairs /
    the SLOWEST rank's seconds, ms_per_step from that rank, n_gpus = world -- so that the first real 8-GPU run cannot
    trip on the aggregation."""
    script = tmp_path / "world8_worker.py"
    script.write_text(_WORLD8_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29621", str(script), ROOT],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "WORLD8_OK 8 1280 15058.824" in out.stdout


def test_bench_config_presets_imply_their_gpu_count():
    """`--config configs2` / `configs4` are BASELINE's 8- and 2-GPU configurations; an explicit --gpus wins."""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--config", "configs4"])
    assert (a.gpus, a.height, a.width, a.batch, a.gpus_given) == (2, 960, 1920, 8, False)
    assert bench.parse(["--config", "configs2"]).gpus == 8
    a = bench.parse(["--config", "configs4", "--gpus", "1"])
    assert a.gpus == 1 and a.gpus_given
    assert bench.parse(["--config", "configs3"]).gpus == 1 and bench.parse(["--config", "configs3"]).use_dc


_EVAL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pwcnet_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
N, h, wd = 7, 6, 10                        # 7 pairs over 2 ranks: 4 + 3
rs = np.random.RandomState(3)
im = rs.rand(N, 2, h, wd, 3).astype(np.float32)
gt = rs.normal(size=(N, h, wd, 2)).astype(np.float32)
def load_pair(i):
    return torch.from_numpy(im[i, 0]), torch.from_numpy(im[i, 1]), torch.from_numpy(gt[i])
def forward(a, b):                         # a stand-in "network": flow = channel differences
    return torch.stack([(a - b)[..., 0], (a + b)[..., 1]], dim=3)
res = sharding.evaluate_pairs(forward, load_pair, N, batch=3, dist=dist, device="cpu", gather=True)
pred = np.stack([im[:, 0, ..., 0] - im[:, 1, ..., 0], im[:, 0, ..., 1] + im[:, 1, ..., 1]], axis=3)
norms = np.sqrt(((gt - pred) ** 2).sum(axis=3))
assert abs(res["epe"] - norms.mean()) < 1e-6 and res["pairs"] == N
assert np.allclose(res["per_pair_epe"], norms.mean(axis=(1, 2)), atol=1e-6)
assert tuple(res["flows"].shape) == (N, h, wd, 2) and np.allclose(res["flows"].numpy(), pred, atol=1e-6)
assert abs(float(sharding.epe(torch.from_numpy(gt), torch.from_numpy(pred))) - norms.mean()) < 1e-6
if r == 0:
    print("EVAL_OK", w)
dist.destroy_process_group()
"""


def test_sharded_evaluation_world2_gloo(tmp_path):
    """pairs sharded 4 + 3 over two ranks, flows and EPE statistics gathered."""
    script = tmp_path / "eval_worker.py"
    script.write_text(_EVAL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29618", str(script), ROOT],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "EVAL_OK 2" in out.stdout


_EMPTY_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pwcnet_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
N, h, wd = 1, 6, 10                        # ONE pair over 2 ranks: rank 1 holds nothing
rs = np.random.RandomState(5)
im = rs.rand(N, 2, h, wd, 3).astype(np.float32)
gt = rs.normal(size=(N, h, wd, 2)).astype(np.float32)
def load_pair(i):
    return torch.from_numpy(im[i, 0]), torch.from_numpy(im[i, 1]), torch.from_numpy(gt[i])
def forward(a, b):
    return torch.stack([(a - b)[..., 0], (a + b)[..., 1]], dim=3)
res = sharding.evaluate_pairs(forward, load_pair, N, batch=3, dist=dist, device="cpu", gather=True)
pred = np.stack([im[:, 0, ..., 0] - im[:, 1, ..., 0], im[:, 0, ..., 1] + im[:, 1, ..., 1]], axis=3)
assert tuple(res["flows"].shape) == (N, h, wd, 2) and np.allclose(res["flows"].numpy(), pred, atol=1e-6)
assert res["pairs"] == 1 and len(res["per_pair_epe"]) == 1
if r == 0:
    print("EMPTY_RANK_OK", w)
dist.destroy_process_group()
"""


def test_sharded_evaluation_with_an_empty_rank_gloo(tmp_path):
    """n_pairs < world: the rank without pairs must pad to the shape the other ranks hold."""
    script = tmp_path / "empty_worker.py"
    script.write_text(_EMPTY_RANK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29619", str(script), ROOT],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "EMPTY_RANK_OK 2" in out.stdout


_ALLREDUCE_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from pwcnet_amd import sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
flat = torch.arange(10, dtype=torch.float32) * (r + 1)       # rank 0: k, rank 1: 2k
world = sharding.allreduce_sum_(flat, dist)
assert world == 2 and torch.equal(flat, torch.arange(10, dtype=torch.float32) * 3)
assert sharding.allreduce_sum_(flat.clone(), None) == 1
if r == 0:
    print("ALLREDUCE_OK", w)
dist.destroy_process_group()
"""


def test_gradient_allreduce_world2_gloo(tmp_path):
    """the training step's one collective -- the flat gradient buffer summed over the ranks."""
    script = tmp_path / "allreduce_worker.py"
    script.write_text(_ALLREDUCE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29620", str(script), ROOT],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ALLREDUCE_OK 2" in out.stdout


# ------------------------------------------------------------------ flow IO (f2)
def test_flo_round_trip_and_layout(tmp_path):
    from pwcnet_amd import flow_io
    flow = np.random.RandomState(0).normal(size=(5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    flow_io.write_flo(p, flow)
    raw = open(p, "rb").read()
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[:4], "<f4")[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], "<i4")) == (7, 5)             # width first, then height
    np.testing.assert_array_equal(flow_io.read_flo(p), flow)
    open(p, "wb").write(b"\x00" * 20)
    with pytest.raises(ValueError):
        flow_io.read_flo(p)


def test_flow_color_wheel_and_crop():
    from pwcnet_amd import flow_io
    wheel = flow_io.color_wheel()
    assert wheel.shape == (55, 3)
    assert tuple(wheel[0]) == (255, 0, 0) and tuple(wheel[15]) == (255, 255, 0) and tuple(wheel[21]) == (0, 255, 0)
    assert tuple(wheel[54]) == (255, 0, 255 - np.floor(255 * 5 / 6))
    img = flow_io.flow_to_color(np.zeros((4, 6, 2), np.float32))
    assert img.dtype == np.uint8 and img.shape == (4, 6, 3) and (img == 255).all()     # zero flow is white
    f = np.zeros((1, 2, 2), np.float32)
    f[0, 0] = (1, 0)
    f[0, 1] = (-1, 0)
    c = flow_io.flow_to_color(f)
    assert not np.array_equal(c[0, 0], c[0, 1])                                        # opposite directions differ
    assert flow_io.factor_crop(np.zeros((130, 200, 3))).shape == (128, 192, 3)


def test_stale_counter_evidence_is_refused(tmp_path):
    """bench.traffic_stale: no stamp, another source hash, or another launch count per forward -> a reason (the traffic is not
    printed); the tree's own stamp with the same launch count -> None.  And the committed profiles/pmc_traffic.json either
    matches the committed kernel sources or will be reported as stale -- never printed as this run's evidence."""
    import json, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from pwcnet_amd.profiler import source_stamp
    now = source_stamp()
    assert len(now) == 16 and now == source_stamp()
    fam = {"launches_per_forward": 23.0}
    assert bench.traffic_stale(None, fam, 23.0) is not None
    assert "kernel sources" in bench.traffic_stale({"source_sha": "0" * 16}, fam, 23.0)
    assert "launches per forward" in bench.traffic_stale({"source_sha": now}, fam, 24.0)
    assert bench.traffic_stale({"source_sha": now}, fam, 23.0) is None
    t = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    st = t.get("stamp")
    verdict = bench.traffic_stale(st, t["kernels"].get("conv3x3_h2_kernel"), None)
    assert verdict is None or isinstance(verdict, str)
    if st and st.get("source_sha") == now:
        assert verdict is None and "conv3x3_h2_kernel" in st["kernel_symbols"]
