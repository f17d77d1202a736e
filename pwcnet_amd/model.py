"""PWCDCNet -- host-side mirror of reference model.py:74-138 on the HIP ops.

Same constructor kwargs, call signature and return values as the reference class; the
forward is eager (each op is enqueued on torch's current HIP stream) instead of a
TF graph.  Inside ``__call__`` the modules' zero-copy ``_run`` forms are composed:

  * both images go through the extractor as ONE stacked 2N batch (shared weights,
    reference model.py:97-98);
  * per level one zero-initialised buffer holds the estimator input
    [cv | f0 | flow_up | feat_up] (+ the dense-connection conv outputs when use_dc);
    the cost-volume kernel, the x2 resizes of the previous level and the convs write
    straight into its channel slices, so no tf.concat copy exists;
  * warp + cost volume + the f0 part of the concat are ONE launch per level
    (pwc_warp_cost_volume_concat_f32, correlation on the matrix pipe; the warped map is never written;
    the `flows_up * scales[l]` multiply of model.py:109 is folded into the flow read).  The two
    coarsest levels use the latency-oriented coarse kernel; `concat_cv=False` restores the separate
    warp + cost-volume launches of rounds 1-2.
"""
import collections
import warnings

import torch

from . import _lib
from . import modules as _m
from .modules import (ContextNetwork, CostVolumeLayer, FeaturePyramidExtractor_custom, LaunchPlan,
                      OpticalFlowEstimator_custom, VariableStore, View, WarpingLayer, _copy_channels,
                      _keep, _resize, as_view, sub_view, variable_scope)
from .weights import ChannelLayout, SCALES, conv_specs


def _shares_queue(dev, a, b, probe, spin_ticks=400_000):
    """True if HIP serves streams `a` and `b` from one hardware queue, decided on DEVICE timestamps: a spin kernel on `a`
    bracketed by timing events gives the spin's own duration; a one-thread kernel issued on `b` right behind it finishes
    either at once (own queue) or only after the spin (shared queue).  No host clock is involved, so a pre-empted host
    thread cannot fake a shared queue."""
    L = _lib.lib()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(a)
    _lib.check(L.pwc_device_spin(spin_ticks, _lib.ctypes.c_void_p(a.cuda_stream)), "stream probe spin")
    e1.record(a)
    _lib.check(L.pwc_device_touch(_lib.ctypes.c_void_p(probe.data_ptr()), _lib.ctypes.c_void_p(b.cuda_stream)),
               "stream probe touch")
    e2.record(b)
    e1.synchronize()
    e2.synchronize()
    spin_ms = e0.elapsed_time(e1)
    touch_ms = e0.elapsed_time(e2)          # when the kernel on b was done, measured from the start of the spin on a
    return touch_ms > 0.5 * spin_ms, spin_ms, touch_ms


def _pick_side_streams(dev, main, count):
    """(streams, report): `count` torch streams for the sub-batches that do not run on the caller's stream, or
    (None, report) if no vetted set exists -- the caller then runs single-stream (-7 %) instead of risking a side stream
    behind the caller's queue (-25 %: 4.08 against 3.26 ms per forward, profiles/r03_exp_side_stream_queues.txt).
    HIP maps streams onto a few hardware queues (4 by default); every candidate is checked against the caller's stream
    and against the streams already picked (_shares_queue), twice if the first verdict says "shared"."""
    report = {"device": str(dev), "main_stream": int(main.cuda_stream), "picked": [], "rejected": 0, "probes": []}
    if torch.cuda.is_current_stream_capturing():
        report["verdict"] = "capturing: no probe, single stream"
        return None, report
    cands = [torch.cuda.Stream(device=dev) for _ in range(max(8, 2 * count + 4))]
    probe = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)

    def shared(a, b):
        for _ in range(2):                               # a "shared" verdict is confirmed once
            sh, spin_ms, touch_ms = _shares_queue(dev, a, b, probe)
            report["probes"].append((int(a.cuda_stream), int(b.cuda_stream), round(spin_ms, 4), round(touch_ms, 4), bool(sh)))
            if spin_ms < 0.02:                           # the spin did not spin (clock counter stuck?): no verdict
                return True
            if not sh:
                return False
        return True

    picked = []
    for s in cands:
        if len(picked) == count:
            break
        if not shared(main, s) and not any(shared(q, s) for q in picked):
            picked.append(s)
        else:
            report["rejected"] += 1
    torch.cuda.synchronize(dev)
    if len(picked) == count:
        report["picked"] = [int(s.cuda_stream) for s in picked]
        report["verdict"] = "vetted"
        return picked, report
    report["verdict"] = f"only {len(picked)} of {count} vetted: single stream"
    return None, report


class PWCDCNet(object):
    def __init__(self, num_levels=6, search_range=4, warp_type="bilinear", use_dc=False,
                 output_level=4, name="pwcdcnet", seed=0, fuse_warp=False, use_plans=True, winograd=True,
                 coarse_cv=True, persistent_outputs=False, max_plans=4, streams=None, concat_cv=True, winograd4=True, f16x2=True,
                 range_check="lazy", track_max=False):
        self.num_levels = num_levels
        self.s_range = search_range
        self.warp_type = warp_type
        self.use_dc = use_dc
        assert output_level < num_levels, "Should set output_level < num_levels"
        self.output_level = output_level
        self.name = name
        self.fuse_warp = fuse_warp
        self.coarse_cv = coarse_cv
        self.concat_cv = concat_cv     # levels with C in {32, 64, 96}: pwc_warp_cost_volume_concat_f32

        self.fp_extractor = FeaturePyramidExtractor_custom(self.num_levels)
        self.warp_layer = WarpingLayer(self.warp_type)
        self.cv_layer = CostVolumeLayer(search_range)
        self.of_estimators = [OpticalFlowEstimator_custom(use_dc=self.use_dc, name=f"optflow_{l}")
                              for l in range(self.num_levels)]
        self.context = ContextNetwork(name="context")
        # Upscale factors from deep -> shallow level (reference model.py:93)
        self.scales = list(SCALES)

        self._mods = [self.fp_extractor, self.context, self.cv_layer] + self.of_estimators
        for mod in self._mods:
            mod.winograd = bool(winograd)
            mod.winograd4 = bool(winograd4)
            mod.f16x2 = bool(f16x2)
            mod.status = None
            mod.track_max = False
        # F16-pipe kernels (f16x2) compute with operands split into fp16 pairs: exact to 22 bits, but only below 65504 in
        # magnitude -- beyond it they produce NaN, NaN survives every later layer, and the forward's last launch (the x4
        # upsampling of model.py:127) sets PWC_STATUS_NONFINITE in `status` (two uint32 in device memory; include/pwc_hip.h)
        # when it writes a value that is not finite.  The reference is plain fp32 (nothing bounds its activations; test.py:31
        # only divides the frames by 255), so a flagged forward is REPEATED on the fp32 kernels into the very tensors it
        # returned, with a warning, and the model stays on fp32 from then on (frames that are not finite themselves take the
        # same path once and come out as non-finite as the reference's would):
        #   range_check="lazy" (default): the words are copied to the host behind every forward (8 bytes, asynchronous, a slot of
        #       its own per forward) and looked at when a later forward is called, or by status() / synchronize() -- no host
        #       synchronisation is added to a forward.  The words are sticky, so the FIRST pending forward whose copy shows a flag
        #       is the one that tripped it; it and every forward issued behind it (they ran with the flag already up) are
        #       repeated on the fp32 kernels into the tensors they returned -- provided the caller has not written into their
        #       input tensors since (torch's version counters): outputs of a forward whose inputs were refilled in place are set
        #       to NaN instead, never to flows of other frames.  A caller who reads results after torch.cuda.synchronize()
        #       without calling status() / synchronize() would see the NaN of a flagged forward (never a wrong number);
        #   range_check="sync": every call ends with status(): results are right when it returns (infer.py, infer_continuous.py and
        #       evaluate.py build their models this way);
        #   range_check="off": no status words at all.
        # track_max: every operand of an F16-pipe kernel is also scanned for its largest magnitude (pwc_absmax_f32, one small
        # launch each: slower) -> status()["max_abs"] -- for the first person with trained weights to see the margin in one run.
        assert range_check in ("lazy", "sync", "off")
        self.range_check = range_check
        self.track_max = bool(track_max)
        self.status_copy_on_caller_stream = False   # ForwardPipeline sets it (see _record)
        self.two_operand = True             # features_0 read from the pyramid tensor where the first conv allows (_est_layout)
        self.three_operand = True           # round 6: [cv | flow], features_0, features_up as three dense tensors (_level_input)
        self.f16x2 = bool(f16x2)
        self._status = {}                   # device -> [device words, pinned host ring (slots x 2), next slot]
        self._pending = collections.deque() # forwards whose status words nobody has looked at: (event, host slot, inputs, their versions, outputs, device)
        self.fallback_reason = None         # set when the model left the F16-pipe kernels
        _lib.lib()  # fail now, loudly, if the HIP library is missing
        self.store = VariableStore(seed=seed)
        # launch plans (one per input shape, device, stream): the forward is recorded once and
        # replayed.  Every plan owns its intermediate buffers (two forwards of one shape on
        # different streams never share them); at most `max_plans` plans are kept (LRU), an
        # evicted plan releases its buffers.
        # Returned tensors are FRESH on every call, like the arrays sess.run hands back
        # (reference test.py:51,55): the plan's output pointers are re-pointed at newly allocated
        # tensors before each replay.  persistent_outputs=True opts into the zero-copy form in
        # which a call returns the plan's own tensors and the NEXT call of that shape overwrites
        # them (fetching into pre-allocated outputs).
        self.use_plans = use_plans
        self.persistent_outputs = bool(persistent_outputs)
        self.max_plans = max(1, int(max_plans))
        self._plans = collections.OrderedDict()
        self._buffers = {}     # buffer set of the forward being run (a plan's, or the eager one's)
        self._eager_buffers = collections.OrderedDict()   # use_plans=False: (shape, device, stream) -> buffers
        # streams = K > 1: a batch divisible by K is run as K sub-batches on side HIP streams, so that the
        # latency-bound coarse levels of one overlap the MFMA-bound layers of another (+4-5 % pairs/s at batch 8,
        # scripts/exp_two_streams.py, profiles/r02_bench_streams2.json).  Same results and the same stream semantics
        # for the caller (the side streams wait for the caller's stream, the caller's stream waits for them).
        # streams=None (default): 1 since round 5 (effective_streams has the measurements); streams=K asks for K sub-batches
        # (per-kernel timings -- profilers, HIP events on the caller's stream -- need the single-stream form).
        self.streams = None if streams is None else max(1, int(streams))
        self._side_streams = {}
        self.side_stream_report = None      # what _pick_side_streams decided the last time it ran (dict), for logs / bench
        self._warm = set()                  # (sub-batch shape, device, weight version) whose weights are packed

    # ------------------------------------------------------------------ variables
    @property
    def vars(self):
        return [v for v in self.store.vars.values() if self.name in v.name]

    def load_weights(self, weights, strict=True):
        """weights: {'<name>/<scope>/conv2d[_k]/kernel' | '.../bias': array} (the
        reference checkpoint's variable names, without the ':0').

        strict (default): the dict must hold exactly the variables of THIS configuration
        (weights.conv_specs: names and shapes) -- like tf.train.Saver.restore, which raises on a
        missing or mis-shaped tensor (reference test.py:39-40) instead of leaving the model on
        its random initial values.  strict=False assigns whatever is given (shapes of variables
        that already exist are still checked)."""
        if strict:
            specs = conv_specs(num_levels=self.num_levels, search_range=self.s_range, use_dc=self.use_dc,
                               output_level=self.output_level, name=self.name)
            want = {}
            for vname, cin, cout in specs:
                want[vname + "/kernel"] = (3, 3, cin, cout)
                want[vname + "/bias"] = (cout,)
            missing = sorted(set(want) - set(weights))
            extra = sorted(set(weights) - set(want))
            if missing or extra:
                raise ValueError(
                    f"load_weights: variable set does not match {self.name} (use_dc={self.use_dc}): "
                    f"{len(missing)} missing (e.g. {missing[:3]}), {len(extra)} unexpected (e.g. {extra[:3]}); "
                    "pass strict=False to assign a partial set")
            for k, shp in want.items():
                got = tuple(getattr(weights[k], "shape", ()))
                if got != shp:
                    raise ValueError(f"load_weights: {k} has shape {got}, this model needs {shp}")
        for k, v in weights.items():
            self.store.assign(k, v)

    # ------------------------------------------------------------------ buffers
    def _zeros(self, tag, shape, device):
        """Zero-initialised level buffer of the forward being run (self._buffers is the buffer
        set of the current plan / of the current (shape, device, stream) in eager mode)."""
        key = (tag,) + tuple(shape) + (str(device),)
        b = self._buffers.get(key)
        if b is None:
            b = torch.zeros(shape, dtype=torch.float32, device=device)
            self._buffers[key] = b
        return b

    # ------------------------------------------------------------------ forward
    def __call__(self, images_0, images_1, with_features=False, reuse=False):
        """(flows_final, flows_pyramid[, pyramid_0]) as reference model.py:95-134.  The returned
        tensors are new on every call unless the model was built with persistent_outputs=True
        (then they are the launch plan's own tensors, overwritten by the next call of that shape)."""
        checking = (self.range_check != "off" and self.f16x2 and _m._RECORDER is None and isinstance(images_0, torch.Tensor)
                    and images_0.is_cuda and not torch.cuda.is_current_stream_capturing())
        if checking:
            self._examine(wait=False)                       # the previous forward's words, if they have arrived
            self._attach_status(images_0.device)
        out = self._dispatch(images_0, images_1, with_features)
        if checking and self.f16x2:
            self._record(images_0, images_1, out)
            if self.range_check == "sync":
                self._examine(wait=True)
        return out

    def _dispatch(self, images_0, images_1, with_features, into=None):
        k = self.effective_streams(getattr(images_0, "shape", (0, 0, 0, 0)))
        if k > 1:
            self.max_plans = max(self.max_plans, k + 1)     # a plan per sub-batch stream + the whole-batch one
        if (k > 1 and self.use_plans and not self.persistent_outputs and not with_features and _m._RECORDER is None
                and not torch.cuda.is_current_stream_capturing()):
            out = self._call_on_side_streams(images_0, images_1, k, into=into)
            if out is not None:
                return out
        return self._call_one(images_0, images_1, with_features, into=into)

    # ------------------------------------------------------------------ status words of the F16-pipe kernels
    _STATUS_SLOTS = 64

    def _attach_status(self, dev):
        key = str(dev)
        if key not in self._status:
            words = torch.zeros(2, dtype=torch.int32, device=dev)
            ring = torch.zeros((self._STATUS_SLOTS, 2), dtype=torch.int32).pin_memory()
            self._status[key] = [words, ring, 0]
        words = self._status[key][0]
        for mod in self._mods:
            mod.status = words
            mod.track_max = self.track_max

    def _record(self, images_0, images_1, out):
        """The status words as they stand behind this forward, copied (asynchronously) into a host slot of its own."""
        if len(self._pending) >= self._STATUS_SLOTS - 1:    # (a slot is reused only after its forward has been looked at)
            self._examine(wait=True)
        st = self._status[str(images_0.device)]
        words, ring, nxt = st[0], st[1], st[2]
        host = ring[nxt % self._STATUS_SLOTS]
        st[2] = nxt + 1
        # the 8-byte copy runs on a stream of its own behind an event: on the caller's stream it is a 4-5 us blit dispatch at the
        # end of every forward (profiles/r05_forward_trace_b8.txt, row 63).  The words are sticky, so a copy that is overtaken by
        # the next forward can only show a flag EARLY -- _examine then repeats one innocent forward more, never one less.
        # (ForwardPipeline: the caller's stream IS a side stream with a hardware queue of its own -- a second stream per replica
        # could land on ANOTHER replica's queue and hold its launches back behind this forward's end: the copy stays in line)
        if self.status_copy_on_caller_stream:
            host.copy_(words, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            if len(st) < 4:
                st.append(torch.cuda.Stream(device=images_0.device))
            side = st[3]
            done = torch.cuda.Event()
            done.record()
            side.wait_event(done)
            with torch.cuda.stream(side):
                host.copy_(words, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
        vers = (getattr(images_0, "_version", None), getattr(images_1, "_version", None))
        self._pending.append((ev, host, images_0, images_1, vers, out, images_0.device))

    def _examine(self, wait):
        """Look at the status words of the pending forwards, oldest first (wait=False: those whose copy has arrived).  The
        words are sticky: the first forward whose copy shows a flag is the one that raised it.  A range violation or a
        stream-K timeout: warn, leave the F16-pipe kernels for good, and repeat that forward AND every forward issued behind it
        on the fp32 kernels into the tensors they returned (stream-ordered on the current stream) -- unless the caller has
        written into a forward's input tensors since: its outputs are then set to NaN (never flows of other frames)."""
        while self._pending:
            ev, host, im0, im1, vers, out, dev = self._pending[0]
            if not wait and not ev.query():
                return 0
            ev.synchronize()
            flags = int(host[0].item())
            if not flags:
                self._pending.popleft()
                continue
            affected = list(self._pending)
            self._pending.clear()
            why = []
            if flags & _lib.STATUS_NONFINITE:
                why.append("the flows are not finite: an activation left fp16's range (|x| >= 65504) in a kernel that splits its "
                           "operands into fp16 pairs (or the frames are not finite)")
            if flags & _lib.STATUS_STREAMK_TIMEOUT:
                why.append("a stream-K workgroup gave up waiting for a partial sum")
                _m.h2_workspaces_refill()
            self.fallback_reason = "; ".join(why)
            stale = [rec for rec in affected
                     if (getattr(rec[2], "_version", None), getattr(rec[3], "_version", None)) != rec[4]]
            warnings.warn(f"PWCDCNet: {self.fallback_reason}: the affected forward" +
                          (f" and the {len(affected) - 1} issued behind it are" if len(affected) > 1 else " is") +
                          " repeated on the fp32 kernels and the model stays on them (f16x2=False) from now on" +
                          (f"; {len(stale)} of them had their input tensors modified since: their outputs are set to NaN"
                           if stale else ""), RuntimeWarning, stacklevel=3)
            self._set_f16x2(False)
            with torch.cuda.device(dev):
                self._status[str(dev)][0].zero_()
                cur = torch.cuda.current_stream(dev)
                for rec in affected:
                    r_ev, _, r0, r1, r_vers, r_out, _ = rec
                    cur.wait_event(r_ev)                    # (a forward issued on another stream)
                    if (getattr(r0, "_version", None), getattr(r1, "_version", None)) != r_vers:
                        r_out[0].fill_(float("nan"))
                        for t in r_out[1]:
                            t.fill_(float("nan"))
                        continue
                    res = self._dispatch(r0, r1, len(r_out) == 3, into=(r_out[0], list(r_out[1])))
                    if len(r_out) == 3:                     # with_features: the pyramid came from the overflowing extractor too
                        for dst, src in zip(r_out[2], res[2]):
                            dst.copy_(src)
            return flags
        return 0

    def _set_f16x2(self, on):
        self.f16x2 = bool(on)
        for mod in self._mods:
            mod.f16x2 = bool(on)
        self._plans.clear()                                  # recorded launches name the kernels
        self._eager_buffers.clear()
        self._warm.clear()

    def status(self):
        """Synchronise with the status words of every forward not yet looked at and act on them (see __init__).  Returns
        {"flags", "f16x2", "fallback_reason"[, "max_abs"]}."""
        flags = self._examine(wait=True)
        rep = {"flags": flags, "f16x2": self.f16x2, "fallback_reason": self.fallback_reason}
        if self.track_max:
            rep["max_abs"] = max([float(st[0][1:2].view(torch.float32).item()) for st in self._status.values()], default=0.0)
        return rep

    def synchronize(self):
        """status() for callers who only want the guarantee: when it returns, every tensor this model has handed out holds
        what the fp32 reference arithmetic gives (a flagged forward has been repeated on the fp32 kernels)."""
        return self.status()

    def effective_streams(self, shape):
        """Number of sub-batches a batch of this (N, H, W, 3) shape is run as: `streams` if given (and dividing N), else 1.
        Rounds 2-4 cut even batches of at least 4 pairs in two (+4-5 %, later +1-2 %: the latency-bound small levels of one
        half ran under the matrix-bound layers of the other).  With the small levels on conv3x3_sk.hip / cost_volume_blk.hip
        (round 5) halving the batch costs what the overlap gains: batch 2 / 4 / 8 / 16 / 32, one stream against two: 2001 /
        1969, 2598 / 2640, 3094 / 3031, 3441 / 3337, 3566 / 3593 pairs/s (profiles/r05_exp_streams_ab.txt) -- so the default is
        the single stream, which also keeps per-kernel timings meaningful; streams=2 remains available."""
        n_batch = int(shape[0]) if len(shape) >= 1 else 0
        k = self.streams
        if k is None:
            k = 1
        if k > 1 and (n_batch % k != 0 or n_batch < k):
            k = 1
        return k

    def _call_on_side_streams(self, images_0, images_1, k, into=None):
        _, images_0 = as_view(images_0, "images_0")
        _, images_1 = as_view(images_1, "images_1")
        dev = images_0.device
        N, H, W, _ = images_0.shape
        n = N // k
        main = torch.cuda.current_stream(dev)
        skey = (str(dev), main.cuda_stream)
        if skey not in self._side_streams:
            while len(self._side_streams) >= 4:              # callers that come with a new stream every time
                self._side_streams.pop(next(iter(self._side_streams)))
            picked, self.side_stream_report = _pick_side_streams(dev, main, k - 1)
            self._side_streams[skey] = picked
        streams = self._side_streams[skey]
        if streams is None or len(streams) != k - 1:
            return None                                      # no vetted side stream: the caller runs single-stream
        if into is not None:
            final, pyr = into
        else:
            final = torch.empty((N, H, W, 2), dtype=torch.float32, device=dev)
            pyr = [torch.empty((N, H >> (self.num_levels - l), W >> (self.num_levels - l), 2), dtype=torch.float32, device=dev)
                   for l in range(self.output_level + 1)]
        # First call of a (sub-batch shape, weight version): sub-batch 0 runs FIRST and alone, on the caller's stream.  It packs
        # every layer's weights there (the per-module caches are filled by whoever launches a layer first, with no event
        # between the pack kernel and a consumer on another stream) and owns the packed tensors' allocator blocks; the
        # side streams start behind it.
        wkey = ((n, H, W), str(dev), self.store.version)
        first = wkey not in self._warm
        if first:
            self._call_one(images_0[:n], images_1[:n], False, into=(final[:n], [p[:n] for p in pyr]))
            if len(self._warm) > 64:
                self._warm.clear()
            self._warm.add(wkey)
        # sub-batch 0 runs on the caller's stream itself (no event in its way), the others on side streams that start when
        # the caller's stream has reached this point and that the caller's stream waits for at the end.  The side streams'
        # launches are issued first: they are the ones that still have an event to wait for.
        ready = torch.cuda.Event()
        ready.record(main)
        dones = []
        for i, st in enumerate(streams, start=1):
            sl = slice(i * n, (i + 1) * n)
            st.wait_event(ready)
            with torch.cuda.stream(st):
                self._call_one(images_0[sl], images_1[sl], False, into=(final[sl], [p[sl] for p in pyr]))
            done = torch.cuda.Event()
            done.record(st)
            dones.append(done)
        if not first:
            self._call_one(images_0[:n], images_1[:n], False, into=(final[:n], [p[:n] for p in pyr]))
        for done in dones:
            main.wait_event(done)
        return final, pyr

    def _call_one(self, images_0, images_1, with_features=False, into=None):
        iv0, images_0 = as_view(images_0, "images_0")
        iv1, images_1 = as_view(images_1, "images_1")
        assert iv0[2:] == iv1[2:], "image batches must have equal shapes"
        dev = images_0.device
        stream = torch.cuda.current_stream().cuda_stream
        if not self.use_plans or iv0.ptr == iv1.ptr or _m._RECORDER is not None:
            # eager: buffers per (shape, device, stream), same LRU bound as the plans
            bkey = (iv0[1:], str(dev), stream)
            bufs = self._eager_buffers.pop(bkey, None)
            self._eager_buffers[bkey] = self._buffers = {} if bufs is None else bufs
            while len(self._eager_buffers) > self.max_plans:
                self._eager_buffers.popitem(last=False)
            return self._hand_over(self._forward(iv0, iv1, dev, with_features), into, with_features)
        # (the alignment of the frames is part of the key: the fused level-1 launch of the extractor takes 16-byte aligned
        # frames only, and a plan recorded with it must not be replayed on a view that is not -- ADVICE r4)
        key = (iv0[1:], str(dev), stream, bool(with_features), self.store.version, iv0.ptr % 16, iv1.ptr % 16,
               self._frames_shared(iv0, iv1))     # (a plan recorded on a sequence reads images_1 through images_0's pointer)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
            patch = {"images_0": iv0.ptr, "images_1": iv1.ptr}
            if self.persistent_outputs:
                plan.replay(patch)
                return plan.outputs
            outputs = self._fresh_outputs(plan, patch, into)
            plan.replay(patch)
            if with_features:
                # pyramid_0 are slices of the extractor's activations (plan-owned): copy them,
                # stream-ordered after the replay
                outputs = outputs + ([t.clone() for t in plan.outputs[2]],)
            return outputs
        plan = LaunchPlan()
        plan.buffers = self._buffers = {}
        _m._RECORDER = plan
        try:
            outputs = self._forward(iv0, iv1, dev, with_features)
        finally:
            _m._RECORDER = None
        out_ptrs = {}
        small = [outputs[0]] + list(outputs[1])
        for i, t in enumerate(small):
            out_ptrs[t.data_ptr()] = f"out{i}"
        for ci, call in enumerate(plan.calls):
            for ai, arg in enumerate(call[1]):
                if not isinstance(arg, _lib.ctypes.c_void_p):
                    continue
                if arg.value in (iv0.ptr, iv1.ptr):
                    plan.patch.setdefault("images_0" if arg.value == iv0.ptr else "images_1", []).append((ci, ai))
                elif arg.value in out_ptrs:
                    plan.patch.setdefault(out_ptrs[arg.value], []).append((ci, ai))
        plan.outputs = outputs
        for k in [k for k in self._plans if k[4] != self.store.version]:
            del self._plans[k]
        self._plans[key] = plan
        while len(self._plans) > self.max_plans:
            self._plans.popitem(last=False)          # least recently used; its buffers go with it
        if into is not None:                         # recording call of a sub-batch / of a repeat: hand the results over
            return self._hand_over(outputs, into, with_features)
        if with_features and not self.persistent_outputs:
            # pyramid_0 are slices of plan-owned extractor activations: the next replay of this shape would
            # rewrite them under the caller (flows_final / flows_pyramid of the recording call are tensors of
            # their own; replays are re-pointed at fresh ones)
            outputs = (outputs[0], outputs[1], [t.clone() for t in outputs[2]])
        return outputs

    @staticmethod
    def _hand_over(outputs, into, with_features=False):
        """Copy a forward's own flows into the caller-provided `into` = (final, pyramid) slices (sub-batches on
        side streams, repeats on the fp32 kernels); with into=None the outputs pass through.  with_features: the pyramid of
        that forward rides along as the third element (copies: the activations behind it belong to the plan)."""
        if into is None:
            return outputs
        into[0].copy_(outputs[0])
        for dst, src in zip(into[1], outputs[1]):
            dst.copy_(src)
        if with_features:
            return into[0], into[1], [t.clone() for t in outputs[2]]
        return into

    def _fresh_outputs(self, plan, patch, into=None):
        """New flows_final / flows_pyramid tensors for this replay (or the caller's `into` slices); their
        pointers are patched into the recorded launches like the inputs'."""
        old = plan.outputs
        if into is not None:
            new = [into[0]] + list(into[1])
            for t, o in zip(new, [old[0]] + list(old[1])):
                assert tuple(t.shape) == tuple(o.shape) and t.is_contiguous(), (t.shape, o.shape)
        else:
            new = [torch.empty_like(t) for t in [old[0]] + list(old[1])]
        for i, t in enumerate(new):
            patch[f"out{i}"] = t.data_ptr()
        return new[0], new[1:]

    @staticmethod
    def _frames_shared(iv0, iv1):
        """True where the two image batches are frames[:-1] and frames[1:] of one dense tensor (same sizes and strides, images_1
        exactly one frame behind images_0)."""
        return (iv0[1:] == iv1[1:] and iv0.cs == iv0.C
                and iv1.ptr == iv0.ptr + 4 * iv0.H * iv0.W * iv0.cs)

    def _forward(self, iv0, iv1, dev, with_features):
        N = iv0.N
        with variable_scope(self.name, store=self.store):
            # Round 6: images_1 = images_0 shifted by one frame of the SAME tensor (frames[:-1], frames[1:] -- a sequence, reference
            # test_continuous.py:55-62 feeds its consecutive pairs one by one): the N + 1 frames go through the extractor once, not
            # as 2 N images; the second pyramid of pair i is the first of pair i + 1.
            f1_off = N
            if self._frames_shared(iv0, iv1):
                stacked = self.fp_extractor._run([View(iv0.ptr, iv0.cs, N + 1, iv0.H, iv0.W, iv0.C)], dev)[::-1]
                f1_off = 1
            else:
                stacked = self.fp_extractor._run([iv0, iv1], dev)[::-1]   # deep -> shallow, 2N batch
            pyramid_0 = [f[:N] for f in stacked]

            flows_pyramid = []
            nxt = None             # the estimator input prepared for the current level (_level_input)
            for l, F in enumerate(stacked):
                _, h, w, C = F.shape
                f0 = View(F.data_ptr(), C, N, h, w, C)
                f1 = View(F.data_ptr() + 4 * f1_off * h * w * C, C, N, h, w, C)
                est = self.of_estimators[l]
                is_out = (l == self.output_level)

                inp = nxt if nxt is not None else self._level_input(l, N, h, w, C, None, dev, is_out)
                lay, cx = inp["lay"], inp["cx"]
                if inp["three"]:
                    # Round 6: [cv | flows_up_prev | 0] as a dense tensor of 84-channel records written by the correlation launch,
                    # features_0 where it is, features_up_prev as a dense tensor: the first conv takes the three of them
                    cvx, flow_v, fu_v = inp["cvx"], inp["flow"], inp["feat_up"]
                    self.cv_layer._run(f0, f1, View(cvx.ptr, cvx.cs, N, h, w, (2 * self.s_range + 1) ** 2), flow=flow_v,
                                       flow_scale=self.scales[l], concat=True, out_pad_writable=2)
                    E = None
                else:
                    E_t, E_off = inp["E_t"], inp["E_off"]
                    E = View(E_t.data_ptr() + 4 * E_off, E_t.shape[3], N, h, w, lay.n_phys)
                    # Warping + cost volume (model.py:105-112).  features_0 either has its segment of the estimator buffer (the
                    # correlation launch copies it there) or stays where it is: the first conv then reads it from the pyramid
                    # tensor (_est_layout)
                    f0_ext = f0 if "f0" in lay.external else None
                    cv_out = sub_view(E, lay.offset("cv"), (2 * self.s_range + 1) ** 2)
                    f0_dst = sub_view(E, lay.offset("f0"), C) if f0_ext is None else None
                    flow_v = sub_view(E, lay.offset("flow"), 2) if l > 0 else None
                    self._corr_level(l, f0, f1, flow_v, cv_out, f0_dst, E, dev)

                def run_est(flows_out, feat_out=None):
                    if inp["three"]:
                        return est._run3(cvx, f0, fu_v, flow_v, flows_out, feat_out=feat_out)
                    if self.use_dc:
                        return est._run(E, lay, flows_out)
                    return est._run(E, lay, flows_out, feat_out=feat_out, f0_ext=f0_ext)

                flows_t = torch.empty((N, h, w, 2), dtype=torch.float32, device=dev)
                _keep(flows_t)
                flows_v = View(flows_t.data_ptr(), 2, N, h, w, 2)
                if not is_out:
                    # Optical flow estimation + x2 upsampling into the next level's buffer
                    # (model.py:114-116, modules.py:282-285)
                    if self.use_dc:
                        run_est(flows_v)
                        feat_v, nfu = View(E.ptr, E.cs, N, h, w, lay.n_phys), list(lay.phys2log)
                    else:
                        feat_v, _feat_t = run_est(flows_v)
                        nfu = list(range(feat_v.C))
                    Fn = stacked[l + 1]
                    _, h2, w2, C2 = Fn.shape
                    assert (h2, w2) == (2 * h, 2 * w), "pyramid levels must double in size"
                    nxt = self._level_input(l + 1, N, h2, w2, C2, nfu, dev, l + 1 == self.output_level)
                    if feat_v.C % 4 == 0 and feat_v.cs % 4 == 0 and nxt["feat_up"].cs % 4 == 0:
                        _m._resize_pair(flows_v, nxt["flow"], feat_v, nxt["feat_up"])
                    else:
                        _resize(flows_v, nxt["flow"])
                        _resize(feat_v, nxt["feat_up"])
                    flows_pyramid.append(flows_t)
                    continue

                # At output level (model.py:117-132)
                cx_t, cx_lay = cx
                CX = View(cx_t.data_ptr(), cx_lay.n_phys, N, h, w, cx_lay.n_phys)
                ctx_flow = sub_view(CX, cx_lay.offset("flow"), 2)
                if self.use_dc:
                    run_est(ctx_flow)
                else:
                    run_est(ctx_flow, feat_out=sub_view(CX, cx_lay.offset("features"), est.filters[-1]))
                self.context._run(CX, cx_lay, flows_v)
                flows_pyramid.append(flows_t)
                upscale = 2 ** (self.num_levels - self.output_level)
                flows_final = torch.empty((N, h * upscale, w * upscale, 2), dtype=torch.float32, device=dev)
                _keep(flows_final)
                # (the last launch also reports non-finite flows: see __init__, range_check)
                _resize(flows_v, View(flows_final.data_ptr(), 2, N, h * upscale, w * upscale, 2), mul=20.0,
                        status=self.context.status)
                if with_features:
                    return flows_final, flows_pyramid, pyramid_0
                else:
                    return flows_final, flows_pyramid

    def _level_input(self, l, N, h, w, C, fu_map, dev, is_out):
        """The estimator input of level l (fu_map: physical -> logical map of features_up_prev, None at level 0): a dict with
        the buffers and Views the level's launches use.  "three" (round 6, _three_operand_level): three dense tensors -- `cvx`
        ([cv | flow | 0] records, written by the correlation launch), features_0 in the pyramid tensor, `feat_up`; `flow` is a
        2-channel tensor of its own.  Otherwise ONE zero-initialised buffer with a ChannelLayout (`lay`), `flow` and `feat_up`
        being its segments.  `flow` / `feat_up` are where the previous level's x2 up-sampling writes."""
        has_flow = l > 0
        cvc = (2 * self.s_range + 1) ** 2
        if has_flow and self._three_operand_level(l, N, h, w, C, fu_map):
            est = self.of_estimators[l]
            cvx_t = self._zeros(f"cvx{l}", (N, h, w, est.CVX_CS), dev)
            flow_t = self._zeros(f"flowup{l}", (N, h, w, 2), dev)
            fu_t = self._zeros(f"featup{l}", (N, h, w, len(fu_map)), dev)
            cx = None
            if is_out:                      # the context network's [flows | features] buffer (modules.py:305)
                cx_lay = ChannelLayout()
                cx_lay.add("flow", 2)
                cx_lay.add("features", est.filters[-1])
                cx_lay.finish(16)
                cx = (self._zeros(f"ctx{l}", (N, h, w, cx_lay.n_phys), dev), cx_lay)
            return {"three": True, "lay": None, "cx": cx,
                    "cvx": View(cvx_t.data_ptr(), est.CVX_CS, N, h, w, 96),
                    "flow": View(flow_t.data_ptr(), 2, N, h, w, 2),
                    "feat_up": View(fu_t.data_ptr(), len(fu_map), N, h, w, len(fu_map))}
        lay = self._est_layout(l, N, h, w, C, has_flow, fu_map)
        E_t, E_off, cx = self._level_buffer(l, lay, N, h, w, dev, is_out)
        E = View(E_t.data_ptr() + 4 * E_off, E_t.shape[3], N, h, w, lay.n_phys)
        return {"three": False, "lay": lay, "cx": cx, "E_t": E_t, "E_off": E_off,
                "flow": sub_view(E, lay.offset("flow"), 2) if has_flow else None,
                "feat_up": sub_view(E, lay.offset("feat_up"), len(fu_map)) if has_flow else None}

    def _three_operand_level(self, l, N, h, w, C, fu_map):
        """True where level l runs on the three-tensor estimator input: non-DC, bilinear warp, the row-walking F16-pipe
        correlation kernel takes the level (not the block-per-workgroup one of the small levels) and the first conv goes to
        the F16-pipe kernel at the stage count of that input."""
        if not (self.three_operand and self.two_operand and self.f16x2 and not self.use_dc and self.concat_cv
                and self.warp_type == "bilinear" and self.s_range == 4 and fu_map is not None
                and list(fu_map) == list(range(len(fu_map)))):
            return False
        est = self.of_estimators[l]
        al = 16      # (any aligned address: the tensors are torch allocations)
        f0 = View(al, C, N, h, w, C)
        out = View(al, est.CVX_CS, N, h, w, 81)
        flow = View(al, 2, N, h, w, 2)
        if self.coarse_cv and (self.cv_layer.blk_ok(f0, f0, out, flow=flow) or self.cv_layer.coarse_ok(f0)):
            return False
        return self.cv_layer.concat_ok(f0, f0, out, flow=flow) and est.three_operand_ok(N, h, w, 81, C, len(fu_map))

    def _est_layout(self, l, N, h, w, C, has_flow, fu_map):
        """Layout of the estimator buffer of level l.  Where the first conv of the (non-DC) estimator runs on the F16-pipe
        kernel, features_0 is NOT part of the buffer (round 5): that kernel takes the channels of a second tensor behind the
        buffer's, so `tf.concat([cv, features_0, ...])` (reference modules.py:261-264) moves no byte of features_0 --
        29 MB written and read again at level 4 of a batch of 8."""
        est = self.of_estimators[l]
        cvc = (2 * self.s_range + 1) ** 2
        ext = self.two_operand and est.two_operand_ok(N, h, w, cvc, C, has_flow, fu_map)
        return est._layout(cvc, C, has_flow, fu_map, f0_external=ext)

    def _corr_level(self, l, f0, f1, flow_v, cv_out, f0_dst, E, dev):
        """Warping + cost volume of pyramid level l (reference model.py:105-112) plus the
        features_0 part of the estimator input's concat (modules.py:264): f1 is warped by
        flow_v * scales[l] (l > 0) and correlated with f0 into cv_out; f0 is copied to f0_dst (None: the estimator reads
        features_0 from the pyramid tensor, nothing to copy)."""
        N, h, w, C = f0.N, f0.H, f0.W, f0.C
        if self.coarse_cv and self.concat_cv and (l == 0 or self.warp_type == "bilinear") and \
                self.cv_layer.blk_ok(f0, f1, cv_out, flow=flow_v, f0_copy=f0_dst):
            # small levels, F16 matrix pipe: one 4 x 4 block per workgroup, its whole window requested at once
            self.cv_layer._run(f0, f1, cv_out, flow=flow_v, flow_scale=self.scales[l] if l > 0 else 1.0,
                               f0_copy=f0_dst, concat=True, out_pad_writable=True, blk=True)
            return
        if self.coarse_cv and self.cv_layer.coarse_ok(f0) and (l == 0 or self.warp_type == "bilinear"):
            # coarse levels: warp + cost volume + the f0 part of the concat in ONE launch
            self.cv_layer._run(f0, f1, cv_out, flow=flow_v, flow_scale=self.scales[l] if l > 0 else 1.0,
                               f0_copy=f0_dst, coarse=True)
            return
        if self.concat_cv and (l == 0 or self.warp_type == "bilinear") and \
                self.cv_layer.concat_ok(f0, f1, cv_out, flow=flow_v, f0_copy=f0_dst):
            # warp + cost volume + the f0 part of the concat in ONE launch on the matrix pipe; the warped map is
            # never written.  Channels 81..83 behind the cost volume are layout padding (ChannelLayout starts every
            # segment on a multiple of 4 channels): the kernel may zero them.
            self.cv_layer._run(f0, f1, cv_out, flow=flow_v, flow_scale=self.scales[l] if l > 0 else 1.0,
                               f0_copy=f0_dst, concat=True, out_pad_writable=True)
            return
        copied = f0_dst is None
        if l == 0:
            self.cv_layer._run(f0, f1, cv_out)
        elif self.warp_type == "bilinear" and self.fuse_warp:
            self.cv_layer._run(f0, f1, cv_out, flow=flow_v, flow_scale=self.scales[l])
        else:
            f1w_t = torch.empty((N, h, w, C), dtype=torch.float32, device=dev)
            _keep(f1w_t)
            f1w = View(f1w_t.data_ptr(), C, N, h, w, C)
            # the f0 part of the concat rides in the warp launch
            fuse_copy = C % 4 == 0 and E.cs % 4 == 0 and f0_dst is not None
            self.warp_layer._run(f1, flow_v, f1w, flow_scale=self.scales[l],
                                 copy=(f0, f0_dst) if fuse_copy else None)
            self.cv_layer._run(f0, f1w, cv_out)
            copied = copied or fuse_copy
        if not copied:
            _copy_channels(f0, f0_dst, C)

    def _level_buffer(self, l, lay, N, h, w, dev, is_out):
        """Zero-initialised estimator buffer of level l.  At the output level with dense
        connections the estimator's buffer IS the `features` segment of the context
        network's input buffer ([flows | features], modules.py:305)."""
        if not is_out:
            return self._zeros(f"est{l}", (N, h, w, lay.n_phys), dev), 0, None
        cx_lay = ChannelLayout()
        cx_lay.add("flow", 2)
        if self.use_dc:
            cx_lay.add("features", lay.n_phys, log_map=list(lay.phys2log))
        else:
            cx_lay.add("features", self.of_estimators[l].filters[-1])
        cx_lay.finish(16)
        cx_t = self._zeros(f"ctx{l}", (N, h, w, cx_lay.n_phys), dev)
        if self.use_dc:
            return cx_t, cx_lay.offset("features"), (cx_t, cx_lay)
        return self._zeros(f"est{l}", (N, h, w, lay.n_phys), dev), 0, (cx_t, cx_lay)
