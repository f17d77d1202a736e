"""ForwardPipeline -- several whole forwards in flight on one MI355X (round 6).

The reference runs one `sess.run` per batch (test.py:55-62, test_continuous.py: one per frame pair).  On MI355X a PWC-Net
forward is a chain of ~64 launches of which the ~40 that serve the coarse pyramid levels are launch- and latency-bound (a few
workgroups each, 5-20 us: profiles/r06_forward_trace_b8.txt) while the fine levels keep all 256 CUs busy.  Within ONE forward
nothing can run beside them (every level needs the flow of the level below).  Consecutive forwards are independent: dealt
round-robin to `depth` replicas of the model -- each with its own activations, stream-K workspace and status words, on a HIP
stream served by a hardware queue of its own -- the launch-bound stretch of one forward runs under the matrix-bound stretch of
another.  Measured (profiles/r06_exp_pipeline.txt): batch 8 x 448 x 1024 2.49 -> 2.2x ms per forward, batch 1 0.76 -> 0.5x.

Every replica computes exactly what PWCDCNet computes (same kernels, same launch plans): results are bit-identical to the
one-stream loop (tests/test_gpu_model.py::test_pipeline_matches_single_stream).

Stream semantics (those of a torch op): submit() orders a forward behind everything already enqueued on the CALLER's current
stream (the frames may still be being written there) and returns a Ticket; Ticket.result() makes the caller's current stream
wait for that forward and returns (flows_final, flows_pyramid) -- no host synchronisation anywhere.  __call__ = submit +
result: a drop-in for PWCDCNet.__call__ that still overlaps with the forwards submitted before it.
"""
import collections

import torch

from .model import PWCDCNet, _pick_side_streams


class Ticket(object):
    """A forward in flight: result() -> (flows_final, flows_pyramid), stream-ordered on the caller's current stream."""
    __slots__ = ("_out", "_done", "_lane")

    def __init__(self, out, done, lane):
        self._out, self._done, self._lane = out, done, lane

    def result(self):
        cur = torch.cuda.current_stream(self._out[0].device)
        cur.wait_event(self._done)
        for t in [self._out[0]] + list(self._out[1]):
            t.record_stream(cur)            # allocated on the lane's stream, read on the caller's
        return self._out

    def done(self):
        return self._done.query()


class ForwardPipeline(object):
    def __init__(self, depth=3, device=None, **net_kwargs):
        """depth: forwards in flight (= model replicas = HIP streams).  net_kwargs: PWCDCNet's (num_levels, search_range,
        warp_type, use_dc, output_level, range_check, ...); `streams` and `persistent_outputs` are the pipeline's business."""
        assert depth >= 1
        for k in ("streams", "persistent_outputs"):
            assert k not in net_kwargs, f"ForwardPipeline: {k} cannot be combined with a pipeline"
        self.depth = int(depth)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.nets = [PWCDCNet(streams=1, **net_kwargs) for _ in range(self.depth)]
        for net in self.nets:
            net.status_copy_on_caller_stream = True     # a lane IS a side stream: no second one per replica (see PWCDCNet._record)
        self._lanes = None
        self.effective_depth = 0            # lanes actually found (set by the first submit)
        self.stream_report = None
        self._next = 0
        self._inflight = collections.deque()

    # ------------------------------------------------------------------ the model's own interface, fanned out
    def load_weights(self, weights, strict=True):
        for net in self.nets:
            net.load_weights(weights, strict=strict)

    @property
    def store(self):
        return self.nets[0].store

    def status(self):
        """PWCDCNet.status() over the replicas: synchronises with every forward not yet looked at."""
        reps = [net.status() for net in self.nets]
        flags = 0
        for r in reps:
            flags |= r["flags"]
        return {"flags": flags, "f16x2": all(r["f16x2"] for r in reps),
                "fallback_reason": next((r["fallback_reason"] for r in reps if r["fallback_reason"]), None), "replicas": reps}

    def synchronize(self):
        for _, done in self._inflight:
            done.synchronize()
        self._inflight.clear()
        return self.status()

    # ------------------------------------------------------------------ lanes
    def _streams(self):
        """`depth` HIP streams on hardware queues of their own (HIP serves all streams from a few queues: two lanes behind one
        queue would run one after the other -- PWCDCNet's probe decides on device timestamps).  Fewer vetted streams than asked
        for: the pipeline is as deep as the streams found; none: the caller's stream (a plain loop)."""
        if self._lanes is not None:
            return self._lanes
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            lanes, report = None, None
            for want in range(self.depth, 0, -1):
                lanes, report = _pick_side_streams(self.device, main, want)
                if lanes is not None:
                    break
        self.stream_report = report
        self._lanes = lanes if lanes is not None else [None]
        self.effective_depth = len(self._lanes) if lanes is not None else 1
        return self._lanes

    # ------------------------------------------------------------------ forwards
    def submit(self, images_0, images_1):
        if not (isinstance(images_0, torch.Tensor) and images_0.is_cuda and images_0.device == self.device):
            raise ValueError(f"ForwardPipeline.submit: images must be CUDA tensors on {self.device} (the lanes' streams live there), "
                             f"got {getattr(images_0, 'device', type(images_0))}")
        lanes = self._streams()
        k = self._next % len(lanes)
        self._next += 1
        net, lane = self.nets[k], lanes[k]
        dev = images_0.device
        if lane is None:                                    # no vetted stream: the caller's own
            out = net(images_0, images_1)
            done = torch.cuda.Event()
            done.record()
            return Ticket(out, done, None)
        cur = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        lane.wait_event(ready)
        with torch.cuda.stream(lane):
            out = net(images_0, images_1)
            done = torch.cuda.Event()
            done.record(lane)
        for t in (images_0, images_1):
            if isinstance(t, torch.Tensor):
                t.record_stream(lane)                       # read on the lane's stream: the allocator must not hand the block on earlier
        self._inflight.append((k, done))
        while len(self._inflight) > 4 * len(lanes):
            self._inflight.popleft()
        return Ticket(out, done, k)

    def __call__(self, images_0, images_1):
        return self.submit(images_0, images_1).result()
