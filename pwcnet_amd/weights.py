"""Variable inventory, initialisation and physical channel layouts of PWCDCNet.

Names and shapes follow the reference's TensorFlow variables exactly (they are what its
checkpoints store): ``<model>/<scope>/conv2d[_k]/{kernel,bias}`` with
kernel HWIO (3,3,Cin,Cout); the k-th tf.layers.Conv2D created inside a variable scope is
``conv2d`` for k = 0 and ``conv2d_k`` after (reference modules.py:58-67, 266-274, 305-324).

Host-side only (numpy): usable without a GPU.
"""
import math

import numpy as np

FILTERS_FP = [16, 32, 64, 96, 128, 192]        # reference modules.py:46
FILTERS_OF = [128, 128, 96, 64, 32]            # reference modules.py:235
CONTEXT = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1), (2, 1)]  # modules.py:306-324
SCALES = [None, 0.625, 1.25, 2.5, 5.0, 10.0, 20.0]   # reference model.py:93


def var_name(model, scope, k):
    return f"{model}/{scope}/conv2d" + ("" if k == 0 else f"_{k}")


def pyramid_channels(num_levels=6):
    """Feature channels per pyramid level in the model's deep->shallow order
    (reference modules.py:71 reverses the list)."""
    return FILTERS_FP[:num_levels][::-1]


def estimator_in_channels(level, use_dc, num_levels=6, search_range=4):
    """Channels of concat[cv, features_0, flows_up_prev, features_up_prev]
    (reference modules.py:261-264) at pyramid level `level` (0 = coarsest)."""
    cv = (2 * search_range + 1) ** 2
    c = pyramid_channels(num_levels)[level]
    if level == 0:
        return cv + c
    return cv + c + 2 + estimator_feature_channels(level - 1, use_dc, num_levels, search_range)


def estimator_feature_channels(level, use_dc, num_levels=6, search_range=4):
    """Channels of the `features` tensor an estimator hands on (modules.py:269-272)."""
    if not use_dc:
        return FILTERS_OF[-1]
    return estimator_in_channels(level, use_dc, num_levels, search_range) + sum(FILTERS_OF)


def conv_specs(num_levels=6, search_range=4, use_dc=False, output_level=4, name="pwcdcnet"):
    """Ordered [(variable scope name, Cin, Cout)] of every convolution the forward
    creates (estimators above output_level are never called, hence have no variables:
    reference model.py:102-132)."""
    specs = []
    cin, k = 3, 0
    for l in range(num_levels):
        for _ in range(3):
            specs.append((var_name(name, "fp_extractor", k), cin, FILTERS_FP[l]))
            cin = FILTERS_FP[l]
            k += 1
    for l in range(output_level + 1):
        cin = estimator_in_channels(l, use_dc, num_levels, search_range)
        for k, f in enumerate(FILTERS_OF):
            specs.append((var_name(name, f"optflow_{l}", k), cin, f))
            cin = cin + f if use_dc else f
        specs.append((var_name(name, f"optflow_{l}", len(FILTERS_OF)), cin, 2))
    cin = 2 + estimator_feature_channels(output_level, use_dc, num_levels, search_range)
    for k, (f, _) in enumerate(CONTEXT):
        specs.append((var_name(name, "context", k), cin, f))
        cin = f
    return specs


def num_parameters(specs):
    return sum(9 * cin * cout + cout for _, cin, cout in specs)


def init_weights(specs, seed=0):
    """tf.layers.Conv2D defaults: kernel glorot_uniform (limit = sqrt(6/(fan_in+fan_out)),
    fan = 9*C), bias zeros.  Deterministic numpy stream, one draw per variable in order."""
    rng = np.random.RandomState(seed)
    w = {}
    for name, cin, cout in specs:
        limit = math.sqrt(6.0 / (9 * cin + 9 * cout))
        w[name + "/kernel"] = rng.uniform(-limit, limit, size=(3, 3, cin, cout)).astype(np.float32)
        w[name + "/bias"] = np.zeros((cout,), np.float32)
    return w


def randomize_biases(weights, seed=1, scale=0.05):
    """Test helper: non-zero biases so that bias handling is actually exercised."""
    rng = np.random.RandomState(seed)
    out = dict(weights)
    for k in sorted(weights):
        if k.endswith("/bias"):
            out[k] = rng.uniform(-scale, scale, size=weights[k].shape).astype(np.float32)
    return out


# ---------------------------------------------------------------- physical layouts

def _round_up(v, m):
    return (v + m - 1) // m * m


class ChannelLayout:
    """Physical channel order of an activation buffer.

    ``phys2log[p]`` is the logical (TensorFlow concat order) channel stored at physical
    channel p, or -1 for a zero padding channel.  Segments start on multiples of 4
    channels (16 bytes) so every kernel can use float4 accesses; the total is a multiple
    of 16 (the MFMA k-group).  tf.concat then costs nothing: producers write straight
    into their segment and the weight packer permutes Cin with this map.
    """

    def __init__(self):
        self.phys2log = []
        self.segments = {}      # name -> (phys offset, logical length)
        self.external = {}      # name -> (first logical channel, length): segments that live in ANOTHER tensor
        self.n_logical = 0

    def add(self, name, length, log_map=None):
        """Append a segment of `length` physical channels.  log_map (optional) gives the
        local logical index (or -1) of each of them; default is 0..length-1."""
        off = len(self.phys2log)
        assert off % 4 == 0
        if log_map is None:
            log_map = list(range(length))
        n_log = max([m for m in log_map if m >= 0], default=-1) + 1
        self.phys2log += [(m + self.n_logical) if m >= 0 else -1 for m in log_map]
        self.segments[name] = (off, length)
        self.n_logical += n_log
        pad = _round_up(len(self.phys2log), 4) - len(self.phys2log)
        self.phys2log += [-1] * pad
        return off

    def reserve(self, name, length):
        """A segment of the concat that stays in the tensor it comes from (round 5: the estimator's first conv reads
        features_0 from the pyramid tensor through a second operand pointer): it takes its place in the LOGICAL order and no
        physical channels of this buffer."""
        self.external[name] = (self.n_logical, length)
        self.n_logical += length

    def cin_map_with(self, name):
        """cin_map() of the whole buffer followed by the logical channels of the external segment `name`: the physical
        channel order of a two-operand conv (buffer channels, then the other tensor's)."""
        first, length = self.external[name]
        return np.concatenate([self.cin_map(0, 0), np.arange(first, first + length, dtype=np.int32)])

    def finish(self, multiple=16):
        pad = _round_up(len(self.phys2log), multiple) - len(self.phys2log)
        self.phys2log += [-1] * pad
        return self

    @property
    def n_phys(self):
        return len(self.phys2log)

    def offset(self, name):
        return self.segments[name][0]

    def cin_map(self, start=0, logical_base=0):
        """int32 map for the weight packer of a conv that reads the physical suffix
        [start:], whose kernel's input channel 0 is logical channel `logical_base`."""
        m = np.asarray(self.phys2log[start:], np.int32).copy()
        m[m >= 0] -= logical_base
        assert (m[m != -1] >= 0).all()
        return m


def estimator_layout(level, use_dc, num_levels=6, search_range=4):
    """Layout of the buffer that holds an estimator's (growing) `features` tensor.

    Logical order = TF concat order.  Non-DC: [cv | f0 | flow_up | feat_up]
    (modules.py:261-264).  DC: each conv output is prepended (modules.py:269-270), so the
    final tensor is [conv5|conv4|conv3|conv2|conv1 | cv | f0 | flow_up | feat_up] and conv k
    reads a physical suffix of it."""
    cvc = (2 * search_range + 1) ** 2
    c = pyramid_channels(num_levels)[level]
    lay = ChannelLayout()
    if use_dc:
        for k in reversed(range(len(FILTERS_OF))):
            lay.add(f"conv{k}", FILTERS_OF[k])
    lay.add("cv", cvc)
    lay.add("f0", c)
    if level > 0:
        lay.add("flow", 2)
        if use_dc:
            prev = estimator_layout(level - 1, True, num_levels, search_range)
            lay.add("feat_up", prev.n_phys, log_map=prev.phys2log)
        else:
            lay.add("feat_up", FILTERS_OF[-1])
    return lay.finish(16)


def context_layout(use_dc, output_level=4, num_levels=6, search_range=4):
    """[flows | features] (modules.py:305)."""
    lay = ChannelLayout()
    lay.add("flow", 2)
    if use_dc:
        est = estimator_layout(output_level, True, num_levels, search_range)
        lay.add("features", est.n_phys, log_map=est.phys2log)
    else:
        lay.add("features", FILTERS_OF[-1])
    return lay.finish(16)
