"""Training step of PWCDCNet on the HIP kernels -- counterpart of the graph the reference builds in
train.py:43-92 (forward, multiscale_loss / multirobust-style level terms, gamma * l2_loss(vars),
tf.train.AdamOptimizer with the piecewise-constant learning rate, global_step), SURVEY.md 8f-4.

One process per GPU; data-parallel ranks average their gradients with ONE all-reduce of the flat
gradient buffer per step (RCCL over xGMI under torch.distributed's `nccl` backend): the model has
5.03 M parameters = 20 MB, far below the size where bucketing or overlap with the backward would pay.

Design: all variables live in ONE flat fp32 buffer (same for gradients and the two Adam moments), each
TensorFlow variable is a view into it -- the optimiser is a single kernel launch and the all-reduce a
single collective.  The forward keeps every activation (this is what tf.gradients does); the backward is
written out by hand, level by level, on pwcnet_amd.grad_ops.  Gradient buffers have the layout of the
activation they belong to, including the padded physical channel layouts of the estimator inputs.

Both use_dc settings: with dense connections (modules.py:269-270) an estimator's `features` tensor is ONE
buffer [conv4|conv3|conv2|conv1|conv0|cv|f0|flow_up|feat_up] (weights.estimator_layout), conv k reads a
physical suffix of it and writes its own segment; the gradient buffer has the same layout, conv k's input
gradient is added onto the suffix, and by the time the backward reaches conv k-1 its segment holds the
complete gradient of that conv's output.  Both level losses: multiscale (losses.py:15-32) and the robust
form (losses.py:34-48 as intended: weight * (L1loss + epsilon) ** q -- the reference's loop names an
undefined variable there).  warp_type must be 'bilinear' (nearest warping has no gradient w.r.t. the flow).
"""
import math

import numpy as np
import torch

from . import grad_ops as G
from . import modules as M
from .modules import View, sub_view
from .sharding import allreduce_sum_
from .weights import CONTEXT, FILTERS_OF, SCALES, context_layout, conv_specs, estimator_layout, init_weights


def piecewise_lr(lr, step, scheduling=True):
    """reference train.py:82-88."""
    if not scheduling:
        return lr
    boundaries = [200000, 250000, 300000, 350000, 4000000]
    i = sum(1 for b in boundaries if step > b)          # tf.train.piecewise_constant: x <= boundaries[0] -> values[0]
    return lr / (2 ** i)


class _Conv:
    """One tf.layers.Conv2D: parameter views + what the backward needs from the forward."""

    def __init__(self, name, cin, cout, kernel, bias, dkernel, dbias, stride=1, dilation=1, act=True):
        self.name, self.cin, self.cout = name, cin, cout
        self.kernel, self.bias, self.dkernel, self.dbias = kernel, bias, dkernel, dbias
        self.stride, self.dilation, self.act = stride, dilation, act
        self.x = self.y = self.y_t = None      # input view, output view, output tensor
        self.cin_map = None                     # physical -> logical map of the input buffer (int32 device tensor)
        self.w_phys = None                      # kernel in the input's physical channel order


class Trainer:
    """Data-parallel semantics: every rank steps on its own pairs and the flat gradient buffer is summed over the ranks
    with weight 1/world.  For the multiscale loss (a mean over the batch) that IS the gradient of the global batch.  For
    the robust loss, weight * (mean_n L1 + eps)^q is not linear in the batch mean: N ranks at batch b follow the mean of
    the per-rank robust losses, not the robust loss of the batch of N*b (the reference trains on one device).  The
    per-level scale q * (L1 + eps)^(q-1) is read back to the host once per level and step."""

    def __init__(self, num_levels=6, search_range=4, warp_type="bilinear", use_dc=False, output_level=4,
                 name="pwcdcnet", weights=(0.32, 0.08, 0.02, 0.01, 0.005), gamma=0.0004, lr=1e-4, lr_scheduling=True,
                 seed=0, device="cuda", dist=None, loss="multiscale", epsilon=0.01, q=0.4, f16x2=True, f16x2_dgrad=False):
        assert loss in ("multiscale", "robust"), loss
        # the F16-pipe convolution kernels (operands as fp16 pairs) in the forward / in the data gradient: grad_ops.F16X2,
        # grad_ops.F16X2_DGRAD (the reasons for the defaults are there); applied at every step()
        self.f16x2, self.f16x2_dgrad = bool(f16x2), bool(f16x2_dgrad)
        self.use_dc, self.loss, self.epsilon, self.q = bool(use_dc), loss, float(epsilon), float(q)
        assert warp_type == "bilinear", "training needs the bilinear warp"
        assert num_levels == 6 and search_range == 4 and output_level < num_levels
        self.num_levels, self.s_range, self.output_level, self.name = num_levels, search_range, output_level, name
        self.loss_weights, self.gamma, self.lr, self.lr_scheduling = list(weights), gamma, lr, lr_scheduling
        self.device = torch.device(device)
        self.dist = dist
        self.specs = conv_specs(num_levels, search_range, self.use_dc, output_level, name)
        # every variable starts on a 16-byte boundary of the flat buffer (the kernels read weights and biases with
        # vector loads); the padding floats stay zero under Adam
        pad4 = lambda k: (k + 3) // 4 * 4
        n = sum(pad4(9 * ci * co) + pad4(co) for _, ci, co in self.specs)
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.global_step = 0
        self.views = {}
        off = 0
        for vname, ci, co in self.specs:
            for suffix, shape in (("/kernel", (3, 3, ci, co)), ("/bias", (co,))):
                k = int(np.prod(shape))
                self.views[vname + suffix] = (off, shape)
                off += pad4(k)
        self.load_weights(init_weights(self.specs, seed=seed))
        _lib_check = M._lib.lib()   # fail loudly without the HIP library
        del _lib_check

    # ------------------------------------------------------------------ variables
    def _view(self, buf, key):
        off, shape = self.views[key]
        return buf[off:off + int(np.prod(shape))].view(shape)

    def load_weights(self, weights):
        missing = sorted(set(self.views) - set(weights))
        if missing:
            raise ValueError(f"load_weights: {len(missing)} variables missing, e.g. {missing[:3]}")
        for k in self.views:
            self._view(self.params, k).copy_(torch.as_tensor(np.ascontiguousarray(weights[k], np.float32)).to(self.device))

    def state_dict(self):
        return {k: self._view(self.params, k).detach().cpu().numpy().copy() for k in self.views}

    def gradients(self):
        return {k: self._view(self.grads, k).detach().cpu().numpy().copy() for k in self.views}

    def _conv(self, scope, k, stride=1, dilation=1, act=True):
        vname = f"{self.name}/{scope}/conv2d" + ("" if k == 0 else f"_{k}")
        _, shape = self.views[vname + "/kernel"]
        return _Conv(vname, shape[2], shape[3], self._view(self.params, vname + "/kernel"), self._view(self.params, vname + "/bias"),
                     self._view(self.grads, vname + "/kernel"), self._view(self.grads, vname + "/bias"), stride, dilation, act)

    # ------------------------------------------------------------------ forward pieces
    def _run_conv(self, c, x, y=None, p2l=None):
        """y = conv(x) [+lrelu]; x a View over physical channels; p2l: physical -> logical (kernel) input channel of
        each of them, -1 for zero padding (None: identity)."""
        dev = self.device
        Ho, Wo = -(-x.H // c.stride), -(-x.W // c.stride)
        if p2l is not None:
            p2l = np.asarray(p2l, np.int32)
            assert len(p2l) == x.C and int(p2l.max()) + 1 == c.cin, (c.name, len(p2l), x.C, int(p2l.max()) + 1, c.cin)
            c.cin_map = torch.from_numpy(p2l).to(dev)
            valid = torch.from_numpy(np.nonzero(p2l >= 0)[0]).to(dev)
            src = torch.from_numpy(p2l[p2l >= 0].astype(np.int64)).to(dev)
            w = torch.zeros((3, 3, x.C, c.cout), dtype=torch.float32, device=dev)
            w[:, :, valid, :] = c.kernel[:, :, src, :]
            c.w_phys = w
        else:
            assert x.C == c.cin
            c.cin_map, c.w_phys = None, c.kernel
        if y is None:
            c.y_t = torch.empty((x.N, Ho, Wo, c.cout), dtype=torch.float32, device=dev)
            y = View(c.y_t.data_ptr(), c.cout, x.N, Ho, Wo, c.cout)
        c.x, c.y = x, y
        G.conv3x3_raw(x, c.w_phys, c.bias, y, c.stride, c.dilation, 0.1 if c.act else None, keep=self._keep)
        return y

    def forward(self, images_0, images_1):
        """Returns flows_pyramid (list of 5 (N,h,w,2) tensors, px/20 units) and keeps the activations."""
        G.F16X2, G.F16X2_DGRAD = self.f16x2, self.f16x2_dgrad
        dev = self.device
        self._keep = []
        N, H, W, _ = images_0.shape
        assert H % 64 == 0 and W % 64 == 0, "image sizes must be multiples of 64 (reference test.py:13-17)"
        x_t = torch.cat([images_0, images_1], dim=0).contiguous()
        self._keep.append(x_t)
        x = View(x_t.data_ptr(), 3, 2 * N, H, W, 3)
        self.ext = []
        feats = []
        k = 0
        for l in range(self.num_levels):
            for j in range(3):
                c = self._conv("fp_extractor", k, stride=2 if j == 0 else 1)
                x = self._run_conv(c, x)
                self.ext.append(c)
                k += 1
            feats.append(self.ext[-1])
        feats = feats[::-1]                        # deep -> shallow
        self.levels = []
        flows_pyramid = []
        prev = None
        dc = self.use_dc
        for l, fc in enumerate(feats):
            F = fc.y
            h, w, C = F.H, F.W, F.C
            f0 = View(F.ptr, C, N, h, w, C)
            f1 = View(F.ptr + 4 * N * h * w * C, C, N, h, w, C)
            # E: the estimator's `features` buffer (dense connections: with the five conv segments in front)
            lay = estimator_layout(l, dc, self.num_levels, self.s_range)
            nfeat = lay.segments["feat_up"][1] if l > 0 else 0
            E_t = torch.zeros((N, h, w, lay.n_phys), dtype=torch.float32, device=dev)
            E = View(E_t.data_ptr(), lay.n_phys, N, h, w, lay.n_phys)
            L = dict(l=l, fc=fc, f0=f0, f1=f1, lay=lay, E_t=E_t, E=E, C=C, nfeat=nfeat)
            if l > 0:
                # x2 upsampling of the previous level's flows and features into this level's buffer (modules.py:282-285)
                M._resize(prev["flows_v"], sub_view(E, lay.offset("flow"), 2))
                M._resize(prev["feat"], sub_view(E, lay.offset("feat_up"), nfeat))
                f1w_t = torch.empty((N, h, w, C), dtype=torch.float32, device=dev)
                f1w = View(f1w_t.data_ptr(), C, N, h, w, C)
                M.WarpingLayer("bilinear")._run(f1, sub_view(E, lay.offset("flow"), 2), f1w, flow_scale=SCALES[l])
                L["f1w_t"], L["f1w"] = f1w_t, f1w
            else:
                L["f1w"] = f1
            cv_v = sub_view(E, lay.offset("cv"), 81)
            M.CostVolumeLayer(self.s_range)._run(f0, L["f1w"], cv_v)
            M._copy_channels(f0, sub_view(E, lay.offset("f0"), C), C)
            convs = []
            xin, start = E, lay.offset("cv")
            for kk in range(5):
                c = self._conv(f"optflow_{l}", kk)
                if dc:
                    # conv kk reads the suffix that starts at conv kk-1's segment and writes its own segment
                    xin = sub_view(E, start, lay.n_phys - start)
                    yv = sub_view(E, lay.offset(f"conv{kk}"), c.cout)
                    self._run_conv(c, xin, y=yv, p2l=lay.cin_map(start, lay.phys2log[start]))
                    start = lay.offset(f"conv{kk}")
                else:
                    xin = self._run_conv(c, xin, p2l=lay.phys2log if kk == 0 else None)
                convs.append(c)
            head = self._conv(f"optflow_{l}", 5, act=False)
            flows_t = torch.empty((N, h, w, 2), dtype=torch.float32, device=dev)
            flows_v = View(flows_t.data_ptr(), 2, N, h, w, 2)
            self._run_conv(head, E if dc else xin, y=flows_v, p2l=lay.phys2log if dc else None)
            head.y_t = flows_t
            if l > 0:
                G.add_(sub_view(E, lay.offset("flow"), 2), flows_v)          # flows += flows_up_prev (modules.py:275-277)
            # `features` handed to the next level / the context network: everything (dc) or conv4's output
            L.update(convs=convs, head=head, flows_t=flows_t, flows_v=flows_v, feat=E if dc else convs[4].y)
            if l == self.output_level:
                cl = context_layout(dc, self.output_level, self.num_levels, self.s_range)
                nf = cl.segments["features"][1]
                CX_t = torch.zeros((N, h, w, cl.n_phys), dtype=torch.float32, device=dev)
                CX = View(CX_t.data_ptr(), cl.n_phys, N, h, w, cl.n_phys)
                G.add_(flows_v, sub_view(CX, 0, 2), accumulate=False)
                G.add_(L["feat"], sub_view(CX, cl.offset("features"), nf), accumulate=False)
                ctx = []
                xin = CX
                for kk, (f, d) in enumerate(CONTEXT):
                    c = self._conv("context", kk, dilation=d, act=kk < len(CONTEXT) - 1)
                    if kk == len(CONTEXT) - 1:
                        out_t = torch.empty((N, h, w, 2), dtype=torch.float32, device=dev)
                        xin = self._run_conv(c, xin, y=View(out_t.data_ptr(), 2, N, h, w, 2))
                        c.y_t = out_t
                    else:
                        xin = self._run_conv(c, xin, p2l=cl.phys2log if kk == 0 else None)
                    ctx.append(c)
                final_t = flows_t + ctx[-1].y_t                                # return flows + x (modules.py:326)
                L.update(ctx=ctx, cl=cl, CX_t=CX_t, CX=CX, final_t=final_t)
                self.levels.append(L)
                flows_pyramid.append(final_t)
                break
            self.levels.append(L)
            flows_pyramid.append(flows_t)
            prev = L
        self.flows_pyramid = flows_pyramid
        return flows_pyramid

    # ------------------------------------------------------------------ backward pieces
    def _conv_backward(self, c, dy_t, need_dx=True, dy=None):
        """dy_t: gradient w.r.t. the conv's (activated) output as a dense tensor, or dy: the same as a View into a
        wider buffer; modified in place.  Fills the variable gradients; returns the gradient w.r.t. the conv's input
        (dense tensor over the physical channels of c.x)."""
        dev = self.device
        if dy is None:
            dy = View(dy_t.data_ptr(), c.cout, c.y.N, c.y.H, c.y.W, c.cout)
        if c.act:
            G.lrelu_grad_channel_sums_(c.y, dy, c.dbias, dev)      # activation mask + bias gradient in one pass over dy
        else:
            G.channel_sums(dy, c.dbias, dev)
        G.conv3x3_wgrad(c.x, dy, c.dkernel, c.cin, c.stride, c.dilation, cin_map=c.cin_map)
        if not need_dx:
            return None
        dx_t = torch.empty((c.x.N, c.x.H, c.x.W, c.x.C), dtype=torch.float32, device=dev)
        dx = View(dx_t.data_ptr(), c.x.C, c.x.N, c.x.H, c.x.W, c.x.C)
        G.conv3x3_dgrad(dy, c.w_phys, dx, c.stride, c.dilation, keep=self._keep, dy_tensor=dy_t)
        return dx_t

    def _estimator_backward(self, L, d_est, dfeat):
        """d_est (N,h,w,2): gradient w.r.t. the head's output; dfeat: gradient w.r.t. the `features` the estimator
        hands on (None, or a tensor shaped like L['feat']).  Returns the gradient of the estimator's input buffer E
        (all physical channels of its layout)."""
        N, h, w = L["E"].N, L["E"].H, L["E"].W
        lay = L["lay"]
        if not self.use_dc:
            d_head = self._conv_backward(L["head"], d_est.clone())
            dcur = dfeat + d_head if dfeat is not None else d_head
            for kk in range(4, 0, -1):
                dcur = self._conv_backward(L["convs"][kk], dcur)
            dE_t = self._conv_backward(L["convs"][0], dcur)
            self._keep += [dcur]
            return dE_t
        # dense connections: dE_t has the layout of E; every conv adds its input gradient onto the suffix it read
        dE_t = dfeat if dfeat is not None else torch.zeros((N, h, w, lay.n_phys), dtype=torch.float32, device=self.device)
        dE = View(dE_t.data_ptr(), lay.n_phys, N, h, w, lay.n_phys)
        t = self._conv_backward(L["head"], d_est.clone())
        G.add_(View(t.data_ptr(), lay.n_phys, N, h, w, lay.n_phys), dE)
        self._keep.append(t)
        for kk in range(4, -1, -1):
            c = L["convs"][kk]
            t = self._conv_backward(c, None, dy=sub_view(dE, lay.offset(f"conv{kk}"), c.cout))
            start = lay.n_phys - c.x.C
            G.add_(View(t.data_ptr(), c.x.C, N, h, w, c.x.C), sub_view(dE, start, c.x.C))
            self._keep.append(t)
        return dE_t

    def _level_loss_scale(self, l, flows_gt):
        """(ord, scale) of level l's loss term for flow_norm_grad: d/dpred of scale * sum_p ||pred - gt||_ord."""
        N = flows_gt.shape[0]
        wl = self.loss_weights[l]
        if self.loss == "multiscale":
            return 2, wl / N                                               # weight * mean_n sum_p ||.||_2
        from . import losses
        l1 = float(losses._norm_sums(self.flows_pyramid[l], flows_gt, 1, gt_div=20.0)[0].mean())   # L1loss(gt_down, fs)
        return 1, wl * self.q * (l1 + self.epsilon) ** (self.q - 1.0) / N

    def backward(self, flows_gt):
        """Gradients of the level losses (losses.py:15-32 / :34-48) w.r.t. every variable, into self.grads (the
        gamma * l2_loss term is applied by the optimiser kernel)."""
        G.F16X2, G.F16X2_DGRAD = self.f16x2, self.f16x2_dgrad
        dev = self.device
        N = flows_gt.shape[0]
        gt = View(flows_gt.data_ptr(), 2, N, flows_gt.shape[1], flows_gt.shape[2], 2)
        self.grads.zero_()
        nl = len(self.levels)
        # loss gradient of every pyramid flow
        dflows = []
        for l, L in enumerate(self.levels):
            fl = self.flows_pyramid[l]
            d = torch.empty_like(fl)
            order, scale = self._level_loss_scale(l, flows_gt)
            G.flow_norm_grad(View(fl.data_ptr(), 2, N, fl.shape[1], fl.shape[2], 2), gt,
                             View(d.data_ptr(), 2, N, fl.shape[1], fl.shape[2], 2), gt_div=20.0, ord=order, scale=scale)
            dflows.append(d)
        # gradient of the pyramid features (2N stacked, deep -> shallow)
        dF = [torch.zeros((2 * N, L["f0"].H, L["f0"].W, L["C"]), dtype=torch.float32, device=dev) for L in self.levels]
        dfeat_next = None        # gradient of this level's `features` coming from the next level's feat_up
        for l in range(nl - 1, -1, -1):
            L = self.levels[l]
            lay, E, h, w, C = L["lay"], L["E"], L["f0"].H, L["f0"].W, L["C"]
            d_est = dflows[l]                                   # gradient w.r.t. this level's estimator flows
            if l == self.output_level:
                # flows_final = flows_est + context(flows_est, features): the loss gradient reaches both terms
                ctx, cl = L["ctx"], L["cl"]
                nf = cl.segments["features"][1]
                dcur = d_est.clone()
                for kk in range(len(ctx) - 1, 0, -1):
                    dcur = self._conv_backward(ctx[kk], dcur)
                dCX_t = self._conv_backward(ctx[0], dcur)
                dCX = View(dCX_t.data_ptr(), cl.n_phys, N, h, w, cl.n_phys)
                G.add_(sub_view(dCX, 0, 2), View(d_est.data_ptr(), 2, N, h, w, 2))
                dfeat = torch.empty((N, h, w, nf), dtype=torch.float32, device=dev)
                G.add_(sub_view(dCX, cl.offset("features"), nf), View(dfeat.data_ptr(), nf, N, h, w, nf), accumulate=False)
                self._keep += [dCX_t, dcur]
            else:
                dfeat = dfeat_next
            dE_t = self._estimator_backward(L, d_est, dfeat)
            dE = View(dE_t.data_ptr(), lay.n_phys, N, h, w, lay.n_phys)
            dF0 = View(dF[l].data_ptr(), C, N, h, w, C)
            dF1 = View(dF[l].data_ptr() + 4 * N * h * w * C, C, N, h, w, C)
            G.add_(sub_view(dE, lay.offset("f0"), C), dF0)                              # features_0 part of the concat
            cv_v = sub_view(E, lay.offset("cv"), 81)
            if l > 0:
                dflow = sub_view(dE, lay.offset("flow"), 2)
                G.add_(View(d_est.data_ptr(), 2, N, h, w, 2), dflow)                     # residual flows += flows_up_prev
                df1w_t = torch.empty((N, h, w, C), dtype=torch.float32, device=dev)
                df1w = View(df1w_t.data_ptr(), C, N, h, w, C)
                G.cost_volume_grad(L["f0"], L["f1w"], cv_v, sub_view(dE, lay.offset("cv"), 81), dF0, None, accumulate=True)
                G.cost_volume_grad(L["f0"], L["f1w"], cv_v, sub_view(dE, lay.offset("cv"), 81), None, df1w, accumulate=False)
                G.warp_grad(L["f1"], sub_view(E, lay.offset("flow"), 2), SCALES[l], df1w, dF1, dflow, dflow_accumulate=True)
                # x2 resizes into this level's buffer: back to the previous level's flows and features
                P = self.levels[l - 1]
                ph, pw, nfeat = P["f0"].H, P["f0"].W, L["nfeat"]
                G.resize_grad(dflow, View(dflows[l - 1].data_ptr(), 2, N, ph, pw, 2), accumulate=True)
                dfeat_next = torch.empty((N, ph, pw, nfeat), dtype=torch.float32, device=dev)
                G.resize_grad(sub_view(dE, lay.offset("feat_up"), nfeat), View(dfeat_next.data_ptr(), nfeat, N, ph, pw, nfeat))
                self._keep += [df1w_t]
            else:
                G.cost_volume_grad(L["f0"], L["f1w"], cv_v, sub_view(dE, lay.offset("cv"), 81), dF0, dF1, accumulate=True)
            self._keep += [dE_t, dfeat]
        # extractor, deep -> shallow; level index in self.ext order is shallow -> deep
        carry = None
        for li in range(self.num_levels - 1, -1, -1):
            lvl = self.num_levels - 1 - li                    # position in self.levels / dF (deep -> shallow)
            if lvl < nl:
                dcur = dF[lvl] if carry is None else dF[lvl] + carry
            else:
                dcur = carry                                   # pyramid levels no estimator reads
            if dcur is None:
                continue
            for j in (2, 1):
                dcur = self._conv_backward(self.ext[3 * li + j], dcur)
            carry = self._conv_backward(self.ext[3 * li], dcur, need_dx=li > 0)

    # ------------------------------------------------------------------ step
    def loss_value(self, flows_gt):
        from . import losses
        if self.loss == "robust":
            return losses.multirobust_loss(flows_gt, self.flows_pyramid, self.loss_weights, self.epsilon, self.q)
        return losses.multiscale_loss(flows_gt, self.flows_pyramid, self.loss_weights)

    def status(self):
        """Status words of the training path's F16-pipe launches since the last call (synchronises): 0, or
        _lib.STATUS_STREAMK_TIMEOUT when a stream-K wait ran out somewhere (the affected step's loss is NaN; the stream-K
        workspaces have been refilled: repeat the step)."""
        return G.read_status()

    def step(self, images_0, images_1, flows_gt):
        """One optimisation step (reference train.py:66-92, 114-118).  Returns the data loss (without the L2 term)."""
        self.forward(images_0, images_1)
        loss = self.loss_value(flows_gt)
        self.backward(flows_gt)
        world = allreduce_sum_(self.grads, self.dist)         # RCCL: one 20 MB collective per step
        self.global_step += 1
        t = self.global_step
        lr = piecewise_lr(self.lr, t - 1, self.lr_scheduling)
        lr_t = lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        G.adam_step_(self.params, self.grads, self.m, self.v, lr_t, l2_gamma=self.gamma, grad_scale=1.0 / world)
        self._keep = []
        return loss
