"""The engine behind the four drop-in layer classes: parameter/buffer set-up with the
reference's names, and a forward that is ONE call into libbtorch_hip.so.

Reference behaviour reproduced (paths under /root/reference/bayesian_torch/layers/):
  parameters / buffers / init     variational_layers/linear_variational.py:90-144, conv_variational.py:291-351
  forward(input, return_kl=True)  linear_variational.py:160-204, conv_variational.py:362-407,
                                  flipout_layers/linear_flipout.py:145-197, conv_flipout.py:370-439
  kl_loss()                       linear_variational.py:146-158
Training: the same fused forward; gradients come from the autograd bridge (autograd.py: draws regenerated from the RNG
coordinates, ATen conv/matmul backward) -- a fused HIP backward is the next step.
"""
import ctypes as C
import warnings

import torch
from torch.nn import Parameter

from .. import _lib, mc, rng
from .. import functional as F
from .base_variational_layer import BaseVariationalLayer_, check_prior_type, get_kernel_size

_warned = [False]


import os as _os
_NO_SEG_CACHE = bool(_os.environ.get("BT_NO_SEG_CACHE"))   # A/B knob of the eager path's host time (tools/profile_eager.py)


class FusedBayesLayer(BaseVariationalLayer_):
    _kind = "linear"     # or "conv"
    _flip = False
    _wname = "weight"    # parameter suffix: mu_weight / mu_kernel

    # ------------------------------------------------------------------ construction
    def _build(self, wshape, bias, eps_bias_last=False, n_out=None):
        wn = self._wname
        n_out = wshape[0] if n_out is None else n_out     # (transposed convolutions: the kernel is [Ci][Co/g]..., the bias [Co])
        self.register_parameter("mu_" + wn, Parameter(torch.empty(wshape)))
        self.register_parameter("rho_" + wn, Parameter(torch.empty(wshape)))
        self.register_buffer("eps_" + wn, torch.zeros(wshape), persistent=False)
        self.register_buffer("prior_weight_mu", torch.empty(wshape), persistent=False)
        self.register_buffer("prior_weight_sigma", torch.empty(wshape), persistent=False)
        if bias:
            self.mu_bias = Parameter(torch.empty(n_out))
            self.rho_bias = Parameter(torch.empty(n_out))
            if not eps_bias_last:
                self.register_buffer("eps_bias", torch.zeros(n_out), persistent=False)
            self.register_buffer("prior_bias_mu", torch.empty(n_out), persistent=False)
            self.register_buffer("prior_bias_sigma", torch.empty(n_out), persistent=False)
            if eps_bias_last:   # LinearFlipout registers eps_bias after the priors (linear_flipout.py:98-104)
                self.register_buffer("eps_bias", torch.zeros(n_out), persistent=False)
        else:
            self.register_parameter("mu_bias", None)
            self.register_parameter("rho_bias", None)
            for b in ("eps_bias", "prior_bias_mu", "prior_bias_sigma"):
                self.register_buffer(b, None, persistent=False)
        self._layer_id = rng.new_layer_id()   # RNG coordinate; dnn_to_bnn / rng.assign_layer_ids renumber it by position
        self._pack = None          # (key, mu_packed, sigma_packed, state): persistent buffers, kept in step by bt_pack_sync
        self._pack_force = False
        self._last = None
        self.post_relu = False    # fused output stage, set by bayesian_torch_amd.fuse (inference-time folding)
        self.post_pool = False    # ... followed by MaxPool2d(3, 2, 1) (fuse.fold_maxpool: the ResNet stem)
        self.register_buffer("post_scale", None, persistent=False)
        self.register_buffer("post_shift", None, persistent=False)
        self.inject_draw = None   # test hook: dict(eps_w [S,*w], eps_b, sign_in, sign_out) consumed instead of a fresh draw (or a list of them)
        self.init_parameters()
        self.quant_prepare = False

    @property
    def _ws_id(self):
        """Key of this layer's KL / pack workspaces: unique among live layers (copies of a model included), unlike the positional
        RNG coordinate ``_layer_id`` -- two converted models may run concurrently on different streams."""
        return id(self)

    def _init_scalars(self):
        mu0, rho0 = self.posterior_mu_init, self.posterior_rho_init
        if isinstance(mu0, tuple):   # the Reparameterization classes keep 1-tuples (trailing commas in the reference)
            mu0, rho0 = mu0[0], rho0[0]
        return mu0, rho0

    def invalidate_pack(self):
        """Force a rebuild of the packed (mu, sigma) copy at the next forward. Never NEEDED: every forward checks a device-side
        fingerprint of (mu, rho) against the one its pack was built from (bt_pack_sync), so writes through ``.data``
        (``p.data.copy_()`` / ``.normal_()``: the reference's own idiom in dnn_to_bnn / MOPED, invisible to ``p._version``),
        optimizer steps and load_state_dict are all picked up. Kept for callers of the round-2 API."""
        self._pack_force = True

    def init_parameters(self):
        mu0, rho0 = self._init_scalars()
        self.prior_weight_mu.fill_(self.prior_mean)
        self.prior_weight_sigma.fill_(self.prior_variance)      # "variance" is used as sigma_p, as in the reference
        self._w("mu").data.normal_(mean=mu0, std=0.1)
        self._w("rho").data.normal_(mean=rho0, std=0.1)
        if self.mu_bias is not None:
            self.prior_bias_mu.fill_(self.prior_mean)
            self.prior_bias_sigma.fill_(self.prior_variance)
            self.mu_bias.data.normal_(mean=mu0, std=0.1)
            self.rho_bias.data.normal_(mean=rho0, std=0.1)

    def prepare(self):
        raise NotImplementedError("post-training quantisation (QuantStub observers) is outside the MI355X hot path")

    def _w(self, what):
        return getattr(self, f"{what}_{self._wname}")

    # ------------------------------------------------------------------ KL
    def _kl_segments(self):
        segs = [(self._w("mu"), self._w("rho"), self.prior_weight_mu, self.prior_weight_sigma)]
        if self.mu_bias is not None:
            segs.append((self.mu_bias, self.rho_bias, self.prior_bias_mu, self.prior_bias_sigma))
        return segs

    def _prior_kind(self):
        return check_prior_type(getattr(self, "prior_type", "normal"))

    def __getstate__(self):     # (copy.deepcopy / pickle: the cached pack-check entry holds ctypes pointers and belongs to THIS object's tensors)
        d = self.__dict__.copy()
        d.pop("_seg_cache", None)
        return d

    def _param_versions(self):
        return tuple(t._version for sg in self._kl_segments() for t in sg[:2])

    def _take_live_kl(self):
        """The differentiable KL tensor the last training forward produced (FusedForward's second output), once, and only while the
        parameters are the ones it was computed from; else None (the caller launches the KL kernel)."""
        live, self._kl_live = getattr(self, "_kl_live", None), None
        if live is None or not torch.is_grad_enabled() or live[1] != self._param_versions():
            return None
        return live

    def kl_loss(self):
        live = self._take_live_kl()
        if live is not None:
            if not live[2]:
                return live[0]
            from ..autograd import KLValue      # a placeholder (training step at one sample): the value from this layer's own launch
            segs = self._kl_segments()
            return KLValue.apply((segs, [0] * len(segs), ("layer", self._ws_id), self._prior_kind() == "laplace"), live[0])
        kind = self._prior_kind()
        segs = self._kl_segments()
        if torch.is_grad_enabled() and any(t.requires_grad for sg in segs for t in sg):
            from ..autograd import KLNormal
            return KLNormal.apply((("layer", self._ws_id), kind), *[t for sg in segs for t in sg])
        return _lib.kl_normal(segs, layer_ids=[0] * len(segs), owner=("layer", self._ws_id), laplace=kind == "laplace")

    # ------------------------------------------------------------------ forward
    def _pack_source(self):
        """(mu, rho) in the [Co][Ci/g][taps...] layout the pack is built from (the family layers re-arrange theirs)."""
        return self._w("mu").detach(), self._w("rho").detach()

    def _pack_segment(self):
        """This layer's entry of a bt_pack_sync call. The pack lives in persistent buffers (a captured graph bakes their
        addresses in; they are re-allocated -- and rebuilt unconditionally -- only when a parameter TENSOR is replaced:
        ``.to(device)``, a new nn.Parameter)."""
        pm, pr = self._w("mu"), self._w("rho")
        cached = getattr(self, "_seg_cache", None)
        if (not _NO_SEG_CACHE and cached is not None and self._pack is not None and cached[0] is self._pack and cached[1] == (pm.data_ptr(), pr.data_ptr(), pm.device)
                and type(self)._pack_source is FusedBayesLayer._pack_source):
            # same parameter tensors, same persistent buffers: the entry of the last call (a layer called on its own builds one per
            # forward -- 21 per model forward in the reference's eager loop, which is host-bound)
            sg = cached[2]
            sg["force"] = self._pack_force
            self._pack_force = False
            return sg
        mu, rho = pm.detach(), pr.detach()
        smu, srho = self._pack_source()
        Co, Ci = smu.shape[0], smu.shape[1]
        taps = 1
        for d in smu.shape[2:]:
            taps *= d
        key = (mu.data_ptr(), rho.data_ptr(), tuple(smu.shape), mu.device)
        force = self._pack_force
        if self._pack is None or self._pack[0] != key:
            self._pack = (key,) + F.pack_buffers(Co, Ci, taps, mu.device)
            force = True
        self._pack_force = False
        same = smu.data_ptr() == mu.data_ptr() and srho.data_ptr() == rho.data_ptr()
        sg = dict(mu=mu, rho=rho, src_mu=None if same else smu.contiguous(), src_rho=None if same else srho.contiguous(), mu_packed=self._pack[1],
                  sigma_packed=self._pack[2], state=self._pack[3], Co=Co, Ci=Ci, taps=taps, force=force)
        self._seg_cache = (self._pack, (pm.data_ptr(), pr.data_ptr(), pm.device), sg) if same else None
        return sg

    def _packed(self):
        """(mu_packed, sigma_packed): tap-major copies of (mu, softplus(rho)) for the fast kernels, verified against the
        parameters on the device before EVERY forward (bt_pack_sync: a fingerprint sweep of (mu, rho) + a rebuild of the
        packs that differ, both in the stream). ``mc_forward`` / ``McGraph`` / ``TrainGraph`` run the check once per model
        (``mc.sync_model_packs``); a layer called on its own checks itself."""
        ctx = mc.current()
        if ctx is not None and id(self) in ctx.synced and self._pack is not None and not self._pack_force:
            if id(self) in ctx.late:
                mc.join_packs(ctx)      # this layer's pack was verified on the side stream: the launch stream waits for it (once)
            return self._pack[1], self._pack[2]
        F.pack_sync([self._pack_segment()], owner=("layer", self._ws_id))
        return self._pack[1], self._pack[2]

    def pack_rebuilds(self):
        """How many times this layer's pack has been (re)built -- a device counter kept by bt_pack_sync (synchronises)."""
        return 0 if self._pack is None else int(self._pack[3][3])

    def _conv_desc(self):
        if getattr(self, "_one_d", False):       # Conv1d: a 1 x k kernel over a 1 x L image
            one = lambda v: v[0] if isinstance(v, (tuple, list)) else v
            if isinstance(self.padding, str):
                raise NotImplementedError("string padding modes are not forwarded by dnn_to_bnn and not supported")
            return dict(stride=(1, one(self.stride)), padding=(0, one(self.padding)), dilation=(1, one(self.dilation)), groups=self.groups)
        pad = self.padding
        if isinstance(pad, str):
            raise NotImplementedError("string padding modes are not forwarded by dnn_to_bnn and not supported")
        return dict(stride=get_kernel_size(self.stride, 2), padding=get_kernel_size(pad, 2),
                    dilation=get_kernel_size(self.dilation, 2), groups=self.groups)

    def _forward(self, x, return_kl=True, residual=None):
        if self.dnn_to_bnn_flag:
            return_kl = False
        ctx = mc.current()
        collect = ctx is not None and ctx.collect_kl
        want_kl = return_kl or collect
        kind = self._prior_kind() if want_kl else "normal"
        x = _lib.dev_f32(x, "input")
        lead = None
        if self._kind == "linear":
            if x.shape[-1] != self.in_features:
                raise RuntimeError(f"{type(self).__name__}: expected last dim {self.in_features}, got {tuple(x.shape)}")
            if x.dim() != 2:
                if ctx is not None:
                    raise RuntimeError("MC batching needs [N, features] inputs for Linear layers")
                lead = tuple(x.shape[:-1])
                x = x.reshape(-1, self.in_features)
            conv = None
        else:
            one_d = getattr(self, "_one_d", False)
            if one_d:
                if x.dim() != 3 or x.shape[1] != self.in_channels:
                    raise RuntimeError(f"{type(self).__name__}: expected [N, {self.in_channels}, L], got {tuple(x.shape)}")
                x = x.unsqueeze(2)
            elif x.dim() != 4 or x.shape[1] != self.in_channels:
                raise RuntimeError(f"{type(self).__name__}: expected [N, {self.in_channels}, H, W], got {tuple(x.shape)}")
            conv = self._conv_desc()
        if x.shape[0] == 0:
            raise RuntimeError("empty batch")
        if ctx is None:
            S, shared = 1, True
        else:
            S = ctx.S
            if x.shape[0] == ctx.batch:
                shared = True
            elif x.shape[0] == S * ctx.batch:
                shared = False
            else:
                raise RuntimeError(f"inside mc_samples(S={S}, batch={ctx.batch}) a Bayesian layer got batch {x.shape[0]}")
        B = x.shape[0] // (1 if shared else S)
        one_d = getattr(self, "_one_d", False)
        mu_t, rho_t = self._w("mu"), self._w("rho")
        if one_d:                      # [Co, Ci/g, k] -> [Co, Ci/g, 1, k] (views: same storage, autograd flows through)
            mu_t, rho_t = mu_t.unsqueeze(2), rho_t.unsqueeze(2)

        sample0 = 0 if ctx is None else ctx.sample0
        call_base = None if ctx is None else ctx.call_base
        call, seed = rng.next_call(), rng.seed()
        if self.inject_draw is not None:
            inj = self.inject_draw.pop(0) if isinstance(self.inject_draw, list) else self.inject_draw   # a list feeds successive calls
            draw = {k: v for k, v in inj.items() if v is not None}
            if draw["eps_w"].shape[0] != S:
                raise RuntimeError(f"inject_draw holds {draw['eps_w'].shape[0]} samples, this call computes {S}")
        else:
            draw = self._draw_torch(x, S, B, conv) if rng.get_mode() == "torch" else {}
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or self._w("mu").requires_grad or self._w("rho").requires_grad
                                                  or (self.mu_bias is not None and self.mu_bias.requires_grad))
        if needs_grad:
            # training path: same fused forward kernel, gradients through the autograd bridge (autograd.py)
            if self.post_scale is not None or residual is not None or self.post_relu or self.post_pool:
                raise RuntimeError("the folded output stage (fuse.py) is inference-only: unfold or run under torch.no_grad()")
            from ..autograd import FusedForward, KLNormal
            opts = dict(flip=self._flip, conv=conv, S=S, shared=shared, seed=seed, call=call, layer_id=self._layer_id, sample0=sample0,
                        eps_w=draw.get("eps_w"), eps_b=draw.get("eps_b"), sign_in=draw.get("sign_in"), sign_out=draw.get("sign_out"),
                        packed=self._packed(), call_base=call_base,      # call_base: a captured training step (mc.TrainGraph)
                        workspace_owner=("layer", self._ws_id))
            fused_kl = ctx is not None and getattr(ctx, "train_fused", False)
            if fused_kl:
                # a training step (mc.TrainGraph): the KL term comes out of the forward kernel's fused sweep and is differentiated in
                # wgrad's finishing pass; get_kl_loss() / kl_loss() pick the live tensor up instead of launching a KL kernel
                opts["kl"] = (self.prior_weight_mu, self.prior_weight_sigma, self.prior_bias_mu, self.prior_bias_sigma, self._prior_kind())
                opts["defer"], opts["layer"] = ctx.deferred, self
                opts["kl_stub"] = not want_kl       # nobody reads this layer's KL by itself: one launch per model computes the value
                out, kl = FusedForward.apply(x, mu_t, rho_t, self.mu_bias, self.rho_bias, opts)
                self._kl_live = (kl, self._param_versions(), opts["kl_stub"])
                ctx.live_layers.append(self)     # (TrainGraph drops whatever the loss did not pick up: the tensor holds this forward's graph)
            else:
                out = FusedForward.apply(x, mu_t, rho_t, self.mu_bias, self.rho_bias, opts)
                kl = None
                if want_kl:
                    flat = [t for sg in self._kl_segments() for t in sg]
                    kl = KLNormal.apply((("layer", self._ws_id), kind), *flat)
        else:
            priors = (self.prior_weight_mu, self.prior_weight_sigma, self.prior_bias_mu, self.prior_bias_sigma) if want_kl else None
            out, kl = F.fused_forward(x, mu_t, rho_t, self.mu_bias, self.rho_bias, flip=self._flip, conv=conv,
                                      S=S, shared_x=shared, priors=priors, eps_w=draw.get("eps_w"), eps_b=draw.get("eps_b"),
                                      sign_in=draw.get("sign_in"), sign_out=draw.get("sign_out"), seed=seed, call=call,
                                      layer_id=self._layer_id, sample0=sample0, call_base=call_base, want_kl=want_kl,
                                      workspace_owner=("layer", self._ws_id), post_scale=self.post_scale, post_shift=self.post_shift,
                                      residual=residual, relu=self.post_relu, pool=self.post_pool, packed=self._packed(), prior_type=kind)
        conv_shape = tuple(out.shape[1:])     # shape of one sample's contraction output (sign_out's shape): before any fused pooling
        if self.post_pool and conv is not None:
            conv_shape = (out.shape[1],) + F.conv_out_hw(x.shape[2], x.shape[3], mu_t.shape[2], mu_t.shape[3], *conv["stride"],
                                                         *conv["padding"], *conv["dilation"])
        self._last = dict(draw=draw or None, rng=(seed, call_base, call, self._layer_id, sample0), S=S,
                          kernel=_lib.lib().bt_last_kernel_name().decode(), launch=_lib.last_launch_info(),
                          shared_x=shared, residual=residual is not None, fused_kl=bool(want_kl and not needs_grad),
                          x_shape=(B,) + tuple(x.shape[1:]), out_shape=(B,) + conv_shape)
        if lead is not None:
            out = out.reshape(lead + (self.out_features,))
        if one_d:
            out = out.squeeze(2)
        if collect:
            ctx.kls.append(kl)
        return (out, kl) if return_kl else out

    # ------------------------------------------------------------------ draws
    def _draw_torch(self, x, S, B, conv):
        """Parity mode: draw with torch's generator on the input device in the reference's order."""
        dev = x.device
        eps_w_buf = getattr(self, "eps_" + self._wname)
        has_b = self.mu_bias is not None
        xs = (S, B) + tuple(x.shape[1:])
        if conv is None:
            os_ = (S, B, self.out_features)
        else:
            kh, kw = (1, eps_w_buf.shape[2]) if eps_w_buf.dim() == 3 else (eps_w_buf.shape[2], eps_w_buf.shape[3])
            os_ = (S, B, self.out_channels) + F.conv_out_hw(x.shape[2], x.shape[3], kh, kw, *conv["stride"], *conv["padding"], *conv["dilation"])

        def eps():
            if S == 1:
                ew = eps_w_buf.data.normal_()
                eb = self.eps_bias.data.normal_() if has_b else None
            else:
                ew = torch.empty((S,) + tuple(eps_w_buf.shape), device=dev).normal_()
                eb = torch.empty((S, eps_w_buf.shape[0]), device=dev).normal_() if has_b else None
            return ew, eb

        d = {}
        if not self._flip:
            d["eps_w"], d["eps_b"] = eps()
        elif self._kind == "conv":      # s_in, s_out, eps_kernel, eps_bias  (conv_flipout.py:385-402)
            d["sign_in"] = torch.empty(xs, device=dev).uniform_(-1, 1).sign_()
            d["sign_out"] = torch.empty(os_, device=dev).uniform_(-1, 1).sign_()
            d["eps_w"], d["eps_b"] = eps()
        else:                            # eps_weight, eps_bias, s_in, s_out  (linear_flipout.py:149-170)
            d["eps_w"], d["eps_b"] = eps()
            d["sign_in"] = torch.empty(xs, device=dev).uniform_(-1, 1).sign_()
            d["sign_out"] = torch.empty(os_, device=dev).uniform_(-1, 1).sign_()
        return d

    def materialize_last_draw(self):
        """The draw the last forward used, as tensors: eps_w [S, *w], eps_b [S, Co], and for Flipout
        sign_in [S, B, ...], sign_out [S, B, ...].  In 'torch' mode these are the tensors that were
        read; in 'philox' mode they are regenerated from the same counters (bt_rng_*_fill)."""
        if self._last is None:
            raise RuntimeError("no forward has run yet")
        st = self._last
        S = st["S"]
        wshape = tuple(self._w("mu").shape)
        if st["draw"] is not None:
            d = st["draw"]
            res = dict(eps_w=d["eps_w"].reshape((S,) + wshape))
            if d.get("eps_b") is not None:
                res["eps_b"] = d["eps_b"].reshape(S, -1)
            for k in ("sign_in", "sign_out"):
                if d.get(k) is not None:
                    res[k] = d[k]
            return res
        seed, call_base, call, lid, sample0 = st["rng"]
        if call_base is not None:
            raise RuntimeError("draws made under a graph call_base cannot be replayed after the word advanced")
        dev = self._w("mu").device
        res = dict(eps_w=F.rng_fill_normal(seed, call, lid, sample0, 0, S, wshape, dev))
        if self.mu_bias is not None:
            res["eps_b"] = F.rng_fill_normal(seed, call, lid, sample0, 1, S, (wshape[0],), dev)
        if self._flip:
            res["sign_in"] = F.rng_fill_sign(seed, call, lid, sample0, 2, S, st["x_shape"], dev)
            res["sign_out"] = F.rng_fill_sign(seed, call, lid, sample0, 3, S, st["out_shape"], dev)
        return res
