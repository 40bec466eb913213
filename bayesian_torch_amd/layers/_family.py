"""Conv3d and ConvTranspose{1,2,3}d (both flavours) on the fused kernels -- SURVEY.md section 8(f) rank 4.

Reference behaviour reproduced (paths under /root/reference/bayesian_torch/layers/):
  Conv3dReparameterization / Flipout            variational_layers/conv_variational.py:650-820, flipout_layers/conv_flipout.py:443-638
  ConvTranspose{1,2,3}dReparameterization       conv_variational.py:822-990, 992-1165, 1167-1340
  ConvTranspose{1,2,3}dFlipout                  conv_flipout.py:640-832, 834-1031, 1033-1230
(parameter names and shapes -- ConvTranspose kernels are [Ci][Co/groups][k...] --, constructor signatures, forward(input,
return_kl=True) -> (out, kl) | out, kl_loss()).

Every one of them is ONE launch of the fused Conv2d kernel (sampling, contraction, Flipout signs on chip) behind index
re-arrangements that are exact (no arithmetic):
  * transposed convolution = stride-1 convolution of the zero-upsampled input ((L-1)*s+1 samples, padded by d*(k-1)-p, plus
    output_padding on the far side) with the kernel transposed in its channel axes and flipped in space; sampling is
    element-wise, so transposing / flipping (mu, rho, eps) commutes with it;
  * Conv3d = Conv2d over B*Do images whose channels are (ci, kd): the depth window is unfolded into the channel axis, and the
    [Co][Ci/g][kd][kh][kw] kernel IS a [Co][(Ci/g)*kd][kh][kw] kernel in memory;
  * Conv1d-like = a 1 x k kernel over 1 x L images.
The re-arrangements of x are torch gathers (differentiable: training goes through the same autograd bridge as Conv2d) and cost
one extra pass over the activations; doing them inside the kernel's x staging (a dgrad-style gather) is the follow-up. KL is
taken by the standalone KL kernel on the parameters in their own layout.
"""
import torch
import torch.nn.functional as TF

from .. import _lib, mc, rng
from .. import functional as F
from ._fused import FusedBayesLayer
from .base_variational_layer import get_kernel_size


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class FamilyConvLayer(FusedBayesLayer):
    _kind, _wname = "conv", "kernel"
    _nd, _transposed = 2, False

    def _setup(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding, prior_mean, prior_variance,
               posterior_mu_init, posterior_rho_init, bias, tuple_inits):
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError('invalid in_channels size')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        if self._transposed:
            self.output_padding = output_padding
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init = (posterior_mu_init,) if tuple_inits else posterior_mu_init      # trailing commas of the Reparameterization classes
        self.posterior_rho_init = (posterior_rho_init,) if tuple_inits else posterior_rho_init
        self.bias = bias
        ks = get_kernel_size(kernel_size, self._nd)
        wshape = (in_channels, out_channels // groups) + ks if self._transposed else (out_channels, in_channels // groups) + ks
        self._build(wshape, bias, n_out=out_channels)

    # ------------------------------------------------------------------ exact re-arrangements
    def _w_eq(self, t, lead=0):
        """[lead...][kernel in this class's layout] -> the equivalent Conv2d kernel [lead...][Co][Cig'][kh][kw] (views / one copy)."""
        nd, g = self._nd, self.groups
        if self._transposed:                       # [Ci][Co/g][k...] -> [Co][Ci/g][k...], flipped in space
            L = t.shape[:lead]
            Ci, Cog = t.shape[lead], t.shape[lead + 1]
            ks = t.shape[lead + 2:]
            t = t.reshape(L + (g, Ci // g, Cog) + ks).transpose(lead + 1, lead + 2).reshape(L + (g * Cog, Ci // g) + ks)
            t = t.flip(tuple(range(lead + 2, lead + 2 + nd)))
        if nd == 1:
            t = t.unsqueeze(lead + 2)
        elif nd == 3:                              # (ci, kd) -> one channel axis: a view of the same memory when t is contiguous
            t = t.reshape(t.shape[:lead + 1] + (t.shape[lead + 1] * t.shape[lead + 2],) + t.shape[lead + 3:])
        return t.contiguous()

    def _geom(self):
        nd = self._nd
        s, p, d = _tup(self.stride, nd), _tup(self.padding, nd), _tup(self.dilation, nd)
        if any(isinstance(v, str) for v in p):
            raise NotImplementedError("string padding modes are not supported")
        op = _tup(getattr(self, "output_padding", 0), nd)
        ks = tuple(self._w("mu").shape[2:])
        return s, p, d, op, ks

    def _x_eq(self, x):
        """[N][C][spatial...] -> ([N'][C'][H][W] for the Conv2d launch, conv dict, function mapping the launch's output back)."""
        nd = self._nd
        s, p, d, op, ks = self._geom()
        if self._transposed:                       # zero-upsample, then pad by d*(k-1)-p (+ output_padding on the far side)
            up = tuple((n - 1) * si + 1 for n, si in zip(x.shape[2:], s))
            xu = x.new_zeros(x.shape[:2] + up)
            xu[(slice(None), slice(None)) + tuple(slice(None, None, si) for si in s)] = x
            pads = []
            for i in reversed(range(nd)):           # F.pad lists the last axis first; a negative amount crops
                lo = d[i] * (ks[i] - 1) - p[i]
                pads += [lo, lo + op[i]]
            x = TF.pad(xu, pads)
            s, p = (1,) * nd, (0,) * nd
        n0 = x.shape[0]
        if nd == 1:
            x = x.unsqueeze(2)
            conv = dict(stride=(1, s[0]), padding=(0, p[0]), dilation=(1, d[0]), groups=self.groups)
            back = lambda o: o.squeeze(2)
        elif nd == 2:
            conv = dict(stride=s, padding=p, dilation=d, groups=self.groups)
            back = lambda o: o
        else:                                       # unfold the depth window into the channel axis
            x = TF.pad(x, (0, 0, 0, 0, p[0], p[0]))
            win = (ks[0] - 1) * d[0] + 1
            xw = x.unfold(2, win, s[0])[..., ::d[0]]                        # [N][C][Do][H][W][kd]
            Do = xw.shape[2]
            x = xw.permute(0, 2, 1, 5, 3, 4).reshape(n0 * Do, x.shape[1] * ks[0], x.shape[3], x.shape[4])
            conv = dict(stride=s[1:], padding=p[1:], dilation=d[1:], groups=self.groups)
            # (the leading dimension comes from the launch's output: under mc_samples it holds S * n0 * Do rows even when x was shared)
            back = lambda o: o.reshape(-1, Do, o.shape[1], o.shape[2], o.shape[3]).permute(0, 2, 1, 3, 4)
        return x.contiguous(), conv, back

    def _sign_out_eq(self, t):
        """[S][B][Co][spatial...] (the reference's layout) -> the Conv2d launch's [S][B'][Co][Ho][Wo]."""
        if self._nd == 1:
            return t.unsqueeze(3).contiguous()
        if self._nd == 3:
            S, B, Co, Do = t.shape[:4]
            return t.permute(0, 1, 3, 2, 4, 5).reshape(S, B * Do, Co, t.shape[4], t.shape[5]).contiguous()
        return t.contiguous()

    def _pack_source(self):
        return self._w_eq(self._w("mu").detach()), self._w_eq(self._w("rho").detach())

    def _w_nat(self, t, lead=0):
        """Inverse of _w_eq: [lead...][Co][Cig'][kh][kw] of the Conv2d launch -> this class's own kernel layout."""
        nd, g = self._nd, self.groups
        ks = tuple(self._w("mu").shape[2:])
        L = tuple(t.shape[:lead])
        if nd == 1:
            t = t.squeeze(lead + 2)
        elif nd == 3:
            t = t.reshape(L + (t.shape[lead], t.shape[lead + 1] // ks[0]) + ks)
        if self._transposed:
            t = t.flip(tuple(range(lead + 2, lead + 2 + nd)))
            Co, Cig = t.shape[lead], t.shape[lead + 1]
            t = t.reshape(L + (g, Co // g, Cig) + ks).transpose(lead + 1, lead + 2).reshape(L + (g * Cig, Co // g) + ks)
        return t.contiguous()

    def materialize_last_draw(self):
        """The last forward's draw in the REFERENCE's layouts where one exists: eps_w [S, *kernel], eps_b [S, Co]. The on-chip
        Flipout signs are defined over the Conv2d launch's operands (``sign_in_eq`` / ``sign_out_eq``: the re-arranged x and the
        launch's output): a transposed convolution's zero-upsampled x carries one sign per real element like the reference, Conv3d's
        depth-unfolded x one sign per (element, depth window) -- see DESIGN.md 4.6."""
        if self._last is None:
            raise RuntimeError("no forward has run yet")
        st = self._last
        if st["draw"] is not None:
            d = dict(st["draw"])
            d["eps_w"] = self._w_nat(d["eps_w"], lead=1)
            return d
        seed, call_base, call, lid, sample0 = st["rng"]
        if call_base is not None:
            raise RuntimeError("draws made under a graph call_base cannot be replayed after the word advanced")
        dev, S = self._w("mu").device, st["S"]
        res = dict(eps_w=self._w_nat(F.rng_fill_normal(seed, call, lid, sample0, 0, S, st["w_eq_shape"], dev), lead=1))
        if self.mu_bias is not None:
            res["eps_b"] = F.rng_fill_normal(seed, call, lid, sample0, 1, S, (self.out_channels,), dev)
        if self._flip:
            res["sign_in_eq"] = F.rng_fill_sign(seed, call, lid, sample0, 2, S, st["x_shape"], dev)
            res["sign_out_eq"] = F.rng_fill_sign(seed, call, lid, sample0, 3, S, st["out_shape"], dev)
        return res

    # ------------------------------------------------------------------ forward
    def forward(self, input, return_kl=True):
        if self.dnn_to_bnn_flag:
            return_kl = False
        ctx = mc.current()
        collect = ctx is not None and ctx.collect_kl
        want_kl = return_kl or collect
        x = _lib.dev_f32(input, "input")
        if x.dim() != self._nd + 2 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"{type(self).__name__}: expected [N, {self.in_channels}, {self._nd} spatial dims], got {tuple(x.shape)}")
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
        sample0 = 0 if ctx is None else ctx.sample0
        call_base = None if ctx is None else ctx.call_base
        call, seed = rng.next_call(), rng.seed()
        xe, conv, back = self._x_eq(x)
        mu_e, rho_e = self._w_eq(self._w("mu")), self._w_eq(self._w("rho"))
        draw = {}
        if self.inject_draw is not None:           # draws in the reference's layouts (test hook), re-arranged like the operands
            inj = self.inject_draw.pop(0) if isinstance(self.inject_draw, list) else self.inject_draw
            draw["eps_w"] = self._w_eq(inj["eps_w"], lead=1)
            if inj.get("eps_b") is not None:
                draw["eps_b"] = inj["eps_b"]
            if self._flip:
                si = inj["sign_in"]
                draw["sign_in"] = self._x_eq(si.reshape((-1,) + tuple(si.shape[2:])))[0].reshape((si.shape[0], -1) + tuple(xe.shape[1:]))
                draw["sign_out"] = self._sign_out_eq(inj["sign_out"])
        elif rng.get_mode() == "torch":
            raise NotImplementedError("rng mode 'torch' covers Linear / Conv1d / Conv2d; the rest of the family draws on chip or takes inject_draw")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from ..autograd import FusedForward
            if call_base is not None:
                raise RuntimeError("graph-replayed draws (call_base) are not supported on the training path")
            opts = dict(flip=self._flip, conv=conv, S=S, shared=shared, seed=seed, call=call, layer_id=self._layer_id, sample0=sample0,
                        eps_w=draw.get("eps_w"), eps_b=draw.get("eps_b"), sign_in=draw.get("sign_in"), sign_out=draw.get("sign_out"), packed=self._packed())
            out = FusedForward.apply(xe, mu_e, rho_e, self.mu_bias, self.rho_bias, opts)
        else:
            out, _ = F.fused_forward(xe, mu_e, rho_e, self.mu_bias, self.rho_bias, flip=self._flip, conv=conv, S=S, shared_x=shared,
                                     eps_w=draw.get("eps_w"), eps_b=draw.get("eps_b"), sign_in=draw.get("sign_in"), sign_out=draw.get("sign_out"),
                                     seed=seed, call=call, layer_id=self._layer_id, sample0=sample0, call_base=call_base, packed=self._packed())
        Be = xe.shape[0] // (1 if shared else S)
        self._last = dict(draw=draw or None, rng=(seed, call_base, call, self._layer_id, sample0), S=S, kernel=_lib.lib().bt_last_kernel_name().decode(),
                          w_eq_shape=tuple(mu_e.shape), x_shape=(Be,) + tuple(xe.shape[1:]), out_shape=(Be,) + tuple(out.shape[1:]))
        out = back(out).contiguous()
        kl = self.kl_loss() if want_kl else None
        if collect:
            ctx.kls.append(kl)
        return (out, kl) if return_kl else out


def _make(name, nd, transposed, flip, doc):
    tuple_inits = not flip

    if transposed:
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, output_padding=0,
                     prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
            FusedBayesLayer.__init__(self)
            self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, output_padding, prior_mean, prior_variance,
                        posterior_mu_init, posterior_rho_init, bias, tuple_inits)
    elif not flip:      # Conv3dReparameterization: the prior / posterior arguments are positional, without defaults (conv_variational.py:651-663)
        def __init__(self, in_channels, out_channels, kernel_size, prior_mean, prior_variance, posterior_mu_init, posterior_rho_init,
                     stride=1, padding=0, dilation=1, groups=1, bias=True):
            FusedBayesLayer.__init__(self)
            self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0, prior_mean, prior_variance,
                        posterior_mu_init, posterior_rho_init, bias, tuple_inits)
    else:
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
            FusedBayesLayer.__init__(self)
            self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, 0, prior_mean, prior_variance,
                        posterior_mu_init, posterior_rho_init, bias, tuple_inits)
    return type(name, (FamilyConvLayer,), dict(__init__=__init__, __doc__=doc, _nd=nd, _transposed=transposed, _flip=flip, __module__=__name__))


Conv3dReparameterization = _make("Conv3dReparameterization", 3, False, False, "Drop-in for reference conv_variational.py:650-820.")
ConvTranspose1dReparameterization = _make("ConvTranspose1dReparameterization", 1, True, False, "Drop-in for reference conv_variational.py:822-990.")
ConvTranspose2dReparameterization = _make("ConvTranspose2dReparameterization", 2, True, False, "Drop-in for reference conv_variational.py:992-1165.")
ConvTranspose3dReparameterization = _make("ConvTranspose3dReparameterization", 3, True, False, "Drop-in for reference conv_variational.py:1167-1340.")
Conv3dFlipout = _make("Conv3dFlipout", 3, False, True, "Drop-in for reference conv_flipout.py:443-638.")
ConvTranspose1dFlipout = _make("ConvTranspose1dFlipout", 1, True, True, "Drop-in for reference conv_flipout.py:640-832.")
ConvTranspose2dFlipout = _make("ConvTranspose2dFlipout", 2, True, True, "Drop-in for reference conv_flipout.py:834-1031.")
ConvTranspose3dFlipout = _make("ConvTranspose3dFlipout", 3, True, True, "Drop-in for reference conv_flipout.py:1033-1230.")

REPARAM = ["Conv3dReparameterization", "ConvTranspose1dReparameterization", "ConvTranspose2dReparameterization", "ConvTranspose3dReparameterization"]
FLIPOUT = ["Conv3dFlipout", "ConvTranspose1dFlipout", "ConvTranspose2dFlipout", "ConvTranspose3dFlipout"]
