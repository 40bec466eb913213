"""Autograd bridge for the fused forward.

The forward is the fused HIP kernel; the backward is the pair of hand-written HIP kernels of csrc/bt_bwd.hip
(bt_conv2d_bwd: data gradient and weight gradient as implicit GEMMs on fp32 MFMA).  Nothing weight-sized is saved for
backward and nothing weight-sized is materialised in it: the draws are a pure function of the RNG coordinates, so the
kernels REGENERATE eps (and the Flipout signs) from the coordinates the forward used, inside their operand staging, and
the weight-gradient kernel applies the chain rule to (mu, rho) in its output stage:

    dL/dmu  = sum_s dL/dW_s                       dL/drho  = sigmoid(rho) * sum_s dL/dW_s * eps_s
    (Flipout: mean path feeds mu, perturbation path feeds rho through Delta = softplus(rho) * eps)

The KL term's gradient is the element-wise HIP kernel bt_kl_normal_bwd.  Bias gradients are row sums of the upstream
gradient (a [S, Co] reduction) times the regenerated bias draw -- bias-sized, done with torch.sum.
``BACKWARD_IMPL = "aten"`` selects round 1's checker path (draws materialised with bt_rng_*_fill, ATen convolution
backward per sample): tests compare the two; the product default is "hip".
Reference arithmetic differentiated: layers/variational_layers/linear_variational.py:163-181,
conv_variational.py:366-385, flipout_layers/linear_flipout.py:149-174, conv_flipout.py:376-417 and the normal-prior
KL of base_variational_layer.py:68-72.
"""
import torch
import torch.nn.functional as TF

from . import _lib
from . import functional as F


def _contract(x, w, conv):
    if conv is None:
        return TF.linear(x, w)
    return TF.conv2d(x, w, None, conv["stride"], conv["padding"], conv["dilation"], conv["groups"])


def _grads(x, w, g, conv, need_x=True):
    """-> (dL/dx or None, dL/dw) of y = contract(x, w) for upstream g."""
    if conv is None:
        return (g @ w if need_x else None), g.t() @ x
    gx = torch.nn.grad.conv2d_input(x.shape, w, g, conv["stride"], conv["padding"], conv["dilation"], conv["groups"]) if need_x else None
    gw = torch.nn.grad.conv2d_weight(x, w.shape, g, conv["stride"], conv["padding"], conv["dilation"], conv["groups"])
    return gx, gw


BACKWARD_IMPL = "hip"     # "aten": the materialising checker path below (tests only)


def _kl_grads_aten(mu, rho, pm, ps, g, kind):
    """(d kl / d mu, d kl / d rho) * g of kl_div's mean over one tensor, in torch ops (checker path)."""
    n = mu.numel()
    sq = TF.softplus(rho)
    if kind == "laplace":    # d/dmu E|w| = erf(mu / (sigma sqrt 2)); d/dsigma = sqrt(2/pi) exp(-mu^2 / (2 sigma^2)) - 1/sigma
        return (torch.erf(mu / (sq * 1.4142135623730951)) * (g / n),
                (0.7978845608028654 * torch.exp(-mu * mu / (2 * sq * sq)) - 1.0 / sq) * torch.sigmoid(rho) * (g / n))
    return (mu - pm) / (ps * ps) * (g / n), (sq / (ps * ps) - 1.0 / sq) * torch.sigmoid(rho) * (g / n)


class FusedForward(torch.autograd.Function):
    """out[S*B, ...] = fused stochastic forward; differentiable in x, mu_w, rho_w, mu_b, rho_b."""

    @staticmethod
    def forward(ctx, x, mu_w, rho_w, mu_b, rho_b, opts):
        """opts["kl"] = (prior_mu_w, prior_sigma_w, prior_mu_b, prior_sigma_b, kind): the layer's KL term rides along -- computed by
        the forward kernel's fused sweep, returned as a second differentiable output, and differentiated inside wgrad's finishing
        pass (bt_conv2d_bwd_kl): a training step then has no KL launches and no KL-gradient tensors for autograd to add.
        opts["kl_stub"]: the second output is a placeholder -- at ONE sample a layer's launch has a handful of workgroups and the fused
        sweep of its parameters would sit on their critical path (ResNet18 step: 3.6 -> 4.6 ms), so the value is computed by one
        bt_kl_normal launch over the whole model (KLValue below) whose backward hands every placeholder the upstream gradient.
        opts["defer"] (a list; mc.TrainGraph): the weight gradients are computed on a side stream beside the dgrad chain and handed
        over through the list -- (layer, dmu, drho) -- instead of through autograd; ``mc.finish_deferred`` joins and assigns them."""
        o = dict(opts)
        klo = o.get("kl")
        stub = klo is not None and bool(o.get("kl_stub"))
        out, kl = F.fused_forward(x, mu_w, rho_w, mu_b, rho_b, flip=o["flip"], conv=o["conv"], S=o["S"], shared_x=o["shared"],
                                  seed=o["seed"], call=o["call"], layer_id=o["layer_id"], sample0=o["sample0"],
                                  eps_w=o.get("eps_w"), eps_b=o.get("eps_b"), sign_in=o.get("sign_in"), sign_out=o.get("sign_out"),
                                  packed=o.get("packed"), workspace_owner=o.get("workspace_owner", ("layer", o["layer_id"])), call_base=o.get("call_base"),
                                  priors=None if (klo is None or stub) else tuple(klo[:4]), want_kl=klo is not None and not stub,
                                  prior_type="normal" if klo is None else klo[4])
        if stub:      # the VALUE comes from one launch over the whole model (KLValue, get_kl_loss); this output only routes its gradient here
            kl = out.new_empty(())
        ctx.o = o
        ctx.save_for_backward(x, mu_w, rho_w, mu_b, rho_b)
        ctx.out_shape = tuple(out.shape)
        ctx.set_materialize_grads(False)
        return out if klo is None else (out, kl)

    @staticmethod
    def backward(ctx, g, g_kl=None):
        x, mu_w, rho_w, mu_b, rho_b = ctx.saved_tensors
        o = ctx.o
        S, shared, conv, flip = o["S"], o["shared"], o["conv"], o["flip"]
        dev = x.device
        g = torch.zeros(ctx.out_shape, dtype=torch.float32, device=dev) if g is None else g.contiguous()
        B = x.shape[0] // (1 if shared else S)
        coords = (o["seed"], o["call"], o["layer_id"], o["sample0"])
        klo = o.get("kl") if g_kl is not None else None
        lap = klo is not None and klo[4] == "laplace"
        if BACKWARD_IMPL == "hip":
            need_x = ctx.needs_input_grad[0]
            need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            packed = o.get("packed") or F.pack_params(mu_w.detach(), rho_w.detach())
            kw = dict(flip=flip, conv=conv, S=S, shared_x=shared, eps_w=o.get("eps_w"), sign_in=o.get("sign_in"), sign_out=o.get("sign_out"),
                      seed=o["seed"], call=o["call"], layer_id=o["layer_id"], sample0=o["sample0"], call_base=o.get("call_base"))
            kl_arg = None if klo is None else (g_kl, klo[0], klo[1], klo[4])
            defer = o.get("defer")
            if defer is not None and need_w:
                # wgrad (+ its finishing pass, + the KL term) on the side stream, forked HERE -- g is ready, the dgrad below is not waited for
                from .mc import _side_stream
                cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _, gmu, grho = F.fused_backward(x, g, mu_w.detach(), rho_w, packed, need_x=False, need_w=True, kl=kl_arg, **kw)
                for t in (x, g, g_kl, packed[0], packed[1]):
                    if t is not None:
                        t.record_stream(side)
                defer.append((o["layer"], gmu, grho, dev))
                gmu = grho = None
                gx = F.fused_backward(x, g, mu_w.detach(), rho_w, packed, need_x=True, need_w=False, **kw)[0] if need_x else None
            else:
                gx, gmu, grho = F.fused_backward(x, g, mu_w.detach(), rho_w, packed, need_x=need_x, need_w=need_w, kl=kl_arg, **kw)
            gmu_b = grho_b = None
            if mu_b is not None:
                Co = mu_w.shape[0]
                gs = g.reshape((S, B, Co, -1))
                if flip:       # out = mean + (pert + Delta_b) o s_out: mu_b sees g, the bias perturbation sees g o s_out
                    s_out = o.get("sign_out")
                    if s_out is None:
                        s_out = F.rng_fill_sign(*coords, 3, S, (B,) + tuple(ctx.out_shape[1:]), dev, call_base=o.get("call_base"))
                    gp = (g.reshape(s_out.shape) * s_out).reshape((S, B, Co, -1)).sum((1, 3))
                else:
                    gp = gs.sum((1, 3))
                eps_b = o.get("eps_b")
                if eps_b is None:
                    eps_b = F.rng_fill_normal(*coords, 1, S, (Co,), dev, call_base=o.get("call_base"))
                gmu_b = gs.sum((0, 1, 3))
                grho_b = (gp * eps_b.reshape(S, Co)).sum(0) * torch.sigmoid(rho_b)
                if klo is not None:      # the bias term of the layer's KL (bias-sized: its own small launch)
                    km, kr = F.kl_backward(mu_b, rho_b, klo[2], klo[3], g_kl, laplace=lap)
                    gmu_b, grho_b = gmu_b + km, grho_b + kr
            return gx, gmu, grho, gmu_b, grho_b, None
        if o.get("call_base") is not None:
            raise RuntimeError("the ATen checker path does not replay graph-captured draws")
        # the draws of the forward, regenerated (or the injected ones)
        eps_w = o.get("eps_w")
        if eps_w is None:
            eps_w = F.rng_fill_normal(*coords, 0, S, mu_w.shape, dev)
        eps_w = eps_w.reshape((S,) + tuple(mu_w.shape))
        has_b = mu_b is not None
        eps_b = o.get("eps_b")
        if has_b and eps_b is None:
            eps_b = F.rng_fill_normal(*coords, 1, S, (mu_w.shape[0],), dev)
        x_s_shape = (B,) + tuple(x.shape[1:])
        o_s_shape = (B,) + tuple(ctx.out_shape[1:])
        s_in = s_out = None
        if flip:
            s_in, s_out = o.get("sign_in"), o.get("sign_out")
            if s_in is None:
                s_in = F.rng_fill_sign(*coords, 2, S, x_s_shape, dev)
                s_out = F.rng_fill_sign(*coords, 3, S, o_s_shape, dev)
            s_in, s_out = s_in.reshape((S,) + x_s_shape), s_out.reshape((S,) + o_s_shape)
        sigma = TF.softplus(rho_w)
        dsig = torch.sigmoid(rho_w)
        need_x = ctx.needs_input_grad[0]
        gx = torch.zeros_like(x) if need_x else None
        gmu = torch.zeros_like(mu_w)
        grho = torch.zeros_like(rho_w)
        gmu_b = torch.zeros_like(mu_b) if has_b else None
        grho_b = torch.zeros_like(rho_b) if has_b else None
        red = [0] + list(range(2, g.dim()))          # every axis but the channel axis
        for s in range(S):
            xs = x if shared else x[s * B:(s + 1) * B]
            gs = g[s * B:(s + 1) * B]
            if not flip:
                w = mu_w + sigma * eps_w[s]
                gxs, gw = _grads(xs, w, gs, conv, need_x)
                gmu += gw
                grho += gw * eps_w[s] * dsig
                if has_b:
                    gb = gs.sum(red)
                    gmu_b += gb
                    grho_b += gb * eps_b[s] * torch.sigmoid(rho_b)
            else:
                gp = gs * s_out[s]
                delta = sigma * eps_w[s]
                gx1, gw_mu = _grads(xs, mu_w, gs, conv, need_x)
                gx2, gw_d = _grads(xs * s_in[s], delta, gp, conv, need_x)
                gmu += gw_mu
                grho += gw_d * eps_w[s] * dsig
                gxs = (gx1 + gx2 * s_in[s]) if need_x else None
                if has_b:
                    gmu_b += gs.sum(red)
                    grho_b += gp.sum(red) * eps_b[s] * torch.sigmoid(rho_b)
            if need_x:
                if shared:
                    gx += gxs
                else:
                    gx[s * B:(s + 1) * B] = gxs
        if klo is not None:
            km, kr = _kl_grads_aten(mu_w, rho_w, klo[0], klo[1], g_kl, klo[4])
            gmu, grho = gmu + km, grho + kr
            if has_b:
                km, kr = _kl_grads_aten(mu_b, rho_b, klo[2], klo[3], g_kl, klo[4])
                gmu_b, grho_b = gmu_b + km, grho_b + kr
        return gx, gmu, grho, gmu_b, grho_b, None


class KLNormal(torch.autograd.Function):
    """kl = sum over the given (mu, rho, prior_mu, prior_sigma) groups of mean_i(...), on the HIP KL kernel; grads to mu, rho."""

    @staticmethod
    def forward(ctx, owner_kind, *tensors):
        owner, kind = owner_kind[0], owner_kind[1]
        segs = [tuple(tensors[i:i + 4]) for i in range(0, len(tensors), 4)]
        lids = list(owner_kind[2]) if len(owner_kind) > 2 else [0] * len(segs)     # (a whole model's segments in ONE launch: get_kl_loss)
        ctx.save_for_backward(*tensors)
        ctx.kind = kind
        return _lib.kl_normal([tuple(t.detach() for t in sg) for sg in segs], layer_ids=lids, owner=owner, laplace=kind == "laplace")

    @staticmethod
    def backward(ctx, g):
        t = ctx.saved_tensors
        grads = [None]
        if BACKWARD_IMPL == "hip":     # every tensor of the call in one launch
            for gmu, grho in F.kl_backward_segs([tuple(t[i:i + 4]) for i in range(0, len(t), 4)], g, laplace=ctx.kind == "laplace"):
                grads += [gmu, grho, None, None]
            return tuple(grads)
        for i in range(0, len(t), 4):
            mu, rho, pm, ps = t[i:i + 4]
            gmu, grho = _kl_grads_aten(mu, rho, pm, ps, g, ctx.kind)
            grads += [gmu, grho, None, None]
        return tuple(grads)


class KLValue(torch.autograd.Function):
    """get_kl_loss of a training step whose layers differentiate their own KL terms (FusedForward, opts["kl"] + opts["kl_stub"]):
    the value from ONE bt_kl_normal launch over all segments; backward gives the upstream gradient to every layer's placeholder --
    the layers' weight-gradient passes do the rest (bt_conv2d_bwd_kl). No KL-gradient tensors, no second sweep."""

    @staticmethod
    def forward(ctx, info, *stubs):
        segs, lids, owner, laplace = info
        ctx.n = len(stubs)
        return _lib.kl_normal([tuple(t.detach() for t in sg) for sg in segs], layer_ids=list(lids), owner=owner, laplace=laplace)

    @staticmethod
    def backward(ctx, g):
        return (None,) + (g,) * ctx.n
