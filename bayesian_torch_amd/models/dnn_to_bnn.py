"""``dnn_to_bnn`` / ``get_kl_loss`` -- drop-in for reference ``models/dnn_to_bnn.py:52-165``.

Same contract: in-place recursive swap of leaf modules whose class name contains "Conv" /
"Linear" by ``<ClassName><params['type']>`` looked up in ``bayesian_torch_amd.layers``
(AttributeError when that class does not exist), same dict keys (KeyError when one is missing), MOPED initialisation, and
``dnn_to_bnn_flag = True`` on every created layer.  ``get_kl_loss`` gathers every fused layer's
(mu, rho, prior) tensors into ONE bt_kl_normal launch instead of ~12 ATen passes per layer.
"""
import bayesian_torch_amd.layers as bayesian_layers
from bayesian_torch_amd import _lib, rng
from bayesian_torch_amd.layers._fused import FusedBayesLayer
from bayesian_torch_amd.layers.base_variational_layer import check_prior_type
from bayesian_torch_amd.utils.util import get_rho


def _moped(layer, d, delta):
    layer.invalidate_pack()
    w = layer._w("mu")
    w.data.copy_(d.weight.data)
    layer._w("rho").data.copy_(get_rho(d.weight.data, delta))
    if layer.mu_bias is not None:
        layer.mu_bias.data.copy_(d.bias.data)
        layer.rho_bias.data.copy_(get_rho(d.bias.data, delta))


def bnn_linear_layer(params, d):
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_features=d.in_features, out_features=d.out_features,
                         prior_mean=params["prior_mu"], prior_variance=params["prior_sigma"],
                         posterior_mu_init=params["posterior_mu_init"], posterior_rho_init=params["posterior_rho_init"],
                         bias=d.bias is not None)
    if params["moped_enable"]:
        _moped(bnn_layer, d, params["moped_delta"])
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer.to(d.weight.device)


def bnn_conv_layer(params, d):
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_channels=d.in_channels, out_channels=d.out_channels, kernel_size=d.kernel_size,
                         stride=d.stride, padding=d.padding, dilation=d.dilation, groups=d.groups,
                         prior_mean=params["prior_mu"], prior_variance=params["prior_sigma"],
                         posterior_mu_init=params["posterior_mu_init"], posterior_rho_init=params["posterior_rho_init"],
                         bias=d.bias is not None)
    if params["moped_enable"]:
        _moped(bnn_layer, d, params["moped_delta"])
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer.to(d.weight.device)


def bnn_lstm_layer(params, d):
    layer_fn = getattr(bayesian_layers, d.__class__.__name__ + params["type"])
    bnn_layer = layer_fn(in_features=d.input_size, out_features=d.hidden_size, prior_mean=params["prior_mu"],
                         prior_variance=params["prior_sigma"], posterior_mu_init=params["posterior_mu_init"],
                         posterior_rho_init=params["posterior_rho_init"], bias=d.bias is not None)
    if params["moped_enable"]:
        print("WARNING: MOPED method is not supported for LSTM layers!!!")
    bnn_layer.dnn_to_bnn_flag = True
    return bnn_layer.to(next(d.parameters()).device)


def dnn_to_bnn(m, bnn_prior_parameters):
    for name in list(m._modules):
        child = m._modules[name]
        cname = child.__class__.__name__
        if child._modules:
            dnn_to_bnn(child, bnn_prior_parameters)
        elif "Conv" in cname:
            setattr(m, name, bnn_conv_layer(bnn_prior_parameters, child))
        elif "Linear" in cname:
            setattr(m, name, bnn_linear_layer(bnn_prior_parameters, child))
        elif "LSTM" in cname:
            setattr(m, name, bnn_lstm_layer(bnn_prior_parameters, child))
    # RNG coordinate of every Bayesian layer = its position in the model (the outermost call of the recursion has the last
    # word): draws then depend on (seed, call, position, sample), not on how many layers the process built before.
    rng.assign_layer_ids(m)
    return


def get_kl_loss(m):
    """Sum of ``layer.kl_loss()`` over every module that has one; None without Bayesian layers."""
    import torch
    fused = [layer for layer in m.modules() if isinstance(layer, FusedBayesLayer)]
    if fused and torch.is_grad_enabled() and all(getattr(layer, "_kl_live", None) is not None for layer in fused):
        # a training step whose forwards carried their KL terms along (mc.TrainGraph): no KL launch, one stack + sum
        live = [layer._take_live_kl() for layer in fused]
        if all(k is not None for k in live):
            from ..autograd import KLValue
            vals = [k[0] for k in live if not k[2]]                 # values out of the forward kernels' fused sweeps
            kl = torch.stack(vals).sum() if vals else None
            by_kind = {}
            for layer, k in zip(fused, live):                      # placeholders: ONE launch per prior kind computes their value
                if k[2]:
                    by_kind.setdefault(layer._prior_kind() == "laplace", []).append((layer, k[0]))
            for lap, items in by_kind.items():
                segs, lids = [], []
                for i, (layer, _) in enumerate(items):
                    sg = layer._kl_segments()
                    segs += sg
                    lids += [i] * len(sg)
                v = KLValue.apply((segs, lids, ("model", id(m), lap), lap), *[st for _, st in items])
                kl = v if kl is None else kl + v
            for layer in m.modules():
                if not isinstance(layer, FusedBayesLayer) and hasattr(layer, "kl_loss"):
                    kl = kl + layer.kl_loss()
            return kl
    segs, lids, others, lap = [], [], [], []
    n = 0
    for layer in m.modules():
        if isinstance(layer, FusedBayesLayer):
            if check_prior_type(getattr(layer, "prior_type", "normal")) == "laplace":
                lap.append(layer)          # (rare) its own launch: one bt_kl_normal call takes one prior kind
                continue
            s = layer._kl_segments()
            segs += s
            lids += [n] * len(s)
            n += 1
        elif hasattr(layer, "kl_loss"):
            others.append(layer)
    others = lap + others
    kl = None
    if segs:
        import torch
        if torch.is_grad_enabled() and any(t.requires_grad for sg in segs for t in sg):   # training: differentiable, still ONE forward launch
            from ..autograd import KLNormal
            kl = KLNormal.apply((("model", id(m)), "normal", tuple(lids)), *[t for sg in segs for t in sg])
        else:
            kl = _lib.kl_normal(segs, layer_ids=lids, owner=("model", id(m)))
    for layer in others:     # foreign modules with a kl_loss of their own
        kl = layer.kl_loss() if kl is None else kl + layer.kl_loss()
    return kl
