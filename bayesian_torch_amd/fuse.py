"""Inference-time folding of the deterministic ops that follow a Bayesian conv into the conv kernel's
output stage (bt_epilogue): BatchNorm2d in eval mode is a per-channel affine map, so
``bn(conv(x))`` becomes ``conv(x) * scale + shift`` inside the kernel's epilogue -- one pass over the
activations instead of three.  (The reference folds BN only on its quantised path,
models/bnn_to_qbnn.py:174-196)  Opt-in; valid while the model stays in eval().
"""
import torch
import torch.nn as nn

from .layers._fused import FusedBayesLayer


def _affine_of(bn):
    if not isinstance(bn, nn.BatchNorm2d) or bn.running_mean is None:
        raise ValueError("only BatchNorm2d with running statistics can be folded")
    if bn.training:
        raise RuntimeError("fold_batchnorm needs model.eval(): batch statistics cannot be folded")
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        g = bn.weight if bn.weight is not None else torch.ones_like(inv)
        b = bn.bias if bn.bias is not None else torch.zeros_like(inv)
        scale = (g * inv).float().contiguous()
        shift = (b - bn.running_mean * g * inv).float().contiguous()
    return scale, shift


def fold_pair(conv, bn, relu=False):
    """Fold ``bn`` (and optionally a following ReLU) into ``conv``'s output stage."""
    if not isinstance(conv, FusedBayesLayer) or conv._kind != "conv":
        raise TypeError("fold_pair needs a Bayesian Conv2d layer of this package")
    if bn.num_features != conv.out_channels:
        raise ValueError("BatchNorm features do not match the conv's out_channels")
    if conv.post_scale is not None:
        raise RuntimeError("this layer already has a folded output stage")
    scale, shift = _affine_of(bn)
    conv.post_scale, conv.post_shift = scale.to(conv.mu_kernel.device), shift.to(conv.mu_kernel.device)
    conv.post_relu = bool(relu)


def fold_maxpool(conv, pool):
    """Fold a following ``nn.MaxPool2d(3, 2, 1)`` into ``conv``'s output stage (after its folded BN / ReLU, if any)."""
    if not isinstance(conv, FusedBayesLayer) or conv._kind != "conv":
        raise TypeError("fold_maxpool needs a Bayesian Conv2d layer of this package")
    as2 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if not isinstance(pool, nn.MaxPool2d) or as2(pool.kernel_size) != (3, 3) or as2(pool.stride) != (2, 2) or as2(pool.padding) != (1, 1) \
            or as2(pool.dilation) != (1, 1) or pool.ceil_mode or pool.return_indices:
        raise ValueError("only MaxPool2d(kernel_size=3, stride=2, padding=1) folds into the conv")
    conv.post_pool = True


def fold_batchnorm(model):
    """Fold every BatchNorm2d that directly follows a Bayesian Conv2d in its parent's registration order
    (conv1/bn1, conv2/bn2, Sequential(conv, bn) ...) and replace it by nn.Identity.  Returns the number folded.
    The pairing is by registration order, the common convention -- check it matches your forward()."""
    n = 0
    for parent in model.modules():
        names = list(parent._modules)
        for a, b in zip(names, names[1:]):
            conv, bn = parent._modules[a], parent._modules[b]
            if isinstance(conv, FusedBayesLayer) and conv._kind == "conv" and isinstance(bn, nn.BatchNorm2d) \
                    and conv.post_scale is None and bn.num_features == conv.out_channels:
                fold_pair(conv, bn)
                parent._modules[b] = nn.Identity()
                n += 1
    return n


def fold_relu(model):
    """Fold every ``nn.ReLU`` that directly follows a Bayesian layer of this package in its parent's registration order
    (Sequential(Linear, ReLU, Linear): an MLP) into that layer's output stage and replace it by nn.Identity. Returns the number
    folded. The same max(v, 0) on the same values, one launch and one pass over the activations fewer."""
    n = 0
    for parent in model.modules():
        names = list(parent._modules)
        for a, b in zip(names, names[1:]):
            layer, act = parent._modules[a], parent._modules[b]
            if isinstance(layer, FusedBayesLayer) and type(act) is nn.ReLU and not layer.post_relu and not layer.post_pool:
                layer.post_relu = True
                parent._modules[b] = nn.Identity()
                n += 1
    return n
