#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (read-only at /root/reference).

Runs only in the build container (the reference does not exist on the GPU box).
Nothing from the reference is copied: the script imports it, feeds it seeded
inputs and records (inputs, parameters, the eps it drew, outputs, KL).  Recipe:
SURVEY.md Appendix B.

  * `termcolor` (imported but unused by conv_variational.py:55) is stubbed.
  * Reparameterization layers run with prior_type='normal'.
  * Flipout layers run with return_kl=False (their kl_div call lacks prior_type
    in this fork and raises); KL is obtained from the layer's own
    `kl_div(mu, log1p(exp(rho)), prior_mu, prior_sigma, 'normal')`
    (base_variational_layer.py:68-72) on the layer's own tensors.
  * eps_* are read back from the layer's buffers after forward; Flipout's
    sign tensors are recovered by replaying the seeded draws in the layer's
    draw order and asserting the replayed eps equals the buffer.

  * Classes whose kl_div call lacks prior_type in this fork (every layer except Linear/Conv2d
    Reparameterization) run with return_kl=False; LSTM wrappers, which always unpack (out, kl) from their
    inner Linear layers, get the documented 4-argument normal KL bound on those inner layers (shim (ii)).
  * 'laplace' fixtures call kl_div(..., 'laplace') (base_variational_layer.py:74-97; it hard-codes
    b_p = 1, mu_p = 0 and prints -- stdout is swallowed here).

Usage:  python tools/make_goldens.py [section ...]     (writes tests/golden/; sections: layers conv1d lstm
        models r50 laplace misc transpose conv3d; no argument = all)
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
_tc = types.ModuleType("termcolor")
_tc.colored = lambda s, *a, **k: s
sys.modules["termcolor"] = _tc

import numpy as np
import torch
import torch.nn as nn

import bayesian_torch.layers as RL                                   # the reference
from bayesian_torch.models.dnn_to_bnn import dnn_to_bnn as ref_dnn_to_bnn, get_kl_loss as ref_get_kl_loss
from bayesian_torch.utils.util import get_rho as ref_get_rho
from bayesian_torch.utils import util as ref_util
from bayesian_torch_amd.harness import resnet as H                    # plain-torch skeletons (ours)

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)


def sp(rho):
    return torch.log1p(torch.exp(rho))


def npf(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float32)


def save(name, meta, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    arrs["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  {name}.npz  {sz/1024:.1f} KiB")


def layer_case(name, cls_name, ctor, x_shape, seeds=(11, 12, 13), random_priors=False):
    seed_p, seed_x, seed_f = seeds
    flip = cls_name.endswith("Flipout")
    cls = getattr(RL, cls_name)
    torch.manual_seed(seed_p)
    kw = dict(ctor)
    has_pt = cls_name in ("LinearReparameterization", "Conv2dReparameterization")   # the only two ctors with prior_type in this fork
    if has_pt:
        kw["prior_type"] = "normal"
    layer = cls(**kw)
    wn = "kernel" if "Conv" in cls_name else "weight"
    if random_priors:  # users overwrite the prior buffers per element (utils/util.py:102-117)
        g = torch.Generator().manual_seed(seed_p + 100)
        layer.prior_weight_mu.copy_(0.05 * torch.randn(layer.prior_weight_mu.shape, generator=g))
        layer.prior_weight_sigma.copy_(0.5 + torch.rand(layer.prior_weight_sigma.shape, generator=g))
        if layer.mu_bias is not None:
            layer.prior_bias_mu.copy_(0.05 * torch.randn(layer.prior_bias_mu.shape, generator=g))
            layer.prior_bias_sigma.copy_(0.5 + torch.rand(layer.prior_bias_sigma.shape, generator=g))
    torch.manual_seed(seed_x)
    x = torch.randn(*x_shape)
    mu_w, rho_w = getattr(layer, "mu_" + wn), getattr(layer, "rho_" + wn)
    with torch.no_grad():
        torch.manual_seed(seed_f)
        if not has_pt:
            out = layer(x, return_kl=False)
        else:
            out, kl_fwd = layer(x)
        eps_w = getattr(layer, "eps_" + wn).clone()
        eps_b = layer.eps_bias.clone() if layer.mu_bias is not None else None
        sign_in = sign_out = None
        if flip:
            # replay the draws (order: linear_flipout.py:149-170 / conv_flipout.py:385-402)
            torch.manual_seed(seed_f)
            if "Conv" in cls_name:
                sign_in = torch.empty_like(x).uniform_(-1, 1).sign()
                sign_out = torch.empty_like(out).uniform_(-1, 1).sign()
                e_w = torch.empty_like(eps_w).normal_()
                e_b = torch.empty_like(eps_b).normal_() if eps_b is not None else None
            else:
                e_w = torch.empty_like(eps_w).normal_()
                e_b = torch.empty_like(eps_b).normal_() if eps_b is not None else None
                sign_in = torch.empty_like(x).uniform_(-1, 1).sign()
                sign_out = torch.empty_like(out).uniform_(-1, 1).sign()
            assert torch.equal(e_w, eps_w), "replayed eps != buffer"
            assert e_b is None or torch.equal(e_b, eps_b)
        pm_w, ps_w = layer.prior_weight_mu, layer.prior_weight_sigma
        if ps_w.shape != mu_w.shape or pm_w.shape != mu_w.shape:
            # ConvTranspose*Flipout register prior_weight_sigma as [Co][Ci/g].. while the kernel is [Ci][Co/g].. (conv_flipout.py:
            # 904-908): their own KL cannot broadcast. The buffers are constant fills, so the KL of record uses the same constants
            # in the kernel's shape (what every other class does).
            pm_w, ps_w = torch.full_like(mu_w, float(layer.prior_mean)), torch.full_like(mu_w, float(layer.prior_variance))
        kl_w = layer.kl_div(mu_w, sp(rho_w), pm_w, ps_w, "normal")
        kl_b = None
        kl = kl_w.clone()
        if layer.mu_bias is not None:
            kl_b = layer.kl_div(layer.mu_bias, sp(layer.rho_bias), layer.prior_bias_mu, layer.prior_bias_sigma, "normal")
            kl = kl_w + kl_b
        if has_pt:
            assert torch.equal(kl, kl_fwd), (kl, kl_fwd)
            assert torch.equal(kl, layer.kl_loss())
    meta = dict(cls=cls_name, ctor=ctor, x_shape=list(x_shape), seeds=list(seeds), torch=torch.__version__,
                random_priors=random_priors, prior_mean=float(layer.prior_mean), prior_variance=float(layer.prior_variance))
    save(name, meta, x=npf(x), mu_w=npf(mu_w), rho_w=npf(rho_w), mu_b=npf(layer.mu_bias), rho_b=npf(layer.rho_bias),
         prior_mu_w=npf(layer.prior_weight_mu) if random_priors else None,
         prior_sigma_w=npf(layer.prior_weight_sigma) if random_priors else None,
         prior_mu_b=npf(layer.prior_bias_mu) if random_priors and layer.mu_bias is not None else None,
         prior_sigma_b=npf(layer.prior_bias_sigma) if random_priors and layer.mu_bias is not None else None,
         eps_w=npf(eps_w), eps_b=npf(eps_b), sign_in=npf(sign_in), sign_out=npf(sign_out),
         out=npf(out), kl_w=npf(kl_w), kl_b=npf(kl_b), kl=npf(kl))


LINEAR_CASES = [
    ("cfg1", dict(in_features=784, out_features=10), (64, 784), False),
    ("nobias", dict(in_features=33, out_features=7, bias=False), (5, 33), False),
    ("k500", dict(in_features=500, out_features=24, prior_mean=0.1, prior_variance=0.7), (6, 500), False),
    ("rprior", dict(in_features=96, out_features=40), (9, 96), True),
]
CONV_CASES = [
    ("c3x8k3", dict(in_channels=3, out_channels=8, kernel_size=3, padding=1), (2, 3, 8, 8), False),
    ("c8x16k3s2", dict(in_channels=8, out_channels=16, kernel_size=3, stride=2, padding=1), (3, 8, 9, 9), False),
    ("c16x32k1s2nb", dict(in_channels=16, out_channels=32, kernel_size=1, stride=2, bias=False), (2, 16, 6, 6), False),
    ("c3x16k7s2", dict(in_channels=3, out_channels=16, kernel_size=7, stride=2, padding=3), (2, 3, 32, 32), False),
    ("c4x4k3d2", dict(in_channels=4, out_channels=4, kernel_size=3, dilation=2, padding=2), (2, 4, 7, 7), False),
    ("c8x12g2", dict(in_channels=8, out_channels=12, kernel_size=3, padding=1, groups=2), (2, 8, 5, 5), False),
    ("c6x10k3x2", dict(in_channels=6, out_channels=10, kernel_size=(3, 2), stride=(2, 1), padding=(1, 0)), (2, 6, 7, 6), True),
    ("c64x64k3hw1", dict(in_channels=64, out_channels=64, kernel_size=3, padding=1, bias=False), (4, 64, 1, 1), False),
]



def lstm_case(name, cls_name, in_f, out_f, x_shape, seeds=(21, 22, 23)):
    """LSTM{Reparameterization,Flipout} (rnn_variational.py:45-153, rnn_flipout.py:46-154): two Linear layers per
    time step. Per-step draws are recovered by replaying the seeded generator in the layers' draw order."""
    seed_p, seed_x, seed_f = seeds
    flip = cls_name.endswith("Flipout")
    torch.manual_seed(seed_p)
    lstm = getattr(RL, cls_name)(in_f, out_f)
    if flip:   # shim (ii): LinearFlipout's own kl_div call lacks prior_type in this fork
        for lin in (lstm.ih, lstm.hh):
            lin.kl_div = (lambda l: (lambda mq, sq, mp, sp_: type(l).kl_div(l, mq, sq, mp, sp_, "normal")))(lin)
    torch.manual_seed(seed_x)
    X = torch.randn(*x_shape)
    B, T = x_shape[0], x_shape[1]
    with torch.no_grad():
        torch.manual_seed(seed_f)
        hs, (_, cs), kl = lstm(X)
        torch.manual_seed(seed_f)
        d = {f"{nm}_{k}": [] for nm in ("ih", "hh") for k in ("eps_w", "eps_b", "sign_in", "sign_out")}
        for t in range(T):
            for nm, fin in (("ih", in_f), ("hh", out_f)):
                lin = getattr(lstm, nm)
                d[nm + "_eps_w"].append(torch.empty_like(lin.eps_weight).normal_())
                d[nm + "_eps_b"].append(torch.empty_like(lin.eps_bias).normal_())
                if flip:    # linear_flipout.py:149-170: eps_w, eps_b, s_in, s_out
                    d[nm + "_sign_in"].append(torch.empty(B, fin).uniform_(-1, 1).sign())
                    d[nm + "_sign_out"].append(torch.empty(B, 4 * out_f).uniform_(-1, 1).sign())
        for nm in ("ih", "hh"):
            assert torch.equal(d[nm + "_eps_w"][-1], getattr(lstm, nm).eps_weight), "replayed eps != buffer"
        if flip:
            kl_loss = sum(l.kl_div(l.mu_weight, sp(l.rho_weight), l.prior_weight_mu, l.prior_weight_sigma)
                          + l.kl_div(l.mu_bias, sp(l.rho_bias), l.prior_bias_mu, l.prior_bias_sigma) for l in (lstm.ih, lstm.hh))
        else:
            kl_loss = lstm.kl_loss()
    arrs = dict(x=npf(X), hidden_seq=npf(hs), c_ts=npf(cs), kl=npf(kl), kl_loss=npf(kl_loss))
    for nm in ("ih", "hh"):
        lin = getattr(lstm, nm)
        arrs.update({nm + "_mu_w": npf(lin.mu_weight), nm + "_rho_w": npf(lin.rho_weight), nm + "_mu_b": npf(lin.mu_bias), nm + "_rho_b": npf(lin.rho_bias)})
    for k, v in d.items():
        if v:
            arrs[k] = npf(torch.stack(v))
    save(name, dict(cls=cls_name, in_features=in_f, out_features=out_f, x_shape=list(x_shape), seeds=list(seeds), torch=torch.__version__), **arrs)


def laplace_cases():
    """kl_div(..., 'laplace') known answers (base_variational_layer.py:74-97). The branch ignores the prior tensors it is
    handed (b_p = 1, mu_p = 0 are hard-coded) and prints; also through LinearReparameterization / Conv2dReparameterization
    constructed with prior_type='laplace' (forward kl == kl_loss())."""
    import contextlib
    import io
    g = torch.Generator().manual_seed(41)
    base = RL.LinearReparameterization(2, 2)
    arrs, meta = {}, dict(cases=[])
    for tag, n, mu_s, rho_m in (("a", 4099, 0.1, -3.0), ("b", 513, 1.5, 0.5), ("c", 64, 0.0, -8.0)):
        mu = torch.randn(n, generator=g) * mu_s
        if tag == "c":
            mu[:8] = 0.0          # exactly-zero means: the folded-normal mean reduces to sigma*sqrt(2/pi)
        rho = torch.randn(n, generator=g) * 0.3 + rho_m
        pm, ps = torch.randn(n, generator=g), torch.rand(n, generator=g) + 0.5      # ignored by the branch
        with contextlib.redirect_stdout(io.StringIO()):
            kl = base.kl_div(mu, sp(rho), pm, ps, "laplace")
        arrs.update({f"{tag}_mu": npf(mu), f"{tag}_rho": npf(rho), f"{tag}_pmu": npf(pm), f"{tag}_psig": npf(ps), f"{tag}_kl": npf(kl)})
        meta["cases"].append(tag)
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        torch.manual_seed(42)
        lin = RL.LinearReparameterization(37, 11, prior_type="laplace")
        out, kl_f = lin(torch.randn(3, 37))
        assert torch.equal(kl_f, lin.kl_loss())
        arrs.update(lin_mu_w=npf(lin.mu_weight), lin_rho_w=npf(lin.rho_weight), lin_mu_b=npf(lin.mu_bias), lin_rho_b=npf(lin.rho_bias), lin_kl=npf(kl_f))
        conv = RL.Conv2dReparameterization(5, 6, 3, prior_type="laplace")
        out, kl_c = conv(torch.randn(2, 5, 6, 6))
        assert torch.equal(kl_c, conv.kl_loss())
        arrs.update(conv_mu_w=npf(conv.mu_kernel), conv_rho_w=npf(conv.rho_kernel), conv_mu_b=npf(conv.mu_bias), conv_rho_b=npf(conv.rho_bias), conv_kl=npf(kl_c))
    save("kl_laplace", meta, **arrs)


CONV1D_CASES = [("c6x10k3s2", dict(in_channels=6, out_channels=10, kernel_size=3, stride=2, padding=1), (3, 6, 17))]
CONVT2D_CASES = [
    ("c6x4k3s2", dict(in_channels=6, out_channels=4, kernel_size=3, stride=2, padding=1, output_padding=1), (2, 6, 5, 5)),
    ("c8x6k4s2g2", dict(in_channels=8, out_channels=6, kernel_size=4, stride=2, padding=1, groups=2), (2, 8, 4, 6)),
    ("c5x7k3d2nb", dict(in_channels=5, out_channels=7, kernel_size=(3, 2), stride=1, padding=(2, 0), dilation=(2, 1), bias=False), (3, 5, 6, 5)),
]
CONVT1D_CASES = [("c6x4k4s2", dict(in_channels=6, out_channels=4, kernel_size=4, stride=2, padding=1), (3, 6, 9))]
CONV3D_CASES = [
    ("c3x5k3", dict(in_channels=3, out_channels=5, kernel_size=3, padding=1), (2, 3, 4, 6, 6)),
    ("c4x6k3s2g2", dict(in_channels=4, out_channels=6, kernel_size=(3, 3, 2), stride=(2, 2, 1), padding=(1, 1, 0), groups=2), (2, 4, 5, 7, 4)),
]
CONVT3D_CASES = [("c4x3k3s2", dict(in_channels=4, out_channels=3, kernel_size=3, stride=2, padding=1, output_padding=1), (2, 4, 3, 3, 3))]


def model_case(name, build, x_shape, btype, S, seed, store_full):
    """dnn_to_bnn(model) -> S sequential MC samples -> logits[S,B,C], kl, mean prob.
    Loop semantics: examples/main_bayesian_cifar_dnn2bnn.py:402-412,551-557."""
    torch.manual_seed(seed)
    model = build()
    params = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
              "type": btype, "moped_enable": False, "moped_delta": 0.5}
    ref_dnn_to_bnn(model, params)
    H.fill_bayes_params(model, seed)
    layers = H.bayes_layers(model)
    flip = btype == "Flipout"
    for _, m in layers:
        if hasattr(m, "prior_type") and m.prior_type is None:
            m.prior_type = "normal"            # oracle shim (i)
    model.eval()
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.randn(*x_shape, generator=g)
    logits, eps_sum = [], []
    with torch.no_grad():
        if flip:   # shim (ii): KL from the layer's own kl_div
            kl = None
            for _, m in layers:
                wn = "kernel" if hasattr(m, "mu_kernel") else "weight"
                k = m.kl_div(getattr(m, "mu_" + wn), sp(getattr(m, "rho_" + wn)), m.prior_weight_mu, m.prior_weight_sigma, "normal")
                if m.mu_bias is not None:
                    k = k + m.kl_div(m.mu_bias, sp(m.rho_bias), m.prior_bias_mu, m.prior_bias_sigma, "normal")
                kl = k if kl is None else kl + k
        else:
            kl = ref_get_kl_loss(model)
        for s in range(S):
            torch.manual_seed(seed * 100 + s)       # per-sample seed: the draw can be replayed sample by sample
            logits.append(model(x))
            eps_sum.append([float(getattr(m, "eps_kernel" if hasattr(m, "mu_kernel") else "eps_weight").double().sum())
                            for _, m in layers])
    logits = torch.stack(logits)
    prob = torch.softmax(logits, -1).mean(0)
    pchk = [[float(p.double().sum()), float((p.double() ** 2).sum())] for _, m in layers for p in m.parameters()]
    meta = dict(kind="model", btype=btype, x_shape=list(x_shape), S=S, seed=seed, torch=torch.__version__,
                layer_names=[n for n, _ in layers], param_checksums=pchk, eps_sums=eps_sum,
                x_checksum=[float(x.double().sum()), float((x.double() ** 2).sum())])
    arrs = dict(logits=npf(logits), kl=npf(kl), mean_prob=npf(prob))
    if store_full:
        arrs["x"] = npf(x)
    save(name, meta, **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:])
    sec = lambda n: not want or n in want
    if sec("layers"):
        print("layer fixtures")
        for tag, ctor, xs, rp in LINEAR_CASES:
            layer_case("linear_reparam_" + tag, "LinearReparameterization", ctor, xs, random_priors=rp)
            layer_case("linear_flipout_" + tag, "LinearFlipout", ctor, xs, random_priors=rp)
        for tag, ctor, xs, rp in CONV_CASES:
            layer_case("conv2d_reparam_" + tag, "Conv2dReparameterization", ctor, xs, random_priors=rp)
            layer_case("conv2d_flipout_" + tag, "Conv2dFlipout", ctor, xs, random_priors=rp)
    if sec("conv1d"):
        for tag, ctor, xs in CONV1D_CASES:
            layer_case("conv1d_reparam_" + tag, "Conv1dReparameterization", ctor, xs)
            layer_case("conv1d_flipout_" + tag, "Conv1dFlipout", ctor, xs)
    if sec("transpose"):
        for tag, ctor, xs in CONVT2D_CASES:
            layer_case("convt2d_reparam_" + tag, "ConvTranspose2dReparameterization", ctor, xs)
            layer_case("convt2d_flipout_" + tag, "ConvTranspose2dFlipout", ctor, xs)
        for tag, ctor, xs in CONVT1D_CASES:
            layer_case("convt1d_reparam_" + tag, "ConvTranspose1dReparameterization", ctor, xs)
            layer_case("convt1d_flipout_" + tag, "ConvTranspose1dFlipout", ctor, xs)
        for tag, ctor, xs in CONVT3D_CASES:
            layer_case("convt3d_reparam_" + tag, "ConvTranspose3dReparameterization", ctor, xs)
            # ConvTranspose3dFlipout.forward with a bias and return_kl=False raises UnboundLocalError in this fork
            # (conv_flipout.py:1192 adds the bias KL unconditionally): its fixture is bias-free
            layer_case("convt3d_flipout_" + tag + "nb", "ConvTranspose3dFlipout", dict(ctor, bias=False), xs)
    if sec("conv3d"):
        for tag, ctor, xs in CONV3D_CASES:
            # Conv3dReparameterization takes its prior / posterior arguments positionally, without defaults (conv_variational.py:651-663)
            layer_case("conv3d_reparam_" + tag, "Conv3dReparameterization",
                       dict(ctor, prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0), xs)
            layer_case("conv3d_flipout_" + tag, "Conv3dFlipout", ctor, xs)
    if sec("lstm"):
        lstm_case("lstm_reparam_7x5", "LSTMReparameterization", 7, 5, (3, 4, 7))
        lstm_case("lstm_flipout_7x5", "LSTMFlipout", 7, 5, (3, 4, 7))
    if sec("laplace"):
        laplace_cases()

    if sec("models"):
        print("model fixtures")
        model_case("model_r18w8_reparam", lambda: H.resnet18(10, width=8), (4, 3, 32, 32), "Reparameterization", 3, 5, True)
        model_case("model_r18w8_flipout", lambda: H.resnet18(10, width=8), (4, 3, 32, 32), "Flipout", 3, 5, True)
        model_case("model_mlp_reparam", lambda: H.mlp((3072, 512, 10)), (8, 3072), "Reparameterization", 2, 6, False)
        # full width, cfg3/cfg4 of BASELINE.json: only (seed, logits, kl, checksums) are stored
        model_case("model_r18_reparam", lambda: H.resnet18(10, width=64), (128, 3, 32, 32), "Reparameterization", 1, 3, False)
        model_case("model_r18_flipout", lambda: H.resnet18(10, width=64), (128, 3, 32, 32), "Flipout", 1, 3, False)
    if sec("r50"):
        # cfg5's model at full width and full image size (ResNet50, 3x224x224); batch 2, one sample: seeds + logits + checksums
        model_case("model_r50_reparam", lambda: H.resnet50(1000, width=64), (2, 3, 224, 224), "Reparameterization", 1, 7, False)
    if not sec("misc"):
        return

    print("get_rho known answers (utils/util.py:63-69)")
    g = torch.Generator().manual_seed(21)
    w = torch.randn(257, generator=g) * 0.2
    w[0] = 0.0
    save("get_rho", dict(deltas=[0.1, 0.5]), w=npf(w), rho_0p1=npf(ref_get_rho(w, 0.1)), rho_0p5=npf(ref_get_rho(w, 0.5)))

    print("uncertainty measures (utils/util.py:41-60)")
    g = torch.Generator().manual_seed(31)
    mc = torch.softmax(torch.randn(7, 9, 10, generator=g) * 2.0, -1).numpy().astype(np.float64)
    save("uncertainty", dict(shape=[7, 9, 10]), mc_preds=mc.astype(np.float32), entropy=ref_util.entropy(mc.astype(np.float32)).astype(np.float32),
         predictive_entropy=ref_util.predictive_entropy(mc.astype(np.float32)).astype(np.float32),
         mutual_information=ref_util.mutual_information(mc.astype(np.float32)).astype(np.float32))

    print("negative paths")
    neg = []
    for desc, fn in [
        ("Conv2dReparameterization(in_channels=3,out_channels=8,kernel_size=3,groups=2)",
         lambda: RL.Conv2dReparameterization(3, 8, 3, groups=2)),
        ("Conv2dReparameterization(in_channels=4,out_channels=6,kernel_size=3,groups=4)",
         lambda: RL.Conv2dReparameterization(4, 6, 3, groups=4)),
        ("LinearReparameterization(4,4,prior_type=None)", lambda: RL.LinearReparameterization(4, 4, prior_type=None)),
        ("dnn_to_bnn missing key 'type'", lambda: ref_dnn_to_bnn(nn.Sequential(nn.Linear(2, 2)), {"prior_mu": 0.0, "prior_sigma": 1.0,
            "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "moped_enable": False, "moped_delta": 0.5})),
        ("dnn_to_bnn unsupported type 'Foo'", lambda: ref_dnn_to_bnn(nn.Sequential(nn.Linear(2, 2)), {"prior_mu": 0.0, "prior_sigma": 1.0,
            "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "type": "Foo", "moped_enable": False, "moped_delta": 0.5})),
        ("kl_div unknown prior_type", lambda: RL.LinearReparameterization(2, 2).kl_div(torch.ones(1), torch.ones(1), torch.ones(1), torch.ones(1), "xyz")),
    ]:
        try:
            fn()
            neg.append(dict(call=desc, raises=None))
        except Exception as e:  # noqa: BLE001 -- recording the reference's behaviour
            neg.append(dict(call=desc, raises=type(e).__name__, message=str(e)))
    # API-surface facts the drop-in must keep
    lr = RL.LinearReparameterization(3, 2)
    cr = RL.Conv2dReparameterization(3, 2, 3, prior_type="normal")
    lf = RL.LinearFlipout(3, 2)
    cf = RL.Conv2dFlipout(3, 2, 3)
    api = {}
    for nm, m in [("LinearReparameterization", lr), ("Conv2dReparameterization", cr), ("LinearFlipout", lf), ("Conv2dFlipout", cf)]:
        api[nm] = dict(state_dict_keys=list(m.state_dict().keys()), buffers=[n for n, _ in m.named_buffers()],
                       repr=repr(m), posterior_mu_init=repr(m.posterior_mu_init), posterior_rho_init=repr(m.posterior_rho_init),
                       quant_prepare=m.quant_prepare, dnn_to_bnn_flag=m.dnn_to_bnn_flag)
    with open(os.path.join(OUT, "api_surface.json"), "w") as f:
        json.dump(dict(negative=neg, api=api), f, indent=1)
    print(json.dumps(neg, indent=1))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
