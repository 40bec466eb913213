"""Conv3d / ConvTranspose{1,2,3}d: the index re-arrangements around the fused Conv2d launch
(host logic, CPU) and the layers themselves against the reference's goldens (GPU)."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import assert_close, golden_names, layer_tensors, load_golden

FIX = golden_names("conv3d_") + golden_names("convt")


def _layer(g):
    import bayesian_torch_amd.layers as L
    m = g["meta"]
    layer = getattr(L, m["cls"])(**m["ctor"])
    with torch.no_grad():
        layer.mu_kernel.copy_(g["mu_w"]), layer.rho_kernel.copy_(g["rho_w"])
        if g["mu_b"] is not None:
            layer.mu_bias.copy_(g["mu_b"]), layer.rho_bias.copy_(g["rho_b"])
    return layer


@pytest.mark.parametrize("name", FIX)
def test_rearrangements_reproduce_the_reference_on_cpu(name):
    """x and kernel re-arranged by the layer + a plain conv2d == the reference's conv3d / conv_transpose output: the host-side
    part of the family layers is exact (the GPU test then swaps conv2d for the fused launch)."""
    from oracle import bt_oracle as O
    g = layer_tensors(load_golden(name))
    layer = _layer(g)
    flip = "flipout" in name
    xe, conv, back = layer._x_eq(g["x"])
    sig_eps = O.softplus_ref(g["rho_w"]) * g["eps_w"]
    bias = None if g["mu_b"] is None else (g["mu_b"] + O.softplus_ref(g["rho_b"]) * g["eps_b"])
    c2 = lambda x_, w_, b_: TF.conv2d(x_, layer._w_eq(w_), b_, conv["stride"], conv["padding"], conv["dilation"], conv["groups"])
    if not flip:
        out = back(c2(xe, g["mu_w"] + sig_eps, bias))
    else:
        si = layer._x_eq(g["sign_in"])[0]
        so = layer._sign_out_eq(g["sign_out"].unsqueeze(0))[0]
        dbias = None if g["mu_b"] is None else O.softplus_ref(g["rho_b"]) * g["eps_b"]
        out = back(c2(xe, g["mu_w"], g["mu_b"]) + c2(xe * si, sig_eps, dbias) * so)
    assert_close(out, g["out"], 1e-5, 1e-6, name)
    assert tuple(layer.mu_kernel.shape) == tuple(g["mu_w"].shape) and repr(layer) == g["meta"]["cls"] + "()"


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIX)
def test_family_layers_match_reference_golden(name):
    from bayesian_torch_amd import rng
    g = layer_tensors(load_golden(name))
    layer = _layer(g).cuda()
    st = lambda k: g[k].cuda().unsqueeze(0) if g.get(k) is not None else None
    layer.inject_draw = dict(eps_w=st("eps_w"), eps_b=st("eps_b"), sign_in=st("sign_in"), sign_out=st("sign_out"))
    with torch.no_grad():
        out, kl = layer(g["x"].cuda())
    assert_close(out.cpu(), g["out"], 1e-4, 1e-5, name + ".out")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, name + ".kl")
    assert_close(layer.kl_loss().cpu(), g["kl"], 1e-5, 0, name + ".kl_loss")
    layer.inject_draw = None
    rng.set_mode("philox")
    with torch.no_grad():
        o2 = layer(g["x"].cuda(), return_kl=False)          # on-chip draws: shape, finiteness, a fresh draw per call
        o3 = layer(g["x"].cuda(), return_kl=False)
    assert o2.shape == out.shape and torch.isfinite(o2).all() and not torch.equal(o2, o3)


@pytest.mark.gpu
def test_family_layers_train_and_convert():
    """Gradients reach (mu, rho) through the re-arrangements (autograd bridge), and dnn_to_bnn converts nn.Conv3d /
    nn.ConvTranspose2d modules like the reference does (class looked up by name)."""
    import torch.nn as nn
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    import bayesian_torch_amd.layers as L
    net = nn.Sequential(nn.Conv3d(2, 8, 3, padding=1), nn.ReLU(), nn.Flatten(0, 1))
    dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "type": "Reparameterization",
                     "moped_enable": False, "moped_delta": 0.5})
    assert isinstance(net[0], L.Conv3dReparameterization) and net[0].dnn_to_bnn_flag
    net = net.cuda()
    y = net(torch.randn(2, 2, 4, 5, 5).cuda())
    assert y.shape == (16, 4, 5, 5)
    (y.square().mean() + get_kl_loss(net)).backward()
    assert torch.isfinite(net[0].mu_kernel.grad).all() and float(net[0].rho_kernel.grad.abs().sum()) > 0
    up = L.ConvTranspose2dFlipout(4, 6, 3, stride=2, padding=1, output_padding=1).cuda()
    o, kl = up(torch.randn(3, 4, 5, 5).cuda())
    assert o.shape == (3, 6, 10, 10)
    (o.mean() + kl).backward()
    assert torch.isfinite(up.mu_kernel.grad).all() and float(up.rho_kernel.grad.abs().sum()) > 0
