"""CPU: the drop-in boundary -- class names, signatures, parameter/buffer names, attribute quirks,
error behaviour, dnn_to_bnn dispatch, the alias package, and the C-ABI library's export table.  No kernels are launched."""
import ctypes
import inspect
import json
import os
import re

import pytest
import torch
import torch.nn as nn

from conftest import GOLD, ROOT

API = json.load(open(os.path.join(GOLD, "api_surface.json")))
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
         "type": "Reparameterization", "moped_enable": False, "moped_delta": 0.5}


def _mk(name):
    import bayesian_torch.layers as L          # the alias package with the reference's import paths
    return {"LinearReparameterization": lambda: L.LinearReparameterization(3, 2),
            "Conv2dReparameterization": lambda: L.Conv2dReparameterization(3, 2, 3, prior_type="normal"),
            "LinearFlipout": lambda: L.LinearFlipout(3, 2), "Conv2dFlipout": lambda: L.Conv2dFlipout(3, 2, 3)}[name]()


@pytest.mark.parametrize("name", sorted(API["api"]))
def test_layer_surface_matches_reference(name):
    ref = API["api"][name]
    m = _mk(name)
    assert list(m.state_dict().keys()) == ref["state_dict_keys"]          # checkpoints of the reference load unchanged
    assert [n for n, _ in m.named_buffers()] == ref["buffers"]
    assert repr(m) == ref["repr"]                                         # MOPED() matches on this string
    assert repr(m.posterior_mu_init) == ref["posterior_mu_init"] and repr(m.posterior_rho_init) == ref["posterior_rho_init"]
    assert m.quant_prepare is False and m.dnn_to_bnn_flag is False
    m.dnn_to_bnn_flag = True
    assert m.dnn_to_bnn_flag is True


def test_constructor_signatures():
    import bayesian_torch_amd.layers as L
    sig = lambda c: list(inspect.signature(c.__init__).parameters)[1:]
    assert sig(L.LinearReparameterization) == ["in_features", "out_features", "prior_mean", "prior_variance", "posterior_mu_init",
                                               "posterior_rho_init", "bias", "prior_type"]
    assert sig(L.Conv2dReparameterization) == ["in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups",
                                               "prior_mean", "prior_variance", "prior_type", "posterior_mu_init", "posterior_rho_init", "bias"]
    assert sig(L.LinearFlipout) == ["in_features", "out_features", "prior_mean", "prior_variance", "posterior_mu_init", "posterior_rho_init", "bias"]
    assert sig(L.Conv2dFlipout) == ["in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups", "prior_mean",
                                    "prior_variance", "posterior_mu_init", "posterior_rho_init", "bias"]
    for c in (L.LinearReparameterization, L.Conv2dReparameterization, L.LinearFlipout, L.Conv2dFlipout):
        assert list(inspect.signature(c.forward).parameters)[1:3] in (["input", "return_kl"], ["x", "return_kl"])
        assert inspect.signature(c.forward).parameters["return_kl"].default is True


def test_seeded_construction_matches_reference_init_order():
    """Same draw order at construction as the reference: a seeded layer gets the reference's initial parameters
    (checked against the golden fixture, whose parameters came from the reference's constructor under seed 11)."""
    import numpy as np
    import bayesian_torch_amd.layers as L
    g = np.load(os.path.join(GOLD, "linear_reparam_cfg1.npz"))
    torch.manual_seed(11)
    m = L.LinearReparameterization(784, 10)
    assert torch.equal(m.mu_weight.detach(), torch.from_numpy(g["mu_w"])) and torch.equal(m.rho_bias.detach(), torch.from_numpy(g["rho_b"]))
    g = np.load(os.path.join(GOLD, "conv2d_flipout_c8x16k3s2.npz"))
    torch.manual_seed(11)
    c = L.Conv2dFlipout(8, 16, 3, stride=2, padding=1)
    assert torch.equal(c.rho_kernel.detach(), torch.from_numpy(g["rho_w"])) and torch.equal(c.mu_bias.detach(), torch.from_numpy(g["mu_b"]))
    assert float(c.prior_weight_sigma.min()) == 1.0 and tuple(c.eps_kernel.shape) == (16, 8, 3, 3)


def test_negative_paths_match_reference():
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    neg = {d["call"]: d for d in API["negative"]}
    with pytest.raises(ValueError, match="invalid in_channels size"):
        L.Conv2dReparameterization(3, 8, 3, groups=2)
    with pytest.raises(ValueError, match="invalid in_channels size"):
        L.Conv2dReparameterization(4, 6, 3, groups=4)
    with pytest.raises(AssertionError, match=re.escape(neg["LinearReparameterization(4,4,prior_type=None)"]["message"])):
        L.LinearReparameterization(4, 4, prior_type=None)
    bad = dict(PRIOR)
    del bad["type"]
    with pytest.raises(KeyError, match="type"):
        dnn_to_bnn(nn.Sequential(nn.Linear(2, 2)), bad)
    with pytest.raises(AttributeError, match="LinearFoo"):
        dnn_to_bnn(nn.Sequential(nn.Linear(2, 2)), dict(PRIOR, type="Foo"))
    with pytest.raises(AttributeError):       # class looked up by name: a layer kind the package does not have
        dnn_to_bnn(nn.Sequential(nn.LazyConv2d(2, 3)), PRIOR)
    m3 = nn.Sequential(nn.Conv3d(2, 2, 3), nn.ConvTranspose2d(2, 4, 3, stride=2))     # the rest of the conv family converts, as in the reference
    dnn_to_bnn(m3, PRIOR)
    assert type(m3[0]).__name__ == "Conv3dReparameterization" and type(m3[1]).__name__ == "ConvTranspose2dReparameterization"
    assert tuple(m3[1].mu_kernel.shape) == (2, 4, 3, 3) and m3[1].dnn_to_bnn_flag
    seq = nn.Sequential(nn.LSTM(3, 5), nn.Conv1d(2, 4, 3))
    dnn_to_bnn(seq, PRIOR)
    assert repr(seq[0]) == "LSTMReparameterization(\n  (ih): LinearReparameterization()\n  (hh): LinearReparameterization()\n)"
    assert repr(seq[1]) == "Conv1dReparameterization()" and tuple(seq[1].mu_kernel.shape) == (4, 2, 3)
    lay = L.LinearReparameterization(2, 2)
    with pytest.raises(ValueError, match="Unknown prior_type: xyz"):
        lay.kl_div(torch.ones(1), torch.ones(1), torch.ones(1), torch.ones(1), "xyz")
    lay.prior_type = "xyz"
    with pytest.raises(ValueError, match="Unknown prior_type: xyz"):
        lay.kl_loss()


def test_no_cpu_fallback():
    """The HIP path is the only implementation: CPU tensors are refused, loudly."""
    import bayesian_torch_amd.layers as L
    with pytest.raises(RuntimeError, match="HIP"):
        L.LinearReparameterization(4, 3)(torch.randn(2, 4))
    with pytest.raises(RuntimeError, match="HIP"):
        L.Conv2dFlipout(2, 3, 3).kl_loss()


def test_dnn_to_bnn_dispatch_and_moped():
    from bayesian_torch.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    from bayesian_torch.utils.util import get_rho
    import bayesian_torch_amd.layers as L

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(4), nn.ReLU())
            self.head = nn.Linear(4, 2)

    net = Net()
    w_conv, w_lin, b_lin = net.features[0].weight.detach().clone(), net.head.weight.detach().clone(), net.head.bias.detach().clone()
    assert dnn_to_bnn(net, dict(PRIOR, moped_enable=True, moped_delta=0.25)) is None      # in place, returns None
    conv, lin = net.features[0], net.head
    assert isinstance(conv, L.Conv2dReparameterization) and isinstance(lin, L.LinearReparameterization)
    assert isinstance(net.features[1], nn.BatchNorm2d)
    assert conv.dnn_to_bnn_flag and lin.dnn_to_bnn_flag
    assert conv.stride == (2, 2) and conv.padding == (1, 1) and conv.mu_bias is None and conv.bias is False
    assert torch.equal(conv.mu_kernel.detach(), w_conv) and torch.equal(lin.mu_bias.detach(), b_lin)
    assert torch.allclose(lin.rho_weight.detach(), get_rho(w_lin, 0.25))
    assert torch.allclose(torch.log1p(torch.exp(lin.rho_weight.detach())), 0.25 * w_lin.abs(), atol=1e-6)
    net2 = Net()
    dnn_to_bnn(net2, dict(PRIOR, type="Flipout"))
    assert isinstance(net2.features[0], L.Conv2dFlipout) and isinstance(net2.head, L.LinearFlipout)
    assert get_kl_loss(nn.Sequential(nn.ReLU())) is None


def test_get_rho_golden():
    import numpy as np
    from bayesian_torch_amd.utils.util import get_rho
    g = np.load(os.path.join(GOLD, "get_rho.npz"))
    w = torch.from_numpy(g["w"])
    assert torch.allclose(get_rho(w, 0.1), torch.from_numpy(g["rho_0p1"]), rtol=1e-6, atol=0)
    assert torch.allclose(get_rho(w, 0.5), torch.from_numpy(g["rho_0p5"]), rtol=1e-6, atol=0)


def test_library_exports_every_declared_symbol():
    """include/bt_hip.h <-> libbtorch_hip.so: every declared entry point is exported (no compute call is made)."""
    from bayesian_torch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "bt_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char \*)\s*\*?(bt_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), name
    L = _lib.lib()
    assert L.bt_version() == 302
    # host-only entry: Philox4x32-10 known answers (Random123 kat_vectors)
    def philox(ctr, key):
        c = (ctypes.c_uint32 * 4)(*ctr)
        out = (ctypes.c_uint32 * 4)()
        assert L.bt_rng_philox_raw(key[0] | (key[1] << 32), c, out) == 0
        return [hex(v) for v in out]
    assert philox([0, 0, 0, 0], [0, 0]) == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    # argument validation happens on the host, before any launch
    assert L.bt_mc_epilogue(0, 1, 1, None, None, None) == -1 and b"bt_mc_epilogue" in L.bt_last_error_string()


def test_struct_layouts_match_header():
    from bayesian_torch_amd import _lib
    assert ctypes.sizeof(_lib.bt_rng) == 32 and ctypes.sizeof(_lib.bt_params) == 88
    assert ctypes.sizeof(_lib.bt_draws) == 64 and ctypes.sizeof(_lib.bt_conv2d_geom) == 56 and ctypes.sizeof(_lib.bt_epilogue) == 40
    assert ctypes.sizeof(_lib.bt_pack_seg) == 88


def test_fold_batchnorm_host_logic():
    from bayesian_torch_amd.fuse import fold_batchnorm
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    net = H.resnet18(10, 8)
    dnn_to_bnn(net, PRIOR)
    with pytest.raises(RuntimeError, match="eval"):
        fold_batchnorm(net)
    net.eval()
    bn = net.layer1[0].bn1
    bn.running_mean.fill_(0.5), bn.running_var.fill_(4.0), bn.weight.data.fill_(2.0), bn.bias.data.fill_(-1.0)
    assert fold_batchnorm(net) == 20
    c = net.layer1[0].conv1
    assert torch.allclose(c.post_scale, torch.full((8,), 2.0 / (4.0 + 1e-5) ** 0.5)) and torch.allclose(c.post_shift, -1.0 - 0.5 * c.post_scale)
    assert isinstance(net.layer1[0].bn1, nn.Identity) and "post_scale" not in net.state_dict()


def test_uncertainty_measures_match_reference():
    import numpy as np
    from bayesian_torch.utils.util import entropy, predictive_entropy, mutual_information
    from bayesian_torch_amd.utils.util import uncertainty_from_mc
    g = np.load(os.path.join(GOLD, "uncertainty.npz"))
    mc = g["mc_preds"]
    assert np.allclose(entropy(mc), g["entropy"], rtol=1e-6, atol=1e-7)
    assert np.allclose(predictive_entropy(mc), g["predictive_entropy"], rtol=1e-6, atol=1e-7)
    assert np.allclose(mutual_information(mc), g["mutual_information"], rtol=1e-5, atol=1e-6)
    # packed-sum form (what the MC epilogue kernel produces): same numbers without the [S, B, C] tensor
    t = torch.from_numpy(mc)
    res = dict(mean_prob=t.mean(0), mean_entropy=(-(t * torch.log(t.clamp_min(1e-30))).sum(-1)).mean(0))
    pe, mi = uncertainty_from_mc(res)
    assert np.allclose(pe.numpy(), g["predictive_entropy"], rtol=1e-5, atol=1e-6)
    assert np.allclose(mi.numpy(), g["mutual_information"], rtol=1e-4, atol=1e-5)


def test_moped_from_checkpoint(tmp_path):
    from bayesian_torch.utils.util import MOPED, get_rho
    import bayesian_torch_amd.layers as L
    det = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Flatten(), nn.Linear(4, 2))
    ck = str(tmp_path / "det.pth")
    torch.save(det.state_dict(), ck)
    bnn = nn.Sequential(L.Conv2dReparameterization(3, 4, 3, prior_type="normal"), nn.BatchNorm2d(4), nn.Flatten(), L.LinearFlipout(4, 2))
    det2 = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Flatten(), nn.Linear(4, 2))
    MOPED(bnn, det2, ck, 0.3)
    assert torch.equal(bnn[0].mu_kernel.data, det[0].weight.data) and torch.equal(bnn[0].prior_weight_mu, det[0].weight.data)
    assert torch.allclose(bnn[3].rho_weight.data, get_rho(det[3].weight.data, 0.3)) and torch.equal(bnn[3].mu_bias.data, det[3].bias.data)


def test_two_models_do_not_share_workspaces():
    """layer ids are positional (RNG coordinate) but the KL / pack workspaces are keyed per live layer object."""
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    nets = []
    for _ in range(2):
        net = H.resnet18(10, 8)
        dnn_to_bnn(net, PRIOR)
        nets.append(net)
    a, b = ([m for _, m in H.bayes_layers(n)] for n in nets)
    assert [m._layer_id for m in a] == [m._layer_id for m in b]
    assert not ({m._ws_id for m in a} & {m._ws_id for m in b})
