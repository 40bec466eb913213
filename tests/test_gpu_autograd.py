"""GPU: gradients of the drop-in layers (fused HIP forward + autograd bridge) against torch autograd of the CPU oracle
on the SAME draws (materialised from the RNG coordinates the forward used)."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _oracle_loss(kind, flip, params, x, draws, conv, gout, with_kl, priors):
    from oracle import bt_oracle as O
    p = {k: (v.detach().cpu().clone().requires_grad_(True) if v is not None else None) for k, v in params.items()}
    xc = x.detach().cpu().clone().requires_grad_(True)
    S = draws["eps_w"].shape[0]
    outs = []
    for s in range(S):
        eb = draws["eps_b"][s].cpu() if "eps_b" in draws else None
        if flip:
            o = O.flipout_fwd_ref(xc, p["mu_w"], p["rho_w"], draws["eps_w"][s].cpu(), draws["sign_in"][s].cpu(), draws["sign_out"][s].cpu(),
                                  p["mu_b"], p["rho_b"], eb, conv)
        else:
            o = O.reparam_fwd_ref(xc, p["mu_w"], p["rho_w"], draws["eps_w"][s].cpu(), p["mu_b"], p["rho_b"], eb, conv)
        outs.append(o)
    out = torch.cat(outs)
    loss = (out * gout.cpu()).sum()
    if with_kl:
        loss = loss + 3.0 * O.kl_layer_ref(p["mu_w"], p["rho_w"], priors[0], priors[1], p["mu_b"], p["rho_b"], priors[2], priors[3])
    loss.backward()
    return out.detach(), xc.grad, {k: (v.grad if v is not None else None) for k, v in p.items()}


CASES = [
    ("LinearReparameterization", dict(in_features=40, out_features=12), (6, 40), None),
    ("Conv2dReparameterization", dict(in_channels=64, out_channels=72, kernel_size=3, padding=1, prior_type="normal"), (5, 64, 8, 8), None),   # split-kernel forward, multi-tile backward
    ("Conv2dReparameterization", dict(in_channels=6, out_channels=10, kernel_size=(3, 2), stride=(2, 1), padding=(1, 0), dilation=(1, 2), groups=2, prior_type="normal"), (3, 6, 9, 8), None),
    ("Conv2dFlipout", dict(in_channels=16, out_channels=24, kernel_size=3, stride=2, padding=1, groups=2), (4, 16, 7, 7), None),
    ("LinearFlipout", dict(in_features=130, out_features=70, bias=False), (9, 130), None),
    ("LinearFlipout", dict(in_features=36, out_features=9), (5, 36), None),
    ("Conv2dReparameterization", dict(in_channels=8, out_channels=12, kernel_size=3, padding=1, prior_type="normal"), (3, 8, 6, 6), None),
    ("Conv2dReparameterization", dict(in_channels=4, out_channels=6, kernel_size=3, stride=2, padding=1, bias=False, prior_type="normal"), (2, 4, 7, 7), None),
    ("Conv2dFlipout", dict(in_channels=8, out_channels=8, kernel_size=3, padding=1), (2, 8, 4, 4), None),
]


@pytest.mark.parametrize("cls,ctor,xshape,_", CASES)
@pytest.mark.parametrize("S", [1, 2])
def test_layer_gradients_match_oracle_autograd(cls, ctor, xshape, _, S):
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd import mc, rng
    rng.set_mode("philox")
    rng.manual_seed(11)
    torch.manual_seed(3)
    layer = getattr(L, cls)(**ctor).cuda()
    flip = "Flipout" in cls
    x = torch.randn(*xshape).cuda().requires_grad_(True)
    if S == 1:
        out, kl = layer(x)
    else:
        with mc.mc_samples(S, xshape[0]):
            out, kl = layer(x)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).cuda()
    loss = (out * gout).sum() + 3.0 * kl
    loss.backward()
    draws = layer.materialize_last_draw()
    wn = layer._wname
    params = dict(mu_w=getattr(layer, "mu_" + wn), rho_w=getattr(layer, "rho_" + wn), mu_b=layer.mu_bias, rho_b=layer.rho_bias)
    conv = layer._conv_desc() if layer._kind == "conv" else None
    priors = [t.cpu() if t is not None else None for t in (layer.prior_weight_mu, layer.prior_weight_sigma, layer.prior_bias_mu, layer.prior_bias_sigma)]
    o_ref, gx_ref, gp_ref = _oracle_loss(layer._kind, flip, params, x, draws, conv, gout, True, priors)
    assert_close(out.detach().cpu(), o_ref, 1e-4, 1e-5, "out")
    assert_close(x.grad.cpu(), gx_ref, 1e-4, 1e-5, "dL/dx")
    for k, v in params.items():
        if v is not None:
            assert_close(v.grad.cpu(), gp_ref[k], 2e-4, 2e-5, "dL/d" + k)


def test_training_step_reduces_loss():
    """A few SGD steps on a converted MLP: the ELBO-style loss (reference README.md:121-127) goes down."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    rng.set_mode("philox")
    torch.manual_seed(0)
    net = H.mlp((32, 64, 4))
    dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
                     "type": "Reparameterization", "moped_enable": False, "moped_delta": 0.5})
    net = net.cuda().train()
    x = torch.randn(64, 32).cuda()
    y = (x[:, :4].argmax(1)).cuda()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(net(x), y) + get_kl_loss(net) / 64
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] - 0.2, losses


def test_hip_backward_equals_the_materialising_checker():
    """The HIP backward kernels against round 1's ATen path (draws materialised, convolution backward per sample) on a
    ResNet18-w8 training step: same parameter gradients within the layer tolerance."""
    import bayesian_torch_amd.autograd as AG
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    rng.set_mode("philox")
    torch.manual_seed(0)
    net = H.resnet18(10, 8)
    dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "type": "Reparameterization",
                     "moped_enable": False, "moped_delta": 0.5})
    net = net.cuda().train()
    x = torch.randn(6, 3, 32, 32).cuda()
    y = torch.randint(0, 10, (6,)).cuda()
    grads = {}
    for impl in ("hip", "aten"):
        AG.BACKWARD_IMPL = impl
        try:
            rng.manual_seed(5)
            net.zero_grad()
            loss = torch.nn.functional.cross_entropy(net(x), y) + get_kl_loss(net) / 6
            loss.backward()
            grads[impl] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            AG.BACKWARD_IMPL = "hip"
    assert grads["hip"].keys() == grads["aten"].keys() and len(grads["hip"]) > 40
    for n in grads["hip"]:
        assert_close(grads["hip"][n].cpu(), grads["aten"][n].cpu(), 2e-4, 2e-5, n)


@pytest.mark.parametrize("fused", [False, True])
def test_train_graph_replays_the_eager_training_loop(fused):
    """mc.TrainGraph (one training step in a HIP graph; forward and backward kernels add the device word call_base to their call
    coordinate) against the same loop run eagerly from the same RNG position: same parameters after the same number of steps,
    bit for bit -- the replayed backward regenerates the draws of ITS forward, and every kernel sums in a fixed order.
    fused=True (the default wiring): the KL terms come out of the forward kernels and are differentiated inside the weight-gradient
    passes (bt_conv2d_bwd_kl), which run on a side stream beside the data-gradient chain -- the PARAMETERS still match the plain
    autograd loop bit for bit (the same gradient bits: contraction path + KL path added in one kernel instead of by AccumulateGrad);
    the loss VALUE adds the per-layer KL means in a different order (21 fp32 values stacked and summed instead of one kernel's
    sum), so it is held to 1e-6 relative."""
    import copy
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import TrainGraph
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    rng.set_mode("philox")
    torch.manual_seed(0)
    net = H.resnet18(10, width=8)
    dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
                     "type": "Reparameterization", "moped_enable": False, "moped_delta": 0.5})
    net = net.cuda().train()
    ref = copy.deepcopy(net)
    x = torch.randn(16, 3, 32, 32).cuda()
    y = torch.randint(0, 10, (16,)).cuda()
    loss_fn = lambda m, out, yy: torch.nn.functional.cross_entropy(out, yy) + get_kl_loss(m) / 16
    n_warm, n_steps = 3, 4

    rng.manual_seed(11)
    opt_r = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9)
    eager_losses = []
    for _ in range(n_warm + n_steps):
        opt_r.zero_grad(set_to_none=True)
        loss = loss_fn(ref, ref(x), y)
        loss.backward()
        opt_r.step()
        eager_losses.append(float(loss))

    rng.manual_seed(11)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
    tg = TrainGraph(net, opt, loss_fn, x, y, warmup=n_warm, fused=fused)        # n_warm real steps, eager
    graph_losses = [float(tg.step()) for _ in range(n_steps)]
    if fused:
        for a, b in zip(graph_losses, eager_losses[n_warm:]):
            assert abs(a - b) <= 1e-6 * abs(b), (graph_losses, eager_losses)
    else:
        assert graph_losses == eager_losses[n_warm:], (graph_losses, eager_losses)
    for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    assert len(set(graph_losses)) == n_steps       # fresh draws every replay


def test_random_geometries_hip_backward_equals_the_checker():
    """40 random conv layers (strides, dilations, groups, stems, Flipout, S in {1, 2, 3}): dx, dmu, drho of the HIP dgrad / wgrad
    kernels -- tap skipping, output-channel pieces, reduction-axis chunks -- against the materialising ATen path on the same draws."""
    import random
    import bayesian_torch_amd.autograd as AG
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd import mc, rng
    rng.set_mode("philox")
    r = random.Random(7)
    for case in range(40):
        flip = r.random() < 0.4
        groups = r.choice([1, 1, 2])
        Ci = groups * r.choice([1, 3, 4, 8, 16, 24, 40, 64])
        Co = groups * r.choice([4, 8, 20, 32, 64, 96])
        kh, kw = r.randint(1, 3), r.randint(1, 3)
        st, dl = (r.choice([1, 1, 2]), r.choice([1, 1, 2])), (r.choice([1, 1, 2]), r.choice([1, 1, 2]))
        pd = (r.randint(0, (kh - 1) * dl[0]), r.randint(0, (kw - 1) * dl[1]))
        H, W = r.choice([1, 2, 4, 5, 8, 12, 16]), r.choice([1, 2, 4, 5, 8, 12, 16])
        H, W = max(H, (kh - 1) * dl[0] + 1 - 2 * pd[0], 1), max(W, (kw - 1) * dl[1] + 1 - 2 * pd[1], 1)
        B, S = r.choice([1, 2, 5, 16, 48]), r.choice([1, 2, 3])
        cls = L.Conv2dFlipout if flip else L.Conv2dReparameterization
        kwargs = dict(in_channels=Ci, out_channels=Co, kernel_size=(kh, kw), stride=st, padding=pd, dilation=dl, groups=groups, bias=r.random() < 0.5)
        if not flip:
            kwargs["prior_type"] = "normal"
        torch.manual_seed(case)
        layer = cls(**kwargs).cuda()
        x0 = torch.randn(B, Ci, H, W).cuda()
        grads = {}
        for impl in ("hip", "aten"):
            AG.BACKWARD_IMPL = impl
            try:
                rng.manual_seed(100 + case)
                layer.zero_grad()
                x = x0.clone().requires_grad_(True)
                with mc.mc_samples(S, B):
                    out, kl = layer(x)
                gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(case)).cuda()
                ((out * gout).sum() + 2.0 * kl).backward()
                grads[impl] = [x.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None]
            finally:
                AG.BACKWARD_IMPL = "hip"
        assert len(grads["hip"]) == len(grads["aten"]) >= 3
        for i, (a, b) in enumerate(zip(grads["hip"], grads["aten"])):
            assert torch.isfinite(a).all(), (case, kwargs, i)
            assert_close(a.cpu(), b.cpu(), 2e-4, 2e-5, f"case {case} {kwargs} H{H} W{W} B{B} S{S} grad {i}")
