"""GPU parity added in round 2: cfg5 at its own size (ResNet50 golden + full-size properties), Laplace-prior KL,
LSTMFlipout, full-size layer1 / layer4 launches of the fast kernel replayed through the plain-C oracle, the parameter-pack
cache, position-derived layer ids and the Flipout sign stream's correlation structure."""
import copy

import pytest
import torch

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "moped_enable": False, "moped_delta": 0.5}
RTOL, ATOL = 1e-4, 1e-5


# ------------------------------------------------------------------------------------------------ cfg5 at its own size
def test_resnet50_full_width_matches_reference_golden():
    """cfg5's model (ResNet50 width 64, 1000 classes, 3x224x224 inputs) at the STATED tolerance: the reference's per-sample
    draws are replayed through the oracle modules and injected into the HIP layers (54 Bayesian layers deep)."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    g = load_golden("model_r50_reparam")
    meta = g["meta"]
    torch.manual_seed(meta["seed"])
    ref = H.resnet50(1000, 64)
    O.ref_dnn_to_bnn(ref, "Reparameterization")
    H.fill_bayes_params(ref, meta["seed"])
    ref.eval()
    net = H.resnet50(1000, 64)
    dnn_to_bnn(net, dict(PRIOR, type="Reparameterization"))
    H.fill_bayes_params(net, meta["seed"])
    net = net.cuda().eval()
    x = torch.randn(*meta["x_shape"], generator=torch.Generator().manual_seed(meta["seed"] + 7))
    with torch.no_grad():
        torch.manual_seed(meta["seed"] * 100)
        want = ref(x)
    assert_close(want, g["logits"][0], RTOL, ATOL, "r50: oracle vs golden")
    for (_, rm), (_, m) in zip(H.bayes_layers(ref), H.bayes_layers(net)):
        m.inject_draw = dict(eps_w=getattr(rm, "eps_" + rm._wn).unsqueeze(0).cuda(),
                             eps_b=rm.eps_bias.unsqueeze(0).cuda() if rm.mu_bias is not None else None)
    logits, kl = mc_forward(net, x.cuda(), 1)
    assert_close(logits[0].cpu(), g["logits"][0], RTOL, ATOL, "r50: HIP vs golden")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, "r50: fused KL vs golden")
    assert_close(get_kl_loss(net).cpu(), g["kl"], 1e-5, 0, "r50: get_kl_loss vs golden")
    for _, m in H.bayes_layers(net):
        m.inject_draw = None


def test_full_size_cfg5_properties():
    """BASELINE cfg5 at its full per-GPU size (ResNet50, batch 256, 3x224x224, fused output stages -- the kernels
    `bench.py --workload cfg5` times): the size-independent properties of test_full_size_cfg3_properties.
    (a) launch-split independence of the MC samples, bit for bit; (b) determinism / fresh draws per call; (c) fused KL ==
    get_kl_loss and independent of S; (d) fused output stages == the module-by-module model on the same draws; (e) batch-row
    independence (a 64-image batch takes other tile geometries)."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    torch.manual_seed(3)
    net = H.resnet50(1000, 64)
    dnn_to_bnn(net, dict(PRIOR, type="Reparameterization"))
    H.fill_bayes_params(net, 3)
    net = net.cuda().eval()
    x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    rng.set_mode("philox")
    rng.manual_seed(321)
    c0 = rng.peek_call()

    def run(model, S, sample0, xx=x):
        rng.set_call(c0)
        out, kl = mc_forward(model, xx, S, sample0=sample0, with_kl=True)
        return out.cpu(), kl

    plain, kl_plain = run(net, 1, 6)                        # unfused: BN / ReLU / add / MaxPool as torch modules
    H.fuse_inference(net)
    two, kl2 = run(net, 2, 5)
    assert torch.isfinite(two).all() and two.shape == (2, 256, 1000)
    assert_close(two[1], plain[0], RTOL, ATOL, "(d) fused vs unfused model, same global sample")
    assert abs(float(kl2) - float(kl_plain)) <= 1e-6 * abs(float(kl_plain))
    ones = torch.cat([run(net, 1, 5)[0], run(net, 1, 6)[0]])
    assert torch.equal(ones, two), "(a) launch split"
    again, _ = run(net, 2, 5)
    assert torch.equal(again, two), "(b) determinism"
    later, _ = mc_forward(net, x, 1, sample0=5, with_kl=False)      # the call counter has advanced
    assert not torch.equal(later.cpu()[0], two[0])
    assert float(run(net, 1, 0)[1]) == float(kl2), "(c) KL independent of S"
    ref_kl = get_kl_loss(net).detach()
    assert abs(float(kl2) - float(ref_kl)) <= 1e-5 * abs(float(ref_kl))
    part, _ = run(net, 1, 5, x[:64].contiguous())
    assert_close(part[0], two[0, :64], RTOL, ATOL, "(e) batch rows are independent of the tile geometry")


# ------------------------------------------------------------------------ full-size fast-kernel launches vs the C oracle
@pytest.mark.parametrize("tag,Ci,Co,H,flip", [("layer1", 64, 64, 8, False), ("layer4", 512, 512, 1, False), ("layer1-flipout", 64, 64, 8, True)])
def test_full_size_fast_kernel_replay_through_c_oracle(tag, Ci, Co, H, flip):
    """One ResNet18/CIFAR layer1 conv (64->64 3x3 on 8x8, K = 576) and one layer4 conv (512->512 3x3 on 1x1 maps,
    K = 4608) at b128 with the packed parameters -- the very kernel instances bench.py times -- on-chip draws materialised
    with bt_rng_*_fill and replayed through the independent plain-C oracle (fp64 accumulation)."""
    from oracle import c_oracle as CO
    from bayesian_torch_amd import functional as F
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(17)
    mu = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    rho = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1 - 3
    S, B = 2, 128
    x = torch.randn(S * B, Ci, H, H, generator=g)
    conv = dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1)
    seed, call, lid, s0 = 99, 3, 11, 40
    out, _ = F.fused_forward(x.to(dev), mu.to(dev), rho.to(dev), flip=flip, conv=conv, S=S, shared_x=False, seed=seed, call=call,
                             layer_id=lid, sample0=s0, packed=F.pack_params(mu.to(dev), rho.to(dev)))
    out = out.reshape(S, B, Co, H, H).cpu()
    eps = F.rng_fill_normal(seed, call, lid, s0, 0, S, mu.shape, dev).cpu()
    if flip:
        s_in = F.rng_fill_sign(seed, call, lid, s0, 2, S, (B, Ci, H, H), dev).cpu()
        s_out = F.rng_fill_sign(seed, call, lid, s0, 3, S, (B, Co, H, H), dev).cpu()
    for s in range(S):
        xs = x[s * B:(s + 1) * B]
        if flip:
            ref = CO.flipout_fwd(xs, mu, rho, eps[s], s_in[s], s_out[s], None, None, None, conv)
        else:
            ref = CO.reparam_fwd(xs, mu, rho, eps[s], None, None, None, conv)
        assert_close(out[s], ref, RTOL, ATOL, f"{tag} sample {s} vs C oracle")


# ------------------------------------------------------------------------------------------------ Laplace prior (f3)
def test_laplace_kl_matches_reference_golden():
    """prior_type='laplace' (base_variational_layer.py:74-97): standalone kernel, kl_div(), and the layers' forward / kl_loss."""
    from bayesian_torch_amd import _lib
    import bayesian_torch_amd.layers as L
    g = load_golden("kl_laplace")
    for tag in g["meta"]["cases"]:
        seg = tuple(g[f"{tag}_{k}"].cuda() for k in ("mu", "rho", "pmu", "psig"))
        assert_close(_lib.kl_normal([seg], laplace=True).cpu(), g[tag + "_kl"], 1e-5, 0, "laplace " + tag)
    base = L.LinearReparameterization(2, 2).cuda()
    sig = torch.log1p(torch.exp(g["a_rho"])).cuda()
    assert_close(base.kl_div(g["a_mu"].cuda(), sig, g["a_pmu"].cuda(), g["a_psig"].cuda(), "laplace").cpu(), g["a_kl"], 1e-5, 0, "kl_div laplace")
    with pytest.raises(ValueError, match="Unknown prior_type"):
        base.kl_div(g["a_mu"].cuda(), sig, 0.0, 1.0, "xyz")
    lin = L.LinearReparameterization(37, 11, prior_type="laplace").cuda()
    conv = L.Conv2dReparameterization(5, 6, 3, prior_type="laplace").cuda()
    with torch.no_grad():
        lin.mu_weight.copy_(g["lin_mu_w"]), lin.rho_weight.copy_(g["lin_rho_w"]), lin.mu_bias.copy_(g["lin_mu_b"]), lin.rho_bias.copy_(g["lin_rho_b"])
        conv.mu_kernel.copy_(g["conv_mu_w"]), conv.rho_kernel.copy_(g["conv_rho_w"]), conv.mu_bias.copy_(g["conv_mu_b"]), conv.rho_bias.copy_(g["conv_rho_b"])
        _, kl_l = lin(torch.randn(3, 37).cuda())
        _, kl_c = conv(torch.randn(2, 5, 6, 6).cuda())
    assert_close(kl_l.cpu(), g["lin_kl"], 1e-5, 0, "Linear laplace forward kl")
    assert_close(lin.kl_loss().cpu(), g["lin_kl"], 1e-5, 0, "Linear laplace kl_loss")
    assert_close(kl_c.cpu(), g["conv_kl"], 1e-5, 0, "Conv2d laplace forward kl")
    # gradient of the differentiable KL (training path) against torch autograd of the oracle's formula
    from oracle import bt_oracle as O
    lin.train()
    kl = lin.kl_loss()
    kl.backward()
    mu = g["lin_mu_w"].clone().requires_grad_(True)
    rho = g["lin_rho_w"].clone().requires_grad_(True)
    mb, rb = g["lin_mu_b"].clone().requires_grad_(True), g["lin_rho_b"].clone().requires_grad_(True)
    (O.kl_laplace_ref(mu, O.softplus_ref(rho)) + O.kl_laplace_ref(mb, O.softplus_ref(rb))).backward()
    assert_close(lin.mu_weight.grad.cpu(), mu.grad, 1e-4, 1e-5, "d kl / d mu")
    assert_close(lin.rho_weight.grad.cpu(), rho.grad, 1e-4, 1e-5, "d kl / d rho")


# ------------------------------------------------------------------------------------------------ LSTMFlipout (f4)
def test_lstm_flipout_matches_reference_golden():
    import bayesian_torch_amd.layers as L
    g = load_golden("lstm_flipout_7x5")
    lstm = L.LSTMFlipout(7, 5).cuda()
    T = g["meta"]["x_shape"][1]
    with torch.no_grad():
        for nm in ("ih", "hh"):
            lin = getattr(lstm, nm)
            lin.mu_weight.copy_(g[nm + "_mu_w"]), lin.rho_weight.copy_(g[nm + "_rho_w"]), lin.mu_bias.copy_(g[nm + "_mu_b"]), lin.rho_bias.copy_(g[nm + "_rho_b"])
            lin.inject_draw = [dict(eps_w=g[nm + "_eps_w"][t:t + 1].cuda(), eps_b=g[nm + "_eps_b"][t:t + 1].cuda(),
                                    sign_in=g[nm + "_sign_in"][t:t + 1].cuda(), sign_out=g[nm + "_sign_out"][t:t + 1].cuda()) for t in range(T)]
        hs, (h2, cs), kl = lstm(g["x"].cuda())
    assert_close(hs.cpu(), g["hidden_seq"], RTOL, ATOL, "hidden_seq")
    assert_close(cs.cpu(), g["c_ts"], RTOL, ATOL, "c_ts")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, "kl")
    assert_close(lstm.kl_loss().cpu(), g["kl_loss"], 1e-5, 0, "kl_loss")


# ------------------------------------------------------------------------------------------------ host-side robustness
def test_pack_cache_follows_parameter_updates():
    """writes through .data do not bump ._version. Training mode rebuilds the pack per call; in eval mode
    load_state_dict / init_parameters / MOPED / invalidate_pack() drop it. Checked against the oracle after each update."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import rng
    import bayesian_torch_amd.layers as L
    torch.manual_seed(0)
    conv = L.Conv2dReparameterization(8, 16, 3, padding=1, bias=False).cuda()
    x = torch.randn(4, 8, 6, 6).cuda()
    cd = dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1)
    rng.set_mode("philox")

    def check(what):
        with torch.no_grad():
            out = conv(x, return_kl=False)
        d = conv.materialize_last_draw()
        ref = O.reparam_fwd_ref(x.cpu(), conv.mu_kernel.detach().cpu(), conv.rho_kernel.detach().cpu(), d["eps_w"][0].cpu(), conv=cd)
        assert_close(out.cpu(), ref, RTOL, ATOL, what)

    conv.eval()
    check("first forward")
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    sd["mu_kernel"] = sd["mu_kernel"] * -2.0
    conv.load_state_dict(sd)
    check("after load_state_dict")
    conv.mu_kernel.data.mul_(0.5)
    conv.invalidate_pack()
    check("after .data update + invalidate_pack()")
    conv.init_parameters()
    check("after init_parameters()")
    with torch.no_grad():
        conv.mu_kernel.add_(1.0)            # autograd-visible in-place update: version counter
    check("after an in-place update under no_grad")
    conv.train()
    out0 = conv(x, return_kl=False)         # grad enabled + training: the pack is rebuilt on every call
    conv.mu_kernel.data.copy_(torch.randn_like(conv.mu_kernel))
    out1 = conv(x, return_kl=False)
    d = conv.materialize_last_draw()
    ref = O.reparam_fwd_ref(x.cpu(), conv.mu_kernel.detach().cpu(), conv.rho_kernel.detach().cpu(), d["eps_w"][0].cpu(), conv=cd)
    assert_close(out1.detach().cpu(), ref, RTOL, ATOL, "training-mode forward after a .data write")


def test_layer_ids_follow_model_position():
    """dnn_to_bnn numbers the Bayesian layers by position: two builds of the same model (whatever else the process
    constructed in between) and a deepcopy draw identically for the same (seed, call)."""
    from bayesian_torch_amd import rng
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn

    def build():
        net = H.resnet18(10, 8)
        dnn_to_bnn(net, dict(PRIOR, type="Flipout"))
        H.fill_bayes_params(net, 2)
        return net.cuda().eval()
    a = build()
    [L.LinearReparameterization(3, 3) for _ in range(5)]      # unrelated constructions in between
    b = build()
    assert [m._layer_id for _, m in H.bayes_layers(a)] == list(range(1, 22)) == [m._layer_id for _, m in H.bayes_layers(b)]
    x = torch.randn(4, 3, 32, 32).cuda()
    rng.set_mode("philox")
    rng.manual_seed(8)
    c0 = rng.peek_call()
    outs = []
    for net in (a, b, copy.deepcopy(a)):
        rng.set_call(c0)
        outs.append(mc_forward(net, x, 2)[0])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_non_current_device_is_guarded():
    """launches follow the tensors' device, not the current one."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    import bayesian_torch_amd.layers as L
    torch.manual_seed(0)
    lin = L.LinearReparameterization(64, 8).to("cuda:1").eval()
    x = torch.randn(5, 64, device="cuda:1")
    with torch.cuda.device(0), torch.no_grad():
        out, kl = lin(x)
    assert out.device == x.device and torch.isfinite(out).all() and torch.isfinite(kl)


# ------------------------------------------------------------------------------------------------ Flipout sign stream
def test_sign_stream_correlation_structure():
    """The Flipout signs are a Philox-keyed integer hash of the element index. Evidence beyond mean / lag-1: correlations at
    the strides a convolution actually sums over (W, H*W, C*H*W of the CIFAR and ImageNet shapes), between sign_in and
    sign_out of one layer, between neighbouring layer ids, sample ids and calls, and a chi-square on 8-bit windows."""
    from bayesian_torch_amd import functional as F
    dev = torch.device("cuda")
    n = 1 << 22
    tol = 5.0 / (n ** 0.5)                  # 5 sigma of a fair-coin correlation estimate

    def stream(call=0, lid=7, s0=0, tensor=2, S=1):
        return F.rng_fill_sign(20240, call, lid, s0, tensor, S, (n,), dev)
    s = stream()[0]
    assert bool(((s == 1) | (s == -1)).all()) and abs(float(s.mean())) < tol
    for lag in (1, 2, 3, 4, 7, 8, 16, 32, 56, 64, 112, 224, 64 * 8 * 8, 8 * 8, 4 * 4, 2 * 2, 56 * 56, 112 * 112, 224 * 224,
                64 * 56 * 56, 3 * 224 * 224, 64 * 32 * 32, 1 << 20, (1 << 20) + 1):
        if lag < n:
            assert abs(float((s[lag:] * s[:-lag]).mean())) < tol * 1.2, f"lag {lag}"
    others = dict(sign_out=stream(tensor=3)[0], next_layer=stream(lid=8)[0], next_sample=stream(s0=1)[0], next_call=stream(call=1)[0],
                  next_sample_same_launch=stream(S=2)[1])
    assert torch.equal(others["next_sample"], others["next_sample_same_launch"])
    for k, o in others.items():
        assert abs(float((s * o).mean())) < tol, k
        assert abs(float((s[1:] * o[:-1]).mean())) < tol * 1.2, k + " lag 1"
    # chi-square on non-overlapping 8-bit windows: 256 cells, expected n/8/256 each
    bits = (s > 0).to(torch.int64).reshape(-1, 8)
    idx = (bits * (2 ** torch.arange(8, device=dev))).sum(1)
    cnt = torch.bincount(idx, minlength=256).double()
    exp = cnt.sum() / 256
    chi2 = float(((cnt - exp) ** 2 / exp).sum())
    assert 170 < chi2 < 350, chi2           # 255 degrees of freedom: mean 255, sd 22.6 -> roughly +-4 sd
    # and per-bit balance inside the windows
    assert float((bits.double().mean(0) - 0.5).abs().max()) < 5.0 / ((n / 8) ** 0.5) / 2 * 1.5
