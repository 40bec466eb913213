"""GPU, model level: dnn_to_bnn-converted ResNet18 / MLP through the drop-in modules vs the golden logits the
reference produced (tests/golden/model_*.npz) -- the per-sample CPU draws are replayed through the oracle modules
and injected into the HIP layers -- plus MC batching, the sequential loop, get_kl_loss and mc_predict."""
import pytest
import torch

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "moped_enable": False, "moped_delta": 0.5}


def _nets(name, meta):
    from oracle import bt_oracle as O
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    mk = (lambda: H.mlp((3072, 512, 10))) if len(meta["x_shape"]) == 2 else (lambda: H.resnet18(10, 8 if "w8" in name else 64))
    torch.manual_seed(meta["seed"])
    ref = mk()
    O.ref_dnn_to_bnn(ref, meta["btype"])
    H.fill_bayes_params(ref, meta["seed"])
    net = mk()
    dnn_to_bnn(net, dict(PRIOR, type=meta["btype"]))
    H.fill_bayes_params(net, meta["seed"])
    return ref.eval(), net.cuda().eval()


def _replay_reference_draws(ref, x, meta):
    """Run the oracle modules with the golden's per-sample seeds; collect logits and every layer's draws."""
    from bayesian_torch_amd.harness import resnet as H
    layers = [m for _, m in H.bayes_layers(ref)]
    logits, draws = [], [dict(eps_w=[], eps_b=[], sign_in=[], sign_out=[]) for _ in layers]
    with torch.no_grad():
        for s in range(meta["S"]):
            torch.manual_seed(meta["seed"] * 100 + s)
            logits.append(ref(x))
            for m, d in zip(layers, draws):
                d["eps_w"].append(getattr(m, "eps_" + m._wn).clone())
                if m.mu_bias is not None:
                    d["eps_b"].append(m.eps_bias.clone())
                if m.flip:
                    d["sign_in"].append(m.last["sign_in"])
                    d["sign_out"].append(m.last["sign_out"])
    stack = lambda lst: torch.stack(lst).cuda() if lst else None
    return torch.stack(logits), [{k: stack(v) for k, v in d.items()} for d in draws]


@pytest.mark.parametrize("name", ["model_r18w8_reparam", "model_r18w8_flipout", "model_mlp_reparam", "model_r18_reparam", "model_r18_flipout"])
def test_model_matches_reference_golden(name):
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import get_kl_loss
    g = load_golden(name)
    meta = g["meta"]
    ref, net = _nets(name, meta)
    x = torch.randn(*meta["x_shape"], generator=torch.Generator().manual_seed(meta["seed"] + 7))
    ref_logits, draws = _replay_reference_draws(ref, x, meta)
    assert_close(ref_logits, g["logits"], 1e-4, 1e-5, name + ": oracle vs golden")       # the box's CPU reproduces the golden
    layers = [m for _, m in H.bayes_layers(net)]
    S = meta["S"]
    for m, d in zip(layers, draws):
        m.inject_draw = d
    logits, kl = mc_forward(net, x.cuda(), S)
    assert_close(logits.cpu(), g["logits"], 1e-4, 1e-5, name + ": HIP (MC-batched) vs golden")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, name + ": fused KL vs golden")
    assert_close(get_kl_loss(net).cpu(), g["kl"], 1e-5, 0, name + ": get_kl_loss vs golden")
    # the reference's sequential loop (one model(x) per sample) gives the same numbers
    with torch.no_grad():
        for s in range(S):
            for m, d in zip(layers, draws):
                m.inject_draw = {k: (v[s:s + 1] if v is not None else None) for k, v in d.items()}
            assert_close(net(x.cuda()).cpu(), g["logits"][s], 1e-4, 1e-5, f"{name}: sequential sample {s}")
    for m in layers:
        m.inject_draw = None


@pytest.mark.parametrize("btype", ["Reparameterization", "Flipout"])
def test_philox_model_replay_and_world_size_independence(btype):
    """On-chip draws at model level: replay through the oracle; and the same global samples computed as one
    launch of 4 or two launches of 2 (what two ranks would do) agree bit for bit."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    meta = dict(x_shape=[6, 3, 32, 32], seed=9, btype=btype, S=4)
    ref, net = _nets("w8", meta)
    x = torch.randn(6, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    rng.set_mode("philox")
    rng.manual_seed(77)
    c0 = rng.peek_call()
    logits, kl = mc_forward(net, x.cuda(), 4, sample0=8)
    layers = [m for _, m in H.bayes_layers(net)]
    draws = [m.materialize_last_draw() for m in layers]
    with torch.no_grad():
        for s in range(4):
            for (_, rm), d in zip(H.bayes_layers(ref), draws):
                rm.inject = {k: v[s].cpu() for k, v in d.items()}
            assert_close(logits[s].cpu(), ref(x), 1e-4, 1e-5, f"philox {btype} sample {s}")
    assert_close(kl.cpu(), O.ref_get_kl_loss(ref), 1e-5, 0, "kl")
    rng.set_call(c0)
    a, _ = mc_forward(net, x.cuda(), 2, sample0=8)
    rng.set_call(c0)
    b, _ = mc_forward(net, x.cuda(), 2, sample0=10)
    assert torch.equal(torch.cat([a, b]), logits)
    # and a later call draws something else
    c, _ = mc_forward(net, x.cuda(), 4, sample0=8)
    assert not torch.equal(c, logits)


def test_torch_mode_fills_eps_buffers_like_reference():
    """rng mode 'torch': eps drawn by torch's device generator into the layer buffers (reference semantics),
    kernel reads them; output equals the oracle on the buffers' content."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import rng
    import bayesian_torch_amd.layers as L
    rng.set_mode("torch")
    try:
        torch.manual_seed(5)
        conv = L.Conv2dReparameterization(5, 7, 3, padding=1, prior_type="normal").cuda()
        x = torch.randn(3, 5, 6, 6).cuda()
        out, kl = conv(x)
        ref = O.reparam_fwd_ref(x.cpu(), conv.mu_kernel.detach().cpu(), conv.rho_kernel.detach().cpu(), conv.eps_kernel.cpu(),
                                conv.mu_bias.detach().cpu(), conv.rho_bias.detach().cpu(), conv.eps_bias.cpu(),
                                dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1))
        assert_close(out.cpu(), ref, 1e-4, 1e-5, "torch-mode conv")
        assert_close(kl.cpu(), conv.kl_loss().cpu(), 1e-6, 0, "kl == kl_loss()")
        lin = L.LinearFlipout(12, 5).cuda()
        y = torch.randn(2, 3, 12).cuda()          # extra leading dims, like F.linear
        o, k = lin(y)
        d = lin.materialize_last_draw()
        ref = O.flipout_fwd_ref(y.reshape(6, 12).cpu(), lin.mu_weight.detach().cpu(), lin.rho_weight.detach().cpu(), lin.eps_weight.cpu(),
                                d["sign_in"][0].cpu(), d["sign_out"][0].cpu(), lin.mu_bias.detach().cpu(), lin.rho_bias.detach().cpu(), lin.eps_bias.cpu())
        assert o.shape == (2, 3, 5)
        assert_close(o.reshape(6, 5).cpu(), ref, 1e-4, 1e-5, "torch-mode linear flipout")
    finally:
        rng.set_mode("philox")


def test_mc_predict_single_process():
    from oracle import bt_oracle as O
    from bayesian_torch_amd import mc_dist, rng
    from bayesian_torch_amd.mc import mc_forward
    meta = dict(x_shape=[5, 3, 32, 32], seed=4, btype="Reparameterization", S=6)
    _, net = _nets("w8", meta)
    x = torch.randn(5, 3, 32, 32).cuda()
    rng.manual_seed(3)
    c0 = rng.peek_call()
    res = mc_dist.mc_predict(net, x, 6)
    rng.set_call(c0)
    logits, kl = mc_forward(net, x, 6)
    p, e, l = O.mc_epilogue_ref(logits.cpu())
    assert_close(res["mean_prob"].cpu(), p / 6, 1e-5, 1e-6, "mean_prob")
    assert_close(res["mean_entropy"].cpu(), e / 6, 1e-5, 1e-6, "entropy")
    assert_close(res["mean_logits"].cpu(), l / 6, 1e-5, 1e-6, "mean logits")
    assert_close(res["kl"].cpu(), kl.cpu(), 1e-6, 0, "kl")


@pytest.mark.parametrize("btype", ["Reparameterization", "Flipout"])
def test_fused_output_stage_equals_unfused_model(btype):
    """fold BN / ReLU / residual add into the conv kernels' epilogue: same logits as the module-by-module model
    (randomised BatchNorm statistics so the affine map is non-trivial), and still the golden's numbers."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.fuse import fold_batchnorm
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    import copy
    meta = dict(x_shape=[6, 3, 32, 32], seed=11, btype=btype, S=3)
    _, net = _nets("w8", meta)
    g = torch.Generator().manual_seed(5)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g).cuda() * 0.3)
            m.running_var.copy_((torch.rand(m.num_features, generator=g) + 0.5).cuda())
            m.weight.data.copy_((torch.rand(m.num_features, generator=g) + 0.5).cuda())
            m.bias.data.copy_(torch.randn(m.num_features, generator=g).cuda() * 0.2)
    x = torch.randn(6, 3, 32, 32).cuda()
    rng.manual_seed(5)
    c0 = rng.peek_call()
    want, kl0 = mc_forward(net, x, 3)
    fused = H.fuse_inference(copy.deepcopy(net))
    for (_, a), (_, b) in zip(H.bayes_layers(net), H.bayes_layers(fused)):
        b._layer_id = a._layer_id                  # same RNG coordinates as the unfused model
    rng.set_call(c0)
    got, kl1 = mc_forward(fused, x, 3)
    assert_close(got.cpu(), want.cpu(), 1e-5, 1e-6, "fused output stage vs unfused")
    assert torch.equal(kl0, kl1)
    folded = copy.deepcopy(net)
    assert fold_batchnorm(folded) == 20
    for (_, a), (_, b) in zip(H.bayes_layers(net), H.bayes_layers(folded)):
        b._layer_id = a._layer_id
    rng.set_call(c0)
    got2, _ = mc_forward(folded, x, 3)
    assert_close(got2.cpu(), want.cpu(), 1e-5, 1e-6, "fold_batchnorm vs unfused")


def test_mc_graph_replay_matches_eager_and_draws_fresh_samples():
    """HIP-graph runner: a replay computes what an eager mc_forward at the same position of the call counter computes
    (device-side call word), consecutive replays differ, and replays / eager calls / a second graph interleave without
    ever reusing a draw coordinate (the host counter advances with every replay)."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.mc import McGraph, mc_forward
    meta = dict(x_shape=[6, 3, 32, 32], seed=13, btype="Reparameterization", S=3)
    _, net = _nets("w8", meta)
    x = torch.randn(6, 3, 32, 32).cuda()
    rng.set_mode("philox")
    rng.manual_seed(21)
    g = McGraph(net, x, 3, sample0=4)
    seen = []
    for r in range(3):
        c = rng.peek_call()
        l, kl, _ = g.replay()
        l, kl = l.clone(), kl.clone()
        assert rng.peek_call() == c + g.calls_per_run
        rng.set_call(c)
        e, ekl = mc_forward(net, x, 3, sample0=4)          # eager at the same coordinates
        assert torch.equal(e, l)
        assert_close(ekl.cpu(), kl.cpu(), 1e-6, 0, "kl")
        assert all(not torch.equal(l, p) for p in seen)
        seen.append(l)
    e, _ = mc_forward(net, x, 3, sample0=4)                # eager after the replays: fresh coordinates
    assert all(not torch.equal(e, p) for p in seen)
    seen.append(e.clone())
    g2 = McGraph(net, x, 3, sample0=4)                     # a second graph of the same model, interleaved with the first
    for gr in (g2, g, g2, g):
        l = gr.replay()[0].clone()
        assert all(not torch.equal(l, p) for p in seen)
        seen.append(l)


@pytest.mark.parametrize("btype", ["Reparameterization", "Flipout"])
def test_resnet50_topology_cfg5_shape_family(btype):
    """cfg5's model family (Bottleneck ResNet50: 1x1 / 3x3-stride-2 / 7x7 stem, ImageNet-style 64x64 inputs here) at small
    width: on-chip draws replayed through the oracle, MC-batched."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    torch.manual_seed(2)
    ref = H.resnet50(10, width=8)
    O.ref_dnn_to_bnn(ref, btype)
    H.fill_bayes_params(ref, 4)
    net = H.resnet50(10, width=8)
    dnn_to_bnn(net, dict(PRIOR, type=btype))
    H.fill_bayes_params(net, 4)
    ref, net = ref.eval(), net.cuda().eval()
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    rng.set_mode("philox")
    rng.manual_seed(5)
    S = 2
    logits, kl = mc_forward(net, x.cuda(), S)
    draws = [m.materialize_last_draw() for _, m in H.bayes_layers(net)]
    with torch.no_grad():
        for s in range(S):
            for (_, rm), d in zip(H.bayes_layers(ref), draws):
                rm.inject = {k: v[s].cpu() for k, v in d.items()}
            assert_close(logits[s].cpu(), ref(x), 2e-4, 2e-5, f"resnet50 {btype} sample {s}")
    assert_close(kl.cpu(), O.ref_get_kl_loss(ref), 1e-5, 0, "kl")


def test_full_size_cfg3_properties():
    """BASELINE cfg3 at its full size (ResNet18 width 64, CIFAR batch 128, fused output stages, the kernels bench.py times):
    size-independent properties instead of an oracle run. (a) MC-sample independence of the launch split: 8 samples in one
    launch == 4 + 4 == 8 x 1 with the same global sample ids, bit for bit; (b) determinism: same (seed, call) twice is
    identical, the next call differs; (c) the fused KL equals the standalone bt_kl_normal launch over the model (get_kl_loss)
    and does not depend on S; (d) the fused model equals the unfused module sequence (BatchNorm / ReLU / add / MaxPool as
    torch modules) on the same draws within the layer tolerance; (e) batch-row independence: the first 32 images alone give
    the same logits for those rows (different tile geometry) within tolerance."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    torch.manual_seed(3)
    net = H.resnet18(10, 64)
    dnn_to_bnn(net, dict(PRIOR, type="Reparameterization"))
    H.fill_bayes_params(net, 3)
    net = net.cuda().eval()
    x = torch.randn(128, 3, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
    rng.set_mode("philox")
    rng.manual_seed(123)
    c0 = rng.peek_call()

    def run(model, S, sample0, xx=x):
        rng.set_call(c0)
        return mc_forward(model, xx, S, sample0=sample0, with_kl=True)

    plain, kl_plain = run(net, 8, 16)                       # unfused: BN / ReLU / add / MaxPool as torch modules
    H.fuse_inference(net)
    full, kl8 = run(net, 8, 16)
    assert torch.isfinite(full).all() and full.shape == (8, 128, 10)
    # (d)
    assert_close(full.cpu(), plain.cpu(), 1e-4, 1e-5, "fused vs unfused model")
    assert abs(float(kl8) - float(kl_plain)) <= 1e-6 * abs(float(kl_plain))
    # (a)
    halves = torch.cat([run(net, 4, 16)[0], run(net, 4, 20)[0]])
    assert torch.equal(halves, full)
    ones = torch.cat([run(net, 1, 16 + s)[0] for s in range(8)])
    assert torch.equal(ones, full)
    # (b)
    again, _ = run(net, 8, 16)
    assert torch.equal(again, full)
    later, _ = mc_forward(net, x, 8, sample0=16, with_kl=True)   # the call counter has advanced
    assert not torch.equal(later, full)
    # (c)
    kl1 = run(net, 1, 0)[1]
    assert float(kl1) == float(kl8)
    ref_kl = get_kl_loss(net).detach()
    assert abs(float(kl8) - float(ref_kl)) <= 1e-5 * abs(float(ref_kl))
    # (e)
    part, _ = run(net, 8, 16, x[:32].contiguous())
    assert_close(part.cpu(), full[:, :32].cpu(), 1e-4, 1e-5, "batch rows are independent of the tile geometry")
