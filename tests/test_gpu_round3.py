"""GPU parity added in round 3: the device-side pack check (a stale pack cannot be read, whatever way the parameters were written),
graph replays that follow parameter updates, and the rest of the layer family inside mc_samples."""
import pytest
import torch

from conftest import assert_close, golden_names, layer_tensors, load_golden

pytestmark = pytest.mark.gpu
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "moped_enable": False, "moped_delta": 0.5}
RTOL, ATOL = 1e-4, 1e-5
CD = dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1)


def _check_conv(conv, x, what):
    from oracle import bt_oracle as O
    with torch.no_grad():
        out = conv(x, return_kl=False)
    d = conv.materialize_last_draw()
    ref = O.reparam_fwd_ref(x.cpu(), conv.mu_kernel.detach().cpu(), conv.rho_kernel.detach().cpu(), d["eps_w"][0].cpu(), conv=CD)
    assert_close(out.cpu(), ref, RTOL, ATOL, what)


def test_data_writes_cannot_leave_a_stale_pack():
    """VERDICT r2 item 8. The reference's own idioms write parameters through ``.data`` (models/dnn_to_bnn.py:95-101
    ``mu_kernel.data.copy_``; utils/util.py:102-117 ``mu_kernel.data = w``), which no version counter sees. Every forward checks a
    device-side fingerprint of (mu, rho) against its pack's (bt_pack_sync), so none of these needs invalidate_pack() -- and an
    unchanged layer is not re-packed."""
    from bayesian_torch_amd import rng
    import bayesian_torch_amd.layers as L
    torch.manual_seed(0)
    conv = L.Conv2dReparameterization(8, 16, 3, padding=1, bias=False).cuda().eval()
    x = torch.randn(4, 8, 6, 6).cuda()
    rng.set_mode("philox")
    _check_conv(conv, x, "first forward")
    n0 = conv.pack_rebuilds()
    assert n0 == 1
    _check_conv(conv, x, "second forward, same parameters")
    assert conv.pack_rebuilds() == n0, "an unchanged layer must not be re-packed"
    conv.mu_kernel.data.copy_(torch.randn_like(conv.mu_kernel) * 0.2)                 # dnn_to_bnn.py:95-101
    _check_conv(conv, x, "after mu.data.copy_ (no invalidate_pack)")
    conv.rho_kernel.data.copy_(torch.randn_like(conv.rho_kernel) * 0.1 - 2.0)
    _check_conv(conv, x, "after rho.data.copy_")
    assert conv.pack_rebuilds() == n0 + 2
    conv.mu_kernel.data.mul_(-1.5)                                                   # in place through .data
    _check_conv(conv, x, "after mu.data.mul_")
    conv.mu_kernel.data[3, 2, 1, 1] += 1e-3                                          # ONE element, one ulp-scale nudge
    _check_conv(conv, x, "after a single-element write")
    assert conv.pack_rebuilds() == n0 + 4
    conv.mu_kernel.data = torch.randn_like(conv.mu_kernel) * 0.3                      # util.py:102-117 (MOPED): a new storage
    conv.rho_kernel.data = torch.full_like(conv.rho_kernel, -2.5)
    _check_conv(conv, x, "after .data = new tensor")
    with torch.no_grad():
        for p in conv.parameters():
            p.add_(0.01)                                                             # what an optimizer step does
    _check_conv(conv, x, "after an in-place update under no_grad")
    sd = {k: v.clone() * 0.5 for k, v in conv.state_dict().items()}
    conv.load_state_dict(sd)
    _check_conv(conv, x, "after load_state_dict")
    # the restore case: parameters go back to an OLDER value -- the pack must follow (a fingerprint kept from that older state must not
    # be trusted over a pack built later)
    old = conv.mu_kernel.detach().clone()
    conv.mu_kernel.data.add_(1.0)
    _check_conv(conv, x, "moved away")
    conv.mu_kernel.data.copy_(old)
    _check_conv(conv, x, "restored to the older value")
    conv.train()
    out = conv(x, return_kl=False)                                                   # training path (autograd bridge) reads the same pack
    d = conv.materialize_last_draw()
    from oracle import bt_oracle as O
    ref = O.reparam_fwd_ref(x.cpu(), conv.mu_kernel.detach().cpu(), conv.rho_kernel.detach().cpu(), d["eps_w"][0].cpu(), conv=CD)
    assert_close(out.detach().cpu(), ref, RTOL, ATOL, "training-mode forward")


def test_model_level_pack_sync_and_graph_replay_follow_updates():
    """mc_forward checks all layers in one bt_pack_sync; a captured McGraph contains that check, so a replay follows in-place
    parameter updates (it used to replay against dropped pack storage). Compared with an eager mc_forward at the same
    RNG coordinates after every kind of update."""
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import McGraph, mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    torch.manual_seed(1)
    net = H.resnet18(10, 16)
    dnn_to_bnn(net, dict(PRIOR, type="Reparameterization"))
    H.fill_bayes_params(net, 3)
    net = net.cuda().eval()
    x = torch.randn(8, 3, 32, 32).cuda()
    rng.set_mode("philox")
    rng.manual_seed(11)
    S = 3
    g = McGraph(net, x, S, with_kl=True, epilogue=False)
    layers = [m for _, m in H.bayes_layers(net)]

    def both(what):
        c = rng.peek_call()
        lg, kg, _ = g.replay()
        lg, kg = lg.clone(), kg.clone()
        rng.set_call(c)
        le, ke = mc_forward(net, x, S)
        assert torch.equal(lg, le), what + ": graph replay != eager forward at the same coordinates"
        assert torch.equal(kg, ke), what + ": kl"
        return lg
    a = both("fresh")
    r0 = [m.pack_rebuilds() for m in layers]
    b = both("again")
    assert [m.pack_rebuilds() for m in layers] == r0 and not torch.equal(a, b)
    layers[5].mu_kernel.data.mul_(1.25)                       # .data write on ONE layer
    both("after a .data write")
    r1 = [m.pack_rebuilds() for m in layers]
    assert r1[5] == r0[5] + 1 and r1[:5] == r0[:5] and r1[6:] == r0[6:], "exactly the changed layer is re-packed"
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.9)                                       # an optimizer-style update of everything
    both("after an in-place update of every parameter")
    sd = {k: (v * 1.1 if v.is_floating_point() else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    both("after load_state_dict")


def test_family_layers_inside_mc_samples():
    """Conv3d / ConvTranspose layers under mc_samples(S > 1) with a shared [B, ...] input (the first layer of a
    model under mc_forward) and with a stacked [S*B, ...] one -- against S separate calls on the same injected draws."""
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd.mc import mc_samples
    torch.manual_seed(0)
    S, B = 2, 3
    cases = [
        (L.Conv3dReparameterization(2, 4, 3, 0, 1, 0, -3.0, padding=1), (2, 5, 6, 6)),
        (L.Conv3dFlipout(2, 4, (2, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)), (2, 4, 7, 7)),
        (L.ConvTranspose2dReparameterization(4, 6, 3, stride=2, padding=1, output_padding=1), (4, 5, 5)),
        (L.ConvTranspose1dFlipout(4, 2, 3, stride=2), (4, 9)),
        (L.ConvTranspose3dReparameterization(2, 3, 2, stride=2), (2, 3, 4, 4)),
    ]
    for layer, xs in cases:
        layer = layer.cuda().eval()
        name = type(layer).__name__
        w = layer.mu_kernel
        xb = torch.randn(B, *xs).cuda()
        xst = torch.randn(S * B, *xs).cuda()
        for shared in (True, False):
            x = xb if shared else xst
            with torch.no_grad():
                probe = layer(x[:B], return_kl=False)
            draw = dict(eps_w=torch.randn(S, *w.shape).cuda(), eps_b=torch.randn(S, layer.out_channels).cuda())
            if layer._flip:
                draw["sign_in"] = torch.randn(S, B, *xs).cuda().sign()
                draw["sign_out"] = torch.randn(S, *probe.shape).cuda().sign()
            layer.inject_draw = draw
            with torch.no_grad(), mc_samples(S, B):
                out = layer(x, return_kl=False)
            assert out.shape == (S * B,) + tuple(probe.shape[1:]), (name, shared, out.shape)
            for s in range(S):
                layer.inject_draw = {k: v[s:s + 1] for k, v in draw.items()}
                with torch.no_grad():
                    one = layer(x if shared else x[s * B:(s + 1) * B], return_kl=False)
                assert_close(out[s * B:(s + 1) * B], one, 1e-5, 1e-6, f"{name} shared={shared} sample {s}")
            layer.inject_draw = None
            with torch.no_grad(), mc_samples(S, B):
                o2 = layer(x, return_kl=False)                  # on-chip draws under MC batching: shape, finite, samples differ
            assert o2.shape == out.shape and torch.isfinite(o2).all()
            d = layer.materialize_last_draw()                  # (_last is populated: ADVICE r2)
            assert d["eps_w"].shape == (S,) + tuple(w.shape) and not torch.equal(d["eps_w"][0], d["eps_w"][1])


def test_family_on_chip_draw_replays_through_the_injection_path():
    """materialize_last_draw() of a family layer returns eps in the REFERENCE's kernel layout: feeding it back through inject_draw
    reproduces the on-chip forward bit for bit (Reparameterization: the draw is eps alone)."""
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd import rng
    torch.manual_seed(0)
    rng.set_mode("philox")
    for layer, xs in [(L.ConvTranspose2dReparameterization(4, 6, 3, stride=2, padding=1, groups=2), (2, 4, 5, 5)),
                      (L.Conv3dReparameterization(2, 4, 3, 0, 1, 0, -3.0, padding=1), (2, 2, 4, 6, 6))]:
        layer = layer.cuda().eval()
        x = torch.randn(*xs).cuda()
        with torch.no_grad():
            a = layer(x, return_kl=False)
        d = layer.materialize_last_draw()
        layer.inject_draw = dict(eps_w=d["eps_w"], eps_b=d["eps_b"])
        with torch.no_grad():
            b = layer(x, return_kl=False)
        assert_close(a, b, 1e-6, 1e-7, type(layer).__name__)


# ------------------------------------------------------------------------------------------------ the direct 1x1 kernel
# Ci, Co, groups, H, W, B, S, bias, extras (scale/shift, residual, relu) [, stride]
DIRECT = {
    "streamed W: K=512 -> 128, 14x14, b48 (reducing 1x1, 4 chunks)": (512, 128, 1, 14, 14, 48, 2, True, False),
    "streamed W: K=1024 -> 96 (partial tile), 7x7, b180, ragged tiles": (1024, 96, 1, 7, 7, 180, 1, False, True),
    "streamed W: K=2048 -> 64, 8x8, b130, S=2": (2048, 64, 1, 8, 8, 130, 2, False, False),
    "downsample: K=256 -> 128, 1x1 stride 2, 28x28 -> 14x14, b48 (resident)": (256, 128, 1, 28, 28, 48, 2, False, True, 2),
    "downsample: K=512 -> 192, 1x1 stride 2, 14x14 -> 7x7, b200 (streamed, odd plane)": (512, 192, 1, 14, 14, 200, 1, True, False, 2),
    "stride (2, 1), groups 2: 128 -> 128, 12x10, b160": (128, 128, 2, 12, 10, 160, 2, True, True, (2, 1)),
    "CIFAR downsample 64 -> 128, 1x1 s2, 8x8 -> 4x4, b128 (few pixels: one sub-tile per wave)": (64, 128, 1, 8, 8, 128, 2, False, True, 2),
    "CIFAR downsample 256 -> 512, 1x1 s2, 2x2 -> 1x1, b128": (256, 512, 1, 2, 2, 128, 2, False, True, 2),
    "K=64 -> 256, 24x24, b16 (ResNet50 layer1 conv3 shape, residual + BN + ReLU)": (64, 256, 1, 24, 24, 16, 2, False, True),
    "K=256 -> 64, 24x24, b16 (reducing 1x1)": (256, 64, 1, 24, 24, 16, 2, True, False),
    "K=128 -> 512, 14x14 (H*W % 4 == 0, W % 4 != 0), b48": (128, 512, 1, 14, 14, 48, 2, False, True),
    "K=256 -> 96 (partial channel tile), 7x7 (odd plane), b200, M % 64 != 0": (256, 96, 1, 7, 7, 200, 1, True, True),
    "groups 2: 128 -> 160 (80 per group: 2 tiles, the second partial), 20x20, b24": (128, 160, 2, 20, 20, 24, 2, True, False),
    "K=192, 33x31, b9 (M = 9207: ragged last sub-tile), S=3": (192, 64, 1, 33, 31, 9, 3, False, True),
}


@pytest.mark.parametrize("name", list(DIRECT))
def test_direct_1x1_kernel_vs_general_kernel_and_c_oracle(name):
    """bt_fused_split_direct.h (persistent workgroups, sampled weights resident in LDS, x straight into registers) against the
    general split kernel on the same RNG coordinates -- same canonical K order and term order: bit for bit -- and against the
    plain-C oracle (fp64 accumulation) on the materialised draws at the unchanged tolerance; KL against the standalone kernel."""
    from oracle import c_oracle as CO
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    Ci, Co, grp, H, W, B, S, bias, extras = DIRECT[name][:9]
    stride = DIRECT[name][9] if len(DIRECT[name]) > 9 else 1
    stride = tuple(stride) if isinstance(stride, tuple) else (stride, stride)
    Ho, Wo = (H - 1) // stride[0] + 1, (W - 1) // stride[1] + 1
    g = torch.Generator().manual_seed(abs(hash(name)) % (1 << 31))
    dev = torch.device("cuda")
    mu = (torch.randn(Co, Ci // grp, 1, 1, generator=g) * 0.1).to(dev)
    rho = (torch.randn(Co, Ci // grp, 1, 1, generator=g) * 0.1 - 3).to(dev)
    mb = (torch.randn(Co, generator=g) * 0.1).to(dev) if bias else None
    rb = (torch.randn(Co, generator=g) * 0.1 - 3).to(dev) if bias else None
    x = torch.randn(S * B, Ci, H, W, generator=g).to(dev)
    conv = dict(stride=stride, padding=(0, 0), dilation=(1, 1), groups=grp)
    kw = {}
    if extras:
        kw = dict(post_scale=(torch.rand(Co, generator=g) + 0.5).to(dev), post_shift=(torch.randn(Co, generator=g) * 0.1).to(dev),
                  residual=torch.randn(S * B, Co, Ho, Wo, generator=g).to(dev), relu=True)
    pri = (torch.zeros_like(mu), torch.ones_like(mu), None if mb is None else torch.zeros_like(mb), None if mb is None else torch.ones_like(mb))
    L = _lib.lib()

    def run(direct):
        L.bt_debug_disable_direct(0 if direct else 1)
        try:
            out, kl = F.fused_forward(x, mu, rho, mb, rb, conv=conv, S=S, shared_x=False, seed=77, call=2, layer_id=9, sample0=5, packed=F.pack_params(mu, rho),
                                      priors=pri, want_kl=True, workspace_owner="t_direct", **kw)
            return out, kl, L.bt_last_kernel_name().decode()
        finally:
            L.bt_debug_disable_direct(0)
    out, kl, kn = run(True)
    assert "fused_split_direct_kernel" in kn and ("streamed" in kn) == (Ci // grp > 256), kn
    ref, klr, kn0 = run(False)
    assert "direct" not in kn0 and "bf16x3" in kn0, kn0
    assert torch.equal(out, ref), f"{name}: direct kernel differs from {kn0}: max abs {float((out - ref).abs().max()):.3e}"
    assert_close(kl, klr, 1e-6, 0, name + ".kl")
    segs = [(mu, rho, pri[0], pri[1])] + ([(mb, rb, pri[2], pri[3])] if bias else [])
    assert_close(kl, _lib.kl_normal(segs, layer_ids=[0] * len(segs), owner="t_direct_kl"), 1e-6, 0, name + ".kl vs standalone")
    # the C oracle on the materialised draws, two samples' worth of the first images (bounded CPU time)
    eps_w = F.rng_fill_normal(77, 2, 9, 5, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(77, 2, 9, 5, 1, S, (Co,), dev).cpu() if bias else None
    nb = min(B, 8)
    o = out.reshape((S, B) + tuple(out.shape[1:]))
    for s in range(S):
        want = CO.reparam_fwd(x[s * B:s * B + nb].cpu(), mu.cpu(), rho.cpu(), eps_w[s], None if mb is None else mb.cpu(), None if rb is None else rb.cpu(),
                              None if eps_b is None else eps_b[s], conv)
        if extras:
            want = torch.relu(want * kw["post_scale"].cpu().view(1, -1, 1, 1) + kw["post_shift"].cpu().view(1, -1, 1, 1) + kw["residual"][s * B:s * B + nb].cpu())
        assert_close(o[s, :nb].cpu(), want, RTOL, ATOL, f"{name}[s={s}] vs C oracle")


def test_direct_kernel_shared_x_and_launch_split_independence():
    """Same global sample ids in one launch or several, shared or stacked x: bit-identical (the direct kernel's eligibility is
    geometric, never a matter of S)."""
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    g = torch.Generator().manual_seed(5)
    dev = torch.device("cuda")
    mu, rho = (torch.randn(128, 64, 1, 1, generator=g) * 0.1).to(dev), (torch.randn(128, 64, 1, 1, generator=g) * 0.1 - 3).to(dev)
    x1 = torch.randn(32, 64, 20, 20, generator=g).to(dev)
    conv = dict(stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1)
    pk = F.pack_params(mu, rho)
    run = lambda x, S, s0, shared: F.fused_forward(x, mu, rho, conv=conv, S=S, shared_x=shared, seed=3, call=1, layer_id=4, sample0=s0, packed=pk)[0]
    full = run(x1, 4, 10, True)
    assert "direct" in _lib.lib().bt_last_kernel_name().decode()
    parts = torch.cat([run(x1, 1, 10 + s, True) for s in range(4)])
    assert torch.equal(parts, full)
    assert torch.equal(run(torch.cat([x1] * 4), 4, 10, False), full)


@pytest.mark.parametrize("case", ["3x3 pad 1 over 1x1 maps, 512 -> 128, b4200 (streamed)", "3x3 dilation 2 pad 2 over 1x1 maps, 256 -> 96, b100", "Linear 256 -> 10, b128", "Linear 1024 -> 200, b4100 (streamed)"])
def test_direct_kernel_on_one_pixel_images_and_linear(case):
    """A padded window over a 1x1 image has ONE live tap (ResNet18 / CIFAR layer4: the centre of the 3x3) and a Linear layer is a 1x1
    kernel over a 1x1 image: both run on the direct kernel with the weights of that tap (tap-major draw index) and 16-byte channel
    vectors for x -- bit-identical to the general kernel, and against the C oracle."""
    from oracle import c_oracle as CO
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    g = torch.Generator().manual_seed(abs(hash(case)) % (1 << 31))
    dev = torch.device("cuda")
    S = 2
    if case.startswith("Linear"):
        In, Out, B = (256, 10, 128) if "256" in case else (1024, 200, 4100)
        mu, rho = (torch.randn(Out, In, generator=g) * 0.1).to(dev), (torch.randn(Out, In, generator=g) * 0.1 - 3).to(dev)
        x = torch.randn(S * B, In, generator=g).to(dev)
        conv = None
    else:
        Ci, Co, B, dil = (512, 128, 4200, 1) if "512" in case else (256, 96, 100, 2)
        mu, rho = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).to(dev), (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1 - 3).to(dev)
        x = torch.randn(S * B, Ci, 1, 1, generator=g).to(dev)
        conv = dict(stride=(1, 1), padding=(dil, dil), dilation=(dil, dil), groups=1)
    mb, rb = (torch.randn(mu.shape[0], generator=g) * 0.1).to(dev), (torch.randn(mu.shape[0], generator=g) * 0.1 - 3).to(dev)
    pk = F.pack_params(mu, rho)
    L = _lib.lib()

    def run(direct):
        L.bt_debug_disable_direct(0 if direct else 1)
        L.bt_debug_disable_skinny(1)   # (narrow heads would go to the split-K flavour first: its own test)
        try:
            out, _ = F.fused_forward(x, mu, rho, mb, rb, conv=conv, S=S, shared_x=False, seed=9, call=4, layer_id=2, sample0=1, packed=pk, relu=True)
            return out, L.bt_last_kernel_name().decode()
        finally:
            L.bt_debug_disable_direct(0)
            L.bt_debug_disable_skinny(0)
    out, kn = run(True)
    assert "fused_split_direct_kernel" in kn and ("streamed" in kn) == ("streamed" in case), kn
    ref, kn0 = run(False)
    assert "direct" not in kn0, kn0
    if "bf16x3" in kn0:     # the general SPLIT kernel: same K order, same terms
        assert torch.equal(out, ref), f"{case}: direct kernel differs from {kn0}: max abs {float((out - ref).abs().max()):.3e}"
    else:                   # (fewer than 112 columns run on the fp32-MFMA kernels when the direct kernel is off: same draws, fp32 rounding apart)
        assert_close(out, ref, 1e-5, 1e-6, case + " vs " + kn0)
    eps_w = F.rng_fill_normal(9, 4, 2, 1, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(9, 4, 2, 1, 1, S, (mu.shape[0],), dev).cpu()
    B = x.shape[0] // S
    for s in range(S):
        xs = x[s * B:s * B + 16].cpu()
        want = torch.relu(CO.reparam_fwd(xs, mu.cpu(), rho.cpu(), eps_w[s], mb.cpu(), rb.cpu(), eps_b[s], conv))
        assert_close(out[s * B:s * B + 16].cpu(), want, RTOL, ATOL, f"{case}[s={s}] vs C oracle")


def test_bench_bare_command_launches_two_ranks_on_one_gpu():
    """VERDICT r2 "missing" 1: `python bench.py --gpus 2` (no torchrun) starts its own ranks, here two gloo ranks sharing this GPU,
    runs the real step (MC-batched forward + KL + epilogue + the packed all-reduce) and relays ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BT_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "cfg2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-parity", "--no-roofline", "--no-extras", "--no-traffic"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    rows = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(rows) == 1 and len(p.stdout.decode().strip().splitlines()) == 1, p.stdout.decode()[:500]
    d = json.loads(rows[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_samples_per_step"] == 16 and d["value"] > 0


# ------------------------------------------------------------------------------------------------ the split-K ("skinny") kernel
# kind, Ci, Co, k, stride, pad, dil, groups, H, B, S, bias, extras
SKINNY = {   # (kind, Ci, Co, k, stride, pad, dil, groups, H, B, S, bias, extras) -- at most 8 workgroups per sample (launch_skinny's gate)
    "classifier head: Linear 512 -> 10, b128, bias (10 of 64 rows; 4 slices)": ("linear", 512, 10, 1, 1, 0, 1, 1, 1, 128, 3, True, False),
    "cfg2 head: Linear 512 -> 10, b256 (2 column tiles x 4 slices), bias + ReLU": ("linear", 512, 10, 1, 1, 0, 1, 1, 1, 256, 2, True, True),
    "Linear 1024 -> 64, b100 (partial column tile, 8 slices), residual + BN + ReLU": ("linear", 1024, 64, 1, 1, 0, 1, 1, 1, 100, 2, True, True),
    "2x2 valid over 2x2 maps (1x1 out, all 4 taps live, slices of 64), 64 -> 48, b100": ("conv", 64, 48, 2, 1, 0, 1, 1, 2, 100, 2, True, False),
    "1x1 stride 2, 2x2 -> 1x1, 256 -> 64 (2 slices)": ("conv", 256, 64, 1, 2, 0, 1, 1, 2, 128, 2, False, True),
    "groups 2: 3x3 pad 1 over 1x1 maps, 128 -> 40 (one slice of 64 per group), b33": ("conv", 128, 40, 3, 1, 1, 1, 2, 1, 33, 2, True, True),
    "3x3 dilation 2 pad 2 over 1x1 maps (centre tap), 128 -> 100 (2 channel tiles)": ("conv", 128, 100, 3, 1, 2, 2, 1, 1, 64, 2, True, False),
}


@pytest.mark.parametrize("name", list(SKINNY))
def test_skinny_split_k_kernel_vs_c_oracle_and_general_kernel(name):
    """bt_fused_split_skinny.h: one-pixel output maps split over K-slices that meet in scratch slabs, combined in slice order by the
    last arriver. Against the C oracle (fp64 accumulation) on the materialised draws at the unchanged tolerance, against the
    general kernels on the same draws (its K order is its own: close, not equal), run twice (deterministic: bit-identical), and
    with the samples launched one at a time (same global sample ids: bit-identical)."""
    from oracle import c_oracle as CO
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    kind, Ci, Co, k, st, pd, dl, grp, H, B, S, bias, extras = SKINNY[name]
    g = torch.Generator().manual_seed(abs(hash(name)) % (1 << 31))
    dev = torch.device("cuda")
    if kind == "linear":
        mu, rho = (torch.randn(Co, Ci, generator=g) * 0.1).to(dev), (torch.randn(Co, Ci, generator=g) * 0.1 - 3).to(dev)
        x = torch.randn(S * B, Ci, generator=g).to(dev)
        conv, oshape = None, (Co,)
    else:
        mu, rho = (torch.randn(Co, Ci // grp, k, k, generator=g) * 0.1).to(dev), (torch.randn(Co, Ci // grp, k, k, generator=g) * 0.1 - 3).to(dev)
        x = torch.randn(S * B, Ci, H, H, generator=g).to(dev)
        conv, oshape = dict(stride=(st, st), padding=(pd, pd), dilation=(dl, dl), groups=grp), (Co, 1, 1)
    mb = (torch.randn(Co, generator=g) * 0.1).to(dev) if bias else None
    rb = (torch.randn(Co, generator=g) * 0.1 - 3).to(dev) if bias else None
    kw = {}
    if extras:
        kw = dict(post_scale=(torch.rand(Co, generator=g) + 0.5).to(dev), post_shift=(torch.randn(Co, generator=g) * 0.1).to(dev),
                  residual=torch.randn((S * B,) + oshape, generator=g).to(dev), relu=True)
    pri = (torch.zeros_like(mu), torch.ones_like(mu), None if mb is None else torch.zeros_like(mb), None if mb is None else torch.ones_like(mb))
    pk = F.pack_params(mu, rho)
    L = _lib.lib()

    def run(skinny, xs=x, S_=S, s0=4, res=None):
        L.bt_debug_disable_skinny(0 if skinny else 1)
        kk = dict(kw)
        if res is not None:
            kk["residual"] = res
        try:
            out, kl = F.fused_forward(xs, mu, rho, mb, rb, conv=conv, S=S_, shared_x=False, seed=21, call=6, layer_id=3, sample0=s0, packed=pk, priors=pri, want_kl=True,
                                      workspace_owner="t_skinny", **kk)
            return out, kl, L.bt_last_kernel_name().decode()
        finally:
            L.bt_debug_disable_skinny(0)
    out, kl, kn = run(True)
    assert "fused_split_skinny_kernel" in kn, kn
    out2, kl2, _ = run(True)
    assert torch.equal(out, out2) and torch.equal(kl, kl2), "the split-K combine must be deterministic"
    ref, klr, kn0 = run(False)
    assert "skinny" not in kn0, kn0
    assert_close(out, ref, 2e-5, 2e-6, f"{name} vs {kn0}")
    assert_close(kl, klr, 1e-6, 0, name + ".kl")
    parts = []
    for s in range(S):     # one sample per launch, same global ids
        o, _, _ = run(True, xs=x[s * B:(s + 1) * B].contiguous(), S_=1, s0=4 + s, res=kw["residual"][s * B:(s + 1) * B].contiguous() if extras else None)
        parts.append(o)
    assert torch.equal(torch.cat(parts), out), "results must not depend on how the samples are launched"
    eps_w = F.rng_fill_normal(21, 6, 3, 4, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(21, 6, 3, 4, 1, S, (Co,), dev).cpu() if bias else None
    nb = min(B, 16)
    o = out.reshape((S, B) + oshape)
    for s in range(S):
        want = CO.reparam_fwd(x[s * B:s * B + nb].cpu(), mu.cpu(), rho.cpu(), eps_w[s], None if mb is None else mb.cpu(), None if rb is None else rb.cpu(),
                              None if eps_b is None else eps_b[s], conv)
        if extras:
            shp = (1, -1) + (1,) * (len(oshape) - 1)
            want = torch.relu(want * kw["post_scale"].cpu().view(shp) + kw["post_shift"].cpu().view(shp) + kw["residual"][s * B:s * B + nb].cpu())
        assert_close(o[s, :nb].cpu(), want, RTOL, ATOL, f"{name}[s={s}] vs C oracle")


def test_skinny_gate_is_geometric():
    """Wide one-pixel layers (ResNet18 layer4: 32 slices per sample) stay off the split-K kernel whatever S is; the head takes it whatever
    S is: the flavour -- and with it the K order -- of a layer never depends on how its samples are launched."""
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    dev = torch.device("cuda")
    L = _lib.lib()
    for (Ci, Co, want) in ((512, 512, False), (512, 10, True)):
        mu, rho = torch.randn(Co, Ci, 1, 1, device=dev) * 0.1, torch.randn(Co, Ci, 1, 1, device=dev) * 0.1 - 3
        pk = F.pack_params(mu, rho)
        for S in (1, 3, 32):
            x = torch.randn(S * 128, Ci, 1, 1, device=dev)
            F.fused_forward(x, mu, rho, conv=dict(stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1), S=S, shared_x=False, seed=1, call=0, layer_id=1, packed=pk,
                            workspace_owner="t_skinny_gate")
            assert ("skinny" in L.bt_last_kernel_name().decode()) == want, (Ci, Co, S, L.bt_last_kernel_name().decode())


@pytest.mark.parametrize("case", ["conv normal prior", "conv laplace prior + bias", "flipout conv + bias", "linear + bias, S=3", "deferred to the side stream"])
def test_kl_term_differentiated_inside_wgrad_equals_autograd_sum(case):
    """bt_conv2d_bwd_kl / FusedForward(opts["kl"]): the layer's KL comes out of the forward kernel as a second differentiable output and
    its gradient is added inside wgrad's finishing pass. Against the plain wiring (FusedForward + KLNormal, autograd adds the two
    gradient tensors) on the same draws: the same bits for every gradient, the same KL value as the fused inference sweep; and
    against the materialising ATen checker at its tolerance. "deferred": the weight gradients computed on the side stream and
    handed over by mc.finish_deferred."""
    import bayesian_torch_amd.autograd as AG
    import bayesian_torch_amd.layers as L
    from bayesian_torch_amd import mc, rng
    rng.set_mode("philox")
    torch.manual_seed(3)
    S, B = (3, 8) if "S=3" in case else (1, 16)
    if case.startswith("linear"):
        layer = L.LinearReparameterization(96, 40, bias=True).cuda()
        x0 = torch.randn(B, 96).cuda()
    elif case.startswith("flipout"):
        layer = L.Conv2dFlipout(16, 24, 3, padding=1, bias=True).cuda()
        x0 = torch.randn(B, 16, 6, 6).cuda()
    else:
        kw = dict(prior_type="laplace", bias=True) if "laplace" in case else dict(bias=False)
        layer = L.Conv2dReparameterization(16, 24, 3, padding=1, **kw).cuda()
        x0 = torch.randn(B, 16, 6, 6).cuda()
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    params = list(layer.parameters())

    def step(fused, impl="hip", defer=False):
        AG.BACKWARD_IMPL = impl
        try:
            rng.manual_seed(5)
            for p in params:
                p.grad = None
            x = x0.clone().requires_grad_(True)
            with mc.mc_samples(S, B) as ctx:
                ctx.train_fused, ctx.deferred = fused, ([] if defer else None)
                out, kl = layer(x)
            loss = (out * torch.linspace(-1, 1, out.numel(), device=out.device).reshape(out.shape)).sum() + kl * 0.37
            loss.backward()
            mc.finish_deferred(ctx)
            torch.cuda.synchronize()
            return kl.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params]
        finally:
            AG.BACKWARD_IMPL = "hip"
    kl0, gx0, g0 = step(False)
    kl1, gx1, g1 = step(True, defer="deferred" in case)
    assert_close(kl1, kl0, 1e-6, 0, case + ".kl")      # (the fused sweep's partial order is the forward kernel's own)
    assert torch.equal(gx1, gx0), case
    for (n, _), a, b in zip(layer.named_parameters(), g1, g0):
        assert torch.equal(a, b), f"{case}: grad of {n} differs: max abs {float((a - b).abs().max()):.3e}"
    if "deferred" not in case:
        _, gx2, g2 = step(True, impl="aten")
        assert_close(gx1, gx2, 2e-4, 2e-5, case + ".dx vs aten")
        for (n, _), a, b in zip(layer.named_parameters(), g1, g2):
            assert_close(a, b, 2e-4, 2e-5, f"{case}: grad of {n} vs aten")


def test_bwd_kl_entry_point_rejects_incomplete_arguments():
    """bt_conv2d_bwd_kl: grad_kl without the weight-gradient outputs, or without mu_w / the priors, is a BT_ERR_BAD_ARG, not a launch."""
    import ctypes as C
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    dev = torch.device("cuda")
    L = _lib.lib()
    mu, rho = torch.randn(8, 8, 1, 1, device=dev), torch.randn(8, 8, 1, 1, device=dev) - 3
    pk = F.pack_params(mu, rho)
    x, g, gkl = torch.randn(4, 8, 2, 2, device=dev), torch.randn(4, 8, 2, 2, device=dev), torch.ones(1, device=dev)
    geom = _lib.bt_conv2d_geom(4, 8, 2, 2, 8, 1, 1, 1, 1, 0, 0, 1, 1, 1)
    ws = torch.empty(max(16, L.bt_conv2d_bwd_workspace(C.byref(geom), 1)), dtype=torch.uint8, device=dev)
    dx, dmu, drho = torch.empty_like(x), torch.empty_like(mu), torch.empty_like(mu)
    D = _lib.bt_draws(None, None, None, None, F._rng(1, 0, 0, 0, None))

    def call(P, dmu_, drho_):
        return L.bt_conv2d_bwd_kl(C.byref(geom), 1, 0, x.data_ptr(), 0, g.data_ptr(), C.byref(P), C.byref(D), gkl.data_ptr(), dx.data_ptr(), _lib.ptr(dmu_), _lib.ptr(drho_),
                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    no_priors = _lib.bt_params(mu.data_ptr(), rho.data_ptr(), None, None, None, None, None, None, pk[0].data_ptr(), pk[1].data_ptr(), 0, 0)
    assert call(no_priors, dmu, drho) < 0 and b"priors" in L.bt_last_error_string()
    pm, ps = torch.zeros_like(mu), torch.ones_like(mu)
    full = _lib.bt_params(mu.data_ptr(), rho.data_ptr(), None, None, pm.data_ptr(), ps.data_ptr(), None, None, pk[0].data_ptr(), pk[1].data_ptr(), 0, 0)
    assert call(full, None, None) < 0 and b"grad_kl needs" in L.bt_last_error_string()
    assert call(full, dmu, drho) == 0
    lap = _lib.bt_params(mu.data_ptr(), rho.data_ptr(), None, None, None, None, None, None, pk[0].data_ptr(), pk[1].data_ptr(), 1, 0)
    assert call(lap, dmu, drho) == 0      # the Laplace branch reads no prior tensors (base_variational_layer.py:74-97)
    torch.cuda.synchronize()
    assert torch.isfinite(dmu).all() and torch.isfinite(drho).all()


@pytest.mark.parametrize("shape", [(3, 5, 112, 112), (2, 3, 7, 9), (1, 1, 1, 1), (2, 2, 16, 130), (1, 4, 33, 2), (1, 2, 5, 300)])
def test_hip_maxpool_pass_equals_torch(shape):
    """bt_maxpool_3x3s2 (the stem's pooling when it cannot be fused: row-band tiles) against torch's max_pool2d(3, 2, 1): bit for bit,
    odd and even widths, rows longer than a wave, -inf and NaN inputs."""
    from bayesian_torch_amd import functional as F
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).cuda()
    if x.numel() > 20:
        x.view(-1)[7] = float("nan")
        x.view(-1)[11] = float("-inf")
    want = torch.nn.functional.max_pool2d(x, 3, 2, 1)
    got = F.maxpool_3x3s2(x)
    assert got.shape == want.shape
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0))
    xv = x[:, :, 1:, :][:, :, :, 1:] if min(shape[2:]) > 2 else x      # a view whose planes are not 8-byte aligned goes through dev_f32's copy
    assert torch.equal(torch.nan_to_num(F.maxpool_3x3s2(xv), nan=0.0), torch.nan_to_num(torch.nn.functional.max_pool2d(xv, 3, 2, 1), nan=0.0))


def test_fold_relu_equals_the_separate_module():
    """fuse.fold_relu: Sequential(Linear, ReLU, Linear) with the ReLU in the first layer's output stage -- the same draws, the same bits."""
    import copy
    from bayesian_torch_amd import rng
    from bayesian_torch_amd.fuse import fold_relu
    from bayesian_torch_amd.harness import resnet as H
    from bayesian_torch_amd.mc import mc_forward
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn
    rng.set_mode("philox")
    torch.manual_seed(1)
    net = H.mlp((192, 128, 10))
    dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "type": "Reparameterization",
                     "moped_enable": False, "moped_delta": 0.5})
    H.fill_bayes_params(net, 2)
    net = net.cuda().eval()
    fused = copy.deepcopy(net)
    assert fold_relu(fused) == 1 and isinstance(fused[1], torch.nn.Identity) and fold_relu(fused) == 0
    x = torch.randn(40, 192).cuda()
    rng.manual_seed(3)
    a, kla = mc_forward(net, x, 3)
    rng.manual_seed(3)
    b, klb = mc_forward(fused, x, 3)
    assert torch.equal(a, b) and torch.equal(kla, klb)


@pytest.mark.parametrize("case", ["layer1 64x64 3x3 8x8, one sample (generic fetch)", "Linear 3072 -> 512, b256, S = 2 (xm 1), bias + ReLU", "3x3 on 2x2 maps 256 -> 96 (xm 2, partial tile)",
                                  "1x1 maps 3x3 512 -> 160, residual + BN + ReLU"])
def test_32_channel_tiles_are_the_same_bits(case):
    """Launches that would leave CUs idle run the general split kernel in 32-channel tiles (launch_split_one: bn32). The K order does
    not depend on the channel tile: the same bits as the 64-channel tiles, KL included; and against the C oracle."""
    from oracle import c_oracle as CO
    from bayesian_torch_amd import _lib
    from bayesian_torch_amd import functional as F
    g = torch.Generator().manual_seed(abs(hash(case)) % (1 << 31))
    dev = torch.device("cuda")
    kw = {}
    if case.startswith("Linear"):
        S, B = 2, 256
        mu, rho = (torch.randn(512, 3072, generator=g) * 0.05).to(dev), (torch.randn(512, 3072, generator=g) * 0.1 - 3).to(dev)
        x, conv, oshape = torch.randn(S * B, 3072, generator=g).to(dev), None, (512,)
        mb, rb = (torch.randn(512, generator=g) * 0.1).to(dev), (torch.randn(512, generator=g) * 0.1 - 3).to(dev)
        kw = dict(relu=True)
    else:
        Ci, Co, H, B, S = (64, 64, 8, 16, 1) if case.startswith("layer1") else (256, 96, 2, 48, 2) if case.startswith("3x3 on") else (512, 160, 1, 128, 2)
        mu, rho = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).to(dev), (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1 - 3).to(dev)
        x, conv, oshape = torch.randn(S * B, Ci, H, H, generator=g).to(dev), dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1), (Co, H, H)
        mb = rb = None
        if "residual" in case:
            kw = dict(post_scale=(torch.rand(Co, generator=g) + 0.5).to(dev), post_shift=(torch.randn(Co, generator=g) * 0.1).to(dev),
                      residual=torch.randn((S * B,) + oshape, generator=g).to(dev), relu=True)
    pri = (torch.zeros_like(mu), torch.ones_like(mu), None if mb is None else torch.zeros_like(mb), None if mb is None else torch.ones_like(mb))
    pk = F.pack_params(mu, rho)
    L = _lib.lib()
    L.bt_debug_force_bn32.argtypes = [__import__("ctypes").c_int]
    L.bt_debug_force_bn32.restype = None

    def run(v):
        L.bt_debug_force_bn32(v)
        L.bt_debug_disable_skinny(1)
        L.bt_debug_disable_direct(1)
        try:
            out, kl = F.fused_forward(x, mu, rho, mb, rb, conv=conv, S=S, shared_x=False, seed=4, call=2, layer_id=6, sample0=3, packed=pk, priors=pri, want_kl=True,
                                      workspace_owner="t_bn32", **kw)
            return out, kl, L.bt_last_kernel_name().decode()
        finally:
            L.bt_debug_force_bn32(-1)
            L.bt_debug_disable_skinny(0)
            L.bt_debug_disable_direct(0)
    o32, k32, n32 = run(1)
    o64, k64, n64 = run(0)
    assert "fused_split_kernel<32,128" in n32 and "fused_split_kernel<64," in n64, (n32, n64)
    assert torch.equal(o32, o64) and torch.equal(k32, k64), f"{case}: max abs {float((o32 - o64).abs().max()):.3e}"
    eps_w = F.rng_fill_normal(4, 2, 6, 3, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(4, 2, 6, 3, 1, S, (mu.shape[0],), dev).cpu() if mb is not None else None
    for s in range(S):
        want = CO.reparam_fwd(x[s * B:s * B + 8].cpu(), mu.cpu(), rho.cpu(), eps_w[s], None if mb is None else mb.cpu(), None if rb is None else rb.cpu(),
                              None if eps_b is None else eps_b[s], conv)
        if "residual" in case:
            shp = (1, -1, 1, 1)
            want = torch.relu(want * kw["post_scale"].cpu().view(shp) + kw["post_shift"].cpu().view(shp) + kw["residual"][s * B:s * B + 8].cpu())
        elif kw.get("relu"):
            want = torch.relu(want)
        assert_close(o32[s * B:s * B + 8].cpu(), want, RTOL, ATOL, f"{case}[s={s}] vs C oracle")
