"""GPU parity (run with -m gpu on the MI355X box): the HIP path, called through the C-ABI, against
(1) the golden vectors produced by running the reference, and (2) the CPU oracle on the same inputs.
Tolerances: out rtol 1e-4 + atol 1e-5*max|out|; KL rtol 1e-4 (tested tighter)."""
import pytest
import torch

from conftest import assert_close, golden_names, layer_tensors, load_golden

pytestmark = pytest.mark.gpu

LAYER_FIX = golden_names("linear_") + golden_names("conv2d_")
RTOL, ATOL = 1e-4, 1e-5


def _cuda(t):
    return None if t is None else t.cuda()


def _run_fixture(g, S=1, want_kl=True, **over):
    from bayesian_torch_amd import functional as F
    flip = "Flipout" in g["meta"]["cls"]
    st = lambda t: None if t is None else _cuda(t).unsqueeze(0)
    kw = dict(flip=flip, conv=g["conv"], S=S, priors=tuple(_cuda(g[k]) for k in ("prior_mu_w", "prior_sigma_w", "prior_mu_b", "prior_sigma_b")),
              eps_w=st(g["eps_w"]), eps_b=st(g["eps_b"]), sign_in=st(g["sign_in"]) if flip else None,
              sign_out=st(g["sign_out"]) if flip else None, want_kl=want_kl)
    kw.update(over)
    return F.fused_forward(_cuda(g["x"]), _cuda(g["mu_w"]), _cuda(g["rho_w"]), _cuda(g["mu_b"]), _cuda(g["rho_b"]), **kw)


@pytest.mark.parametrize("name", LAYER_FIX)
def test_fixture_forward_and_kl(name):
    g = layer_tensors(load_golden(name))
    out, kl = _run_fixture(g)
    assert_close(out.cpu(), g["out"], RTOL, ATOL, name + ".out")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, name + ".kl")
    # the workspace is left zeroed and a second call gives the identical result (deterministic reduction)
    out2, kl2 = _run_fixture(g)
    assert torch.equal(out, out2) and torch.equal(kl, kl2)


@pytest.mark.parametrize("name", LAYER_FIX)
def test_standalone_kl_matches_fused(name):
    from bayesian_torch_amd import _lib
    g = layer_tensors(load_golden(name))
    segs = [tuple(_cuda(g[k]) for k in ("mu_w", "rho_w", "prior_mu_w", "prior_sigma_w"))]
    if g["mu_b"] is not None:
        segs.append(tuple(_cuda(g[k]) for k in ("mu_b", "rho_b", "prior_mu_b", "prior_sigma_b")))
    kl = _lib.kl_normal(segs, layer_ids=[0] * len(segs))
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, name + ".kl_loss")
    assert_close(_lib.kl_normal(segs[:1]).cpu(), g["kl_w"], 1e-5, 0, name + ".kl_w")


@pytest.mark.parametrize("name", ["linear_reparam_k500", "conv2d_reparam_c8x16k3s2", "conv2d_flipout_c8x16k3s2", "linear_flipout_cfg1",
                                  "conv2d_flipout_c64x64k3hw1", "conv2d_reparam_c8x12g2"])
@pytest.mark.parametrize("shared", [True, False])
def test_mc_batched_equals_per_sample_oracle(name, shared):
    """S samples in one launch == the oracle run sample by sample with the same injected draws."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F
    g = layer_tensors(load_golden(name))
    flip = "flipout" in name
    S = 3
    gen = torch.Generator().manual_seed(99)
    B = g["x"].shape[0]
    xs = torch.randn((S,) + tuple(g["x"].shape), generator=gen) if not shared else g["x"].unsqueeze(0).expand(S, *g["x"].shape)
    eps_w = torch.randn((S,) + tuple(g["mu_w"].shape), generator=gen)
    eps_b = torch.randn((S, g["mu_w"].shape[0]), generator=gen) if g["mu_b"] is not None else None
    sgn = lambda shape: torch.empty(shape).uniform_(-1, 1, generator=gen).sign()
    s_in = sgn((S,) + tuple(g["x"].shape)) if flip else None
    s_out = sgn((S,) + tuple(g["out"].shape)) if flip else None
    x_dev = _cuda(g["x"]) if shared else _cuda(xs.reshape((S * B,) + tuple(g["x"].shape[1:])))
    out, _ = F.fused_forward(x_dev, _cuda(g["mu_w"]), _cuda(g["rho_w"]), _cuda(g["mu_b"]), _cuda(g["rho_b"]), flip=flip, conv=g["conv"],
                             S=S, shared_x=shared, eps_w=_cuda(eps_w), eps_b=_cuda(eps_b), sign_in=_cuda(s_in), sign_out=_cuda(s_out))
    out = out.reshape((S,) + tuple(g["out"].shape)).cpu()
    for s in range(S):
        eb = None if eps_b is None else eps_b[s]
        if flip:
            ref = O.flipout_fwd_ref(xs[s], g["mu_w"], g["rho_w"], eps_w[s], s_in[s], s_out[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        else:
            ref = O.reparam_fwd_ref(xs[s], g["mu_w"], g["rho_w"], eps_w[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        assert_close(out[s], ref, RTOL, ATOL, f"{name}[s={s}]")


@pytest.mark.parametrize("name", ["linear_reparam_cfg1", "linear_flipout_k500", "conv2d_reparam_c3x16k7s2", "conv2d_flipout_c3x8k3",
                                  "conv2d_flipout_c8x12g2", "conv2d_reparam_c16x32k1s2nb"])
def test_philox_mode_replays_through_oracle(name):
    """On-chip draws: materialise the same counter streams with bt_rng_*_fill, feed them to the CPU oracle,
    and require the same output -- proves the fused kernel consumes exactly the documented stream."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F
    g = layer_tensors(load_golden(name))
    flip = "flipout" in name
    S, seed, call, lid, s0 = 2, 1234567890123, 7, 42, 5
    out, _ = F.fused_forward(_cuda(g["x"]), _cuda(g["mu_w"]), _cuda(g["rho_w"]), _cuda(g["mu_b"]), _cuda(g["rho_b"]), flip=flip,
                             conv=g["conv"], S=S, seed=seed, call=call, layer_id=lid, sample0=s0)
    out = out.reshape((S,) + tuple(g["out"].shape)).cpu()
    dev = torch.device("cuda")
    eps_w = F.rng_fill_normal(seed, call, lid, s0, 0, S, g["mu_w"].shape, dev).cpu()
    eps_b = F.rng_fill_normal(seed, call, lid, s0, 1, S, (g["mu_w"].shape[0],), dev).cpu() if g["mu_b"] is not None else None
    if flip:
        s_in = F.rng_fill_sign(seed, call, lid, s0, 2, S, g["x"].shape, dev).cpu()
        s_out = F.rng_fill_sign(seed, call, lid, s0, 3, S, g["out"].shape, dev).cpu()
    for s in range(S):
        eb = None if eps_b is None else eps_b[s]
        if flip:
            ref = O.flipout_fwd_ref(g["x"], g["mu_w"], g["rho_w"], eps_w[s], s_in[s], s_out[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        else:
            ref = O.reparam_fwd_ref(g["x"], g["mu_w"], g["rho_w"], eps_w[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        assert_close(out[s], ref, RTOL, ATOL, f"{name}[s={s}]")
        from oracle import c_oracle as CO          # and the independent plain-C oracle (fp64 accumulation)
        if flip:
            refc = CO.flipout_fwd(g["x"], g["mu_w"], g["rho_w"], eps_w[s], s_in[s], s_out[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        else:
            refc = CO.reparam_fwd(g["x"], g["mu_w"], g["rho_w"], eps_w[s], g["mu_b"], g["rho_b"], eb, g["conv"])
        assert_close(out[s], refc, RTOL, ATOL, f"{name}[s={s}] vs C oracle")
    # sample identity is global: the same (sample0 + s) drawn in a different launch gives the same result
    out_b, _ = F.fused_forward(_cuda(g["x"]), _cuda(g["mu_w"]), _cuda(g["rho_w"]), _cuda(g["mu_b"]), _cuda(g["rho_b"]), flip=flip,
                               conv=g["conv"], S=1, seed=seed, call=call, layer_id=lid, sample0=s0 + 1)
    assert torch.equal(out_b.cpu().reshape(out[1].shape), out[1])


def test_softplus_and_kl_extremes():
    """rho from -30 to +30 and inf-overflow region; per-element priors."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import _lib
    rho = torch.linspace(-30, 30, 4001)
    mu = torch.linspace(-2, 2, 4001)
    pmu = torch.full_like(mu, 0.3)
    psig = torch.linspace(0.05, 3.0, 4001)
    kl = _lib.kl_normal([(mu.cuda(), rho.cuda(), pmu.cuda(), psig.cuda())]).cpu()
    ref = O.kl_normal_ref(mu.double(), O.softplus_ref(rho.double()), pmu.double(), psig.double())
    assert_close(kl, ref.float(), 2e-6, 0, "kl over rho range")
    # unaligned base pointer + odd length -> scalar path
    n = 1237
    a = torch.randn(4 * n + 4).cuda()
    segs = [(a[1:n + 1], a[n + 2:2 * n + 2] - 3, a[2 * n + 3:3 * n + 3] * 0.1, a[3 * n + 3:4 * n + 3].abs() + 0.5)]
    ref = O.kl_normal_ref(segs[0][0].cpu(), O.softplus_ref(segs[0][1].cpu()), segs[0][2].cpu(), segs[0][3].cpu())
    assert_close(_lib.kl_normal(segs).cpu(), ref, 1e-5, 0, "kl unaligned")


def test_rng_streams_statistics():
    from bayesian_torch_amd import functional as F
    dev = torch.device("cuda")
    n = 1 << 20
    z = F.rng_fill_normal(2024, 0, 1, 0, 0, 4, (n,), dev)
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.std()) - 1) < 3e-3
    assert abs(float((z ** 3).mean())) < 1e-2 and abs(float((z ** 4).mean()) - 3) < 3e-2
    c = torch.corrcoef(z)                      # independence across samples
    assert float((c - torch.eye(4, device=dev)).abs().max()) < 5e-3
    assert abs(float((z[0, 1:] * z[0, :-1]).mean())) < 5e-3      # lag-1
    z2 = F.rng_fill_normal(2024, 1, 1, 0, 0, 1, (n,), dev)       # next call: a fresh stream
    assert abs(float((z2[0] * z[0]).mean())) < 5e-3
    s = F.rng_fill_sign(2024, 0, 1, 0, 2, 4, (n,), dev)
    zc = F.rng_fill_normal(2024, 0, 1, 0, 0, 1, (64, 32, 3, 3), dev)[0]   # conv-shaped: the tap-major stream, natural layout
    assert abs(float(zc.mean())) < 2e-2 and abs(float(zc.std()) - 1) < 2e-2
    flat = F.rng_fill_normal(2024, 0, 1, 0, 0, 1, (64 * 9 * 32,), dev)[0]   # same Philox blocks, e-order
    assert torch.equal(zc.permute(0, 2, 3, 1).reshape(-1), flat)
    assert bool(((s == 1) | (s == -1)).all())
    assert abs(float(s.mean())) < 3e-3
    assert float((torch.corrcoef(s) - torch.eye(4, device=dev)).abs().max()) < 5e-3
    assert abs(float((s[0, 1:] * s[0, :-1]).mean())) < 5e-3
    # reproducible
    assert torch.equal(z, F.rng_fill_normal(2024, 0, 1, 0, 0, 4, (n,), dev))


def test_mc_epilogue_matches_oracle():
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F
    torch.manual_seed(3)
    for (S, B, Cc) in [(5, 7, 10), (3, 4, 1000), (70, 9, 64), (32, 128, 10)]:
        logits = torch.randn(S, B, Cc) * 4
        packed = F.mc_epilogue(logits.cuda()).cpu()
        p, e, l = O.mc_epilogue_ref(logits)
        assert_close(packed[:B * Cc].reshape(B, Cc), p, 1e-5, 1e-6, "psum")
        assert_close(packed[B * Cc:B * Cc + B], e, 1e-5, 1e-6, "entropy")
        assert_close(packed[B * Cc + B:].reshape(B, Cc), l, 1e-5, 1e-6, "lsum")


def test_fast_kernel_is_bit_identical_to_general_kernel(tmp_path):
    """The specialised kernel (packed parameters, LDS patch; what bench.py runs) and the general kernel (what the golden
    fixtures with injected draws run) use the same arithmetic order: on the same on-chip draws their outputs are equal bit
    for bit. BT_FORCE_GENERIC is read once per process, hence two child processes."""
    import os
    import subprocess
    import sys
    script = r'''
import sys, torch
sys.path.insert(0, %r)
from bayesian_torch_amd import functional as F
torch.manual_seed(0)
dev = torch.device("cuda")
outs = []
for (Ci, Co, k, st, pd, H, B, S, flip) in [(64, 64, 3, 1, 1, 8, 8, 2, False), (128, 256, 3, 2, 1, 4, 16, 2, False), (256, 256, 3, 1, 1, 2, 32, 2, True),
                                            (512, 512, 3, 1, 1, 1, 64, 2, False), (3, 64, 7, 2, 3, 32, 4, 2, False), (64, 128, 1, 2, 0, 8, 8, 1, True)]:
    mu = torch.randn(Co, Ci, k, k, device=dev) * 0.1
    rho = torch.randn(Co, Ci, k, k, device=dev) * 0.1 - 3
    x = torch.randn(S * B, Ci, H, H, device=dev)
    conv = dict(stride=(st, st), padding=(pd, pd), dilation=(1, 1), groups=1)
    pri = (torch.zeros_like(mu), torch.ones_like(mu), None, None)
    o, kl = F.fused_forward(x, mu, rho, flip=flip, conv=conv, S=S, shared_x=False, priors=pri, want_kl=True, seed=5, call=1, layer_id=2,
                            packed=F.pack_params(mu, rho))
    outs += [o.flatten().cpu(), kl.reshape(1).cpu()]
for (In, Out, B, S, flip) in [(512, 10, 128, 2, False), (3072, 512, 64, 1, True)]:
    mu = torch.randn(Out, In, device=dev) * 0.1
    rho = torch.randn(Out, In, device=dev) * 0.1 - 3
    mb, rb = torch.randn(Out, device=dev) * 0.1, torch.randn(Out, device=dev) * 0.1 - 3
    x = torch.randn(B, In, device=dev)
    o, _ = F.fused_forward(x, mu, rho, mb, rb, flip=flip, S=S, seed=5, call=1, layer_id=2, packed=F.pack_params(mu, rho))
    outs.append(o.flatten().cpu())
torch.save(torch.cat(outs), sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for tag, env in (("fast", {"BT_CONTRACTION": "f32"}), ("general", {"BT_CONTRACTION": "f32", "BT_FORCE_GENERIC": "1"})):   # the two fp32-MFMA flavours
        f = str(tmp_path / (tag + ".pt"))
        subprocess.run([sys.executable, "-c", script, f], check=True, env=dict(os.environ, **env), timeout=300)
        res.append(torch.load(f))
    assert res[0].numel() == res[1].numel() and torch.isfinite(res[0]).all()
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("name", ["conv1d_reparam_c6x10k3s2", "conv1d_flipout_c6x10k3s2"])
def test_conv1d_layers_match_reference_golden(name):
    """Conv1d{Reparameterization,Flipout} drop-ins (a 1 x k kernel on the fused conv kernel) vs the reference's outputs."""
    import bayesian_torch_amd.layers as L
    g = load_golden(name)
    m = g["meta"]
    layer = getattr(L, m["cls"])(**m["ctor"]).cuda()
    with torch.no_grad():
        layer.mu_kernel.copy_(g["mu_w"]), layer.rho_kernel.copy_(g["rho_w"]), layer.mu_bias.copy_(g["mu_b"]), layer.rho_bias.copy_(g["rho_b"])
    st = lambda k: g[k].cuda().unsqueeze(0) if k in g else None
    layer.inject_draw = dict(eps_w=st("eps_w"), eps_b=st("eps_b"), sign_in=st("sign_in"), sign_out=st("sign_out"))
    with torch.no_grad():
        out, kl = layer(g["x"].cuda())
    assert_close(out.cpu(), g["out"], RTOL, ATOL, name + ".out")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, name + ".kl")
    layer.inject_draw = None
    from bayesian_torch_amd import rng
    rng.set_mode("philox")
    with torch.no_grad():
        o2 = layer(g["x"].cuda(), return_kl=False)      # on-chip draws: shape and finiteness
    assert o2.shape == out.shape and torch.isfinite(o2).all()


def test_lstm_reparameterization_matches_reference_golden():
    import bayesian_torch_amd.layers as L
    g = load_golden("lstm_reparam_7x5")
    lstm = L.LSTMReparameterization(7, 5).cuda()
    with torch.no_grad():
        for nm in ("ih", "hh"):
            lin = getattr(lstm, nm)
            lin.mu_weight.copy_(g[nm + "_mu_w"]), lin.rho_weight.copy_(g[nm + "_rho_w"]), lin.mu_bias.copy_(g[nm + "_mu_b"]), lin.rho_bias.copy_(g[nm + "_rho_b"])
            lin.inject_draw = [dict(eps_w=g[nm + "_eps_w"][t:t + 1].cuda(), eps_b=g[nm + "_eps_b"][t:t + 1].cuda()) for t in range(4)]
        hs, (h2, cs), kl = lstm(g["x"].cuda())
    assert_close(hs.cpu(), g["hidden_seq"], RTOL, ATOL, "hidden_seq")
    assert_close(cs.cpu(), g["c_ts"], RTOL, ATOL, "c_ts")
    assert_close(kl.cpu(), g["kl"], 1e-5, 0, "kl")
    assert_close(lstm.kl_loss().cpu(), g["kl_loss"], 1e-5, 0, "kl_loss")


@pytest.mark.parametrize("Ci,Co,k,st,pd,H,W,B,flip", [
    (8, 16, 3, 1, 1, 7, 7, 11, False),      # 49 pixels: whole-image tiles that do not divide the tile width
    (8, 16, 3, 1, 1, 14, 14, 3, True),      # 196 pixels: one image per tile
    (4, 8, 3, 1, 1, 30, 30, 2, False),      # 900 pixels: row bands of one image
    (4, 8, 3, 2, 1, 45, 37, 2, False),      # stride 2, odd sizes, bands
    (4, 8, 1, 1, 0, 1, 300, 2, False),      # one row wider than a tile: row segments
    (3, 8, 7, 2, 3, 56, 56, 2, False),      # 7x7 stem on a larger image
    (12, 8, 3, 1, 2, 20, 24, 2, True),      # padding 2 (wide halo), Wo % 4 == 0
    (16, 24, 3, 1, 1, 1, 1, 70, False),     # 1x1 input: x staged as 4-channel vectors (one centre tap)
    (20, 8, 3, 1, 1, 1, 1, 33, True),       # ... Flipout, channels not a multiple of the stage width
    (16, 24, 3, 1, 1, 2, 2, 70, False),     # 2x2 input, pixel-major tiles: x staged as whole-plane vectors
    (12, 8, 3, 1, 1, 2, 2, 33, True),       # ... Flipout
    (8, 16, 3, 2, 1, 2, 2, 40, False),      # 2x2 -> 1x1 (stride 2): four taps on a whole-plane patch
    (6, 8, 3, 1, 1, 2, 2, 9, False),        # channel count that is not a multiple of 4: scalar staging
])
def test_fast_kernel_tile_geometries_vs_oracle(Ci, Co, k, st, pd, H, W, B, flip):
    """Spatial sizes that exercise every tile form of the specialised kernel (whole images / row bands / row segments,
    dead columns, odd widths -> scalar stores, Wo % 4 == 0 -> float4 stores): on-chip draws replayed through the oracle."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F
    gen = torch.Generator().manual_seed(Ci * 100 + H)
    mu = torch.randn(Co, Ci, k, k, generator=gen) * 0.1
    rho = torch.randn(Co, Ci, k, k, generator=gen) * 0.1 - 3
    mb, rb = torch.randn(Co, generator=gen) * 0.1, torch.randn(Co, generator=gen) * 0.1 - 3
    x = torch.randn(B, Ci, H, W, generator=gen)
    conv = dict(stride=(st, st), padding=(pd, pd), dilation=(1, 1), groups=1)
    S, seed, call, lid, s0 = 2, 77, 3, 9, 1
    dev = torch.device("cuda")
    out, _ = F.fused_forward(x.cuda(), mu.cuda(), rho.cuda(), mb.cuda(), rb.cuda(), flip=flip, conv=conv, S=S, seed=seed, call=call,
                             layer_id=lid, sample0=s0, packed=F.pack_params(mu.cuda(), rho.cuda()))
    out = out.reshape((S, B) + tuple(out.shape[1:])).cpu()
    eps_w = F.rng_fill_normal(seed, call, lid, s0, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(seed, call, lid, s0, 1, S, (Co,), dev).cpu()
    for s in range(S):
        if flip:
            s_in = F.rng_fill_sign(seed, call, lid, s0, 2, S, x.shape, dev).cpu()
            s_out = F.rng_fill_sign(seed, call, lid, s0, 3, S, out.shape[1:], dev).cpu()
            ref = O.flipout_fwd_ref(x, mu, rho, eps_w[s], s_in[s], s_out[s], mb, rb, eps_b[s], conv)
        else:
            ref = O.reparam_fwd_ref(x, mu, rho, eps_w[s], mb, rb, eps_b[s], conv)
        assert_close(out[s], ref, RTOL, ATOL, f"sample {s}")


@pytest.mark.gpu
@pytest.mark.parametrize("Ci,Co,k,st,pd,H,W,B,flip,fusable", [
    (3, 64, 7, 2, 3, 32, 32, 6, False, True),     # the CIFAR ResNet stem: two 16x16 images per 512-wide tile
    (3, 24, 7, 2, 3, 32, 32, 5, True, True),      # the Flipout stem: one 16x16 image per 256-wide tile of the split quad flavour
    (6, 24, 3, 1, 1, 8, 8, 5, True, False),       # Flipout stages x through registers (no row-chunk kernel): separate pass
    (6, 24, 3, 1, 1, 8, 8, 5, False, True),       # 8x8 images, channel count that is not a tile multiple, ragged batch
    (8, 16, 3, 1, 1, 12, 8, 3, False, True),      # non-square, even sizes
    (4, 8, 3, 1, 1, 7, 8, 2, False, True),        # odd height: the last pooled row sees a 2-row window
    (3, 8, 7, 2, 3, 96, 96, 2, False, False),     # 48x48 outputs: row-band tiles -> separate pooling pass
    (4, 8, 3, 1, 1, 9, 9, 2, False, False),       # Wo % 4 != 0: scalar store path -> separate pooling pass
])
def test_fused_maxpool_output_stage(Ci, Co, k, st, pd, H, W, B, flip, fusable):
    """bt_epilogue.pool: conv -> scale/shift -> ReLU -> MaxPool2d(3, 2, 1) in one launch, against the oracle's forward
    followed by torch's max_pool2d on the same on-chip draws; geometries the kernel cannot fuse report
    BT_ERR_UNSUPPORTED at the C ABI and the Python layer pools separately (same result)."""
    import ctypes as C
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F, _lib
    gen = torch.Generator().manual_seed(Ci * 10 + H)
    mu = torch.randn(Co, Ci, k, k, generator=gen) * 0.1
    rho = torch.randn(Co, Ci, k, k, generator=gen) * 0.1 - 3
    mb, rb = torch.randn(Co, generator=gen) * 0.1, torch.randn(Co, generator=gen) * 0.1 - 3
    sc, sh = torch.rand(Co, generator=gen) + 0.5, torch.randn(Co, generator=gen) * 0.3
    x = torch.randn(B, Ci, H, W, generator=gen)
    conv = dict(stride=(st, st), padding=(pd, pd), dilation=(1, 1), groups=1)
    S, seed, call, lid, s0 = 2, 5, 1, 4, 0
    dev = torch.device("cuda")
    cu = lambda t: t.cuda()
    packed = F.pack_params(cu(mu), cu(rho))
    kw = dict(flip=flip, conv=conv, S=S, seed=seed, call=call, layer_id=lid, sample0=s0, packed=packed, post_scale=cu(sc), post_shift=cu(sh), relu=True)
    direct = F._fused_forward(cu(x), cu(mu), cu(rho), cu(mb), cu(rb), pool=True, **kw)
    pooled_on_split = "split" in _lib.lib().bt_last_kernel_name().decode()      # (stems: the quad flavour pools too)
    assert (direct is not None) == fusable, "fusability of this geometry changed"
    out, _ = F.fused_forward(cu(x), cu(mu), cu(rho), cu(mb), cu(rb), pool=True, **kw)
    out_on_split = "split" in _lib.lib().bt_last_kernel_name().decode()     # (fused, or the unpooled launch behind a separate pooling pass)
    assert direct is None or out_on_split == pooled_on_split
    _lib.lib().bt_set_contraction(0 if out_on_split else 1)       # compare like with like: same contraction flavour as the launch that made `out`
    try:
        full, _ = F.fused_forward(cu(x), cu(mu), cu(rho), cu(mb), cu(rb), **kw)
    finally:
        _lib.lib().bt_set_contraction(0)
    Ho, Wo = full.shape[2], full.shape[3]
    assert tuple(out.shape) == (S * B, Co, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1)
    # the fused pool is exactly max_pool2d of the unpooled launch's output (same draws, same arithmetic)
    assert torch.equal(out, torch.nn.functional.max_pool2d(full, 3, 2, 1))
    out = out.reshape((S, B) + tuple(out.shape[1:])).cpu()
    eps_w = F.rng_fill_normal(seed, call, lid, s0, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(seed, call, lid, s0, 1, S, (Co,), dev).cpu()
    for s in range(S):
        if flip:
            s_in = F.rng_fill_sign(seed, call, lid, s0, 2, S, x.shape, dev).cpu()
            s_out = F.rng_fill_sign(seed, call, lid, s0, 3, S, (B, Co, Ho, Wo), dev).cpu()
            ref = O.flipout_fwd_ref(x, mu, rho, eps_w[s], s_in[s], s_out[s], mb, rb, eps_b[s], conv)
        else:
            ref = O.reparam_fwd_ref(x, mu, rho, eps_w[s], mb, rb, eps_b[s], conv)
        ref = torch.nn.functional.max_pool2d(torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
        assert_close(out[s], ref, RTOL, ATOL, f"sample {s}")


@pytest.mark.gpu
def test_fused_maxpool_propagates_nan():
    from bayesian_torch_amd import functional as F
    mu = torch.zeros(4, 4, 1, 1); mu[range(4), range(4)] = 1.0     # identity 1x1 conv, sigma ~ 0
    rho = torch.full((4, 4, 1, 1), -40.0)
    x = torch.randn(2, 4, 8, 8)
    x[0, 1, 3, 4] = float("nan")
    conv = dict(stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1)
    kw = dict(conv=conv, S=1, seed=1, packed=F.pack_params(mu.cuda(), rho.cuda()))
    out, _ = F.fused_forward(x.cuda(), mu.cuda(), rho.cuda(), pool=True, **kw)
    full, _ = F.fused_forward(x.cuda(), mu.cuda(), rho.cuda(), **kw)      # NaN in every channel of that pixel (0 * NaN)
    ref = torch.nn.functional.max_pool2d(full, 3, 2, 1)                   # the device op the unfused path runs: NaN wins
    assert torch.equal(torch.isnan(out), torch.isnan(ref)) and torch.isnan(ref).any() and not torch.isnan(ref).all()


@pytest.mark.gpu
@pytest.mark.parametrize("flip", [True, False])
def test_wide_tiles_at_chip_filling_sizes_vs_oracle(flip):
    """The wide tiles are only chosen when a launch fills the chip (>= 256 workgroups), which the small cases above never do:
    128 images of 8x8, 8 samples -> Flipout's 64x256 tile / the 64x512 row-chunk tile. Two of the samples are replayed through
    the oracle (on-chip draws), ragged channel counts included."""
    from oracle import bt_oracle as O
    from bayesian_torch_amd import functional as F
    Ci, Co, B, S = 12, 48, 128, 8
    gen = torch.Generator().manual_seed(21)
    mu = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.1
    rho = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.1 - 3
    mb, rb = torch.randn(Co, generator=gen) * 0.1, torch.randn(Co, generator=gen) * 0.1 - 3
    x = torch.randn(B, Ci, 8, 8, generator=gen)
    conv = dict(stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=1)
    seed, call, lid, s0 = 31, 4, 6, 2
    dev = torch.device("cuda")
    out, _ = F.fused_forward(x.cuda(), mu.cuda(), rho.cuda(), mb.cuda(), rb.cuda(), flip=flip, conv=conv, S=S, seed=seed, call=call,
                             layer_id=lid, sample0=s0, packed=F.pack_params(mu.cuda(), rho.cuda()))
    out = out.reshape(S, B, Co, 8, 8).cpu()
    eps_w = F.rng_fill_normal(seed, call, lid, s0, 0, S, mu.shape, dev).cpu()
    eps_b = F.rng_fill_normal(seed, call, lid, s0, 1, S, (Co,), dev).cpu()
    for s in (0, S - 1):
        if flip:
            s_in = F.rng_fill_sign(seed, call, lid, s0, 2, S, x.shape, dev).cpu()
            s_out = F.rng_fill_sign(seed, call, lid, s0, 3, S, (B, Co, 8, 8), dev).cpu()
            ref = O.flipout_fwd_ref(x, mu, rho, eps_w[s], s_in[s], s_out[s], mb, rb, eps_b[s], conv)
        else:
            ref = O.reparam_fwd_ref(x, mu, rho, eps_w[s], mb, rb, eps_b[s], conv)
        assert_close(out[s], ref, RTOL, ATOL, f"sample {s}")
