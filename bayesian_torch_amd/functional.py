"""Functional form of the four fused forwards: tensors in, one C-ABI call, tensors out.
The layer modules (layers/_fused.py) are thin wrappers over ``fused_forward``."""
import ctypes as C

import torch

from . import _lib


def conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1


def fused_forward(x, mu_w, rho_w, mu_b=None, rho_b=None, *, pool=False, **kw):
    """See _fused_forward.  pool=True appends MaxPool2d(3, 2, 1) to the output stage (the ResNet stem): fused into the
    launch when its tiles hold whole output images, else run as a separate pooling pass on the launch's output."""
    if not pool:
        return _fused_forward(x, mu_w, rho_w, mu_b, rho_b, pool=False, **kw)
    r = _fused_forward(x, mu_w, rho_w, mu_b, rho_b, pool=True, **kw)
    if r is None:  # BT_ERR_UNSUPPORTED: nothing was launched
        out, kl = _fused_forward(x, mu_w, rho_w, mu_b, rho_b, pool=False, **kw)
        return maxpool_3x3s2(out), kl
    return r


def maxpool_3x3s2(x):
    """MaxPool2d(3, 2, 1) of an NCHW tensor on the HIP pooling pass (bt_maxpool_3x3s2): the same bits as torch's max_pool2d."""
    x = _lib.dev_f32(x, "input")
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    with _lib.on(x.device):
        _lib.check(_lib.lib().bt_maxpool_3x3s2(x.data_ptr(), out.data_ptr(), N * Cc, H, W, _lib.stream_ptr(x.device)))
    return out


def _fused_forward(x, mu_w, rho_w, mu_b=None, rho_b=None, *, flip=False, conv=None, S=1, shared_x=True,
                   priors=None, eps_w=None, eps_b=None, sign_in=None, sign_out=None,
                   seed=0, call=0, layer_id=0, sample0=0, call_base=None, want_kl=False, workspace_owner="functional",
                   post_scale=None, post_shift=None, residual=None, relu=False, packed=None, pool=False, prior_type="normal"):
    """x: [B, In] (conv=None) or [B, Ci, H, W]; when ``shared_x`` is False x holds S stacked batches
    ([S*B, ...]).  conv: dict(stride=(sh,sw), padding=(ph,pw), dilation=(dh,dw), groups=g) for Conv2d.
    priors: (prior_mu_w, prior_sigma_w, prior_mu_b, prior_sigma_b) -- required when want_kl.
    eps_*/sign_*: injected draws with a leading S axis, or None for the on-chip generators.
    post_scale/post_shift [Co], residual ([S*B, ...] like out, or [B, ...] shared), relu: fused output stage
    (v*scale+shift, +residual, max(.,0)).  packed: (mu_packed, sigma_packed) from pack_params() -- selects the fast kernel.
    prior_type: "normal" | "laplace" (kl_div's branch; only matters when want_kl).
    Returns (out [S*B, ...], kl or None)."""
    x = _lib.dev_f32(x, "input")
    dev = x.device
    tens = dict(mu_w=mu_w, rho_w=rho_w, mu_b=mu_b, rho_b=rho_b, eps_w=eps_w, eps_b=eps_b, sign_in=sign_in, sign_out=sign_out,
                post_scale=post_scale, post_shift=post_shift, residual=residual, mu_packed=None if packed is None else packed[0],
                sigma_packed=None if packed is None else packed[1])
    for k, t in tens.items():
        t = _lib.dev_f32(t, k)
        if t is not None and t.device != dev:
            raise RuntimeError(f"{k} on {t.device} but input on {dev}")
        tens[k] = t
    if x.shape[0] % (1 if shared_x else S):
        raise RuntimeError("stacked input rows are not a multiple of S")
    B = x.shape[0] // (1 if shared_x else S)
    Co = mu_w.shape[0]
    if conv is None:
        In = mu_w.shape[1]
        if x.dim() != 2 or x.shape[1] != In:
            raise RuntimeError(f"expected [N, {In}] input, got {tuple(x.shape)}")
        tail = (Co,)
    else:
        kh, kw = mu_w.shape[2], mu_w.shape[3]
        (sh, sw), (ph, pw), (dh, dw), groups = conv["stride"], conv["padding"], conv["dilation"], conv["groups"]
        Ci, H, W = x.shape[1], x.shape[2], x.shape[3]
        if Ci != mu_w.shape[1] * groups:
            raise RuntimeError(f"input has {Ci} channels, weight expects {mu_w.shape[1] * groups}")
        Ho, Wo = conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError("convolution output would be empty")
        tail = (Co, Ho, Wo)
        geom = _lib.bt_conv2d_geom(B, Ci, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups)
    x_elems = x.numel() // (1 if shared_x else S)
    if pool:
        if conv is None or residual is not None:
            raise RuntimeError("pool=True needs a Conv2d launch without residual")
        tail = (Co, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1)
    out = torch.empty((S * B,) + tail, dtype=torch.float32, device=dev)
    kl = ws = None
    pr = [None] * 4
    L = _lib.lib()
    # layers whose output map is one pixel may run split over K-slices that meet in scratch behind the workspace (include/bt_hip.h)
    gq = geom if conv is not None else _lib.bt_conv2d_geom(B, In, 1, 1, Co, 1, 1, 1, 1, 0, 0, 1, 1, 1)
    scratch = int(L.bt_fused_scratch_bytes(C.byref(gq), S)) if (eps_w is None and not flip and packed is not None) else 0
    if scratch:
        ws = _lib.workspace(workspace_owner, dev, scratch)
    if want_kl:
        if priors is None:
            raise ValueError("want_kl needs priors")
        pr = [_lib.dev_f32(t, "prior") for t in priors]
        kl = torch.empty((), dtype=torch.float32, device=dev)
        ws = _lib.workspace(workspace_owner, dev, scratch)
    P = _lib.bt_params(tens["mu_w"].data_ptr(), tens["rho_w"].data_ptr(), _lib.ptr(tens["mu_b"]), _lib.ptr(tens["rho_b"]),
                       _lib.ptr(pr[0]), _lib.ptr(pr[1]), _lib.ptr(pr[2]), _lib.ptr(pr[3]), _lib.ptr(tens["mu_packed"]), _lib.ptr(tens["sigma_packed"]),
                       _lib.PRIOR_LAPLACE if prior_type == "laplace" else _lib.PRIOR_NORMAL, 0)
    R = _lib.bt_rng(int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(call_base), int(call) & 0xFFFFFFFF, int(layer_id), int(sample0), 0)
    D = _lib.bt_draws(_lib.ptr(tens["eps_w"]), _lib.ptr(tens["eps_b"]), _lib.ptr(tens["sign_in"]), _lib.ptr(tens["sign_out"]), R)
    E = None
    if tens["post_scale"] is not None or tens["residual"] is not None or relu or pool:
        res, rstride = tens["residual"], 0
        if res is not None:
            if res.numel() == out.numel():
                rstride = out.numel() // S
            elif res.numel() * S != out.numel():
                raise RuntimeError(f"residual has {res.numel()} elements, out has {out.numel()} (S={S})")
        if tens["post_scale"] is not None and (tens["post_scale"].numel() != Co or tens["post_shift"] is None or tens["post_shift"].numel() != Co):
            raise RuntimeError("post_scale / post_shift must both have Co elements")
        E = C.byref(_lib.bt_epilogue(_lib.ptr(tens["post_scale"]), _lib.ptr(tens["post_shift"]), _lib.ptr(res), rstride, 1 if relu else 0, 1 if pool else 0))
    tail_args = (x.data_ptr(), 0 if shared_x else x_elems, C.byref(P), C.byref(D), E, out.data_ptr(), _lib.ptr(kl), _lib.ptr(ws),
                 ws.numel() if ws is not None else 0, _lib.stream_ptr(dev))
    with _lib.on(dev):
        if conv is None:
            fn = L.bt_flipout_linear_fwd if flip else L.bt_reparam_linear_fwd
            _lib.check(fn(B, In, Co, S, *tail_args))
        else:
            fn = L.bt_flipout_conv2d_fwd if flip else L.bt_reparam_conv2d_fwd
            rc = fn(C.byref(geom), S, *tail_args)
            if pool and rc == _lib.ERR_UNSUPPORTED:
                return None
            _lib.check(rc)
    return out, kl


def _rng(seed, call, layer_id, sample0, call_base):
    return _lib.bt_rng(int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(call_base), int(call) & 0xFFFFFFFF, int(layer_id), int(sample0), 0)


def rng_fill_normal(seed, call, layer_id, sample0, tensor_id, S, shape, device, call_base=None):
    """Materialise the on-chip eps stream of a weight ([Co, Ci/g, kh, kw] or [Out, In]) or bias ([Co]) tensor
    -> [S, *shape].  The stream is tap-major (include/bt_hip.h), the returned tensor is in natural order."""
    shape = tuple(shape)
    rows, inner = (shape[0], shape[1]) if len(shape) > 1 else (1, shape[0])   # a vector is one row (bias: block co >> 2)
    taps = 1
    for d in shape[2:]:
        taps *= d
    out = torch.empty((S,) + shape, dtype=torch.float32, device=device)
    R = _rng(seed, call, layer_id, sample0, call_base)
    with _lib.on(out.device):
        _lib.check(_lib.lib().bt_rng_normal_fill(C.byref(R), tensor_id, S, rows, inner, taps, out.data_ptr(), _lib.stream_ptr(out.device)))
    return out


def rng_fill_sign(seed, call, layer_id, sample0, tensor_id, S, shape, device, call_base=None):
    """Materialise the on-chip Flipout sign stream (tensor_id 2: sign_in over one sample's x, 3: sign_out) -> [S, *shape]."""
    shape = tuple(shape)
    n = 1
    for d in shape:
        n *= d
    out = torch.empty((S,) + shape, dtype=torch.float32, device=device)
    R = _rng(seed, call, layer_id, sample0, call_base)
    with _lib.on(out.device):
        _lib.check(_lib.lib().bt_rng_sign_fill(C.byref(R), tensor_id, S, n, out.data_ptr(), _lib.stream_ptr(out.device)))
    return out


def pack_params(mu_w, rho_w):
    """Tap-major re-layout of a layer's (mu, softplus(rho)) -> (mu_packed, sigma_packed), each [Co, taps, Ci4]
    (include/bt_hip.h, bt_params). A cache of a pure function of the parameters; rebuild when they change."""
    mu_w, rho_w = _lib.dev_f32(mu_w, "mu_w"), _lib.dev_f32(rho_w, "rho_w")
    Co, Ci = mu_w.shape[0], mu_w.shape[1]
    taps = 1
    for d in mu_w.shape[2:]:
        taps *= d
    C4 = (Ci + 3) // 4 * 4
    mp = torch.empty((Co, taps, C4), dtype=torch.float32, device=mu_w.device)
    sp = torch.empty_like(mp)
    with _lib.on(mu_w.device):
        _lib.check(_lib.lib().bt_pack_params(mu_w.data_ptr(), rho_w.data_ptr(), Co, Ci, taps, mp.data_ptr(), sp.data_ptr(), _lib.stream_ptr(mu_w.device)))
    return mp, sp


def pack_buffers(Co, Ci, taps, device):
    """Persistent storage of one layer's pack: (mu_packed, sigma_packed, state) -- state = the 4 device words bt_pack_sync keeps
    (accumulator, fingerprint of the packed copy, dirty flag of the last call, rebuild count)."""
    C4 = (Ci + 3) // 4 * 4
    mp = torch.empty((Co, taps, C4), dtype=torch.float32, device=device)
    return mp, torch.empty_like(mp), torch.zeros(4, dtype=torch.int64, device=device)


def pack_sync(segments, owner="pack"):
    """segments: list of dict(mu, rho, src_mu|None, src_rho|None, mu_packed, sigma_packed, state, Co, Ci, taps, force) on ONE device.
    Re-packs, ON THE DEVICE and in the current stream, exactly the layers whose (mu, rho) no longer match the fingerprint their pack
    was built from (bt_pack_sync: two launches per 64 layers, no host synchronisation, graph-capturable)."""
    if not segments:
        return
    L = _lib.lib()
    dev = segments[0]["mu"].device
    if len(segments) == 1 and "_c" in segments[0]:      # a layer checking itself again: its marshalled entry is still valid (same tensors, same buffers)
        arr = segments[0]["_c"][0]
        arr[0].force = 1 if segments[0].get("force") else 0
        with _lib.on(dev):
            _lib.check(L.bt_pack_sync(1, arr, _lib.workspace((owner, "pack"), dev).data_ptr(), _lib.WORKSPACE_BYTES, _lib.stream_ptr(dev)))
        return
    for c0 in range(0, len(segments), _lib.PACK_MAX_SEGMENTS):
        chunk = segments[c0:c0 + _lib.PACK_MAX_SEGMENTS]
        arr = (_lib.bt_pack_seg * len(chunk))()
        keep = []
        for i, sg in enumerate(chunk):
            mu, rho = _lib.dev_f32(sg["mu"], "mu_w"), _lib.dev_f32(sg["rho"], "rho_w")
            smu = _lib.dev_f32(sg.get("src_mu"), "src_mu")
            srho = _lib.dev_f32(sg.get("src_rho"), "src_rho")
            if mu.device != dev:
                raise RuntimeError("pack_sync: all layers of one call must live on one device")
            keep.append((mu, rho, smu, srho))
            n_src = (smu if smu is not None else mu).numel()
            if n_src != sg["Co"] * sg["Ci"] * sg["taps"] or mu.numel() != n_src or rho.numel() != n_src:
                raise RuntimeError("pack_sync: geometry does not match the parameter tensors")
            arr[i] = _lib.bt_pack_seg(mu.data_ptr(), rho.data_ptr(), _lib.ptr(smu), _lib.ptr(srho), sg["mu_packed"].data_ptr(), sg["sigma_packed"].data_ptr(),
                                      sg["state"].data_ptr(), sg["Co"], sg["Ci"], sg["taps"], 1 if sg.get("force") else 0, 0)
        if len(segments) == 1:
            segments[0]["_c"] = (arr, keep)
        with _lib.on(dev):
            _lib.check(L.bt_pack_sync(len(chunk), arr, _lib.workspace((owner, "pack"), dev).data_ptr(), _lib.WORKSPACE_BYTES, _lib.stream_ptr(dev)))


def mc_epilogue(logits):
    """logits [S, B, C] -> packed [B*C + B + B*C] = [sum_s softmax | sum_s entropy | sum_s logits]."""
    logits = _lib.dev_f32(logits, "logits")
    S, B, Cc = logits.shape
    packed = torch.empty(B * Cc + B + B * Cc, dtype=torch.float32, device=logits.device)
    with _lib.on(logits.device):
        _lib.check(_lib.lib().bt_mc_epilogue(S, B, Cc, logits.data_ptr(), packed.data_ptr(), _lib.stream_ptr(logits.device)))
    return packed


def fused_backward(x, grad_out, mu_w, rho_w, packed, *, flip=False, conv=None, S=1, shared_x=True, need_x=True, need_w=True,
                   eps_w=None, sign_in=None, sign_out=None, seed=0, call=0, layer_id=0, sample0=0, call_base=None, kl=None):
    """Gradients of ``fused_forward`` on the HIP backward kernels (bt_conv2d_bwd): the draws are regenerated on chip from the
    forward's RNG coordinates (or the injected ones are read).  x / grad_out as the forward saw / produced them.
    -> (dx like x or None, dmu_w, drho_w like mu_w or None).  Bias gradients are row sums of grad_out (caller).
    ``kl = (grad_kl device scalar, prior_mu_w, prior_sigma_w, prior kind)``: the layer's weight-KL term is differentiated in the same
    pass and added to dmu_w / drho_w (bt_conv2d_bwd_kl)."""
    x, g = _lib.dev_f32(x, "input"), _lib.dev_f32(grad_out, "grad_out")
    dev = x.device
    mu_w, rho_w = _lib.dev_f32(mu_w, "mu_w"), _lib.dev_f32(rho_w.detach(), "rho_w")
    B = x.shape[0] // (1 if shared_x else S)
    Co = mu_w.shape[0]
    if conv is None:
        geom = _lib.bt_conv2d_geom(B, mu_w.shape[1], 1, 1, Co, 1, 1, 1, 1, 0, 0, 1, 1, 1)
    else:
        (sh, sw), (ph, pw), (dh, dw), groups = conv["stride"], conv["padding"], conv["dilation"], conv["groups"]
        geom = _lib.bt_conv2d_geom(B, x.shape[1], x.shape[2], x.shape[3], Co, mu_w.shape[2], mu_w.shape[3], sh, sw, ph, pw, dh, dw, groups)
    x_elems = x.numel() // (1 if shared_x else S)
    dx = torch.empty((S,) + (B,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev) if need_x else None
    dmu = torch.empty_like(mu_w) if need_w else None
    drho = torch.empty_like(mu_w) if need_w else None
    L = _lib.lib()
    # partials of wgrad's sample / reduction groups and of dgrad's output-channel pieces (contents need not be initialised)
    ws = torch.empty(max(16, L.bt_conv2d_bwd_workspace(C.byref(geom), S)), dtype=torch.uint8, device=dev)
    inj = [None if t is None else _lib.dev_f32(t, "draw") for t in (eps_w, sign_in, sign_out)]
    gkl = pm = ps = None
    lap = 0
    if kl is not None and need_w:
        gkl = _lib.dev_f32(kl[0].reshape(1), "grad_kl")
        lap = 1 if kl[3] == "laplace" else 0
        pm, ps = (None, None) if lap else (_lib.dev_f32(kl[1], "prior_mu"), _lib.dev_f32(kl[2], "prior_sigma"))
    P = _lib.bt_params(mu_w.data_ptr(), rho_w.data_ptr(), None, None, _lib.ptr(pm), _lib.ptr(ps), None, None, packed[0].data_ptr(), packed[1].data_ptr(), lap, 0)
    R = _rng(seed, call, layer_id, sample0, call_base)     # call_base: the device word of a captured training step (mc.TrainGraph)
    D = _lib.bt_draws(_lib.ptr(inj[0]), None, _lib.ptr(inj[1]), _lib.ptr(inj[2]), R)
    with _lib.on(dev):
        _lib.check(L.bt_conv2d_bwd_kl(C.byref(geom), S, 1 if flip else 0, x.data_ptr(), 0 if shared_x else x_elems, g.data_ptr(), C.byref(P), C.byref(D), _lib.ptr(gkl),
                                      _lib.ptr(dx), _lib.ptr(dmu), _lib.ptr(drho), _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.stream_ptr(dev)))
    if need_x:
        dx = dx.sum(0) if shared_x else dx.reshape(x.shape)
    return dx, dmu, drho


def kl_backward(mu, rho, prior_mu, prior_sigma, grad_kl, laplace=False):
    """(d kl / d mu, d kl / d rho) * grad_kl for one tensor of bt_kl_normal's mean (HIP kernel bt_kl_normal_bwd)."""
    mu, rho = _lib.dev_f32(mu.detach(), "mu"), _lib.dev_f32(rho.detach(), "rho")
    pm = None if laplace else _lib.dev_f32(prior_mu, "prior_mu")
    ps = None if laplace else _lib.dev_f32(prior_sigma, "prior_sigma")
    g = _lib.dev_f32(grad_kl.reshape(1).contiguous(), "grad_kl")
    dmu, drho = torch.empty_like(mu), torch.empty_like(mu)
    with _lib.on(mu.device):
        _lib.check(_lib.lib().bt_kl_normal_bwd(mu.data_ptr(), rho.data_ptr(), _lib.ptr(pm), _lib.ptr(ps), g.data_ptr(), mu.numel(),
                                               _lib.KL_PRIOR_LAPLACE if laplace else 0, dmu.data_ptr(), drho.data_ptr(), _lib.stream_ptr(mu.device)))
    return dmu, drho


def kl_backward_segs(segments, grad_kl, laplace=False):
    """[(d kl / d mu, d kl / d rho) * grad_kl] for every (mu, rho, prior_mu, prior_sigma) segment of bt_kl_normal's sum, ONE HIP
    launch per BT_KL_MAX_SEGMENTS tensors (bt_kl_normal_bwd_segs): the backward of a whole model's get_kl_loss."""
    L = _lib.lib()
    dev = segments[0][0].device
    g = _lib.dev_f32(grad_kl.reshape(1).contiguous(), "grad_kl")
    out = []
    for c0 in range(0, len(segments), _lib.KL_MAX_SEGMENTS):
        chunk = segments[c0:c0 + _lib.KL_MAX_SEGMENTS]
        n = len(chunk)
        arrs = [(C.c_void_p * n)() for _ in range(6)]
        numel = (C.c_int64 * n)()
        keep = []
        for i, (mu, rho, pm, ps) in enumerate(chunk):
            mu, rho = _lib.dev_f32(mu.detach(), "mu"), _lib.dev_f32(rho.detach(), "rho")
            pm = None if laplace else _lib.dev_f32(pm, "prior_mu")
            ps = None if laplace else _lib.dev_f32(ps, "prior_sigma")
            dmu, drho = torch.empty_like(mu), torch.empty_like(mu)
            keep.append((mu, rho, pm, ps))
            out.append((dmu, drho))
            for a, t in zip(arrs, (mu, rho, pm, ps, dmu, drho)):
                a[i] = None if t is None else t.data_ptr()
            numel[i] = mu.numel()
        with _lib.on(dev):
            _lib.check(L.bt_kl_normal_bwd_segs(n, arrs[0], arrs[1], arrs[2] if not laplace else None, arrs[3] if not laplace else None, numel, g.data_ptr(),
                                               _lib.KL_PRIOR_LAPLACE if laplace else 0, arrs[4], arrs[5], _lib.stream_ptr(dev)))
    return out
