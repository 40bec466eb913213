import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


def load_golden(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    out = {k: (torch.from_numpy(d[k]) if k != "meta" else json.loads(str(d[k]))) for k in d.files}
    return out


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith(prefix) and f.endswith(".npz"))


def layer_tensors(g):
    """Expand a layer fixture into the full tensor set (constant priors are stored as scalars)."""
    m = g["meta"]
    t = dict(g)
    if "prior_mu_w" not in t:
        t["prior_mu_w"] = torch.full_like(g["mu_w"], m["prior_mean"])
        t["prior_sigma_w"] = torch.full_like(g["mu_w"], m["prior_variance"])
        if "mu_b" in g:
            t["prior_mu_b"] = torch.full_like(g["mu_b"], m["prior_mean"])
            t["prior_sigma_b"] = torch.full_like(g["mu_b"], m["prior_variance"])
    for k in ("mu_b", "rho_b", "eps_b", "prior_mu_b", "prior_sigma_b", "sign_in", "sign_out"):
        t.setdefault(k, None)
    conv = None
    c = m["ctor"]
    if "Conv" in m["cls"]:
        nd = g["mu_w"].dim() - 2
        tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * nd
        conv = dict(stride=tup(c.get("stride", 1)), padding=tup(c.get("padding", 0)),
                    dilation=tup(c.get("dilation", 1)), groups=c.get("groups", 1))
        if "Transpose" in m["cls"]:
            conv.update(transposed=True, output_padding=tup(c.get("output_padding", 0)))
    t["conv"] = conv
    return t


def assert_close(a, b, rtol, atol_scale, what=""):
    """|a-b| <= atol_scale*max|b| + rtol*|b|  -- the tolerance form SURVEY.md section 8(c) states."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    atol = atol_scale * float(b.abs().max()) if b.numel() else 0.0
    err = (a - b).abs()
    bad = err > (atol + rtol * b.abs())
    assert not bool(bad.any()), f"{what}: max abs err {float(err.max()):.3e} (atol {atol:.3e}, rtol {rtol}), {int(bad.sum())} bad of {b.numel()}"
