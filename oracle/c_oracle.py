"""ctypes front end of oracle/bt_oracle_c.c (plain-C oracle; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbt_oracle_c.so")
_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "bt_oracle_c.c")):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = C.CDLL(_SO)
        _lib.bto_kl_normal.restype = C.c_double
        _lib.bto_kl_normal.argtypes = [_fp] * 4 + [C.c_int64]
        _lib.bto_kl_laplace.restype = C.c_double
        _lib.bto_kl_laplace.argtypes = [_fp] * 2 + [C.c_int64]
        _lib.bto_sample.argtypes = [_fp] * 3 + [C.c_int64, _fp, _fp]
        _lib.bto_conv2d.argtypes = [_fp] * 4 + [C.c_int] * 14
        _lib.bto_linear.argtypes = [_fp] * 4 + [C.c_int] * 3
        _lib.bto_mul.argtypes = [_fp] * 3 + [C.c_int64]
        _lib.bto_fma_sign.argtypes = [_fp] * 4 + [C.c_int64]
    return _lib


def _a(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def kl_normal(mu, rho, pmu, psig):
    m, r, p, s = _a(mu), _a(rho), _a(pmu), _a(psig)
    return float(lib().bto_kl_normal(_p(m), _p(r), _p(p), _p(s), m.size))


def kl_laplace(mu, rho):
    m, r = _a(mu), _a(rho)
    return float(lib().bto_kl_laplace(_p(m), _p(r), m.size))


def kl_layer(mu_w, rho_w, pmu_w, psig_w, mu_b=None, rho_b=None, pmu_b=None, psig_b=None):
    kl = np.float32(kl_normal(mu_w, rho_w, pmu_w, psig_w))
    if mu_b is not None:
        kl = np.float32(kl + np.float32(kl_normal(mu_b, rho_b, pmu_b, psig_b)))
    return float(kl)


def _sample(mu, rho, eps, want_delta=False):
    m, r, e = _a(mu), _a(rho), _a(eps)
    w, d = np.empty_like(m), np.empty_like(m)
    lib().bto_sample(_p(m), _p(r), _p(e), m.size, _p(w), _p(d))
    return (w, d) if want_delta else w


def _contract(x, w, b, conv):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if conv is None:
        out = np.empty((x.shape[0], w.shape[0]), dtype=np.float32)
        lib().bto_linear(_p(x), _p(w), _p(b), _p(out), x.shape[0], x.shape[1], w.shape[0])
        return out
    (sh, sw), (ph, pw), (dh, dw), g = conv["stride"], conv["padding"], conv["dilation"], conv["groups"]
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho, Wo = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    out = np.empty((B, Co, Ho, Wo), dtype=np.float32)
    lib().bto_conv2d(_p(x), _p(w), _p(b), _p(out), B, Ci, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, g)
    return out


def reparam_fwd(x, mu_w, rho_w, eps_w, mu_b=None, rho_b=None, eps_b=None, conv=None):
    w = _sample(mu_w, rho_w, eps_w)
    b = None if mu_b is None else _sample(mu_b, rho_b, eps_b)
    return torch.from_numpy(_contract(_a(x), w, b, conv))


def flipout_fwd(x, mu_w, rho_w, eps_w, sign_in, sign_out, mu_b=None, rho_b=None, eps_b=None, conv=None):
    xa = _a(x)
    mean = _contract(xa, _a(mu_w), _a(mu_b), conv)
    _, dw = _sample(mu_w, rho_w, eps_w, True)
    db = None if mu_b is None else _sample(mu_b, rho_b, eps_b, True)[1]
    xs = np.empty_like(xa)
    lib().bto_mul(_p(xa), _p(_a(sign_in)), _p(xs), xa.size)
    pert = _contract(xs, dw, db, conv)
    out = np.empty_like(mean)
    lib().bto_fma_sign(_p(mean), _p(pert), _p(_a(sign_out)), _p(out), mean.size)
    return torch.from_numpy(out)
