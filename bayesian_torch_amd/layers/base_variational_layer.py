"""Base class of the Bayesian layers (drop-in for reference
``layers/base_variational_layer.py:37-100``): ``dnn_to_bnn_flag`` property, ``kl_div`` and
``get_kernel_size``.  ``kl_div`` runs the HIP KL kernel (bt_kl_normal; both of the reference's prior types)."""
import collections.abc
from itertools import repeat

import torch
import torch.nn as nn

from .. import _lib


def get_kernel_size(x, n):
    return tuple(x) if isinstance(x, collections.abc.Iterable) else tuple(repeat(x, n))


def check_prior_type(prior_type):
    """None and 'normal' select the Gaussian closed form.  (The fork's Conv2d default of None
    makes its own kl_loss raise -- SURVEY.md section 0.3; this build treats None as 'normal',
    which is upstream bayesian-torch behaviour.)"""
    if prior_type is None or prior_type == "normal":
        return "normal"
    if prior_type == "laplace":     # base_variational_layer.py:74-97: closed form against Laplace(0, 1)
        return "laplace"
    raise ValueError(f"Unknown prior_type: {prior_type}")


class BaseVariationalLayer_(nn.Module):
    def __init__(self):
        super().__init__()
        self._dnn_to_bnn_flag = False

    @property
    def dnn_to_bnn_flag(self):
        return self._dnn_to_bnn_flag

    @dnn_to_bnn_flag.setter
    def dnn_to_bnn_flag(self, value):
        self._dnn_to_bnn_flag = value

    def kl_div(self, mu_q, sigma_q, mu_p, sigma_p, prior_type="normal"):
        """KL(Q || P) between element-wise Gaussians ('normal'), or of the Gaussians Q against Laplace(0, 1) ('laplace': the
        reference hard-codes that prior and ignores mu_p / sigma_p), MEAN over elements; 0-dim tensor.
        Takes sigma (not rho), like the reference's method."""
        kind = check_prior_type(prior_type)
        mu_q = _lib.dev_f32(mu_q, "mu_q")
        like = lambda v: torch.broadcast_to(torch.as_tensor(v, dtype=torch.float32, device=mu_q.device), mu_q.shape).contiguous()
        return _lib.kl_normal([(mu_q, like(sigma_q), like(mu_p), like(sigma_p))], rho_is_sigma=True, owner=("kl_div", id(self)), laplace=kind == "laplace")
