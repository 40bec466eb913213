"""``LinearFlipout`` -- drop-in for reference ``layers/flipout_layers/linear_flipout.py:49-197`` on the
two-accumulator fused HIP kernel (bt_flipout_linear_fwd).  Unlike the fork (whose 4-argument kl_div call
raises), forward(x) returns the Gaussian KL, as upstream bayesian-torch does."""
from .._fused import FusedBayesLayer

__all__ = ["LinearFlipout"]


class LinearFlipout(FusedBayesLayer):
    _kind, _flip, _wname = "linear", True, "weight"

    def __init__(self, in_features, out_features, prior_mean=0, prior_variance=1, posterior_mu_init=0,
                 posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init, self.posterior_rho_init = posterior_mu_init, posterior_rho_init
        self._build((out_features, in_features), bias, eps_bias_last=True)

    def forward(self, x, return_kl=True, residual=None):
        return self._forward(x, return_kl, residual)
