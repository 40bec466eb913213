"""``LinearReparameterization`` -- drop-in for reference
``layers/variational_layers/linear_variational.py:54-204`` on the fused HIP kernel (bt_reparam_linear_fwd)."""
from .._fused import FusedBayesLayer

__all__ = ["LinearReparameterization"]


class LinearReparameterization(FusedBayesLayer):
    _kind, _flip, _wname = "linear", False, "weight"

    def __init__(self, in_features, out_features, prior_mean=0, prior_variance=1, posterior_mu_init=0,
                 posterior_rho_init=-3.0, bias=True, prior_type='normal'):
        super().__init__()
        assert prior_type is not None, "prior_type must be specified for LinearReparameterization layer"
        self.in_features, self.out_features = in_features, out_features
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init = (posterior_mu_init,)      # 1-tuples: the reference's attribute quirk, kept
        self.posterior_rho_init = (posterior_rho_init,)
        self.prior_type = prior_type
        self.bias = bias
        self._build((out_features, in_features), bias)

    def forward(self, input, return_kl=True, residual=None):
        return self._forward(input, return_kl, residual)
