"""``Conv2dReparameterization`` -- drop-in for reference
``layers/variational_layers/conv_variational.py:234-407`` on the fused implicit-GEMM HIP kernel
(bt_reparam_conv2d_fwd); Conv1d on the same kernel; Conv3d and ConvTranspose{1,2,3}d come from layers/_family.py (exact index
re-arrangements around the same launch).  The Multivariate variant is outside this build's scope."""
from .._family import (Conv3dReparameterization, ConvTranspose1dReparameterization,  # noqa: F401
                       ConvTranspose2dReparameterization, ConvTranspose3dReparameterization)
from .._fused import FusedBayesLayer
from ..base_variational_layer import get_kernel_size

__all__ = ["Conv2dReparameterization", "Conv1dReparameterization", "Conv3dReparameterization", "ConvTranspose1dReparameterization",
           "ConvTranspose2dReparameterization", "ConvTranspose3dReparameterization"]


class Conv2dReparameterization(FusedBayesLayer):
    _kind, _flip, _wname = "conv", False, "kernel"

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 prior_mean=0, prior_variance=1, prior_type=None, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError('invalid in_channels size')
        if out_channels % groups != 0:
            raise ValueError('invalid in_channels size')      # sic: same message as the reference
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.prior_mean, self.prior_variance, self.prior_type = prior_mean, prior_variance, prior_type
        self.posterior_mu_init = (posterior_mu_init,)
        self.posterior_rho_init = (posterior_rho_init,)
        self.bias = bias
        kh, kw = get_kernel_size(kernel_size, 2)
        self._build((out_channels, in_channels // groups, kh, kw), bias)

    def forward(self, input, return_kl=True, residual=None):
        return self._forward(input, return_kl, residual)


class Conv1dReparameterization(FusedBayesLayer):
    """Drop-in for reference ``conv_variational.py:68-232`` -- a 1 x k kernel over a 1 x L image on the same fused kernel."""
    _kind, _flip, _wname, _one_d = "conv", False, "kernel", True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError('invalid in_channels size')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init = (posterior_mu_init,)
        self.posterior_rho_init = (posterior_rho_init,)
        self.bias = bias
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        self._build((out_channels, in_channels // groups, k), bias)

    def forward(self, input, return_kl=True, residual=None):
        return self._forward(input, return_kl, residual)
