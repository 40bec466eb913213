"""``Conv2dFlipout`` -- drop-in for reference ``layers/flipout_layers/conv_flipout.py:247-439`` on the
two-accumulator fused implicit-GEMM HIP kernel (bt_flipout_conv2d_fwd)."""
from .._family import Conv3dFlipout, ConvTranspose1dFlipout, ConvTranspose2dFlipout, ConvTranspose3dFlipout  # noqa: F401
from .._fused import FusedBayesLayer
from ..base_variational_layer import get_kernel_size

__all__ = ["Conv2dFlipout", "Conv1dFlipout", "Conv3dFlipout", "ConvTranspose1dFlipout", "ConvTranspose2dFlipout", "ConvTranspose3dFlipout"]


class Conv2dFlipout(FusedBayesLayer):
    _kind, _flip, _wname = "conv", True, "kernel"

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError('invalid in_channels size')   # the reference defers this to F.conv2d; fail at construction
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init, self.posterior_rho_init = posterior_mu_init, posterior_rho_init
        self.bias = bias
        self.kl = 0
        kh, kw = get_kernel_size(kernel_size, 2)
        self._build((out_channels, in_channels // groups, kh, kw), bias)

    def forward(self, x, return_kl=True, residual=None):
        return self._forward(x, return_kl, residual)


class Conv1dFlipout(FusedBayesLayer):
    """Drop-in for reference ``conv_flipout.py:57-245`` on the two-accumulator fused kernel (1 x k kernel, 1 x L image)."""
    _kind, _flip, _wname, _one_d = "conv", True, "kernel", True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 prior_mean=0, prior_variance=1, posterior_mu_init=0, posterior_rho_init=-3.0, bias=True):
        super().__init__()
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError('invalid in_channels size')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init, self.posterior_rho_init = posterior_mu_init, posterior_rho_init
        self.bias = bias
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        self._build((out_channels, in_channels // groups, k), bias)

    def forward(self, x, return_kl=True, residual=None):
        return self._forward(x, return_kl, residual)
