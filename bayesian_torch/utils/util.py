from bayesian_torch_amd.utils.util import get_rho  # noqa: F401
