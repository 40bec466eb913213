from bayesian_torch_amd.utils.util import get_rho, entropy, predictive_entropy, mutual_information, MOPED  # noqa: F401
