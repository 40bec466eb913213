from bayesian_torch_amd.layers import *  # noqa: F401,F403
