from bayesian_torch_amd.layers.variational_layers import *  # noqa: F401,F403
