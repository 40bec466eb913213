from bayesian_torch_amd.layers.variational_layers.linear_variational import *  # noqa: F401,F403
