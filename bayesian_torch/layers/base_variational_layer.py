from bayesian_torch_amd.layers.base_variational_layer import *  # noqa: F401,F403
