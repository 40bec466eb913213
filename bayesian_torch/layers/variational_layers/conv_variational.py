from bayesian_torch_amd.layers.variational_layers.conv_variational import *  # noqa: F401,F403
