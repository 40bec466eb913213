from bayesian_torch_amd.layers.flipout_layers import *  # noqa: F401,F403
