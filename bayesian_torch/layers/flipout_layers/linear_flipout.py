from bayesian_torch_amd.layers.flipout_layers.linear_flipout import *  # noqa: F401,F403
