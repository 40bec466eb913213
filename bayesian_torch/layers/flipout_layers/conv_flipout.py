from bayesian_torch_amd.layers.flipout_layers.conv_flipout import *  # noqa: F401,F403
