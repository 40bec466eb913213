"""bayesian_torch_amd -- MI355X (gfx950) implementation of bayesian-torch's stochastic
variational-layer forward + KL hot path behind the reference's module API.

    from bayesian_torch_amd.layers import LinearReparameterization, Conv2dReparameterization, LinearFlipout, Conv2dFlipout
    from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
    from bayesian_torch_amd.mc import mc_forward

(``import bayesian_torch`` resolves to a thin alias package with the reference's module paths.)
Everything numeric runs in libbtorch_hip.so (hand-written HIP, include/bt_hip.h); importing this
package does not load it -- the first kernel call does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"
