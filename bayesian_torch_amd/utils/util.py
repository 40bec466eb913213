"""``get_rho`` -- drop-in for reference ``utils/util.py:63-69`` (MOPED initialisation of rho).
Runs once per layer at conversion time on whatever device the deterministic weights live on;
not part of the per-sample hot path."""
import torch


def get_rho(sigma, delta):
    """rho such that log1p(exp(rho)) == delta * |w| (up to the 1e-20 guard)."""
    return torch.log(torch.expm1(delta * torch.abs(sigma)) + 1e-20)


# ---- predictive-uncertainty measures (drop-in for reference utils/util.py:41-60; numpy, host side) ----------------------
def entropy(prob):
    """Shannon entropy along the last axis (the reference's 1e-15 guard inside the log)."""
    import numpy as np
    return -np.sum(prob * np.log(prob + 1e-15), axis=-1)


def predictive_entropy(mc_preds):
    """Entropy of the MC-averaged predictive distribution; mc_preds is [S, ..., classes]."""
    import numpy as np
    return entropy(np.mean(mc_preds, axis=0))


def mutual_information(mc_preds):
    """Predictive entropy minus the mean per-sample entropy (epistemic part)."""
    import numpy as np
    return entropy(np.mean(mc_preds, axis=0)) - np.mean(entropy(mc_preds), axis=0)


def uncertainty_from_mc(res):
    """The same measures from the packed device sums mc_dist.mc_predict() returns (dict with mean_prob, mean_entropy):
    -> (predictive_entropy [B], mutual_information [B]) as device tensors, no [S, B, C] tensor needed."""
    p = res["mean_prob"]
    pe = -(p * torch.log(p + 1e-15)).sum(-1)
    return pe, pe - res["mean_entropy"]


def MOPED(model, det_model, det_checkpoint, delta):
    """Empirical-Bayes initialisation from a deterministic checkpoint (drop-in for reference utils/util.py:72-136):
    priors' means <- deterministic weights, mu <- weights, rho <- get_rho(weights, delta); BatchNorm state copied.
    Layers are matched by walking both module lists in step, Bayesian layers recognised by their repr string."""
    det_model.load_state_dict(torch.load(det_checkpoint, weights_only=True))
    for layer, det_layer in zip(model.modules(), det_model.modules()):
        name = str(layer)
        if name.endswith("Reparameterization()") or name.endswith("Flipout()"):
            wn = "kernel" if name.startswith("Conv") else "weight"
            layer.prior_weight_mu = det_layer.weight.data
            getattr(layer, "mu_" + wn).data = det_layer.weight.data
            getattr(layer, "rho_" + wn).data = get_rho(det_layer.weight.data, delta)
            if layer.mu_bias is not None:
                layer.prior_bias_mu = det_layer.bias.data
                layer.mu_bias.data = det_layer.bias.data
                layer.rho_bias.data = get_rho(det_layer.bias.data, delta)
        elif name.startswith("Batch"):
            layer.weight.data = det_layer.weight.data
            if layer.bias is not None:
                layer.bias.data = det_layer.bias.data
            layer.running_mean.data = det_layer.running_mean.data
            layer.running_var.data = det_layer.running_var.data
            layer.num_batches_tracked.data = det_layer.num_batches_tracked.data
    return model
