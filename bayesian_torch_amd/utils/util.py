"""``get_rho`` -- drop-in for reference ``utils/util.py:63-69`` (MOPED initialisation of rho).
Runs once per layer at conversion time on whatever device the deterministic weights live on;
not part of the per-sample hot path."""
import torch


def get_rho(sigma, delta):
    """rho such that log1p(exp(rho)) == delta * |w| (up to the 1e-20 guard)."""
    return torch.log(torch.expm1(delta * torch.abs(sigma)) + 1e-20)
