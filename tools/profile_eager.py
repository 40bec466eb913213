#!/usr/bin/env python3
"""cProfile of the reference harness's loop on the drop-in layers (GPU box): where the host time of an eager model(x) goes."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd.harness import resnet as H
from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss
net = H.resnet18(10)
dnn_to_bnn(net, {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0, "type": "Reparameterization", "moped_enable": False, "moped_delta": 0.5})
net = net.cuda().eval()
x = torch.randn(128, 3, 32, 32, device="cuda")
def loop(n):
    with torch.no_grad():
        for _ in range(n):
            net(x); get_kl_loss(net)
    torch.cuda.synchronize()
loop(5)
import time
t = time.time(); loop(32); print("32 samples: %.1f ms" % ((time.time() - t) * 1e3))
pr = cProfile.Profile(); pr.enable(); loop(32); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
