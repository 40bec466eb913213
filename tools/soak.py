#!/usr/bin/env python3
"""Soak run: many graph replays of the bench workloads (fresh draws every replay), every output checked for non-finite values
and the KL for drift. usage: python tools/soak.py [replays]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bayesian_torch_amd import rng
from bayesian_torch_amd.mc import McGraph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
for wl in ("cfg3", "cfg4"):
    w = bench.WORKLOADS[wl]
    net = bench.build_model(w, dev)
    rng.set_mode("philox"); rng.manual_seed(1)
    x = torch.randn(*w["x"], device=dev)
    g = McGraph(net, x, w["S"])
    kl0 = None
    t0 = time.time()
    bad = 0
    for i in range(n):
        logits, kl, packed = g.replay()
        if i % 10 == 0:
            ok = bool(torch.isfinite(logits).all()) and bool(torch.isfinite(packed).all()) and bool(torch.isfinite(kl))
            bad += 0 if ok else 1
            kl0 = float(kl) if kl0 is None else kl0
            assert abs(float(kl) - kl0) <= 1e-6 * abs(kl0), (float(kl), kl0)
    torch.cuda.synchronize()
    print(f"{wl}: {n} replays, {bad} non-finite checks, kl {kl0:.6f}, {time.time() - t0:.1f} s")
    assert bad == 0
