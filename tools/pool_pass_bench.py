#!/usr/bin/env python3
"""bt_maxpool_3x3s2 against torch's max_pool2d on the ImageNet stem's output (GPU box): python tools/pool_pass_bench.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import functional as F
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(N, 64, 112, 112, device="cuda")
for name, fn in (("torch", lambda: torch.nn.functional.max_pool2d(x, 3, 2, 1)), ("hip", lambda: F.maxpool_3x3s2(x))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {ms:.3f} ms  {(x.numel() * 4 * 1.25) / ms / 1e9:.2f} TB/s (read + write)")
