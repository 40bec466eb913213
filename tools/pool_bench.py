#!/usr/bin/env python3
"""conv1 (ResNet stem) launch with and without the fused max-pool output stage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import functional as F
dev = torch.device("cuda")
S, B = 32, 128
mu = torch.randn(64, 3, 7, 7, device=dev) * 0.1; rho = torch.randn(64, 3, 7, 7, device=dev) * 0.1 - 3
x = torch.randn(S * B, 3, 32, 32, device=dev)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
conv = dict(stride=(2, 2), padding=(3, 3), dilation=(1, 1), groups=1)
pk = F.pack_params(mu, rho)
kw = dict(conv=conv, S=S, shared_x=False, seed=1, layer_id=3, packed=pk, post_scale=sc, post_shift=sh, relu=True)
print("fused pool available:", F._fused_forward(x, mu, rho, pool=True, call=0, **kw) is not None)
for pool in (False, True):
    for i in range(3): F.fused_forward(x, mu, rho, pool=pool, call=i, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(20): out, _ = F.fused_forward(x, mu, rho, pool=pool, call=i, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"pool={pool}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  out={tuple(out.shape)}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
full, _ = F.fused_forward(x, mu, rho, call=0, **kw)
torch.cuda.synchronize(); e0.record()
for i in range(20): torch.nn.functional.max_pool2d(full, 3, 2, 1)
e1.record(); torch.cuda.synchronize()
print(f"torch max_pool2d alone: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
