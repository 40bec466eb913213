#!/usr/bin/env python3
"""Single-layer microbenchmark of the fused forward (GPU box). Usage:
  python tools/microbench.py --shape layer1 [--S 32] [--B 128] [--iters 20] [--flip] [--kl]
Prints avg kernel time (HIP events), nominal TFLOP/s."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import functional as F

SHAPES = {  # Ci, Co, k, stride, pad, H
    "conv1": (3, 64, 7, 2, 3, 32), "layer1": (64, 64, 3, 1, 1, 8), "layer2": (128, 128, 3, 1, 1, 4), "layer2s": (64, 128, 3, 2, 1, 8),
    "r50l3c3": (256, 1024, 1, 1, 0, 14), "r50l3c1": (1024, 256, 1, 1, 0, 14), "r50l1c3": (64, 256, 1, 1, 0, 56), "r50l2ds": (256, 512, 1, 2, 0, 56),
    "r50l3ds": (512, 1024, 1, 2, 0, 28), "r50l4ds": (1024, 2048, 1, 2, 0, 14), "r50l4c1": (2048, 512, 1, 1, 0, 7), "r50l4c3": (512, 2048, 1, 1, 0, 7),
    "r50l2c1": (512, 128, 1, 1, 0, 28), "r50l2c3": (128, 512, 1, 1, 0, 28), "r50l1ds": (64, 256, 1, 1, 0, 56),
    "lin1024": (1024, 512, 1, 1, 0, 1), "lin2048": (2048, 512, 1, 1, 0, 1), "lin3072": (3072, 512, 1, 1, 0, 1), "lin4096x1024": (4096, 1024, 1, 1, 0, 1),
    "layer3": (256, 256, 3, 1, 1, 2), "layer4": (512, 512, 3, 1, 1, 1), "ds4": (256, 512, 1, 2, 0, 2), "l4s": (256, 512, 3, 2, 1, 2),
}
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="layer1")
ap.add_argument("--S", type=int, default=32)
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--flip", action="store_true")
ap.add_argument("--kl", action="store_true")
ap.add_argument("--shared", action="store_true")
ap.add_argument("--sigma", action="store_true")
ap.add_argument("--mode", type=int, default=0, help="bt_set_contraction: 0 auto (bf16x3 split), 1 fp32 MFMA, 2 bf16x2")
a = ap.parse_args()
Ci, Co, k, st, pd, H = SHAPES[a.shape]
dev = torch.device("cuda")
from bayesian_torch_amd import _lib
_lib.check(_lib.lib().bt_set_contraction(a.mode))
torch.manual_seed(0)
mu = (torch.randn(Co, Ci, k, k, device=dev) * 0.1)
rho = (torch.randn(Co, Ci, k, k, device=dev) * 0.1 - 3)
x = torch.randn((a.B if a.shared else a.S * a.B), Ci, H, H, device=dev)
conv = dict(stride=(st, st), padding=(pd, pd), dilation=(1, 1), groups=1)
pri = (torch.zeros_like(mu), torch.ones_like(mu), None, None)
sig = F.pack_params(mu, rho) if a.sigma else None
def run(i):
    return F.fused_forward(x, mu, rho, flip=a.flip, conv=conv, S=a.S, shared_x=a.shared, priors=pri, want_kl=a.kl, seed=1, call=i, layer_id=3, packed=sig)
for i in range(3):
    out, _ = run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.iters):
    run(i)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
Ho = out.shape[-1]
fl = 2.0 * a.S * a.B * Co * Ho * Ho * Ci * k * k * (2 if a.flip else 1)
print(f"{a.shape} S={a.S} B={a.B} flip={a.flip} kl={a.kl} mode={a.mode}: {ms*1e3:.1f} us  {fl/ms/1e9:.2f} TF/s nominal  out={tuple(out.shape)}  {_lib.lib().bt_last_kernel_name().decode()}")
