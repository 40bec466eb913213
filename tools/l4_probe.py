import sys, torch
sys.path.insert(0, ".")
from bayesian_torch_amd import _lib, functional as F
dev = torch.device("cuda")
mu = torch.randn(512, 2048, 1, 1, device=dev) * 0.05; rho = torch.randn(512, 2048, 1, 1, device=dev) * 0.1 - 3
x = torch.randn(16 * 256, 2048, 7, 7, device=dev)
conv = dict(stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1)
pk = F.pack_params(mu, rho)
for i in range(3): F.fused_forward(x, mu, rho, conv=conv, S=16, shared_x=False, seed=1, call=i, layer_id=3, packed=pk)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5): F.fused_forward(x, mu, rho, conv=conv, S=16, shared_x=False, seed=1, call=10 + i, layer_id=3, packed=pk)
e1.record(); torch.cuda.synchronize()
print(_lib.lib().bt_last_kernel_name().decode(), "%.2f ms" % (e0.elapsed_time(e1) / 5))
