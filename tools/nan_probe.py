import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_split as T
from bayesian_torch_amd import _lib
mu, rho, mb, rb, x, conv, B, S = T._case("odd octet count, one tap")
bad = 0
for it in range(40):
    o0, k0 = T._run(mu, rho, mb, rb, x, conv, S, 0)
    o1, k1 = T._run(mu, rho, mb, rb, x, conv, S, 1)
    n1 = (~torch.isfinite(o1)).nonzero()
    n0 = (~torch.isfinite(o0)).nonzero()
    if len(n1) or len(n0):
        bad += 1
        print(it, "split nonfinite", len(n0), "f32 nonfinite", len(n1), k1)
        if len(n1):
            idx = n1.cpu()
            print("  images", sorted(set(idx[:, 0].tolist()))[:20], "channels", sorted(set(idx[:, 1].tolist()))[:20], "rows", sorted(set(idx[:, 2].tolist())), "cols", sorted(set(idx[:, 3].tolist())))
print("bad iterations", bad)
