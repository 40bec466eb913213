#!/usr/bin/env python3
"""Per-stage timeline of block 0 (hook bt_debug_set_stamp_buffer). Needs the diagnostic build:
   make -C bayesian_torch_amd/csrc clean && make -C bayesian_torch_amd/csrc -j8 STAMPS=1   (rebuild without STAMPS afterwards)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_torch_amd import _lib, functional as F
sys.argv = [sys.argv[0]] + sys.argv[1:]
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--shape", default="layer1"); ap.add_argument("--S", type=int, default=32); ap.add_argument("--B", type=int, default=128); ap.add_argument("--sigma", action="store_true"); ap.add_argument("--noprio", action="store_true"); ap.add_argument("--prio", type=int, default=0); ap.add_argument("--wgs", action="store_true", help="per-workgroup timeline"); ap.add_argument("--pool", action="store_true")
a = ap.parse_args()
SH = {"conv1": (3, 64, 7, 2, 3, 32), "r50conv1": (3, 64, 7, 2, 3, 224), "layer1": (64, 64, 3, 1, 1, 8), "layer2": (128, 128, 3, 1, 1, 4), "layer3": (256, 256, 3, 1, 1, 2), "layer4": (512, 512, 3, 1, 1, 1), "ds2": (64, 128, 1, 2, 0, 8), "l2s": (64, 128, 3, 2, 1, 8), "ds4": (256, 512, 1, 2, 0, 2)}
pri = None
Ci, Co, k, st, pd, H = SH[a.shape]
dev = torch.device("cuda")
mu = torch.randn(Co, Ci, k, k, device=dev) * 0.1; rho = torch.randn(Co, Ci, k, k, device=dev) * 0.1 - 3
x = torch.randn(a.S * a.B, Ci, H, H, device=dev)
conv = dict(stride=(st, st), padding=(pd, pd), dilation=(1, 1), groups=1)
buf = torch.zeros(256 + 4 * 16384, dtype=torch.int64, device=dev)
if a.noprio: buf[200] = 1
if a.prio: buf[200] = a.prio
if a.wgs: buf[201] = 1
L = _lib.lib(); L.bt_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; L.bt_debug_set_stamp_buffer.restype = None
for i in range(3):
    F.fused_forward(x, mu, rho, conv=conv, S=a.S, shared_x=False, seed=1, call=i, layer_id=3)
L.bt_debug_set_stamp_buffer(buf.data_ptr())
pri = (torch.zeros_like(mu), torch.ones_like(mu), None, None)
F.fused_forward(x, mu, rho, conv=conv, S=a.S, shared_x=False, seed=1, call=9, layer_id=3, packed=(F.pack_params(mu, rho) if a.sigma else None), pool=a.pool, relu=a.pool, priors=pri, want_kl=True)
print(L.bt_last_kernel_name().decode())
torch.cuda.synchronize()
pk = F.pack_params(mu, rho) if a.sigma else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    F.fused_forward(x, mu, rho, conv=conv, S=a.S, shared_x=False, seed=1, call=10 + i, layer_id=3, packed=pk, pool=a.pool, relu=a.pool, priors=pri, want_kl=True)
e1.record(); torch.cuda.synchronize()
print("avg launch (prio mode %d): %.1f us" % (int(buf[200]), e0.elapsed_time(e1) * 50))
L.bt_debug_set_stamp_buffer(None)
t = buf.cpu().tolist()
if a.wgs:
    import collections
    rows = []
    for b in range(16384):
        s0, s1, dur, hw = t[256 + 4 * b:256 + 4 * b + 4]
        if s0 == 0: break
        rows.append((s0, s1, dur, hw & 0xffffffff, hw >> 32, b))
    base = min(r[0] for r in rows)
    print(f"{len(rows)} workgroups; kernel span (100 MHz clock) {(max(r[1] for r in rows) - base) / 100:.1f} us")
    durs = sorted((r[1] - r[0]) / 100 for r in rows)
    print("WG wall duration us: min %.1f  median %.1f  p90 %.1f  max %.1f" % (durs[0], durs[len(durs) // 2], durs[int(len(durs) * .9)], durs[-1]))
    clk = sorted(r[2] for r in rows)
    print("WG shader-clock duration: min %d median %d max %d  -> clock %.2f GHz" % (clk[0], clk[len(clk) // 2], clk[-1], clk[len(clk) // 2] / durs[len(durs) // 2] / 1e3))
    starts = sorted((r[0] - base) / 100 for r in rows)
    print("WG start times us (sorted), every 32nd:", [round(v, 1) for v in starts[::32]])
    ends = sorted((r[1] - base) / 100 for r in rows)
    print("WG end times us (sorted), every 32nd:", [round(v, 1) for v in ends[::32]])
    percu = collections.Counter((r[4] & 0xf, r[3] & 0xfffffff0) for r in rows)   # (xcc, HW_ID without the wave slot)
    print("distinct (xcc, hw id) slots:", len(percu), " WGs per slot histogram:", collections.Counter(percu.values()))
t0 = t[0]
if "quad" in L.bt_last_kernel_name().decode():
    print("quad kernel (cycles after entry): patch staged %d  loop starts %d  loop ends %d  staged+barrier %d  end %d; stages:" % (
        t[212] - t[210], t[0] - t[210], t[1] - t[210], t[121] - t[210], t[126] - t[210]), [t[3 + 2 * i] - t[2 + 2 * i] for i in range(3)])
    sys.exit(0)
print("prologue (cycles after kernel entry): tap table %d  buffers cleared %d  | producer: units decoded %d  items decoded %d  loads issued %d  stage 0 begins %d | consumer: ready %d  loop starts %d" % (
    t[211] - t[210], t[212] - t[210], t[213] - t[210], t[214] - t[210], t[215] - t[210], t[128] - t[210], t[216] - t[210], t0 - t[210]))
print("  fast: start->Wloads-issued", t[240]-t[128+6], " Xloads-issued", t[251]-t[240])
print("producer phases st3: loads-issued", t[251]-t[250], " draws", t[252]-t[251], " W->LDS", t[253]-t[252], " X->LDS(end)", t[128+7]-t[253])
print("s_memtime ticks (100 MHz realtime? or shader clock) relative to consumer loop start")
print("consumer: loop", t[1] - t0, "end-of-kernel", t[126] - t0, " epilogue stamps (rel. loop end):", [v - t[1] for v in t[120:124]])
for s in range(60):
    if t[2 + 2 * s] == 0: break
    c0, c1, p0, p1 = t[2 + 2 * s] - t0, t[3 + 2 * s] - t0, t[128 + 2 * s] - t0, t[129 + 2 * s] - t0
    print(f"st {s:2d}  consumer [{c0:7d} .. {c1:7d}] = {c1-c0:6d}   producer [{p0:7d} .. {p1:7d}] = {p1-p0:6d}")
