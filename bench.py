#!/usr/bin/env python3
"""Headline benchmark: MC-samples/sec (forward+KL), Bayesian-ResNet18 CIFAR batch=128 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1: this process starts N rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their
environment -- the env:// contract of the reference's launcher, utils/utils.py:483-495) BEFORE it makes any GPU call, relays rank
0's JSON line and exits non-zero if any rank does. Started by `python -m torch.distributed.run` (WORLD_SIZE already set) it is
one of the ranks. Backend "nccl" (= RCCL); BT_DIST_BACKEND=gloo rehearses the N > 1 path with the ranks sharing one GPU.

One step = one pass of the hot path over one synthetic batch: the device-side pack check (bt_pack_sync), S MC samples (default 32
per GPU, BASELINE.json cfg3) of the dnn_to_bnn-converted ResNet18 forward, every Bayesian layer's KL (fused into the forward
kernels), the softmax / entropy MC epilogue, and -- for N > 1 -- the one packed all-reduce. Inputs and parameters are resident in
HBM before the timed region. scaling: "weak" = S samples per rank (cfg3's default; the job does N*S per step), "strong" = a fixed
global sample count sharded over the ranks with mc_dist.shard (cfg5's default: 128 samples over the node).

Prints ONE JSON line (rank 0).
  roofline      a second pass of the same K steps with a HIP-event pair around every fused launch (events on the launch stream).
                The headline fields describe the DOMINANT kernel instance against the pipe it issues on: the split kernels run on
                the bf16 matrix pipe (2.5 PFLOP/s dense), `achieved` = 6 x executed fp32-equivalent FLOPs (the six piece products)
                / launch time; `frac_issued` adds the K slots the schedule pads (odd tap counts). Per kernel group: the MFMA, the
                VALU weight-synthesis and the HBM bound of the launch and which one binds. `traffic`: FETCH_SIZE / WRITE_SIZE of
                the same command, collected by two child rocprofv3 --pmc passes started before this process touches the GPU.
  extras        value_with_pack_rebuild (every pack rebuilt inside the timed step), eager_caller_loop (the reference harness's own
                loop -- `for s in range(S): model(x)` then get_kl_loss, no graph, no BN folding:
                examples/main_bayesian_cifar_dnn2bnn.py:402-410,541-545).
  parity        outside the timed region: one sample through the same kernels (on-chip draws), every layer's output and the logits
                against the CPU oracle on the same draws.
  cpu_baseline  the oracle -- the reference's exact ATen op sequence -- on this process's host cores (count stated) for a bounded
                sample, plus a 1-thread line.
"""
import argparse
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bayesian_torch_amd import functional as BF  # noqa: E402
from bayesian_torch_amd import mc_dist, rng  # noqa: E402
from bayesian_torch_amd.harness import resnet as H  # noqa: E402
from bayesian_torch_amd.mc import mc_forward  # noqa: E402
from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn  # noqa: E402

# MI355X_MICROARCH.md: dense matrix peaks, HBM3E peak; VALU = 256 CUs x 4 SIMDs x 16 lanes at the 2.4 GHz boost clock
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
PEAK_VALU_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# VALU instructions of the producers per sampled weight (one Philox4x32-10 block + two Box-Muller pairs = ~130 per 4 weights, + mu +
# sigma*eps, the three-piece split and the packing) and per staged activation (split + packing): an estimate from the ISA of the
# producer loop, used only to say which bound a launch is nearest to.
VALU_PER_WEIGHT, VALU_PER_X = 40.0, 7.0
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
         "moped_enable": False, "moped_delta": 0.5}

WORKLOADS = {
    "cfg3": dict(desc="cfg3: Bayesian-ResNet18 via dnn_to_bnn (Conv2dReparameterization), CIFAR 3x32x32, batch=128", scaling="weak",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Reparameterization", S=32, S_total=32),
    "cfg4": dict(desc="cfg4: Bayesian-ResNet18 Flipout (Conv2dFlipout/LinearFlipout), CIFAR 3x32x32, batch=128", scaling="weak",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Flipout", S=32, S_total=32),
    "cfg5": dict(desc="cfg5: Bayesian-ResNet50 Reparameterization, ImageNet 3x224x224, batch=256 (128 MC samples / 8 GPUs = 16 per GPU)", scaling="strong",
                 net=lambda: H.resnet50(1000, 64), x=(256, 3, 224, 224), btype="Reparameterization", S=16, S_total=128),
    "cfg2": dict(desc="cfg2: MLP 3072->512->10 (LinearReparameterization), batch=256", scaling="weak",
                 net=lambda: H.mlp((3072, 512, 10)), x=(256, 3072), btype="Reparameterization", S=8, S_total=8),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=0, help="MC samples per GPU per step (weak) / in total (strong); default: the workload's")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: S samples per rank; strong: a fixed global sample count sharded over the ranks (mc_dist.shard). Default: cfg5 strong, the rest weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_with_pack_rebuild and eager_caller_loop")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two child rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a captured HIP graph")
    ap.add_argument("--no-fuse", action="store_true", help="keep BatchNorm/ReLU/add as separate torch modules")
    ap.add_argument("--layers-json", default="", help="write the per-layer roofline table here")
    ap.add_argument("--train", action="store_true", help="secondary line: one TRAINING step per MC sample (forward + KL + HIP backward + SGD step, S = 1, BatchNorm in train mode)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launcher / rendezvous / sharding / collective rehearsal WITHOUT any kernel (runs on CPU-only hosts under BT_DIST_BACKEND=gloo): prints a line with value null")
    return ap.parse_args(argv)


# ======================================================================================================== multi-rank launcher
def launch_ranks(n, argv):
    """Parent of `python bench.py --gpus N`: N fresh rank processes, no GPU call in this process, rank 0's stdout relayed."""
    backend = os.environ.get("BT_DIST_BACKEND", "nccl")
    if backend == "nccl":
        have = torch.cuda.device_count()          # (counts devices without initialising HIP on this image)
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs for the RCCL backend, found {have} (BT_DIST_BACKEND=gloo rehearses with shared GPUs)", file=sys.stderr)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    import threading
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    deadline = time.time() + float(os.environ.get("BT_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            rc = next(c for c in codes if c not in (None, 0))
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            print("bench.py: the ranks did not finish in time", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.2)
    for p in procs:                                # a failed or late run: end exactly the processes started here
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    rd.join(timeout=5)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    for ln in text.splitlines():          # rank 0's JSON line goes to stdout; anything else it printed there (a backend's banner) to stderr
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr)
    sys.stdout.flush()
    if rc == 0 and not any(ln.startswith("{") for ln in text.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 3
    return rc if rc >= 0 else 128 - rc


def init_ranks(need_gpu=True):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BT_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path with ranks sharing one GPU
        if need_gpu:
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, rank, local


def plumbing_only(args, w):
    """Everything of the N-rank path except the kernels: rendezvous, sample sharding, the packed all-reduce, barrier + MAX timing."""
    world, rank, _ = init_ranks(need_gpu=False)
    scaling = args.scaling or w["scaling"]
    if scaling == "strong":
        S_total = args.samples or w["S_total"]
        first, S = mc_dist.shard(S_total, rank, world)
    else:
        S = args.samples or w["S"]
        S_total, first = S * world, rank * S
    B, C = 4, 3
    shards = [None] * world
    if world > 1:
        dist.all_gather_object(shards, (first, S))
        dist.barrier()
    else:
        shards = [(first, S)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        packed = torch.full((2 * B * C + B,), float(S))                   # stands for sum_s over this rank's samples of (softmax | entropy | logits)
        buf = mc_dist.finish_pack(packed, torch.tensor(7.0), world)
        mc_dist.reduce_packed(buf)
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    res = mc_dist.unpack(buf, B, C, S_total)
    ok = bool(torch.allclose(res["mean_prob"], torch.ones(B, C)) and abs(float(res["kl"]) - 7.0) < 1e-5)
    covered = sorted(shards) == sorted(mc_dist.shard(S_total, r, world) for r in range(world)) if scaling == "strong" else True
    if rank == 0:
        print(json.dumps(dict(metric="launcher rehearsal (no kernels)", value=None, unit="MC-samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                              plumbing_only=True, scaling=scaling, global_samples_per_step=S_total, shards=shards, shards_cover_all_samples=covered and sum(c for _, c in shards) == S_total,
                              packed_allreduce_ok=ok, backend=os.environ.get("BT_DIST_BACKEND", "nccl") if world > 1 else None,
                              master=f"{os.environ.get('MASTER_ADDR', '-')}:{os.environ.get('MASTER_PORT', '-')}", ms_per_step=round(dt / max(1, args.steps) * 1e3, 4))))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 4


# ======================================================================================================== model + per-launch model
def build_model(w, dev, seed=0):
    torch.manual_seed(seed)
    net = w["net"]()
    dnn_to_bnn(net, dict(PRIOR, type=w["btype"]))
    H.fill_bayes_params(net, 1)
    return net.to(dev).eval()


def _steps_for(n, quad):
    """K16 slots the schedule issues for n active taps of one channel octet (bt_fused_split.h: taps in pairs, an odd last tap with an
    empty half; one active tap: octets in pairs -- no padding; stems, bt_fused_split_quad.h: taps in groups of four)."""
    if n <= 0:
        return 0
    if quad:
        return (n + 3) // 4 * 4
    return n if n == 1 else (n + 1) // 2 * 2


def layer_model(m, flip, S):
    """Per-LAUNCH figures (S samples) of one Bayesian layer from its geometry, the kernel instance that ran and the launch's tile
    grid (bt_last_launch_info): fp32-equivalent FLOPs three ways (nominal: padding taps included; executed: the kernels' tap
    schedule, channels padded to the quad; effective: products with real pixels), MFMA FLOPs issued on the kernel's pipe, weights
    drawn, activations staged, algorithmic HBM bytes -- and the time each of the three bounds allows."""
    w = m._w("mu")
    st = m._last
    xs, os_ = st["x_shape"], st["out_shape"]
    kn, li = st.get("kernel", "?"), st.get("launch", {})
    B, Co = os_[0], os_[1]
    f = 2 if flip else 1
    if m._kind == "linear":
        Cig, kh, kw, Ho, Wo, Hh, Ww = w.shape[1], 1, 1, 1, 1, 1, 1
        rows, cols = [[True]], [[True]]
        g = 1
    else:
        cd = m._conv_desc()
        (sh, sw), (ph, pw), (dh, dw), g = cd["stride"], cd["padding"], cd["dilation"], cd["groups"]
        Cig = w.shape[1]
        kh, kw = (1, w.shape[2]) if w.dim() == 3 else (w.shape[2], w.shape[3])
        Hh, Ww = (1, xs[2]) if len(xs) == 3 else (xs[2], xs[3])
        Ho, Wo = BF.conv_out_hw(Hh, Ww, kh, kw, sh, sw, ph, pw, dh, dw)
        rows = [[0 <= ho * sh - ph + a * dh < Hh for a in range(kh)] for ho in range(Ho)]     # rows[ho][kh]: tap row meets data
        cols = [[0 <= wo * sw - pw + b * dw < Ww for b in range(kw)] for wo in range(Wo)]
    nrow, ncol = [sum(r) for r in rows], [sum(c) for c in cols]
    eff_taps = sum(nrow) * sum(ncol)                                                         # sum over output pixels of in-bounds taps
    gh, gw = sum(any(r[a] for r in rows) for a in range(kh)), sum(any(c[b] for c in cols) for b in range(kw))
    glob_taps = gh * gw
    pixel_major = bool(li["pixel_major"]) if "pixel_major" in li else (m._kind != "linear" and 2 <= Ho * Wo <= 4 and (ph > 0 or pw > 0))
    row_tiles = bool(li.get("row_tiles", 0))
    Cig4 = (Cig + 3) // 4 * 4
    nominal = 2.0 * B * Co * Ho * Wo * Cig * kh * kw * f * S
    effective = 2.0 * B * Co * Cig * eff_taps * f * S
    quad = "quad" in kn
    split = "bf16" in kn
    # taps per output pixel as the tile that holds it schedules them, and K slots issued for them
    if pixel_major:
        per_px = [(nrow[ho] * ncol[wo]) for ho in range(Ho) for wo in range(Wo)]
    elif row_tiles:
        per_px = [nrow[ho] * gw for ho in range(Ho) for _ in range(Wo)]
    else:
        per_px = [glob_taps] * (Ho * Wo)
    executed = 2.0 * B * Co * Cig4 * sum(per_px) * f * S
    slots = sum(_steps_for(n, quad) for n in per_px) if split else sum(per_px)
    terms = (3 if "bf16x2" in kn else 6) if split else 1
    issued_useful = executed * terms
    issued = 2.0 * B * Co * Cig4 * slots * f * S * terms
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    # weights drawn: every workgroup (n-tile, sample, m-tile) synthesises its rows for the taps ITS tile keeps
    m_tiles = max(1, li.get("m_tiles", 1))
    if pixel_major:
        n_bt = max(1, li.get("n_bt", 1))
        taps_tiles = n_bt * sum(per_px)
    elif row_tiles:
        n_bt = max(1, li.get("n_bt", 1))
        taps_tiles = n_bt * sum(nrow[ho] * gw for ho in range(Ho))
    else:
        taps_tiles = m_tiles * glob_taps
    draws = float(Co) * Cig4 * taps_tiles * S
    n_tiles = max(1, li.get("n_tiles", (Co // g + 63) // 64))
    x_el = 1.0
    for d in xs:
        x_el *= d
    out_el = 1.0
    for d in tuple(m._last["out_shape"]):
        out_el *= d
    if getattr(m, "post_pool", False):
        out_el /= 4.0
    staged = x_el * S * n_tiles * (1.0 if (kh == 1 and kw == 1) else 1.15)                     # every n-tile stages the patch again (+ halo overlap of 3x3 tiles)
    wel = float(w.numel())
    bytes_ = 8.0 * wel + (16.0 * wel if st.get("fused_kl") else 0.0) + 4.0 * x_el * (1 if st.get("shared_x") else S) + 4.0 * out_el * S * (2 if st.get("residual") else 1)
    t_mfma = issued / (peak * 1e12)
    t_valu = (draws * VALU_PER_WEIGHT * f + staged * VALU_PER_X) / (PEAK_VALU_TLANEOPS * 1e12)
    t_hbm = bytes_ / (PEAK_HBM_GBS * 1e9)
    return dict(nominal=nominal, executed=executed, effective=effective, issued=issued, issued_useful=issued_useful, pipe="bf16" if split else "f32", peak=peak,
                draws=draws, staged=staged, bytes=bytes_, t_mfma=t_mfma, t_valu=t_valu, t_hbm=t_hbm)


def _binding(t_mfma, t_valu, t_hbm):
    return max((("mfma", t_mfma), ("valu_synthesis", t_valu), ("hbm", t_hbm)), key=lambda kv: kv[1])


# ======================================================================================================== CPU baseline / parity
def _cgroup_cpu_quota():
    """CPUs granted by the cgroup's bandwidth limit (v2 cpu.max, v1 cfs quota), or 0 when unlimited / unreadable."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return 0 if q == "max" else max(1, int(int(q) / int(p) + 0.5))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return 0 if q <= 0 else max(1, int(q / p + 0.5))
    except (OSError, ValueError):
        return 0


def cpu_baseline(w, budget_s=12.0, budget_1t=8.0):
    """The oracle (kind 'port': the reference's ATen op sequence, oracle/bt_oracle.py) on this host: all of the
    process's cores (stated), then one thread."""
    from oracle import bt_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpu_quota()          # a 1-GPU slice of the host: the affinity mask may list every core of the machine
    usable = max(1, min(avail, quota)) if quota else avail
    torch.manual_seed(0)
    net = w["net"]()
    O.ref_dnn_to_bnn(net, w["btype"])
    net.eval()
    x = torch.randn(*w["x"])
    how = f"affinity mask {avail}, cgroup cpu quota {quota or 'unreadable'}"
    if "BT_CPU_THREADS" in os.environ:
        cores = max(1, int(os.environ["BT_CPU_THREADS"]))
    elif quota or usable <= 32:
        cores = usable
    else:
        # The mask lists more cores than a 1-GPU slice of the host is scheduled on and the quota cannot be read: pick the
        # thread count that actually runs fastest, on a probe that costs milliseconds (the largest Bayesian layer alone:
        # weight sampling + its contraction + KL -- the op mix of the whole sample).
        big = max((m for m in net.modules() if isinstance(m, O._RefBayes)), key=lambda m: m.mu_w.numel())
        xin = torch.randn(8, big.mu_w.shape[1], *([4, 4] if big.kind == "conv" else []))
        best = None
        for c in sorted({16, 32, 64, 128, usable} & set(range(1, usable + 1))):
            torch.set_num_threads(c)
            ts = []
            with torch.no_grad():
                for _ in range(3):
                    t0 = time.perf_counter()
                    big(xin)
                    big.kl_loss()
                    ts.append(time.perf_counter() - t0)
            if best is None or min(ts) < best[1]:
                best = (c, min(ts))
        cores = best[0]
        how += f"; thread count chosen by a probe over {{16, 32, 64, 128, {usable}}}: {cores} was fastest"

    def timed(threads, budget, warm):
        torch.set_num_threads(threads)
        with torch.no_grad():
            for _ in range(warm):
                net(x)
                O.ref_get_kl_loss(net)
            n, t0 = 0, time.perf_counter()
            while True:
                net(x)                      # one MC sample: forward with a fresh draw in every layer ...
                O.ref_get_kl_loss(net)      # ... plus the model's KL (dnn_to_bnn.py:157-165)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= budget or n >= 400:
                    break
        return n, dt
    n1, dt1 = timed(1, budget_1t, 1)
    n, dt = timed(cores, budget_s, 2)
    return dict(value=n / dt, unit="MC-samples/s", cores=cores, kind="port",
                sample=f"{n} sequential MC samples (forward + get_kl_loss) of the same workload, {dt:.1f} s, torch {torch.__version__} CPU, "
                       f"{cores} threads ({how})",
                one_thread=dict(value=n1 / dt1, unit="MC-samples/s", cores=1, sample=f"{n1} samples, {dt1:.1f} s"))


def parity_check(w, dev):
    """One MC sample through the same kernels (on-chip draws, unfused so every layer's own output is visible), each
    Bayesian layer's output and the logits against the CPU oracle ON THE SAME DRAWS (materialised from the counters).
    Reported as max|hip - ref| / max|ref| per tensor; worst layer, logits and KL."""
    from oracle import bt_oracle as O
    net = build_model(w, dev)
    torch.manual_seed(0)
    ref = w["net"]()
    O.ref_dnn_to_bnn(ref, w["btype"])
    H.fill_bayes_params(ref, 1)
    ref.eval()
    B = min(w["x"][0], 32)                                     # bounded: a CPU forward of the full batch adds nothing here
    x = torch.randn(B, *w["x"][1:], generator=torch.Generator().manual_seed(1))
    outs = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, (i[0].detach(), o.detach()))) for n, m in H.bayes_layers(net)]
    logits, kl = mc_forward(net, x.to(dev), 1, sample0=3)
    for h in hooks:
        h.remove()
    worst, worst_name = 0.0, ""
    with torch.no_grad():
        for (n, m), (_, rm) in zip(H.bayes_layers(net), H.bayes_layers(ref)):
            d = m.materialize_last_draw()
            rm.inject = {k: v[0].cpu() for k, v in d.items()}
            xi, yo = outs[n]
            want = rm(xi.cpu())                                 # the layer alone, on the input the HIP model fed it
            err = float((yo.cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))
            if err > worst:
                worst, worst_name = err, n
        want = ref(x)
        lerr = float((logits[0].cpu() - want).abs().max() / want.abs().max())
        klr = O.ref_get_kl_loss(ref)
    return dict(parity_max_rel_err=max(worst, lerr), worst_layer=worst_name, worst_layer_rel_err=worst, logits_rel_err=lerr,
                kl_rel_err=abs(float(kl) - float(klr)) / abs(float(klr)),
                how=f"1 sample, batch {B}, on-chip draws replayed through oracle/bt_oracle.py per layer and end to end; max|hip-ref|/max|ref|")


# ======================================================================================================== HBM traffic (child PMC passes)
def _rocprof_name(kn):
    """Our kernel-instance name -> a regex for the demangled name rocprofv3 reports (template arguments BN, BM, NP, NPW, XM)."""
    m = re.match(r"fused_split_kernel<(\d+),(\d+),bf16x(\d),\d terms,npw=(\d+),xm=(\d+)>", kn)
    if m:
        return r"fused_split_kernel<\s*%s,\s*%s,\s*%s,\s*%s,\s*%s,\s*(false|0)" % m.groups()
    if kn.startswith("fused_split_quad_kernel"):
        return r"fused_split_quad_kernel<"
    if kn.startswith("fused_split_direct_kernel"):
        return r"fused_split_direct_kernel<\s*(true|1)\s*>" if "resident" in kn else r"fused_split_direct_kernel<\s*(false|0)\s*>"
    return None


def pmc_traffic(argv_base, timeout_s=240):
    """FETCH_SIZE and WRITE_SIZE of this same command (fewer steps, the extra legs off), each in its own child `rocprofv3 --pmc`
    run (the two do not fit one pass; nothing but the counters is collected). Returns {kernel name: (fetch KB, write KB, launches)}
    or a string saying why not. Must run BEFORE this process initialises the GPU."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return "rocprofv3 not found"
    import csv
    import glob
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bt_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + argv_base
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return f"rocprofv3 --pmc {ctr} pass exceeded {timeout_s} s"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            tail = p.stdout.decode(errors="replace")[-300:].replace("\n", " | ")
            shutil.rmtree(d, ignore_errors=True)
            return f"rocprofv3 --pmc {ctr} pass failed (rc {p.returncode}): {tail}"
        for fn in files:
            for r in csv.DictReader(open(fn)):
                e = res.setdefault(r["Kernel_Name"], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_FETCH_SIZE": 0, "n_WRITE_SIZE": 0})
                e[ctr] += float(r["Counter_Value"])
                e["n_" + ctr] += 1
        shutil.rmtree(d, ignore_errors=True)
    return res


def traffic_object(pmc, dom_name, dom_bytes_alg, step_bytes_alg, launches_per_step):
    if not isinstance(pmc, dict):
        return dict(measured=False, reason=pmc or "disabled (--no-traffic, N > 1, or a secondary mode)")

    def agg(pred):
        f = sum(v["FETCH_SIZE"] for k, v in pmc.items() if pred(k))
        wr = sum(v["WRITE_SIZE"] for k, v in pmc.items() if pred(k))
        nf = sum(v["n_FETCH_SIZE"] for k, v in pmc.items() if pred(k))
        nw = sum(v["n_WRITE_SIZE"] for k, v in pmc.items() if pred(k))
        return f, wr, nf, nw
    pat = _rocprof_name(dom_name)
    out = dict(measured=True, how="two child `rocprofv3 --pmc` passes of this command (FETCH_SIZE, WRITE_SIZE; 3 steps, graph replay), started before this "
               "process touched the GPU; corrected per MI355X_MICROARCH.md (HBM): FETCH_SIZE x 2 (16-byte-per-lane streaming reads are tallied at half), WRITE_SIZE as is; counter unit KB. "
               "The guide calibrates the x2 for 16-byte-per-lane reads only: the direct 1x1 kernels fetch x with 4-byte-per-lane loads, for which the raw figure may be the right one -- both are given")
    f, wr, nf, nw = agg(lambda k: "fused_" in k)
    if nf and nw:
        out["fused_launch_avg_bytes"] = round((2.0 * f / nf + wr / nw) * 1024.0)
        out["fused_launch_avg_algorithmic_bytes"] = round(step_bytes_alg / max(1, launches_per_step))
        out["ratio_to_algorithmic"] = round(out["fused_launch_avg_bytes"] / max(1.0, out["fused_launch_avg_algorithmic_bytes"]), 3)
    if pat:
        f, wr, nf, nw = agg(lambda k: re.search(pat, k) is not None)
        if nf and nw:
            out["kernel_bytes_per_launch"] = round((2.0 * f / nf + wr / nw) * 1024.0)
            out["kernel_fetch_kb_raw"], out["kernel_write_kb_raw"] = round(f / nf, 1), round(wr / nw, 1)
            out["kernel_algorithmic_bytes_per_launch"] = round(dom_bytes_alg)
    return out


# ======================================================================================================== training line
def train_bench(args, w, dev, world, rank):
    """Training throughput of the drop-in (reference loop: examples/main_bayesian_cifar_dnn2bnn.py:402-420 with one MC sample per
    step): forward (fused HIP kernels) + get_kl_loss + cross-entropy + backward (HIP dgrad / wgrad with on-chip regeneration)
    + SGD step. Each rank trains its own replica on its own synthetic batch (no gradient all-reduce: the path under test is the
    layer kernels)."""
    from bayesian_torch_amd.models.dnn_to_bnn import get_kl_loss
    net = build_model(w, dev).train()
    torch.manual_seed(rank)
    x = torch.randn(*w["x"], device=dev)
    B = x.shape[0]
    ncls = 1000 if args.workload == "cfg5" else 10
    y = torch.randint(0, ncls, (B,), device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    rng.set_mode("philox")
    rng.manual_seed(0)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(net(x), y) + get_kl_loss(net) / B
        loss.backward()
        opt.step()
        return loss
    if not args.no_graph:      # the whole step as one HIP graph launch (mc.TrainGraph); --no-graph: ~400 eager launches per step
        from bayesian_torch_amd.mc import TrainGraph
        tg = TrainGraph(net, opt, lambda m, out, yy: torch.nn.functional.cross_entropy(out, yy) + get_kl_loss(m) / B, x, y, warmup=max(2, args.warmup))
        step = tg.step
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert torch.isfinite(loss)
    if rank == 0:
        print(json.dumps(dict(metric="training MC-samples/sec (forward+KL+backward+SGD step, 1 MC sample per step), " + w["desc"],
                              value=round(world * args.steps / dt, 2), unit="MC-samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                              ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
                              data="synthetic", config=dict(workload=w["desc"], batch=B, mc_samples_per_step=1, optimizer="SGD momentum 0.9",
                                                            backward="HIP dgrad + wgrad kernels (one launch per layer), draws regenerated on chip; the KL term's gradient is taken inside the weight-gradient pass (bt_conv2d_bwd_kl), its value by one bt_kl_normal launch per model",
                                                            launch="eager" if args.no_graph else "hip graph replay (mc.TrainGraph)"))))
    if world > 1:
        dist.destroy_process_group()


# ======================================================================================================== main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    w = WORKLOADS[args.workload]
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, argv)
    if args.plumbing_only:
        return plumbing_only(args, w)

    # HBM traffic of this same command: two child PMC passes, before this process initialises the GPU (N = 1, headline mode only)
    pmc = None
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1
    if single and not (args.no_traffic or args.no_roofline or args.train or os.environ.get("BT_BENCH_CHILD")):
        os.environ["BT_BENCH_CHILD"] = "1"
        base = ["--workload", args.workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-parity", "--no-extras", "--no-traffic"]
        if args.samples:
            base += ["--samples", str(args.samples)]
        if args.no_fuse:
            base += ["--no-fuse"]
        if args.no_graph:
            base += ["--no-graph"]
        pmc = pmc_traffic(base)
        del os.environ["BT_BENCH_CHILD"]

    world, rank, local = init_ranks()
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    if args.train:
        train_bench(args, w, dev, world, rank)
        return 0
    scaling = args.scaling or w["scaling"]
    if scaling == "strong":
        S_total = args.samples or w["S_total"]
        if S_total < world:
            print(f"bench.py: --scaling strong needs at least one sample per rank ({S_total} < {world})", file=sys.stderr)
            return 2
        first, S = mc_dist.shard(S_total, rank, world)
    else:
        S = args.samples or w["S"]
        S_total, first = S * world, rank * S
    # A rank's S samples run as `reps` launches of S_launch samples each (activations of S_launch samples are resident at a time):
    # 1 for every weak-scaling default; strong scaling on fewer ranks than the job was sized for walks its larger share in pieces
    # of the workload's per-GPU size (cfg5 on one GPU: 128 samples = 8 x 16).
    reps = max(1, -(-S // w["S"])) if scaling == "strong" else 1
    while S % reps:
        reps += 1
    S_launch = S // reps
    net = build_model(w, dev)
    fused = (not args.no_fuse) and hasattr(net, "layer1")
    if not args.no_fuse and not hasattr(net, "layer1"):      # an MLP: the ReLU behind a Linear layer goes into its output stage
        from bayesian_torch_amd.fuse import fold_relu
        fold_relu(net)
    if fused:
        H.fuse_inference(net)      # BN(eval)/ReLU/residual add folded into the conv kernels' output stage
    torch.manual_seed(0)
    x = torch.randn(*w["x"], device=dev)
    B = x.shape[0]
    rng.set_mode("philox")
    rng.manual_seed(0)

    def eager_step(i):
        packed = None
        for r in range(reps):
            logits, kl = mc_forward(net, x, S_launch, sample0=i * S_total + first + r * S_launch, with_kl=True)
            p = BF.mc_epilogue(logits.reshape(S_launch, B, -1))
            packed = p if packed is None else packed + p
        buf = mc_dist.finish_pack(packed, kl, world)
        mc_dist.reduce_packed(buf)
        return buf

    # Default: the step's kernels (pack check, 21 fused forwards, epilogue) are captured once in a HIP graph; the draw counter
    # lives on the device and every replay draws at fresh coordinates. The rank's sample ids are fixed at capture and the
    # per-step freshness comes from the call counter. The packed all-reduce stays outside the graph.
    graph = None
    if not args.no_graph:
        try:
            from bayesian_torch_amd.mc import McGraph
            graph = McGraph(net, x, S_launch, sample0=first, with_kl=True, epilogue=True)
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation; report and fall back to eager launches
            print(f"bench.py: HIP graph capture failed ({type(e).__name__}: {e}); using eager launches", file=sys.stderr)
            graph = None
    used_graph = graph is not None

    def graph_step(i, g=None):
        packed = None
        for r in range(reps):             # every replay draws at fresh RNG coordinates (the device-side call word advances)
            _, kl, p = (g or graph).replay()
            packed = p if reps == 1 else (p.clone() if packed is None else packed + p)
        buf = mc_dist.finish_pack(packed, kl, world)
        mc_dist.reduce_packed(buf)
        return buf

    step = graph_step if graph is not None else eager_step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warm, steps):
        for i in range(warm):
            fn(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            buf = fn(warm + i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt, buf

    dt, buf = timed(step, args.warmup, args.steps)
    res = mc_dist.unpack(buf, B, (buf.numel() - 1 - B) // (2 * B), S_total)
    assert torch.isfinite(res["kl"]).all()

    # ---- roofline pass: same steps, an event pair around every fused-forward launch --------------------------
    roof = None
    flip = w["btype"] == "Flipout"
    if not args.no_roofline:   # every rank runs the same steps (they contain the collective); only rank 0 keeps the numbers
        layers = H.bayes_layers(net)
        recs = {n: [] for n, _ in layers}
        handles = []
        for n, m in layers:
            def pre(mod, inp, n=n):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                recs[n].append([e, None])
            def post(mod, inp, out, n=n):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                recs[n][-1][1] = e
            handles += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
        for i in range(args.steps):
            eager_step(args.warmup + args.steps + i)   # eager: module hooks do not run inside a captured graph
        torch.cuda.synchronize()
        for h in handles:
            h.remove()
        barrier()
        table, kern = [], {}
        keys = ("nominal", "executed", "effective", "issued", "issued_useful", "draws", "bytes", "t_mfma", "t_valu", "t_hbm")
        tot = dict(ms=0.0, **{k: 0.0 for k in keys})
        for n, m in layers:
            d = sorted(a.elapsed_time(b) for a, b in recs[n])
            ms = d[len(d) // 2]          # median over the K launches: an event pair also spans host-side hiccups between record and launch
            lm = layer_model(m, flip, S_launch)
            kn = m._last.get("kernel", "?")
            bname, bt_ = _binding(lm["t_mfma"], lm["t_valu"], lm["t_hbm"])
            table.append(dict(layer=n, kernel=kn, ms=ms, pipe=lm["pipe"], gflop_nominal=lm["nominal"] / 1e9, gflop_executed=lm["executed"] / 1e9,
                              gflop_effective=lm["effective"] / 1e9, tflops_fp32_equiv=lm["executed"] / ms / 1e9,
                              mfma_tflops_issued=lm["issued"] / ms / 1e9, frac_mfma=lm["issued_useful"] / (lm["peak"] * 1e12) / (ms * 1e-3),
                              us_mfma_bound=lm["t_mfma"] * 1e6, us_valu_bound=lm["t_valu"] * 1e6, us_hbm_bound=lm["t_hbm"] * 1e6, bound=bname,
                              frac_of_bound=bt_ / (ms * 1e-3), mb_algorithmic=lm["bytes"] / 1e6, weights_drawn=lm["draws"],
                              launch=m._last.get("launch"), x=list(m._last["x_shape"]), out=list(m._last["out_shape"])))
            k = kern.setdefault(kn, dict(ms=0.0, launches=0, peak=lm["peak"], pipe=lm["pipe"], **{q: 0.0 for q in keys}))
            k["ms"] += ms
            k["launches"] += 1
            tot["ms"] += ms
            for q in keys:
                k[q] += lm[q]
                tot[q] += lm[q]
        nl = len(layers)
        dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"])
        dsec = dom["ms"] * 1e-3
        d_mfma, d_hbm = dom["issued_useful"] / dsec / 1e12, dom["bytes"] / dsec / 1e9
        if dom["t_hbm"] > dom["t_mfma"]:
            head = dict(bound="hbm", achieved=round(d_hbm, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(d_hbm / PEAK_HBM_GBS, 4))
        else:
            head = dict(bound="mfma", achieved=round(d_mfma, 1), peak=dom["peak"], unit="TFLOP/s", frac=round(d_mfma / dom["peak"], 4))

        def group(v):
            bname, bt_ = _binding(v["t_mfma"], v["t_valu"], v["t_hbm"])
            sec = v["ms"] * 1e-3
            return dict(ms_per_step=round(v["ms"], 4), launches=v["launches"], pipe=v["pipe"], frac_mfma=round(v["issued_useful"] / (v["peak"] * 1e12) / sec, 4),
                        frac_mfma_issued=round(v["t_mfma"] / sec, 4), frac_valu_synthesis=round(v["t_valu"] / sec, 4), frac_hbm=round(v["t_hbm"] / sec, 4),
                        bound=bname, frac_of_bound=round(bt_ / sec, 4), tflops_fp32_equiv=round(v["executed"] / sec / 1e12, 2))
        tsec = tot["ms"] * 1e-3
        bind_t = sum(_binding(r["us_mfma_bound"], r["us_valu_bound"], r["us_hbm_bound"])[1] for r in table) * 1e-6
        roof = dict(**head, traffic=traffic_object(pmc if rank == 0 else None, dom_name, dom["bytes"] / dom["launches"], tot["bytes"], nl),
                    kernel=dom_name, kernel_pipe=dom["pipe"], kernel_share_of_launch_time=round(dom["ms"] / tot["ms"], 4), kernel_launches_per_step=dom["launches"],
                    kernel_avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    frac_issued=round(dom["t_mfma"] / dsec, 4), frac_hbm=round(dom["t_hbm"] / dsec, 4), frac_valu_synthesis=round(dom["t_valu"] / dsec, 4),
                    achieved_fp32_equiv=round(dom["executed"] / dsec / 1e12, 2),
                    step=dict(launches=nl, launch_ms=round(tot["ms"], 4), avg_launch_ms=round(tot["ms"] / nl, 4),
                              frac_mfma=round(sum(v["issued_useful"] / v["peak"] for v in kern.values()) / 1e12 / tsec, 4),
                              frac_mfma_issued=round(tot["t_mfma"] / tsec, 4), frac_valu_synthesis=round(tot["t_valu"] / tsec, 4), frac_hbm=round(tot["t_hbm"] / tsec, 4),
                              frac_of_binding_bounds=round(bind_t / tsec, 4),
                              achieved_fp32_equiv=round(tot["executed"] / tsec / 1e12, 2), achieved_fp32_equiv_nominal=round(tot["nominal"] / tsec / 1e12, 2),
                              achieved_fp32_equiv_effective=round(tot["effective"] / tsec / 1e12, 2),
                              flop_nominal=tot["nominal"], flop_executed=tot["executed"], flop_effective=tot["effective"], algorithmic_bytes=tot["bytes"]),
                    peaks=dict(bf16_mfma_tflops=PEAK_BF16_MFMA_TFLOPS, f32_mfma_tflops=PEAK_F32_MFMA_TFLOPS, hbm_gbs=PEAK_HBM_GBS, valu_tlaneops=round(PEAK_VALU_TLANEOPS, 2),
                               valu_instr_per_weight=VALU_PER_WEIGHT, valu_instr_per_staged_activation=VALU_PER_X),
                    note="frac = MFMA FLOPs of the six bf16 piece products (6 x executed fp32-equivalent FLOPs; executed = active taps only, channels padded to the "
                         "kernel's quad) x S samples per launch / event-measured launch time (per layer: median over the K steps) / the dense peak of the pipe the kernel "
                         "issues on; frac_issued also counts the K slots the schedule pads. *_fp32_equiv figures are throughput in the reference's arithmetic, never a fraction. "
                         "valu_synthesis is a modelled bound (instruction estimates in `peaks`), hbm uses algorithmic bytes (packed params 8 B + fused KL sweep 16 B per weight, "
                         "x once or per sample, out (+ residual) per sample).",
                    per_kernel={kn: group(v) for kn, v in kern.items()})
        if args.layers_json and rank == 0:
            with open(args.layers_json, "w") as f:
                json.dump(table, f, indent=1)
        for r in (table if rank == 0 else []):
            print(f"  {r['layer']:24s} {r['ms']*1e3:8.1f} us  mfma {r['frac_mfma']:5.2f}  bound {r['bound']:14s} {r['frac_of_bound']:5.2f}  {r['tflops_fp32_equiv']:7.2f} TF/s fp32-eq  {r['kernel']}", file=sys.stderr)

    # ---- extras (rank 0 of a 1-GPU run): pack rebuild inside the step; the reference harness's own loop -----------------------------
    extras = {}
    if world == 1 and not args.no_extras:
        try:
            k2 = max(3, min(args.steps, 10))
            if used_graph:
                from bayesian_torch_amd.mc import McGraph
                g2 = McGraph(net, x, S_launch, sample0=first, with_kl=True, epilogue=True, force_pack=True)
                dt2, _ = timed(lambda i: graph_step(i, g2), 2, k2)
                del g2
            else:
                from bayesian_torch_amd.mc import sync_model_packs

                def forced(i):
                    for _, m in H.bayes_layers(net):
                        m.invalidate_pack()
                    return eager_step(i)
                dt2, _ = timed(forced, 2, k2)
            extras["value_with_pack_rebuild"] = dict(value=round(S_total * k2 / dt2, 2), unit="MC-samples/s", ms_per_step=round(dt2 / k2 * 1e3, 4), steps=k2,
                                                     what="the same step with every layer's (mu, softplus(rho)) pack rebuilt inside it (bt_pack_sync with force): what a step costs "
                                                          "right after a parameter update; the headline step contains the pack CHECK (fingerprint sweep), not the rebuild")
        except Exception as e:  # noqa: BLE001
            extras["value_with_pack_rebuild"] = dict(value=None, error=f"{type(e).__name__}: {e}")
        try:
            from bayesian_torch_amd.models.dnn_to_bnn import get_kl_loss
            graph = None
            torch.cuda.empty_cache()
            net_u = build_model(w, dev)          # converted by dnn_to_bnn, BatchNorm / ReLU / add as the separate torch modules they are

            def caller_loop(i):
                with torch.no_grad():
                    outs, kls = [], []
                    for _ in range(S_launch):     # examples/main_bayesian_cifar_dnn2bnn.py:402-410 / 541-545
                        outs.append(net_u(x))
                        kls.append(get_kl_loss(net_u))
                    return torch.stack(outs).mean(0), torch.stack(kls).mean()
            k3 = 3
            dt3, _ = timed(caller_loop, 1, k3)
            extras["eager_caller_loop"] = dict(value=round(S_launch * k3 / dt3, 2), unit="MC-samples/s", ms_per_step=round(dt3 / k3 * 1e3, 3), steps=k3,
                                               what=f"the reference harness's loop unchanged -- for s in range({S_launch}): model(x); get_kl_loss(model) -- on the drop-in layers: one fused "
                                                    "launch + a pack check per layer call, no MC batching, no HIP graph, BatchNorm/ReLU/add as separate torch modules")
            del net_u
        except Exception as e:  # noqa: BLE001
            extras["eager_caller_loop"] = dict(value=None, error=f"{type(e).__name__}: {e}")

    parity = None
    if rank == 0 and not args.no_parity:
        parity = parity_check(w, dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w)

    if rank == 0:
        total = S_total * args.steps
        metric = "MC-samples/sec (forward+KL), Bayesian-ResNet18 CIFAR batch=128"      # BASELINE.json (cfg3, and cfg4 = its Flipout variant)
        if args.workload in ("cfg2", "cfg5"):
            metric = "MC-samples/sec (forward+KL), " + w["desc"]
        kernels = sorted({m._last.get("kernel", "?") for _, m in H.bayes_layers(net) if m._last})
        split = any("bf16" in k for k in kernels)
        line = dict(metric=metric, value=round(total / dt, 2), unit="MC-samples/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4),
                    higher_is_better=True, scaling=scaling, vs_baseline=None,
                    dtype="f32 (bf16x3 split operands, 6 product terms, f32 accumulate)" if split else "f32", data="synthetic",
                    config=dict(workload=w["desc"], mc_samples_per_gpu_per_step=S, global_samples_per_step=S_total, samples_per_launch=S_launch, launches_of_the_model_per_step=reps, batch=B,
                                rng="on-chip philox", parallelism=f"mc{world}", kl="fused into forward kernels (computed once per launch: it does not depend on the sample)",
                                scaling_note=("strong: %d samples in total, sharded contiguously over the %d rank(s)" % (S_total, world)) if scaling == "strong"
                                else ("weak: %d samples per rank" % S),
                                output_stage="bn+relu+residual (+ the stem max-pool) folded into the conv kernels" if fused else "separate torch modules",
                                launch="hip graph replay" if used_graph else "eager",
                                parameter_pack="tap-major (mu, softplus(rho)) copies live in persistent buffers; every step verifies them on the device (bt_pack_sync: a "
                                               "fingerprint sweep of all (mu, rho), inside the timed region) and rebuilds only layers whose parameters changed"),
                    roofline=roof, cpu_baseline=cpu, parity=parity, **extras)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
