#!/usr/bin/env python3
"""Headline benchmark: MC-samples/sec (forward+KL), Bayesian-ResNet18 CIFAR batch=128 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over one synthetic batch: S MC samples (default 32 per GPU,
BASELINE.json cfg3) of the dnn_to_bnn-converted ResNet18 forward, every Bayesian layer's KL (fused
into the forward kernels), the softmax/entropy MC epilogue, and -- for N > 1 -- the one packed
all-reduce.  Inputs and parameters are resident in HBM before the timed region.  Default scaling is
weak (each rank draws its own S samples, global sample ids rank*S ...: the job does N*S samples per
step); `--scaling strong` fixes the global sample count (cfg5's contract: 128 samples over 8 GPUs)
and shards it with mc_dist.shard.

Prints ONE JSON line (rank 0).
  roofline      measured in a second pass of the same K steps with a HIP-event pair around every fused
                launch (events on the launch stream). FLOPs are counted three ways per layer: nominal
                (2*B*Co*Ho*Wo*K, padding taps included), executed (what the kernel issues: only taps
                that can ever meet data, channels padded to the kernel's quad) and effective (products
                with real input pixels only). `frac` uses EXECUTED flops; no per-layer figure can
                exceed the peak. The dominant kernel instance (largest share of the launch time) is
                reported with its own fraction.
  parity        outside the timed region: one sample through the same kernels (on-chip draws), every
                layer's output and the logits against the CPU oracle on the same draws.
  cpu_baseline  the oracle -- the reference's exact ATen op sequence -- on ALL of this process's host
                cores (count stated) for a bounded sample, plus a 1-thread line.
"""
import argparse
import json
import os
import sys
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

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (no xf32 on gfx950)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
         "moped_enable": False, "moped_delta": 0.5}

WORKLOADS = {
    "cfg3": dict(desc="cfg3: Bayesian-ResNet18 via dnn_to_bnn (Conv2dReparameterization), CIFAR 3x32x32, batch=128",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Reparameterization", S=32, S_total=32),
    "cfg4": dict(desc="cfg4: Bayesian-ResNet18 Flipout (Conv2dFlipout/LinearFlipout), CIFAR 3x32x32, batch=128",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Flipout", S=32, S_total=32),
    "cfg5": dict(desc="cfg5: Bayesian-ResNet50 Reparameterization, ImageNet 3x224x224, batch=256 (128 MC samples / 8 GPUs = 16 per GPU)",
                 net=lambda: H.resnet50(1000, 64), x=(256, 3, 224, 224), btype="Reparameterization", S=16, S_total=128),
    "cfg2": dict(desc="cfg2: MLP 3072->512->10 (LinearReparameterization), batch=256",
                 net=lambda: H.mlp((3072, 512, 10)), x=(256, 3072), btype="Reparameterization", S=8, S_total=8),
}


def build_model(w, dev, seed=0):
    torch.manual_seed(seed)
    net = w["net"]()
    dnn_to_bnn(net, dict(PRIOR, type=w["btype"]))
    H.fill_bayes_params(net, 1)
    return net.to(dev).eval()


def layer_flops(m, flip):
    """(nominal, executed, effective) FLOPs of ONE sample through a Bayesian layer (x2 for Flipout's two contractions).
    nominal: 2*B*Co*Ho*Wo*(Ci/g)*kh*kw. executed: the kernels' tap schedule -- a tap that can only ever meet zero padding
    is dropped (for the whole layer, or per output pixel on pixel-major tiles: 2..4-pixel outputs with padding), and the
    channel axis is padded to the kernel's quad. effective: products whose input pixel exists."""
    w = m._w("mu")
    xs, os_ = m._last["x_shape"], m._last["out_shape"]
    B, Co = os_[0], os_[1]
    f = 2 if flip else 1
    if m._kind == "linear":
        fl = 2.0 * B * Co * w.shape[1] * f
        return fl, fl, fl
    cd = m._conv_desc()
    (sh, sw), (ph, pw), (dh, dw), g = cd["stride"], cd["padding"], cd["dilation"], cd["groups"]
    Cig = w.shape[1]
    kh, kw = (1, w.shape[2]) if w.dim() == 3 else (w.shape[2], w.shape[3])
    Hh, Ww = (1, xs[2]) if len(xs) == 3 else (xs[2], xs[3])
    Ho, Wo = BF.conv_out_hw(Hh, Ww, kh, kw, sh, sw, ph, pw, dh, dw)
    rows = [[0 <= ho * sh - ph + a * dh < Hh for a in range(kh)] for ho in range(Ho)]     # rows[ho][kh]: tap row meets data
    cols = [[0 <= wo * sw - pw + b * dw < Ww for b in range(kw)] for wo in range(Wo)]
    eff_taps = sum(sum(r) for r in rows) * sum(sum(c) for c in cols)                        # sum over pixels of in-bounds taps
    glob_taps = sum(any(r[a] for r in rows) for a in range(kh)) * sum(any(c[b] for c in cols) for b in range(kw))
    pixel_major = 2 <= Ho * Wo <= 4 and (ph > 0 or pw > 0)                                    # bt_fused_api.hip
    Cig4 = (Cig + 3) // 4 * 4
    nominal = 2.0 * B * Co * Ho * Wo * Cig * kh * kw * f
    executed = 2.0 * B * Co * Cig4 * (eff_taps if pixel_major else Ho * Wo * glob_taps) * f
    effective = 2.0 * B * Co * Cig * eff_taps * f
    return nominal, executed, effective


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


def _offline_traffic(workload):
    """The newest committed PMC summary of the default command (profiles/r*_bench_summary.json), LABELLED as a file read: it was
    not measured by this run. FETCH_SIZE / WRITE_SIZE in the counters' own units (KB) per fused launch, averaged over the step's
    launches; the guide's x2 correction for 16-byte fetch streams is NOT applied."""
    if workload != "cfg3":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_bench_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return dict(source=os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__))) + " (offline rocprofv3 PMC passes; not measured in this run)",
                    fetch_kb_per_launch=round(d["fetch"]["fused"]["per_launch"], 1), write_kb_per_launch=round(d["write"]["fused"]["per_launch"], 1))
    except Exception:
        return None


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
                                                            backward="HIP dgrad/wgrad kernels, draws regenerated on chip",
                                                            launch="eager" if args.no_graph else "hip graph replay (mc.TrainGraph)"))))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=0, help="MC samples per GPU per step (weak) / in total (strong); default: the workload's")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: S samples per rank; strong: a fixed global sample count sharded over the ranks (mc_dist.shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a captured HIP graph")
    ap.add_argument("--no-fuse", action="store_true", help="keep BatchNorm/ReLU/add as separate torch modules")
    ap.add_argument("--layers-json", default="", help="write the per-layer roofline table here")
    ap.add_argument("--train", action="store_true", help="secondary line: one TRAINING step per MC sample (forward + KL + HIP backward + SGD step, S = 1, BatchNorm in train mode)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BT_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path with ranks sharing one GPU
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    elif args.gpus > 1:
        print("bench.py: --gpus > 1 must be launched through torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    w = WORKLOADS[args.workload]
    if args.train:
        return train_bench(args, w, dev, world, rank)
    if args.scaling == "strong":
        S_total = args.samples or w["S_total"]
        first, S = mc_dist.shard(S_total, rank, world)
        if S_total < world:
            print(f"bench.py: --scaling strong needs at least one sample per rank ({S_total} < {world})", file=sys.stderr)
            sys.exit(2)
    else:
        S = args.samples or w["S"]
        S_total, first = S * world, rank * S
    net = build_model(w, dev)
    fused = (not args.no_fuse) and hasattr(net, "layer1")
    if fused:
        H.fuse_inference(net)      # BN(eval)/ReLU/residual add folded into the conv kernels' output stage
    torch.manual_seed(0)
    x = torch.randn(*w["x"], device=dev)
    B = x.shape[0]
    rng.set_mode("philox")
    rng.manual_seed(0)

    def eager_step(i):
        logits, kl = mc_forward(net, x, S, sample0=i * S_total + first, with_kl=True)
        packed = BF.mc_epilogue(logits.reshape(S, B, -1))
        buf = mc_dist.finish_pack(packed, kl, world)
        mc_dist.reduce_packed(buf)
        return buf

    # Default: the step's kernels (21 fused forwards, pooling, epilogue) are captured once in a HIP graph; the draw counter
    # lives on the device and every replay draws at fresh coordinates. The rank's sample ids are fixed at capture and the
    # per-step freshness comes from the call counter. The packed all-reduce stays outside the graph.
    graph = None
    if not args.no_graph:
        try:
            from bayesian_torch_amd.mc import McGraph
            graph = McGraph(net, x, S, sample0=first, with_kl=True, epilogue=True)
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation; report and fall back to eager launches
            print(f"bench.py: HIP graph capture failed ({type(e).__name__}: {e}); using eager launches", file=sys.stderr)
            graph = None

    def graph_step(i):
        _, kl, packed = graph.replay()
        buf = mc_dist.finish_pack(packed, kl, world)
        mc_dist.reduce_packed(buf)
        return buf

    step = graph_step if graph is not None else eager_step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        buf = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
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
        bf16 = any("bf16" in m._last.get("kernel", "") for _, m in layers)
        table, kern = [], {}
        tot = dict(ms=0.0, nom=0.0, exe=0.0, eff=0.0)
        for n, m in layers:
            d = sorted(a.elapsed_time(b) for a, b in recs[n])
            ms = d[len(d) // 2]          # median over the K launches: an event pair also spans host-side hiccups between record and launch
            nom, exe, eff = (v * S for v in layer_flops(m, flip))
            kn = m._last.get("kernel", "?")
            table.append(dict(layer=n, kernel=kn, ms=ms, gflop_nominal=nom / 1e9, gflop_executed=exe / 1e9, gflop_effective=eff / 1e9,
                              tflops_executed=exe / ms / 1e9, x=list(m._last["x_shape"]), out=list(m._last["out_shape"])))
            k = kern.setdefault(kn, dict(ms=0.0, exe=0.0, launches=0))
            k["ms"] += ms
            k["exe"] += exe
            k["launches"] += 1
            tot["ms"] += ms
            tot["nom"] += nom
            tot["exe"] += exe
            tot["eff"] += eff
        nl = len(layers)
        peak = PEAK_F32_MFMA_TFLOPS
        ach = tot["exe"] / tot["ms"] / 1e9
        dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"])
        dom_ach = dom["exe"] / dom["ms"] / 1e9
        over = [r["layer"] for r in table if r["tflops_executed"] > peak and not bf16]
        roof = dict(bound="mfma", achieved=round(ach, 3), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                    frac_effective=round(tot["eff"] / tot["ms"] / 1e9 / peak, 4), frac_nominal=round(tot["nom"] / tot["ms"] / 1e9 / peak, 4),
                    traffic=None, traffic_note="HBM bytes are measured offline with rocprofv3 PMC passes of this same command (tools/profile_bench.sh -> profiles/): PMC cannot run inside the process",
                    traffic_offline=_offline_traffic(args.workload),
                    kernel=dom_name, kernel_share_of_launch_time=round(dom["ms"] / tot["ms"], 4), kernel_launches_per_step=dom["launches"],
                    kernel_achieved=round(dom_ach, 3), kernel_frac=round(dom_ach / peak, 4), kernel_avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    launches_per_step=nl, avg_launch_ms=round(tot["ms"] / nl, 4),
                    flop_per_step=tot["nom"], flop_executed=tot["exe"], flop_effective=tot["eff"],
                    layers_above_peak=over,
                    bf16_pipe=(dict(note="split kernels issue 6 bf16 MFMAs per fp32-equivalent K16 step (odd tap counts: 10/9 more)",
                                    kernel_issue_tflops=round(6 * dom_ach, 1), peak=PEAK_BF16_MFMA_TFLOPS, kernel_frac=round(6 * dom_ach / PEAK_BF16_MFMA_TFLOPS, 4))
                               if "split" in dom_name else None),
                    note="fp32-equivalent FLOPs; frac = executed FLOPs (active taps only, channels padded to the kernel's quad) x S samples per launch / "
                         "event-measured launch time (per layer: median over the K steps) / the fp32-MFMA peak"
                         + ("; the contraction runs as bf16-split products on the bf16 matrix pipe with fp32 accumulation, so a figure above the fp32-MFMA peak is legitimate here" if bf16 else ""),
                    per_kernel={kn: dict(ms_per_step=round(v["ms"], 4), launches=v["launches"], tflops_executed=round(v["exe"] / v["ms"] / 1e9, 2)) for kn, v in kern.items()})
        if args.layers_json and rank == 0:
            with open(args.layers_json, "w") as f:
                json.dump(table, f, indent=1)
        for r in (table if rank == 0 else []):
            print(f"  {r['layer']:24s} {r['ms']*1e3:9.1f} us  exec {r['gflop_executed']:8.2f} GF  {r['tflops_executed']:7.2f} TF/s  {r['kernel']}", file=sys.stderr)

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
                    higher_is_better=True, scaling=args.scaling, vs_baseline=None,
                    dtype="f32 (bf16x3 split operands, 6 product terms, f32 accumulate)" if split else "f32", data="synthetic",
                    config=dict(workload=w["desc"], mc_samples_per_gpu_per_step=S, global_samples_per_step=S_total, batch=B,
                                rng="on-chip philox", parallelism=f"mc{world}", kl="fused into forward kernels (computed once per launch: it does not depend on the sample)",
                                output_stage="bn+relu+residual (+ the stem max-pool) folded into the conv kernels" if fused else "separate torch modules",
                                launch="hip graph replay" if graph is not None else "eager",
                                parameter_pack="tap-major (mu, softplus(rho)) copies are built once per parameter version, outside the timed region; "
                                               "in training they are rebuilt every step (one pass over the parameters)"),
                    roofline=roof, cpu_baseline=cpu, parity=parity)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
