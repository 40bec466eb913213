#!/usr/bin/env python3
"""Headline benchmark: MC-samples/sec (forward+KL), Bayesian-ResNet18 CIFAR batch=128 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over one synthetic batch: S MC samples (default 32 per GPU,
BASELINE.json cfg3) of the dnn_to_bnn-converted ResNet18 forward, every Bayesian layer's KL (fused
into the forward kernels), the softmax/entropy MC epilogue, and -- for N > 1 -- the one packed
all-reduce.  Inputs and parameters are resident in HBM before the timed region.  Weak scaling: each
rank draws its own S samples (global sample ids rank*S ...), so the job does N*S samples per step.

Prints ONE JSON line (rank 0).  `roofline` is measured in a second pass of the same K steps with a
HIP-event pair around every fused-forward launch (the events sit on the launch stream); `cpu_baseline`
times the oracle -- the reference's exact ATen op sequence -- on this host's cores for a bounded sample.
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
PRIOR = {"prior_mu": 0.0, "prior_sigma": 1.0, "posterior_mu_init": 0.0, "posterior_rho_init": -3.0,
         "moped_enable": False, "moped_delta": 0.5}

WORKLOADS = {
    "cfg3": dict(desc="cfg3: Bayesian-ResNet18 via dnn_to_bnn (Conv2dReparameterization), CIFAR 3x32x32, batch=128",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Reparameterization", S=32),
    "cfg4": dict(desc="cfg4: Bayesian-ResNet18 Flipout (Conv2dFlipout/LinearFlipout), CIFAR 3x32x32, batch=128",
                 net=lambda: H.resnet18(10, 64), x=(128, 3, 32, 32), btype="Flipout", S=32),
    "cfg5": dict(desc="cfg5: Bayesian-ResNet50 Reparameterization, ImageNet 3x224x224, batch=256 (128 MC samples / 8 GPUs = 16 per GPU)",
                 net=lambda: H.resnet50(1000, 64), x=(256, 3, 224, 224), btype="Reparameterization", S=16),
    "cfg2": dict(desc="cfg2: MLP 3072->512->10 (LinearReparameterization), batch=256",
                 net=lambda: H.mlp((3072, 512, 10)), x=(256, 3072), btype="Reparameterization", S=8),
}


def build_model(w, dev, seed=0):
    torch.manual_seed(seed)
    net = w["net"]()
    dnn_to_bnn(net, dict(PRIOR, type=w["btype"]))
    H.fill_bayes_params(net, 1)
    return net.to(dev).eval()


def layer_flops(m, x_shape, out_shape, flip):
    """Nominal FLOPs of one sample through a Bayesian layer: 2*B*Co*Ho*Wo*(Ci/g)*kh*kw (x2 for Flipout)."""
    w = m._w("mu")
    k = w[0].numel()
    outs = 1
    for d in out_shape:
        outs *= d
    return 2.0 * outs * k * (2 if flip else 1)


def cpu_baseline(w, budget_s=12.0):
    """The oracle (kind 'port': the reference's ATen op sequence, oracle/bt_oracle.py) on this host."""
    from oracle import bt_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a 1-GPU slice of the host owns about 16 cores; more ATen threads than that only oversubscribe it
    cores = max(1, min(avail, int(os.environ.get("BT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = w["net"]()
    O.ref_dnn_to_bnn(net, w["btype"])
    net.eval()
    x = torch.randn(*w["x"])
    with torch.no_grad():
        for _ in range(2):
            net(x)
            O.ref_get_kl_loss(net)
        n, t0 = 0, time.perf_counter()
        while True:
            net(x)                      # one MC sample: forward with a fresh draw in every layer ...
            O.ref_get_kl_loss(net)      # ... plus the model's KL (dnn_to_bnn.py:157-165)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or n >= 400:
                break
    return dict(value=n / dt, unit="MC-samples/s", cores=cores, kind="port",
                sample=f"{n} sequential MC samples (forward + get_kl_loss) of the same workload, {dt:.1f} s, torch {torch.__version__} CPU, {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=0, help="MC samples per GPU per step (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a captured HIP graph")
    ap.add_argument("--no-fuse", action="store_true", help="keep BatchNorm/ReLU/add as separate torch modules")
    ap.add_argument("--layers-json", default="", help="write the per-layer roofline table here")
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
    w = WORKLOADS[args.workload]
    S = args.samples or w["S"]
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
        logits, kl = mc_forward(net, x, S, sample0=(i * world + rank) * S, with_kl=True)
        packed = BF.mc_epilogue(logits.reshape(S, B, -1))
        buf = mc_dist.finish_pack(packed, kl, world)
        mc_dist.reduce_packed(buf)
        return buf

    # Default: the step's kernels (21 fused forwards, pooling, epilogue) are captured once in a HIP graph; the draw counter
    # lives on the device and advances per replay. The rank's sample ids are fixed at capture (sample0 = rank*S) and the
    # per-step freshness comes from the call counter. The packed all-reduce stays outside the graph.
    graph = None
    if not args.no_graph:
        try:
            from bayesian_torch_amd.mc import McGraph
            graph = McGraph(net, x, S, sample0=rank * S, with_kl=True, epilogue=True)
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
    res = mc_dist.unpack(buf, B, (buf.numel() - 1 - B) // (2 * B), S * world)
    assert torch.isfinite(res["kl"]).all()

    # ---- roofline pass: same steps, an event pair around every fused-forward launch --------------------------
    roof = None
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
        table, tot_ms, tot_fl = [], 0.0, 0.0
        barrier()
        flip = w["btype"] == "Flipout"
        for n, m in layers:
            d = sorted(a.elapsed_time(b) for a, b in recs[n])
            ms = d[len(d) // 2]          # median over the K launches: an event pair also spans host-side hiccups between record and launch
            fl = layer_flops(m, m._last["x_shape"], m._last["out_shape"], flip) * S
            table.append(dict(layer=n, ms=ms, gflop=fl / 1e9, tflops=fl / ms / 1e9, x=list(m._last["x_shape"]), out=list(m._last["out_shape"])))
            tot_ms += ms
            tot_fl += fl
        nl = len(layers)
        ach = tot_fl / tot_ms / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01r_pmc_traffic.json")   # measured offline: PMC passes cannot run inside this process
        if args.workload == "cfg3" and S == 32 and fused and world == 1 and os.path.exists(tfile):
            traffic = json.load(open(tfile))["traffic_bytes_per_launch"]
        roof = dict(bound="mfma", achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    traffic=traffic, kernel="bt::fused_fast_kernel / bt::fused_fwd_kernel (fp32 MFMA implicit GEMM, all tile instances)",
                    launches_per_step=nl, avg_launch_ms=round(tot_ms / nl, 4), flop_per_step=tot_fl,
                    note="nominal FLOPs (2*B*Co*Ho*Wo*K per sample, padding taps included) x S samples per launch / event-measured launch time (per layer: median over the K steps)")
        if args.layers_json and rank == 0:
            with open(args.layers_json, "w") as f:
                json.dump(table, f, indent=1)
        for r in (table if rank == 0 else []):
            print(f"  {r['layer']:24s} {r['ms']*1e3:9.1f} us  {r['gflop']:8.2f} GF  {r['tflops']:7.2f} TF/s", file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w)

    if rank == 0:
        total = S * world * args.steps
        metric = "MC-samples/sec (forward+KL), Bayesian-ResNet18 CIFAR batch=128"      # BASELINE.json (cfg3, and cfg4 = its Flipout variant)
        if args.workload in ("cfg2", "cfg5"):
            metric = "MC-samples/sec (forward+KL), " + w["desc"]
        line = dict(metric=metric, value=round(total / dt, 2), unit="MC-samples/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=w["desc"], mc_samples_per_gpu_per_step=S, global_samples_per_step=S * world, batch=B,
                                rng="on-chip philox", parallelism=f"mc{world}", kl="fused into forward kernels",
                                output_stage="bn+relu+residual (+ the stem max-pool) folded into the conv kernels" if fused else "separate torch modules",
                                launch="hip graph replay" if graph is not None else "eager"),
                    roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
