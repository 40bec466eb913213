"""Monte-Carlo batching: S weight draws of a model in ONE pass.

The reference evaluates MC samples with a sequential Python loop that re-runs the whole
model per sample (examples/main_bayesian_cifar_dnn2bnn.py:402-408, 542-545).  Here the
sample axis is folded into the batch axis: inside ``mc_samples(S, batch)`` every Bayesian
layer treats its input as ``[S*B, ...]`` (sample-major; or ``[B, ...]`` for an input shared by
all samples, e.g. the first layer) and its kernel uses a different weight draw per sample
chunk -- one launch per layer for all S samples, (mu, rho) fetched once and re-read from L2.
Deterministic layers in between (BatchNorm in eval mode, ReLU, pooling, residual adds) are
per-example, so they simply see a larger batch.
"""
import contextlib
import threading

import torch

_tls = threading.local()


class McContext:
    def __init__(self, S, batch, sample0=0, collect_kl=False, call_base=None):
        self.S, self.batch, self.sample0 = int(S), int(batch), int(sample0)
        self.collect_kl = collect_kl
        self.kls = []            # per-layer 0-dim KL tensors in execution order (collect_kl)
        self.call_base = call_base  # device uint32 word (graph replay) or None
        self.synced = set()      # id() of the layers whose packs sync_model_packs has verified inside this context
        self.pack_event = None   # recorded on the side stream that verifies every layer but the first (sync_model_packs)
        self.late = set()        # id() of the layers verified there: the first of them to run makes the launch stream wait
        self.train_fused = False  # TrainGraph: forwards carry their KL term along (FusedForward's second output) ...
        self.live_layers = []    # layers holding a live KL tensor of this context (layers/_fused.py: _kl_live)
        self.deferred = None     # ... and (a list) hand their weight gradients over here, computed on the side stream: (layer, dmu, drho, device)


def current():
    return getattr(_tls, "ctx", None)


@contextlib.contextmanager
def mc_samples(S, batch, sample0=0, collect_kl=False, call_base=None):
    prev = current()
    ctx = McContext(S, batch, sample0, collect_kl, call_base)
    _tls.ctx = ctx
    try:
        yield ctx
    finally:
        _tls.ctx = prev


_side_streams = {}


def _side_stream(dev):
    st = _side_streams.get(dev)
    if st is None:
        st = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return st


def sync_model_packs(model, ctx=None, force=False, overlap=True):
    """Verify -- on the device, in the stream -- that every Bayesian layer's packed (mu, softplus(rho)) copy still matches its
    parameters, and rebuild the ones that do not (bt_pack_sync: one fingerprint launch + one conditional pack launch per 64
    layers and device, instead of two launches per layer). Inside ``ctx`` the layers then skip their own check.
    With a context the sweep is split: the FIRST layer is verified in the launch stream (a few KB), all the others on a side
    stream that forks here and is joined by the first of them to run (``join_packs``) -- the HBM-bound sweep of the whole model
    (ResNet18: 89 MB, ~30 us) then runs beside the first layer's kernel instead of in front of it. Under graph capture the fork
    and the join become edges of the graph."""
    from . import functional as F
    by_dev = {}
    for m in model.modules():
        if hasattr(m, "_pack_segment") and m._w("mu").is_cuda:
            if force:
                m._pack_force = True
            by_dev.setdefault(m._w("mu").device, []).append(m)
    for dev, layers in by_dev.items():
        if ctx is None or not overlap or len(layers) < 2 or len(by_dev) > 1:
            F.pack_sync([m._pack_segment() for m in layers], owner=("model", id(model)))
        else:
            F.pack_sync([layers[0]._pack_segment()], owner=("model", id(model), "first"))
            cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                F.pack_sync([m._pack_segment() for m in layers[1:]], owner=("model", id(model), "rest"))
                ctx.pack_event = torch.cuda.Event()
                ctx.pack_event.record(side)
            ctx.late.update(id(m) for m in layers[1:])
        if ctx is not None:
            ctx.synced.update(id(m) for m in layers)


def join_packs(ctx):
    """Make the current stream wait for the side-stream pack check of ``ctx`` (once; no-op when there is none)."""
    if ctx is not None and ctx.pack_event is not None:
        torch.cuda.current_stream().wait_event(ctx.pack_event)
        ctx.pack_event = None


def mc_forward(model, x, S, sample0=0, with_kl=True):
    """S MC samples of ``model`` on batch ``x`` -> (logits[S, B, ...], kl or None).

    Semantics of ``for s in range(S): out_s = model(x); kl = get_kl_loss(model)`` of the
    reference's loop; KL does not depend on the sample and is produced by the layers' own
    forward kernels (fused) when ``with_kl``."""
    B = x.shape[0]
    with torch.no_grad(), mc_samples(S, B, sample0, collect_kl=with_kl) as ctx:
        sync_model_packs(model, ctx)
        out = model(x)
        join_packs(ctx)          # (a model whose later layers never ran: the side stream still has to come back)
    if isinstance(out, tuple):   # native Bayesian models return (logits, kl_sum)
        out = out[0]
    out = out.reshape(S, B, *out.shape[1:])
    kl = None
    if with_kl and ctx.kls:
        kl = torch.stack(ctx.kls).sum()
    return out, kl


class McGraph:
    """``mc_forward`` (+ the MC epilogue) captured once in a HIP graph and replayed per batch: the ~25 kernel launches of a
    model become one graph launch.  The draws stay fresh: every fused kernel adds a device-side word (``call_base``,
    bt_rng.call_base_dev) to its baked-in ``call`` coordinate, and ``replay()`` sets that word so that the replay draws at the
    host counter's current position and then advances the host counter by the number of layer calls -- a replay consumes
    exactly the coordinates an eager ``mc_forward`` at that moment would have, so eager calls, replays and several graphs
    can interleave without ever reusing a draw.  Parameters are baked in by address; the layers' packed copies live in
    persistent buffers and the captured step begins with the device-side pack check (``sync_model_packs``), so a replay
    FOLLOWS parameter updates made in place (optimizer steps, ``.data`` writes, load_state_dict).  Re-capture only when a
    parameter tensor is replaced (``.to()``, a new nn.Parameter).  ``force_pack``: rebuild every pack on every replay (the
    benchmark's with-rebuild figure).
    ("capturing the whole model per sample in a HIP graph".)"""

    def __init__(self, model, x, S, sample0=0, with_kl=True, epilogue=True, force_pack=False):
        from . import functional as F, rng
        self.S, self.B = int(S), x.shape[0]
        self.x = x.clone()
        self.call_base = torch.zeros(1, dtype=torch.int32, device=x.device)

        def run():
            B = self.B
            with torch.no_grad(), mc_samples(self.S, B, sample0, collect_kl=with_kl, call_base=self.call_base) as ctx:
                sync_model_packs(model, ctx, force=force_pack)
                out = model(self.x)
                join_packs(ctx)
            out = out[0] if isinstance(out, tuple) else out
            logits = out.reshape(self.S, B, *out.shape[1:])
            kl = torch.stack(ctx.kls).sum() if (with_kl and ctx.kls) else None
            packed = F.mc_epilogue(logits.reshape(self.S, B, -1)) if epilogue else None
            return logits, kl, packed

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture: parameter packs, workspaces, LDS attributes
            run()
            c0 = rng.peek_call()
            run()
            self.calls_per_run = rng.peek_call() - c0
        torch.cuda.current_stream().wait_stream(side)
        self.call0 = rng.peek_call()           # the call coordinate baked into the first captured layer
        self._packs = [m._pack for m in model.modules() if hasattr(m, "_pack_segment")]   # the buffers whose addresses the graph bakes in
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logits, self.kl, self.packed = run()
        rng.set_call(self.call0)               # capture executes nothing: its coordinates are still unused

    def replay(self, x=None):
        """-> (logits [S, B, ...], kl, packed epilogue sums) -- tensors owned by the graph, overwritten by the next replay."""
        from . import rng
        if x is not None:
            self.x.copy_(x)
        c = rng.peek_call()
        d = (c - self.call0) & 0xFFFFFFFF          # the device adds the word mod 2^32; the tensor holds it as int32
        self.call_base.fill_(d - (1 << 32) if d >= (1 << 31) else d)
        self.graph.replay()
        rng.set_call(c + self.calls_per_run)
        return self.logits, self.kl, self.packed


def finish_deferred(ctx):
    """Join the side stream that computed the deferred weight gradients of a training step (autograd.FusedForward, opts["defer"]) and
    give them to their parameters -- ``p.grad`` exactly as autograd would have left it: assigned, or added to what is there."""
    if not ctx.deferred:
        return
    devs = {d for (_, _, _, d) in ctx.deferred}
    for dev in devs:
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
    for layer, gmu, grho, dev in ctx.deferred:
        cur = torch.cuda.current_stream(dev)
        for p, g in ((layer._w("mu"), gmu), (layer._w("rho"), grho)):
            g = g.view_as(p)
            g.record_stream(cur)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
    ctx.deferred.clear()


class TrainGraph:
    """One TRAINING step -- zero_grad, forward (fused kernels, one MC sample), ``loss_fn(model, out, y)``, backward (HIP dgrad /
    wgrad with the draws regenerated on chip), optimizer step -- captured once in a HIP graph and replayed per batch: the ~400
    kernel launches of a ResNet18 step become one graph launch (eager mode is launch-bound).  Fresh draws per step exactly as
    in McGraph: every forward AND backward kernel adds the device word ``call_base`` to its baked-in call coordinate, so a
    replay regenerates in its backward the very draws its forward made.  Parameters, gradients and optimizer state live at
    fixed addresses and are updated in place by the replays; ``loss_fn`` typically is
    ``lambda m, out, y: cross_entropy(out, y) + get_kl_loss(m) / batch`` (reference loop:
    examples/main_bayesian_cifar_dnn2bnn.py:402-420)."""

    def __init__(self, model, optimizer, loss_fn, x, y, warmup=3, fused=True, side_wgrad=None):
        """fused=True: every layer's KL term comes out of its forward kernel and is differentiated inside its weight-gradient pass
        (no KL launches, no KL-gradient tensors), and the weight-gradient passes run on a side stream -- a parallel branch of the
        captured graph -- beside the data-gradient chain. fused=False: plain autograd wiring (the checker the tests compare with)."""
        from . import rng
        import os
        if side_wgrad is None:
            side_wgrad = os.environ.get("BT_TRAIN_SIDE_WGRAD", "0") not in ("", "0")
        side_wgrad = bool(side_wgrad and fused)
        self.x, self.y = x.clone(), y.clone()
        self.call_base = torch.zeros(1, dtype=torch.int32, device=x.device)
        B = x.shape[0]

        def run():
            optimizer.zero_grad(set_to_none=True)
            with mc_samples(1, B, 0, collect_kl=False, call_base=self.call_base) as ctx:
                ctx.train_fused, ctx.deferred = fused, ([] if side_wgrad else None)
                # the optimizer step of the previous replay changed every parameter: one check + rebuild per model, in the launch stream
                # (beside a one-sample stem there is nothing to hide the rebuild behind: the forked form measured 6 % slower here)
                sync_model_packs(model, ctx, overlap=False)
                out = model(self.x)
                join_packs(ctx)
            out = out[0] if isinstance(out, tuple) else out
            loss = loss_fn(model, out, self.y)
            loss.backward()
            finish_deferred(ctx)
            for layer in ctx.live_layers:      # a loss that never asked for the KL leaves the tensors (and their graph) behind
                layer._kl_live = None
            optimizer.step()
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture: optimizer state, workspaces, LDS attributes
            for _ in range(max(2, warmup) - 1):
                run()
            c0 = rng.peek_call()
            run()
            self.calls_per_run = rng.peek_call() - c0
        torch.cuda.current_stream().wait_stream(side)
        self.call0 = rng.peek_call()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = run()
        rng.set_call(self.call0)               # capture executes nothing: its coordinates are still unused

    def step(self, x=None, y=None):
        """One optimizer step on (x, y) (default: the captured batch) -> loss (a tensor owned by the graph)."""
        from . import rng
        if x is not None:
            self.x.copy_(x)
        if y is not None:
            self.y.copy_(y)
        c = rng.peek_call()
        d = (c - self.call0) & 0xFFFFFFFF
        self.call_base.fill_(d - (1 << 32) if d >= (1 << 31) else d)
        self.graph.replay()
        rng.set_call(c + self.calls_per_run)
        return self.loss
