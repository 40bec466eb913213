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


def mc_forward(model, x, S, sample0=0, with_kl=True):
    """S MC samples of ``model`` on batch ``x`` -> (logits[S, B, ...], kl or None).

    Semantics of ``for s in range(S): out_s = model(x); kl = get_kl_loss(model)`` of the
    reference's loop; KL does not depend on the sample and is produced by the layers' own
    forward kernels (fused) when ``with_kl``."""
    B = x.shape[0]
    with torch.no_grad(), mc_samples(S, B, sample0, collect_kl=with_kl) as ctx:
        out = model(x)
    if isinstance(out, tuple):   # native Bayesian models return (logits, kl_sum)
        out = out[0]
    out = out.reshape(S, B, *out.shape[1:])
    kl = None
    if with_kl and ctx.kls:
        kl = torch.stack(ctx.kls).sum()
    return out, kl
