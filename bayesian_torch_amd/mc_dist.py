"""MC-sample sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).

Samples are i.i.d. given (mu, rho) -- the reference just loops over them
(examples/main_bayesian_cifar_dnn2bnn.py:542-545) -- so the sample axis shards with no data-path
exchange: parameters and the input batch are replicated, rank r draws global sample ids
[sample0 + r*S_local, ...) (counter-based RNG: results do not depend on the world size), and the
only collective is ONE all-reduce(sum) per batch over a packed fp32 buffer
    [ sum_s softmax(logits_s) : B*C | sum_s H(softmax_s) : B | sum_s logits_s : B*C | KL : 1 ]
(5.6-10.8 KB at CIFAR b128; latency-bound on xGMI, so a single call on the compute stream).
KL is a function of the parameters only and identical on every rank; it rides along pre-divided by
the world size so that the sum reproduces it.
"""
import torch
import torch.distributed as dist


def shard(S_total, rank, world):
    """Contiguous split of S_total samples: -> (first global sample of this rank, local count)."""
    base, rem = divmod(int(S_total), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def pack_size(B, C):
    return 2 * B * C + B + 1


def finish_pack(packed_local, kl, world):
    """Append the KL slot to a [2BC+B] epilogue buffer."""
    k = torch.zeros(1, dtype=packed_local.dtype, device=packed_local.device) if kl is None else (kl.reshape(1) / world)
    return torch.cat([packed_local, k.to(packed_local.dtype)])


def reduce_packed(buf, group=None):
    """The one collective of the path. In place; no-op without an initialised process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def unpack(buf, B, C, S_total):
    """-> dict(mean_prob [B,C], mean_entropy [B], mean_logits [B,C], kl 0-dim)."""
    bc = B * C
    return dict(mean_prob=buf[:bc].reshape(B, C) / S_total, mean_entropy=buf[bc:bc + B] / S_total,
                mean_logits=buf[bc + B:2 * bc + B].reshape(B, C) / S_total, kl=buf[2 * bc + B])


def mc_predict(model, x, S_total, sample0=0, group=None, with_kl=True):
    """MC prediction with the sample axis sharded over the process group (or all local without one)."""
    from . import functional as F
    from .mc import mc_forward
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if S_total < world:      # checked identically on every rank BEFORE any work: all ranks raise together, none enters the collective
        raise RuntimeError(f"S_total={S_total} is smaller than the world size {world}: every rank needs at least one sample")
    first, count = shard(S_total, rank, world)
    B = x.shape[0]
    logits, kl = mc_forward(model, x, count, sample0=sample0 + first, with_kl=with_kl)
    logits = logits.reshape(count, B, -1)
    packed = F.mc_epilogue(logits)
    buf = finish_pack(packed, kl, world)
    reduce_packed(buf, group)
    return unpack(buf, B, logits.shape[-1], S_total)
